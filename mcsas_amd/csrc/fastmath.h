// fastmath.h — fp64 sincos and division tuned for the form-factor inner loops.
//
// The device libm (ocml) sincos carries a Payne-Hanek large-argument path and costs ~110 executed
// instructions per call; q·R on the hot path never exceeds ~1e5, so a 3-term FMA Cody-Waite
// reduction + the fdlibm minimax kernels (|r| <= pi/4, error < 2^-58) is enough: ~34 instructions,
// measured error <= 1.0 ulp over |x| < 2^20 (tests/test_fastmath.py compiles this header for the
// host and checks it against long-double libm).  Larger arguments take the libm path.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MCSAS_HD __host__ __device__ __forceinline__
#else
#define MCSAS_HD static inline
#endif

namespace mcsas {

MCSAS_HD void sincos_fast(double x, double *sn, double *cs) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00;     // pi/2 rounded to 53 bits
    const double P2 = 6.12323399573676603587e-17;     // pi/2 - P1, rounded
    const double P3 = -1.49738490485916983294e-33;    // pi/2 - P1 - P2, rounded
    if (!(fabs(x) < 1048576.0)) {                     // 2^20; also catches NaN/Inf
#if defined(__HIP_DEVICE_COMPILE__)
        sincos(x, sn, cs);
#else
        *sn = sin(x); *cs = cos(x);
#endif
        return;
    }
    double n = rint(x * TWO_OVER_PI);
    double r = fma(-n, P1, x);
    r = fma(-n, P2, r);
    r = fma(-n, P3, r);
    int q = (int)n;
    double r2 = r * r;
    // fdlibm __kernel_sin / __kernel_cos coefficients
    double ps = fma(r2, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(r2, ps, 2.75573137070700676789e-06);
    ps = fma(r2, ps, -1.98412698298579493134e-04);
    ps = fma(r2, ps, 8.33333333332248946124e-03);
    ps = fma(r2, ps, -1.66666666666666324348e-01);
    double s = fma(r * r2, ps, r);
    double pc = fma(r2, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(r2, pc, -2.75573143513906633035e-07);
    pc = fma(r2, pc, 2.48015872894767294178e-05);
    pc = fma(r2, pc, -1.38888888888741095749e-03);
    pc = fma(r2, pc, 4.16666666666666019037e-02);
    double c = fma(r2 * r2, pc, fma(-0.5, r2, 1.0));
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    *sn = (q & 2) ? -so : so;
    *cs = ((q + 1) & 2) ? -co : co;
}

// branch-free core of sincos_fast for |x| < 2^20 (the caller guarantees the range): two-term
// Cody-Waite reduction (the third term is < 2e-27 absolute for n < 2^20) + the same kernels
MCSAS_HD void sincos_core(double x, double *sn, double *cs) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00;
    const double P2 = 6.12323399573676603587e-17;
    double n = rint(x * TWO_OVER_PI);
    double r = fma(-n, P1, x);
    r = fma(-n, P2, r);
    int q = (int)n;
    double r2 = r * r;
    double ps = fma(r2, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(r2, ps, 2.75573137070700676789e-06);
    ps = fma(r2, ps, -1.98412698298579493134e-04);
    ps = fma(r2, ps, 8.33333333332248946124e-03);
    ps = fma(r2, ps, -1.66666666666666324348e-01);
    double s = fma(r * r2, ps, r);
    double pc = fma(r2, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(r2, pc, -2.75573143513906633035e-07);
    pc = fma(r2, pc, 2.48015872894767294178e-05);
    pc = fma(r2, pc, -1.38888888888741095749e-03);
    pc = fma(r2, pc, 4.16666666666666019037e-02);
    double c = fma(r2 * r2, pc, fma(-0.5, r2, 1.0));
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    *sn = (q & 2) ? -so : so;
    *cs = ((q + 1) & 2) ? -co : co;
}

// a / b for normal-range operands (no subnormal / overflow scaling): reciprocal seed, two Newton
// steps, one residual correction; <= 1 ulp
MCSAS_HD double div_fast(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(b);
#else
    double y = 1.0 / (double)(float)b;                 // host stand-in for the hardware seed
#endif
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    double qv = a * y;
    double r = fma(-b, qv, a);
    return fma(r, y, qv);
}

}  // namespace mcsas
