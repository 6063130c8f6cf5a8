// fastmath.h — fp64 sincos and division tuned for the form-factor inner loops.
//
// The device libm (ocml) sincos carries a Payne-Hanek large-argument path and costs ~110 executed
// instructions per call; q·R on the hot path never exceeds ~1e5, so a 3-term FMA Cody-Waite
// reduction + the fdlibm minimax kernels (|r| <= pi/4, error < 2^-58) is enough: ~34 instructions.
// Measured against extended-precision libm over |x| < 2^20 (tests/test_fastmath.py compiles this header
// for the host: dense sweeps, the doubles next to every multiple of pi/2, the 2^20 hand-off):
//   sincos_fast  <= 1.6 ulp relative (1.55 seen), <= 1.8e-16 absolute, next to multiples of pi/2 included;
//   sincos_core  <= 1.8e-16 absolute everywhere, <= 1.6 ulp where |value| > 1e-9; closer to a multiple of
//                pi/2 its dropped third reduction term (< 2e-27) shows as a relative error (absolute <= 2e-26);
//   j1_fast / j1_core  <= 5e-16 absolute against scipy's Cephes j1;  div_fast <= 1 ulp;  rsqrt_fast <= 1.5 ulp.
// Larger arguments take the libm path.
#pragma once
#ifndef __HIPCC_RTC__
#include <math.h>
#endif

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

// The two kernels of sincos_core without the final quadrant selection: for |x| < 2^20 (the caller guarantees the
// range) x = n pi/2 + r by a two-term Cody-Waite reduction (the third term is < 2e-27 absolute for n < 2^20),
// *s = sin r, *c = cos r (fdlibm kernels, |r| <= pi/4), *q = n.  sin x, cos x = +-s / +-c, swapped when n is odd.
MCSAS_HD void sincos_poly(double x, double *s, double *c, int *q) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00;
    const double P2 = 6.12323399573676603587e-17;
    double n = rint(x * TWO_OVER_PI);
    double r = fma(-n, P1, x);
    r = fma(-n, P2, r);
    *q = (int)n;
    double r2 = r * r;
    double ps = fma(r2, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(r2, ps, 2.75573137070700676789e-06);
    ps = fma(r2, ps, -1.98412698298579493134e-04);
    ps = fma(r2, ps, 8.33333333332248946124e-03);
    ps = fma(r2, ps, -1.66666666666666324348e-01);
    *s = fma(r * r2, ps, r);
    double pc = fma(r2, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(r2, pc, -2.75573143513906633035e-07);
    pc = fma(r2, pc, 2.48015872894767294178e-05);
    pc = fma(r2, pc, -1.38888888888741095749e-03);
    pc = fma(r2, pc, 4.16666666666666019037e-02);
    *c = fma(r2 * r2, pc, fma(-0.5, r2, 1.0));
}

// branch-free core of sincos_fast for |x| < 2^20: sincos_poly + the quadrant selection
MCSAS_HD void sincos_core(double x, double *sn, double *cs) {
    double s, c;
    int q;
    sincos_poly(x, &s, &c, &q);
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    *sn = (q & 2) ? -so : so;
    *cs = ((q + 1) & 2) ? -co : co;
}

// |sin x - x cos x| up to its sign, for callers that square it (the sphere form factor): with x = n pi/2 + r,
//   n = 0, 2 (mod 4):  sin x - x cos x = +-(sin r - x cos r),      n = 1, 3:  +-(cos r + x sin r),
// and fma(-x, -c, -s) == -fma(-x, c, s) bit for bit, so the square equals that of fma(-x, cos x, sin x) taken
// from sincos_core — without its two sign flips and with one selection instead of two.
MCSAS_HD double sin_minus_xcos_abs(double x) {
    double s, c;
    int q;
    sincos_poly(x, &s, &c, &q);
    const double fe = fma(-x, c, s), fo = fma(x, s, c);
    return (q & 1) ? fo : fe;
}

// sin x - x cos x with its sign (callers that add several such terms: the core-shell ellipsoid): the same
// selection, then one sign flip for n = 2, 3 (mod 4).  Bit-identical to fma(-x, cos x, sin x) from sincos_core.
MCSAS_HD double sin_minus_xcos(double x) {
    double s, c;
    int q;
    sincos_poly(x, &s, &c, &q);
    const double fe = fma(-x, c, s), fo = fma(x, s, c);
    const double h = (q & 1) ? fo : fe;
    return (q & 2) ? -h : h;
}

#if defined(__HIPCC__)
#define MCSAS_UNROLL _Pragma("unroll")
#else
#define MCSAS_UNROLL
#endif

// sincos_poly over N independent arguments, written stage by stage so that the N dependent chains are issued
// interleaved (one chain after the other runs at the fp64 pipe's latency, not its issue rate).  The operations
// and their order per element are exactly sincos_poly's: the results are bit-identical.
template <int N>
MCSAS_HD void sincos_poly_n(const double (&x)[N], double (&s)[N], double (&c)[N], int (&q)[N]) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00;
    const double P2 = 6.12323399573676603587e-17;
    double n[N], r[N], r2[N], ps[N], pc[N];
    MCSAS_UNROLL for (int i = 0; i < N; ++i) n[i] = rint(x[i] * TWO_OVER_PI);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) r[i] = fma(-n[i], P1, x[i]);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) r[i] = fma(-n[i], P2, r[i]);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) { q[i] = (int)n[i]; r2[i] = r[i] * r[i]; }
    MCSAS_UNROLL for (int i = 0; i < N; ++i) ps[i] = fma(r2[i], 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) pc[i] = fma(r2[i], -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) ps[i] = fma(r2[i], ps[i], 2.75573137070700676789e-06);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) pc[i] = fma(r2[i], pc[i], -2.75573143513906633035e-07);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) ps[i] = fma(r2[i], ps[i], -1.98412698298579493134e-04);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) pc[i] = fma(r2[i], pc[i], 2.48015872894767294178e-05);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) ps[i] = fma(r2[i], ps[i], 8.33333333332248946124e-03);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) pc[i] = fma(r2[i], pc[i], -1.38888888888741095749e-03);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) ps[i] = fma(r2[i], ps[i], -1.66666666666666324348e-01);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) pc[i] = fma(r2[i], pc[i], 4.16666666666666019037e-02);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) {
        s[i] = fma(r[i] * r2[i], ps[i], r[i]);
        c[i] = fma(r2[i] * r2[i], pc[i], fma(-0.5, r2[i], 1.0));
    }
}

// sincos_core over N independent arguments (sincos_poly_n + the quadrant selections): bit-identical per element
template <int N>
MCSAS_HD void sincos_core_n(const double (&x)[N], double (&sn)[N], double (&cs)[N]) {
    double s[N], c[N];
    int q[N];
    sincos_poly_n<N>(x, s, c, q);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) {
        const double so = (q[i] & 1) ? c[i] : s[i];
        const double co = (q[i] & 1) ? s[i] : c[i];
        sn[i] = (q[i] & 2) ? -so : so;
        cs[i] = ((q[i] + 1) & 2) ? -co : co;
    }
}

// sin_minus_xcos_abs over N independent arguments: bit-identical per element
template <int N>
MCSAS_HD void sin_minus_xcos_abs_n(const double (&x)[N], double (&g)[N]) {
    double s[N], c[N];
    int q[N];
    sincos_poly_n<N>(x, s, c, q);
    MCSAS_UNROLL for (int i = 0; i < N; ++i) {
        const double fe = fma(-x[i], c[i], s[i]), fo = fma(x[i], s[i], c[i]);
        g[i] = (q[i] & 1) ? fo : fe;
    }
}
#undef MCSAS_UNROLL

// Bessel J1, the Cephes algorithm (the one behind scipy.special.j1 that the reference calls,
// cylindersisotropic.py:74, kholodenko.py:43): rational approximation on [0, 5], Hankel asymptotic
// form with rational P, Q beyond.  Same coefficients, so the values track scipy's to ~1e-17 absolute;
// the trigonometric part goes through sincos_fast.  ~30 / ~110 instructions per branch against
// several hundred for the generic device libm j1.
MCSAS_HD double j1_fast(double xin) {
    const double ax = fabs(xin);
    double res;
    if (ax <= 5.0) {
        const double z = ax * ax;
        double n = -8.99971225705559398224E8;
        n = fma(n, z, 4.52228297998194034323E11);
        n = fma(n, z, -7.27494245221818276015E13);
        n = fma(n, z, 3.68295732863852883286E15);
        double d = z + 6.20836478118054335476E2;
        d = fma(d, z, 2.56987256757748830383E5);
        d = fma(d, z, 8.35146791431949253037E7);
        d = fma(d, z, 2.21511595479792499675E10);
        d = fma(d, z, 4.74914122079991414898E12);
        d = fma(d, z, 7.84369607876235854894E14);
        d = fma(d, z, 8.95222336184627338078E16);
        d = fma(d, z, 5.32278620332680085395E18);
        res = (n / d) * ax * (z - 1.46819706421238932572E1) * (z - 4.92184563216946036703E1);
    } else {
        const double w = 5.0 / ax, z = w * w;
        double pn = 7.62125616208173112003E-4;
        pn = fma(pn, z, 7.31397056940917570436E-2);
        pn = fma(pn, z, 1.12719608129684925192E0);
        pn = fma(pn, z, 5.11207951146807644818E0);
        pn = fma(pn, z, 8.42404590141772420927E0);
        pn = fma(pn, z, 5.21451598682361504063E0);
        pn = fma(pn, z, 1.00000000000000000254E0);
        double pd = 5.71323128072548699714E-4;
        pd = fma(pd, z, 6.88455908754495404082E-2);
        pd = fma(pd, z, 1.10514232634061696926E0);
        pd = fma(pd, z, 5.07386386128601488557E0);
        pd = fma(pd, z, 8.39985554327604159757E0);
        pd = fma(pd, z, 5.20982848682361821619E0);
        pd = fma(pd, z, 9.99999999999999997461E-1);
        double qn = 5.10862594750176621635E-2;
        qn = fma(qn, z, 4.98213872951233449420E0);
        qn = fma(qn, z, 7.58238284132545283818E1);
        qn = fma(qn, z, 3.66779609360150777800E2);
        qn = fma(qn, z, 7.10856304998926107277E2);
        qn = fma(qn, z, 5.97489612400613639965E2);
        qn = fma(qn, z, 2.11688757100572135698E2);
        qn = fma(qn, z, 2.52070205858023719784E1);
        double qd = z + 7.42373277035675149943E1;
        qd = fma(qd, z, 1.05644886038262816351E3);
        qd = fma(qd, z, 4.98641058337653607651E3);
        qd = fma(qd, z, 9.56231892404756170795E3);
        qd = fma(qd, z, 7.99704160447350683650E3);
        qd = fma(qd, z, 2.82619278517639096600E3);
        qd = fma(qd, z, 3.36093607810698293419E2);
        double sn, cs;
        sincos_fast(ax - 2.35619449019234492885, &sn, &cs);
        const double p = (pn / pd) * cs - w * (qn / qd) * sn;
        res = p * 0.79788456080286535588 / sqrt(ax);
    }
    return xin < 0. ? -res : res;
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

// 1/sqrt(x) for normal-range x > 0: hardware seed (~2^-26) + one third-order correction
MCSAS_HD double rsqrt_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
#else
    const double y = 1.0 / sqrt((double)(float)x);     // host stand-in for the hardware seed
#endif
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// J1(x) for 0 < x < 2^20 with 1/x supplied by the caller (the integration loops have it as a product
// of two precomputed reciprocals): same Cephes rationals as j1_fast, one division per branch, the
// branch-free sincos core and the hardware reciprocal square root.
// exp(x) - 1 for x <= 0 (the worm-like chain quadrature calls it with x = -z, 0 < z <= 2, once per point):
// x = k ln2 + r with |r| <= ln2 / 2 (two-term Cody-Waite), expm1(r) by its Taylor polynomial to r^13 (next term
// < 5e-18 relative), and e^x - 1 = (2^k - 1) + 2^k expm1(r) — both summands are exact scalings / differences, so
// there is no cancellation for any k <= 0 (k = 0: the polynomial itself).  Branch-free, <= 2 ulp (tests/test_fastmath.py).
MCSAS_HD double expm1_neg_fast(double x) {
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(-k, 6.93147180369123816490e-01, x);        // ln2 hi (fdlibm split)
    r = fma(-k, 1.90821492927058770002e-10, r);               // ln2 lo
    double p = 1.6059043836821613e-10;                        // 1/13!
    p = fma(p, r, 2.08767569878680990e-09);                   // 1/12!
    p = fma(p, r, 2.50521083854417188e-08);                   // 1/11!
    p = fma(p, r, 2.75573192239858907e-07);                   // 1/10!
    p = fma(p, r, 2.75573192239858907e-06);                   // 1/9!
    p = fma(p, r, 2.48015873015873016e-05);                   // 1/8!
    p = fma(p, r, 1.98412698412698413e-04);                   // 1/7!
    p = fma(p, r, 1.38888888888888894e-03);                   // 1/6!
    p = fma(p, r, 8.33333333333333322e-03);                   // 1/5!
    p = fma(p, r, 4.16666666666666644e-02);                   // 1/4!
    p = fma(p, r, 1.66666666666666657e-01);                   // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p * r, r, r);                                     // r + r² (1/2 + r/6 + ...)
    const double e = ldexp(1.0, (int)k);                      // 2^k, exact (k >= -1075: 0 below that, result -1)
    return fma(e, p, e - 1.0);
}

// the two ranges of j1_core as functions of their own: a caller that knows (wave-uniformly) which range a group of its
// arguments is in can run several of them interleaved in one basic block
MCSAS_HD double j1_core_small(double x) {                   // x <= 5
    const double z = x * x;
    double n = -8.99971225705559398224E8;
    n = fma(n, z, 4.52228297998194034323E11);
    n = fma(n, z, -7.27494245221818276015E13);
    n = fma(n, z, 3.68295732863852883286E15);
    double d = z + 6.20836478118054335476E2;
    d = fma(d, z, 2.56987256757748830383E5);
    d = fma(d, z, 8.35146791431949253037E7);
    d = fma(d, z, 2.21511595479792499675E10);
    d = fma(d, z, 4.74914122079991414898E12);
    d = fma(d, z, 7.84369607876235854894E14);
    d = fma(d, z, 8.95222336184627338078E16);
    d = fma(d, z, 5.32278620332680085395E18);
    return div_fast(n * x * (z - 1.46819706421238932572E1) * (z - 4.92184563216946036703E1), d);
}
MCSAS_HD double j1_core_large(double x, double invx) {      // x > 5
    const double w = 5.0 * invx, z = w * w;
    double pn = 7.62125616208173112003E-4;
    pn = fma(pn, z, 7.31397056940917570436E-2);
    pn = fma(pn, z, 1.12719608129684925192E0);
    pn = fma(pn, z, 5.11207951146807644818E0);
    pn = fma(pn, z, 8.42404590141772420927E0);
    pn = fma(pn, z, 5.21451598682361504063E0);
    pn = fma(pn, z, 1.00000000000000000254E0);
    double pd = 5.71323128072548699714E-4;
    pd = fma(pd, z, 6.88455908754495404082E-2);
    pd = fma(pd, z, 1.10514232634061696926E0);
    pd = fma(pd, z, 5.07386386128601488557E0);
    pd = fma(pd, z, 8.39985554327604159757E0);
    pd = fma(pd, z, 5.20982848682361821619E0);
    pd = fma(pd, z, 9.99999999999999997461E-1);
    double qn = 5.10862594750176621635E-2;
    qn = fma(qn, z, 4.98213872951233449420E0);
    qn = fma(qn, z, 7.58238284132545283818E1);
    qn = fma(qn, z, 3.66779609360150777800E2);
    qn = fma(qn, z, 7.10856304998926107277E2);
    qn = fma(qn, z, 5.97489612400613639965E2);
    qn = fma(qn, z, 2.11688757100572135698E2);
    qn = fma(qn, z, 2.52070205858023719784E1);
    double qd = z + 7.42373277035675149943E1;
    qd = fma(qd, z, 1.05644886038262816351E3);
    qd = fma(qd, z, 4.98641058337653607651E3);
    qd = fma(qd, z, 9.56231892404756170795E3);
    qd = fma(qd, z, 7.99704160447350683650E3);
    qd = fma(qd, z, 2.82619278517639096600E3);
    qd = fma(qd, z, 3.36093607810698293419E2);
    double sn, cs;
    sincos_core(x - 2.35619449019234492885, &sn, &cs);
    // (pn/pd) cs - w (qn/qd) sn over one common denominator
    const double num = (pn * qd) * cs - (w * (qn * pd)) * sn;
    return div_fast(num, pd * qd) * (0.79788456080286535588 * rsqrt_fast(x));
}
MCSAS_HD double j1_core(double x, double invx) {
    if (x <= 5.0) return j1_core_small(x);
    return j1_core_large(x, invx);
}

}  // namespace mcsas
