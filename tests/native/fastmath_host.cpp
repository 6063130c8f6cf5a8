// Host build of mcsas_amd/csrc/fastmath.h for tests/test_fastmath.py (CPU test infrastructure only).
// The header is written for both sides (MCSAS_HD); on the host the hardware reciprocal seeds are replaced
// by float-precision stand-ins, everything else is the arithmetic the kernels execute (fma = one rounding).
#include "../../mcsas_amd/csrc/fastmath.h"
#include <math.h>

extern "C" {
void fm_sincos_fast(int n, const double *x, double *s, double *c) { for (int i = 0; i < n; ++i) mcsas::sincos_fast(x[i], s + i, c + i); }
void fm_sincos_core(int n, const double *x, double *s, double *c) { for (int i = 0; i < n; ++i) mcsas::sincos_core(x[i], s + i, c + i); }
void fm_j1_fast(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = mcsas::j1_fast(x[i]); }
void fm_j1_core(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = mcsas::j1_core(x[i], 1.0 / x[i]); }
void fm_div_fast(int n, const double *a, const double *b, double *y) { for (int i = 0; i < n; ++i) y[i] = mcsas::div_fast(a[i], b[i]); }
void fm_expm1_neg_fast(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = mcsas::expm1_neg_fast(x[i]); }
void fm_ref_expm1(int n, const double *x, double *hi, double *lo) {
    for (int i = 0; i < n; ++i) { const long double q = expm1l((long double)x[i]); hi[i] = (double)q; lo[i] = (double)(q - (long double)hi[i]); }
}
void fm_rsqrt_fast(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = mcsas::rsqrt_fast(x[i]); }
// references in x87 extended precision (64-bit significand): error 2^-64 relative, far below half an ulp of double
void fm_ref_sincos(int n, const double *x, double *s_hi, double *s_lo, double *c_hi, double *c_lo) {
    for (int i = 0; i < n; ++i) {
        const long double s = sinl((long double)x[i]), c = cosl((long double)x[i]);
        s_hi[i] = (double)s; s_lo[i] = (double)(s - (long double)s_hi[i]);
        c_hi[i] = (double)c; c_lo[i] = (double)(c - (long double)c_hi[i]);
    }
}
void fm_ref_div(int n, const double *a, const double *b, double *hi, double *lo) {
    for (int i = 0; i < n; ++i) { const long double q = (long double)a[i] / (long double)b[i]; hi[i] = (double)q; lo[i] = (double)(q - (long double)hi[i]); }
}
void fm_ref_rsqrt(int n, const double *x, double *hi, double *lo) {
    for (int i = 0; i < n; ++i) { const long double q = 1.0L / sqrtl((long double)x[i]); hi[i] = (double)q; lo[i] = (double)(q - (long double)hi[i]); }
}
int fm_long_double_digits(void) { return __LDBL_MANT_DIG__; }
}
