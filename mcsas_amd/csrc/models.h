// models.h — device form factors: one struct per ScatteringModel subclass on the hot path.
// Each `Contrib<M>` holds the per-contribution scalars (what volume()/absVolume()/surface()/
// weight() return for one parameter row) and evaluates I(q) = F(q)^2 * volume()^(2c), i.e.
// SASModel.calcIntensity()[0] (bases/model/sasmodel.py:46-79, smearing off).
#pragma once
#include "device_util.h"
#include "fastmath.h"
#ifdef __HIPCC_RTC__
#include "mcsas_hip.h"               // (run-time compiler: the headers come from memory, by plain name)
#else
#include "../../include/mcsas_hip.h"
#endif

#ifndef MCSAS_ROW_GROUP
#define MCSAS_ROW_GROUP 4       // sphere rows: q slots per lane evaluated in one interleaved group
#endif

namespace mcsas {

// kernel-argument view of the model half of mcsas_problem
struct ModelArgs {
    int32_t model_id;
    int32_t n_active;
    int32_t active_index[MCSAS_MAX_ACTIVE];
    double  params[MCSAS_MAX_PARAMS];
    double  clip_lo[MCSAS_MAX_ACTIVE];
    double  clip_hi[MCSAS_MAX_ACTIVE];
    double  comp_exp;
    int32_t int_div;       // orientation / quadrature points K (1 for the sphere)
    int32_t use_rowtab;    // per-row orientation tables in LDS (host: table_doubles_host), see RowEval
    double  qmax;          // largest q of the data set: bounds q*R for the branch-free sincos
    // beam-profile smearing (sasmodel.py:56-73), canSmear models only; smear_nk = 0: off
    int32_t smear_nk, smear_stride;    // integration points per q; row stride of locs_t (>= nq)
    const double *smear_locs_t;        // [smear_nk][smear_stride]: evaluation points, transposed for coalescing
    const double *smear_cw;            // [smear_nk]: 2 * trapezoid coefficient * beam-profile weight
};

// read-only per-block tables in LDS
struct QTables {
    const double *q;       // [qpad]
    const double *q3inv;   // [qpad] 1/q^3
    const double *tab;     // orientation table of the model
    double *rowtab;        // this wave's scratch for the per-row table (ROWTAB*K doubles) or null
    const double *locs_t, *cw;   // smearing (ModelArgs), smear_nk = 0: off
    int smear_nk, smear_stride;
};

// One specialisation per model: everything the library knows about a model is in it.
//   ROWTAB         doubles per orientation point of the per-row table (0: the model has none)
//   INT_DIV_PARAM  index of the quadrature point count in params[], -1 if the model has no orientation / contour integral
//   ROW_CLASS      0: a row costs a few hundred instructions (no integral) — the pipeline keeps its `d` rows in LDS and
//                  re-evaluates stale rows instead of storing new ones; 1: a row costs an integral; 2: ... whose cost also
//                  varies with the parameter set (row_cost below deals such rows out by predicted cost)
//   CAN_SMEAR      the reference's canSmear flag (sasmodel.py:56-60)
//   table_doubles(K), fill_table()   shared orientation table in LDS;  prepare() / bcast() / intensity()   a contribution
// Adding a built-in model: a specialisation here, its id in include/mcsas_hip.h and in model_list.h.
template <int M> struct Contrib;

// smeared intensity at data point i: sum_m cw[m] * I(locs[i][m])  (I = F^2 w, so this is
// 2 trapz(F^2 w weights, x = qOffset), sasmodel.py:72-73)
template <int M>
__device__ __forceinline__ double smeared_intensity(const Contrib<M> &c, const double *locs_t, const double *cw,
                                                    int nk, int stride, int i, const double *tab) {
    double acc = 0.;
    for (int m = 0; m < nk; ++m) acc = fma(glb(cw)[m], c.intensity(glb(locs_t)[(size_t)m * stride + i], tab), acc);
    return acc;
}

// full parameter vector for one contribution: active columns from `row`, clipped into their
// valueRange as Parameter.setValue does (bases/algorithm/parameter.py:405-414,489-495)
__device__ __forceinline__ void full_params(const ModelArgs &a, const double *row, double *p) {
#pragma unroll
    for (int i = 0; i < MCSAS_MAX_PARAMS; ++i) p[i] = a.params[i];
#pragma unroll
    for (int c = 0; c < MCSAS_MAX_ACTIVE; ++c)
        if (c < a.n_active) {
            double v = fmin(fmax(row[c], a.clip_lo[c]), a.clip_hi[c]);
#pragma unroll
            for (int i = 0; i < MCSAS_MAX_PARAMS; ++i)
                if (a.active_index[c] == i) p[i] = v;
        }
}

constexpr double PI = 3.141592653589793;

// ---------------------------------------------------------------------------------- Sphere
// models/sphere.py:32-63
template <> struct Contrib<MCSAS_MODEL_SPHERE> {
    static constexpr int ROWTAB = 0;
    static constexpr int INT_DIV_PARAM = -1, ROW_CLASS = 0;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = true;
    double r, v, w, s, invr3;
    int fast;              // q*r < 2^20 for every q of the data set: branch-free sincos is valid
    static __host__ __device__ __forceinline__ int table_doubles(int) { return 0; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &, double *, int, int) {}
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        r = p[0];
        double vol = (PI * 4. / 3.) * (r * r * r);   // sphere.py:44
        v = vol * (p[1] * p[1]);                      // sphere.py:53
        s = 4. * PI * r * r;                          // sphere.py:37
        w = pow(vol, 2. * a.comp_exp);                // sasmodel.py:44
        invr3 = 1.0 / (r * r * r);
        fast = (a.qmax * r < 1048576.0) && (r > 0.);
    }
    // copy of lane `lane`'s contribution into wave-uniform registers
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.r = readlane_f64(r, lane); o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.;
        o.invr3 = readlane_f64(invr3, lane); o.fast = __builtin_amdgcn_readlane(fast, lane);
        return o;
    }
    // branch-free evaluation, valid when `fast`: 1/x^3 from the two precomputed reciprocals; sin x - x cos x up to its
    // sign (it is squared: fastmath.h, sin_minus_xcos_abs)
    __device__ __forceinline__ double intensity_fast(double q, double q3inv) const {
        const double x = q * r;
        const double f = (3. * sin_minus_xcos_abs(x)) * (q3inv * invr3);
        return f * f * w;
    }
    // the same for N points at once, the N chains interleaved (fastmath.h: sincos_poly_n); bit-identical per point
    template <int N>
    __device__ __forceinline__ void intensity_fast_n(const double (&q)[N], const double (&q3inv)[N], double (&out)[N]) const {
        double x[N], g[N];
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = q[i] * r;
        sin_minus_xcos_abs_n<N>(x, g);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double f = (3. * g[i]) * (q3inv[i] * invr3);
            out[i] = f * f * w;
        }
    }
    __device__ __forceinline__ double intensity(double q, const double *) const {
        double x = q * r, sn, cs;
        sincos_fast(x, &sn, &cs);
        double f = div_fast(3. * fma(-x, cs, sn), x * x * x);  // sphere.py:62
        return f * f * w;
    }
};

// ---------------------------------------------------------------------------------- Cylinders
// models/cylindersisotropic.py:50-101.  table: x_k (ends replaced by 0.5, :60-61) and sqrt(1-x_k^2)
template <> struct Contrib<MCSAS_MODEL_CYL_ISO> {
    static constexpr int ROWTAB = 4;       // per orientation k: r sqrt(1-x^2), hl x, 1/(2 r sqrt(1-x^2) hl x), 1/(r sqrt(1-x^2))
    static constexpr int INT_DIV_PARAM = 4, ROW_CLASS = 2;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = false;
    double r, hl, v, w, s, step;
    int K, fast;
    static __host__ __device__ __forceinline__ int table_doubles(int K) { return 2 * K; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &a, double *tab, int tid, int nt) {
        int K = a.int_div;
        double step = 1.0 / (double)(K - 1);
        for (int k = tid; k < K; k += nt) {
            double x = (k == K - 1) ? 1.0 : (double)k * step;   // numpy.linspace(0, 1, K)
            if (k == 0 || k == K - 1) x = 0.5;
            tab[k] = x;
            tab[K + k] = sqrt(1. - x * x);
        }
    }
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        r = p[0];
        hl = (p[1] != 0.0) ? r * p[3] : 0.5 * p[2];   // :65-68 useAspect ? radius*aspect : length/2
        double vol = PI * (r * r) * (hl * 2.);        // :97
        v = vol * (p[5] * p[5]);                      // :101
        s = 0.;
        w = pow(vol, 2. * a.comp_exp);
        K = a.int_div;
        step = 1.0 / (double)(K - 1);
        fast = (a.qmax * fmax(r, hl) < 1048576.0) && (r > 0.) && (hl > 0.);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.r = readlane_f64(r, lane); o.hl = readlane_f64(hl, lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.; o.step = step; o.K = K;
        o.fast = __builtin_amdgcn_readlane(fast, lane);
        return o;
    }
    // q-independent factors of the integrand, once per row (the wave's lanes share the K points)
    __device__ __forceinline__ void fill_rowtab(const double *tab, double *rt, int lane) const {
        for (int k = lane; k < K; k += WAVE) {
            const double A = r * tab[K + k], B = hl * tab[k];
            rt[4 * k + 0] = A; rt[4 * k + 1] = B;
            rt[4 * k + 2] = 1.0 / (2. * A * B); rt[4 * k + 3] = 1.0 / A;
        }
    }
    // same integral as intensity(): f_k = J1(q A_k) sin(q B_k) / (q^2 2 A_k B_k), without divisions
    // or large-argument branches in the loop (valid when `fast`).  One q slot of the lane at a time, the orientations in
    // groups of seven: J1 changes formula at x = 5, and with x = q A_k varying slowly along k a whole group — all 64 lanes,
    // whose q lie within a factor two of each other — is nearly always on one side, so the side is chosen once per
    // group (wave-uniform) and its evaluations run interleaved in one basic block.  (The earlier form — all q slots
    // of the lane inside the k loop, J1's branch per evaluation — reached 0.63 of the fp64 issue rate: eight
    // data-dependent branches per k cut the loop body into blocks that cannot overlap.)  The sum over k is taken in the
    // same order as before: same bits.
    template <int QPL>
    __device__ __forceinline__ void rows_rt(const QTables &t, int lane, double (&out)[QPL]) const {
#pragma unroll
        for (int j = 0; j < QPL; ++j) out[j] = 0.;
        rows_rt_each<QPL>(t, lane, [&](int js, double o) {
#pragma unroll
            for (int j = 0; j < QPL; ++j) out[j] = (j == js) ? o : out[j];
        });
    }
    // the same, each q slot handed to `sink(slot, value)` as soon as it is finished — end columns, trapezoid, weight inside the
    // slot's own iteration — so that NO per-slot array lives across the slot loop (q, 1/q and the sums of all slots used to: spilled
    // and reloaded once per slot; a caller that stores the value at once keeps the evaluation's registers to the evaluation)
    template <int QPL, class Sink>
    __device__ __forceinline__ void rows_rt_each(const QTables &t, int lane, Sink &&sink) const {
        const double *rt = t.rowtab;
        constexpr int G = 7;                                  // (K = 100: 98 interior orientations = 14 groups; groups of 4: 2 % slower)
#pragma nounroll
        for (int js = 0; js < QPL; ++js) {
            const double qj = t.q[lane + WAVE * js];
            const double iqj = (qj * qj) * t.q3inv[lane + WAVE * js];
            double a = 0.;
            int k = 1;
            for (; k + G <= K - 1; k += G) {
                double A[G], B[G], C[G], iA[G], x[G], jv[G], sl[G], cl[G];
                bool all_gt = true, all_le = true;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    A[g] = rt[4 * (k + g)]; B[g] = rt[4 * (k + g) + 1]; C[g] = rt[4 * (k + g) + 2]; iA[g] = rt[4 * (k + g) + 3];
                    x[g] = qj * A[g];
                    all_gt = all_gt && x[g] > 5.0; all_le = all_le && x[g] <= 5.0;
                }
                if (__all(all_gt)) {
#pragma unroll
                    for (int g = 0; g < G; ++g) { sincos_core(qj * B[g], &sl[g], &cl[g]); jv[g] = j1_core_large(x[g], iqj * iA[g]); }
                } else if (__all(all_le)) {
#pragma unroll
                    for (int g = 0; g < G; ++g) { sincos_core(qj * B[g], &sl[g], &cl[g]); jv[g] = j1_core_small(x[g]); }
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) { sincos_core(qj * B[g], &sl[g], &cl[g]); jv[g] = j1_core(x[g], iqj * iA[g]); }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double gg = (jv[g] * sl[g]) * C[g];
                    a = fma(gg, gg, a);
                }
            }
            for (; k < K - 1; ++k) {
                const double A = rt[4 * k], B = rt[4 * k + 1], C = rt[4 * k + 2], iA = rt[4 * k + 3];
                double s1, c1;
                sincos_core(qj * B, &s1, &c1);
                const double gg = (j1_core(qj * A, iqj * iA) * s1) * C;
                a = fma(gg, gg, a);
            }
            const double qr = qj * r, qh = qj * hl;
            const double f0 = 0.5 * (j1_fast(qr) / qr);          // end columns: analytic limits (:79-82)
            double sq, cq;
            sincos_core(qh, &sq, &cq);
            const double fl = sq / qh;
            const double iq2 = iqj * iqj;
            const double trapz = (0.5 * step) * (f0 * f0 + fl * fl + 2. * (a * (iq2 * iq2)));
            sink(js, (16. * trapz) * w);
        }
    }
    __device__ __forceinline__ double intensity(double q, const double *tab) const {
        // end columns: analytic limits (:79-82)
        double qr = q * r, qh = q * hl;
        double f0 = 0.5 * (j1_fast(qr) / qr);
        double sq, cq;
        sincos_fast(qh, &sq, &cq);
        double fl = sq / qh;
        double prev = f0 * f0, acc = 0.;
        for (int k = 1; k < K - 1; ++k) {
            double qrs = q * (r * tab[K + k]);
            double qlx = q * (2. * hl * tab[k]);
            double sl, cl;
            sincos_fast(qlx / 2., &sl, &cl);
            double f = (j1_fast(qrs) * sl) / (qrs * qlx);
            double cur = f * f;
            acc += (cur + prev);
            prev = cur;
        }
        acc += (fl * fl + prev);
        double ff = sqrt(16. * (acc * step * 0.5));   // numpy.trapz(fsplit**2, dx=step) (:90)
        return ff * ff * w;
    }
};

// ---------------------------------------------------------------------------------- Core-shell ellipsoid
// models/ellipsoidalcoreshell.py:59-97.  table: mu_k^2 and 1-mu_k^2
template <> struct Contrib<MCSAS_MODEL_ELL_CS> {
    static constexpr int ROWTAB = 4;       // per orientation k: R_core, R_total, 3 c1 / R_core^3, 3 c2 / R_total^3
    static constexpr int INT_DIV_PARAM = 6, ROW_CLASS = 1;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = false;
    double a2, b2, at2, bt2, c1, c2, v, w, s, invK;
    int K, fast;
    static __host__ __device__ __forceinline__ int table_doubles(int K) { return 2 * K; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &a, double *tab, int tid, int nt) {
        int K = a.int_div;
        double step = 1.0 / (double)(K - 1);
        for (int k = tid; k < K; k += nt) {
            double mu = (k == K - 1) ? 1.0 : (double)k * step;
            tab[k] = mu * mu;
            tab[K + k] = 1. - mu * mu;
        }
    }
    __device__ __forceinline__ void prepare(const ModelArgs &ar, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(ar, row, p);
        double a = p[0], b = p[1], t = p[2];
        double vc = 4. / 3. * PI * a * (b * b);
        double vt = 4. / 3. * PI * (a + t) * ((b + t) * (b + t));
        double vr = vc / vt;
        c1 = (p[3] - p[4]) * vr;          // (eta_c - eta_s) * vRatio
        c2 = (p[4] - p[5]) * 1.;          // (eta_s - eta_sol)
        a2 = a * a; b2 = b * b; at2 = (a + t) * (a + t); bt2 = (b + t) * (b + t);
        v = vt; s = 0.;                   // volume() == absVolume() (:92-97)
        w = pow(vt, 2. * ar.comp_exp);
        K = ar.int_div;
        invK = 1.0 / (double)K;
        fast = (ar.qmax * ar.qmax * fmax(at2, bt2) < 1048576.0 * 1048576.0) && (a2 > 0.) && (b2 > 0.);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o;
        o.a2 = readlane_f64(a2, lane); o.b2 = readlane_f64(b2, lane);
        o.at2 = readlane_f64(at2, lane); o.bt2 = readlane_f64(bt2, lane);
        o.c1 = readlane_f64(c1, lane); o.c2 = readlane_f64(c2, lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.; o.invK = invK; o.K = K;
        o.fast = __builtin_amdgcn_readlane(fast, lane);
        return o;
    }
    __device__ __forceinline__ void fill_rowtab(const double *tab, double *rt, int lane) const {
        for (int k = lane; k < K; k += WAVE) {
            const double m2 = tab[k], n2 = tab[K + k];
            const double Rc = sqrt(a2 * m2 + b2 * n2), Rt = sqrt(at2 * m2 + bt2 * n2);
            rt[4 * k + 0] = Rc; rt[4 * k + 1] = Rt;
            rt[4 * k + 2] = (3. * c1) / (Rc * Rc * Rc); rt[4 * k + 3] = (3. * c2) / (Rt * Rt * Rt);
        }
    }
    // f_k = [C1_k (sin xc - xc cos xc) + C2_k (sin xt - xt cos xt)] / q^3 with x = q R_k (valid when `fast`)
    template <int QPL>
    __device__ __forceinline__ void rows_rt(const QTables &t, int lane, double (&out)[QPL]) const {
        double q[QPL], acc[QPL];
#pragma unroll
        for (int j = 0; j < QPL; ++j) { q[j] = t.q[lane + WAVE * j]; acc[j] = 0.; }
        const double *rt = t.rowtab;
        for (int k = 0; k < K; ++k) {
            const double Rc = rt[4 * k], Rt = rt[4 * k + 1], C1 = rt[4 * k + 2], C2 = rt[4 * k + 3];
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const double xc = q[j] * Rc, xt = q[j] * Rt;
                const double g = fma(C1, sin_minus_xcos(xc), C2 * sin_minus_xcos(xt));
                acc[j] = fma(g, g, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const double q3 = t.q3inv[lane + WAVE * j];
            out[j] = ((acc[j] * (q3 * q3)) * invK) * w;
        }
    }
    __device__ __forceinline__ double intensity(double q, const double *tab) const {
        double acc = 0.;
        for (int k = 0; k < K; ++k) {
            double m2 = tab[k], n2 = tab[K + k];
            double xc = q * sqrt(a2 * m2 + b2 * n2);
            double xt = q * sqrt(at2 * m2 + bt2 * n2);
            double sc, cc, st, ct;
            sincos_fast(xc, &sc, &cc);
            sincos_fast(xt, &st, &ct);
            double jc = div_fast(sc - xc * cc, xc * xc);
            double jt = div_fast(st - xt * ct, xt * xt);
            double f = c1 * div_fast(3. * jc, xc) + c2 * div_fast(3. * jt, xt);
            acc += f * f;
        }
        double ff = sqrt(acc * invK);     // numpy.sqrt(numpy.mean(fsplit**2, axis=1))
        return ff * ff * w;
    }
};

// ---------------------------------------------------------------------------------- Kholodenko worm
// models/kholodenko.py:16-94.  F = sqrt(P0) * 2 J1(qr)/(qr),  P0 = (2/x) ∫_0^x f(z) (1 - z/x) dz,
// x = 3 L / l_k, f(z) = sinh(e z)/(e sinh z) for q < 3/l_k (e = sqrt(1 - (q l_k/3)^2)),
// sin(F z)/(F sinh z) for q > 3/l_k (F = sqrt((q l_k/3)^2 - 1)), z/sinh z at equality.
// The reference integrates with QUADPACK (epsrel 1e-10).  Here: composite 16-point Gauss-Legendre on
// a fixed panel scheme with a known error bound —
//   smooth branch: panels [0,1],[1,2],[2,4],[4,8],...; the integrand is exp(-(1-e) z) times a factor
//     that is analytic with decay rate >= 2, so (rate * width / 2)^32 / 32! <= 1e-7 of a panel whose own
//     weight is already <= exp(-8); truncated where exp(-(1-e) z) < 6e-19;
//   oscillatory branch: uniform panels of width min(1, 8/F) on [0, 2] (phase advance <= 8 rad per panel
//     => GL-16 error (4)^32/32! = 7e-17), the rest of the range in closed form term by term of
//     1/sinh z = 2 sum exp(-(2m+1) z).
// exp(-z)/(1 - exp(-2z)) replaces 1/(2 sinh z) so nothing overflows for x > 710 (the reference would).
// table: GL-16 nodes then weights on [-1, 1].
template <> struct Contrib<MCSAS_MODEL_KHOLODENKO> {
    static constexpr int ROWTAB = 0;
    static constexpr int INT_DIV_PARAM = -1, ROW_CLASS = 2;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = false;
    double r, lk, x, ratio, v, w, s;
    static __host__ __device__ __forceinline__ int table_doubles(int) { return 32; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &, double *tab, int tid, int) {
        const double nd[8] = {9.50125098376374544e-02, 2.81603550779258915e-01, 4.58016777657227370e-01,
                              6.17876244402643771e-01, 7.55404408355002999e-01, 8.65631202387831755e-01,
                              9.44575023073232600e-01, 9.89400934991649939e-01};
        const double wt[8] = {1.89450610455068585e-01, 1.82603415044923612e-01, 1.69156519395002619e-01,
                              1.49595988816576764e-01, 1.24628971255534030e-01, 9.51585116824925914e-02,
                              6.22535239386477063e-02, 2.71524594117540374e-02};
        if (tid < 16) {
            const int i = tid < 8 ? 7 - tid : tid - 8;
            tab[tid] = tid < 8 ? -nd[i] : nd[i];
            tab[16 + tid] = wt[i];
        }
    }
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        r = p[0]; lk = p[1];
        const double lc = p[2];
        x = 3. * lc / lk;                            // kholodenko.py:86
        ratio = 3.0 / lk;                            // :19
        const double vol = PI * lc * (r * r);        // :92-94
        v = vol; s = 0.;
        w = pow(vol, 2. * a.comp_exp);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.r = readlane_f64(r, lane); o.lk = readlane_f64(lk, lane); o.x = readlane_f64(x, lane);
        o.ratio = readlane_f64(ratio, lane); o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.;
        return o;
    }
    // cD phi1(t) - D2x phi2(t), phi1(t) = (1 - exp(-t))/t, phi2(t) = (1 - (1+t) exp(-t))/t² (series below t = 0.5)
    static __device__ __forceinline__ double kho_tail(double t, double cD, double D2x) {
        const double em1 = -expm1(-t);                         // 1 - exp(-t)
        const double p1 = em1 / t;
        double p2;
        if (t < 0.5) {                                         // sum_k (-1)^k (k+1)/(k+2)! t^k
            p2 = -1.07060292245477428e-11;                      // k = 13
            p2 = fma(p2, t, 1.49119692770486430e-10);
            p2 = fma(p2, t, -1.92708526041859370e-09);
            p2 = fma(p2, t, 2.29644326866549102e-08);
            p2 = fma(p2, t, -2.50521083854417176e-07);
            p2 = fma(p2, t, 2.48015873015873016e-06);
            p2 = fma(p2, t, -2.20458553791887140e-05);
            p2 = fma(p2, t, 1.73611111111111118e-04);
            p2 = fma(p2, t, -1.19047619047619058e-03);
            p2 = fma(p2, t, 6.94444444444444406e-03);
            p2 = fma(p2, t, -3.33333333333333329e-02);
            p2 = fma(p2, t, 1.25000000000000000e-01);
            p2 = fma(p2, t, -3.33333333333333315e-01);
            p2 = fma(p2, t, 5.00000000000000000e-01);
        } else {
            p2 = (em1 - t * (1.0 - em1)) / (t * t);
        }
        return cD * p1 - D2x * p2;
    }
    __device__ __forceinline__ double intensity(double q, const double *tab) const {
        const double invx = 1.0 / x;
        double acc = 0.;
        if (q > ratio) {                                                  // oscillatory branch (:24-26)
            const double F = sqrt(q * q * lk * lk / 9. - 1.0);
            const double invF = 1.0 / F;
            // [0, z0]: composite GL-16, panels of phase advance <= 8 rad
            const double z0 = fmin(x, 2.0);
            const int n = (int)ceil(z0 / fmin(1.0, 8.0 / F));
            const double h = z0 / (double)n, hw = 0.5 * h;
            // (F z < 2^20 for every lane of the wave — in practice always —: the branch-free sincos, 1.8e-16 absolute, lets the
            // unrolled points overlap)
            const bool small_arg = __all(F * z0 < 1048576.0);
            for (int pnl = 0; pnl < n; ++pnl) {
                const double mid = ((double)pnl + 0.5) * h;
                double pa = 0.;
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const double z = fma(hw, tab[i], mid);
                    double sn, cs;
                    if (small_arg) sincos_core(F * z, &sn, &cs);
                    else sincos_fast(F * z, &sn, &cs);
                    // 1/sinh z = 2 e^{-z} / (1 - e^{-2z}) from ONE expm1: u = e^{-z} - 1, e^{-z} = 1 + u, 1 - e^{-2z} = -u (2 + u)
                    // (no cancellation for small z; one transcendental call per point instead of two)
                    const double u = expm1_neg_fast(-z);
                    const double fz = (sn * invF) * div_fast(2. * (1.0 + u), -u * (2.0 + u));
                    pa = fma(tab[16 + i], fz * fma(-z, invx, 1.0), pa);
                }
                acc = fma(pa, hw, acc);
            }
            // [z0, x]: 1/sinh z = 2 sum_m exp(-(2m+1) z); each term integrates in closed form,
            //   A(z) = int sin(Fz) e^{-az} dz   = -e^{-az} (a sin Fz + F cos Fz) / (a²+F²)
            //   B(z) = int z sin(Fz) e^{-az} dz = -e^{-az} [ z (a sin Fz + F cos Fz)/(a²+F²)
            //                                                + ((a²-F²) sin Fz + 2aF cos Fz)/(a²+F²)² ]
            // and exp(-2 z0 m) < 3e-18 after eleven terms (checked against QUADPACK at 1e-13: 2e-14)
            if (x > z0) {
                double s0, c0, s1, c1;
                sincos_fast(F * z0, &s0, &c0);
                sincos_fast(F * x, &s1, &c1);
                const double r0 = exp(-2. * z0), r1 = exp(-2. * x);
                double e0 = exp(-z0), e1 = exp(-x), tail = 0.;
                const double F2 = F * F;
#pragma unroll 1
                for (int m = 0; m < 11; ++m) {
                    const double a = (double)(2 * m + 1);
                    const double iD = 1.0 / fma(a, a, F2);
                    const double p = (a * a - F2) * iD, r = 2. * a * F * iD;
                    const double u0 = fma(a, s0, F * c0), u1 = fma(a, s1, F * c1);
                    const double A0 = -e0 * u0 * iD, A1 = -e1 * u1 * iD;
                    const double B0 = -e0 * iD * (z0 * u0 + fma(p, s0, r * c0));
                    const double B1 = -e1 * iD * (x * u1 + fma(p, s1, r * c1));
                    tail += (A1 - A0) - (B1 - B0) * invx;
                    e0 *= r0; e1 *= r1;
                }
                acc = fma(2. * invF, tail, acc);
            }
        } else {                                                          // smooth branches (:21-23, :27-28)
            const double e2 = 1.0 - q * q * lk * lk / 9.;
            const double e = (q < ratio && e2 > 0.) ? sqrt(e2) : 0.;
            const double a1 = 1.0 - e;
            // away from e = 0 the range beyond z0 = 2 is integrated in closed form like the oscillatory branch:
            //   sinh(e z)/(e sinh z) = (1/e) sum_m [exp(-(a-e) z) - exp(-(a+e) z)],  a = 2m+1,
            //   int_{z0}^{x} exp(-b z)(1 - z/x) dz = exp(-b z0) [ (1 - z0/x) D phi1(b D) - D²/x phi2(b D) ],  D = x - z0,
            //   phi1(t) = (1 - exp(-t))/t,  phi2(t) = (1 - (1+t) exp(-t))/t²   (no cancellation for small b D);
            // checked against QUADPACK at 1e-13: 3e-15.  Near e = 0 the difference quotient cancels: panels.
            const bool closed = e >= 0.05;
            const double zsplit = closed ? fmin(x, 2.0) : x;
            double left = 0., width = 1.0;
            while (left < zsplit) {
                const double right = fmin(zsplit, left + width);
                const double hw = 0.5 * (right - left), mid = 0.5 * (right + left);
                double pa = 0.;
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const double z = fma(hw, tab[i], mid);
                    const double den = -expm1_neg_fast(-2. * z);
                    // sinh(e z)/(e sinh z) = exp(-(1-e) z) (1 - exp(-2 e z)) / (e (1 - exp(-2 z))); e -> 0: 2 z exp(-z)/(1-exp(-2z))
                    const double fz = (e > 0.) ? exp(-a1 * z) * (-expm1_neg_fast(-2. * e * z)) / (e * den)
                                               : 2. * z * exp(-z) / den;
                    pa = fma(tab[16 + i], fz * fma(-z, invx, 1.0), pa);
                }
                acc = fma(pa, hw, acc);
                left = right;
                if (left >= 2.0) width = left;
                if (a1 * left > 42.0) break;
            }
            if (closed && x > zsplit) {
                const double z0 = zsplit, D = x - z0, cD = (1.0 - z0 * invx) * D, D2x = D * D * invx;
                const double r0 = exp(-2. * z0), ep = exp(e * z0), en = 1.0 / ep;
                double ea = exp(-z0), tail = 0.;
#pragma unroll 1
                for (int m = 0; m < 11; ++m) {
                    const double a = (double)(2 * m + 1);
                    tail += ea * (ep * kho_tail(( a - e) * D, cD, D2x) - en * kho_tail((a + e) * D, cD, D2x));
                    ea *= r0;
                }
                acc += tail / e;
            }
        }
        const double p0 = sqrt(acc * (2.0 * invx));                      // coreIntegral (:32-37)
        const double u = q * r;
        const double pcs = (u <= 0.) ? 1.0 : 2. * j1_fast(u) / u;         // calcPcs (:40-45)
        const double ff = p0 * pcs;                                       // :90
        return ff * ff * w;
    }
};

// ---------------------------------------------------------------------------------- Isotropic ellipsoids
// models/ellipsoidsisotropic.py:51-84.  table: sin^2(alpha_k), cos^2(alpha_k), sin(alpha_k),
// alpha = linspace(0, pi/2, K)
template <> struct Contrib<MCSAS_MODEL_ELL_ISO> {
    static constexpr int ROWTAB = 2;       // per orientation k: R_k, 3 sqrt(sin alpha_k) / R_k^3
    static constexpr int INT_DIV_PARAM = 4, ROW_CLASS = 1;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = false;
    double ra2, rc2, v, w, s, invK;
    int K, fast;
    static __host__ __device__ __forceinline__ int table_doubles(int K) { return 3 * K; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &a, double *tab, int tid, int nt) {
        const int K = a.int_div;
        const double step = (PI / 2.) / (double)(K - 1);
        for (int k = tid; k < K; k += nt) {
            const double al = (k == K - 1) ? PI / 2. : (double)k * step;   // numpy.linspace(0, pi/2, K)
            const double sn = sin(al), cs = cos(al);
            tab[k] = sn * sn; tab[K + k] = cs * cs; tab[2 * K + k] = sn;
        }
    }
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        const double ra = p[0];
        const double rc = (p[1] != 0.0) ? ra * p[3] : p[2];          // :63-66
        ra2 = ra * ra; rc2 = rc * rc;
        const double vol = 4. / 3. * PI * (ra * ra) * rc;            // :80
        v = vol * (p[5] * p[5]);                                     // :83-84
        s = 0.;
        w = pow(vol, 2. * a.comp_exp);
        K = a.int_div; invK = 1.0 / (double)K;
        fast = (a.qmax * a.qmax * fmax(ra2, rc2) < 1048576.0 * 1048576.0) && (ra2 > 0.) && (rc2 > 0.);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.ra2 = readlane_f64(ra2, lane); o.rc2 = readlane_f64(rc2, lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.; o.invK = invK; o.K = K;
        o.fast = __builtin_amdgcn_readlane(fast, lane);
        return o;
    }
    __device__ __forceinline__ void fill_rowtab(const double *tab, double *rt, int lane) const {
        for (int k = lane; k < K; k += WAVE) {
            const double R = sqrt(ra2 * tab[k] + rc2 * tab[K + k]);
            rt[2 * k + 0] = R; rt[2 * k + 1] = (3. * sqrt(tab[2 * K + k])) / (R * R * R);
        }
    }
    template <int QPL>
    __device__ __forceinline__ void rows_rt(const QTables &t, int lane, double (&out)[QPL]) const {
        double q[QPL], acc[QPL];
#pragma unroll
        for (int j = 0; j < QPL; ++j) { q[j] = t.q[lane + WAVE * j]; acc[j] = 0.; }
        const double *rt = t.rowtab;
        for (int k = 0; k < K; ++k) {
            const double R = rt[2 * k], D = rt[2 * k + 1];
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const double x = q[j] * R;
                const double g = D * sin_minus_xcos_abs(x);
                acc[j] = fma(g, g, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const double q3 = t.q3inv[lane + WAVE * j];
            out[j] = ((acc[j] * (q3 * q3)) * invK) * w;
        }
    }
    __device__ __forceinline__ double intensity(double q, const double *tab) const {
        double acc = 0.;
        for (int k = 0; k < K; ++k) {
            const double x = q * sqrt(ra2 * tab[k] + rc2 * tab[K + k]);   // rPlugin (:56-58)
            double sn, cs;
            sincos_fast(x, &sn, &cs);
            const double f = div_fast(3. * (sn - x * cs), x * x * x);
            acc += (f * f) * tab[2 * K + k];
        }
        const double ff = sqrt(acc * invK);                           // :73
        return ff * ff * w;
    }
};

// ---------------------------------------------------------------------------------- Core-shell sphere
// models/sphericalcoreshell.py:50-77
template <> struct Contrib<MCSAS_MODEL_SPH_CS> {
    static constexpr int ROWTAB = 0;
    static constexpr int INT_DIV_PARAM = -1, ROW_CLASS = 0;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = false;
    double r, rt, ds, dc, vr, v, w, s;
    static __host__ __device__ __forceinline__ int table_doubles(int) { return 0; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &, double *, int, int) {}
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        r = p[0]; rt = p[0] + p[1];
        const double vc = 4. / 3 * PI * (r * r * r);
        const double vt = 4. / 3 * PI * (rt * rt * rt);
        vr = vc / vt;
        ds = p[3] - p[4];                 // eta_s - eta_sol
        dc = p[3] - p[2];                 // eta_s - eta_c
        v = vt; s = 0.;                   // volume() == absVolume()
        w = pow(vt, 2. * a.comp_exp);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.r = readlane_f64(r, lane); o.rt = readlane_f64(rt, lane); o.ds = readlane_f64(ds, lane);
        o.dc = readlane_f64(dc, lane); o.vr = readlane_f64(vr, lane); o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.;
        return o;
    }
    __device__ __forceinline__ double intensity(double q, const double *) const {
        double sn, cs;
        const double xs = q * rt;
        sincos_fast(xs, &sn, &cs);
        const double ks = div_fast(ds * 3. * (sn - xs * cs), xs * xs * xs);
        const double xc = q * r;
        sincos_fast(xc, &sn, &cs);
        const double kc = div_fast(dc * 3. * (sn - xc * cs), xc * xc * xc);
        const double f = ks - vr * kc;
        return f * f * w;
    }
};

// ---------------------------------------------------------------------------------- Gaussian chain
// models/gaussianchain.py:54-66
template <> struct Contrib<MCSAS_MODEL_GAUSS_CHAIN> {
    static constexpr int ROWTAB = 0;
    static constexpr int INT_DIV_PARAM = -1, ROW_CLASS = 0;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = false;
    double rg, beta, v, w, s;
    static __host__ __device__ __forceinline__ int table_doubles(int) { return 0; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &, double *, int, int) {}
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        rg = p[0];
        const double vol = p[3] * (rg * rg);          // k * rg^2 (:63-65)
        beta = p[1] - vol * p[2];                     // bp - (k rg^2) etas (:56)
        v = vol; s = 0.;
        w = pow(vol, 2. * a.comp_exp);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.rg = readlane_f64(rg, lane); o.beta = readlane_f64(beta, lane); o.w = readlane_f64(w, lane);
        o.v = 0.; o.s = 0.;
        return o;
    }
    __device__ __forceinline__ double intensity(double q, const double *) const {
        const double x = q * rg, u = x * x;
        double f = sqrt(2.) * sqrt(expm1(-u) + u) / u;
        f *= beta;
        if (q <= 0.0) f = beta;
        return f * f * w;
    }
};

// ---------------------------------------------------------------------------------- LMA dense spheres
// models/lmadensesphere.py:58-106: sphere form factor times a Percus-Yevick structure factor
template <> struct Contrib<MCSAS_MODEL_LMA_SPHERE> {
    static constexpr int ROWTAB = 0;
    static constexpr int INT_DIV_PARAM = -1, ROW_CLASS = 0;   // index of intDiv in params[] (-1: none); cost class of a row (see the head of this file)
    static constexpr bool CAN_SMEAR = true;
    double r, rh, mu, al, be, ga, v, w, s;
    static __host__ __device__ __forceinline__ int table_doubles(int) { return 0; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &, double *, int, int) {}
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        double p[MCSAS_MAX_PARAMS];
        full_params(a, row, p);
        r = p[0]; mu = p[1];
        double mf = p[2];
        if (mf == -1.) mf = pow(0.634 / mu, 1. / 3);                  // :72-73
        rh = mf * r;
        const double om = (1. - mu) * (1. - mu) * (1. - mu) * (1. - mu);
        al = (1. + 2. * mu) * (1. + 2. * mu) / om;                    // :76
        be = -6. * mu * (1. + mu / 2.) * (1. + mu / 2.) / om;         // :77
        ga = mu * al / 2.;                                            // :78
        const double vol = (PI * 4. / 3.) * (r * r * r);
        v = vol * (p[3] * p[3]); s = 0.;
        w = pow(vol, 2. * a.comp_exp);
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.r = readlane_f64(r, lane); o.rh = readlane_f64(rh, lane); o.mu = readlane_f64(mu, lane);
        o.al = readlane_f64(al, lane); o.be = readlane_f64(be, lane); o.ga = readlane_f64(ga, lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.;
        return o;
    }
    __device__ __forceinline__ double intensity(double q, const double *) const {
        double sn, cs;
        const double x = q * r;
        sincos_fast(x, &sn, &cs);
        const double f = div_fast(3. * (sn - x * cs), x * x * x);
        const double A = 2. * q * rh;                                 // :90
        sincos_fast(A, &sn, &cs);
        const double A2 = A * A, A3 = A2 * A, A4 = A2 * A2, A5 = A4 * A;
        const double G = al * (sn - A * cs) / A2
                       + be * (2. * A * sn + (2. - A2) * cs - 2.) / A3
                       + ga * (-1. * A4 * cs + 4. * ((3. * A2 - 6.) * cs + (A3 - 6. * A) * sn + 6.)) / A5;   // :79-85
        const double S = 1. / (1. + 24. * mu * G / A);                // :92
        const double ff = sqrt(f * f * S);                            // :99
        return ff * ff * w;
    }
};

// ---------------------------------------------------------------------------------- row cost
// What one row of a contribution will cost its wave, in arbitrary units, from the parameter set alone — used by the
// pipeline's producers to hand every wave rows of about the same TOTAL cost (chain_pipe.h, rows with an integral).  Only the
// order matters; models whose rows all cost the same return 0.  q_lo[j] / q_hi[j]: smallest / largest q of q slot j
// (the lanes of a wave take the 64 consecutive q of a slot in lockstep: a divergent loop runs to its slowest lane).
template <int M, int QPL>
__device__ __forceinline__ double row_cost(const Contrib<M> &c, const double *lq) {
    if constexpr (M == MCSAS_MODEL_KHOLODENKO) {
        // panels of 16 quadrature points: oscillatory branch (q > 3 / l_k) ceil(min(x, 2) max(1, F / 8)) + the closed-form tail,
        // smooth branch 2 + tail (closed form beyond z = 2) or the doubling panels near e = 0
        double cost = 0.;
        const double z0 = fmin(c.x, 2.0);
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const double qh = lq[WAVE * j + WAVE - 1];
            if (qh > c.ratio) {
                const double F = sqrt(fmax(qh * qh * c.lk * c.lk / 9. - 1.0, 0.));
                cost += ceil(z0 * fmax(1.0, F * 0.125)) + 1.0;
            } else {
                cost += 3.0;
            }
        }
        return cost;
    } else if constexpr (M == MCSAS_MODEL_CYL_ISO) {
        // per orientation group J1 takes its small-argument form (x <= 5 for every lane), its asymptotic form (x > 5 for every
        // lane: ~2.3x the instructions) or both; x = q r sqrt(1 - x_k^2) with sqrt(1 - x_k^2) spread over (0, 1)
        double cost = 0.;
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const double ql = lq[WAVE * j], qh = lq[WAVE * j + WAVE - 1];
            const double tl = 5.0 / (ql * c.r), th = 5.0 / (qh * c.r);
            const double f_gt = tl < 1.0 ? sqrt(1.0 - tl * tl) : 0.0;            // orientations on the asymptotic form for the whole slot
            const double f_le = th < 1.0 ? 1.0 - sqrt(1.0 - th * th) : 1.0;      // ... on the small-argument form for the whole slot
            cost += 144. * f_gt + 64. * f_le + 180. * (1.0 - f_gt - f_le);
        }
        return cost;
    } else {
        return 0.;
    }
}

// ---------------------------------------------------------------------------------- row evaluation
// out[j] = I(q[lane + 64 j]) for one contribution; the wave-uniform fast/slow choice is made once
// per row so the QPL evaluations stay in one basic block and interleave.
// Models with an orientation integral (ROWTAB > 0): the q-independent factors of the K integrand points
// are worked out once per row into the wave's LDS scratch, which leaves two multiplies, the branch-free
// sincos (and J1) and two FMAs per (q, k).
template <int M, int QPL> struct RowEval {
    // The row one q slot at a time, slot j's value handed to sink(j, value) in slot order as soon as it is known.  Rows that cost an
    // integral per point are evaluated slot by slot anyway, and a caller that consumes the value at once (the pipeline's
    // producer: d = new - old, its stores and sums) leaves no row array alive across the evaluation.
    template <class Sink>
    static __device__ __forceinline__ void run_each(const Contrib<M> &c, const QTables &t, int lane, Sink &&sink) {
        if constexpr (Contrib<M>::ROW_CLASS != 0 && !Contrib<M>::CAN_SMEAR) {
            if constexpr (M == MCSAS_MODEL_CYL_ISO) {
                if (t.rowtab && c.fast) {
                    c.fill_rowtab(t.tab, t.rowtab, lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    c.template rows_rt_each<QPL>(t, lane, sink);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    return;
                }
            }
            if constexpr (Contrib<M>::ROWTAB == 0) {
#pragma nounroll
                for (int js = 0; js < QPL; ++js) sink(js, c.intensity(t.q[lane + WAVE * js], t.tab));
                return;
            }
        }
        double out[QPL];
        run(c, t, lane, out);
#pragma unroll
        for (int j = 0; j < QPL; ++j) sink(j, out[j]);
    }
    static __device__ __forceinline__ void run(const Contrib<M> &c, const QTables &t, int lane, double (&out)[QPL]) {
        if constexpr (Contrib<M>::CAN_SMEAR) {
            if (t.smear_nk > 0) {
#pragma unroll
                for (int j = 0; j < QPL; ++j)
                    out[j] = smeared_intensity<M>(c, t.locs_t, t.cw, t.smear_nk, t.smear_stride, lane + WAVE * j, t.tab);
                return;
            }
        }
        if constexpr (Contrib<M>::ROWTAB > 0) {
            if (t.rowtab && c.fast) {
                c.fill_rowtab(t.tab, t.rowtab, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                c.template rows_rt<QPL>(t, lane, out);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                return;
            }
        }
        if constexpr (Contrib<M>::ROW_CLASS != 0) {
            // a point of these models is an integral (hundreds to thousands of instructions, loops and branches of its own):
            // ONE copy of it, run once per q slot — QPL inlined copies multiply the code and the values live at their joins
#pragma unroll
            for (int j = 0; j < QPL; ++j) out[j] = 0.;
#pragma nounroll
            for (int js = 0; js < QPL; ++js) {
                const double o = c.intensity(t.q[lane + WAVE * js], t.tab);
#pragma unroll
                for (int j = 0; j < QPL; ++j) out[j] = (j == js) ? o : out[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < QPL; ++j) out[j] = c.intensity(t.q[lane + WAVE * j], t.tab);
        }
    }
};

// the block's tables as the row evaluators see them; `tab` is followed by one row scratch per wave
template <int M>
__device__ __forceinline__ QTables make_qtables(const ModelArgs &a, const double *q, const double *q3inv, double *tab) {
    double *rt = nullptr;
    if constexpr (Contrib<M>::ROWTAB > 0) {
        if (a.use_rowtab)
            rt = tab + Contrib<M>::table_doubles(a.int_div) + (size_t)(threadIdx.x >> 6) * Contrib<M>::ROWTAB * a.int_div;
    }
    return QTables{q, q3inv, tab, rt, a.smear_locs_t, a.smear_cw, Contrib<M>::CAN_SMEAR ? a.smear_nk : 0, a.smear_stride};
}
template <int QPL> struct RowEval<MCSAS_MODEL_SPHERE, QPL> {
    template <class Sink>                                      // (see the general template; a sphere's row is evaluated whole)
    static __device__ __forceinline__ void run_each(const Contrib<MCSAS_MODEL_SPHERE> &c, const QTables &t, int lane, Sink &&sink) {
        double out[QPL];
        run(c, t, lane, out);
#pragma unroll
        for (int j = 0; j < QPL; ++j) sink(j, out[j]);
    }
    static __device__ __forceinline__ void run(const Contrib<MCSAS_MODEL_SPHERE> &c, const QTables &t, int lane,
                                               double (&out)[QPL]) {
        if (t.smear_nk > 0) {
#pragma unroll
            for (int j = 0; j < QPL; ++j)
                out[j] = smeared_intensity<MCSAS_MODEL_SPHERE>(c, t.locs_t, t.cw, t.smear_nk, t.smear_stride, lane + WAVE * j, t.tab);
            return;
        }
        if (c.fast) {
            // all operands first, then the QPL evaluations in ONE basic block: their dependent chains interleave and a
            // single wave keeps its SIMD's fp64 pipe busy (one evaluation after the other is latency bound: ~8 cycles
            // per instruction instead of ~4)
            // (four at a time: four interleaved chains keep the pipe busy as well as eight do and leave the registers of the
            // other four to the caller — the pipeline's producer carries Gram accumulators across this call)
            constexpr int RG = QPL < MCSAS_ROW_GROUP ? QPL : MCSAS_ROW_GROUP;
#pragma unroll
            for (int j0 = 0; j0 < QPL; j0 += RG) {
                double qq[RG], q3[RG], o[RG];
#pragma unroll
                for (int j = 0; j < RG; ++j) { qq[j] = t.q[lane + WAVE * (j0 + j)]; q3[j] = t.q3inv[lane + WAVE * (j0 + j)]; }
                c.template intensity_fast_n<RG>(qq, q3, o);
#pragma unroll
                for (int j = 0; j < RG; ++j) out[j0 + j] = o[j];
                if (j0 + RG < QPL) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            asm volatile("" ::: "memory");                    // keeps the optimiser from folding the two loops into one with a branch per element
#pragma unroll
            for (int j = 0; j < QPL; ++j) out[j] = c.intensity(t.q[lane + WAVE * j], t.tab);
        }
    }
};

}  // namespace mcsas
