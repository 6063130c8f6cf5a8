// models.h — device form factors: one struct per ScatteringModel subclass on the hot path.
// Each `Contrib<M>` holds the per-contribution scalars (what volume()/absVolume()/surface()/
// weight() return for one parameter row) and evaluates I(q) = F(q)^2 * volume()^(2c), i.e.
// SASModel.calcIntensity()[0] (bases/model/sasmodel.py:46-79, smearing off).
#pragma once
#include "device_util.h"
#include "fastmath.h"
#include "../../include/mcsas_hip.h"

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
    int32_t pad;
    double  qmax;          // largest q of the data set: bounds q*R for the branch-free sincos
};

// read-only per-block tables in LDS
struct QTables {
    const double *q;       // [qpad]
    const double *q3inv;   // [qpad] 1/q^3
    const double *tab;     // orientation table of the model
};

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

template <int M> struct Contrib;

// ---------------------------------------------------------------------------------- Sphere
// models/sphere.py:32-63
template <> struct Contrib<MCSAS_MODEL_SPHERE> {
    double r, v, w, s, invr3;
    int fast;              // q*r < 2^20 for every q of the data set: branch-free sincos is valid
    static __device__ __forceinline__ int table_doubles(int) { return 0; }
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
    // branch-free evaluation, valid when `fast`: 1/x^3 from the two precomputed reciprocals
    __device__ __forceinline__ double intensity_fast(double q, double q3inv) const {
        double x = q * r, sn, cs;
        sincos_core(x, &sn, &cs);
        double f = (3. * (sn - x * cs)) * (q3inv * invr3);
        return f * f * w;
    }
    __device__ __forceinline__ double intensity(double q, const double *) const {
        double x = q * r, sn, cs;
        sincos_fast(x, &sn, &cs);
        double f = div_fast(3. * (sn - x * cs), x * x * x);  // sphere.py:62
        return f * f * w;
    }
};

// ---------------------------------------------------------------------------------- Cylinders
// models/cylindersisotropic.py:50-101.  table: x_k (ends replaced by 0.5, :60-61) and sqrt(1-x_k^2)
template <> struct Contrib<MCSAS_MODEL_CYL_ISO> {
    double r, hl, v, w, s, step;
    int K;
    static __device__ __forceinline__ int table_doubles(int K) { return 2 * K; }
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
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o; o.r = readlane_f64(r, lane); o.hl = readlane_f64(hl, lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.; o.step = step; o.K = K;
        return o;
    }
    __device__ __forceinline__ double intensity(double q, const double *tab) const {
        // end columns: analytic limits (:79-82)
        double qr = q * r, qh = q * hl;
        double f0 = 0.5 * (j1(qr) / qr);
        double fl = sin(qh) / qh;
        double prev = f0 * f0, acc = 0.;
        for (int k = 1; k < K - 1; ++k) {
            double qrs = q * (r * tab[K + k]);
            double qlx = q * (2. * hl * tab[k]);
            double f = (j1(qrs) * sin(qlx / 2.)) / (qrs * qlx);
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
    double a2, b2, at2, bt2, c1, c2, v, w, s, invK;
    int K;
    static __device__ __forceinline__ int table_doubles(int K) { return 2 * K; }
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
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o;
        o.a2 = readlane_f64(a2, lane); o.b2 = readlane_f64(b2, lane);
        o.at2 = readlane_f64(at2, lane); o.bt2 = readlane_f64(bt2, lane);
        o.c1 = readlane_f64(c1, lane); o.c2 = readlane_f64(c2, lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.; o.invK = invK; o.K = K;
        return o;
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

// ---------------------------------------------------------------------------------- row evaluation
// out[j] = I(q[lane + 64 j]) for one contribution; the wave-uniform fast/slow choice is made once
// per row so the QPL evaluations stay in one basic block and interleave.
template <int M, int QPL> struct RowEval {
    static __device__ __forceinline__ void run(const Contrib<M> &c, const QTables &t, int lane, double (&out)[QPL]) {
#pragma unroll
        for (int j = 0; j < QPL; ++j) out[j] = c.intensity(t.q[lane + WAVE * j], t.tab);
    }
};
template <int QPL> struct RowEval<MCSAS_MODEL_SPHERE, QPL> {
    static __device__ __forceinline__ void run(const Contrib<MCSAS_MODEL_SPHERE> &c, const QTables &t, int lane,
                                               double (&out)[QPL]) {
        if (c.fast) {
#pragma unroll
            for (int j = 0; j < QPL; ++j) out[j] = c.intensity_fast(t.q[lane + WAVE * j], t.q3inv[lane + WAVE * j]);
        } else {
#pragma unroll
            for (int j = 0; j < QPL; ++j) out[j] = c.intensity(t.q[lane + WAVE * j], t.tab);
        }
    }
};

}  // namespace mcsas
