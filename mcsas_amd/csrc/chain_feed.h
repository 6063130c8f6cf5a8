// chain_feed.h — Monte-Carlo chains whose intensity rows are evaluated by the HOST (mcsas_hip_analyse_host_rows).
//
// The reference's plug-in contract is a ScatteringModel subclass whose formfactor / volume are Python
// (bases/model/scatteringmodel.py:16-58, sasmodel.py:46-79).  A proposal depends on the random stream only, never on the state
// of its chain, so the caller can evaluate calcIntensity for a window of proposals ahead of the decisions; what is left of
// McSAS.mcFit (mcsas.py:287-439) runs here: the row cache (`old`, :362), test = ft - old + new (:367), the scale / background fit
// and chi² (:376), the decision (:379-390), the bookkeeping, the final fit (:424-430).
//
// One wavefront per chain, rows [qpad] in device memory, q index i in lane i & 63: the three weighted sums are taken in the order of
// chain_wave.h (slot-major per lane, then the DPP reduction), and the decision is the same division-free comparison.  The kernel is
// not tuned: the caller's row evaluation (microseconds to milliseconds per row in Python) is what such a run waits for.
#pragma once
#include "chain_common.h"

namespace mcsas {

struct FeedState {                 // per chain, lives in device memory between the windows of an attempt
    double  X, chi2, A, b;         // chi²·Q of the current state (division-free comparison), chi², scale, background
    int64_t num_iter, num_moves;
    int32_t ri, ended, converged, pad;
};

enum { FEED_STEPS = 0, FEED_INIT = 1, FEED_END = 2 };

struct FeedArgs {
    ChainArgs c;                   // data vectors and their sums, settings, rset, cache [n_reps][n_contrib][qpad], fit
    double *ft;                    // [n_reps][qpad] the running model intensity
    FeedState *state;              // [n_reps]
    const double *rows;            // [rows of this window][qpad]: calcIntensity()[0] of every proposal, zero-padded
    const double *pvals;           // [rows of this window][n_active]: the proposals' parameter values
    const int32_t *first, *count, *kind;   // [n_reps]: this chain's rows in the window; FEED_*
};

__device__ __forceinline__ void feed_sums(const ChainArgs &a, const double *ft, int lane, double &s1, double &s2, double &s3) {
    s1 = 0.; s2 = 0.; s3 = 0.;
    for (int i = lane; i < a.qpad; i += WAVE) {
        const double v = ft[i], wt = a.w[i] * v;
        s1 += wt; s2 = fma(wt, v, s2); s3 = fma(a.wI[i], v, s3);
    }
    wave_sum3(s1, s2, s3);
}

__global__ __launch_bounds__(64) void feed_rows_kernel(const FeedArgs f) {
    const ChainArgs &a = f.c;
    const int lane = threadIdx.x, rep = blockIdx.x;
    const int n = f.count[rep], kind = f.kind[rep];
    if (n == 0 && kind != FEED_END) return;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad;
    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * N * qpad;
    double *ft = f.ft + (size_t)rep * qpad;
    const double *rows = f.rows + (size_t)f.first[rep] * qpad;
    const double *pv = f.pvals + (size_t)f.first[rep] * P;
    FeedState s = f.state[rep];
    const double nqd = (double)a.nq;

    if (kind == FEED_INIT) {
        // the initial parameter set (mcsas.py:317) and model.calc over it (:319): rows summed in contribution order
        for (int i = lane; i < qpad; i += WAVE) ft[i] = 0.;
        for (int c = 0; c < n; ++c) {
            for (int i = lane; i < qpad; i += WAVE) {
                const double v = rows[(size_t)c * qpad + i];
                cache[(size_t)c * qpad + i] = v;
                ft[i] += v;
            }
            if (lane < P) rset[(size_t)c * P + lane] = pv[(size_t)c * P + lane];
        }
        double s1, s2, s3;
        feed_sums(a, ft, lane, s1, s2, s3);
        const FitResult cur = solve_fit(a, s1, s2, s3);                       // mcsas.py:327-343
        s.chi2 = cur.chi2; s.A = cur.A; s.b = cur.b; s.X = cur.chi2 * nqd;
        s.num_iter = 0; s.num_moves = 0; s.ri = 0; s.ended = 0; s.converged = 0;
    } else if (kind == FEED_STEPS) {
        const double invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
        for (int k = 0; k < n && N > 1; ++k) {                                // mcsas.py:354-404
            if (!(s.chi2 > a.conv_crit) || !(s.num_iter < a.max_iter)) break;
            const double *nw = rows + (size_t)k * qpad, *od = cache + (size_t)s.ri * qpad;
            double s1 = 0., s2 = 0., s3 = 0.;
            for (int i = lane; i < qpad; i += WAVE) {
                const double t = ft[i] + (nw[i] - od[i]);                     // (:367; the order of chain_wave.h)
                const double wt = a.w[i] * t;
                s1 += wt; s2 = fma(wt, t, s2); s3 = fma(a.wI[i], t, s3);
            }
            wave_sum3(s1, s2, s3);
            double S = a.SII, num = s3, den = s2;
            if (a.find_bg) {
                const double numc = fma(-SIoSw, s1, s3), denc = fma(-(s1 * invSw), s1, s2);
                const bool neg_b = a.pos_bg && (fma(a.SI, denc, -(numc * s1)) < 0.);
                if (!neg_b) { S = Scen; num = numc; den = denc; }
            }
            if (num * num > (S - s.X) * den) {                                // :379-390
                s.X = S - num * num / den;
                s.chi2 = s.X / nqd;
                for (int i = lane; i < qpad; i += WAVE) {
                    const double v = nw[i];
                    ft[i] = ft[i] + (v - od[i]);
                    cache[(size_t)s.ri * qpad + i] = v;
                }
                if (lane < P) rset[(size_t)s.ri * P + lane] = pv[(size_t)k * P + lane];
                ++s.num_moves;
            }
            s.ri = (s.ri + 1 == N) ? 0 : s.ri + 1;                            // :403-404
            ++s.num_iter;
        }
    }
    const bool over = kind == FEED_END || N <= 1 || !(s.chi2 > a.conv_crit) || !(s.num_iter < a.max_iter);
    if (over && !s.ended) {
        // the final fit on ft (mcsas.py:424-426), chi² as the direct residual sum (backgroundscalingfit.py:72-77), the fit (:430)
        double s1, s2, s3;
        feed_sums(a, ft, lane, s1, s2, s3);
        const FitResult cur = solve_fit(a, s1, s2, s3);
        double rs = 0.;
        for (int i = lane; i < qpad; i += WAVE) {
            const double r = a.I[i] - (ft[i] * cur.A + cur.b);
            rs += a.w[i] * r * r;
        }
        s.chi2 = wave_sum(rs) / nqd; s.A = cur.A; s.b = cur.b;
        for (int i = lane; i < qpad; i += WAVE) a.fit[(size_t)rep * qpad + i] = ft[i] * cur.A + cur.b;
        s.ended = 1;
        s.converged = !(s.chi2 > a.conv_crit);
    }
    if (lane == 0) f.state[rep] = s;
}

}  // namespace mcsas
