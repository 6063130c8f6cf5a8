// chain_common.h — kernel argument blocks and the closed-form scale/background fit shared by the
// chain kernels.
#pragma once
#include "models.h"

namespace mcsas {

// Per-repetition outputs (what mcFit returns besides rset/fit, mcsas.py:420-439)
struct ChainOut {
    double  chisq, scaling, background, seconds;
    int64_t num_iter, num_moves, draws, total_steps;
    int32_t attempts, converged, stream_overflow, stopped;
#ifdef MCSAS_STAMPS
    int64_t dbg[20];         // diagnostic build only (make EXTRA=-DMCSAS_STAMPS): s_memtime sums per phase
#endif
};

// In-kernel stamps (cdna_hip_programming.md §7): compiled in ONLY with -DMCSAS_STAMPS; such a build is
// for reading phase SHARES, never for timing.
#ifdef MCSAS_STAMPS
#define MCSAS_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MCSAS_STAMP_DECL(...) uint64_t __VA_ARGS__
#define MCSAS_STAMP_ADD(acc, t1, t0) (acc) += (int64_t)((t1) - (t0))
#else
#define MCSAS_STAMP(var) do { } while (0)
#define MCSAS_STAMP_DECL(...)
#define MCSAS_STAMP_ADD(acc, t1, t0) do { } while (0)
#endif

// The tuning / ablation word (mcsas_problem.reserved0 -> ChainArgs::pad0) exists in measurement builds only
// (make tuning: -DMCSAS_TUNING); the release library refuses a non-zero word and compiles every use of it to 0.
#ifdef MCSAS_TUNING
#define MCSAS_TUNE_BITS(a) ((a).pad0)
#else
#define MCSAS_TUNE_BITS(a) 0
#endif

struct ChainArgs {
    ModelArgs model;
    // data, padded to qpad = 64*QPL entries (pad: q = q[0], w = wI = I = 0)
    int32_t nq, qpad;
    const double *q, *w, *wI, *I;      // w = 1/sigma^2, wI = I/sigma^2
    double  Sw, SI, SII;               // sum w, sum w*I, sum w*I*I over the real entries
    double  Ssig2;                     // sum sigma^2 (aGoFs denominator)
    // settings
    int32_t n_contrib, n_reps;
    int32_t find_bg, pos_bg, start_from_min, max_retries;
    int64_t max_iter;
    double  conv_crit;
    double  gen_lo[MCSAS_MAX_ACTIVE], gen_hi[MCSAS_MAX_ACTIVE], start_value[MCSAS_MAX_ACTIVE];
    int32_t gen_kind[MCSAS_MAX_ACTIVE];
    // random numbers
    uint64_t seed;
    int32_t  rep_offset, pad0;
    const double *replay;              // [n_reps][replay_len] or null
    int64_t  replay_len;
    const int32_t *stop_flag;          // host-mapped word, polled
    // per-chain state / outputs in HBM
    double  *rset;                     // [n_reps][n_contrib][n_active]
    double  *cache;                    // [n_reps][cache_rows][qpad] per-contribution intensities (or null)
    int32_t  cache_rows, pad1;
    double  *fit;                      // [n_reps][qpad]
    ChainOut *out;                     // [n_reps]
};

struct FitResult { double A, b, chi2; };

// argmin_{A,b} sum w (I - A*C - b)^2 from the five weighted sums, and the reduced chi-squared
// at that optimum (what scipy.optimize.leastsq converges to in backgroundscalingfit.py:94-103,
// then chiSqr :72-77).  positiveBackground replaces b by |b| in the residual (:59-63): a negative
// free optimum therefore lands on the b = 0 boundary.
__device__ __forceinline__ FitResult solve_fit(const ChainArgs &a, double SC, double SCC, double SIC) {
    FitResult r;
    if (a.find_bg) {
        double det = a.Sw * SCC - SC * SC;
        r.A = (a.Sw * SIC - a.SI * SC) / det;
        r.b = (a.SI - r.A * SC) / a.Sw;
        if (a.pos_bg && r.b < 0.) { r.A = SIC / SCC; r.b = 0.; }
    } else {
        r.A = SIC / SCC; r.b = 0.;
    }
    double q2 = a.SII - 2. * r.A * SIC - 2. * r.b * a.SI
              + r.A * r.A * SCC + 2. * r.A * r.b * SC + r.b * r.b * a.Sw;
    r.chi2 = q2 / (double)a.nq;
    return r;
}

// draw number `idx` of this chain's uniform stream
struct DrawSource {
    const double *replay; int64_t replay_len; uint64_t seed; uint32_t chain;
    __device__ __forceinline__ double at(uint64_t idx, int &overflow) const {
        if (replay) {
            if ((int64_t)idx < replay_len) return glb(replay)[idx];
            overflow = 1;
            return 0.5;
        }
        return philox_uniform(seed, chain, idx);
    }
};

}  // namespace mcsas
