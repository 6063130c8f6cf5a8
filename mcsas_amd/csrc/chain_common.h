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
    const int32_t *stop_flag;          // McSAS.stop as the kernels see it: a word in HOST memory (pinned, mapped)
    int32_t *stop_relay;               // device memory: [0] the relayed stop word, [2..3] (uint64) time of the last look at the host word (stop_requested)
    // per-chain state / outputs in HBM
    double  *rset;                     // [n_reps][n_contrib][n_active]
    double  *cache;                    // [n_reps][cache_rows][qpad] per-contribution intensities (or null)
    int32_t  cache_rows, pad1;
    double  *fit;                      // [n_reps][qpad]
    ChainOut *out;                     // [n_reps]
};

// McSAS.stop (mcsas.py:357: polled once per step).  The caller's word lives in host memory, and a read of it crosses the host link:
// microseconds each, served one at a time — 8192 chains looking every 64 steps were measured to spend HALF of a launch queued
// up on those reads (3.7e8 instead of 7.5e8 steps/s, whatever the q count).  So the chains look at a word in device memory (an L2
// hit) and whoever comes by more than 100 us after the last look at the host word (one compare-and-swap on a device time stamp
// elects it) reads the host word and relays it: a few thousand host reads per second whatever the number of chains, and a stop
// is seen by every chain within its next 64 steps + 100 us.
__device__ __forceinline__ bool stop_requested(const ChainArgs &a) {
    if (!a.stop_flag) return false;
    // one lane asks, the wave gets one answer (every lane of a wave must leave its loop at the same step)
    const int first = (int)__builtin_ctzll(__ballot(1));
    int r = 0;
    if ((int)(threadIdx.x & 63) == first) {
        if (__hip_atomic_load(a.stop_relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) r = 1;
        else {
            unsigned long long *stamp = reinterpret_cast<unsigned long long *>(a.stop_relay + 2);
            const unsigned long long now = wall_clock64() | 1ull;     // 100 MHz; never the "nobody has looked yet" value 0
            unsigned long long last = __hip_atomic_load(stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((last == 0ull || now - last >= 10000ull) &&
                __hip_atomic_compare_exchange_strong(stamp, &last, now, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
                __hip_atomic_load(a.stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                __hip_atomic_store(a.stop_relay, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r = 1;
            }
        }
    }
    return __builtin_amdgcn_readfirstlane(r) != 0;
}

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
