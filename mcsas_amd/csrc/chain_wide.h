// chain_wide.h — one workgroup = one Monte-Carlo chain, the q-points split over its waves: the kernel for data sets with
// more than 1024 q-points (un-binned files, nBin = 0: dataobj/dataconfig.py:99; McSAS.analyse takes any data.count,
// mcsas.py:210).  Same chain as chain_wave.h (McSAS.mcFit, mcsas.py:287-439; retry loop of McSAS.analyse, :220-246), same
// arithmetic per q-point; what differs is where a q-point lives and how the three weighted sums of a step are put together.
//
// Layout: NW = blockDim.x / 64 waves (<= 8), qpad = 64 * QPL * NW; wave v owns the q slice [64 QPL v, 64 QPL (v + 1)):
// q index i = 64 (QPL v + j) + lane, j = register slot.  ft, w, wI of the slice live in registers; q and 1/q^3 in LDS when
// they fit beside the model's tables (ChainArgs::pad1 = 1), else they are read from HBM/L2 through the same pointers.
// Per step: every wave evaluates its slice of the proposal's row, reads its slice of the `old` row from the row cache
// (always on), reduces its three partial sums with DPP and parks them in LDS; ONE workgroup barrier; every wave adds the NW
// partials in wave order — the same numbers in every wave, so the accept/reject decision is uniform over the workgroup
// without a second exchange.  The partial slots alternate by step parity: a wave can only reach its write of step s + 2
// through the barrier of step s + 1, which every wave passes after reading step s.
#pragma once
#include "chain_common.h"

namespace mcsas {

constexpr int WIDE_MAX_WAVES = 8;        // 2 waves per SIMD: the row evaluators keep their 256 registers (16 waves: 128, and the sphere's interleaved sincos spills)
constexpr int WIDE_PART_DOUBLES = 2 * WIDE_MAX_WAVES * 4 + 8;       // partial sums [parity][wave][4], then the stop word

// block-wide sums of three per-lane values, the same in every thread; `par` alternates per call
__device__ __forceinline__ void wide_sum3(double &s1, double &s2, double &s3, double *part, int &par, int wave, int lane, int NW) {
    wave_sum3(s1, s2, s3);
    double *slot = part + (size_t)(par * WIDE_MAX_WAVES) * 4;
    if (lane == 0) { slot[wave * 4 + 0] = s1; slot[wave * 4 + 1] = s2; slot[wave * 4 + 2] = s3; }
    __syncthreads();
    double t1 = 0., t2 = 0., t3 = 0.;
    for (int v = 0; v < NW; ++v) { t1 += slot[v * 4 + 0]; t2 += slot[v * 4 + 1]; t3 += slot[v * 4 + 2]; }
    s1 = t1; s2 = t2; s3 = t3;
    par ^= 1;
}

template <int M, int QPL>
__global__ __launch_bounds__(WIDE_MAX_WAVES * 64) void chain_wide_kernel(const ChainArgs a, const double *q3inv_glb) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int rep = blockIdx.x;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad;
    const int q0 = wave * QPL * WAVE;                              // first q index of this wave's slice
    double *part = lds, *tab = lds + WIDE_PART_DOUBLES;
    int32_t *stop_word = reinterpret_cast<int32_t *>(part + 2 * WIDE_MAX_WAVES * 4);
    // the model's tables and the per-wave row scratch (make_qtables finds the latter behind the former)
    int tabd = Contrib<M>::table_doubles(a.model.int_div);
    if constexpr (Contrib<M>::ROWTAB > 0) { if (a.model.use_rowtab) tabd += NW * Contrib<M>::ROWTAB * a.model.int_div; }
    const double *qsrc = a.q, *q3src = q3inv_glb;
    if (a.pad1) {                                                  // q and 1/q^3 fit in LDS
        double *lq = tab + tabd, *lq3 = lq + qpad;
        for (int i = tid; i < qpad; i += blockDim.x) { lq[i] = a.q[i]; lq3[i] = q3inv_glb[i]; }
        qsrc = lq; q3src = lq3;
    }
    Contrib<M>::fill_table(a.model, tab, tid, blockDim.x);
    if (tid == 0) *stop_word = 0;
    __syncthreads();
    QTables qt = make_qtables<M>(a.model, qsrc + q0, q3src + q0, tab);
    if (qt.locs_t) qt.locs_t += q0;                                // smearing: evaluation points of this slice
    double lw[QPL], lwI[QPL];
#pragma unroll
    for (int j = 0; j < QPL; ++j) { lw[j] = a.w[q0 + lane + WAVE * j]; lwI[j] = a.wI[q0 + lane + WAVE * j]; }

    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * a.cache_rows * qpad + q0 + lane;     // this lane's column of the rows
    DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                   (uint32_t)(a.rep_offset + rep)};
    int overflow = 0, par = 0;
    uint64_t draw_pos = 0;
    const uint64_t t_start = wall_clock64();

    double ft[QPL];
    FitResult cur{1.0, 0.0, 0.0};
    int64_t num_iter = 0, num_moves = 0, total_steps = 0;
    int attempts = 0, converged = 0, stopped = 0;

    for (int attempt = 0; attempt <= a.max_retries; ++attempt) {
        ++attempts;
        // ------------------------------------------------------------ initial parameter set (scatteringmodel.py:117-127)
#pragma unroll
        for (int j = 0; j < QPL; ++j) ft[j] = 0.;
        for (int n0 = 0; n0 < N; n0 += WAVE) {
            const int n = n0 + lane;
            double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
            if (n < N) {
#pragma unroll
                for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                    if (p < P) {
                        if (a.start_from_min) row[p] = a.start_value[p];         // mcsas.py:310-315
                        else {
                            double u = src.at(draw_pos + (uint64_t)p * N + n, overflow);   // parameter-major
                            row[p] = gen_transform(a.gen_kind[p], u) * (a.gen_hi[p] - a.gen_lo[p]) + a.gen_lo[p];
                        }
                        if (wave == 0) rset[(size_t)n * P + p] = row[p];
                    }
            } else {
#pragma unroll
                for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p) row[p] = a.gen_lo[p] > 0. ? a.gen_lo[p] : 1e-9;
            }
            Contrib<M> mine;                                       // (every wave prepares all 64: the same numbers)
            mine.prepare(a.model, row);
            const int cnt = min(WAVE, N - n0);
            for (int i = 0; i < cnt; ++i) {                        // rows accumulated in contribution order (scatteringmodel.py:90-101)
                const Contrib<M> c = mine.bcast(__builtin_amdgcn_readfirstlane(i));
                double it[QPL];
                RowEval<M, QPL>::run(c, qt, lane, it);
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    ft[j] += it[j];
                    cache[(size_t)(n0 + i) * qpad + WAVE * j] = it[j];
                }
            }
        }
        if (!a.start_from_min) draw_pos += (uint64_t)N * P;

        // ------------------------------------------------------------ initial fit (mcsas.py:327-343)
        {
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                double wt = lw[j] * ft[j];
                s1 += wt; s2 = fma(wt, ft[j], s2); s3 = fma(lwI[j], ft[j], s3);
            }
            wide_sum3(s1, s2, s3, part, par, wave, lane, NW);
            cur = solve_fit(a, s1, s2, s3);
        }
        num_iter = 0; num_moves = 0;
        int ri = 0;
        // division-free comparison as in chain_wave.h: chi²_t < chi²  <=>  num² > (S - X) den
        const double nqd = (double)a.nq, invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
        double X = cur.chi2 * nqd;

        // ------------------------------------------------------------ MC loop (mcsas.py:354-404)
        bool running = (N > 1);
        while (running) {
            if (!(cur.chi2 > a.conv_crit) || !(num_iter < a.max_iter)) break;
            // McSAS.stop: one thread looks, everybody acts on what it saw (a wave of its own could see the word change
            // between two waves' reads and leave the others at the next barrier)
            if (a.stop_flag) {
                if (tid == 0) *stop_word = stop_requested(a) ? 1 : 0;
                __syncthreads();
                if (*stop_word) { stopped = 1; break; }
            }
            // proposals for the next 64 steps, one per lane (mcsas.py:358), the same in every wave
            double prow[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
            int povf = 0;
#pragma unroll
            for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                if (p < P) {
                    double u = 0.5;
                    if (num_iter + lane < a.max_iter)
                        u = src.at(draw_pos + (uint64_t)(num_iter + lane) * P + p, povf);
                    prow[p] = gen_transform(a.gen_kind[p], u) * (a.gen_hi[p] - a.gen_lo[p]) + a.gen_lo[p];
                }
            Contrib<M> prop;
            prop.prepare(a.model, prow);

            for (int k = 0; k < WAVE; ++k) {
                if (!(cur.chi2 > a.conv_crit) || !(num_iter < a.max_iter)) { running = false; break; }
                const int kk = __builtin_amdgcn_readfirstlane(k);
                const Contrib<M> cnew = prop.bcast(kk);
                if (__builtin_amdgcn_readlane(povf, kk)) overflow = 1;
                double inew[QPL], test[QPL];
                const double *orow = cache + (size_t)ri * qpad;
#pragma unroll
                for (int j = 0; j < QPL; ++j) test[j] = orow[WAVE * j];
                RowEval<M, QPL>::run(cnew, qt, lane, inew);
                double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    test[j] = ft[j] + (inew[j] - test[j]);        // ft + d, d = new - old: as in every execution mode (chain_wave.h)
                    double wt = lw[j] * test[j];
                    s1 += wt; s2 = fma(wt, test[j], s2); s3 = fma(lwI[j], test[j], s3);
                }
                wide_sum3(s1, s2, s3, part, par, wave, lane, NW);
                double S = a.SII, num = s3, den = s2;
                if (a.find_bg) {
                    const double numc = fma(-SIoSw, s1, s3), denc = fma(-(s1 * invSw), s1, s2);
                    const bool neg_b = a.pos_bg && (fma(a.SI, denc, -(numc * s1)) < 0.);
                    if (!neg_b) { S = Scen; num = numc; den = denc; }
                }
                if (num * num > (S - X) * den) {                                   // mcsas.py:379-390
                    X = S - num * num / den;
                    cur.chi2 = X / nqd;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        ft[j] = test[j];
                        cache[(size_t)ri * qpad + WAVE * j] = inew[j];
                    }
#pragma unroll
                    for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                        if (p < P) {
                            double val = readlane_f64(prow[p], kk);
                            if (tid == 0) rset[(size_t)ri * P + p] = val;
                        }
                    ++num_moves;
                }
                ri = (ri + 1 == N) ? 0 : ri + 1;                                   // mcsas.py:403-404
                ++num_iter;
            }
        }
        draw_pos += (uint64_t)num_iter * P;
        total_steps += num_iter;

        // ------------------------------------------------------------ final fit on ft (mcsas.py:424-426)
        {
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                double wt = lw[j] * ft[j];
                s1 += wt; s2 = fma(wt, ft[j], s2); s3 = fma(lwI[j], ft[j], s3);
            }
            wide_sum3(s1, s2, s3, part, par, wave, lane, NW);
            cur = solve_fit(a, s1, s2, s3);
            // reported chi-squared: direct residual sum, like chiSqr (backgroundscalingfit.py:72-77)
            double rs = 0., z1 = 0., z2 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                double r = a.I[q0 + lane + WAVE * j] - (ft[j] * cur.A + cur.b);
                rs += lw[j] * r * r;
            }
            wide_sum3(rs, z1, z2, part, par, wave, lane, NW);
            cur.chi2 = rs / (double)a.nq;
        }
        converged = !(cur.chi2 > a.conv_crit);
        if (converged || stopped) break;
    }

    // ---------------------------------------------------------------- outputs (mcsas.py:428-439)
#pragma unroll
    for (int j = 0; j < QPL; ++j)
        a.fit[(size_t)rep * qpad + q0 + lane + WAVE * j] = ft[j] * cur.A + cur.b;
    overflow = __any(overflow);
    if (tid == 0) {
        ChainOut o;
        o.chisq = cur.chi2; o.scaling = cur.A; o.background = cur.b;
        o.seconds = (double)(wall_clock64() - t_start) * 1e-8;
        o.num_iter = num_iter; o.num_moves = num_moves; o.draws = (int64_t)draw_pos;
        o.total_steps = total_steps;
        o.attempts = attempts; o.converged = converged; o.stream_overflow = overflow; o.stopped = stopped;
        a.out[rep] = o;
    }
}

}  // namespace mcsas
