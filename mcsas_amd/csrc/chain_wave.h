// chain_wave.h — one wavefront = one Monte-Carlo chain (McSAS.mcFit, mcsas.py:287-439), all
// attempts of one repetition (the retry loop of McSAS.analyse, mcsas.py:220-246).
//
// Layout: q index i lives in lane (i & 63), register slot (i >> 6); QPL = slots per lane.
//   registers : ft[QPL] (running model intensity), per-step new/old/test rows
//   LDS       : q, w = 1/sigma^2, wI = I/sigma^2 (read-only, shared by the block), orientation table
//   HBM       : rset[N][P], cached per-contribution intensity rows [N][qpad] (CACHE), outputs
// No barriers in the step loop: the three weighted sums are reduced with DPP inside the wave,
// the accept/reject decision is wave-uniform.
#pragma once
#include "chain_common.h"

namespace mcsas {

template <int M, int QPL, bool CACHE>
__global__ __launch_bounds__(64, (QPL < 8 || (QPL == 8 && CACHE)) ? 2 : 1) void chain_wave_kernel(const ChainArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int rep = blockIdx.x;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad;
    double *lq = lds, *lw = lds + qpad, *lwI = lds + 2 * qpad, *lq3 = lds + 3 * qpad, *tab = lds + 4 * qpad;
    for (int i = lane; i < qpad; i += WAVE) {
        const double qq = a.q[i];
        lq[i] = qq; lw[i] = a.w[i]; lwI[i] = a.wI[i]; lq3[i] = 1.0 / (qq * qq * qq);
    }
    const QTables qt = make_qtables<M>(a.model, lq, lq3, tab);
    Contrib<M>::fill_table(a.model, tab, lane, WAVE);
    __syncthreads();

    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = CACHE ? a.cache + (size_t)rep * a.cache_rows * qpad : nullptr;
    DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                   (uint32_t)(a.rep_offset + rep)};
    int overflow = 0;
    uint64_t draw_pos = 0;                 // uniforms consumed so far by this rep (all attempts)
    const uint64_t t_start = wall_clock64();

    double ft[QPL];
    FitResult cur{1.0, 0.0, 0.0};
    int64_t num_iter = 0, num_moves = 0, total_steps = 0;
    int attempts = 0, converged = 0, stopped = 0;

    for (int attempt = 0; attempt <= a.max_retries; ++attempt) {
        ++attempts;
        // ------------------------------------------------------------ initial parameter set
        // generateParameters(N): N draws per active parameter, parameter-major (scatteringmodel.py:117-127)
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
                        rset[(size_t)n * P + p] = row[p];
                    }
            } else {
#pragma unroll
                for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p) row[p] = a.gen_lo[p] > 0. ? a.gen_lo[p] : 1e-9;
            }
            Contrib<M> mine;
            mine.prepare(a.model, row);
            // model.calc(data, rset, c): rows accumulated in contribution order (scatteringmodel.py:90-101)
            const int cnt = min(WAVE, N - n0);
            for (int i = 0; i < cnt; ++i) {
                const Contrib<M> c = mine.bcast(__builtin_amdgcn_readfirstlane(i));
                double it[QPL];
                RowEval<M, QPL>::run(c, qt, lane, it);
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    ft[j] += it[j];
                    if (CACHE) cache[(size_t)(n0 + i) * qpad + lane + WAVE * j] = it[j];
                }
            }
        }
        if (!a.start_from_min) draw_pos += (uint64_t)N * P;

        // ------------------------------------------------------------ initial fit (mcsas.py:327-343)
        {
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                double wt = lw[lane + WAVE * j] * ft[j];
                s1 += wt; s2 = fma(wt, ft[j], s2); s3 = fma(lwI[lane + WAVE * j], ft[j], s3);
            }
            wave_sum3(s1, s2, s3);
            cur = solve_fit(a, s1, s2, s3);
        }
        num_iter = 0; num_moves = 0;
        int ri = 0;
        // The comparison chi²_t < chi² (mcsas.py:379) is made without divisions, as in the other kernels:
        // chi²·Q = S - num²/den at the optimum (centred sums when a background is fitted), so
        // chi²_t < chi²  <=>  num² > (S - X) den  with X = chi²·Q of the current state; the one division is
        // paid when a move is accepted.  Scale and background are only needed at the end of the attempt.
        const double nqd = (double)a.nq, invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
        double X = cur.chi2 * nqd;

        // ------------------------------------------------------------ MC loop (mcsas.py:354-404)
        bool running = (N > 1);
        while (running) {
            if (!(cur.chi2 > a.conv_crit) || !(num_iter < a.max_iter)) break;
            if (stop_requested(a)) { stopped = 1; break; }   // (one answer per wave: chain_common.h)
            // proposals for the next 64 steps, one per lane: generateParameters() draws P uniforms
            // per step in parameter order (mcsas.py:358)
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

            int k = 0;
            for (; k < WAVE; ++k) {
                if (!(cur.chi2 > a.conv_crit) || !(num_iter < a.max_iter)) { running = false; break; }
                const int kk = __builtin_amdgcn_readfirstlane(k);
                const Contrib<M> cnew = prop.bcast(kk);
                if (__builtin_amdgcn_readlane(povf, kk)) overflow = 1;
                double inew[QPL], test[QPL];
                if (CACHE) {
                    const double *orow = cache + (size_t)ri * qpad + lane;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) test[j] = orow[WAVE * j];
                } else {
                    // model.calc(data, rset[ri]) re-evaluated like mcsas.py:362
                    double orow[MCSAS_MAX_ACTIVE];
#pragma unroll
                    for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p) orow[p] = (p < P) ? rset[(size_t)ri * P + p] : 0.;
                    Contrib<M> cold;
                    cold.prepare(a.model, orow);
                    RowEval<M, QPL>::run(cold, qt, lane, test);
                }
                double s1 = 0., s2 = 0., s3 = 0.;
                RowEval<M, QPL>::run(cnew, qt, lane, inew);
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    // mcsas.py:367 has (ft - old) + new; every execution mode here adds the fp64 difference d = new - old
                    // instead (the workgroup and pipeline kernels carry d rows), so that ft is the SAME number in all
                    // three modes — at most one ulp per accepted move away from the reference's order
                    test[j] = ft[j] + (inew[j] - test[j]);
                    double wt = lw[lane + WAVE * j] * test[j];
                    s1 += wt; s2 = fma(wt, test[j], s2); s3 = fma(lwI[lane + WAVE * j], test[j], s3);
                }
                wave_sum3(s1, s2, s3);
                // s1 = Σ w C, s2 = Σ w C², s3 = Σ w I C of the candidate (mcsas.py:376)
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
                        if (CACHE) cache[(size_t)ri * qpad + lane + WAVE * j] = inew[j];
                    }
#pragma unroll
                    for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                        if (p < P) {
                            double val = readlane_f64(prow[p], kk);
                            if (lane == 0) rset[(size_t)ri * P + p] = val;
                        }
                    if (!CACHE) __threadfence_block();   // rset[ri] is re-read by every lane N steps later
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
                double wt = lw[lane + WAVE * j] * ft[j];
                s1 += wt; s2 = fma(wt, ft[j], s2); s3 = fma(lwI[lane + WAVE * j], ft[j], s3);
            }
            wave_sum3(s1, s2, s3);
            cur = solve_fit(a, s1, s2, s3);
            // reported chi-squared: direct residual sum, like chiSqr (backgroundscalingfit.py:72-77)
            double rs = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const int i = lane + WAVE * j;
                double r = a.I[i] - (ft[j] * cur.A + cur.b);
                rs += lw[i] * r * r;
            }
            cur.chi2 = wave_sum(rs) / (double)a.nq;
        }
        converged = !(cur.chi2 > a.conv_crit);
        if (converged || stopped) break;
    }

    // ---------------------------------------------------------------- outputs (mcsas.py:428-439)
#pragma unroll
    for (int j = 0; j < QPL; ++j)
        a.fit[(size_t)rep * qpad + lane + WAVE * j] = ft[j] * cur.A + cur.b;      // ifinal*sc[0]+sc[1]
    overflow = __any(overflow);
    if (lane == 0) {
        ChainOut o;
        o.chisq = cur.chi2; o.scaling = cur.A; o.background = cur.b;
        o.seconds = (double)(wall_clock64() - t_start) * 1e-8;                    // 100 MHz counter
        o.num_iter = num_iter; o.num_moves = num_moves; o.draws = (int64_t)draw_pos;
        o.total_steps = total_steps;
        o.attempts = attempts; o.converged = converged; o.stream_overflow = overflow; o.stopped = stopped;
        a.out[rep] = o;
    }
}

}  // namespace mcsas
