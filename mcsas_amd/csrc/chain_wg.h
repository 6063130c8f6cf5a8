// chain_wg.h — one WORKGROUP per Monte-Carlo chain, for runs with few chains (numReps ≲ CUs).
//
// A chain is sequential only in its accept/reject decisions.  Everything a step needs besides the
// running intensity ft is known in advance: the proposal of step s is a pure function of the
// chain's random stream (mcsas.py:358), and the "old" intensity of contribution ri = s mod N
// (mcsas.py:362) cannot change between two visits N steps apart.  So the kernel speculates a
// window of K steps (2K <= N), exactly — no proposal is ever recomputed or discarded because of
// an earlier decision:
//
//   producer waves (1..W-1): for step s of window i evaluate the proposal's intensity row `new`
//       (the expensive part), fetch the cached row `old` of contribution ri from HBM, write `new`
//       to a spare row slot in HBM, put d = new - old in LDS together with the three sums that do
//       not depend on ft:  a = Σ w d,  e = Σ wI d,  g = Σ w d².
//   scanner wave (0): walks window i-1 sequentially.  With test = ft + d the fit sums are
//       SC_t = SC + a, SIC_t = SIC + e, SCC_t = SCC + 2h + g, h = Σ (w·ft)·d — ONE wave reduction on
//       the critical path — and chi²_t < chi² is decided without a division:
//       chi²·Q = S - num²/den  =>  accept  <=>  num² > (S - chi²·Q)·den.
//   one workgroup barrier per window; d rows are double-buffered in LDS.
//
// The two roles are separate code paths (wave specialisation): each has its own register
// allocation, both execute the same sequence of barriers.
//
// Row slots: the HBM cache holds N + 2K rows per chain.  A proposal is written to a spare slot; on
// accept the contribution's slot pointer moves to it and the old row's slot becomes spare, on
// reject the spare is reused — rows are never copied.
#pragma once
#include "chain_common.h"

namespace mcsas {

struct WgGeom {
    int32_t waves, window, qpl, tab_doubles;
    int32_t off_q3, off_tab, off_ft, off_wft, off_drow, off_scal, off_pval, off_ints;   // in doubles from the LDS base
    int32_t n_contrib;
    uint64_t lds_bytes;
};

constexpr int WG_MAX_WAVES = 8;

// host side: pick the proposal window that fits LDS; nonzero = does not fit
static inline int wg_geometry(int nq, int n_contrib, int tab_doubles, int waves, WgGeom *g) {
    int qpl = 1;
    while (qpl * 64 < nq) qpl *= 2;
    if (qpl > 16 || waves < 2 || waves > WG_MAX_WAVES) return 1;
    const int qpad = qpl * 64, np = waves - 1;
    const size_t budget = 160 * 1024;
    for (int mult = 8; mult >= 1; --mult) {
        int K = np * mult;
        if (2 * K > n_contrib || K > 64) continue;
        size_t dbl = 4 * (size_t)qpad + tab_doubles;          // q, w, wI, 1/q^3, table
        g->off_q3 = 3 * qpad; g->off_tab = 4 * qpad;
        g->off_ft = (int32_t)dbl; dbl += qpad;                // the scanner's ft
        g->off_wft = (int32_t)dbl; dbl += qpad;               // and w*ft
        g->off_drow = (int32_t)dbl; dbl += 2 * (size_t)K * qpad;
        g->off_scal = (int32_t)dbl; dbl += 2 * (size_t)K * 4;
        g->off_pval = (int32_t)dbl; dbl += 2 * (size_t)K * MCSAS_MAX_ACTIVE;
        g->off_ints = (int32_t)dbl;
        size_t ints = 2 * (size_t)K * 2 + n_contrib + 16;      // stage_slot, povf, slot_of, control
        size_t bytes = dbl * 8 + ints * 4;
        if (bytes <= budget) {
            g->waves = waves; g->window = K; g->qpl = qpl; g->tab_doubles = tab_doubles;
            g->n_contrib = n_contrib; g->lds_bytes = (bytes + 15) & ~(size_t)15;
            return 0;
        }
    }
    return 1;
}

// control words (int32) at the start of the int region.  DONE is double-buffered by window parity:
// a wave that is slow to leave barrier i must not see the flag the scanner raises during window i+1.
enum { CTL_DONE = 0, CTL_NUM_ITER = 2, CTL_FINISHED = 4, CTL_OVF = 5, CTL_COUNT = 16 };

struct WgShared {
    double *lq, *lw, *lwI, *lq3, *tab, *lft, *lwft, *drow, *scal, *pval;
    int32_t *ctl, *stage_slot, *povf, *slot_of;
};

__device__ __forceinline__ WgShared wg_carve(double *lds, const WgGeom &g, int qpad) {
    WgShared s;
    s.lq = lds; s.lw = lds + qpad; s.lwI = lds + 2 * qpad; s.lq3 = lds + g.off_q3; s.tab = lds + g.off_tab;
    s.lft = lds + g.off_ft; s.lwft = lds + g.off_wft;
    s.drow = lds + g.off_drow; s.scal = lds + g.off_scal; s.pval = lds + g.off_pval;
    s.ctl = reinterpret_cast<int32_t *>(lds + g.off_ints);
    s.stage_slot = s.ctl + CTL_COUNT;
    s.povf = s.stage_slot + 2 * g.window;
    s.slot_of = s.povf + 2 * g.window;
    return s;
}

// ---- shared by both roles: the initial parameter set and its intensity rows (mcsas.py:317-319)
template <int M, int QPL>
__device__ __forceinline__ void wg_init_rows(const ChainArgs &a, const WgGeom &g, const WgShared &sh,
                                             const DrawSource &src, uint64_t draw_pos, double *rset, double *cache) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = g.waves, T = W * WAVE, N = a.n_contrib, P = a.model.n_active, qpad = a.qpad;
    int ovf_init = 0;
    // contribution n = lane*W + wave + 64*W*i: every wave owns ~N/W rows
    for (int nb = 0; nb < N; nb += T) {
        const int n = nb + lane * W + wave;
        double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
        if (n < N) {
#pragma unroll
            for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                if (p < P) {
                    if (a.start_from_min) row[p] = a.start_value[p];
                    else {
                        double u = src.at(draw_pos + (uint64_t)p * N + n, ovf_init);
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
        for (int l = 0; l < WAVE; ++l) {
            const int nn = nb + l * W + wave;
            if (nn >= N) break;
            const Contrib<M> c = mine.bcast(__builtin_amdgcn_readfirstlane(l));
            double it[QPL];
            RowEval<M, QPL>::run(c, make_qtables<M>(a.model, sh.lq, sh.lq3, sh.tab), lane, it);
#pragma unroll
            for (int j = 0; j < QPL; ++j) cache[(size_t)nn * qpad + lane + WAVE * j] = it[j];
        }
    }
    if (__any(ovf_init) && lane == 0) atomicOr(&sh.ctl[CTL_OVF], 1);
}

// ---------------------------------------------------------------------------------- producer role
template <int M, int QPL>
__device__ __forceinline__ void wg_producer(const ChainArgs &a, const WgGeom &g, const WgShared &sh, int rep) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = g.waves, NP = W - 1, K = g.window, T = W * WAVE;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad;
    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * a.cache_rows * qpad;
    const DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                         (uint32_t)(a.rep_offset + rep)};
    const int rpw = K / NP;
    const bool chain_runs = (N > 1);
    uint64_t draw_pos = 0;

    for (int attempt = 0; attempt <= a.max_retries; ++attempt) {
        for (int i = tid; i < N; i += T) sh.slot_of[i] = i;
        for (int i = tid; i < 2 * K; i += T) sh.stage_slot[i] = N + i;
        __syncthreads();                                // B0: tables / control words in place
        wg_init_rows<M, QPL>(a, g, sh, src, draw_pos, rset, cache);
        const uint64_t step_base = draw_pos + (a.start_from_min ? 0 : (uint64_t)N * P);
        __syncthreads();                                // B1: rows of the initial set are in HBM

        // wave w owns window slots k = (w-1) + NP*jj; its m-th row overall is step
        // s(m) = (m / rpw)*K + (w-1) + NP*(m % rpw); proposals are prepared 64 rows at a time
        int64_t m0 = -64;
        Contrib<M> prop;
        double prow[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
        int pov = 0;
        for (int64_t win = 0;; ++win) {
            const int buf = (int)(win & 1);
            if (chain_runs && !(MCSAS_TUNE_BITS(a) & 1)) {
                for (int jj = 0; jj < rpw; ++jj) {
                    const int k = (wave - 1) + NP * jj;
                    const int64_t s = win * K + k;
                    if (s >= a.max_iter) break;
                    const int64_t m = win * rpw + jj;
                    if (m >= m0 + WAVE) {
                        m0 = m;
                        const int64_t ml = m + lane;
                        const int64_t sl = (ml / rpw) * K + (wave - 1) + (int64_t)NP * (ml % rpw);
                        pov = 0;
#pragma unroll
                        for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                            if (p < P) {
                                double u = 0.5;
                                if (sl < a.max_iter) u = src.at(step_base + (uint64_t)sl * P + p, pov);
                                prow[p] = gen_transform(a.gen_kind[p], u) * (a.gen_hi[p] - a.gen_lo[p]) + a.gen_lo[p];
                            }
                        prop.prepare(a.model, prow);
                    }
                    const int bl = __builtin_amdgcn_readfirstlane((int)(m - m0));
                    const Contrib<M> cnew = prop.bcast(bl);
                    const int ri = (int)(s % N);
                    const int oslot = sh.slot_of[ri], sslot = sh.stage_slot[buf * K + k];
                    const double *orow = cache + (size_t)oslot * qpad + lane;
                    double *nrow = cache + (size_t)sslot * qpad + lane;
                    double *dr = sh.drow + ((size_t)buf * K + k) * qpad + lane;
                    double d[QPL], nwv[QPL];
#pragma unroll
                    for (int j = 0; j < QPL; ++j) d[j] = orow[WAVE * j];
                    RowEval<M, QPL>::run(cnew, make_qtables<M>(a.model, sh.lq, sh.lq3, sh.tab), lane, nwv);
                    double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        const int i = lane + WAVE * j;
                        const double nw = nwv[j];
                        nrow[WAVE * j] = nw;
                        d[j] = nw - d[j];
                        dr[WAVE * j] = d[j];
                        const double wd = sh.lw[i] * d[j];
                        s1 += wd; s2 = fma(sh.lwI[i], d[j], s2); s3 = fma(wd, d[j], s3);
                    }
                    wave_sum3(s1, s2, s3);
                    if (lane == 0) {
                        double *sc = sh.scal + ((size_t)buf * K + k) * 4;
                        sc[0] = s1; sc[1] = s2; sc[2] = s3;
                    }
#pragma unroll
                    for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                        if (p < P) {
                            const double v = readlane_f64(prow[p], bl);
                            if (lane == 0) sh.pval[((size_t)buf * K + k) * MCSAS_MAX_ACTIVE + p] = v;
                        }
                    const int ov = __builtin_amdgcn_readlane(pov, bl);
                    if (lane == 0) sh.povf[buf * K + k] = ov;
                }
            }
            __syncthreads();                            // BW: window `win` produced, window win-1 scanned
            if (sh.ctl[CTL_DONE + buf]) break;
        }
        __syncthreads();                                // B2: scanner published num_iter / finished
        const int64_t it = (int64_t)(uint32_t)sh.ctl[CTL_NUM_ITER] | ((int64_t)sh.ctl[CTL_NUM_ITER + 1] << 32);
        draw_pos = step_base + (uint64_t)it * P;
        const int finished = sh.ctl[CTL_FINISHED];
        __syncthreads();                                // B3
        if (finished) break;
    }
}

// ---------------------------------------------------------------------------------- scanner role
template <int M, int QPL>
__device__ __forceinline__ void wg_scanner(const ChainArgs &a, const WgGeom &g, const WgShared &sh, int rep) {
    const int lane = threadIdx.x & 63;
    const int W = g.waves, K = g.window, T = W * WAVE;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad;
    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * a.cache_rows * qpad;
    const DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                         (uint32_t)(a.rep_offset + rep)};
    const uint64_t t_start = wall_clock64();
    __builtin_amdgcn_s_setprio(3);                      // the latency-critical wave
    const double invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
    const double nqd = (double)a.nq;
    const bool chain_runs = (N > 1);
    double *lft = sh.lft + lane, *lwft = sh.lwft + lane;

    FitResult cur{1.0, 0.0, 0.0};
    int64_t num_iter = 0, num_moves = 0, total_steps = 0;
    int attempts = 0, converged = 0, stopped = 0, overflow = 0;
    uint64_t draw_pos = 0;

    for (int attempt = 0; attempt <= a.max_retries; ++attempt) {
        ++attempts;
        for (int i = threadIdx.x; i < N; i += T) sh.slot_of[i] = i;
        for (int i = threadIdx.x; i < 2 * K; i += T) sh.stage_slot[i] = N + i;
        if (lane == 0) { sh.ctl[CTL_DONE] = 0; sh.ctl[CTL_DONE + 1] = 0; if (attempt == 0) sh.ctl[CTL_OVF] = 0; }
        __syncthreads();                                // B0
        wg_init_rows<M, QPL>(a, g, sh, src, draw_pos, rset, cache);
        const uint64_t step_base = draw_pos + (a.start_from_min ? 0 : (uint64_t)N * P);
        __syncthreads();                                // B1

        double SC, SIC, SCC;
        {   // model.calc: rows accumulated in contribution order (scatteringmodel.py:90-101)
            double ft[QPL];
#pragma unroll
            for (int j = 0; j < QPL; ++j) ft[j] = 0.;
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int j = 0; j < QPL; ++j) ft[j] += cache[(size_t)n * qpad + lane + WAVE * j];
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const double wf = sh.lw[lane + WAVE * j] * ft[j];
                s1 += wf; s2 = fma(wf, ft[j], s2); s3 = fma(sh.lwI[lane + WAVE * j], ft[j], s3);
                lft[WAVE * j] = ft[j]; lwft[WAVE * j] = wf;
            }
            wave_sum3(s1, s2, s3);
            SC = s1; SCC = s2; SIC = s3;
            cur = solve_fit(a, SC, SCC, SIC);
        }
        num_iter = 0; num_moves = 0;
        double X = cur.chi2 * nqd;                      // chi²·Q of the current state
        bool live = chain_runs;

        for (int64_t win = 0;; ++win) {
            const int buf = (int)(win & 1);
            if (win > 0 && live) {
                const int sb = buf ^ 1;                 // window win-1
                bool touched = false;
                if ((win & 15) == 1 && stop_requested(a)) stopped = 1;
                int ri = (int)(((win - 1) * K) % N);
                const int64_t budget = a.max_iter - num_iter;
                const int kmax = budget < K ? (int)budget : K;
                const double *dbase = sh.drow + (size_t)sb * K * qpad + lane;
                const double *sbase = sh.scal + (size_t)sb * K * 4;
                int k = 0;
                if (!(cur.chi2 > a.conv_crit) || stopped) k = kmax;      // nothing to do
                if (MCSAS_TUNE_BITS(a) & 2) { num_iter += kmax - k; k = kmax; }       // diagnostic: skip the scan
                // Steps are decided EIGHT at a time: h_g = Σ (w·ft)·d_g for the next eight steps is
                // valid for all of them as long as none is accepted (the usual case); lane g of every
                // octet then evaluates step g's decision.  The first accepted step (if any) is applied
                // and the group restarts behind it — the sequence of decisions is exactly the serial one.
                const int g = lane & 7;
                const int gslot = ((g & 1) << 2) | (g & 2) | ((g >> 2) & 1);   // accumulator holding step g (bit-reversed)
                while (k < kmax) {
                    const int gcount = (kmax - k) < 8 ? (kmax - k) : 8;
                    double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        const double wf = lwft[WAVE * j];
#pragma unroll
                        for (int gg = 0; gg < 8; ++gg) {
                            const int kk = (k + gg < K) ? k + gg : K - 1;            // stay inside the window buffer
                            const int slot = ((gg & 1) << 2) | (gg & 2) | ((gg >> 2) & 1);
                            acc[slot] = fma(wf, dbase[(size_t)kk * qpad + WAVE * j], acc[slot]);
                        }
                    }
                    const double h = wave_sum8_transposed(acc, lane);               // lane g: h of step k+g
                    (void)gslot;
                    const int kg = (k + g < K) ? k + g : K - 1;
                    const double *sc = sbase + kg * 4;
                    const double SCt = SC + sc[0], SICt = SIC + sc[1], SCCt = SCC + fma(2., h, sc[2]);
                    // chi²·Q = S - num²/den for the candidate (centred sums when a background is fitted)
                    double S = a.SII, num = SICt, den = SCCt;
                    if (a.find_bg) {
                        const double numc = fma(-SIoSw, SCt, SICt), denc = fma(-(SCt * invSw), SCt, SCCt);
                        const bool neg_b = a.pos_bg && (fma(a.SI, denc, -(numc * SCt)) < 0.);
                        if (!neg_b) { S = Scen; num = numc; den = denc; }
                    }
                    const bool acc_g = (g < gcount) && (num * num > (S - X) * den);   // chi²_t < chi² (mcsas.py:379)
                    const unsigned amask = (unsigned)(__ballot(acc_g) & 0xFFull);
                    const unsigned ovm = (unsigned)(__ballot((g < gcount) && sh.povf[sb * K + kg]) & 0xFFull);
                    if (amask == 0u) {
                        if (ovm) overflow = 1;
                        k += gcount; num_iter += gcount;
                        ri += gcount; if (ri >= N) ri -= N;
                        continue;
                    }
                    const int ga = __builtin_ctz(amask);                            // first accepted step of the group
                    if (ovm & ((2u << ga) - 1u)) overflow = 1;
                    const int ka = k + ga;
                    int ria = ri + ga; if (ria >= N) ria -= N;
                    {
                        const double *dr = dbase + (size_t)ka * qpad;
                        double fo[QPL], wv[QPL], dl[QPL];
#pragma unroll
                        for (int j = 0; j < QPL; ++j) { fo[j] = lft[WAVE * j]; wv[j] = sh.lw[lane + WAVE * j]; dl[j] = dr[WAVE * j]; }
#pragma unroll
                        for (int j = 0; j < QPL; ++j) {
                            const double f = fo[j] + dl[j];
                            lft[WAVE * j] = f;
                            lwft[WAVE * j] = wv[j] * f;
                        }
                    }
                    SC = readlane_f64(SCt, ga); SIC = readlane_f64(SICt, ga); SCC = readlane_f64(SCCt, ga);
                    cur = solve_fit(a, SC, SCC, SIC);
                    X = cur.chi2 * nqd;
                    {
                        const int fresh = sh.stage_slot[sb * K + ka];
                        const int freed = sh.slot_of[ria];
                        if (lane == 0) {
                            sh.slot_of[ria] = fresh; sh.stage_slot[sb * K + ka] = freed;
                            for (int p = 0; p < P; ++p)
                                rset[(size_t)ria * P + p] = sh.pval[((size_t)sb * K + ka) * MCSAS_MAX_ACTIVE + p];
                        }
                    }
                    ++num_moves;
                    touched = true;
                    k = ka + 1; num_iter += ga + 1;
                    ri = ria + 1; if (ri >= N) ri -= N;
                    if (!(cur.chi2 > a.conv_crit)) break;
                }
                if (touched) {
                    // re-sum the fit sums from ft so the incremental updates cannot drift
                    double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        const double f = lft[WAVE * j], wf = lwft[WAVE * j];
                        s1 += wf; s2 = fma(wf, f, s2); s3 = fma(sh.lwI[lane + WAVE * j], f, s3);
                    }
                    wave_sum3(s1, s2, s3);
                    SC = s1; SCC = s2; SIC = s3;
                    cur = solve_fit(a, SC, SCC, SIC);
                    X = cur.chi2 * nqd;
                }
                if (!(cur.chi2 > a.conv_crit) || !(num_iter < a.max_iter) || stopped) { live = false; sh.ctl[CTL_DONE + buf] = 1; }
            }
            if ((!chain_runs || a.max_iter <= 0) && lane == 0) sh.ctl[CTL_DONE + buf] = 1;
            __syncthreads();                            // BW
            if (sh.ctl[CTL_DONE + buf]) break;
        }

        // ------------------------------------------------------------ end of attempt (mcsas.py:424-426)
        if (lane == 0) { sh.ctl[CTL_NUM_ITER] = (int32_t)num_iter; sh.ctl[CTL_NUM_ITER + 1] = (int32_t)(num_iter >> 32); }
        total_steps += num_iter;
        {
            double ft[QPL];
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                ft[j] = lft[WAVE * j];
                const double wf = sh.lw[lane + WAVE * j] * ft[j];
                s1 += wf; s2 = fma(wf, ft[j], s2); s3 = fma(sh.lwI[lane + WAVE * j], ft[j], s3);
            }
            wave_sum3(s1, s2, s3);
            cur = solve_fit(a, s1, s2, s3);
            double rs = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const int i = lane + WAVE * j;
                const double r = a.I[i] - (ft[j] * cur.A + cur.b);
                rs += sh.lw[i] * r * r;
            }
            cur.chi2 = wave_sum(rs) / nqd;              // chiSqr, backgroundscalingfit.py:72-77
        }
        converged = !(cur.chi2 > a.conv_crit);
        if (lane == 0) sh.ctl[CTL_FINISHED] = (converged || stopped) ? 1 : 0;
        __syncthreads();                                // B2
        draw_pos = step_base + (uint64_t)num_iter * P;
        __syncthreads();                                // B3
        if (converged || stopped) break;
    }

#pragma unroll
    for (int j = 0; j < QPL; ++j)
        a.fit[(size_t)rep * qpad + lane + WAVE * j] = lft[WAVE * j] * cur.A + cur.b;   // ifinal*sc[0]+sc[1]
    overflow = __any(overflow) | sh.ctl[CTL_OVF];
    if (lane == 0) {
        ChainOut o;
        o.chisq = cur.chi2; o.scaling = cur.A; o.background = cur.b;
        o.seconds = (double)(wall_clock64() - t_start) * 1e-8;
        o.num_iter = num_iter; o.num_moves = num_moves; o.draws = (int64_t)draw_pos;
        o.total_steps = total_steps;
        o.attempts = attempts; o.converged = converged; o.stream_overflow = overflow; o.stopped = stopped;
        a.out[rep] = o;
    }
}

template <int M, int QPL>
__global__ __launch_bounds__(WG_MAX_WAVES * 64) void chain_wg_kernel(const ChainArgs a, const WgGeom g) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const WgShared sh = wg_carve(lds, g, a.qpad);
    const int T = g.waves * WAVE;
    for (int i = tid; i < a.qpad; i += T) {
        const double qq = a.q[i];
        sh.lq[i] = qq; sh.lw[i] = a.w[i]; sh.lwI[i] = a.wI[i]; sh.lq3[i] = 1.0 / (qq * qq * qq);
    }
    Contrib<M>::fill_table(a.model, sh.tab, tid, T);
    // (the first barrier of either role publishes the tables)
    if ((tid >> 6) == 0) wg_scanner<M, QPL>(a, g, sh, blockIdx.x);
    else wg_producer<M, QPL>(a, g, sh, blockIdx.x);
}

}  // namespace mcsas
