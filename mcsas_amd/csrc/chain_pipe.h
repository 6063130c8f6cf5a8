// chain_pipe.h — the speculative proposal window of chain_wg.h spread over the WHOLE chip for runs
// with fewer chains than CUs (BASELINE config 2: 50 chains on 256 CUs).
//
// Two kernels per "tick", on two HIP streams:
//   pipe_prod_kernel  (grid = chains x row blocks, every CU): the form-factor rows of one window of
//       Kb steps per chain -> `new` row into a spare HBM row slot, d = new - old and the three
//       ft-independent sums into the window buffer in HBM;
//   pipe_scan_kernel  (grid = chains, one workgroup each): loader waves stage the window's d rows
//       HBM -> LDS in sub-windows, the scanner wave makes the accept/reject decisions eight steps at
//       a time (same arithmetic as chain_wg.h).
// PROD(t+1) runs concurrently with SCAN(t): a window's rows depend only on the random stream and
// on row slots settled two windows earlier (2*Kb <= N), never on the decisions of the window
// before.  Dependencies between launches are HIP events; there are no in-kernel spin waits.
//
// Chain schedule: an attempt (one mcFit call) is initialised at tick t_init (PROD evaluates the N
// rows of the initial set, SCAN sums them and fits), then tick t > t_init handles window
// t - t_init - 1.  SCAN(t) publishes the small schedule record PROD(t+2) reads, double-buffered by
// tick parity, so a producer never reads a record that a concurrently running scan is writing.
#pragma once
#include "chain_common.h"

namespace mcsas {

struct PipeSnap {                 // what the producer needs to know about a chain
    int32_t attempt, t_init, alive, pad;
    uint64_t init_base;           // draw index of the initial parameter set of this attempt
    uint64_t step_base;           // draw index of step 0 of this attempt
};

struct PipeChain {                // per-chain scanner state, lives in HBM between ticks
    PipeSnap snap[2];
    double SC, SIC, SCC, A, b, chi2;
    int64_t num_iter, num_moves, total_steps;
    uint64_t draw_pos, t_start;
    int32_t attempts, converged, stopped, overflow, done, pad;
};

struct PipeGeom {
    int32_t kb;                   // steps per window (tick)
    int32_t ks;                   // steps per LDS sub-window in the scan kernel
    int32_t rows_per_wave;        // producer: rows per wave
    int32_t prod_blocks_y;        // producer grid.y
    int32_t scan_waves;           // scan kernel waves (1 scanner + loaders)
    int32_t qpl;
    uint64_t prod_lds, scan_lds;
};

struct PipeArgs {
    ChainArgs c;                  // c.cache_rows = N + 2*kb
    PipeGeom g;
    PipeChain *chains;            // [R]
    double *ft, *wft;             // [R][qpad]
    int32_t *slot_of;             // [R][N]
    int32_t *stage_slot;          // [R][2][kb]
    double *dwin;                 // [R][2][kb][qpad]
    double *scal;                 // [R][2][kb][4]
    double *pval;                 // [R][2][kb][MAX_ACTIVE]
    int32_t *povf;                // [R][2][kb]
    int32_t *n_done;              // host-mapped: number of finished chains
    int32_t tick;
    int32_t pad;
};

static inline int pipe_geometry(int nq, int n_contrib, int tab_doubles, PipeGeom *g) {
    int qpl = 1;
    while (qpl * 64 < nq) qpl *= 2;
    if (qpl > 16) return 1;
    const int qpad = qpl * 64;
    int kb = 128;
    while (kb > 8 && 2 * kb > n_contrib) kb /= 2;
    if (2 * kb > n_contrib) return 1;
    g->kb = kb; g->qpl = qpl;
    g->rows_per_wave = kb >= 32 ? 8 : (kb >= 16 ? 4 : 2);
    g->prod_blocks_y = kb / (4 * g->rows_per_wave);
    if (g->prod_blocks_y < 1) { g->prod_blocks_y = 1; g->rows_per_wave = kb / 4; }
    g->prod_lds = sizeof(double) * (4 * (size_t)qpad + tab_doubles);
    // scan: w, wI, ft, wft + 2 sub-window buffers of ks rows + scalars
    int ks = 16;
    while (ks > 8 && sizeof(double) * ((4 + 2 * (size_t)ks) * qpad + 2 * ks * 4) + 1024 > 150 * 1024) ks /= 2;
    if (ks > kb) ks = kb;
    g->ks = ks; g->scan_waves = 4;
    g->scan_lds = sizeof(double) * ((4 + 2 * (size_t)ks) * qpad + 2 * ks * 4) + 2 * ks * 4 + 64;
    if (g->scan_lds > 160 * 1024) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------ producer
template <int M, int QPL>
__global__ __launch_bounds__(256) void pipe_prod_kernel(const PipeArgs pa) {
    extern __shared__ double lds[];
    const ChainArgs &a = pa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rep = blockIdx.x, by = blockIdx.y, t = pa.tick;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad, Kb = pa.g.kb;
    const PipeSnap sn = pa.chains[rep].snap[t & 1];
    if (!sn.alive || t < sn.t_init) return;

    double *lq = lds, *lw = lds + qpad, *lwI = lds + 2 * qpad, *lq3 = lds + 3 * qpad, *tab = lds + 4 * qpad;
    for (int i = tid; i < qpad; i += 256) {
        const double qq = a.q[i];
        lq[i] = qq; lw[i] = a.w[i]; lwI[i] = a.wI[i]; lq3[i] = 1.0 / (qq * qq * qq);
    }
    Contrib<M>::fill_table(a.model, tab, tid, 256);
    __syncthreads();
    const QTables qt{lq, lq3, tab};
    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * a.cache_rows * qpad;
    const DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                         (uint32_t)(a.rep_offset + rep)};
    int32_t *slot_of = pa.slot_of + (size_t)rep * N;
    int32_t *stage = pa.stage_slot + (size_t)rep * 2 * Kb;
    const int gw = by * 4 + wave, nw = gridDim.y * 4;        // this wave's index among the chain's producer waves

    if (t == sn.t_init) {
        // ---- initial parameter set of the attempt (mcsas.py:310-319): rows n = gw*64 + lane + 64*nw*i
        for (int i = tid + by * 256; i < N; i += 256 * gridDim.y) slot_of[i] = i;
        for (int i = tid + by * 256; i < 2 * Kb; i += 256 * gridDim.y) stage[i] = N + i;
        int ovf = 0;
        // contribution n = lane*nw + gw + 64*nw*i: every producer wave of the chain owns ~N/nw rows
        for (int nb = 0; nb < N; nb += nw * WAVE) {
            const int n = nb + lane * nw + gw;
            double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
            if (n < N) {
#pragma unroll
                for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                    if (p < P) {
                        if (a.start_from_min) row[p] = a.start_value[p];
                        else {
                            double u = src.at(sn.init_base + (uint64_t)p * N + n, ovf);
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
                const int nn = nb + l * nw + gw;
                if (nn >= N) break;
                const Contrib<M> c = mine.bcast(__builtin_amdgcn_readfirstlane(l));
                double it[QPL];
                RowEval<M, QPL>::run(c, qt, lane, it);
#pragma unroll
                for (int j = 0; j < QPL; ++j) cache[(size_t)nn * qpad + lane + WAVE * j] = it[j];
            }
        }
        if (__any(ovf) && lane == 0) atomicOr(&pa.chains[rep].overflow, 1);
        return;
    }

    // ---- window w of the attempt: this wave's rows k = gw*rpw .. +rpw-1, one proposal per lane
    const int64_t w = (int64_t)t - sn.t_init - 1;
    const int rpw = pa.g.rows_per_wave, buf = t & 1;
    const int k0 = gw * rpw;
    if (k0 >= Kb) return;
    const int64_t s0 = w * Kb + k0;
    double prow[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
    int pov = 0;
    {
        const int64_t sl = s0 + lane;
#pragma unroll
        for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
            if (p < P) {
                double u = 0.5;
                if (lane < rpw && sl < a.max_iter) u = src.at(sn.step_base + (uint64_t)sl * P + p, pov);
                prow[p] = gen_transform(a.gen_kind[p], u) * (a.gen_hi[p] - a.gen_lo[p]) + a.gen_lo[p];
            }
    }
    Contrib<M> prop;
    prop.prepare(a.model, prow);
    double *dwin = pa.dwin + ((size_t)rep * 2 + buf) * Kb * qpad;
    double *scal = pa.scal + ((size_t)rep * 2 + buf) * Kb * 4;
    double *pval = pa.pval + ((size_t)rep * 2 + buf) * Kb * MCSAS_MAX_ACTIVE;
    int32_t *povf = pa.povf + ((size_t)rep * 2 + buf) * Kb;
    int ri = (int)(s0 % N);
    for (int i = 0; i < rpw; ++i) {
        const int k = k0 + i;
        if (s0 + i >= a.max_iter) break;
        const int bl = __builtin_amdgcn_readfirstlane(i);
        const Contrib<M> cnew = prop.bcast(bl);
        const int oslot = slot_of[ri], sslot = stage[buf * Kb + k];
        const double *orow = cache + (size_t)oslot * qpad + lane;
        double *nrow = cache + (size_t)sslot * qpad + lane;
        double *dr = dwin + (size_t)k * qpad + lane;
        double d[QPL], nwv[QPL];
#pragma unroll
        for (int j = 0; j < QPL; ++j) d[j] = orow[WAVE * j];
        RowEval<M, QPL>::run(cnew, qt, lane, nwv);
        double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const int iq = lane + WAVE * j;
            nrow[WAVE * j] = nwv[j];
            d[j] = nwv[j] - d[j];
            dr[WAVE * j] = d[j];
            const double wd = lw[iq] * d[j];
            s1 += wd; s2 += lwI[iq] * d[j]; s3 += wd * d[j];
        }
        wave_sum3(s1, s2, s3);
        if (lane == 0) { scal[k * 4 + 0] = s1; scal[k * 4 + 1] = s2; scal[k * 4 + 2] = s3; }
#pragma unroll
        for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
            if (p < P) {
                const double v = readlane_f64(prow[p], bl);
                if (lane == 0) pval[k * MCSAS_MAX_ACTIVE + p] = v;
            }
        const int ov = __builtin_amdgcn_readlane(pov, bl);
        if (lane == 0) povf[k] = ov;
        ri = (ri + 1 == N) ? 0 : ri + 1;
    }
}

// ------------------------------------------------------------------------------------ scanner
// LDS: lw, lwI, lft, lwft [qpad each]; dsub[2][ks][qpad]; ssub[2][ks][4]; osub[2][ks] (int)
template <int QPL>
__global__ __launch_bounds__(256) void pipe_scan_kernel(const PipeArgs pa) {
    extern __shared__ double lds[];
    const ChainArgs &a = pa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rep = blockIdx.x, t = pa.tick;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad, Kb = pa.g.kb, Ks = pa.g.ks;
    const int NW = pa.g.scan_waves, NL = NW - 1, T = NW * WAVE;
    PipeChain &ch = pa.chains[rep];
    if (ch.done) return;                                      // uniform for the block
    const PipeSnap sn = ch.snap[(t + 1) & 1];                 // the record in force for tick t (written at t-1; host for t = 0)

    double *lw = lds, *lwI = lds + qpad, *lft = lds + 2 * qpad, *lwft = lds + 3 * qpad;
    double *dsub = lds + 4 * qpad;                            // [2][Ks][qpad]
    double *ssub = dsub + 2 * (size_t)Ks * qpad;              // [2][Ks][4]
    int32_t *osub = reinterpret_cast<int32_t *>(ssub + 2 * Ks * 4);   // [2][Ks]
    int32_t *ctl = osub + 2 * Ks;                             // [4]
    double *gft = pa.ft + (size_t)rep * qpad, *gwft = pa.wft + (size_t)rep * qpad;
    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * a.cache_rows * qpad;
    const int buf = t & 1;
    const double *dwin = pa.dwin + ((size_t)rep * 2 + buf) * Kb * qpad;
    const double *scal = pa.scal + ((size_t)rep * 2 + buf) * Kb * 4;
    const double *pval = pa.pval + ((size_t)rep * 2 + buf) * Kb * MCSAS_MAX_ACTIVE;
    const int32_t *povf = pa.povf + ((size_t)rep * 2 + buf) * Kb;
    int32_t *slot_of = pa.slot_of + (size_t)rep * N;
    int32_t *stage = pa.stage_slot + ((size_t)rep * 2 + buf) * Kb;
    const double nqd = (double)a.nq;

    for (int i = tid; i < qpad; i += T) { lw[i] = a.w[i]; lwI[i] = a.wI[i]; }
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }

    // scanner-side chain state (meaningful in wave 0)
    FitResult cur{ch.A, ch.b, ch.chi2};
    double SC = ch.SC, SIC = ch.SIC, SCC = ch.SCC;
    int64_t num_iter = ch.num_iter, num_moves = ch.num_moves;
    int stopped = ch.stopped, overflow = 0;
    bool attempt_over = false;

    if (t < sn.t_init) {
        // nothing scheduled for this chain at this tick; just republish below
    } else if (t == sn.t_init) {
        // ---- model.calc over the initial set: rows summed in contribution order (scatteringmodel.py:90-101)
        if (wave == 0) {
            double ft[QPL];
#pragma unroll
            for (int j = 0; j < QPL; ++j) ft[j] = 0.;
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int j = 0; j < QPL; ++j) ft[j] += cache[(size_t)n * qpad + lane + WAVE * j];
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const double wf = a.w[lane + WAVE * j] * ft[j];
                s1 += wf; s2 += wf * ft[j]; s3 += a.wI[lane + WAVE * j] * ft[j];
                gft[lane + WAVE * j] = ft[j]; gwft[lane + WAVE * j] = wf;
            }
            wave_sum3(s1, s2, s3);
            SC = s1; SCC = s2; SIC = s3;
            cur = solve_fit(a, SC, SCC, SIC);
            num_iter = 0; num_moves = 0;
            if (N <= 1 || a.max_iter <= 0 || !(cur.chi2 > a.conv_crit)) attempt_over = true;
        }
    } else {
        // ---- window w = t - t_init - 1: loaders stage sub-windows, the scanner decides
        for (int i = tid; i < qpad; i += T) { lft[i] = gft[i]; lwft[i] = gwft[i]; }
        const int64_t w = (int64_t)t - sn.t_init - 1;
        const int64_t budget = a.max_iter - w * Kb;
        const int kmax_all = budget < Kb ? (budget < 0 ? 0 : (int)budget) : Kb;
        const int nsub = (kmax_all + Ks - 1) / Ks;
        const double invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
        double X = cur.chi2 * nqd;
        bool touched = false, live = true;
        int ri = (int)((w * Kb) % N);
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);
            if (a.stop_flag && __hip_atomic_load(a.stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) stopped = 1;
        }
        __syncthreads();
        for (int sw = 0; sw <= nsub; ++sw) {
            const int sbuf = sw & 1;
            if (wave > 0 && sw < nsub) {
                // loaders: sub-window sw -> LDS (rows are 8*qpad bytes, contiguous)
                const int kbeg = sw * Ks, kend = min(kmax_all, kbeg + Ks);
                const int rows = kend - kbeg;
                const double *src = dwin + (size_t)kbeg * qpad;
                double *dst = dsub + (size_t)sbuf * Ks * qpad;
                for (int i = tid - WAVE; i < rows * qpad; i += NL * WAVE) dst[i] = src[i];
                for (int i = tid - WAVE; i < rows * 4; i += NL * WAVE) ssub[sbuf * Ks * 4 + i] = scal[kbeg * 4 + i];
                for (int i = tid - WAVE; i < rows; i += NL * WAVE) osub[sbuf * Ks + i] = povf[kbeg + i];
            }
            if (wave == 0 && sw > 0 && live) {
                const int pb = sbuf ^ 1;                      // sub-window sw-1
                const int kbeg = (sw - 1) * Ks;
                const int kmax = min(kmax_all, kbeg + Ks) - kbeg;
                const double *dbase = dsub + (size_t)pb * Ks * qpad + lane;
                const double *sbase = ssub + pb * Ks * 4;
                const int32_t *obase = osub + pb * Ks;
                const int g = lane & 7;
                int k = 0;
                if (!(cur.chi2 > a.conv_crit) || stopped) { live = false; k = kmax; }
                while (k < kmax) {
                    const int gcount = (kmax - k) < 8 ? (kmax - k) : 8;
                    double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        const double wf = lwft[lane + WAVE * j];
#pragma unroll
                        for (int gg = 0; gg < 8; ++gg) {
                            const int kk = (k + gg < Ks) ? k + gg : Ks - 1;
                            const int slot = ((gg & 1) << 2) | (gg & 2) | ((gg >> 2) & 1);
                            acc[slot] = fma(wf, dbase[(size_t)kk * qpad + WAVE * j], acc[slot]);
                        }
                    }
                    const double h = wave_sum8_transposed(acc, lane);
                    const int kg = (k + g < Ks) ? k + g : Ks - 1;
                    const double *sc = sbase + kg * 4;
                    const double SCt = SC + sc[0], SICt = SIC + sc[1], SCCt = SCC + (2. * h + sc[2]);
                    double S = a.SII, num = SICt, den = SCCt;
                    if (a.find_bg) {
                        const double numc = SICt - SIoSw * SCt, denc = SCCt - SCt * invSw * SCt;
                        const bool neg_b = a.pos_bg && (a.SI * denc - numc * SCt < 0.);
                        if (!neg_b) { S = Scen; num = numc; den = denc; }
                    }
                    const bool acc_g = (g < gcount) && (num * num > (S - X) * den);
                    const unsigned amask = (unsigned)(__ballot(acc_g) & 0xFFull);
                    const unsigned ovm = (unsigned)(__ballot((g < gcount) && obase[kg]) & 0xFFull);
                    if (amask == 0u) {
                        if (ovm) overflow = 1;
                        k += gcount; num_iter += gcount;
                        ri += gcount; if (ri >= N) ri -= N;
                        continue;
                    }
                    const int ga = __builtin_ctz(amask);
                    if (ovm & ((2u << ga) - 1u)) overflow = 1;
                    const int ka = k + ga;
                    int ria = ri + ga; if (ria >= N) ria -= N;
                    {
                        const double *dr = dbase + (size_t)ka * qpad;
                        double fo[QPL], wv[QPL], dl[QPL];
#pragma unroll
                        for (int j = 0; j < QPL; ++j) { fo[j] = lft[lane + WAVE * j]; wv[j] = lw[lane + WAVE * j]; dl[j] = dr[WAVE * j]; }
#pragma unroll
                        for (int j = 0; j < QPL; ++j) {
                            const double f = fo[j] + dl[j];
                            lft[lane + WAVE * j] = f;
                            lwft[lane + WAVE * j] = wv[j] * f;
                        }
                    }
                    SC = readlane_f64(SCt, ga); SIC = readlane_f64(SICt, ga); SCC = readlane_f64(SCCt, ga);
                    cur = solve_fit(a, SC, SCC, SIC);
                    X = cur.chi2 * nqd;
                    {
                        const int kglob = kbeg + ka;
                        const int fresh = stage[kglob];
                        const int freed = slot_of[ria];
                        if (lane == 0) {
                            slot_of[ria] = fresh; stage[kglob] = freed;
                            for (int p = 0; p < P; ++p) rset[(size_t)ria * P + p] = pval[(size_t)kglob * MCSAS_MAX_ACTIVE + p];
                        }
                    }
                    ++num_moves;
                    touched = true;
                    k = ka + 1; num_iter += ga + 1;
                    ri = ria + 1; if (ri >= N) ri -= N;
                    if (!(cur.chi2 > a.conv_crit)) { live = false; break; }
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
            if (touched) {
                // re-sum the fit sums from ft so the incremental updates cannot drift; park ft in HBM
                double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    const double f = lft[lane + WAVE * j], wf = lwft[lane + WAVE * j];
                    s1 += wf; s2 += wf * f; s3 += lwI[lane + WAVE * j] * f;
                    gft[lane + WAVE * j] = f; gwft[lane + WAVE * j] = wf;
                }
                wave_sum3(s1, s2, s3);
                SC = s1; SCC = s2; SIC = s3;
                cur = solve_fit(a, SC, SCC, SIC);
            }
            if (!(cur.chi2 > a.conv_crit) || !(num_iter < a.max_iter) || stopped) attempt_over = true;
        }
    }

    // ---- bookkeeping by the scanner wave: end of attempt (mcsas.py:424-439), schedule record for t+2
    if (wave == 0) {
        PipeSnap next = sn;
        int done = 0;
        uint64_t draw_pos = ch.draw_pos;
        int64_t total_steps = ch.total_steps;
        int attempts = ch.attempts, converged = ch.converged;
        if (attempt_over) {
            double ft[QPL];
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                ft[j] = gft[lane + WAVE * j];
                const double wf = a.w[lane + WAVE * j] * ft[j];
                s1 += wf; s2 += wf * ft[j]; s3 += a.wI[lane + WAVE * j] * ft[j];
            }
            wave_sum3(s1, s2, s3);
            cur = solve_fit(a, s1, s2, s3);
            double rs = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const int i = lane + WAVE * j;
                const double r = a.I[i] - (ft[j] * cur.A + cur.b);
                rs += a.w[i] * r * r;
            }
            cur.chi2 = wave_sum(rs) / nqd;                    // chiSqr, backgroundscalingfit.py:72-77
            converged = !(cur.chi2 > a.conv_crit);
            total_steps += num_iter;
            draw_pos = sn.step_base + (uint64_t)num_iter * P;
            if (converged || stopped || sn.attempt >= a.max_retries) {
                done = 1;
#pragma unroll
                for (int j = 0; j < QPL; ++j)
                    a.fit[(size_t)rep * qpad + lane + WAVE * j] = ft[j] * cur.A + cur.b;
                next.alive = 0;
            } else {
                ++attempts;
                next.attempt = sn.attempt + 1;
                next.t_init = t + 2;
                next.init_base = draw_pos;
                next.step_base = draw_pos + (a.start_from_min ? 0 : (uint64_t)N * P);
                next.alive = 1;
            }
        }
        overflow = __any(overflow);
        if (lane == 0) {
            ch.snap[t & 1] = next;                            // read by PROD(t+2) and SCAN(t+1)
            ch.SC = SC; ch.SIC = SIC; ch.SCC = SCC; ch.A = cur.A; ch.b = cur.b; ch.chi2 = cur.chi2;
            ch.num_iter = num_iter; ch.num_moves = num_moves; ch.total_steps = total_steps;
            ch.draw_pos = draw_pos; ch.attempts = attempts; ch.converged = converged; ch.stopped = stopped;
            if (overflow) atomicOr(&ch.overflow, 1);
            if (done) {
                ch.done = 1;
                ChainOut o;
                o.chisq = cur.chi2; o.scaling = cur.A; o.background = cur.b;
                o.seconds = (double)(wall_clock64() - ch.t_start) * 1e-8;
                o.num_iter = num_iter; o.num_moves = num_moves; o.draws = (int64_t)draw_pos;
                o.total_steps = total_steps;
                o.attempts = attempts; o.converged = converged; o.stream_overflow = ch.overflow | overflow; o.stopped = stopped;
                a.out[rep] = o;
                __hip_atomic_fetch_add(pa.n_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// first tick bookkeeping: schedule records for ticks 0 and 1, counters
template <int UNUSED>
__global__ void pipe_reset_kernel(const PipeArgs pa) {
    const int rep = blockIdx.x * blockDim.x + threadIdx.x;
    if (rep >= pa.c.n_reps) return;
    const ChainArgs &a = pa.c;
    PipeChain ch{};
    PipeSnap s{};
    s.attempt = 0; s.t_init = 0; s.alive = 1;
    s.init_base = 0;
    s.step_base = a.start_from_min ? 0 : (uint64_t)a.n_contrib * a.model.n_active;
    ch.snap[0] = s; ch.snap[1] = s;
    ch.attempts = 1; ch.chi2 = 0.; ch.A = 1.; ch.t_start = wall_clock64();
    pa.chains[rep] = ch;
}

}  // namespace mcsas
