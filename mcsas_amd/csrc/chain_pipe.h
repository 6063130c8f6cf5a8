// chain_pipe.h — the speculative proposal window of chain_wg.h spread over the WHOLE chip for runs
// with fewer chains than CUs (BASELINE config 2: 50 chains on 256 CUs).
//
// One launch per "tick" t (pipe_tick_kernel), two kinds of workgroup in its grid:
//   scan blocks (one per chain, block ids 0..R-1) do SCAN(t): eight waves stream window t's d rows
//       HBM -> private LDS rings with LDS-DMA and reduce h = sum (w ft) d for their rows; wave 0 takes
//       the accept/reject decisions 8 or 16 steps at a time (same arithmetic as chain_wg.h);
//   producer blocks (the rest of the grid, every CU) do PROD(t+1): the form-factor rows of the NEXT
//       window of Kb steps per chain -> `new` row into a spare HBM row slot, d = new - old and the
//       three ft-independent sums into the window buffer in HBM.
// The two run concurrently inside one launch because a window's rows depend only on the random
// stream and on row slots settled two windows earlier (2*Kb <= N), never on the decisions of the
// window before.  Launch t+1 follows launch t on the same stream: the kernel boundary is the only
// synchronisation; there are no in-kernel spin waits and no cross-workgroup flags.
//
// Chain schedule: an attempt (one mcFit call) is initialised at tick t_init (PROD evaluates the N
// rows of the initial set, SCAN sums them and fits), then tick t > t_init handles window
// t - t_init - 1.  SCAN(t) publishes the small schedule record PROD(t+2) reads, double-buffered by
// tick parity, so a producer never reads a record that a concurrently running scan is writing.
#pragma once
#include "chain_common.h"

namespace mcsas {

// Keep a wave-uniform double in a VGPR: the scan loop has far more uniform fp64 state than the 102
// SGPRs can hold, and spilled SGPRs come back one v_readlane at a time on the critical path.
#define MCSAS_IN_VGPR(x) asm volatile("" : "+v"(x))

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
#ifdef MCSAS_STAMPS
    int64_t dbg[20];
    uint64_t last_end;            // wall clock (10 ns) at the end of this chain's previous scan block
#endif
};

struct PipeGeom {
    int32_t kb;                   // steps per window (tick)
    int32_t ks;                   // steps per decision group after an accepted move (8; 16 after a group without one when the ring holds it)
    int32_t rows_per_wave;        // producer: rows per wave
    int32_t prod_blocks_y;        // producer grid.y
    int32_t scan_waves;           // scan kernel waves (1 scanner + loaders)
    int32_t qpl;
    int32_t ring, pad;            // rows each scan wave keeps in its private LDS ring (4, 2 or 1)
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
    int32_t tick, pad;            // unused: the tick travels as its own kernel argument
};

constexpr int PIPE_BLOCK = 512;      // threads per workgroup of the tick kernel (8 waves)
constexpr int PIPE_RING = 4;         // most rows a scan wave keeps in flight in its private LDS ring

static inline int pipe_geometry(int nq, int n_contrib, int tab_doubles, int heavy_rows, PipeGeom *g) {
    int qpl = 1;
    while (qpl * 64 < nq) qpl *= 2;
    if (qpl > 16) return 1;
    const int qpad = qpl * 64;
    // window: as many steps as 2*Kb <= N allows (a multiple of the 8 producer waves x rows per wave)
    int rpw = 8;
    while (rpw > 1 && 2 * 8 * rpw > n_contrib) rpw /= 2;
    if (2 * 8 * rpw > n_contrib) return 1;
    int by = n_contrib / (2 * 8 * rpw);
    if (by * 8 * rpw > 256) by = 256 / (8 * rpw);
    g->kb = by * 8 * rpw; g->qpl = qpl;
    // rows that cost a numerical integration each (cylinders, ellipsoids, worm-like chains): one row per
    // producer wave, so that a window is R*Kb waves for the 1024 SIMDs instead of R*Kb/8
    if (heavy_rows) { by *= rpw; rpw = 1; }
    g->rows_per_wave = rpw;
    g->prod_blocks_y = by;
    g->prod_lds = sizeof(double) * (4 * (size_t)qpad + tab_doubles);
    // scan block LDS: one private ring of `ring` rows per wave + the window's scalars/tables
    g->ks = 8; g->scan_waves = PIPE_BLOCK / 64;
    for (int ring = PIPE_RING; ring >= 1; ring /= 2) {
        g->ring = ring;
        g->scan_lds = sizeof(double) * ((size_t)g->scan_waves * ring * qpad + (size_t)g->kb * 4 + 16)
                    + sizeof(int32_t) * (4 * g->kb + 32) + 64;
        if (g->scan_lds <= 160 * 1024) break;
    }
    if (g->scan_lds > 160 * 1024) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------ producer
template <int M, int QPL>
__device__ __forceinline__ void pipe_prod_block(const PipeArgs &pa, double *lds, int rep, int by, int gy, int t) {
    const ChainArgs &a = pa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WPB = PIPE_BLOCK / 64;
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad, Kb = pa.g.kb;
    const PipeSnap sn = pa.chains[rep].snap[t & 1];
    if (!sn.alive || t < sn.t_init) return;

    double *lq = lds, *lw = lds + qpad, *lwI = lds + 2 * qpad, *lq3 = lds + 3 * qpad, *tab = lds + 4 * qpad;
    for (int i = tid; i < qpad; i += PIPE_BLOCK) {
        const double qq = a.q[i];
        lq[i] = qq; lw[i] = a.w[i]; lwI[i] = a.wI[i]; lq3[i] = 1.0 / (qq * qq * qq);
    }
    Contrib<M>::fill_table(a.model, tab, tid, PIPE_BLOCK);
    __syncthreads();
    const QTables qt = make_qtables<M>(a.model, lq, lq3, tab);
    double *rset = a.rset + (size_t)rep * N * P;
    double *cache = a.cache + (size_t)rep * a.cache_rows * qpad;
    const DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                         (uint32_t)(a.rep_offset + rep)};
    int32_t *slot_of = pa.slot_of + (size_t)rep * N;
    int32_t *stage = pa.stage_slot + (size_t)rep * 2 * Kb;
    const int gw = by * WPB + wave, nw = gy * WPB;          // this wave's index among the chain's producer waves

    if (t == sn.t_init) {
        // ---- initial parameter set of the attempt (mcsas.py:310-319): rows n = gw*64 + lane + 64*nw*i
        for (int i = tid + by * PIPE_BLOCK; i < N; i += PIPE_BLOCK * gy) slot_of[i] = i;
        for (int i = tid + by * PIPE_BLOCK; i < 2 * Kb; i += PIPE_BLOCK * gy) stage[i] = N + i;
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

    if (a.pad0 & 16) return;                                  // diagnostic: no window rows
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
// LDS: one ring of `ring` rows per wave; the window's scalars, flags and slot tables; control words
template <int QPL>
__device__ __forceinline__ void pipe_scan_block(const PipeArgs &pa, double *lds, int rep, int t, int stop_now) {
    const ChainArgs &a = pa.c;
    // the wave index is wave-uniform: keep it (and the row bookkeeping that hangs on it) on the scalar unit
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad, Kb = pa.g.kb;
    const int NW = pa.g.scan_waves, T = NW * WAVE;
    PipeChain &ch = pa.chains[rep];
    if (ch.done) return;                                      // uniform for the block
    MCSAS_STAMP_DECL(sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0);
    MCSAS_STAMP(sb0);
#ifdef MCSAS_STAMPS
    const uint64_t wc0 = wall_clock64();
#endif
    const PipeSnap sn = ch.snap[(t + 1) & 1];                 // the record in force for tick t (written at t-1; host for t = 0)

    const int RING = pa.g.ring, RMASK = RING - 1;             // ring depth is 4, 2 or 1
    double *ring = lds + (size_t)wave * RING * qpad;                       // this wave's RING rows
    double *ssub = lds + (size_t)NW * RING * qpad;                   // [Kb][4] scalars of the whole window
    double *hbuf = ssub + (size_t)Kb * 4;                                  // [8] h of the current group, by step offset
    int32_t *osub = reinterpret_cast<int32_t *>(hbuf + 16);                // [Kb] replay-overflow flags
    int32_t *lstage = osub + Kb, *lslot = lstage + Kb;          // [Kb] spare row slot of step k / row slot of its contribution
    int32_t *lacc = lslot + Kb;                                // [Kb + 1] accepted steps of this window, count in lacc[Kb]
    int32_t *ctl = lacc + Kb + 1;                              // [k_next, accepted row or -1, live]
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
        // ---- window w = t - t_init - 1.  Eight symmetric waves: wave v owns the window's rows r = v (mod 8),
        // streams them from HBM into its private LDS ring with LDS-DMA (PIPE_RING rows in flight, no
        // registers), computes h = Σ (w ft) d for its rows of the current group of steps, and wave 0
        // decides the group.
        const int64_t w = (int64_t)t - sn.t_init - 1;
        const int64_t budget = a.max_iter - w * Kb;
        const int kmax_all = budget < Kb ? (budget < 0 ? 0 : (int)budget) : Kb;
        const int ri0 = (int)((w * Kb) % N);
        const bool dbg_noload = a.pad0 & 4, dbg_noscan = a.pad0 & 8;
        typedef __attribute__((address_space(3))) void *lds_vp;
        typedef __attribute__((address_space(1))) const void *glb_vp;
        constexpr int CALLS = (QPL >= 2) ? QPL / 2 : 1;        // 1 KB DMA calls per row (QPL = 1: half a call, 32 lanes)
        const bool dma_lane = (QPL >= 2) || lane < 32;
        auto issue_row = [&](int m) {                          // m-th row of this wave: r = wave + 8 m
            const char *gsrc = reinterpret_cast<const char *>(dwin + (size_t)(wave + 8 * m) * qpad) + lane * 16;
            double *ldst = ring + (size_t)(m & RMASK) * qpad;
            if (dma_lane) {
#pragma unroll
                for (int c = 0; c < CALLS; ++c)
                    __builtin_amdgcn_global_load_lds((glb_vp)(gsrc + c * 1024), (lds_vp)(ldst + c * 128), 16, 0, 0);
            }
        };
        // The whole prologue is ONE memory round trip: the first ring rows go out first, then every
        // other load of the block, and only then the single wait.
        const int my_rows = (kmax_all > wave) ? (kmax_all - wave + 7) / 8 : 0;   // rows this wave owns
        int m_issue = 0, m_cur = 0;
        if (!dbg_noload)
            for (; m_issue < RING && m_issue < my_rows; ++m_issue) issue_row(m_issue);
        // every wave keeps its own copy of ft, w*ft and w in registers and applies accepted rows to it
        // itself (same two operations in every wave, so the copies stay bit-identical)
        double wftr[QPL], ftr[QPL], wr[QPL], wIr[QPL];
#pragma unroll
        for (int j = 0; j < QPL; ++j) {
            const int i = lane + WAVE * j;
            wr[j] = a.w[i]; wIr[j] = a.wI[i]; ftr[j] = gft[i]; wftr[j] = gwft[i];
        }
        {
            const int n4 = kmax_all * 4;                       // Kb <= 256: at most two scalars per thread
            double sv0 = 0., sv1 = 0.;
            int ov = 0, stg = 0, sl = 0;
            if (tid < n4) sv0 = scal[tid];
            if (tid + PIPE_BLOCK < n4) sv1 = scal[tid + PIPE_BLOCK];
            if (tid < kmax_all) {
                ov = povf[tid]; stg = stage[tid];
                int r = ri0 + tid; if (r >= N) r -= N;
                sl = slot_of[r];
            }
            if (tid < n4) ssub[tid] = sv0;
            if (tid + PIPE_BLOCK < n4) ssub[tid + PIPE_BLOCK] = sv1;
            if (tid < kmax_all) { osub[tid] = ov; lstage[tid] = stg; lslot[tid] = sl; }
        }
        if (tid == 0) lacc[Kb] = 0;
        const double invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
        double X = cur.chi2 * nqd;
        bool touched = false, live = true;
        int num_acc_win = 0;
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(1);
            if (stop_now) stopped = 1;                         // McSAS.stop as the host saw it when it launched this tick
            if (lane == 0) ctl[2] = (!(cur.chi2 > a.conv_crit) || stopped) ? 0 : 1;   // `live`, shared by all waves
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int k = 0;
        live = ctl[2] != 0;
        if (dbg_noscan) { num_iter += kmax_all; k = kmax_all; }
        // loop-invariant fit constants and the running sums, pinned in VGPRs (see MCSAS_IN_VGPR)
        double cSII = a.SII, cSI = a.SI, cScen = Scen, cSIoSw = SIoSw, cinvSw = invSw, cCrit = a.conv_crit, cnq = nqd;
        MCSAS_IN_VGPR(cSII); MCSAS_IN_VGPR(cSI); MCSAS_IN_VGPR(cScen); MCSAS_IN_VGPR(cSIoSw); MCSAS_IN_VGPR(cinvSw);
        MCSAS_IN_VGPR(cCrit); MCSAS_IN_VGPR(cnq);
        MCSAS_IN_VGPR(SC); MCSAS_IN_VGPR(SIC); MCSAS_IN_VGPR(SCC); MCSAS_IN_VGPR(X);
        const bool find_bg = a.find_bg, pos_bg = a.pos_bg, never_accept = a.pad0 & 32;
#ifdef MCSAS_STAMPS
        int64_t ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        MCSAS_STAMP_DECL(s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0, s8 = 0, s9 = 0);
        // Steps are decided in groups: 8 after a group that accepted a move, 16 after one that did not
        // (moves come in bursts early in a run and become rare later; a longer group amortises the fixed
        // LDS round trips and barriers).  Wave v contributes its rows r = v (mod 8) of the group.
        const int GMAX = (RING >= 4) ? 16 : 8;
        int G = 8;
        const int g16 = lane & 15;
        MCSAS_STAMP(sb1);
        while (k < kmax_all && live) {
            MCSAS_STAMP(s0);
            const int gcount = (kmax_all - k) < G ? (kmax_all - k) : G;
            // my rows of this group (if any): r0 = the smallest r >= k with r = wave (mod 8), then r0 + 8
            const int r0 = k + ((wave - (k & 7) + 8) & 7);
            const int m0 = r0 >> 3;
            const bool mine = r0 < k + gcount, mine1 = r0 + 8 < k + gcount;
            const int mlast = mine1 ? m0 + 1 : m0;
            // (rows behind the group start are normally retired and the ring refilled AFTER the barrier,
            // while the decision is being taken, so the DMA issue costs the critical path nothing; only a
            // ring too shallow to hold the next row has to catch up here)
            while (mine && m_issue <= mlast && m_cur < m0 && !dbg_noload) {
                ++m_cur;
                if (m_issue < my_rows) { issue_row(m_issue); ++m_issue; }
            }
            MCSAS_STAMP(s1);
            if (mine) {
                // wait until my last row of the group has landed: only younger rows' DMAs may be in flight
                const int younger = m_issue - 1 - mlast;
                if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CALLS) : "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CALLS) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * CALLS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MCSAS_STAMP(s2);
                const double *dr = ring + (size_t)(m0 & RMASK) * qpad + lane;
                const double *dr1 = ring + (size_t)((m0 + 1) & RMASK) * qpad + lane;
                double h0 = 0., h1 = 0., e0 = 0., e1 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; j += 2) {
                    h0 = fma(wftr[j], dr[WAVE * j], h0);
                    if (j + 1 < QPL) h1 = fma(wftr[j + 1], dr[WAVE * (j + 1)], h1);
                }
                if (mine1) {
#pragma unroll
                    for (int j = 0; j < QPL; j += 2) {
                        e0 = fma(wftr[j], dr1[WAVE * j], e0);
                        if (j + 1 < QPL) e1 = fma(wftr[j + 1], dr1[WAVE * (j + 1)], e1);
                    }
                }
                double hs = h0 + h1, es = e0 + e1;
                MCSAS_STAMP(s3);
                if (mine1) wave_sum2(hs, es); else hs = wave_sum(hs);
                MCSAS_STAMP(s4);
                if (lane == 0) { hbuf[r0 - k] = hs; if (mine1) hbuf[r0 + 8 - k] = es; }
            }
            // the scalars of my lane's step do not depend on the other waves: fetch them before the barrier
            const int kg = (k + g16 < kmax_all) ? k + g16 : kmax_all - 1;
            double sc0 = 0., sc1 = 0., sc2 = 0.;
            int ovg = 0;
            if (wave == 0) { const double *sc = ssub + kg * 4; sc0 = sc[0]; sc1 = sc[1]; sc2 = sc[2]; ovg = osub[kg]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MCSAS_STAMP(s5);
            __builtin_amdgcn_s_barrier();                                  // B1: hbuf complete
            MCSAS_STAMP(s6);
            if (wave != 0) {
                // workers: retire the rows behind this group's start and refill the ring while wave 0 decides
                const int m_keep = (k + ((wave - (k & 7) + 8) & 7)) >> 3;   // my first row at or after k
                while (m_cur < m_keep) {
                    ++m_cur;
                    if (m_issue < my_rows && !dbg_noload) { issue_row(m_issue); ++m_issue; }
                }
            }
            if (wave == 0) {
                // lane g decides step k+g (all groups of 16 lanes do the same work)
                const double h = hbuf[g16];
                const double SCt = SC + sc0, SICt = SIC + sc1, SCCt = SCC + (2. * h + sc2);
                // chi²·Q = S - num²/den for the candidate (centred sums when a background is fitted)
                double S = cSII, num = SICt, den = SCCt;
                if (find_bg) {
                    const double numc = SICt - cSIoSw * SCt, denc = SCCt - SCt * cinvSw * SCt;
                    const bool neg_b = pos_bg && (cSI * denc - numc * SCt < 0.);
                    if (!neg_b) { S = cScen; num = numc; den = denc; }
                }
                const bool acc_g = (g16 < gcount) && (num * num > (S - X) * den);   // chi²_t < chi² (mcsas.py:379)
                unsigned amask = (unsigned)(__ballot(acc_g) & 0xFFFFull);
                if (never_accept) amask = 0u;                       // diagnostic: never accept
                const unsigned ovm = (unsigned)(__ballot((g16 < gcount) && ovg) & 0xFFFFull);
                int k_next, acc_row = -1, g_next = GMAX;
                if (amask == 0u) {
                    if (ovm) overflow = 1;
                    k_next = k + gcount; num_iter += gcount;
                } else {
                    const int ga = __builtin_ctz(amask);
                    if (ovm & ((2u << ga) - 1u)) overflow = 1;
                    acc_row = k + ga;
                    g_next = 8;
                    SC = readlane_f64(SCt, ga); SIC = readlane_f64(SICt, ga); SCC = readlane_f64(SCCt, ga);
                    // chi²·Q of the accepted state from the same three numbers the decision used (one
                    // division); scale and background are only needed at the end of the attempt
                    X = readlane_f64(S - num * num / den, ga);
                    MCSAS_IN_VGPR(SC); MCSAS_IN_VGPR(SIC); MCSAS_IN_VGPR(SCC); MCSAS_IN_VGPR(X);
                    cur.chi2 = X / cnq;
                    const int fresh = lstage[acc_row], freed = lslot[acc_row];
                    if (lane == 0) {
                        lslot[acc_row] = fresh; lstage[acc_row] = freed;      // slot swap: rows are never copied
                        lacc[num_acc_win] = acc_row;
                    }
                    ++num_acc_win; ++num_moves;
                    touched = true;
                    k_next = acc_row + 1; num_iter += ga + 1;
                    if (!(X > cCrit * cnq)) live = false;
                }
                if (lane == 0) { ctl[0] = k_next; ctl[1] = acc_row; ctl[2] = live ? 1 : 0; ctl[3] = g_next; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MCSAS_STAMP(s7);
            __builtin_amdgcn_s_barrier();                                  // B2: decision published
            MCSAS_STAMP(s8);
            const int k_next = ctl[0], acc_row = ctl[1];
            live = ctl[2] != 0;
            G = ctl[3];
            if (wave == 0) {
                const int m_keep = (k + ((wave - (k & 7) + 8) & 7)) >> 3;
                while (m_cur < m_keep) {
                    ++m_cur;
                    if (m_issue < my_rows && !dbg_noload) { issue_row(m_issue); ++m_issue; }
                }
            }
            if (acc_row >= 0) {
                // ft += d, w ft refreshed (mcsas.py:381-382): the accepted row sits in its owner's ring,
                // landed before B1 and not refilled before the next B1
                const double *dr = lds + ((size_t)(acc_row & 7) * RING + (size_t)((acc_row >> 3) & RMASK)) * qpad;
                double dv[QPL];
#pragma unroll
                for (int j = 0; j < QPL; ++j) dv[j] = dr[lane + WAVE * j];
#pragma unroll
                for (int j = 0; j < QPL; ++j) { ftr[j] += dv[j]; wftr[j] = wr[j] * ftr[j]; }
                if (RING < 2 * (GMAX / 8)) {
                    // a ring too shallow for a whole group refills at the top of the loop: nobody may
                    // still be reading the accepted row then
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                          // B3
                }
            }
            k = k_next;
#ifdef MCSAS_STAMPS
            MCSAS_STAMP(s9);
            if (mine) { ph[0] += s1 - s0; ph[1] += s2 - s1; ph[2] += s3 - s2; ph[3] += s4 - s3; ph[4] += s5 - s4; ph[10] += 1; }
            ph[5] += s6 - s5; ph[6] += s7 - s6; ph[7] += s8 - s7; ph[8] += s9 - s8; ph[9] += s9 - s0; ph[11] += 1;
#endif
        }
        MCSAS_STAMP(sb2);
#ifdef MCSAS_STAMPS
        if (wave == 0 && lane == 0) for (int i = 0; i < 12; ++i) ch.dbg[i] += ph[i];
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // no DMA may outlive the ring
        if (wave == 0 && lane == 0) lacc[Kb] = num_acc_win;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {   // write the window's slot tables back and store the accepted proposals (mcsas.py:381), all waves
            const int nacc = lacc[Kb];
            if (nacc > 0) {
                for (int i = tid; i < kmax_all; i += T) {
                    stage[i] = lstage[i];
                    int r = ri0 + i; if (r >= N) r -= N;
                    slot_of[r] = lslot[i];
                }
                for (int i = tid; i < nacc * P; i += T) {
                    const int kk = lacc[i / P], p = i % P;
                    int r = ri0 + kk; if (r >= N) r -= N;
                    rset[(size_t)r * P + p] = pval[(size_t)kk * MCSAS_MAX_ACTIVE + p];
                }
            }
        }
        if (wave == 0) {
            if (touched) {
                // re-sum the fit sums from ft so the incremental updates cannot drift; park ft in HBM
                double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    const double f = ftr[j], wf = wftr[j];
                    s1 += wf; s2 += wf * f; s3 += wIr[j] * f;
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
        MCSAS_STAMP(sb3);
#ifdef MCSAS_STAMPS
        if (lane == 0 && sb1 != 0) { ch.dbg[12] += sb1 - sb0; ch.dbg[13] += sb3 - sb2; ch.dbg[14] += 1; ch.dbg[15] += sb3 - sb0; }
        if (lane == 0) {
            if (ch.last_end) { ch.dbg[16] += (int64_t)(wc0 - ch.last_end); ch.dbg[17] += 1; }
            ch.dbg[18] += (int64_t)(wall_clock64() - wc0);
            ch.last_end = wall_clock64();
        }
#endif
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
#ifdef MCSAS_STAMPS
                for (int i = 0; i < 20; ++i) o.dbg[i] = ch.dbg[i];
#endif
                a.out[rep] = o;
                __hip_atomic_fetch_add(pa.n_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ one tick
// launch t: blocks [0, R) do SCAN(t) (skipped for t < 0), the others PROD(t + 1)
template <int M, int QPL>
__global__ __launch_bounds__(PIPE_BLOCK) void pipe_tick_kernel(const PipeArgs *pap, const int tick, const int stop_now) {
    // The argument block lives in device memory and is read where it is needed: passed by value it
    // would sit in SGPRs for the whole kernel (600+ bytes) and the scan loop would run on spilled
    // scalars (one v_readlane per use).
    extern __shared__ double lds[];
    const PipeArgs &pa = *pap;
    const int R = pa.c.n_reps, b = blockIdx.x, t = tick;
    if (b < R) {
        if (t >= 0) pipe_scan_block<QPL>(pa, lds, b, t, stop_now);
    } else {
        const int gy = pa.g.prod_blocks_y;
        pipe_prod_block<M, QPL>(pa, lds, (b - R) / gy, (b - R) % gy, gy, t + 1);
    }
}

// first tick bookkeeping: schedule records for ticks 0 and 1, counters
template <int UNUSED>
__global__ void pipe_reset_kernel(const PipeArgs *pap) {
    const PipeArgs &pa = *pap;
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
