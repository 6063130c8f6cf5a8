// chain_pipe.h — the speculative proposal window of chain_wg.h spread over the WHOLE chip for runs
// with fewer chains than CUs (BASELINE config 2: 50 chains on 256 CUs).
//
// One launch per "tick" t (pipe_tick_kernel), two kinds of workgroup in its grid:
//   producer blocks (four per chain at config 2) do PROD(t+1): the form-factor rows of the NEXT window of Kb steps.
//       A block owns 8 * rows_per_wave consecutive steps = one or more SUB-WINDOWS of W steps: per step
//       d = new - old to the window buffer, the three ft-independent sums (a = Σ w d, e = Σ wI d, g = Σ w d²),
//       and — fp64 MFMA, v_mfma_f64_16x16x4_f64 — the sub-window's Gram block G[a][k] = Σ_q w d_a d_k, which does
//       not depend on ft either.  Rows without an integral: the sub-window's d rows stay in LDS for the MFMAs, no
//       `new` row is stored and a contribution's cached row is evaluated again when it has gone stale (lazy rows);
//       Rows with an integral (or a smeared model) — the kernel instantiation pipe_tick_kernel<M, QPL, true>: the producer waves
//       of a chain PULL the window's rows from a queue, most expensive first, blocks whose queue is empty join other chains'
//       (pipe_prod_rowq); `new` row into a spare HBM row slot, slots swapped on acceptance; the Gram blocks (8 steps) are taken by
//       the scan block from the rows it has in LDS.
//   scan blocks (one per chain) do SCAN(t): per sub-window ONE pass over its d rows gives h_k = Σ (w ft) d_k for
//       the ft the sub-window starts from (eight waves, rows streamed HBM/L2 -> registers); then ONE wave takes
//       the W decisions with lane g = step g: a candidate's fit sums are SC + a, SIC + e, SCC + 2h + g, and after
//       an accepted row `acc` the later steps of the sub-window need only h_k += G[acc][k] (one LDS read) — no
//       barrier, no re-reduction and no restart per accepted move; the accepted rows are applied to ft once per
//       sub-window, in order, as ft += d.
// PROD(t+1) and SCAN(t) run concurrently inside one launch because a window's rows depend only on the random
// stream and on row slots settled two windows earlier (2*Kb <= N), never on the decisions of the window
// before.  Launch t+1 follows launch t on the same stream: the kernel boundary is the only synchronisation
// between workgroups; there are no in-kernel spin waits and no cross-workgroup flags (the row queue's counters are
// fetch-and-add tickets: nobody waits on them).
//
// Chain schedule: an attempt (one mcFit call) is initialised at tick t_init (PROD evaluates the N
// rows of the initial set, SCAN sums them and fits), then tick t > t_init handles window
// t - t_init - 1.  SCAN(t) publishes the small schedule record PROD(t+2) reads, double-buffered by
// tick parity, so a producer never reads a record that a concurrently running scan is writing.
#pragma once
#include "chain_common.h"
#include <type_traits>

namespace mcsas {

// Workgroup barrier for data handed over through LDS only: wait for this wave's LDS traffic, not for its global
// loads and stores.  __syncthreads() also drains vmcnt — in the scan loop that would make every barrier wait
// for the row batch that was prefetched just before it and for the stores of the accepted rows (a memory round
// trip per barrier).  The "memory" clobbers keep the compiler from moving LDS accesses across (the barrier
// builtin itself is IntrNoMem).
#define PIPE_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// Keep a wave-uniform double in a VGPR: the scan loop has far more uniform fp64 state than the 102
// SGPRs can hold, and spilled SGPRs come back one v_readlane at a time on the critical path.
#define MCSAS_IN_VGPR(x) asm volatile("" : "+v"(x))

// Producer row loop: the memory counter of gfx950 is in order over loads AND stores, so a wait for the `old` row of the
// next step that comes behind this step's sixteen row stores drains them (a memory round trip per row).  The rows are
// therefore waited for HERE, in front of the stores: everything outstanding at this point was issued before the row was
// evaluated.  Routing the values through an empty asm pins the wait (and the stores behind it) to this place.
#define PIPE_TL_WORDS 30                 /* timeline record of a wave: start, end, HW_ID, XCC_ID, then 26 marks */
#ifdef MCSAS_STAMPS
#define PIPE_TL_WRITE(pa, t, i) do { if ((pa).timeline && (t) - 1 == (pa).timeline_tick && (threadIdx.x & 63) == 0) (pa).timeline[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * PIPE_TL_WORDS + 4 + (i)] = wall_clock64(); } while (0)
#else
#define PIPE_TL_WRITE(pa, t, i) do {} while (0)
#endif
#define PIPE_TLX_MARK(pa, t, i) PIPE_TL_WRITE(pa, t, 18 + (i))   /* start-up marks 18..21: tables in LDS, proposals prepared, stale rows refreshed, row loop */
#define PIPE_TL_MARK(pa, t, i) PIPE_TL_WRITE(pa, t, i)
#ifdef MCSAS_STAMPS                      /* entry marks 22..25: clocks taken into registers (no argument-block read in the way), written later */
#define PIPE_TL_CLOCK(var) const uint64_t var = wall_clock64()
#define PIPE_TL_PUT(pa, t, i, var) do { if ((pa).timeline && (t) - 1 == (pa).timeline_tick && (threadIdx.x & 63) == 0) (pa).timeline[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * PIPE_TL_WORDS + 4 + (i)] = (var); } while (0)
#else
#define PIPE_TL_CLOCK(var) do {} while (0)
#define PIPE_TL_PUT(pa, t, i, var) do {} while (0)
#endif
#define PIPE_PIN_ROW(arr) do { _Pragma("unroll") for (int j_ = 0; j_ < QPL; ++j_) asm volatile("" : "+v"(arr[j_])); } while (0)

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

struct PipeSnap {                 // what the producer needs to know about a chain
    int32_t attempt, t_init, alive, pad;
    uint64_t init_base;           // draw index of the initial parameter set of this attempt
    uint64_t step_base;           // draw index of step 0 of this attempt
};

struct PipeChain {                // per-chain scanner state, lives in HBM between ticks
    PipeSnap snap[2];
    double SC, SIC, SCC, A, b, chi2;
    double X;                     // chi²·Q as the decisions carry it (PipeGeom::resum_every)
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
    int32_t w;                    // steps per sub-window = rows per producer block = 8 * rows_per_wave (<= 64)
    int32_t rows_per_wave;        // producer: rows per wave
    int32_t prod_blocks_y;        // producer blocks (= sub-windows) per chain and tick
    int32_t scan_waves;           // waves of a scan block
    int32_t qpl;
    int32_t gram_off;             // producer LDS: offset (doubles) of the Gram reduction buffer
    int32_t sub_per_block;        // scan sub-windows per producer block (8 * rows_per_wave / w)
    int32_t lazy_rows;            // no `new` rows are stored: an accepted step marks its contribution's cached row stale and the producer that
                                  // next needs it as `old` evaluates it again from the parameter set (rows without an integral only)
    int32_t overlap;              // producer variant (tuning): the Gram MFMAs of sub-window s are issued between the rows of s + 1, operands from HBM/L2
    int32_t gram_lds;             // producer: the sub-window's d rows are also kept in LDS and the Gram MFMAs read them from there
    int32_t drow_off;             // producer LDS: offset (doubles) of those rows, row stride qpad + PIPE_DROW_PAD
    int32_t resum_every;          // 0: the running sums are re-derived from ft at the end of every window (the window is fixed: rows without an
                                  // integral); n: at every n-th step of the attempt instead, and chi²·Q is carried across windows exactly —
                                  // nothing a chain decides then depends on the window, which follows the chain count for rows with an integral
    int32_t rowq;                 // rows with an integral (round 4): the producer waves of a chain PULL the window's rows from a queue, most expensive
                                  // first (no static deal), and the scan block works out the 8-step Gram blocks itself from the rows in its LDS
    int32_t rec_off;              // rowq: producer LDS offset (doubles) of the window's proposal records
    int32_t help;                 // rowq: a producer block whose chain's queue is empty joins another chain's (pipe_prod_rowq)
    uint64_t prod_lds, scan_lds;
};

struct PipeArgs {
    ChainArgs c;                  // c.cache_rows = N + 2*kb
    PipeGeom g;
    PipeChain *chains;            // [R]
    double *ft, *wft;             // [R][qpad]
    int32_t *slot_of;             // [R][N]
    int32_t *stage_slot;          // [R][2][kb]
    double *dwin;                 // [R][2][kb][qpad]   d rows of the window
    double *gwin;                 // [R][2][kb][w]      Gram blocks: row = step in the window, column = step in ITS sub-window
    double *scal;                 // [R][2][kb][4]   a = Σ w d, e = Σ wI d, g = Σ w d² of every step's row
    int32_t *row_valid;           // [R][N]  lazy_rows: 1 = the contribution's cached row is current
    double *pval;                 // [R][2][kb][MAX_ACTIVE]
    int32_t *povf;                // [R][2][kb]
    int32_t *rowq;                // [R][2]  rowq: next row of the window to hand out, by tick parity (zeroed by the scan block a tick ahead)
    int32_t *n_done;              // host-mapped: set to the number of chains when the last one has finished
    int32_t *n_done_dev;          // device counter behind it (one system-scope atomic per chain cost the last tick 30 us)
    int32_t tick, pad;            // unused: the tick travels as its own kernel argument
    uint64_t *timeline;           // stamps build: [blocks][8 waves][2] wall clock (10 ns) at wave start / end of tick `timeline_tick`
    int32_t timeline_tick, pad1;
};

// What a workgroup needs in its first microsecond, passed BY VALUE (kernel-argument segment, scalar loads): reading these
// through the argument block in device memory costs a dependent global round trip before the first useful load can be
// issued — pointer, then data — on the critical path of every tick.
struct PipeHot {
    const double *q, *w, *wI, *q3inv;                          // q3inv = 1 / q^3, host-made (the same IEEE operations as on the device)
    PipeChain *chains;
    int32_t n_reps, n_contrib, n_active, qpad, kb, prod_blocks_y, w_sub, pad;
    int64_t max_iter;
};

// schedule records go through scalar global loads / stores (a struct copy out of an address-space-qualified
// reference does not exist in C++)
__device__ __forceinline__ PipeSnap load_snap(const PipeSnap *p) {
    PipeSnap s;
    s.attempt = glb(&p->attempt)[0]; s.t_init = glb(&p->t_init)[0]; s.alive = glb(&p->alive)[0]; s.pad = 0;
    s.init_base = glb(&p->init_base)[0]; s.step_base = glb(&p->step_base)[0];
    return s;
}
__device__ __forceinline__ void store_snap(PipeSnap *p, const PipeSnap &s) {
    glb(&p->attempt)[0] = s.attempt; glb(&p->t_init)[0] = s.t_init; glb(&p->alive)[0] = s.alive; glb(&p->pad)[0] = 0;
    glb(&p->init_base)[0] = s.init_base; glb(&p->step_base)[0] = s.step_base;
}

constexpr int PIPE_BLOCK = 512;      // threads per workgroup of the tick kernel (8 waves)
constexpr int PIPE_WAVES = PIPE_BLOCK / 64;
constexpr int PIPE_GRAM_TILES_PER_ROUND = 2;
constexpr int PIPE_DROW_PAD = 8;         // LDS d rows: stride qpad + 8 doubles, so that the 64 16-byte operands of one Gram load hit 64 different bank groups
constexpr int PIPE_GRAM_NT_MAX = 3;          // overlapped producer: tiles per sub-window — W <= 32 (two 16-row groups: 3 tiles) or 24 packed (2)
constexpr int PIPE_RESUM_STEPS = 64;        // rows with an integral: the running sums are re-derived from ft every 64 steps of an attempt (a multiple of the 8-step sub-window)
constexpr int PIPE_MAX_ROW_DOUBLES = 32;   // scan block: doubles per lane held in row registers (rows per wave and sub-window x q per lane)   // 16x16 tiles reduced across the 8 waves per LDS round (32 KB)

// rows_per_wave_req: 0 = automatic, else the requested rows per producer wave (diagnostic / tuning)
static inline int pipe_geometry(int nq, int n_contrib, int tab_doubles, int heavy_rows, int rows_per_wave_req, int sub_req, int gram_global_req, int eager_req, int n_chains, int n_cus, PipeGeom *g,
                                int contrib_doubles) {
    int qpl = 1;
    while (qpl * 64 < nq) qpl *= 2;
    if (qpl > 16) return 1;
    const int qpad = qpl * 64;
    g->rowq = 0; g->rec_off = 0; g->help = 0;
    if (heavy_rows) {
        // Rows that cost an integral each (round 4).  The window is as long as 2 Kb <= N allows (a multiple of the 8-step Gram
        // blocks, at most 512 steps: one proposal per thread of a block) — it no longer follows the chain count
        // —, every chain gets the producer blocks that are left beside the scan blocks, and their waves pull the window's rows from
        // a queue in order of predicted cost: a wave that drew a cheap row simply comes back sooner.
        int kb = (n_contrib / 2) & ~7;
        int cap = PIPE_BLOCK;                                  // (one proposal per thread; 13 worm chains: 296 steps per window 4.12e6 steps/s, 256: 3.86e6)
#ifndef __HIPCC_RTC__
        if (const char *e = getenv("MCSAS_HIP_PIPE_KB_CAP")) { const int v = atoi(e) & ~7; if (v >= 8 && v <= PIPE_BLOCK) cap = v; }   // (measurement knob, host only)
#endif
        if (kb > cap) kb = cap;
        // Few chains (round 5): a tick hands R x Kb rows to 8 waves per CU, and while that is only a few "rounds" of rows the
        // tick lasts c0 + ceil(rounds) x (a row's time) — config 3's per-GPU share, 25 chains x 200 rows on 2048 wave slots, is 2.44
        // rounds: a third round for a sixth of the rows; a window of 160 steps (1.95 rounds) runs 8.5 % faster (tools/kb_probe.py:
        // 5.72 against 5.27e6 steps/s; the fixed part c0 of a tick — proposal records, scan blocks, the boundary — measured 0.7-0.95
        // of a row's time, which is why halving the window to get ONE round loses: 13 worm chains, 296 -> 152 steps, -22 %).  Between two
        // and four rounds the window is the multiple of 8 in [Kb/2, Kb] that maximises Kb / (0.75 + ceil(rounds)), the longest unless another
        // is 2 % better; from four rounds on the queue evens the rounds out and the longest window wins (measured: configs 3 at 200
        // chains, 4 at 50).  Nothing a chain decides depends on the window (PipeGeom::resum_every): same arrays, bit for bit.
        if (n_chains > 0 && n_cus > 0) {
            const double slots = 8.0 * (double)n_cus;
            auto rate = [&](int c) { const double r = (double)n_chains * c / slots; return (double)c / (0.75 + (double)(long long)(r + 1.0 - 1e-9)); };
            const double rmax = (double)n_chains * kb / slots;
            if (rmax > 2.0 && rmax < 4.0) {                      // (a third or fourth round to shed; 2 -> 1 loses: 6 chains of config 4, 496 -> 336 steps, -10 %)
                int best = kb;
                double best_v = rate(kb);
                for (int c = kb - 8; c >= 8 && 2 * c >= kb; c -= 8)
                    if (rate(c) > best_v * 1.02) { best_v = rate(c); best = c; }
                kb = best;
            }
        }
#ifndef __HIPCC_RTC__
        if (const char *e = getenv("MCSAS_HIP_PIPE_KB")) { const int v = atoi(e) & ~7; if (v >= 8 && v <= ((n_contrib / 2) & ~7) && v <= cap) kb = v; }     // (measurement knob, host only)
#endif
        if (kb < 8) return 1;
        // producer blocks per chain: enough to cover every CU by themselves — the launch then holds more workgroups than CUs, the
        // scan blocks (dispatched first) are done within a tenth of a tick, and the producer blocks that were waiting take over
        // their CUs and pull what is left of their chain's window (a static deal would leave those CUs idle for the rest of the tick)
        int by = 8;
        if (n_chains > 0 && n_cus > 0) { by = (n_cus + n_chains - 1) / n_chains; if (by < 1) by = 1; if (by > 32) by = 32; if (by * 8 > kb) by = (kb + 7) / 8; }
        g->kb = kb; g->qpl = qpl; g->w = 8; g->sub_per_block = 1; g->rows_per_wave = 1; g->prod_blocks_y = by;
        g->gram_off = 4 * qpad + tab_doubles; g->resum_every = PIPE_RESUM_STEPS;
        g->overlap = 0; g->gram_lds = 0; g->drow_off = 0; g->lazy_rows = 0;
        g->rowq = 1; g->rec_off = g->gram_off + 16;
        const size_t rec = (size_t)kb * (contrib_doubles + MCSAS_MAX_ACTIVE + 2) + ((size_t)3 * kb + 1) / 2;      // records; rank -> step and the two row slots (int32)
        g->prod_lds = sizeof(double) * ((size_t)g->rec_off + rec);
        g->help = (n_chains > 1 && (size_t)n_chains <= 2 * rec) ? 1 : 0;       // (the helpers' table of rows left per chain takes the records' place)
#ifndef __HIPCC_RTC__
        if (const char *e = getenv("MCSAS_HIP_PIPE_HELP")) g->help = atoi(e) ? g->help : 0;                           // (measurement knob, host only)
#endif
        g->scan_waves = PIPE_WAVES;
        g->scan_lds = sizeof(double) * ((size_t)g->w * qpad + 2 * (size_t)g->w * g->w + 3 * (size_t)qpad + (size_t)g->kb * 4 + 64)
                    + sizeof(int32_t) * (4 * (size_t)g->kb + 1 + 1 + 64 + 4 + 8) + 64;
        if (g->scan_lds > 160 * 1024 || g->prod_lds > 160 * 1024) return 2;     // (2: the window's records / row buffers do not fit the LDS)
        return 0;
    }
    // window: as many steps as 2*Kb <= N allows, Kb = (sub-windows) x (8 producer waves) x (rows per wave).
    // Rows per wave set the sub-window W = 8 rpw: measured on config 2 (tools/sweep_flags.sh) W = 48 beats 64
    // (Gram tiles per step fall from 10/64 to 6/48 and four producer blocks per chain instead of three fill
    // the CUs the scan blocks leave free) and 32 (more scan sub-windows per tick): candidates in that order,
    // the first one whose window is within 15 % of the largest wins.
    // scan sub-window for `r` rows per producer wave: the largest multiple of 8 that divides the producer block's rows and
    // whose d rows fit the scan block's LDS row buffer (the accepted rows are applied to ft from there, not from HBM).
    // The overlapped producer (Gram of sub-window s between the rows of s + 1) has tile schemes for W <= 32.
    const bool overlap = gram_global_req;
    auto pick_w = [&](int r) {
        int w = 8;
        for (int ws = 8; ws <= 8 * r && ws <= (overlap ? 32 : 64); ws += 8) {
            const int rps = ws / 8;
            if (rps == 5 || rps == 7 || rps * qpl > PIPE_MAX_ROW_DOUBLES) continue;   // the kernels instantiate 1, 2, 3, 4, 6, 8 rows per wave
            if ((8 * r) % ws == 0 && sizeof(double) * (size_t)ws * qpad <= 96 * 1024 && (sub_req == 0 || ws <= 8 * sub_req)) w = ws;
        }
        return w;
    };
    int rpw = 0, by = 0;
    if (rows_per_wave_req >= 1 && rows_per_wave_req <= 8) {
        rpw = rows_per_wave_req;
        while (rpw > 1 && 2 * 8 * rpw > n_contrib) --rpw;
        if (2 * 8 * rpw > n_contrib) return 1;
        by = n_contrib / (2 * 8 * rpw);
        if (by * 8 * rpw > 256) by = 256 / (8 * rpw);
    } else {
        static const int order[6] = {6, 8, 4, 3, 2, 1};
        int kbs[6], best_kb = 0;
        for (int c = 0; c < 6; ++c) {
            const int r = order[c];
            int b = (2 * 8 * r > n_contrib) ? 0 : n_contrib / (2 * 8 * r);
            if (b * 8 * r > 256) b = 256 / (8 * r);
            kbs[c] = b * 8 * r;
            if (kbs[c] > best_kb) best_kb = kbs[c];
        }
        if (best_kb == 0) return 1;
        // default: the first candidate whose window is within 15 % of the largest
        int rpw_d = 0, by_d = 0;
        for (int c = 0; c < 6 && !rpw_d; ++c)
            if (kbs[c] > 0 && 20 * kbs[c] >= 17 * best_kb) { rpw_d = order[c]; by_d = kbs[c] / (8 * rpw_d); }
        // Few chains: more, smaller producer blocks per chain shorten the producers' critical path (start-up + rows
        // per wave + Gram) down to where the scan block becomes the longer one — as long as every block still gets a
        // CU of its own.  Measured at 512 q x 400 (tools/sweep_rpw.sh): 3 rows per wave 3.1-3.3 ms per launch up to 28
        // chains, 6 rows 3.8-4.0 ms up to 51.  Only candidates with the SAME window and the same scan sub-window as
        // the default qualify: a chain's decisions depend on both (where the running sums are re-derived from ft, which
        // pairs of steps go through the Gram block), and a repetition must come out the same whether it runs beside 6
        // others (one of eight GPUs) or beside 49.
        if (n_chains > 0 && n_cus > 0) {
            static const int small_first[6] = {3, 4, 6, 8, 2, 1};
            for (int c = 0; c < 6 && !rpw; ++c) {
                const int r = small_first[c];
                int b = (2 * 8 * r > n_contrib) ? 0 : n_contrib / (2 * 8 * r);
                if (b * 8 * r > 256) b = 256 / (8 * r);
                if (b > 0 && b * 8 * r == by_d * 8 * rpw_d && pick_w(r) == pick_w(rpw_d) && n_chains * (b + 1) <= n_cus) { rpw = r; by = b; }
            }
        }
        if (!rpw) { rpw = rpw_d; by = by_d; }
    }
    g->kb = by * 8 * rpw; g->qpl = qpl;
    g->w = pick_w(rpw);
    g->sub_per_block = 8 * rpw / g->w;
    g->rows_per_wave = rpw;
    g->prod_blocks_y = by;
    g->gram_off = 4 * qpad + tab_doubles;
    g->resum_every = 0;
    {
        // reduction buffer of the Gram tiles: [8 waves][tiles][256]; the overlapped producer keeps two of them (the block of
        // sub-window s is summed while the partial tiles of s + 1 are being parked)
        const size_t red = overlap ? (size_t)2 * PIPE_WAVES * PIPE_GRAM_NT_MAX * 256 : (size_t)PIPE_WAVES * PIPE_GRAM_TILES_PER_ROUND * 256;
        g->prod_lds = sizeof(double) * ((size_t)g->gram_off + 16 + red);    // 16 doubles: counters
        g->overlap = overlap ? 1 : 0;
        // Rows without an integral: the Gram phase is a third of the producer's tick; with the sub-window's d rows parked
        // in LDS on their way to HBM it is MFMA-bound instead of waiting for an L2 round trip per sub-window.
        g->gram_lds = 0; g->drow_off = 0;
        const size_t with_rows = g->prod_lds + sizeof(double) * (size_t)g->w * (qpad + PIPE_DROW_PAD);
        if (!overlap && with_rows <= 160 * 1024) {
            g->gram_lds = 1; g->drow_off = g->gram_off + 16 + (int)red; g->prod_lds = with_rows;
        }
        // ... and no `new` rows go to HBM either (4 KB per step at Q = 512, a fifth of the tick's memory traffic): see lazy_rows
        g->lazy_rows = ((g->gram_lds || overlap) && !eager_req) ? 1 : 0;
    }
    g->scan_waves = PIPE_WAVES;
    // scan block LDS: the sub-window's d rows, two Gram blocks (double buffer), ft and w*ft, the window's scalars, h of
    // the sub-window, flags / slot tables / accepted lists
    g->scan_lds = sizeof(double) * ((size_t)g->w * qpad + 2 * (size_t)g->w * g->w + 2 * (size_t)qpad + (size_t)g->kb * 4 + 64)
                + sizeof(int32_t) * (4 * (size_t)g->kb + 1 + 1 + 64 + 4 + 8) + 64;
    if (g->scan_lds > 160 * 1024 || g->prod_lds > 160 * 1024) return 2;
    return 0;
}

// Models whose row costs a few hundred instructions (no orientation / contour integral): their producers store no `new`
// rows (4 KB per step) — a contribution's cached row goes stale when its proposal is accepted and is evaluated again,
// one q per thread, by the producer block that needs it as `old` N steps later (PipeGeom::lazy_rows).
// Same function, same inputs as RowEval -> the same bits.
// (Contrib<M>::ROW_CLASS == 0: sphere, core-shell sphere, Gaussian chain, LMA dense spheres)
template <int M> constexpr bool pipe_light_model_v = Contrib<M>::ROW_CLASS == 0;
template <int M>
__device__ __forceinline__ double pipe_point_intensity(const Contrib<M> &c, double q, double q3inv, const double *tab) {
    if constexpr (M == MCSAS_MODEL_SPHERE) return c.fast ? c.intensity_fast(q, q3inv) : c.intensity(q, tab);
    else return c.intensity(q, tab);
}

// one row of the window buffers as every kernel here holds it in registers: 16-byte loads, lane l and
// register pair c <-> q = 128 c + 2 l + {0, 1}  (QPL = 1: one 8-byte load, q = l)
template <int QPL>
__device__ __forceinline__ void load_row_pairs(const MCSAS_GLOBAL double *row, int lane, double (&r)[QPL]) {
    if constexpr (QPL >= 2) {
#pragma unroll
        for (int c = 0; c < QPL / 2; ++c) {
            const v2f64 v = *(const MCSAS_GLOBAL v2f64 *)(row + 128 * c + 2 * lane);
            r[2 * c] = v.x; r[2 * c + 1] = v.y;
        }
    } else {
        r[0] = row[lane];
    }
}
template <int QPL>
__device__ __forceinline__ void load_row_pairs_lds(const double *row, int lane, double (&r)[QPL]) {
    if constexpr (QPL >= 2) {
#pragma unroll
        for (int c = 0; c < QPL / 2; ++c) {
            const v2f64 v = *reinterpret_cast<const v2f64 *>(row + 128 * c + 2 * lane);
            r[2 * c] = v.x; r[2 * c + 1] = v.y;
        }
    } else {
        r[0] = row[lane];
    }
}

// ------------------------------------------------------------------------------------ producer
// Gram block of one sub-window, G[a][k] = Σ_q w_q d_a(q) d_k(q) over the W rows this block has just written,
// with v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) B(4x16), lane l supplies A[l % 16][l / 16] and
// B[l / 16][l % 16], result register r holds D[4 r + l / 16][l % 16].  With A = d rows and B = (w d) rows both
// operands of lane l are the SAME element (row l % 16, q-slot l / 16) — one load, one multiply, one MFMA.
// Wave v takes the q slice [v 8 QPL, (v + 1) 8 QPL) for ALL tiles (every row element is loaded exactly once
// per block); its load number c covers 8 consecutive q of the slice, two per lane slot kk = l / 16 (any
// assignment of q to MFMA k-slots is fine, the sum runs over all of them): the four slots of a row read 64
// contiguous bytes, so a 128-byte line is consumed by two consecutive loads instead of lingering in L1.
// The eight partial tiles are then summed in wave order through LDS (deterministic) and written to
// gout[a][k], a, k < W.
// The MFMA part: this wave's share (q slice gw of NWV) of every upper-triangular tile.  T = 16-row groups of the
// sub-window (1..4), compile time: straight-line MFMA code.
template <int QPL, int T, int NWV>
__device__ __forceinline__ void pipe_gram_mfma(const MCSAS_GLOBAL double *drows, int qpad, int nvalid, const double *lw, int gw,
                                               v4f64 (&acc)[T * (T + 1) / 2]) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kk = lane >> 4;
    constexpr int NT = T * (T + 1) / 2;                       // upper-triangular tiles (gi <= gj), row-major
    constexpr int SLICE = 64 * QPL / NWV;                     // q per wave; a load covers 8 of them (two per k-slot)
    static_assert(SLICE >= 8, "too many waves for this q count");
    const int qs = gw * SLICE + kk * 2;
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (v4f64){0., 0., 0., 0.};
    const MCSAS_GLOBAL double *rowp[T];
    bool rowok[T];
#pragma unroll
    for (int gi = 0; gi < T; ++gi) {
        const int rr = 16 * gi + m;
        rowok[gi] = rr < nvalid;
        rowp[gi] = drows + (size_t)(rowok[gi] ? rr : 0) * qpad + qs;
    }
    // loads run one step-pair ahead of the MFMAs that consume them (an L2 round trip per step-pair otherwise)
    v2f64 nxt[T], wnx;
    auto fetch = [&](int s, v2f64 (&dst)[T], v2f64 &wdst) {
        wdst = *reinterpret_cast<const v2f64 *>(lw + qs + 4 * s);
#pragma unroll
        for (int gi = 0; gi < T; ++gi) dst[gi] = *(const MCSAS_GLOBAL v2f64 *)(rowp[gi] + 4 * s);
    };
    fetch(0, nxt, wnx);
#pragma unroll
    for (int s = 0; s < SLICE / 4; s += 2) {
        v2f64 av[T], bv[T];
        const v2f64 wv = wnx;
#pragma unroll
        for (int gi = 0; gi < T; ++gi) av[gi] = rowok[gi] ? nxt[gi] : (v2f64){0., 0.};
        if (s + 2 < SLICE / 4) fetch(s + 2, nxt, wnx);
#pragma unroll
        for (int gi = 0; gi < T; ++gi) bv[gi] = av[gi] * wv;
        int ti = 0;
#pragma unroll
        for (int gi = 0; gi < T; ++gi)
#pragma unroll
            for (int gj = gi; gj < T; ++gj) {
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].x, bv[gj].x, acc[ti], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].y, bv[gj].y, acc[ti], 0, 0, 0);
                ++ti;
            }
    }
}

// tile number -> (row group, column group) of the upper triangle, and the store of one summed element
template <int T>
__device__ __forceinline__ void pipe_gram_store(int tsel, int idx, double sum, int W, MCSAS_GLOBAL double *gout, MCSAS_GLOBAL double *scal_sub = nullptr) {
    int ti = 0, tgi = 0, tgj = 0;
#pragma unroll
    for (int gi = 0; gi < T; ++gi)
#pragma unroll
        for (int gj = gi; gj < T; ++gj) { if (ti == tsel) { tgi = gi; tgj = gj; } ++ti; }
    const int i = 4 * (idx >> 6) + ((idx & 63) >> 4), j = idx & 15;   // result register r of lane l holds D[4 r + l / 16][l % 16]
    const int ar = 16 * tgi + i, kc = 16 * tgj + j;
    if (ar < W && kc < W) {
        gout[(size_t)ar * W + kc] = sum;
        if (scal_sub && ar == kc) scal_sub[ar * 4 + 2] = sum;
    }
}

// All eight waves of the block, partial tiles summed in wave order through LDS between two barriers.
template <int QPL, int T>
__device__ __forceinline__ void pipe_prod_gram_t(const MCSAS_GLOBAL double *drows, int qpad, int W, int nvalid, const double *lw,
                                                 double *gred, MCSAS_GLOBAL double *gout) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = T * (T + 1) / 2;
    v4f64 acc[NT];
    pipe_gram_mfma<QPL, T, PIPE_WAVES>(drows, qpad, nvalid, lw, wave, acc);
    // cross-wave sum, PIPE_GRAM_TILES_PER_ROUND tiles per round
    constexpr int TPR = PIPE_GRAM_TILES_PER_ROUND;
#pragma unroll
    for (int r0 = 0; r0 < NT; r0 += TPR) {
#pragma unroll
        for (int u = 0; u < TPR; ++u)
            if (r0 + u < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gred[((size_t)(wave * TPR + u) * 4 + r) * 64 + lane] = acc[r0 + u < NT ? r0 + u : 0][r];
            }
        PIPE_LDS_BARRIER();
        {
            const int u = tid >> 8, idx = tid & 255;          // 512 threads <-> TPR (= 2) tiles x 256 elements
            const int tsel = r0 + u;
            if (tsel < NT) {
                double sum = 0.;
#pragma unroll
                for (int v = 0; v < PIPE_WAVES; ++v) sum += gred[(size_t)(v * TPR + u) * 256 + idx];
                pipe_gram_store<T>(tsel, idx, sum, W, gout);
            }
        }
        if (r0 + TPR < NT) PIPE_LDS_BARRIER();
    }
}


template <int QPL>
__device__ __forceinline__ void pipe_prod_gram(const MCSAS_GLOBAL double *drows, int qpad, int W, int nvalid, const double *lw,
                                               double *gred, MCSAS_GLOBAL double *gout) {
    switch ((W + 15) >> 4) {                                   // uniform for the launch
        case 1: pipe_prod_gram_t<QPL, 1>(drows, qpad, W, nvalid, lw, gred, gout); break;
        case 2: pipe_prod_gram_t<QPL, 2>(drows, qpad, W, nvalid, lw, gred, gout); break;
        case 3: pipe_prod_gram_t<QPL, 3>(drows, qpad, W, nvalid, lw, gred, gout); break;
        default: pipe_prod_gram_t<QPL, 4>(drows, qpad, W, nvalid, lw, gred, gout); break;
    }
}


// The same Gram block with the operands read from the LDS copy of the sub-window's d rows (row stride dstr).
// PACK (W = 24, three 8-row groups g0 g1 g2): the six upper-triangular 8x8 blocks fit TWO 16x16 tiles instead of the
// three of the 16-row grouping — tile 0 = rows [g0 g1] x columns [g1 g2] (blocks 01 02 11 12), tile 1 = rows and
// columns [g0 g2] (blocks 00 22; its 02 is a duplicate and not stored): a third fewer MFMAs.
template <int QPL, int T, bool PACK>
__device__ __forceinline__ void pipe_gram_mfma_lds(const double *drows, int dstr, int nvalid, const double *lw, int gw,
                                                   v4f64 (&acc)[PACK ? 2 : T * (T + 1) / 2]) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kk = lane >> 4;
    constexpr int NT = PACK ? 2 : T * (T + 1) / 2;
    constexpr int NL = PACK ? 3 : T;                          // row operands per lane and step-pair
    constexpr int SLICE = 64 * QPL / PIPE_WAVES;
    static_assert(SLICE >= 8, "too many waves for this q count");
    const int qs = gw * SLICE + kk * 2;
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (v4f64){0., 0., 0., 0.};
    const double *rowp[NL];
    bool rowok[NL];
#pragma unroll
    for (int gi = 0; gi < NL; ++gi) {
        const int rr = PACK ? (gi == 0 ? m : gi == 1 ? 8 + m : (m < 8 ? m : m + 8)) : 16 * gi + m;
        rowok[gi] = rr < nvalid;
        rowp[gi] = drows + (size_t)(rowok[gi] ? rr : 0) * dstr + qs;
    }
#pragma unroll
    for (int sp = 0; sp < SLICE / 8; ++sp) {
        v2f64 av[NL], bv[NL];
        const v2f64 wv = *reinterpret_cast<const v2f64 *>(lw + qs + 8 * sp);
#pragma unroll
        for (int gi = 0; gi < NL; ++gi) {
            const v2f64 x = *reinterpret_cast<const v2f64 *>(rowp[gi] + 8 * sp);
            av[gi] = rowok[gi] ? x : (v2f64){0., 0.};
            bv[gi] = av[gi] * wv;
        }
        if constexpr (PACK) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0].x, bv[1].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2].x, bv[2].x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0].y, bv[1].y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2].y, bv[2].y, acc[1], 0, 0, 0);
        } else {
            int ti = 0;
#pragma unroll
            for (int gi = 0; gi < T; ++gi)
#pragma unroll
                for (int gj = gi; gj < T; ++gj) {
                    acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].x, bv[gj].x, acc[ti], 0, 0, 0);
                    acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].y, bv[gj].y, acc[ti], 0, 0, 0);
                    ++ti;
                }
        }
    }
}

// element idx of packed tile tsel -> (row, column) of the 24-step sub-window, or skipped
// (the block's diagonal, g_k = sum_q w d_k^2, is the third of a step's ft-independent sums: to scal_sub[k][2])
__device__ __forceinline__ void pipe_gram_store_pack(int tsel, int idx, double sum, MCSAS_GLOBAL double *gout, MCSAS_GLOBAL double *scal_sub) {
    const int i = 4 * (idx >> 6) + ((idx & 63) >> 4), j = idx & 15;
    if (tsel == 0) {
        gout[(size_t)i * 24 + 8 + j] = sum;
        if (i == 8 + j) scal_sub[i * 4 + 2] = sum;
    } else if ((i < 8) == (j < 8)) {
        const int ar = i < 8 ? i : i + 8, kc = j < 8 ? j : j + 8;
        gout[(size_t)ar * 24 + kc] = sum;
        if (ar == kc) scal_sub[ar * 4 + 2] = sum;
    }
}

template <int QPL, int T, bool PACK>
__device__ __forceinline__ void pipe_prod_gram_lds_t(const double *drows, int dstr, int W, int nvalid, const double *lw,
                                                     double *gred, MCSAS_GLOBAL double *gout, MCSAS_GLOBAL double *scal_sub) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = PACK ? 2 : T * (T + 1) / 2;
    v4f64 acc[NT];
    pipe_gram_mfma_lds<QPL, T, PACK>(drows, dstr, nvalid, lw, wave, acc);
    constexpr int TPR = PIPE_GRAM_TILES_PER_ROUND;
#pragma unroll
    for (int r0 = 0; r0 < NT; r0 += TPR) {
#pragma unroll
        for (int u = 0; u < TPR; ++u)
            if (r0 + u < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gred[((size_t)(wave * TPR + u) * 4 + r) * 64 + lane] = acc[r0 + u < NT ? r0 + u : 0][r];
            }
        PIPE_LDS_BARRIER();
        {
            const int u = tid >> 8, idx = tid & 255;
            const int tsel = r0 + u;
            if (tsel < NT) {
                double sum = 0.;
#pragma unroll
                for (int v = 0; v < PIPE_WAVES; ++v) sum += gred[(size_t)(v * TPR + u) * 256 + idx];
                if constexpr (PACK) pipe_gram_store_pack(tsel, idx, sum, gout, scal_sub);
                else pipe_gram_store<T>(tsel, idx, sum, W, gout, scal_sub);
            }
        }
        if (r0 + TPR < NT) PIPE_LDS_BARRIER();
    }
}

template <int QPL>
__device__ __forceinline__ void pipe_prod_gram_lds(const double *drows, int dstr, int W, int nvalid, const double *lw,
                                                   double *gred, MCSAS_GLOBAL double *gout, MCSAS_GLOBAL double *scal_sub) {
    if (W == 24) { pipe_prod_gram_lds_t<QPL, 2, true>(drows, dstr, W, nvalid, lw, gred, gout, scal_sub); return; }
    switch ((W + 15) >> 4) {                                   // uniform for the launch
        case 1: pipe_prod_gram_lds_t<QPL, 1, false>(drows, dstr, W, nvalid, lw, gred, gout, scal_sub); break;
        case 2: pipe_prod_gram_lds_t<QPL, 2, false>(drows, dstr, W, nvalid, lw, gred, gout, scal_sub); break;
        case 3: pipe_prod_gram_lds_t<QPL, 3, false>(drows, dstr, W, nvalid, lw, gred, gout, scal_sub); break;
        default: pipe_prod_gram_lds_t<QPL, 4, false>(drows, dstr, W, nvalid, lw, gred, gout, scal_sub); break;
    }
}


// ---- the Gram block in UNITS, for producers that evaluate the next sub-window's rows at the same time ----------------
// Rows without an integral: the Gram MFMAs of sub-window s are issued BETWEEN the rows of sub-window s + 1 (the matrix
// pipe runs beside the vector pipe: while one wave of a SIMD is inside a run of MFMAs its partner has the vector issue
// slots to itself), so a wave's share of a block — its q slice of 8 QPL points, every row — is cut into QPL units of 8 q
// (two MFMA k-steps per tile) that are done a few at a time.  Operands come from the window buffer the rows were just
// stored to (HBM/L2; same CU, same L1: visible to the whole workgroup once the storing waves have waited for their
// stores and passed a barrier).  The accumulators stay in registers between the calls.
// PACK (W = 24, three 8-row groups g0 g1 g2): the six upper-triangular 8x8 blocks fit TWO 16x16 tiles instead of the
// three of the 16-row grouping — tile 0 = rows [g0 g1] x columns [g1 g2] (blocks 01 02 11 12), tile 1 = rows and
// columns [g0 g2] (blocks 00 22; its 02 is a duplicate and not stored): a third fewer MFMAs.

template <int QPL, int T, bool PACK>
__device__ __forceinline__ void pipe_gram_units(const MCSAS_GLOBAL double *drows, int qpad, int nvalid, const double *lw, int gw,
                                                int u0, int u1, v4f64 (&acc)[PIPE_GRAM_NT_MAX]) {
    static_assert(T <= 2, "at most two 16-row groups per sub-window");
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kk = lane >> 4;
    constexpr int NL = PACK ? 3 : T;                          // row operands per lane and unit
    constexpr int SLICE = 64 * QPL / PIPE_WAVES;              // q per wave = 8 * (units per wave)
    static_assert(SLICE >= 8, "too many waves for this q count");
    const int qs = gw * SLICE + kk * 2;
    if (u0 == 0) {
#pragma unroll
        for (int i = 0; i < PIPE_GRAM_NT_MAX; ++i) acc[i] = (v4f64){0., 0., 0., 0.};
    }
    const MCSAS_GLOBAL double *rowp[NL];
    bool rowok[NL];
#pragma unroll
    for (int gi = 0; gi < NL; ++gi) {
        const int rr = PACK ? (gi == 0 ? m : gi == 1 ? 8 + m : (m < 8 ? m : m + 8)) : 16 * gi + m;
        rowok[gi] = rr < nvalid;
        rowp[gi] = drows + (size_t)(rowok[gi] ? rr : 0) * qpad + qs;
    }
    v2f64 cur[NL], nxt[NL];
#pragma unroll
    for (int gi = 0; gi < NL; ++gi) cur[gi] = *(const MCSAS_GLOBAL v2f64 *)(rowp[gi] + 8 * u0);
    for (int u = u0; u < u1; ++u) {
        const int un = u + 1 < u1 ? u + 1 : u;                // (the last unit is requested twice: no load under a condition)
#pragma unroll
        for (int gi = 0; gi < NL; ++gi) nxt[gi] = *(const MCSAS_GLOBAL v2f64 *)(rowp[gi] + 8 * un);
        const v2f64 wv = *reinterpret_cast<const v2f64 *>(lw + qs + 8 * u);
        v2f64 av[NL], bv[NL];
#pragma unroll
        for (int gi = 0; gi < NL; ++gi) {
            av[gi] = rowok[gi] ? cur[gi] : (v2f64){0., 0.};
            bv[gi] = av[gi] * wv;
        }
        if constexpr (PACK) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0].x, bv[1].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2].x, bv[2].x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0].y, bv[1].y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2].y, bv[2].y, acc[1], 0, 0, 0);
        } else {
            int ti = 0;
#pragma unroll
            for (int gi = 0; gi < T; ++gi)
#pragma unroll
                for (int gj = gi; gj < T; ++gj) { acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].x, bv[gj].x, acc[ti], 0, 0, 0); ++ti; }
            ti = 0;
#pragma unroll
            for (int gi = 0; gi < T; ++gi)
#pragma unroll
                for (int gj = gi; gj < T; ++gj) { acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].y, bv[gj].y, acc[ti], 0, 0, 0); ++ti; }
        }
#pragma unroll
        for (int gi = 0; gi < NL; ++gi) cur[gi] = nxt[gi];
    }
}

// The same units with the operand loads and the MFMAs as two calls, so that a row evaluation fits between them: the loads
// of up to UCAP = 12 / NL units are in flight while the row is computed, the MFMAs run on landed operands.
constexpr int PIPE_GRAM_PREF = 12;                             // 16-byte operand registers per lane held across a row evaluation
template <int T, bool PACK> struct PipeGramScheme {
    static constexpr int NL = PACK ? 3 : T;
    static constexpr int UCAP = PIPE_GRAM_PREF / NL;
};
template <int QPL, int T, bool PACK>
__device__ __forceinline__ void pipe_gram_fetch(const MCSAS_GLOBAL double *drows, int qpad, int nvalid, int gw, int u0, int n,
                                                v2f64 (&G)[PIPE_GRAM_PREF]) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kk = lane >> 4;
    constexpr int NL = PipeGramScheme<T, PACK>::NL, UCAP = PipeGramScheme<T, PACK>::UCAP;
    constexpr int SLICE = 64 * QPL / PIPE_WAVES;
    const int qs = gw * SLICE + kk * 2;
#pragma unroll
    for (int gi = 0; gi < NL; ++gi) {
        const int rr = PACK ? (gi == 0 ? m : gi == 1 ? 8 + m : (m < 8 ? m : m + 8)) : 16 * gi + m;
        const MCSAS_GLOBAL double *rowp = drows + (size_t)(rr < nvalid ? rr : 0) * qpad + qs;
#pragma unroll
        for (int i = 0; i < UCAP; ++i) {
            const int u = u0 + (i < n ? i : n - 1);           // (past the chunk: its last unit again — no load under a condition)
            G[i * NL + gi] = *(const MCSAS_GLOBAL v2f64 *)(rowp + 8 * u);
        }
    }
}
template <int QPL, int T, bool PACK>
__device__ __forceinline__ void pipe_gram_consume(const v2f64 (&G)[PIPE_GRAM_PREF], int nvalid, const double *lw, int gw, int u0, int n,
                                                  v4f64 (&acc)[PIPE_GRAM_NT_MAX]) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kk = lane >> 4;
    constexpr int NL = PipeGramScheme<T, PACK>::NL, UCAP = PipeGramScheme<T, PACK>::UCAP;
    constexpr int SLICE = 64 * QPL / PIPE_WAVES;
    const int qs = gw * SLICE + kk * 2;
    if (u0 == 0) {
#pragma unroll
        for (int i = 0; i < PIPE_GRAM_NT_MAX; ++i) acc[i] = (v4f64){0., 0., 0., 0.};
    }
    bool rowok[NL];
#pragma unroll
    for (int gi = 0; gi < NL; ++gi) {
        const int rr = PACK ? (gi == 0 ? m : gi == 1 ? 8 + m : (m < 8 ? m : m + 8)) : 16 * gi + m;
        rowok[gi] = rr < nvalid;
    }
#pragma unroll
    for (int i = 0; i < UCAP; ++i)
        if (i < n) {                                          // uniform
            const v2f64 wv = *reinterpret_cast<const v2f64 *>(lw + qs + 8 * (u0 + i));
            v2f64 av[NL], bv[NL];
#pragma unroll
            for (int gi = 0; gi < NL; ++gi) {
                av[gi] = rowok[gi] ? G[i * NL + gi] : (v2f64){0., 0.};
                bv[gi] = av[gi] * wv;
            }
            if constexpr (PACK) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0].x, bv[1].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2].x, bv[2].x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0].y, bv[1].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2].y, bv[2].y, acc[1], 0, 0, 0);
            } else {
                int ti = 0;
#pragma unroll
                for (int gi = 0; gi < T; ++gi)
#pragma unroll
                    for (int gj = gi; gj < T; ++gj) { acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].x, bv[gj].x, acc[ti], 0, 0, 0); ++ti; }
                ti = 0;
#pragma unroll
                for (int gi = 0; gi < T; ++gi)
#pragma unroll
                    for (int gj = gi; gj < T; ++gj) { acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[gi].y, bv[gj].y, acc[ti], 0, 0, 0); ++ti; }
            }
        }
}

// the launch-uniform choice of the tile scheme (W <= 32 by pipe_geometry for these rows)
template <int QPL>
__device__ __forceinline__ void pipe_gram_units_w(int W, const MCSAS_GLOBAL double *drows, int qpad, int nvalid, const double *lw, int gw,
                                                  int u0, int u1, v4f64 (&acc)[PIPE_GRAM_NT_MAX]) {
    if (W == 24) pipe_gram_units<QPL, 2, true>(drows, qpad, nvalid, lw, gw, u0, u1, acc);
    else if (W <= 16) pipe_gram_units<QPL, 1, false>(drows, qpad, nvalid, lw, gw, u0, u1, acc);
    else pipe_gram_units<QPL, 2, false>(drows, qpad, nvalid, lw, gw, u0, u1, acc);
}
__device__ __forceinline__ int pipe_gram_tiles(int W) { return W == 24 ? 2 : (W <= 16 ? 1 : 3); }

// a wave's partial tiles -> its slots of the reduction buffer gred[wave][tile][256]
__device__ __forceinline__ void pipe_gram_park(const v4f64 (&acc)[PIPE_GRAM_NT_MAX], int nt, double *gred, int wave, int lane) {
#pragma unroll
    for (int ti = 0; ti < PIPE_GRAM_NT_MAX; ++ti)
        if (ti < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gred[((size_t)(wave * PIPE_GRAM_NT_MAX + ti) * 4 + r) * 64 + lane] = acc[ti][r];
        }
}

// all threads: the eight waves' partial tiles summed in wave order (deterministic), Gram block to gout[a][k] (a, k < W), and
// its diagonal — g_k = sum_q w d_k^2, the third of a step's ft-independent sums — to scal_sub[k][2]
__device__ __forceinline__ void pipe_gram_sum_store(int W, const double *gred, MCSAS_GLOBAL double *gout, MCSAS_GLOBAL double *scal_sub) {
    const int tid = threadIdx.x;
    const int nt = pipe_gram_tiles(W);
    for (int e = tid; e < nt * 256; e += PIPE_BLOCK) {
        const int ti = e >> 8, idx = e & 255;
        double sum = 0.;
#pragma unroll
        for (int v = 0; v < PIPE_WAVES; ++v) sum += gred[(size_t)(v * PIPE_GRAM_NT_MAX + ti) * 256 + idx];
        const int i = 4 * (idx >> 6) + ((idx & 63) >> 4), j = idx & 15;   // result register r of lane l holds D[4 r + l / 16][l % 16]
        int ar = -1, kc = -1;
        if (W == 24) {
            if (ti == 0) { ar = i; kc = 8 + j; }
            else if ((i < 8) == (j < 8)) { ar = i < 8 ? i : i + 8; kc = j < 8 ? j : j + 8; }
        } else {
            const int tgi = ti == 2 ? 1 : 0, tgj = ti == 0 ? 0 : 1;     // tiles (0,0), (0,1), (1,1)
            ar = 16 * tgi + i; kc = 16 * tgj + j;
        }
        if (ar >= 0 && ar < W && kc < W) {
            gout[(size_t)ar * W + kc] = sum;
            if (ar == kc) scal_sub[ar * 4 + 2] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------ producer, rows with an integral
// Rows of these models (or of a smeared one) cost 10^4 .. 10^5 instructions and up to five times their neighbour's (a worm's Kuhn
// length sets the number of quadrature panels, a cylinder's radius the Bessel function's branch): with a static deal a tick lasted as
// long as its unluckiest wave (13 worm chains: 0.47 of the issue rate; 256 chains, whose many blocks the dispatcher balances: 0.70).
// The producer waves of a chain PULL rows instead, from a counter per chain and tick parity in device memory (PipeArgs::rowq, zeroed a
// tick ahead by the scan block):
//   window tick  1. the block works out the proposals of ALL Kb steps of the window, one per thread (draw, generator transform,
//                   prepare(), predicted cost: models.h row_cost), and parks the records in LDS — 500 instructions per step against
//                   10^5 for its row;
//                2. every thread ranks its step by predicted cost (most expensive first; steps behind max_iter last);
//                3. every wave takes the next rank from the counter until the window is handed out: longest rows first, the short
//                   ones fill the gaps.
//   initial tick the contributions of the initial set, four at a time.
// No Gram phase here: the scan block has the rows of an 8-step sub-window in its LDS anyway and takes the 28 dot products there
// (pipe_scan_block) — the steps of a sub-window are no longer evaluated by one workgroup.
// One visit = one chain's queue worked on until it is empty.  `helper`: the chain is not the block's own (pipe_prod_rowq).
template <int M, int QPL>
__device__ __forceinline__ void pipe_rowq_visit(const PipeArgs &pa, const PipeHot &hot, double *lds, const QTables &qt, int rep, int by, int gy, int t,
                                                const PipeSnap &sn, bool helper) {
    const ChainArgs &a = pa.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = hot.n_contrib, P = hot.n_active, qpad = hot.qpad, Kb = hot.kb;
    const int64_t max_iter = hot.max_iter;
    double *lq = lds, *lw = lds + qpad, *lwI = lds + 2 * qpad;
    auto rset = glb(a.rset) + (size_t)rep * N * P;
    auto cache = glb(a.cache) + (size_t)rep * a.cache_rows * qpad;
    const DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                         (uint32_t)(a.rep_offset + rep)};
    auto slot_of = glb(pa.slot_of) + (size_t)rep * N;
    auto stage = glb(pa.stage_slot) + (size_t)rep * 2 * Kb;
    auto row_valid = glb(pa.row_valid) + (size_t)rep * N;
    int32_t *rowq = pa.rowq + (size_t)rep * 2 + (t & 1);

    if (t == sn.t_init) {
        // ---- initial parameter set of the attempt (mcsas.py:310-319).  (A static share per wave would make the workgroups that
        // wait for a CU — the launch has more of them than the chip — a second round as long as the first.)
        if (!helper) {
            for (int i = tid + by * PIPE_BLOCK; i < N; i += PIPE_BLOCK * gy) { slot_of[i] = i; row_valid[i] = 1; }
            for (int i = tid + by * PIPE_BLOCK; i < 2 * Kb; i += PIPE_BLOCK * gy) stage[i] = N + i;
        }
        int ovf = 0;
        constexpr int CH = 4;
        for (int pulls = 0; pulls * CH <= N; ++pulls) {
            int n0 = 0;
            if (lane == 0) n0 = __hip_atomic_fetch_add(rowq, CH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n0 = __builtin_amdgcn_readfirstlane(n0);
            if (n0 < 0 || n0 >= N) break;
            const int n = n0 + lane;
            double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
            if (lane < CH && n < N) {
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
            for (int l = 0; l < CH && n0 + l < N; ++l) {
                const Contrib<M> c = mine.bcast(__builtin_amdgcn_readfirstlane(l));
                double it[QPL];
                RowEval<M, QPL>::run(c, qt, lane, it);
#pragma unroll
                for (int j = 0; j < QPL; ++j) cache[(size_t)(n0 + l) * qpad + lane + WAVE * j] = it[j];
            }
        }
        if (__any(ovf) && lane == 0) atomicOr(&pa.chains[rep].overflow, 1);
        return;
    }

    // ---- window w of the attempt
    const int64_t w = (int64_t)t - sn.t_init - 1;
    const int buf = t & 1;
    const int64_t s0 = w * Kb;
    const int64_t left = max_iter - s0;
    const int nvalid = left >= Kb ? Kb : (left > 0 ? (int)left : 0);
    constexpr int CON = (int)(sizeof(Contrib<M>) / 8), REC = CON + MCSAS_MAX_ACTIVE + 2;   // Contrib | proposal values | overflow flag | cost
    static_assert(sizeof(Contrib<M>) % 8 == 0, "Contrib record");
    double *rec = lds + pa.g.rec_off;                         // [Kb][REC]
    int32_t *order = reinterpret_cast<int32_t *>(rec + (size_t)Kb * REC);   // [Kb] rank -> step of the window
    int32_t *rslot = order + Kb;                              // [Kb][2] row slot of the step's contribution, spare slot for its new row
    for (int k = tid; k < Kb; k += PIPE_BLOCK) {
        double prow[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
        int pov = 0;
#pragma unroll
        for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
            if (p < P) {
                double u = 0.5;
                if (k < nvalid) u = src.at(sn.step_base + (uint64_t)(s0 + k) * P + p, pov);
                prow[p] = gen_transform(a.gen_kind[p], u) * (a.gen_hi[p] - a.gen_lo[p]) + a.gen_lo[p];
            }
        Contrib<M> prop;
        prop.prepare(a.model, prow);
        double cost = -1.0;                                   // (steps behind max_iter: last)
        if (k < nvalid) cost = Contrib<M>::ROW_CLASS == 2 ? row_cost<M, QPL>(prop, lq) : 0.0;
        double tmp[CON];
        __builtin_memcpy(tmp, &prop, sizeof(Contrib<M>));
#pragma unroll
        for (int i = 0; i < CON; ++i) rec[(size_t)k * REC + i] = tmp[i];
#pragma unroll
        for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p) rec[(size_t)k * REC + CON + p] = prow[p];
        rec[(size_t)k * REC + CON + MCSAS_MAX_ACTIVE] = (double)pov;
        rec[(size_t)k * REC + CON + MCSAS_MAX_ACTIVE + 1] = cost;
        const int r = (int)((s0 + k) % N);
        rslot[2 * k] = slot_of[r]; rslot[2 * k + 1] = stage[buf * Kb + k];
    }
    __syncthreads();
    for (int k = tid; k < Kb; k += PIPE_BLOCK) {
        int rank = k;
        if constexpr (Contrib<M>::ROW_CLASS == 2) {
            const double cst = rec[(size_t)k * REC + CON + MCSAS_MAX_ACTIVE + 1];
            rank = 0;
            for (int j = 0; j < Kb; ++j) {
                const double cj = rec[(size_t)j * REC + CON + MCSAS_MAX_ACTIVE + 1];
                rank += (cj > cst || (cj == cst && j < k)) ? 1 : 0;
            }
        }
        order[rank] = k;
    }
    PIPE_LDS_BARRIER();
    auto dwin = glb(pa.dwin) + ((size_t)rep * 2 + buf) * Kb * qpad;
    auto scal = glb(pa.scal) + ((size_t)rep * 2 + buf) * Kb * 4;
    auto pval = glb(pa.pval) + ((size_t)rep * 2 + buf) * Kb * MCSAS_MAX_ACTIVE;
    auto povf = glb(pa.povf) + ((size_t)rep * 2 + buf) * Kb;
    if (MCSAS_TUNE_BITS(a) & 16) return;                                  // diagnostic: no window rows
    for (int pulls = 0; pulls <= Kb; ++pulls) {               // (a wave can draw at most every row of the window: the loop ends whatever the counter holds)
        int idx = 0;
        if (lane == 0) idx = __hip_atomic_fetch_add(rowq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx < 0 || idx >= nvalid) break;                  // (the ranks behind nvalid are the steps behind max_iter)
        const int k = __builtin_amdgcn_readfirstlane(order[idx]);
        Contrib<M> cnew;
        {
            double tmp[CON];
#pragma unroll
            for (int i = 0; i < CON; ++i) tmp[i] = readlane_f64(rec[(size_t)k * REC + i], 0);   // one address for the wave: into scalar registers
            __builtin_memcpy(&cnew, tmp, sizeof(Contrib<M>));
        }
        const int oslot = __builtin_amdgcn_readfirstlane(rslot[2 * k]), sslot = __builtin_amdgcn_readfirstlane(rslot[2 * k + 1]);
        const auto nrow = cache + (size_t)sslot * qpad + lane;
        const auto dr = dwin + (size_t)k * qpad + lane;
        // d = new - old and the three sums that do not depend on ft: a = Σ w d, e = Σ wI d, g = Σ w d².  Every q slot is
        // consumed the moment the evaluator has it (RowEval::run_each) and nothing of the row stays in registers across the
        // evaluation of the next slot (the row arrays used to be spilled to scratch around every slot: 20-30 KB per step); the
        // `old` value of a slot is requested one slot ahead and lands while that slot is evaluated.
        const auto orow = cache + (size_t)oslot * qpad + lane;
        double s1 = 0., s2 = 0., s3 = 0.;
        double o_ahead = orow[0];
        RowEval<M, QPL>::run_each(cnew, qt, lane, [&](int j, double v) {
            const int iq = lane + WAVE * j;
            const double o = o_ahead;
            o_ahead = orow[WAVE * (j + 1 < QPL ? j + 1 : j)];
            nrow[WAVE * j] = v;
            const double dj = v - o;
            dr[WAVE * j] = dj;
            const double wd = lw[iq] * dj;
            s1 += wd; s2 = fma(lwI[iq], dj, s2); s3 = fma(wd, dj, s3);
        });
        wave_sum3(s1, s2, s3);
        if (lane == 0) { scal[k * 4 + 0] = s1; scal[k * 4 + 1] = s2; scal[k * 4 + 2] = s3; }
        if (lane < P) pval[k * MCSAS_MAX_ACTIVE + lane] = rec[(size_t)k * REC + CON + lane];
        if (lane == 0) povf[k] = (int)rec[(size_t)k * REC + CON + MCSAS_MAX_ACTIVE];
    }
}

// The block's own chain first; then it HELPS.  The launch holds more producer workgroups than the chip has CUs (so that the CUs the
// scan blocks leave after a tenth of a tick are taken over), the chains do not get their workgroups at the same time, and chains that
// have converged — or wait for their next attempt — need none: a block whose queue is empty looks at what is left in EVERY chain's
// queue (one chain per thread: schedule record and counter), and joins one picked with probability proportional to the rows left
// (by a hash of the block index, so that the helpers spread like the work) if that is worth the 500 instructions per step of working
// out that chain's proposals again.  The tick then ends when the rows of ALL chains are done, and the last chains of an analysis that
// runs to its criterion get the whole chip.  Bounded: PIPE_HELP_TRIES visits per block, none once every queue is (nearly) empty.
constexpr int PIPE_HELP_TRIES = 8;
constexpr int PIPE_HELP_MIN_ROWS = 4;
template <int M, int QPL>
__device__ __forceinline__ void pipe_prod_rowq(const PipeArgs &pa, const PipeHot &hot, double *lds, const QTables &qt, int rep, int by, int gy, int t,
                                               const PipeSnap &sn, bool own) {
    const int tid = threadIdx.x, lane = tid & 63, R = hot.n_reps, N = hot.n_contrib, Kb = hot.kb;
    if (own) pipe_rowq_visit<M, QPL>(pa, hot, lds, qt, rep, by, gy, t, sn, false);
    if (!pa.g.help) return;
    int32_t *box = reinterpret_cast<int32_t *>(lds + pa.g.gram_off);       // (the 16 doubles ahead of the proposal records)
    int32_t *rem = reinterpret_cast<int32_t *>(lds + pa.g.rec_off);        // [R] rows left per chain (the records' place: pipe_geometry sets `help` only if they fit)
    for (int tries = 0; tries < PIPE_HELP_TRIES; ++tries) {
        __syncthreads();                                          // every wave is done with the records of the last visit
        for (int c = tid; c < R; c += PIPE_BLOCK) {
            int r = 0;
            const PipeSnap cs = load_snap(&hot.chains[c].snap[t & 1]);
            if (cs.alive && t >= cs.t_init) {
                int total = N;
                if (t > cs.t_init) {
                    const int64_t left = hot.max_iter - ((int64_t)t - cs.t_init - 1) * Kb;
                    total = left >= Kb ? Kb : (left > 0 ? (int)left : 0);
                }
                r = total - __hip_atomic_load(pa.rowq + (size_t)c * 2 + (t & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r < PIPE_HELP_MIN_ROWS) r = 0;
            }
            rem[c] = r;
        }
        __syncthreads();
        if (tid < 64) {
            const int chunk = (R + 63) / 64, lo = lane * chunk, hi = min(R, lo + chunk);
            int sum = 0;
            for (int c = lo; c < hi; ++c) sum += rem[c];
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (lane == 0) box[0] = -1;
            if (total > 0) {
                const uint32_t hsh = ((uint32_t)blockIdx.x * 2654435761u) ^ ((uint32_t)(tries + 1) * 0x9E3779B9u) ^ ((uint32_t)t * 0x85EBCA6Bu);
                const int pos = (int)((hsh >> 8) % (uint32_t)total);
                if (pos >= incl - sum && pos < incl) {            // exactly one lane
                    int acc = incl - sum, pick = lo;
                    for (int c = lo; c < hi; ++c) { if (pos < acc + rem[c]) { pick = c; break; } acc += rem[c]; }
                    box[0] = pick;
                }
            }
        }
        __syncthreads();
        const int target = box[0];
        if (target < 0) break;
        const PipeSnap cs = load_snap(&hot.chains[target].snap[t & 1]);
        pipe_rowq_visit<M, QPL>(pa, hot, lds, qt, target, by, gy, t, cs, true);
    }
}

template <int M, int QPL, bool RQ>                            // RQ: rows pulled from a queue (PipeGeom::rowq), a kernel of its own
__device__ __forceinline__ void pipe_prod_block(const PipeArgs &pa, const PipeHot &hot, double *lds, int rep, int by, int gy, int t) {
    const ChainArgs &a = pa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WPB = PIPE_BLOCK / 64;
    const int N = hot.n_contrib, P = hot.n_active, qpad = hot.qpad, Kb = hot.kb;
    const int64_t max_iter = hot.max_iter;                    // (a field of the argument block read inside the row loop would be a global load + full wait per row)
    // the data tables do not depend on the chain's schedule record: both round trips run side by side
    constexpr int QTB = (QPL * 64 + PIPE_BLOCK - 1) / PIPE_BLOCK;
    double tq[QTB], tw[QTB], twI[QTB], tq3[QTB];
#pragma unroll
    for (int x = 0; x < QTB; ++x) {
        const int i = tid + PIPE_BLOCK * x < qpad ? tid + PIPE_BLOCK * x : 0;
        tq[x] = glb(hot.q)[i]; tw[x] = glb(hot.w)[i]; twI[x] = glb(hot.wI)[i]; tq3[x] = glb(hot.q3inv)[i];
    }
    PIPE_TL_CLOCK(c_entry);                                   // (hot arguments in registers, loads issued)
    const PipeSnap sn = load_snap(&hot.chains[rep].snap[t & 1]);
    const bool own = sn.alive && t >= sn.t_init;
    if (!own && !(RQ && pa.g.help)) return;                   // (row queues: a block whose chain has nothing to do this tick helps the others)
    PIPE_TL_CLOCK(c_snap);

    double *lq = lds, *lw = lds + qpad, *lwI = lds + 2 * qpad, *lq3 = lds + 3 * qpad, *tab = lds + 4 * qpad;
#pragma unroll
    for (int x = 0; x < QTB; ++x) {
        const int i = tid + PIPE_BLOCK * x;
        if (i < qpad) { lq[i] = tq[x]; lw[i] = tw[x]; lwI[i] = twI[x]; lq3[i] = tq3[x]; }
    }
    PIPE_TL_CLOCK(c_tab);
    Contrib<M>::fill_table(a.model, tab, tid, PIPE_BLOCK);
    if (tid == 0) *reinterpret_cast<int32_t *>(lds + pa.g.gram_off + 16) = 0;   // lazy rows: the block's stale-row count
    __syncthreads();
    PIPE_TL_CLOCK(c_bar);
    PIPE_TL_PUT(pa, t, 22, c_entry); PIPE_TL_PUT(pa, t, 23, c_snap); PIPE_TL_PUT(pa, t, 24, c_tab); PIPE_TL_PUT(pa, t, 25, c_bar);
    const QTables qt = make_qtables<M>(a.model, lq, lq3, tab);
    if constexpr (RQ) { pipe_prod_rowq<M, QPL>(pa, hot, lds, qt, rep, by, gy, t, sn, own); return; }
    auto rset = glb(a.rset) + (size_t)rep * N * P;
    auto cache = glb(a.cache) + (size_t)rep * a.cache_rows * qpad;
    const DrawSource src{a.replay ? a.replay + (size_t)rep * a.replay_len : nullptr, a.replay_len, a.seed,
                         (uint32_t)(a.rep_offset + rep)};
    auto slot_of = glb(pa.slot_of) + (size_t)rep * N;
    auto stage = glb(pa.stage_slot) + (size_t)rep * 2 * Kb;
    auto row_valid = glb(pa.row_valid) + (size_t)rep * N;
    const int gw = by * WPB + wave, nw = gy * WPB;          // this wave's index among the chain's producer waves

    if (t == sn.t_init) {
        // ---- initial parameter set of the attempt (mcsas.py:310-319): rows n = gw*64 + lane + 64*nw*i
        for (int i = tid + by * PIPE_BLOCK; i < N; i += PIPE_BLOCK * gy) { slot_of[i] = i; row_valid[i] = 1; }
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

    if (MCSAS_TUNE_BITS(a) & 16) return;                                  // diagnostic: no window rows
    if constexpr (pipe_light_model_v<M>) if (pa.g.overlap || pa.g.gram_lds) {   // (rows with an integral never take this path: not instantiated for them)
        // ---- overlapped producer.  The block's rows are nsb sub-windows of W; phase ss = the rows of sub-window ss, every
        // wave its share, d = new - old straight to the window buffer.  The Gram block of sub-window ss - 1 is worked off
        // in units BETWEEN the rows of phase ss (matrix pipe beside the vector pipe: while one wave of a SIMD is inside a run
        // of MFMAs its partner has the vector issue slots to itself), its operands read back from the window buffer.
        // One barrier per phase, and it waits for no memory: a wave passes B(ss - 1) — "the rows of ss - 1 are visible to the
        // workgroup" — behind its FIRST row of phase ss, after a counted wait that covers exactly its stores of phase
        // ss - 1 (the counter is in order: everything older than that row's own stores has completed by then).  The partial
        // tiles of a block are parked in LDS when a wave has done its last unit and summed by all threads behind the next
        // barrier (two reduction buffers, by parity).  Only the last sub-window's Gram block runs with nothing beside it.
        const int W = pa.g.w, nsb = pa.g.sub_per_block, BR = nsb * W;
        // Rows per wave and sub-window: W / 8 on average; tuning bits 19-20 shift rows from the four waves that share
        // their SIMDs with an older wave (4-7) to the older ones (0-3): 0 = equal shares, 1 / 2 = one / two rows.
        // (the LDS variant's default is one row: the SIMD arbitrates oldest-first, so with equal shares the older wave is done
        // early and the younger one finishes the phase alone, latency-bound — measured 4 + 2 rows 3.92 ms, 3 + 3 4.03, 5 + 1 4.2;
        // bits 19-20 = 3 there: equal shares)
        const int rw_even = W >> 3, skew_bits = (MCSAS_TUNE_BITS(a) >> 19) & 3;
        const int skew_req = pa.g.gram_lds ? (skew_bits == 0 ? 1 : (skew_bits == 3 ? 0 : skew_bits)) : skew_bits;
        const int skew = skew_req < rw_even ? skew_req : rw_even - 1;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int RW = wv < 4 ? rw_even + skew : rw_even - skew;                        // my rows per sub-window
        const int rbase = wv < 4 ? wv * (rw_even + skew) : 4 * (rw_even + skew) + (wv - 4) * (rw_even - skew);   // my first row in a sub-window
        const int buf = t & 1;
        const int64_t w = (int64_t)t - sn.t_init - 1;
        const int64_t sb0 = w * Kb + (int64_t)by * BR;                                 // global step of the block's first row
        auto dwin = glb(pa.dwin) + ((size_t)rep * 2 + buf) * Kb * qpad;
        auto scal = glb(pa.scal) + ((size_t)rep * 2 + buf) * Kb * 4;
        auto pval = glb(pa.pval) + ((size_t)rep * 2 + buf) * Kb * MCSAS_MAX_ACTIVE;
        auto povf = glb(pa.povf) + ((size_t)rep * 2 + buf) * Kb;
        auto gwin = glb(pa.gwin) + ((size_t)rep * 2 + buf) * Kb * W;
        double *gred = lds + pa.g.gram_off + 16;                                       // [2][8 waves][PIPE_GRAM_NT_MAX][256]
        constexpr size_t GRED = (size_t)PIPE_WAVES * PIPE_GRAM_NT_MAX * 256;
        const int nmine = nsb * RW;                                                    // my rows (<= 8), lane l <-> my l-th row
        const bool no_gram = MCSAS_TUNE_BITS(a) & 64;                                              // diagnostic: no Gram blocks (uniform)
        const int lrow = (lane / RW) * W + rbase + (lane % RW);                        // its offset in the block
        const bool lazy = pa.g.lazy_rows;
        PIPE_TLX_MARK(pa, t, 0);
        // ---- lazy rows: the block's stale `old` rows (their last proposal, N steps ago, was accepted: ~6 % of them) are
        // evaluated again from the parameter set, one q per thread and row — an eighth of a wave's row time for the whole
        // block, and no wave ends up with more rows than the others — and written back to the row cache.  Lanes 32 + l of
        // a wave mirror its lanes l: the same rows, their `old` side — validity flag and parameter set in one round trip
        // (under way while the proposals are drawn), and ONE prepare() call serves the proposals and the old sets.
        constexpr int CON = 12;                                   // doubles per Contrib record in LDS
        static_assert(sizeof(Contrib<M>) <= 8 * CON && sizeof(Contrib<M>) % 8 == 0, "Contrib record");
        int32_t *stl = reinterpret_cast<int32_t *>(gred);         // [0] count (zeroed before the tables' barrier), then the stale contributions
        double *scon = gred + 64;                                 // their Contrib records
        double prow[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
        const int l2 = lane - 32;
        const bool old_lane = lazy && l2 >= 0 && l2 < nmine;
        int stale_r = -1;
        if (old_lane) {
            const int lrow_o = (l2 / RW) * W + rbase + (l2 % RW);
            if (sb0 + lrow_o < max_iter) {
                const int r = (int)((sb0 + lrow_o) % N);
                const int v = row_valid[r];
#pragma unroll
                for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p) if (p < P) prow[p] = rset[(size_t)r * P + p];
                if (!v) stale_r = r;
            }
        }
        int pov = 0, my_oslot = 0, my_sslot = 0;
        {
            const int r = (int)((sb0 + lrow) % N);
            if (lane < nmine) {
                if (lazy) my_oslot = r;
                else { my_oslot = slot_of[r]; my_sslot = stage[buf * Kb + by * BR + lrow]; }
            }
            const int64_t sl = sb0 + lrow;
#pragma unroll
            for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p)
                if (p < P) {
                    double u = 0.5;
                    if (lane < nmine && sl < max_iter) u = src.at(sn.step_base + (uint64_t)sl * P + p, pov);
                    const double pv = gen_transform(a.gen_kind[p], u) * (a.gen_hi[p] - a.gen_lo[p]) + a.gen_lo[p];
                    if (!(old_lane && sb0 + ((l2 / RW) * W + rbase + (l2 % RW)) < max_iter)) prow[p] = pv;
                }
        }
        Contrib<M> prop;
        prop.prepare(a.model, prow);
        PIPE_TLX_MARK(pa, t, 1);
        int nst = 0;
        if (lazy) {
            if (stale_r >= 0) {
                const int e = atomicAdd(&stl[0], 1);
                stl[1 + e] = stale_r;
                double tmp[CON] = {};
                __builtin_memcpy(tmp, &prop, sizeof(Contrib<M>));
#pragma unroll
                for (int i = 0; i < (int)(sizeof(Contrib<M>) / 8); ++i) scon[e * CON + i] = tmp[i];
            }
            PIPE_LDS_BARRIER();
            nst = stl[0];
            for (int i = 0; i < nst; ++i) {                       // (list order varies from run to run, the rows do not depend on it)
                const int r = stl[1 + i];
                Contrib<M> c;
                {
                    double tmp[CON];
#pragma unroll
                    for (int x = 0; x < (int)(sizeof(Contrib<M>) / 8); ++x) tmp[x] = scon[i * CON + x];
                    __builtin_memcpy(&c, tmp, sizeof(Contrib<M>));
                }
#pragma unroll
                for (int x = 0; x < QTB; ++x) {
                    const int iq = tid + PIPE_BLOCK * x;
                    if (iq < qpad) cache[(size_t)r * qpad + iq] = pipe_point_intensity<M>(c, lq[iq], lq3[iq], tab);
                }
                if (tid == 0) row_valid[r] = 1;
            }
            if (nst) __syncthreads();                             // the refreshed rows have landed before the row loop loads them (uniform)
            else PIPE_LDS_BARRIER();                              // (the stale list shares the reduction buffer: read by all before it is reused)
        }
        PIPE_TLX_MARK(pa, t, 2);
        PIPE_TLX_MARK(pa, t, 3);
        // proposals and replay-overflow flags of all my rows: one store per wave (lane l <-> my l-th row)
        if (lane < nmine) {
            const int k = by * BR + lrow;
#pragma unroll
            for (int p = 0; p < MCSAS_MAX_ACTIVE; ++p) if (p < P) pval[k * MCSAS_MAX_ACTIVE + p] = prow[p];
            povf[k] = pov;
        }
        auto nvalid_of = [&](int ss) {
            const int64_t left = max_iter - (w * Kb + (int64_t)(by * nsb + ss) * W);
            return left >= W ? W : (left > 0 ? (int)left : 0);
        };
        if (pa.g.gram_lds) {
            // ---- default: sub-window by sub-window — every wave evaluates its rows of the sub-window (d also into the LDS row
            // buffer), barrier, the eight waves take the Gram block from LDS, next sub-window.  Only the LDS traffic is waited
            // for at the barriers: the rows' global stores drain behind the MFMAs.
            const int dstr = qpad + PIPE_DROW_PAD;
            double *dbuf = lds + pa.g.drow_off;
            double ocur[QPL], onext[QPL];
            {
                const auto orow0 = cache + (size_t)__builtin_amdgcn_readlane(my_oslot, 0) * qpad + lane;
#pragma unroll
                for (int j = 0; j < QPL; ++j) ocur[j] = orow0[WAVE * j];
            }
            PIPE_PIN_ROW(ocur);                                   // (a pending load carried into the loop would be waited for at its head, every iteration)
            for (int ss = 0; ss < nsb; ++ss) {
                for (int jr = 0; jr < RW; ++jr) {
                    const int l = ss * RW + jr, bl = __builtin_amdgcn_readfirstlane(l);
                    const int kl = ss * W + rbase + jr, k = by * BR + kl;
                    const Contrib<M> cnew = prop.bcast(bl);
                    const int sslot = __builtin_amdgcn_readlane(my_sslot, bl);
                    double d[QPL], nwv[QPL];
                    {
                        const int bn = __builtin_amdgcn_readfirstlane(l + 1 < nmine ? l + 1 : l);
                        const auto orow = cache + (size_t)__builtin_amdgcn_readlane(my_oslot, bn) * qpad + lane;
#pragma unroll
                        for (int j = 0; j < QPL; ++j) onext[j] = orow[WAVE * j];
                    }
                    // (a row behind max_iter — the last window of a run only — is evaluated like any other: its proposal is the
                    // generators' midpoint, its stores land in slots nobody reads, the Gram block masks it)
                    const auto nrow = cache + (size_t)sslot * qpad + lane;
                    const auto dr = dwin + (size_t)k * qpad + lane;
                    double *dl = dbuf + (size_t)(rbase + jr) * dstr + lane;
                    RowEval<M, QPL>::run(cnew, qt, lane, nwv);
#ifdef MCSAS_STAMPS                                               /* marks 4..11: my first four rows — evaluated / `old` rows there and stores out */
                    if (l < 4) { PIPE_PIN_ROW(nwv); PIPE_TL_MARK(pa, t, 4 + 2 * l); }
#endif
                    PIPE_PIN_ROW(ocur); PIPE_PIN_ROW(onext); PIPE_PIN_ROW(nwv);   // both `old` rows have landed before the first store is issued
                    double s1 = 0., s2 = 0.;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        const int iq = lane + WAVE * j;
                        if (!lazy) nrow[WAVE * j] = nwv[j];
                        d[j] = nwv[j] - ocur[j];
                        dr[WAVE * j] = d[j];
                        dl[WAVE * j] = d[j];
                        s1 = fma(lw[iq], d[j], s1); s2 = fma(lwI[iq], d[j], s2);
                    }
                    // a = sum w d (even lanes), e = sum wI d (odd lanes); g = sum w d^2 is the Gram block's diagonal
                    const double ae = wave_sum2_split(s1, s2, lane);
                    if (lane < 2) scal[(size_t)k * 4 + lane] = ae;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) ocur[j] = onext[j];
#ifdef MCSAS_STAMPS
                    if (l < 4) PIPE_TL_MARK(pa, t, 5 + 2 * l);
#endif
                }
                PIPE_TL_MARK(pa, t, 2 * (ss < 4 ? ss : 3));
                if (!no_gram) PIPE_LDS_BARRIER();                 // the sub-window's rows are in LDS
                const int nvalid = nvalid_of(ss);
                if (nvalid > 0 && !no_gram)                       // uniform in the block
                    pipe_prod_gram_lds<QPL>(dbuf, dstr, W, nvalid, lw, gred, gwin + (size_t)(by * nsb + ss) * W * W,
                                            scal + (size_t)(by * BR + ss * W) * 4);
                // (the next sub-window's rows overwrite dbuf only behind the reduction's first barrier, which every wave
                // passes after its last operand read; gred is written again behind the next rows -> Gram barrier)
                PIPE_TL_MARK(pa, t, 2 * (ss < 4 ? ss : 3) + 1);
            }
            PIPE_TL_MARK(pa, t, 17);
            return;
        }
#ifdef MCSAS_TUNING                                           // the overlapped variant (pipe_geometry: bit 18) is a measurement build's
        const int ntiles = pipe_gram_tiles(W);
        constexpr int UN = QPL;                                   // Gram units (8 q each) of my q slice
        // the tile scheme is uniform for the launch; everything below is compiled once per scheme
        auto rows_and_gram = [&](auto scheme_t, auto scheme_pack) {
            constexpr int T = decltype(scheme_t)::value;
            constexpr bool PACK = decltype(scheme_pack)::value;
            constexpr int UCAP = PipeGramScheme<T, PACK>::UCAP;
            v4f64 gacc[PIPE_GRAM_NT_MAX];
            v2f64 G[PIPE_GRAM_PREF];
            for (int ss = 0; ss < nsb; ++ss) {
                const int nv_prev = ss > 0 ? nvalid_of(ss - 1) : 0;
                const bool gram_live = ss > 0 && !no_gram && nv_prev > 0;             // uniform in the block
                const auto dprev = dwin + (size_t)(by * BR + (ss > 0 ? ss - 1 : 0) * W) * qpad;
                for (int jr = 0; jr < RW; ++jr) {
                    const int l = ss * RW + jr, bl = __builtin_amdgcn_readfirstlane(l);
                    const int kl = ss * W + rbase + jr, k = by * BR + kl;
                    const Contrib<M> cnew = prop.bcast(bl);
                    const int sslot = __builtin_amdgcn_readlane(my_sslot, bl);
                    // the `old` row of THIS step is requested here and used behind the row evaluation, which is longer than the
                    // round trip (a row of lookahead would hold another QPL doubles per lane across the evaluation, beside the
                    // Gram operands and accumulators)
                    double d[QPL], nwv[QPL], ocur[QPL];
                    {
                        const auto orow = cache + (size_t)__builtin_amdgcn_readlane(my_oslot, bl) * qpad + lane;
#pragma unroll
                        for (int j = 0; j < QPL; ++j) ocur[j] = orow[WAVE * j];
                    }
                    // my units of block ss - 1 that go with this row (none with the first row of a phase: B(ss - 1) comes
                    // behind it): their operands are requested now and land while the row is evaluated
                    const int u0 = (RW > 1 && jr > 0) ? UN * (jr - 1) / (RW - 1) : 0, u1 = (RW > 1 && jr > 0) ? UN * jr / (RW - 1) : 0;
                    const int npre = u1 - u0 < UCAP ? u1 - u0 : UCAP;
                    const bool chunk = gram_live && npre > 0;                        // uniform in the wave
                    if (chunk) pipe_gram_fetch<QPL, T, PACK>(dprev, qpad, nv_prev, wv, u0, npre, G);
                    else {
#pragma unroll
                        for (int i = 0; i < PIPE_GRAM_PREF; ++i) asm volatile("" : "=v"(G[i]));   // (defined on both paths: no copy at the join)
                    }
                    // (a row behind max_iter — the last window of a run only — is evaluated like any other: its proposal is the
                    // generators' midpoint, its stores land in slots nobody reads, the Gram block masks it; no branch around the
                    // row means no join at which the compiler would wait for this row's stores)
                    const auto nrow = cache + (size_t)sslot * qpad + lane;
                    const auto dr = dwin + (size_t)k * qpad + lane;
                    RowEval<M, QPL>::run(cnew, qt, lane, nwv);
                    PIPE_PIN_ROW(ocur); PIPE_PIN_ROW(nwv);                        // the `old` row has landed before the first store is issued
                    if (chunk) {
                        pipe_gram_consume<QPL, T, PACK>(G, nv_prev, lw, wv, u0, npre, gacc);
                        if (u0 + npre < u1) pipe_gram_units<QPL, T, PACK>(dprev, qpad, nv_prev, lw, wv, u0 + npre, u1, gacc);
                        if (u1 == UN) pipe_gram_park(gacc, ntiles, gred + (size_t)((ss - 1) & 1) * GRED, wv, lane);
                    }
                    double s1 = 0., s2 = 0.;
#pragma unroll
                    for (int j = 0; j < QPL; ++j) {
                        const int iq = lane + WAVE * j;
                        if (!lazy) nrow[WAVE * j] = nwv[j];
                        d[j] = nwv[j] - ocur[j];
                        dr[WAVE * j] = d[j];
                        s1 = fma(lw[iq], d[j], s1); s2 = fma(lwI[iq], d[j], s2);
                    }
                    // a = sum w d (even lanes), e = sum wI d (odd lanes); g = sum w d^2 is the Gram block's diagonal
                    const double ae = wave_sum2_split(s1, s2, lane);
                    if (lane < 2) scal[(size_t)k * 4 + lane] = ae;
                    PIPE_TL_MARK(pa, t, (ss < 2 ? ss : 1) * 4 + (jr < 3 ? jr : 3));
                    if (ss > 0 && !no_gram && jr == 0) {
                        // B(ss - 1): my stores of phase ss - 1 are older than this row's QPL (or more) stores
                        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(QPL < 8 ? QPL : 8) : "memory");
                        PIPE_LDS_BARRIER();
                        if (ss > 1 && nvalid_of(ss - 2) > 0)                          // every wave's tiles of block ss - 2 are parked: sum them
                            pipe_gram_sum_store(W, gred + (size_t)((ss - 2) & 1) * GRED, gwin + (size_t)(by * nsb + ss - 2) * W * W,
                                                scal + (size_t)(by * BR + (ss - 2) * W) * 4);
                        if (RW == 1 && nv_prev > 0) {                                 // one row per wave and phase: nothing to put the units beside
                            pipe_gram_units<QPL, T, PACK>(dprev, qpad, nv_prev, lw, wv, 0, UN, gacc);
                            pipe_gram_park(gacc, ntiles, gred + (size_t)((ss - 1) & 1) * GRED, wv, lane);
                        }
                    }
                }
                PIPE_TL_MARK(pa, t, 8 + (ss < 3 ? ss : 3));
            }
            // ---- the tail: block nsb - 2 is summed, the last sub-window's Gram block has nothing to run beside
            if (!no_gram) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PIPE_TL_MARK(pa, t, 12);
                PIPE_LDS_BARRIER();                                   // B(nsb - 1)
                PIPE_TL_MARK(pa, t, 13);
                if (nsb > 1 && nvalid_of(nsb - 2) > 0)
                    pipe_gram_sum_store(W, gred + (size_t)((nsb - 2) & 1) * GRED, gwin + (size_t)(by * nsb + nsb - 2) * W * W,
                                        scal + (size_t)(by * BR + (nsb - 2) * W) * 4);
                const int nv = nvalid_of(nsb - 1);
                if (nv > 0) {                                         // uniform in the block
                    const auto dlast = dwin + (size_t)(by * BR + (nsb - 1) * W) * qpad;
                    // batches of UCAP units, the next batch's operands requested before this one's MFMAs
                    v2f64 G2[PIPE_GRAM_PREF];
                    PIPE_TL_MARK(pa, t, 14);
                    pipe_gram_fetch<QPL, T, PACK>(dlast, qpad, nv, wv, 0, UN < UCAP ? UN : UCAP, G);
                    for (int u = 0; u < UN; u += 2 * UCAP) {
                        const int n0 = UN - u < UCAP ? UN - u : UCAP;
                        const int un = u + UCAP, n1 = un < UN ? (UN - un < UCAP ? UN - un : UCAP) : 0;
                        if (n1 > 0) pipe_gram_fetch<QPL, T, PACK>(dlast, qpad, nv, wv, un, n1, G2);
                        pipe_gram_consume<QPL, T, PACK>(G, nv, lw, wv, u, n0, gacc);
                        const int u2 = u + 2 * UCAP, n2 = u2 < UN ? (UN - u2 < UCAP ? UN - u2 : UCAP) : 0;
                        if (n2 > 0) pipe_gram_fetch<QPL, T, PACK>(dlast, qpad, nv, wv, u2, n2, G);
                        if (n1 > 0) pipe_gram_consume<QPL, T, PACK>(G2, nv, lw, wv, un, n1, gacc);
                    }
                    PIPE_TL_MARK(pa, t, 15);
                    pipe_gram_park(gacc, ntiles, gred + (size_t)((nsb - 1) & 1) * GRED, wv, lane);
                    PIPE_LDS_BARRIER();
                    PIPE_TL_MARK(pa, t, 16);
                    pipe_gram_sum_store(W, gred + (size_t)((nsb - 1) & 1) * GRED, gwin + (size_t)(by * nsb + nsb - 1) * W * W,
                                        scal + (size_t)(by * BR + (nsb - 1) * W) * 4);
                }
            }
        };
        if (W == 24) rows_and_gram(std::integral_constant<int, 2>{}, std::integral_constant<bool, true>{});
        else if (W <= 16) rows_and_gram(std::integral_constant<int, 1>{}, std::integral_constant<bool, false>{});
        else rows_and_gram(std::integral_constant<int, 2>{}, std::integral_constant<bool, false>{});
#endif
        PIPE_TL_MARK(pa, t, 17);
        return;
    }

}

// ------------------------------------------------------------------------------------ scanner
// LDS: two Gram blocks, ft and w*ft, the window's scalars, h of the current sub-window, flags and slot tables
template <int M, int QPL, int RPS, bool RQ>                    // RPS = rows per wave and sub-window (W / 8), compile time: see `request`
__device__ __forceinline__ void pipe_scan_block(const PipeArgs &pa, double *lds, int rep, int t, int stop_now) {
    static_assert(PIPE_GRAM_TILES_PER_ROUND * 256 == PIPE_BLOCK, "Gram reduction maps one thread to one tile element");
    const ChainArgs &a = pa.c;
    // the wave index is wave-uniform: keep it (and the row bookkeeping that hangs on it) on the scalar unit
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.n_contrib, P = a.model.n_active, qpad = a.qpad, Kb = pa.g.kb, W = pa.g.w;
    constexpr int T = PIPE_BLOCK;
    MCSAS_GLOBAL PipeChain &ch = glb(pa.chains)[rep];
    if (ch.done) return;                                      // uniform for the block
    if (RQ && tid == 0) pa.rowq[(size_t)rep * 2 + (t & 1)] = 0;   // the queue of PROD(t + 2) (this launch's producers use the other parity)
    MCSAS_STAMP_DECL(sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0);
    MCSAS_STAMP(sb0);
#ifdef MCSAS_STAMPS
    const uint64_t wc0 = wall_clock64();
#endif
    const PipeSnap sn = load_snap(&pa.chains[rep].snap[(t + 1) & 1]);   // the record in force for tick t (written at t-1; host for t = 0)

    double *rowbuf = lds;                                     // [W][qpad] d rows of the current sub-window (accepted ones are applied from here)
    double *Gl = rowbuf + (size_t)W * qpad;                   // [2][W*W] Gram block of the current / next sub-window
    double *lft = Gl + 2 * (size_t)W * W;                     // [qpad] ft, q-indexed
    double *lwft = lft + qpad;                                // [qpad] w * ft
    double *ssub = lwft + qpad;                               // [Kb][4] a, e, g of every step of the window
    double *hsub = ssub + (size_t)Kb * 4;                     // [64] h of the current sub-window, by step offset
    int32_t *osub = reinterpret_cast<int32_t *>(hsub + 64);   // [Kb] replay-overflow flags
    int32_t *lstage = osub + Kb, *lslot = lstage + Kb;        // [Kb] spare row slot of step k / row slot of its contribution
    int32_t *lacc = lslot + Kb;                               // [Kb + 1] accepted steps of this window, count in lacc[Kb]
    int32_t *sacc = lacc + Kb + 1;                            // [1 + 64] this sub-window: count, then the accepted steps' offsets in it
    int32_t *ctl = sacc + 1 + 64;                             // [4]: [2] = live
    double *lwq = reinterpret_cast<double *>((reinterpret_cast<unsigned long long>(ctl + 4) + 7ull) & ~7ull);   // [qpad] w (row queues: the Gram blocks are taken here)
    auto gft = glb(pa.ft) + (size_t)rep * qpad, gwft = glb(pa.wft) + (size_t)rep * qpad;
    auto rset = glb(a.rset) + (size_t)rep * N * P;
    auto cache = glb(a.cache) + (size_t)rep * a.cache_rows * qpad;
    const int buf = t & 1;
    const auto dwin = glb((const double *)pa.dwin) + ((size_t)rep * 2 + buf) * Kb * qpad;
    const auto gwin = glb((const double *)pa.gwin) + ((size_t)rep * 2 + buf) * Kb * W;
    const auto scal = glb((const double *)pa.scal) + ((size_t)rep * 2 + buf) * Kb * 4;
    const auto pval = glb((const double *)pa.pval) + ((size_t)rep * 2 + buf) * Kb * MCSAS_MAX_ACTIVE;
    const auto povf = glb((const int32_t *)pa.povf) + ((size_t)rep * 2 + buf) * Kb;
    auto slot_of = glb(pa.slot_of) + (size_t)rep * N;
    auto stage = glb(pa.stage_slot) + ((size_t)rep * 2 + buf) * Kb;
    const auto gw_ = glb(a.w), gwI_ = glb(a.wI), gI_ = glb(a.I);
    const double nqd = (double)a.nq;

    // scanner-side chain state (meaningful in wave 0)
    FitResult cur{ch.A, ch.b, ch.chi2};
    double SC = ch.SC, SIC = ch.SIC, SCC = ch.SCC;
    int64_t num_iter = ch.num_iter, num_moves = ch.num_moves;
    int stopped = ch.stopped, overflow = 0;
    bool attempt_over = false;
    double Xwin = 0.;                                           // chi²·Q at the end of this tick's window (PipeGeom::resum_every)
    bool had_window = false;

    if (t < sn.t_init) {
        // nothing scheduled for this chain at this tick; just republish below
    } else if (t == sn.t_init) {
        // ---- model.calc over the initial set: rows summed in contribution order (scatteringmodel.py:90-101)
        // One q per thread, the N rows in batches of 16 loads (one wave walking the rows one dependent load at a time took
        // ~100 us at N = 400: a fortieth of a 20 000-step launch); every q is summed in contribution order, as before.
        for (int i = tid; i < qpad; i += T) {
            double f = 0.;
            int n = 0;
            for (; n + 16 <= N; n += 16) {
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = cache[(size_t)(n + k) * qpad + i];
#pragma unroll
                for (int k = 0; k < 16; ++k) f += v[k];
            }
            for (; n < N; ++n) f += cache[(size_t)n * qpad + i];
            lft[i] = f;
        }
        PIPE_LDS_BARRIER();                                    // (t == t_init for every thread of the block)
        if (wave == 0) {
            double ft[QPL];
#pragma unroll
            for (int j = 0; j < QPL; ++j) ft[j] = lft[lane + WAVE * j];
            double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const double wf = gw_[lane + WAVE * j] * ft[j];
                s1 += wf; s2 = fma(wf, ft[j], s2); s3 = fma(gwI_[lane + WAVE * j], ft[j], s3);
                gft[lane + WAVE * j] = ft[j]; gwft[lane + WAVE * j] = wf;
                lft[lane + WAVE * j] = ft[j];                 // the end-of-attempt code below reads ft from LDS
            }
            wave_sum3(s1, s2, s3);
            SC = s1; SCC = s2; SIC = s3;
            cur = solve_fit(a, SC, SCC, SIC);
            num_iter = 0; num_moves = 0;
            if (N <= 1 || a.max_iter <= 0 || !(cur.chi2 > a.conv_crit)) attempt_over = true;
        }
    } else {
        // ---- window w = t - t_init - 1, sub-window by sub-window
        const int64_t w = (int64_t)t - sn.t_init - 1;
        const int64_t budget = a.max_iter - w * Kb;
        const int kmax_all = budget < Kb ? (budget < 0 ? 0 : (int)budget) : Kb;
        const int ri0 = (int)((w * Kb) % N);
        const int nsub = (kmax_all + W - 1) / W;
        // The d rows travel HBM/L2 -> registers -> (dot product with w ft) -> LDS row buffer.  A wave owns the rows
        // wave, wave + 8, ... of a sub-window (RPS of them).
        // Row i of EVERY sub-window sits in register set i, and the row for the next sub-window is requested as soon as
        // the set has been used: RPS rows per wave are under way all the time, also across the decision and apply
        // phases.  RPS is a template parameter because a load whose target depends on a run-time choice (or sits
        // under a condition) becomes a load into scratch registers, a wait and a copy at the join: no prefetch.
        const int total_r = nsub * RPS;
        int r_load = 0;
        double rs[RPS][QPL];
        auto request = [&](double (&dst)[QPL]) {
            const int rr = r_load < total_r ? r_load : 0;      // (past the end: row 0 again, never used)
            const int k = (rr / RPS) * W + wave + 8 * (rr % RPS);
            load_row_pairs<QPL>(dwin + (size_t)(k < kmax_all ? k : 0) * qpad, lane, dst);
            ++r_load;
        };
#pragma unroll
        for (int i = 0; i < RPS; ++i) request(rs[i]);
        // Gram block of sub-window s -> LDS buffer s & 1 (W*W doubles, contiguous in HBM): the loads are issued at the
        // top of the previous sub-window and parked in registers, the LDS stores follow behind that sub-window's rows
        // (a load-store copy loop would drain every outstanding row load at its first store)
        constexpr int NG = (RPS * RPS * 64 / 2 + T - 1) / T;   // 16-byte pieces per thread (W*W / 2 pieces in all)
        v2f64 gtmp[NG];
        auto gram_fetch = [&](int s) {
            const auto src = gwin + (size_t)s * W * W;
#pragma unroll
            for (int x = 0; x < NG; ++x) {
                int i = 2 * (tid + T * x);
                if (i > W * W - 2) i = W * W - 2;              // (clamped, not skipped: see `request`)
                gtmp[x] = *(const MCSAS_GLOBAL v2f64 *)(src + i);
            }
        };
        auto gram_store = [&](int s) {
            double *dst = Gl + (size_t)(s & 1) * W * W;
#pragma unroll
            for (int x = 0; x < NG; ++x) {
                const int i = 2 * (tid + T * x);
                if (i < W * W) *reinterpret_cast<v2f64 *>(dst + i) = gtmp[x];
            }
        };
        constexpr bool gram_here = RQ;                         // rows with an integral: the Gram blocks are worked out below, from the rows in LDS
        if (nsub > 0 && !gram_here) { gram_fetch(0); gram_store(0); }
        // ft, w ft -> LDS; the thread's own q in the apply phase: q = tid (+ 512)
        constexpr int QT = (QPL * 64 + T - 1) / T;            // q per thread in the apply phase (1 or 2)
        double wq[QT];
#pragma unroll
        for (int x = 0; x < QT; ++x) {
            const int i = tid + T * x;
            wq[x] = 0.;
            if (i < qpad) { wq[x] = gw_[i]; lft[i] = gft[i]; lwft[i] = gwft[i]; if constexpr (RQ) lwq[i] = wq[x]; }
        }
        static_assert(PIPE_BLOCK >= 512, "one window step per thread: pipe_geometry caps Kb at PIPE_BLOCK");
        if (tid < kmax_all) {                                  // Kb <= PIPE_BLOCK = threads (pipe_geometry)
            osub[tid] = povf[tid];
            if (!pa.g.lazy_rows) {                             // (lazy rows never move: no slot tables)
                int r = ri0 + tid; if (r >= N) r -= N;
                lstage[tid] = stage[tid]; lslot[tid] = slot_of[r];
            }
        }
        for (int i = tid; i < kmax_all * 4; i += T) ssub[i] = scal[i];
        if (tid == 0) { lacc[Kb] = 0; sacc[0] = 0; }
        const double invSw = 1.0 / a.Sw, SIoSw = a.SI / a.Sw, Scen = a.SII - a.SI * a.SI / a.Sw;
        const int resum = pa.g.resum_every;
        const bool swap_slots = !pa.g.lazy_rows;
        double X = resum ? ch.X : cur.chi2 * nqd;
        bool touched = false, live = true;
        int num_acc_win = 0;
        if (wave == 0) {
            if (stop_now) stopped = 1;                         // McSAS.stop as the host saw it when it launched this tick
            if (lane == 0) ctl[2] = (!(cur.chi2 > a.conv_crit) || stopped) ? 0 : 1;   // `live`, shared by all waves
        }
        PIPE_LDS_BARRIER();
        live = ctl[2] != 0;
        double wftp[QPL];
        load_row_pairs_lds<QPL>(lwft, lane, wftp);
        // loop-invariant fit constants and the running sums, pinned in VGPRs (see MCSAS_IN_VGPR)
        double cSII = a.SII, cSI = a.SI, cScen = Scen, cSIoSw = SIoSw, cinvSw = invSw, cCrit = a.conv_crit, cnq = nqd;
        MCSAS_IN_VGPR(cSII); MCSAS_IN_VGPR(cSI); MCSAS_IN_VGPR(cScen); MCSAS_IN_VGPR(cSIoSw); MCSAS_IN_VGPR(cinvSw);
        MCSAS_IN_VGPR(cCrit); MCSAS_IN_VGPR(cnq);
        MCSAS_IN_VGPR(SC); MCSAS_IN_VGPR(SIC); MCSAS_IN_VGPR(SCC); MCSAS_IN_VGPR(X);
        const bool find_bg = a.find_bg, pos_bg = a.pos_bg, never_accept = MCSAS_TUNE_BITS(a) & 32;
#ifdef MCSAS_STAMPS
        int64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        MCSAS_STAMP_DECL(s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0);
        MCSAS_STAMP(sb1);
        for (int s = 0; s < nsub && live; ++s) {
            MCSAS_STAMP(s0);
            const int k0 = s * W;
            const int cnt = (kmax_all - k0) < W ? (kmax_all - k0) : W;
            if (!gram_here) gram_fetch(s + 1 < nsub ? s + 1 : s);   // (its LDS buffer was last read two sub-windows ago)
            // ---- my rows of this sub-window: h = Σ (w ft) d, and the row itself into the LDS row buffer
            double acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.;
            auto use_row = [&](int i, const double (&row)[QPL]) {
                double h0 = 0., h1 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; j += 2) {
                    h0 = fma(wftp[j], row[j], h0);
                    if (j + 1 < QPL) h1 = fma(wftp[j + 1], row[j + 1], h1);
                }
                const double hs = h0 + h1;
#pragma unroll
                for (int x = 0; x < 8; ++x) acc[x] = (x == i) ? hs : acc[x];
                const int g = wave + 8 * i;
                if (g < cnt) {
                    double *dst = rowbuf + (size_t)g * qpad;
                    if constexpr (QPL >= 2) {
#pragma unroll
                        for (int c = 0; c < QPL / 2; ++c)
                            *reinterpret_cast<v2f64 *>(dst + 128 * c + 2 * lane) = (v2f64){row[2 * c], row[2 * c + 1]};
                    } else {
                        dst[lane] = row[0];
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < RPS; ++i) { use_row(i, rs[i]); request(rs[i]); }
            {
                // eight sums for the price of ~1.5: lane l < 8 ends up with the total of acc[4 (l&1) + 2 ((l>>1)&1) + ((l>>2)&1)]
                const double tot = wave_sum8_transposed(acc, lane);
                const int c = 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
                if (lane < 8 && c < RPS && wave + 8 * c < cnt) hsub[wave + 8 * c] = tot;
            }
            if (!gram_here) gram_store(s + 1);                 // read at the earliest after B1 of the next sub-window
            MCSAS_STAMP(s1);
            PIPE_LDS_BARRIER();                                            // B1: hsub, the row buffer and this sub-window's Gram block complete
            if constexpr (gram_here) {
                // G[a][k] = Σ_q w d_a d_k of the sub-window's (eight) rows, which are all in the row buffer now: wave a takes row a
                // against every row, q = lane + 64 j, the eight sums reduced together.  (The producers' MFMA pass did this when one
                // workgroup evaluated the eight steps of a sub-window; with rows pulled from a queue no workgroup has them all.)
                if (wave < cnt) {
                    double ga[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) ga[x] = 0.;
                    // (not unrolled over q: sixteen slots of nine operands each in flight took the scan loop's registers — 929 spills at
                    // Q = 1024 and a sub-window in 40 us instead of 5.  Rows behind cnt are stale LDS: their sums are never read.)
#pragma nounroll
                    for (int j = 0; j < QPL; ++j) {
                        const int iq = lane + WAVE * j;
                        const double wa = lwq[iq] * rowbuf[(size_t)wave * qpad + iq];
#pragma unroll
                        for (int x = 0; x < 8; ++x) ga[x] = fma(wa, rowbuf[(size_t)x * qpad + iq], ga[x]);
                    }
                    const double tot = wave_sum8_transposed(ga, lane);
                    const int c = 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
                    if (lane < 8) Gl[(size_t)(s & 1) * W * W + (size_t)wave * W + c] = tot;
                }
                PIPE_LDS_BARRIER();
            }
            MCSAS_STAMP(s2);
            if (wave == 0) {
                // ---- the W decisions of the sub-window: lane g <-> step k0 + g
                __builtin_amdgcn_s_setprio(1);
                const int g = lane;
                const bool in = g < cnt;
                const int kg = in ? k0 + g : k0;
                double h = hsub[in ? g : 0];
                const double sc0 = ssub[kg * 4 + 0], sc1 = ssub[kg * 4 + 1], sc2 = ssub[kg * 4 + 2];
                const int ovg = osub[kg];
                const double *Gs = Gl + (size_t)(s & 1) * W * W;
                int nacc_sub = 0;
                // One round per accepted step: every remaining candidate (lanes in `cmask`) is judged against the current
                // state at once, the first that passes is taken, the state moves on, again from the step behind it.  A round is
                // a dependent chain on ONE wave — early in a chain, when most proposals pass, the rounds are the whole tick —
                // so everything that is not on that chain is kept out of it: the candidate and overflow masks are scalars, the
                // fit flags are compile-time (three copies of the loop), the accepted steps are recorded as a bit mask and
                // published to LDS once per sub-window, chi²·Q of every candidate is divided out beside its comparison.
                auto decide = [&](auto fb_t, auto pb_t) {
                    constexpr bool FB = decltype(fb_t)::value, PB = decltype(pb_t)::value;
                    const unsigned long long inmask = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
                    const unsigned long long ovm_all = __ballot(in && ovg) & inmask;
                    unsigned long long cmask = inmask, accm = 0ull, leftover = 0ull;
                    // The current state's chi²·Q enters a comparison as the pair (Na, Da), X = Sa - Na / Da, with (S - X, 1) at the
                    // head of a sub-window — "chi²_t < chi²" (mcsas.py:379), num² / den > Na / Da + (S - Sa), is taken across:
                    // num² Da > (Na + (S - Sa) Da) den.  With (S - X, 1) that is the very expression num² > (S - X) den; behind an
                    // accepted step it is that step's own (num², den), exact, and the division that gives X leaves the chain of
                    // dependent operations a round consists of (it is made once, when the sub-window is through).  S is the same
                    // for every candidate unless positiveBackground can switch a candidate to the uncentred sums.
                    double Sa = PB ? X : 0., Na = PB ? 0. : (FB ? cScen : cSII) - X, Da = 1.0;
                    MCSAS_IN_VGPR(Na); MCSAS_IN_VGPR(Da);
                    // Round 5: ONE exit (no candidate passes) and nothing in a round but the dependent chain itself — the step and
                    // overflow counts are taken from the masks behind the loop, the accepted candidate's own convergence test is made
                    // on its two numbers; a chain that ends inside the sub-window empties the candidate mask instead of leaving the
                    // loop (one more, empty, round at the very end of an attempt).  The loop went from ~125 to ~60 instructions a round.
                    for (;;) {
                        const double SCt = SC + sc0, SICt = SIC + sc1, SCCt = SCC + fma(2., h, sc2);
                        double S = cSII, num = SICt, den = SCCt;
                        if constexpr (FB) {
                            const double numc = fma(-cSIoSw, SCt, SICt), denc = fma(-(SCt * cinvSw), SCt, SCCt);
                            if constexpr (PB) {
                                const bool neg_b = fma(cSI, denc, -(numc * SCt)) < 0.;
                                if (!neg_b) { S = cScen; num = numc; den = denc; }
                            } else {
                                S = cScen; num = numc; den = denc;
                            }
                        }
                        const double n2 = num * num;
                        bool pass;
                        if constexpr (PB) {
                            pass = n2 * Da > fma(S - Sa, Da, Na) * den;
                        } else {
                            pass = n2 * Da > Na * den;
                        }
                        unsigned long long amask = __ballot(pass) & cmask;
                        if (never_accept) amask = 0ull;            // diagnostic: never accept
                        if (amask == 0ull) break;
                        const int ga = __builtin_ctzll(amask);
                        const unsigned long long upto = (2ull << ga) - 1ull;          // steps 0 .. ga
                        // the steps behind the accepted one see ft + d_acc: h_k += Σ w d_acc d_k (read issued first)
                        const double gk = Gs[(size_t)ga * W + (in ? g : 0)];
                        SC = readlane_f64(SCt, ga); SIC = readlane_f64(SICt, ga); SCC = readlane_f64(SCCt, ga);
                        Na = readlane_f64(n2, ga); Da = readlane_f64(den, ga);
                        double Sg = FB ? cScen : cSII;
                        if constexpr (PB) { Sa = readlane_f64(S, ga); MCSAS_IN_VGPR(Sa); Sg = Sa; }
                        h += gk;
                        accm |= 1ull << ga;
                        cmask &= ~upto;
                        // this candidate, accepted, ends the attempt: !(chi² > criterion), on its own (num², den)
                        // (the ballot tells the compiler that the test — the same in every lane — is wave-uniform: masks stay scalar)
                        if (__ballot(!((Sg - cCrit * cnq) * Da > Na)) != 0ull) { live = false; leftover = cmask; cmask = 0ull; }
                    }
                    {
                        const unsigned long long consumed = inmask & ~leftover;       // the steps this sub-window went through
                        num_iter += __builtin_popcountll(consumed);
                        num_moves += __builtin_popcountll(accm);
                        if (ovm_all & consumed) overflow = 1;
                    }
                    // the accepted steps of the sub-window, in order: sacc[1 + i] = step in the sub-window, lacc[...] = step in the window
                    nacc_sub = __builtin_popcountll(accm);
                    if (nacc_sub) {
                        X = (PB ? Sa : (FB ? cScen : cSII)) - Na / Da;   // chi²·Q of the state the sub-window ends in
                        MCSAS_IN_VGPR(X);
                        touched = true;
                    }
                    if ((accm >> lane) & 1ull) {
                        const int pos = __builtin_popcountll(accm & ((1ull << lane) - 1ull));
                        const int acc_row = k0 + lane;
                        sacc[1 + pos] = lane;
                        lacc[num_acc_win + pos] = acc_row;
                        if (swap_slots) {                          // slot swap: rows are never copied (lazy rows never move: nothing to swap)
                            const int fresh = lstage[acc_row], freed = lslot[acc_row];
                            lslot[acc_row] = fresh; lstage[acc_row] = freed;
                        }
                    }
                    num_acc_win += nacc_sub;
                };
                if (!find_bg) decide(std::false_type{}, std::false_type{});
                else if (pos_bg) decide(std::true_type{}, std::true_type{});
                else decide(std::true_type{}, std::false_type{});
                cur.chi2 = X / cnq;                               // scale and background are only needed at the end of the attempt
                if (lane == 0) { sacc[0] = nacc_sub; ctl[2] = live ? 1 : 0; }
                __builtin_amdgcn_s_setprio(0);
            }
            MCSAS_STAMP(s3);
            PIPE_LDS_BARRIER();                                            // B2: decisions published
            MCSAS_STAMP(s4);
            live = ctl[2] != 0;
            const int nacc = sacc[0];
            if (nacc > 0) {
                // ---- ft += d for the accepted steps, in order (mcsas.py:381), straight from the LDS row buffer: no
                // memory round trip on the way to the next sub-window.  (d = new - old is the producers' fp64
                // difference; the wavefront kernel's (ft - old) + new differs from ft + d in the last bit at most.)
#pragma unroll
                for (int x = 0; x < QT; ++x) {
                    const int i = tid + T * x;
                    if (i < qpad) {
                        double f = lft[i];
                        int n = 0;
                        // (four rows at a time: their LDS reads are independent, the additions keep their order — early in a chain a
                        // sub-window has a dozen accepted rows and a dependent read per row was a third of the tick's apply phase)
                        for (; n + 4 <= nacc; n += 4) {
                            const int r0 = sacc[1 + n], r1 = sacc[2 + n], r2 = sacc[3 + n], r3 = sacc[4 + n];
                            const double v0 = rowbuf[(size_t)r0 * qpad + i], v1 = rowbuf[(size_t)r1 * qpad + i];
                            const double v2 = rowbuf[(size_t)r2 * qpad + i], v3 = rowbuf[(size_t)r3 * qpad + i];
                            f += v0; f += v1; f += v2; f += v3;
                        }
                        for (; n < nacc; ++n) f += rowbuf[(size_t)sacc[1 + n] * qpad + i];
                        lft[i] = f; lwft[i] = wq[x] * f;
                    }
                }
                PIPE_LDS_BARRIER();                                        // B3: ft complete; sacc and the row buffer may be rewritten
                load_row_pairs_lds<QPL>(lwft, lane, wftp);
            }
            if (resum && wave == 0 && live && num_iter % resum == 0) {
                // (rows with an integral) every `resum` steps of the attempt — wherever that falls in a window — the running
                // sums are re-derived from ft so that the incremental updates cannot drift; ft is complete here (B3, or
                // nothing was accepted in this sub-window)
                double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    const double f = lft[lane + WAVE * j], wf = lwft[lane + WAVE * j];
                    s1 += wf; s2 = fma(wf, f, s2); s3 = fma(gwI_[lane + WAVE * j], f, s3);
                }
                wave_sum3(s1, s2, s3);
                SC = s1; SCC = s2; SIC = s3;
                cur = solve_fit(a, SC, SCC, SIC);
                X = cur.chi2 * nqd;
            }
#ifdef MCSAS_STAMPS
            MCSAS_STAMP(s5);
            ph[0] += s1 - s0; ph[1] += s2 - s1; ph[2] += s3 - s2; ph[3] += s4 - s3; ph[4] += s5 - s4; ph[5] += 1; ph[6] += nacc;
#endif
        }
        MCSAS_STAMP(sb2);
#ifdef MCSAS_STAMPS
        if (wave == 0 && lane == 0) for (int i = 0; i < 8; ++i) ch.dbg[i] += ph[i];
#endif
        if (wave == 0 && lane == 0) lacc[Kb] = num_acc_win;
        PIPE_LDS_BARRIER();
        {   // write the window's slot tables back and store the accepted proposals (mcsas.py:381), all waves
            const int nacc = lacc[Kb];
            if (nacc > 0) {
                if (!pa.g.lazy_rows) {
                    for (int i = tid; i < kmax_all; i += T) {
                        stage[i] = lstage[i];
                        int r = ri0 + i; if (r >= N) r -= N;
                        slot_of[r] = lslot[i];
                    }
                } else {
                    // the accepted contributions' cached rows are stale from here on (the producer that proposes for them
                    // next, N steps from now, evaluates them again from the parameters stored just below)
                    auto row_valid = glb(pa.row_valid) + (size_t)rep * N;
                    for (int i = tid; i < nacc; i += T) {
                        int r = ri0 + lacc[i]; if (r >= N) r -= N;
                        row_valid[r] = 0;
                    }
                }
                for (int i = tid; i < nacc * P; i += T) {
                    const int kk = lacc[i / P], p = i % P;
                    int r = ri0 + kk; if (r >= N) r -= N;
                    rset[(size_t)r * P + p] = pval[(size_t)kk * MCSAS_MAX_ACTIVE + p];
                }
                // park ft in HBM for the next tick
                for (int i = tid; i < qpad; i += T) { gft[i] = lft[i]; gwft[i] = lwft[i]; }
            }
        }
        if (wave == 0) {
            Xwin = X; had_window = true;
            if (touched && !resum) {
                // re-sum the fit sums from ft so the incremental updates cannot drift
                double s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll
                for (int j = 0; j < QPL; ++j) {
                    const double f = lft[lane + WAVE * j], wf = lwft[lane + WAVE * j];
                    s1 += wf; s2 = fma(wf, f, s2); s3 = fma(gwI_[lane + WAVE * j], f, s3);
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
                ft[j] = lft[lane + WAVE * j];
                const double wf = gw_[lane + WAVE * j] * ft[j];
                s1 += wf; s2 = fma(wf, ft[j], s2); s3 = fma(gwI_[lane + WAVE * j], ft[j], s3);
            }
            wave_sum3(s1, s2, s3);
            cur = solve_fit(a, s1, s2, s3);
            double rs = 0.;
#pragma unroll
            for (int j = 0; j < QPL; ++j) {
                const int i = lane + WAVE * j;
                const double r = gI_[i] - (ft[j] * cur.A + cur.b);
                rs += gw_[i] * r * r;
            }
            cur.chi2 = wave_sum(rs) / nqd;                    // chiSqr, backgroundscalingfit.py:72-77
            converged = !(cur.chi2 > a.conv_crit);
            total_steps += num_iter;
            draw_pos = sn.step_base + (uint64_t)num_iter * P;
            if (converged || stopped || sn.attempt >= a.max_retries) {
                done = 1;
#pragma unroll
                for (int j = 0; j < QPL; ++j)
                    glb(a.fit)[(size_t)rep * qpad + lane + WAVE * j] = ft[j] * cur.A + cur.b;
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
            store_snap(&pa.chains[rep].snap[t & 1], next);    // read by PROD(t+2) and SCAN(t+1)
            // a finished chain: the OTHER record's `alive` goes to 0 as well, so that the producers of the odd ticks stop
            // evaluating rows for it too (one word; a producer of this very launch that still reads 1 only does work nobody uses)
            if (done) glb(&pa.chains[rep].snap[(t + 1) & 1].alive)[0] = 0;
            ch.SC = SC; ch.SIC = SIC; ch.SCC = SCC; ch.A = cur.A; ch.b = cur.b; ch.chi2 = cur.chi2;
            ch.X = had_window ? Xwin : cur.chi2 * nqd;            // (no window this tick: the attempt's initial fit)
            ch.num_iter = num_iter; ch.num_moves = num_moves; ch.total_steps = total_steps;
            ch.draw_pos = draw_pos; ch.attempts = attempts; ch.converged = converged; ch.stopped = stopped;
            if (overflow) atomicOr(&pa.chains[rep].overflow, 1);
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
                if (__hip_atomic_fetch_add(pa.n_done_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == a.n_reps)
                    __hip_atomic_store(pa.n_done, a.n_reps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ one tick
// launch t: blocks [0, R) do SCAN(t) (skipped for t < 0), the others PROD(t + 1)
template <int M, int QPL, bool RQ>
__global__ __launch_bounds__(PIPE_BLOCK) void pipe_tick_kernel(const PipeArgs *pap, const int tick, const int stop_now, const PipeHot hot) {
    // The argument block lives in device memory and is read where it is needed: passed by value it
    // would sit in SGPRs for the whole kernel (600+ bytes) and the scan loop would run on spilled
    // scalars (one v_readlane per use).
    extern __shared__ double lds[];
    const PipeArgs &pa = *pap;
    const int R = hot.n_reps, b = blockIdx.x, t = tick;
#ifdef MCSAS_STAMPS
    struct Timeline {                                          // one record per wave, written when the wave leaves the kernel
        uint64_t *p, t0; uint32_t hw;
        __device__ Timeline(const PipeArgs &pa, int tick) {
            p = (pa.timeline && tick == pa.timeline_tick && (threadIdx.x & 63) == 0) ? pa.timeline + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * PIPE_TL_WORDS : nullptr;
            t0 = wall_clock64();
            hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        }
        __device__ ~Timeline() { if (p) { p[0] = t0; p[1] = wall_clock64(); p[2] = hw; p[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); } }   // XCC_ID
    } timeline_record(pa, tick);
#endif
    if (b < R) {
        if (t >= 0) {
            // rows per wave and sub-window: uniform for the launch; the host only picks combinations instantiated here
            if constexpr (RQ) pipe_scan_block<M, QPL, 1, true>(pa, lds, b, t, stop_now);       // (sub-windows of 8 steps)
            else switch (hot.w_sub >> 3) {
#define PIPE_SCAN_CASE(r) case r: if constexpr (r * QPL <= PIPE_MAX_ROW_DOUBLES) pipe_scan_block<M, QPL, r, false>(pa, lds, b, t, stop_now); break;
                PIPE_SCAN_CASE(1) PIPE_SCAN_CASE(2) PIPE_SCAN_CASE(3) PIPE_SCAN_CASE(4) PIPE_SCAN_CASE(6) PIPE_SCAN_CASE(8)
#undef PIPE_SCAN_CASE
                default: break;
            }
        }
    } else {
        // chain-major block -> (chain, sub-window) map: round-robin dispatch then spreads every chain's blocks
        // over the 8 XCDs.  The XCD-aware alternative (diagnostic bit 128: chain r's producer blocks on block ids
        // congruent to r modulo 8, i.e. on the XCD of its scan block, so that next tick's d rows sit in that
        // XCD's L2) measured 10-40 % SLOWER on config 2: 50 chains x (1 + gy) blocks do not divide over 8 XCDs of
        // 32 CUs with one block per CU — two XCDs get 35 blocks and their chains take two rounds.
        const int gy = hot.prod_blocks_y;
        int rep = (b - R) / gy, y = (b - R) % gy;
        // rows pulled from a queue: block-major over the chains (b = R + y R + rep), so that the workgroups that have to wait for a
        // CU — the launch has more of them than the chip — are the LAST block of every chain, not all blocks of the last chains
        if constexpr (RQ) { rep = (b - R) % R; y = (b - R) / R; }
        if (MCSAS_TUNE_BITS(pa.c) & 128) { const int x = b & 7, j = (b - R) >> 3; rep = x + 8 * (j / gy); y = j % gy; }
        if (rep < R) pipe_prod_block<M, QPL, RQ>(pa, hot, lds, rep, y, gy, t + 1);
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
    if (pa.rowq) { pa.rowq[(size_t)rep * 2] = 0; pa.rowq[(size_t)rep * 2 + 1] = 0; }
}

}  // namespace mcsas
