/* mcsas_hip.h — C ABI of the MI355X-native McSAS Monte-Carlo core (libmcsas_hip.so).
 *
 * This is the drop-in boundary.  The reference has no FFI; the path it replaces is
 * the pure-Python call chain
 *     McSAS.analyse()                         src/mcsas/mcsas/mcsas.py:191-285
 *       -> McSAS.mcFit()                      src/mcsas/mcsas/mcsas.py:287-439
 *            -> ScatteringModel.calc()        src/mcsas/bases/model/scatteringmodel.py:79-109
 *            -> BackgroundScalingFit.calc()   src/mcsas/mcsas/backgroundscalingfit.py:112-139
 *            -> ScatteringModel.generateParameters()  src/mcsas/bases/model/scatteringmodel.py:117-127
 * Every entry point below names the reference interface it stands in for.  Plain pointers and
 * sizes only; all arrays are caller-owned host memory, double = IEEE binary64, C order.
 * All functions return 0 on success or a negative MCSAS_E* code and never throw; the message of
 * the last failure on the calling thread is available from mcsas_hip_last_error().
 * One in-flight call per plan (mirrors the reference's non-reentrant model objects).
 */
#ifndef MCSAS_HIP_H
#define MCSAS_HIP_H

#ifndef __HIPCC_RTC__   /* (the library's run-time compiler has the fixed-width types built in) */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MCSAS_ABI_VERSION 4
#define MCSAS_MAX_ACTIVE 4   /* active (fitted) parameters per contribution: columns of rset */
#define MCSAS_MAX_PARAMS 8   /* full parameter vector of a model */
#define MCSAS_MAX_DEVICES 16 /* devices one mcsas_hip_analyse call may spread its repetitions over */

/* model_id: which ScatteringModel.formfactor/volume/absVolume/surface set is evaluated.
 * Parameter vector order = the reference's `parameters` tuple of that class. */
enum {
    MCSAS_MODEL_SPHERE = 0,      /* models/sphere.py:12-63            (radius, sld) */
    MCSAS_MODEL_CYL_ISO = 1,     /* models/cylindersisotropic.py:17-101 (radius, useAspect, length, aspect, intDiv, sld) */
    MCSAS_MODEL_ELL_CS = 2,      /* models/ellipsoidalcoreshell.py:14-97 (a, b, t, eta_c, eta_s, eta_sol, intDiv) */
    MCSAS_MODEL_KHOLODENKO = 3,  /* models/kholodenko.py:51-94         (radius, lenKuhn, lenContour) */
    MCSAS_MODEL_ELL_ISO = 4,     /* models/ellipsoidsisotropic.py:18-84 (a, useAspect, c, aspect, intDiv, sld) */
    MCSAS_MODEL_SPH_CS = 5,      /* models/sphericalcoreshell.py:14-77 (radius, t, eta_c, eta_s, eta_sol) */
    MCSAS_MODEL_GAUSS_CHAIN = 6, /* models/gaussianchain.py:14-66      (rg, bp, etas, k) */
    MCSAS_MODEL_LMA_SPHERE = 7,  /* models/lmadensesphere.py:14-106    (radius, volFrac, mf, sld) */
    MCSAS_MODEL_COUNT = 8,
    MCSAS_MODEL_PLUGIN = 8,      /* the template slot run-time plug-ins are compiled into (csrc/plugin_model.h); never a caller's model_id */
    MCSAS_MODEL_PLUGIN0 = 64     /* first id mcsas_hip_plugin_compile hands out */
};
#define MCSAS_MAX_PLUGINS 32 /* plug-in models a process may register */

/* gen_kind: NumberGenerator subclass of an active parameter (bases/algorithm/numbergenerator.py) */
enum {
    MCSAS_GEN_UNIFORM = 0,  /* RandomUniform       :28-31   u */
    MCSAS_GEN_EXP1 = 1,     /* RandomExponential   :168-175 (10^u - 1)/10 */
    MCSAS_GEN_EXP2 = 2,     /* RandomExponential2  :181-184 (10^(2u) - 1)/100 */
    MCSAS_GEN_EXP3 = 3      /* RandomExponential3  :186-189 (10^(3u) - 1)/1000 */
};

/* exec_mode: how chains are mapped onto the chip.  Every mode replays the reference's trajectories decision for decision
 * (tests/test_parity_gpu.py); the modes sum the same terms in different orders, so two FREE-RUNNING chains with the same
 * seed can take different turns at a numerically tied step (replacing one negligible contribution by another moves chi²
 * by less than its rounding error).  A repetition's result is reproducible for the same mode, whatever the number of
 * repetitions or devices beside it: nothing a chain decides depends on how many chains share a launch (the pipeline's window
 * is fixed for rows without an integral; for rows with an integral it follows the chain count, and the decisions are kept
 * independent of it: csrc/chain_pipe.h, PipeGeom::resum_every).
 * q-points (the reference takes any data.count, mcsas.py:210): up to 1024 in every mode; 1025..16384 (un-binned data files,
 * nBin = 0) one workgroup per chain with the q-points split over its waves (csrc/chain_wide.h: MCSAS_EXEC_WORKGROUP, and
 * what MCSAS_EXEC_AUTO picks); up to 4096 MCSAS_EXEC_WAVE also runs; beyond 16384 MCSAS_EINVAL. */
enum {
    MCSAS_EXEC_AUTO = 0,
    MCSAS_EXEC_WAVE = 1,      /* one wavefront per chain: many repetitions (>= ~1000) */
    MCSAS_EXEC_WORKGROUP = 2, /* one workgroup per chain: speculative proposal window in LDS (nq <= 1024), q-split (nq > 1024) */
    MCSAS_EXEC_PIPELINE = 3   /* producer kernels on every CU + one scan workgroup per chain: few repetitions (nq <= 1024) */
};

enum {
    MCSAS_OK = 0,
    MCSAS_EINVAL = -1,      /* bad argument / unsupported size */
    MCSAS_ENODEV = -2,      /* no usable HIP device */
    MCSAS_EHIP = -3,        /* HIP runtime error (see mcsas_hip_last_error) */
    MCSAS_ENOMEM = -4,
    MCSAS_ESTREAM = -5,     /* replay stream shorter than the draws a chain consumed */
    MCSAS_ECALLBACK = -6    /* the caller's rows callback reported a failure (mcsas_hip_analyse_host_rows) */
};

/* Everything McSAS.analyse() reads from self.data, self.model and the algorithm settings. */
typedef struct mcsas_problem {
    uint32_t struct_size;        /* sizeof(mcsas_problem), ABI check */
    int32_t  model_id;           /* MCSAS_MODEL_* */

    /* data: SASData.q, data.f.binnedData, data.f.binnedDataU (dataobj/sasdata.py:51-55,
     * backgroundscalingfit.py:113-117; sigma == 0 is treated as 1 like the reference). SI units. */
    int32_t  nq;
    const double *q;
    const double *intensity;
    const double *sigma;

    /* model: values of ALL parameters (inactive ones are used as they are) + the active set */
    double   params[MCSAS_MAX_PARAMS];
    int32_t  n_active;                           /* model.activeParamCount() */
    int32_t  active_index[MCSAS_MAX_ACTIVE];     /* ascending indices into params[] (activeParams() order) */
    double   gen_lo[MCSAS_MAX_ACTIVE];           /* activeRange ∩ valueRange (utils/parameter.py:715-728, */
    double   gen_hi[MCSAS_MAX_ACTIVE];           /*   bases/algorithm/parameter.py:66-84) */
    int32_t  gen_kind[MCSAS_MAX_ACTIVE];         /* MCSAS_GEN_* */
    double   clip_lo[MCSAS_MAX_ACTIVE];          /* valueRange: Parameter.setValue clips into it */
    double   clip_hi[MCSAS_MAX_ACTIVE];          /*   (bases/algorithm/parameter.py:405-414,489-495) */
    double   start_value[MCSAS_MAX_ACTIVE];      /* startFromMinimum fill value (mcsas.py:310-315) */

    /* algorithm settings (mcsas/mcsasparameters.json) */
    int32_t  n_contrib;          /* numContribs */
    int32_t  n_reps;             /* numReps handled by THIS call (a shard when multi-GPU) */
    int64_t  max_iter;           /* maxIterations */
    double   comp_exp;           /* compensationExponent */
    double   conv_crit;          /* convergenceCriterion */
    int32_t  max_retries;        /* maxRetries: up to max_retries+1 attempts per rep (mcsas.py:220-246) */
    int32_t  find_background;    /* findBackground */
    int32_t  positive_background;
    int32_t  start_from_minimum;

    /* random numbers.  Free-running: Philox4x32-10 keyed by `seed`, one independent stream per
     * chain id (rep_offset + local rep).  Replay: replay_stream != NULL supplies, per rep, the
     * raw uniforms numpy.random.uniform would have returned, consumed in the reference's order
     * (N draws per active parameter at chain start, then n_active per step; retries continue). */
    uint64_t seed;
    int32_t  rep_offset;         /* global index of this shard's first rep */
    int32_t  reserved0;          /* must be 0: MCSAS_EINVAL otherwise */
    const double *replay_stream; /* [n_reps][replay_len] or NULL */
    int64_t  replay_len;

    /* cooperative stop (McSAS.stop, mcsas.py:357): polled by the host while the kernel runs and
     * forwarded to the device; may be NULL */
    const volatile int32_t *stop;

    /* execution */
    int32_t  device;             /* HIP device ordinal, -1 = current device */
    int32_t  waves_per_chain;    /* 0 = auto; 1 = one wavefront per chain; >1 = workgroup per chain */
    int32_t  cache_intensities;  /* -1 auto, 0 re-evaluate `old` every step like mcsas.py:362, 1 keep rows in HBM */
    int32_t  exec_mode;          /* MCSAS_EXEC_*: 0 auto, 1 wavefront per chain, 2 workgroup per chain, 3 whole-chip pipeline */

    /* beam-profile smearing (ABI 2): what SASConfig.prepareSmearing left in data.locs and
     * data.config.smearing.prepared (dataobj/sasconfig.py:308-339, dataobj/sasdata.py:165).  With
     * smear_nk > 0 and a model whose class has canSmear = True (Sphere, LMADenseSphere) every
     * calcIntensity is 2 * trapz(F(locs)^2 * w * weights, x = q_offset) (bases/model/sasmodel.py:56-73);
     * other models ignore it, as in the reference.  smear_nk = 0: off. */
    int32_t  smear_nk;               /* integration points per q (nSteps + 1, or 2 ceil(nSteps/2) + 1) */
    int32_t  reserved1;
    const double *smear_locs;        /* [nq][smear_nk], row-major: where F is evaluated */
    const double *smear_q_offset;    /* [smear_nk] */
    const double *smear_weights;     /* [smear_nk] beam-profile weights */

    /* several GPUs (ABI 3).  The repetition loop of McSAS.analyse (mcsas.py:214-262) is a serial `for nr in
     * range(numReps)` over chains that share nothing but read-only data; with n_devices > 1 mcsas_hip_analyse runs
     * contiguous blocks of repetitions on devices[0..n_devices) at the same time (one host thread, one plan and one
     * stream per device; blocks differ by at most one repetition) and writes every block into the caller's arrays
     * at its place.  The chain id of a repetition is rep_offset + its index whatever the split, so a repetition's
     * random stream does not depend on the device count — and neither does its result as long as every block
     * runs in the same execution mode (set exec_mode; MCSAS_EXEC_AUTO looks at the block's own repetition count).
     * n_devices = 0 or 1: `device` alone.  A device may be listed more than once.  Plans take `device` only. */
    int32_t  n_devices;
    int32_t  devices[MCSAS_MAX_DEVICES];
    int32_t  reserved2;
} mcsas_problem;

/* What mcFit returns per repetition (mcsas.py:428-439) gathered the way analyse() stores it
 * (mcsas.py:203-210,233-251).  Caller allocates every non-NULL array. */
typedef struct mcsas_result {
    uint32_t struct_size;
    uint32_t reserved;
    double  *contribs;     /* [n_contrib][n_active][n_reps]  == contributions, mcsas.py:203-205 */
    double  *fit;          /* [nq][n_reps]                   == contribMeasVal[0], mcsas.py:210 */
    double  *chisq;        /* [n_reps] final reduced chi-squared (conval) */
    double  *scaling;      /* [n_reps] details['scaling'] */
    double  *background;   /* [n_reps] details['background'] */
    int64_t *num_iter;     /* [n_reps] details['numIterations'] of the last attempt */
    int64_t *num_moves;    /* [n_reps] details['numMoves'] */
    int32_t *attempts;     /* [n_reps] mcFit calls made for this rep */
    int32_t *converged;    /* [n_reps] chisq <= conv_crit */
    double  *seconds;      /* [n_reps] device wall time of the chain, all attempts */
    int64_t *draws;        /* [n_reps] uniforms consumed (replay bookkeeping); may be NULL */
} mcsas_result;

/* ---- one-shot: replaces McSAS.analyse()'s repetition loop (mcsas.py:214-262) ----------------
 * n_active == 0 (no active fit parameter, mcsas.py:198-201, 238-239, 322-323): the reference runs ONE
 * repetition of ONE contribution and mcFit returns the model intensity at the fixed parameter values;
 * here: fit[nq] = that intensity, chisq[0] = -1, scaling[0] = 1, background[0] = 0, num_iter[0] = 0 — the
 * result arrays are then used as if n_contrib = n_reps = 1. */
int mcsas_hip_analyse(const mcsas_problem *problem, mcsas_result *result);
/* how mcsas_hip_analyse splits n_reps repetitions over n_devices devices: block `index` = repetitions
 * [*first, *first + *count) (contiguous, in device-list order, sizes differ by at most one; count may be 0) */
int mcsas_hip_shard(int32_t n_reps, int32_t n_devices, int32_t index, int32_t *first, int32_t *count);

/* ---- McSAS.analyse() for a model that exists only as HOST code (ABI 4) -------------------------------------------------
 * The reference's plug-in contract is a ScatteringModel subclass with Python formfactor() / volume() / absVolume() / surface()
 * (bases/model/scatteringmodel.py:16-58), evaluated one parameter set at a time by ScatteringModel.calc (:79-105: p.setValue(v)
 * on every active parameter — which clips into the valueRange —, then calcIntensity, bases/model/sasmodel.py:46-79), found by
 * FindModels in any file under models/ (utils/findmodels.py:120-186).  Such a model has no device code.  A proposal depends on the random
 * stream only, never on the state of its chain (mcsas.py:358), so the library draws the proposals of the next `window` steps of
 * every chain on the host (the same counter-based / replayed streams and generator transforms as every other entry point), hands
 * ALL of them to `rows_cb` in one call, and runs everything else of McSAS.mcFit on the device (csrc/chain_feed.h): the row
 * cache (`old`, mcsas.py:362), test = ft - old + new, the scale / background fit and chi² (:376), the decisions (:379-390), the
 * final fit (:424-430); the retry loop of analyse() (:220-246) and McSAS.stop are followed between windows.
 *   rows_cb(user, n, pset, rows): pset[n][n_active] parameter sets (activeParams() order, NOT yet clipped: setValue does that)
 *     -> rows[n][nq] = calcIntensity(data, compensationExponent)[0] of each; return 0, anything else aborts with MCSAS_ECALLBACK.
 *     Called from the calling thread only, with 1 <= n <= max(n_contrib, window) * n_reps rows (bounded by ~256 MB of rows).
 *   window: proposals evaluated ahead per chain and call (< 1: 64); a chain that ends inside a window wastes the rest of it.
 * Uses problem->{nq, q, intensity, sigma, n_active, gen_*, start_value, n_contrib, n_reps, max_iter, conv_crit, max_retries,
 * find_background, positive_background, start_from_minimum, seed, rep_offset, replay_*, stop, device}; model_id, params, clip_*,
 * comp_exp and smear_* are the callback's business (calcIntensity smears by itself, sasmodel.py:56-73).  nq <= 16384.
 * result: as mcsas_hip_analyse; seconds[] = host wall time from the start of the call to the end of the repetition. */
typedef int (*mcsas_rows_callback)(void *user, int32_t n, const double *pset, double *rows);
int mcsas_hip_analyse_host_rows(const mcsas_problem *problem, mcsas_rows_callback rows_cb, void *user, int32_t window,
                                mcsas_result *result);

/* ---- resident plan: same work split so that inputs/workspaces live in HBM across runs ------- */
typedef struct mcsas_plan mcsas_plan;
int  mcsas_hip_plan_create(const mcsas_problem *problem, mcsas_plan **plan);
/* enqueue all chains on `hip_stream` (a hipStream_t, NULL = default stream).  Wavefront / workgroup modes:
 * one asynchronous kernel launch.  Pipeline mode: one launch per window of steps; the call stays ahead of the
 * GPU by at most 64 launches and otherwise waits on an event (every 16 launches) while it forwards
 * problem->stop and watches for the last chain to finish — it returns when everything is ENQUEUED, which for
 * long runs is shortly before everything has run. */
int  mcsas_hip_plan_launch(mcsas_plan *plan, void *hip_stream);
/* wait for the launch (forwarding problem->stop meanwhile) and copy results out */
int  mcsas_hip_plan_fetch(mcsas_plan *plan, mcsas_result *result);
/* Result slots.  A plan keeps MCSAS_PLAN_SLOTS sets of what a finished analysis is read back from (parameter sets, fits,
 * per-chain outputs, timing events) over ONE set of workspaces, so that a series of analyses — one data set after the other,
 * gui/calc.py:271-330 — keeps the device busy: launch into slot 1 while slot 0 is being fetched and unpacked.  The slots of a
 * plan share its workspaces: launch them on the same stream (they then run one after the other).  mcsas_hip_plan_launch /
 * _fetch are slot 0; _last_ms and _total_steps report the slot fetched last. */
#define MCSAS_PLAN_SLOTS 2
int  mcsas_hip_plan_launch_slot(mcsas_plan *plan, void *hip_stream, int32_t slot);
int  mcsas_hip_plan_fetch_slot(mcsas_plan *plan, int32_t slot, mcsas_result *result);
/* device time of the last launch measured with HIP events on its stream, milliseconds */
int  mcsas_hip_plan_last_ms(mcsas_plan *plan, double *ms);
/* total MC steps executed by the last launch (sum of iterations over chains and attempts) */
int  mcsas_hip_plan_total_steps(mcsas_plan *plan, int64_t *steps);
/* how the plan executes: info[0] = exec mode chosen (MCSAS_EXEC_*), [1] = waves per chain, [2] = q slots
 * per lane, [3] = speculative window (steps), [4] = kernel launches of the last mcsas_hip_plan_launch,
 * [5] = 1 if per-contribution intensity rows are cached in HBM, [6..7] reserved */
int  mcsas_hip_plan_info(mcsas_plan *plan, int32_t info[8]);
/* change seed / rep_offset between launches without re-uploading anything else */
int  mcsas_hip_plan_reseed(mcsas_plan *plan, uint64_t seed, int32_t rep_offset);
void mcsas_hip_plan_destroy(mcsas_plan *plan);

/* ---- ScatteringModel.calc(data, pset, compensationExponent) (scatteringmodel.py:79-109) ------
 * Uses problem->{model_id, params, n_active, active_index, clip_*, nq, q, comp_exp, device}.
 * pset[n][n_active] -> cum_int[nq] (rows summed in order), vset/wset/sset[n], optional
 * rows[n][nq] (each contribution's F²·w, i.e. SASModel.calcIntensity()[0], sasmodel.py:46-79). */
int mcsas_hip_model_calc(const mcsas_problem *problem, const double *pset, int32_t n,
                         double *cum_int, double *vset, double *wset, double *sset, double *rows);

/* ---- BackgroundScalingFit.calc (backgroundscalingfit.py:112-139), closed-form minimiser ------
 * out[4] = { sc[0] (scaling), sc[1] (background), conval (reduced chi²), aGoFs } */
int mcsas_hip_bgfit(int32_t nq, const double *intensity, const double *sigma, const double *model_int,
                    int32_t find_background, int32_t positive_background, int32_t num_params,
                    int32_t device, double out[4]);

/* ---- McSAS.histogram()'s per-contribution visibility limits (mcsas.py:575-594) --------------
 * For each rep: min over q of sigma·vf_c / (A·I_c(q)) where I_c != 0.  contribs in the
 * (n_contrib, n_active, n_reps) layout of mcsas_result, vol_frac/min_req_vol [n_contrib][n_reps]. */
int mcsas_hip_observability(const mcsas_problem *problem, const double *contribs,
                            const double *scaling, const double *vol_frac, double *min_req_vol);

/* ---- McSAS.histogram(), first half, for ALL repetitions in one call (mcsas.py:549-594) ----------
 * Per rep: model.calc over its contributions (:552), scale/background fit of the summed intensity to
 * the data (:559), per-contribution visibility limits min_q sigma*vf_c / (A*I_c(q)) (:575-590) with
 * vf_c = wset_c * A / vset_c (modeldata.py:57-61).  Uses problem->{model, nq, q, intensity, sigma,
 * n_contrib, n_reps, comp_exp, find_background, positive_background, smear_*, device}; contribs in the
 * (n_contrib, n_active, n_reps) layout of mcsas_result.  scaling[2][n_reps] = (A, b) per rep;
 * vset/wset/sset/min_req_vol [n_contrib][n_reps]. */
int mcsas_hip_histogram_prep(const mcsas_problem *problem, const double *contribs, double *scaling,
                             double *vset, double *wset, double *sset, double *min_req_vol);

/* ---- McSAS.histogram(), ALL of it for all repetitions in one call (mcsas.py:445-615) --------------
 * What mcsas_hip_histogram_prep does, then on the device as well: the volume / number / intensity / surface fractions and
 * their visibility limits with the per-repetition normalisation (mcsas.py:561-604), and for every configured histogram
 * (utils/parameter.py:187-538: one per (parameter, range, weighting)) the bins, the mean visibility limit of a bin's
 * members, the cumulative distribution (Histogram._calcBins / _calcCDF :441-479) and the moments per repetition
 * (Moments :84-122) — every sum taken over the contributions in contribution order, the order of the reference's
 * builtin sum().  The means / standard deviations over the repetitions (VectorResult :156-184) are left to the caller.
 *   specs[h]: param_index = column of contribs; weighting 0 vol, 1 num, 2 int, 3 surf; edges = n_bin + 1 lower bin edges
 *             (Histogram._setXLowerEdge :349-362, evaluated by the caller so that bin membership is numpy's); lower / upper =
 *             the histogram's value range (Moments count lower < x < upper).
 *   scaling[2][n_reps]; fractions[8][n_contrib][n_reps] = vol, num, int, surf fractions, then their visibility limits
 *   (may be NULL); out = per histogram, one after the other: bins[n_bin][n_reps], obs[n_bin][n_reps] (0 for an empty bin),
 *   cdf[n_bin][n_reps], moments[5][n_reps] (total, mean, variance, skew, kurtosis).
 * n_contrib <= 4096 (the contributions of a repetition are staged in LDS): MCSAS_EINVAL beyond, use mcsas_hip_histogram_prep. */
typedef struct mcsas_histogram_spec {
    int32_t param_index, weighting, n_bin, reserved;
    double  lower, upper;
    const double *edges;
} mcsas_histogram_spec;
int mcsas_hip_histogram(const mcsas_problem *problem, const double *contribs, int32_t n_hist,
                        const mcsas_histogram_spec *specs, double *scaling, double *fractions, double *out);

/* ---- input preparation ----------------------------------------------------------
 * DataObj._prepareUncertainty (dataobj/dataobj.py:204-227): sigma_out = max(sigma_raw, fu_min * I),
 * fu_min * I when sigma_raw is NULL (no uncertainty column), +inf where that is not finite. */
int mcsas_hip_prepare_uncertainty(int32_t n, const double *intensity, const double *sigma_raw, double fu_min,
                                  int32_t device, double *sigma_out);
/* DataObj._reBin (dataobj/dataobj.py:288-345) on the sanitized vectors x, f, fu [n]: bin b takes the
 * points with edges[b] <= x < edges[b+1] (edges[n_bin+1]: the caller's numpy.logspace, :312-316); one
 * point: copied; more: mean x, mean f, max(std(f, ddof=1)/sqrt(count), sqrt(sum fu^2 / count)); empty
 * bins are dropped.  Outputs hold up to n_bin entries, *n_out = bins kept. */
int mcsas_hip_rebin(int32_t n, const double *x, const double *f, const double *fu, int32_t n_bin,
                    const double *edges, int32_t device, double *x_out, double *f_out, double *fu_out,
                    int32_t *n_out);

/* Run-time model plug-in — what the reference's model discovery does for any file under models/ with a ScatteringModel subclass
 * (utils/findmodels.py:120-186; the four methods a model supplies: bases/model/scatteringmodel.py:15-58, sasmodel.py:36-79).
 * `source` is HIP C++ text that defines, per q point and on the model's full parameter vector p[MCSAS_MAX_PARAMS] (active
 * parameters substituted and clipped into their valueRange):
 *     __device__ double mcsas_plugin_formfactor(double q, const double *p);   // ScatteringModel.formfactor
 *     __device__ double mcsas_plugin_volume(const double *p);                  // .volume()
 *     __device__ double mcsas_plugin_absvolume(const double *p);               // .absVolume()
 *     __device__ double mcsas_plugin_surface(const double *p);                 // .surface()
 * (csrc/fastmath.h and device_util.h are in scope, namespace mcsas.)  The text is compiled with hiprtc for gfx950 against
 * the library's own kernel headers, which it carries inside; no GPU is needed for that.  *model_id (>= MCSAS_MODEL_PLUGIN0)
 * is then valid as mcsas_problem.model_id in every entry point and every execution mode (the chain kernel of the mode is
 * compiled for the plug-in on first use: 0.5 - 15 s once per q-slot count; more than 1024 q-points: the q-split workgroup
 * kernel, MCSAS_EXEC_AUTO or MCSAS_EXEC_WORKGROUP); beam-profile smearing if the text
 * says `#define MCSAS_PLUGIN_CAN_SMEAR 1` (the model class's canSmear).  A form
 * factor that loops over orientations / a contour should say `#define MCSAS_PLUGIN_ROW_CLASS 1` (csrc/plugin_model.h: how
 * the pipeline spreads such rows over the chip; results do not depend on it).  The same text registered twice
 * gives the same id.  MCSAS_EINVAL + mcsas_hip_plugin_log() (compiler output, calling thread) if it does not compile. */
int         mcsas_hip_plugin_compile(const char *source, int32_t *model_id);
const char *mcsas_hip_plugin_log(void);

/* A (non-blocking) HIP stream on `device` for mcsas_hip_plan_launch, for hosts that have no HIP binding of their own: plans on
 * different streams overlap on the chip — two analyses side by side finish sooner than one after the other. */
int         mcsas_hip_stream_create(int32_t device, void **stream);
void        mcsas_hip_stream_destroy(void *stream);
/* Device and pinned memory that destroyed plans gave back stays parked for the next plan of the same shape (a series of
 * analyses pays ~3 ms per plan in hipFree / hipHostFree otherwise), up to 4 GiB per process; this returns it to the driver. */
int         mcsas_hip_release_cached_memory(void);
int         mcsas_hip_device_count(void);
int         mcsas_hip_abi_version(void);
/* 0: the release library.  1: a measurement build (-DMCSAS_TUNING) in which mcsas_problem.reserved0 selects tuning and
 * ablation variants of the pipeline mode (csrc/mcsas_hip.hip, mcsas_hip_plan_create); never shipped as libmcsas_hip.so */
int         mcsas_hip_is_tuning_build(void);
const char *mcsas_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MCSAS_HIP_H */
