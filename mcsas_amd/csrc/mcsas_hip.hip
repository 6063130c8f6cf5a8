// mcsas_hip.hip — libmcsas_hip.so: C ABI (include/mcsas_hip.h) over the gfx950 chain kernels.
// Built only for MI355X (gfx950); there is no CPU path in here.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcsas_hip.h"
#include "chain_common.h"
#include "small_kernels.h"   // model_rows_kernel, observability_kernel, hist_rows_kernel
#include "chain_feed.h"     // feed_rows_kernel: chains whose rows the caller evaluates (mcsas_hip_analyse_host_rows)
#include "chain_wg.h"   // WgGeom / wg_geometry only; the kernels are instantiated in kern_*.hip
#include "chain_pipe.h" // PipeArgs / pipe_geometry only
#include "chain_wide.h" // WIDE_* constants only
#include "auto_table.h" // measured rates of the execution modes (tools/make_auto_table.py)
#include "model_list.h" // MCSAS_FOR_MODELS: the built-in models

using namespace mcsas;

// ------------------------------------------------------------------------------ error plumbing
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? MCSAS_ENOMEM : MCSAS_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

// ------------------------------------------------------------------------------ small kernels
// cumInt += it, contribution by contribution (scatteringmodel.py:101): thread per q, fixed order
__global__ void rows_cumsum_kernel(int nq, int n, const double *rows, double *cum) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nq) return;
    double s = 0.;
    for (int i = 0; i < n; ++i) s += rows[(size_t)i * nq + k];
    cum[k] = s;
}

// BackgroundScalingFit.calc with the closed-form minimiser; one wave
__global__ __launch_bounds__(64) void bgfit_kernel(int nq, const double *I, const double *sigma, const double *C,
                                                   int find_bg, int pos_bg, int num_params, double *out) {
    const int lane = threadIdx.x;
    double sw = 0, si = 0, sii = 0, sc = 0, scc = 0, sic = 0, ss2 = 0;
    for (int k = lane; k < nq; k += WAVE) {
        double e = sigma[k] == 0.0 ? 1.0 : sigma[k];       // backgroundscalingfit.py:117
        double w = 1.0 / (e * e);
        sw += w; si += w * I[k]; sii += w * I[k] * I[k];
        sc += w * C[k]; scc += w * C[k] * C[k]; sic += w * I[k] * C[k]; ss2 += e * e;
    }
    wave_sum3(sw, si, sii); wave_sum3(sc, scc, sic); ss2 = wave_sum(ss2);
    ChainArgs a{};
    a.Sw = sw; a.SI = si; a.SII = sii; a.nq = nq; a.find_bg = find_bg; a.pos_bg = pos_bg;
    FitResult f = solve_fit(a, sc, scc, sic);
    double rs = 0, r2 = 0;
    for (int k = lane; k < nq; k += WAVE) {
        double e = sigma[k] == 0.0 ? 1.0 : sigma[k];
        double r = I[k] - (C[k] * f.A + f.b);
        rs += (r / e) * (r / e); r2 += r * r;
    }
    rs = wave_sum(rs); r2 = wave_sum(r2);
    if (lane == 0) {
        out[0] = f.A; out[1] = f.b; out[2] = rs / nq;
        out[3] = r2 / ss2 * ((double)nq / (double)(nq - num_params));   // aGoFsAlpha :79-84, :136-138
    }
}

// ------------------------------------------------------------------------------ histogram(), batched
// per rep: cumInt = sum of its rows in contribution order (scatteringmodel.py:101), then the closed-form
// scale/background fit (mcsas.py:559); one wave per rep
__global__ __launch_bounds__(64) void hist_fit_kernel(int nq, const double *I, const double *sigma, int N, int R, int r0,
                                                      const double *rows, int find_bg, int pos_bg, double *scaling) {
    const int rl = blockIdx.x, r = r0 + rl, lane = threadIdx.x;
    const double *base = rows + (size_t)rl * N * nq;
    double sw = 0, si = 0, sii = 0, sc = 0, scc = 0, sic = 0;
    for (int k = lane; k < nq; k += WAVE) {
        double C = 0.;
        for (int n = 0; n < N; ++n) C += base[(size_t)n * nq + k];
        const double e = sigma[k] == 0.0 ? 1.0 : sigma[k];
        const double w = 1.0 / (e * e);
        sw += w; si += w * I[k]; sii += w * I[k] * I[k];
        sc += w * C; scc += w * C * C; sic += w * I[k] * C;
    }
    wave_sum3(sw, si, sii); wave_sum3(sc, scc, sic);
    ChainArgs a{};
    a.Sw = sw; a.SI = si; a.SII = sii; a.nq = nq; a.find_bg = find_bg; a.pos_bg = pos_bg;
    const FitResult f = solve_fit(a, sc, scc, sic);
    if (lane == 0) { scaling[r] = f.A; scaling[R + r] = f.b; }
}

// min over q of sigma*vf / (A*I_c(q)), I_c != 0 (mcsas.py:582-590) from the stored rows
__global__ __launch_bounds__(64) void hist_obs_kernel(int nq, const double *sigma, int N, int R, int r0, const double *rows,
                                                      const double *scaling, const double *vset, const double *wset,
                                                      double *min_req) {
    const int c = blockIdx.x, rl = blockIdx.y, r = r0 + rl;
    const double A = scaling[r];
    const double vf = wset[(size_t)c * R + r] * A / vset[(size_t)c * R + r];      // modeldata.py:57-61
    const double *row = rows + ((size_t)rl * N + c) * nq;
    double best = INFINITY;
    for (int k = threadIdx.x; k < nq; k += WAVE) {
        const double scaled = A * row[k];
        if (scaled != 0.) best = fmin(best, (sigma[k] * vf) / scaled);
    }
    best = wave_min(best);
    if (threadIdx.x == 0) min_req[(size_t)c * R + r] = best;
}


// ---- McSAS.histogram() on the device (mcsas_hip_histogram) -----------------------------------------------------------------
// cum[rl][k] = sum_n rows[rl][n][k] in contribution order (scatteringmodel.py:101): one thread per (q, rep), loads in batches of 8
__global__ void hist_colsum_kernel(int nq, int N, const double *rows, double *cum) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, rl = blockIdx.y;
    if (k >= nq) return;
    const double *base = rows + (size_t)rl * N * nq + k;
    double s = 0.;
    int n = 0;
    for (; n + 8 <= N; n += 8) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = base[(size_t)(n + i) * nq];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; n < N; ++n) s += base[(size_t)n * nq];
    cum[(size_t)rl * nq + k] = s;
}
// hist_fit_kernel with the summed intensity given (same sums in the same order): one wave per rep
__global__ __launch_bounds__(64) void hist_fit_cum_kernel(int nq, const double *I, const double *sigma, int R, int r0, const double *cum,
                                                          int find_bg, int pos_bg, double *scaling) {
    const int rl = blockIdx.x, r = r0 + rl, lane = threadIdx.x;
    double sw = 0, si = 0, sii = 0, sc = 0, scc = 0, sic = 0;
    for (int k = lane; k < nq; k += WAVE) {
        const double C = cum[(size_t)rl * nq + k];
        const double e = sigma[k] == 0.0 ? 1.0 : sigma[k];
        const double w = 1.0 / (e * e);
        sw += w; si += w * I[k]; sii += w * I[k] * I[k];
        sc += w * C; scc += w * C * C; sic += w * I[k] * C;
    }
    wave_sum3(sw, si, sii); wave_sum3(sc, scc, sic);
    ChainArgs a{};
    a.Sw = sw; a.SI = si; a.SII = sii; a.nq = nq; a.find_bg = find_bg; a.pos_bg = pos_bg;
    const FitResult f = solve_fit(a, sc, scc, sic);
    if (lane == 0) { scaling[r] = f.A; scaling[R + r] = f.b; }
}
// fractions and their visibility limits, normalised per repetition (mcsas.py:561-604): frac8 = [vf nf qf sf | mv mn mq ms][N][R].
// One wave per rep; the three totals are taken by one lane each, over the contributions in order (builtin sum()).
__global__ __launch_bounds__(64) void hist_fractions_kernel(int N, int R, const double *scaling, const double *vset, const double *wset,
                                                            const double *sset, const double *mv, double *frac8) {
    __shared__ double tot[3];
    const int r = blockIdx.x, lane = threadIdx.x;
    const size_t NR = (size_t)N * R;
    const double A = scaling[r];
    double *vf = frac8, *nf = frac8 + NR, *qf = frac8 + 2 * NR, *sf = frac8 + 3 * NR;
    double *omv = frac8 + 4 * NR, *mn = frac8 + 5 * NR, *mq = frac8 + 6 * NR, *ms = frac8 + 7 * NR;
    for (int c = lane; c < N; c += WAVE) {
        const size_t i = (size_t)c * R + r;
        const double v = vset[i], w = wset[i], s = sset[i], m = mv[i];
        const double vfc = w * A / v;                          // modeldata.py:57-61
        const double nfc = vfc / v, mnc = m / v;               // mcsas.py:567-571, :591-594
        vf[i] = vfc; nf[i] = nfc; qf[i] = vfc * v; sf[i] = nfc * s;
        omv[i] = m; mn[i] = mnc; mq[i] = mnc * m * m; ms[i] = mnc * s;
    }
    __syncthreads();
    if (lane < 3) {
        const double *src = lane == 0 ? nf : (lane == 1 ? qf : sf);
        double t = 0.;
        for (int c = 0; c < N; ++c) t += src[(size_t)c * R + r];
        tot[lane] = t;
    }
    __syncthreads();
    const double tn = tot[0], tq = tot[1], ts = tot[2];
    for (int c = lane; c < N; c += WAVE) {
        const size_t i = (size_t)c * R + r;
        if (tn != 0.) { nf[i] /= tn; mn[i] /= tn; }           // :596-604
        if (tq != 0.) { qf[i] /= tq; mq[i] /= tq; }
        if (ts != 0.) { sf[i] /= ts; ms[i] /= ts; }
    }
}
struct HistSpecDev { int32_t pidx, weight, nb, pad; int64_t edge_off, out_off; double lo, hi; };
// One wave per (histogram, repetition).  The repetition's parameter values, fractions and limits are staged in LDS (3 N doubles);
// lane b owns bin b (64 bins per pass) and walks the contributions in order; the cumulative distribution and the moments are
// sequential sums again, taken by one lane (two for skew and kurtosis).  utils/parameter.py:84-122, :441-479.
__global__ __launch_bounds__(64) void hist_bins_kernel(int N, int P, int R, const double *contribs, const double *frac8, const double *edges_all,
                                                       const HistSpecDev *specs, double *out) {
    extern __shared__ double hl[];
    __shared__ double mom[8];
    const HistSpecDev s = specs[blockIdx.x];
    const int r = blockIdx.y, lane = threadIdx.x, nb = s.nb;
    const size_t NR = (size_t)N * R;
    const double *fr = frac8 + (size_t)s.weight * NR, *lim = frac8 + (size_t)(4 + s.weight) * NR;
    const double *edges = edges_all + s.edge_off;
    double *bins = out + s.out_off, *obs = bins + (size_t)nb * R, *cdf = obs + (size_t)nb * R, *mo = cdf + (size_t)nb * R;
    double *lx = hl, *lf = hl + N, *ll = hl + 2 * (size_t)N;
    for (int c = lane; c < N; c += WAVE) {
        lx[c] = contribs[((size_t)c * P + s.pidx) * R + r];
        lf[c] = fr[(size_t)c * R + r]; ll[c] = lim[(size_t)c * R + r];
    }
    __syncthreads();
    const double last = nb > 0 ? edges[nb] : 0.;
    for (int b0 = 0; b0 < nb; b0 += WAVE) {
        const int b = b0 + lane;
        if (b < nb) {
            const double elo = edges[b], ehi = edges[b + 1];
            double sb = 0., so = 0., cnt = 0.;
            for (int c = 0; c < N; ++c) {
                const double x = lx[c];
                if (x >= elo && x < ehi && x < last) { sb += lf[c]; so += ll[c]; cnt += 1.; }
            }
            bins[(size_t)b * R + r] = (sb != sb) ? 0. : sb;    // bins[isnan(bins)] = 0
            obs[(size_t)b * R + r] = cnt > 0. ? so / cnt : 0.;
        }
    }
    __syncthreads();
    if (lane == 0) {
        double run = 0., top = -__builtin_inf();
        for (int b = 0; b < nb; ++b) { run += bins[(size_t)b * R + r]; cdf[(size_t)b * R + r] = run; top = fmax(top, run); }
        for (int b = 0; b < nb; ++b) cdf[(size_t)b * R + r] = (top == 0.) ? 0. : cdf[(size_t)b * R + r] / top;
    }
    // moments: total, mean, variance by lane 32 (runs beside lane 0's distribution), then skew / kurtosis by lanes 32 / 33
    if (lane == 32) {
        double tot = 0., m1 = 0.;
        int any = 0;
        for (int c = 0; c < N; ++c) { const double x = lx[c]; const bool ok = x > s.lo && x < s.hi; any |= ok; tot += ok ? lf[c] : 0.; }
        for (int c = 0; c < N; ++c) { const double x = lx[c]; const bool ok = x > s.lo && x < s.hi; m1 += ok ? x * lf[c] : 0.; }
        if (tot != 0.) m1 /= tot;
        double v2 = 0.;
        for (int c = 0; c < N; ++c) { const double x = lx[c], d = x - m1; const bool ok = x > s.lo && x < s.hi; v2 += ok ? (d * d) * lf[c] : 0.; }
        const double var = v2 / tot, sg = sqrt(fabs(var));
        mom[0] = tot; mom[1] = m1; mom[2] = var; mom[3] = sg; mom[4] = (double)any;
    }
    __syncthreads();
    if (lane == 32 || lane == 33) {
        const double tot = mom[0], m1 = mom[1], sg = mom[3];
        const bool any = mom[4] != 0., ok3 = any && (tot * sg) != 0.;      // (utils/parameter.py:106-107: only an exact zero is skipped; NaN goes on)
        const double sg2 = sg * sg;
        double acc = 0.;
        for (int c = 0; c < N; ++c) {
            const double x = lx[c], d = x - m1, d2 = d * d;
            const bool ok = x > s.lo && x < s.hi;
            acc += ok ? (lane == 32 ? d2 * d : d2 * d2) * lf[c] : 0.;
        }
        const double val = ok3 ? acc / (tot * (lane == 32 ? sg2 * sg : sg2 * sg2)) : 0.;
        mo[(size_t)(3 + (lane - 32)) * R + r] = val;
        if (lane == 32) { mo[r] = any ? tot : 0.; mo[(size_t)R + r] = any ? m1 : 0.; mo[(size_t)2 * R + r] = any ? mom[2] : 0.; }
    }
}

// ------------------------------------------------------------------------------ input preparation
// DataObj._prepareUncertainty (dataobj/dataobj.py:204-227)
__global__ void prepare_uncertainty_kernel(int n, const double *I, const double *su, double fu_min, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double floor_u = fu_min * I[i];
    double u = su ? fmax(su[i], floor_u) : floor_u;          // numpy.maximum propagates NaN; fmax would not:
    if (su && (su[i] != su[i] || floor_u != floor_u)) u = NAN;
    out[i] = isfinite(u) ? u : INFINITY;
}

// DataObj._reBin (dataobj/dataobj.py:319-337): one wavefront per bin
__global__ __launch_bounds__(64) void rebin_kernel(int n, const double *x, const double *f, const double *fu,
                                                   const double *edges, double *xb, double *fb, double *ub, int32_t *cnt_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double lo = edges[b], hi = edges[b + 1];
    double cnt = 0., sx = 0., sf = 0., su2 = 0., x1 = 0., f1 = 0., u1 = 0.;
    for (int i = lane; i < n; i += WAVE) {
        const double xi = x[i];
        if (xi >= lo && xi < hi) { cnt += 1.; sx += xi; sf += f[i]; su2 += fu[i] * fu[i]; x1 = xi; f1 = f[i]; u1 = fu[i]; }
    }
    const double mine = cnt;
    cnt = wave_sum(cnt); sx = wave_sum(sx); sf = wave_sum(sf); su2 = wave_sum(su2);
    const int c = (int)cnt;
    if (c == 1) {                                          // the lane that holds the point copies it (:327-329)
        if (mine == 1.) { xb[b] = x1; fb[b] = f1; ub[b] = u1; }
    } else if (c > 1) {
        const double mf = sf / cnt;
        double ss = 0.;
        for (int i = lane; i < n; i += WAVE) {
            const double xi = x[i];
            if (xi >= lo && xi < hi) { const double dlt = f[i] - mf; ss += dlt * dlt; }
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            xb[b] = sx / cnt; fb[b] = mf;
            const double sem = sqrt(ss / (cnt - 1.)) / sqrt(cnt), prop = sqrt(su2 / cnt);
            ub[b] = (sem != sem || prop != prop) ? NAN : fmax(sem, prop);     // numpy.maximum
        }
    }
    if (lane == 0) cnt_out[b] = c;
}

// ------------------------------------------------------------------------------ run-time model plug-ins (hiprtc)
// The reference takes any models/*.py that subclasses ScatteringModel (utils/findmodels.py:120-186); here a model outside
// the eight built-in ones arrives as HIP source text defining the four functions of plugin_model.h.  It is compiled for gfx950
// against the library's OWN kernel headers (embedded at build time: embedded_headers.inc) — the wave-per-chain kernel and the
// three model-templated small kernels, instantiated for Contrib<MCSAS_MODEL_PLUGIN> — and loaded as a code-object module on
// every device that uses it.  Compilation needs no GPU; loading does.
#include "embedded_headers.inc"

struct Plugin {
    std::string source;
    int row_class = 0;                  // `#define MCSAS_PLUGIN_ROW_CLASS n` in the text (plugin_model.h)
    bool can_smear = false;             // `#define MCSAS_PLUGIN_CAN_SMEAR 1`
    std::mutex mu;
    struct Program { std::vector<char> code; std::map<std::string, std::string> lowered; };
    std::map<std::string, Program> programs;                                  // by program key ("small", "wave 8 1", ...)
    std::map<std::pair<int, std::string>, hipModule_t> modules;               // (device, program key)
};
static std::mutex g_plugins_mu;
static std::vector<std::unique_ptr<Plugin>> g_plugins;
static thread_local std::string g_plugin_log;

static bool is_plugin_model(int model_id) { return model_id >= MCSAS_MODEL_PLUGIN0 && model_id < MCSAS_MODEL_PLUGIN0 + MCSAS_MAX_PLUGINS; }
static Plugin *plugin_of(int model_id) {
    std::lock_guard<std::mutex> lk(g_plugins_mu);
    const int k = model_id - MCSAS_MODEL_PLUGIN0;
    return (k >= 0 && k < (int)g_plugins.size()) ? g_plugins[k].get() : nullptr;
}

// one translation unit: the kernel headers, the plug-in's Contrib, the plug-in text; `exprs` = the kernels to instantiate
// value of the LAST `#define KEY value` line of the text (whitespace after '#' and around the name, parentheses around the
// value allowed), `dflt` when there is none.  What this returns is checked against the preprocessor (see mcsas_hip_plugin_compile).
static long plugin_scan_define(const std::string &src, const char *key, long dflt) {
    long val = dflt;
    const size_t klen = strlen(key);
    size_t pos = 0;
    while (pos < src.size()) {
        size_t eol = src.find('\n', pos);
        if (eol == std::string::npos) eol = src.size();
        size_t i = pos;
        auto skip_ws = [&]() { while (i < eol && (src[i] == ' ' || src[i] == '\t')) ++i; };
        skip_ws();
        if (i < eol && src[i] == '#') {
            ++i; skip_ws();
            if (src.compare(i, 6, "define") == 0) {
                i += 6;
                const size_t before = i;
                skip_ws();
                if (i > before && src.compare(i, klen, key) == 0 && (i + klen >= eol || src[i + klen] == ' ' || src[i + klen] == '\t')) {
                    i += klen; skip_ws();
                    while (i < eol && src[i] == '(') { ++i; skip_ws(); }
                    val = i < eol ? strtol(src.c_str() + i, nullptr, 0) : 1;     // `#define KEY` alone: defined, i.e. 1 where it is used as a flag
                }
            }
        }
        pos = eol + 1;
    }
    return val;
}

static int plugin_compile_program(const Plugin &pg, const std::vector<std::string> &exprs, Plugin::Program *out) {
    const std::string &source = pg.source;
    // hiprtc has the fixed-width integer types in a namespace of its own
    std::string tu =
        "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
        "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;\n"
        "#include \"chain_common.h\"\n#line 1 \"plugin\"\n";
    tu += source;
    tu += "\n#include \"plugin_model.h\"\n#include \"chain_wave.h\"\n#include \"chain_wg.h\"\n#include \"chain_wide.h\"\n#include \"chain_pipe.h\"\n#include \"small_kernels.h\"\n";
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, tu.c_str(), "mcsas_plugin.hip", mcsas_embedded_count, const_cast<const char **>(mcsas_embedded_texts),
                                         const_cast<const char **>(mcsas_embedded_names));
    if (r != HIPRTC_SUCCESS) return fail(MCSAS_EHIP, "hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
    for (const std::string &e : exprs) hiprtcAddNameExpression(prog, e.c_str());
    // the flags of the Makefile: one contribution row must come out the same from every call site (-ffp-contract=off)
    char host_rc[48], host_cs[48];
    snprintf(host_rc, sizeof host_rc, "-DMCSAS_HOST_ROW_CLASS=%d", pg.row_class);
    snprintf(host_cs, sizeof host_cs, "-DMCSAS_HOST_CAN_SMEAR=%d", pg.can_smear ? 1 : 0);
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", host_rc, host_cs};
    r = hiprtcCompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    size_t ls = 0;
    g_plugin_log.clear();
    if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { g_plugin_log.resize(ls); hiprtcGetProgramLog(prog, &g_plugin_log[0]); }
    if (r != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        return fail(MCSAS_EINVAL, "model plug-in does not compile (%s): see mcsas_hip_plugin_log()", hiprtcGetErrorString(r));
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    out->code.resize(cs);
    hiprtcGetCode(prog, out->code.data());
    for (const std::string &e : exprs) {
        const char *ln = nullptr;
        if (hiprtcGetLoweredName(prog, e.c_str(), &ln) != HIPRTC_SUCCESS || !ln) { hiprtcDestroyProgram(&prog); return fail(MCSAS_EHIP, "no lowered name for %s", e.c_str()); }
        out->lowered[e] = ln;
    }
    hiprtcDestroyProgram(&prog);
    return MCSAS_OK;
}

static const char *const PLUGIN_SMALL_EXPRS[3] = {"mcsas::model_rows_kernel<MCSAS_MODEL_PLUGIN>", "mcsas::observability_kernel<MCSAS_MODEL_PLUGIN>",
                                                  "mcsas::hist_rows_kernel<MCSAS_MODEL_PLUGIN>"};

// kernel `expr` of program `key` (compiled on first use) as a function of the CURRENT device's module
static int plugin_function(int model_id, const std::string &key, const std::vector<std::string> &exprs, const std::string &expr, hipFunction_t *fn) {
    Plugin *pg = plugin_of(model_id);
    if (!pg) return fail(MCSAS_EINVAL, "model_id %d: no such plug-in (mcsas_hip_plugin_compile returns the id)", model_id);
    std::lock_guard<std::mutex> lk(pg->mu);
    auto it = pg->programs.find(key);
    if (it == pg->programs.end()) {
        Plugin::Program prg;
        int rc = plugin_compile_program(*pg, exprs, &prg);
        if (rc) return rc;
        it = pg->programs.emplace(key, std::move(prg)).first;
    }
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    auto mk = std::make_pair(dev, key);
    auto mi = pg->modules.find(mk);
    if (mi == pg->modules.end()) {
        hipModule_t mod = nullptr;
        HIPCHK(hipModuleLoadData(&mod, it->second.code.data()));
        mi = pg->modules.emplace(mk, mod).first;
    }
    HIPCHK(hipModuleGetFunction(fn, mi->second, it->second.lowered.at(expr).c_str()));
    return MCSAS_OK;
}
static int plugin_small_function(int model_id, int which, hipFunction_t *fn) {
    const std::vector<std::string> exprs(PLUGIN_SMALL_EXPRS, PLUGIN_SMALL_EXPRS + 3);
    return plugin_function(model_id, "small", exprs, exprs[which], fn);
}
static int plugin_wg_function(int model_id, int qpl, hipFunction_t *fn) {
    char e[128], k[32];
    snprintf(e, sizeof e, "mcsas::chain_wg_kernel<MCSAS_MODEL_PLUGIN, %d>", qpl);
    snprintf(k, sizeof k, "wg %d", qpl);
    return plugin_function(model_id, k, {e}, e, fn);
}
static int plugin_wide_function(int model_id, int qpl, hipFunction_t *fn) {
    char e[128], k[32];
    snprintf(e, sizeof e, "mcsas::chain_wide_kernel<MCSAS_MODEL_PLUGIN, %d>", qpl);
    snprintf(k, sizeof k, "wide %d", qpl);
    return plugin_function(model_id, k, {e}, e, fn);
}
static int plugin_pipe_function(int model_id, int qpl, bool rowq, hipFunction_t *fn) {
    char e[128], k[32];
    snprintf(e, sizeof e, "mcsas::pipe_tick_kernel<MCSAS_MODEL_PLUGIN, %d, %s>", qpl, rowq ? "true" : "false");
    snprintf(k, sizeof k, "pipe %d %d", qpl, rowq ? 1 : 0);
    return plugin_function(model_id, k, {e}, e, fn);
}
static int plugin_wave_function(int model_id, int qpl, bool cache, hipFunction_t *fn) {
    char e[128], k[32];
    snprintf(e, sizeof e, "mcsas::chain_wave_kernel<MCSAS_MODEL_PLUGIN, %d, %s>", qpl, cache ? "true" : "false");
    snprintf(k, sizeof k, "wave %d %d", qpl, cache ? 1 : 0);
    return plugin_function(model_id, k, {e}, e, fn);
}

// launch of small kernel `which` (PLUGIN_SMALL_EXPRS) of a plug-in on the null stream; the arguments are passed by address,
// so their types must be the kernel's parameter types exactly
template <class... A>
static int plugin_small_launch(int model_id, int which, dim3 grid, size_t lds, A... args) {
    hipFunction_t fn = nullptr;
    int rc = plugin_small_function(model_id, which, &fn);
    if (rc) return rc;
    void *ka[] = {(void *)&args...};
    HIPCHK(hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, WAVE, 1, 1, (unsigned)lds, nullptr, ka, nullptr));
    return MCSAS_OK;
}

extern "C" int mcsas_hip_plugin_compile(const char *source, int32_t *model_id) {
    if (!source || !model_id) return fail(MCSAS_EINVAL, "null argument");
    *model_id = -1;
    {
        std::lock_guard<std::mutex> lk(g_plugins_mu);
        for (size_t k = 0; k < g_plugins.size(); ++k)
            if (g_plugins[k]->source == source) { *model_id = MCSAS_MODEL_PLUGIN0 + (int)k; return MCSAS_OK; }   // the same text again
    }
    // the small kernels are compiled here, so that a plug-in that does not compile is refused before anything uses it
    auto pg = std::make_unique<Plugin>();
    pg->source = source;
    // The row class and the canSmear flag the text declares: the host needs the numbers for the pipeline's geometry and for the
    // smearing tables, the kernels see the macros.  The host's reading of the text (a line-wise scan for `# define KEY value`)
    // is passed to EVERY compilation of this plug-in as -DMCSAS_HOST_ROW_CLASS / -DMCSAS_HOST_CAN_SMEAR and plugin_model.h
    // static_asserts that the preprocessor's values are the same: a text the scan misreads (a define inside a comment or an
    // `#if 0`, an expression for a value) does not compile and is refused here instead of running with two different answers.
    pg->row_class = (int)plugin_scan_define(pg->source, "MCSAS_PLUGIN_ROW_CLASS", 0);
    pg->can_smear = plugin_scan_define(pg->source, "MCSAS_PLUGIN_CAN_SMEAR", 0) != 0;
    if (pg->row_class < 0 || pg->row_class > 1) return fail(MCSAS_EINVAL, "MCSAS_PLUGIN_ROW_CLASS %d (0 or 1)", pg->row_class);
    Plugin::Program prg;
    const std::vector<std::string> exprs(PLUGIN_SMALL_EXPRS, PLUGIN_SMALL_EXPRS + 3);
    int rc = plugin_compile_program(*pg, exprs, &prg);
    if (rc) return rc;
    pg->programs.emplace("small", std::move(prg));
    std::lock_guard<std::mutex> lk(g_plugins_mu);
    if ((int)g_plugins.size() >= MCSAS_MAX_PLUGINS) return fail(MCSAS_EINVAL, "more than %d model plug-ins", MCSAS_MAX_PLUGINS);
    g_plugins.push_back(std::move(pg));
    *model_id = MCSAS_MODEL_PLUGIN0 + (int)g_plugins.size() - 1;
    return MCSAS_OK;
}
extern "C" const char *mcsas_hip_plugin_log(void) { return g_plugin_log.c_str(); }

// ------------------------------------------------------------------------------ host helpers
// what the host needs to know about a model, read off its Contrib<M> (models.h) — no per-model code below this table
struct ModelTraits { int int_div_param, rowtab, row_class; bool can_smear; int (*table_doubles)(int); int contrib_doubles; };
static ModelTraits model_traits(int model_id) {
#define TRAITS_OF(m) {Contrib<m>::INT_DIV_PARAM, Contrib<m>::ROWTAB, Contrib<m>::ROW_CLASS, Contrib<m>::CAN_SMEAR, &Contrib<m>::table_doubles, (int)(sizeof(Contrib<m>) / 8)},
    static const ModelTraits builtin[] = {MCSAS_FOR_MODELS(TRAITS_OF)};
#undef TRAITS_OF
    static_assert(sizeof builtin / sizeof builtin[0] == MCSAS_MODEL_COUNT, "model_list.h and include/mcsas_hip.h disagree");
    if (model_id >= 0 && model_id < MCSAS_MODEL_COUNT) return builtin[model_id];
    const Plugin *pg = plugin_of(model_id);                                           // plugin_model.h
    // (Contrib<MCSAS_MODEL_PLUGIN>, plugin_model.h: the full parameter vector and three doubles)
    return ModelTraits{-1, 0, pg ? pg->row_class : 0, pg ? pg->can_smear : false, [](int) { return 0; }, MCSAS_MAX_PARAMS + 3};
}
static int model_int_div(const mcsas_problem *p) {
    const int i = model_traits(p->model_id).int_div_param;
    return i < 0 ? 1 : (int)p->params[i];
}

// per-wave scratch for the per-row orientation table (Contrib<M>::ROWTAB * K doubles, models.h); beyond
// K = 256 the chain kernels evaluate the integrand directly
static int rowtab_doubles_host(int model_id, int K) { return K > 256 ? 0 : model_traits(model_id).rowtab * K; }

static int fill_model_args(const mcsas_problem *p, ModelArgs *m) {
    if ((p->model_id < 0 || p->model_id >= MCSAS_MODEL_COUNT) && !(is_plugin_model(p->model_id) && plugin_of(p->model_id)))
        return fail(MCSAS_EINVAL, "unknown model_id %d", p->model_id);
    if (p->n_active < 0 || p->n_active > MCSAS_MAX_ACTIVE) return fail(MCSAS_EINVAL, "n_active %d out of range", p->n_active);
    memset(m, 0, sizeof *m);
    m->model_id = p->model_id; m->n_active = p->n_active; m->comp_exp = p->comp_exp;
    for (int i = 0; i < MCSAS_MAX_PARAMS; ++i) m->params[i] = p->params[i];
    for (int c = 0; c < MCSAS_MAX_ACTIVE; ++c) {
        m->active_index[c] = c < p->n_active ? p->active_index[c] : -1;
        m->clip_lo[c] = p->clip_lo[c]; m->clip_hi[c] = p->clip_hi[c];
        if (c < p->n_active && (p->active_index[c] < 0 || p->active_index[c] >= MCSAS_MAX_PARAMS))
            return fail(MCSAS_EINVAL, "active_index[%d]=%d out of range", c, p->active_index[c]);
    }
    m->int_div = model_int_div(p);
    m->use_rowtab = rowtab_doubles_host(p->model_id, m->int_div) > 0;
    m->qmax = 0.;
    if (p->q) for (int i = 0; i < p->nq; ++i) m->qmax = std::max(m->qmax, std::fabs(p->q[i]));
    if (model_traits(p->model_id).int_div_param >= 0 && (m->int_div < 2 || m->int_div > 4096))
        return fail(MCSAS_EINVAL, "intDiv %d unsupported (2..4096)", m->int_div);
    return MCSAS_OK;
}

static int table_doubles_host(int model_id, int K) { return model_traits(model_id).table_doubles(K); }
// Device copy of the smearing tables of a problem: locs transposed to [K][stride] (pad columns repeat
// column 0, like the padded q) and cw[m] = 2 * trapezoid coefficient(q_offset)[m] * weights[m], so that
// sum_m cw[m] y[m] = 2 trapz(y * weights, x = q_offset) (sasmodel.py:72-73).
struct SmearDev {
    double *locs_t = nullptr, *cw = nullptr;
    ~SmearDev() { if (locs_t) hipFree(locs_t); if (cw) hipFree(cw); }
    int upload(const mcsas_problem *p, int stride, ModelArgs *m) {
        m->smear_nk = 0; m->smear_stride = 0; m->smear_locs_t = nullptr; m->smear_cw = nullptr;
        if (p->smear_nk <= 0) return MCSAS_OK;
        if (!model_traits(p->model_id).can_smear) return MCSAS_OK;   // canSmear = False
        const int K = p->smear_nk;
        if (K < 2 || K > 4096 || !p->smear_locs || !p->smear_q_offset || !p->smear_weights)
            return fail(MCSAS_EINVAL, "smearing: smear_nk %d (2..4096) needs locs, q_offset and weights", K);
        std::vector<double> lt((size_t)K * stride), cw(K);
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < stride; ++i) lt[(size_t)k * stride + i] = p->smear_locs[(size_t)(i < p->nq ? i : 0) * K + k];
        for (int k = 0; k < K; ++k) {
            const double dl = k > 0 ? p->smear_q_offset[k] - p->smear_q_offset[k - 1] : 0.;
            const double dr = k + 1 < K ? p->smear_q_offset[k + 1] - p->smear_q_offset[k] : 0.;
            cw[k] = 2. * (0.5 * (dl + dr)) * p->smear_weights[k];
        }
        HIPCHK(hipMalloc(&locs_t, sizeof(double) * lt.size()));
        HIPCHK(hipMalloc(&cw_dev(), sizeof(double) * K));
        HIPCHK(hipMemcpy(locs_t, lt.data(), sizeof(double) * lt.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(this->cw, cw.data(), sizeof(double) * K, hipMemcpyHostToDevice));
        m->smear_nk = K; m->smear_stride = stride; m->smear_locs_t = locs_t; m->smear_cw = this->cw;
        for (size_t i = 0; i < lt.size(); ++i) m->qmax = std::max(m->qmax, std::fabs(lt[i]));
        return MCSAS_OK;
    }
    double *&cw_dev() { return cw; }
};

static int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MCSAS_ENODEV, "no HIP device available");
    if (device >= n) return fail(MCSAS_ENODEV, "device %d requested, %d present", device, n);
    if (device >= 0) HIPCHK(hipSetDevice(device));
    return MCSAS_OK;
}

// Every entry point that selects a device puts the calling thread's current device back on the way out: a
// host that shares the HIP runtime (torch with RCCL in bench.py, the hosts of INTEGRATION.md) keeps its own.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// ------------------------------------------------------------------------------ plan
// Device memory of a plan.  The big arrays (row cache, window buffers, replay streams: tens of MB) are allocations of their
// own; everything else — the data vectors, the chains' records and running sums, parameter sets, the windows' scalars and
// proposals, ~3 MB at config 2 — is carved out of 8 MB chunks, 256-byte aligned: a tick's blocks each touch a dozen of
// these small arrays before their first row, and packed into one 2 MB-aligned range they share a handful of page-table
// entries instead of one scattered 4 KB / 64 KB page per array (and a plan costs 6 hipMalloc calls instead of 25).
// Memory that plans give back is kept for the next plan (per device, exact size): a series of analyses — McSAS.calc() per data
// set, one plan each — otherwise pays ~3 ms of hipFree / hipHostFree per plan for 3.4 ms of kernels, plus as much for stream and
// pinned-memory creation.  Up to 4 GiB of device memory (MCSAS_HIP_CACHE_MB in the environment: another cap, 0 = none) stay parked;
// mcsas_hip_release_cached_memory() frees them.
struct MemCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> dev;        // (device, bytes) -> free device blocks
    std::multimap<std::pair<unsigned, size_t>, void *> host;  // (flags, bytes) -> free pinned blocks
    std::map<int, hipStream_t> copy_stream;                   // per device: the non-blocking stream results come back on
    size_t dev_bytes = 0;
    // how much device memory may stay parked per process: 4 GiB, or MCSAS_HIP_CACHE_MB from the environment (0: nothing is kept —
    // for hosts whose own allocator (torch's, say) should see every byte this library is not using)
    size_t cap_bytes = (size_t)4 << 30;
    MemCache() {
        if (const char *e = getenv("MCSAS_HIP_CACHE_MB")) { char *end = nullptr; const long long mb = strtoll(e, &end, 10); if (end != e && mb >= 0) cap_bytes = (size_t)mb << 20; }
    }
};
static MemCache &mem_cache() { static MemCache c; return c; }
static hipError_t cached_dev_malloc(void **p, size_t n) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    {
        MemCache &c = mem_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.dev.find({d, n});
        if (it != c.dev.end()) { *p = it->second; c.dev.erase(it); c.dev_bytes -= n; return hipSuccess; }
    }
    e = hipMalloc(p, n);
    if (e == hipErrorOutOfMemory) {                           // parked blocks of other sizes may be in the way
        { MemCache &c = mem_cache(); std::lock_guard<std::mutex> lk(c.mu); for (auto &kv : c.dev) hipFree(kv.second); c.dev.clear(); c.dev_bytes = 0; }
        (void)hipGetLastError();
        e = hipMalloc(p, n);
    }
    return e;
}
static void cached_dev_free(void *p, size_t n, int d) {
    MemCache &c = mem_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.dev_bytes + n <= c.cap_bytes) { c.dev.insert({{d, n}, p}); c.dev_bytes += n; }
    else hipFree(p);
}
static hipError_t cached_host_malloc(void **p, size_t n, unsigned flags) {
    {
        MemCache &c = mem_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.host.find({flags, n});
        if (it != c.host.end()) { *p = it->second; c.host.erase(it); return hipSuccess; }
    }
    return hipHostMalloc(p, n, flags);
}
static void cached_host_free(void *p, size_t n, unsigned flags) {
    if (!p) return;
    MemCache &c = mem_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.cap_bytes > 0 && c.host.size() < 64) c.host.insert({{flags, n}, p});
    else hipHostFree(p);
}
static hipError_t cached_copy_stream(hipStream_t *st) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    MemCache &c = mem_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.copy_stream.find(d);
    if (it == c.copy_stream.end()) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        it = c.copy_stream.emplace(d, s).first;
    }
    *st = it->second;
    return hipSuccess;
}
extern "C" int mcsas_hip_release_cached_memory(void) {
    MemCache &c = mem_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    for (auto &kv : c.dev) hipFree(kv.second);
    for (auto &kv : c.host) hipHostFree(kv.second);
    c.dev.clear(); c.host.clear(); c.dev_bytes = 0;
    return MCSAS_OK;
}

struct DevPool {
    static constexpr size_t CHUNK = (size_t)8 << 20, BIG = (size_t)2 << 20, ALIGN = 256;
    std::vector<std::pair<void *, size_t>> blocks;   // everything to give back
    int dev = 0;
    char *cur = nullptr;
    size_t left = 0;
    hipError_t get(void **out, size_t n) {
        *out = nullptr;
        n = (n ? n : 1);
        n = (n + ALIGN - 1) / ALIGN * ALIGN;
        void *p = nullptr;
        if (blocks.empty()) (void)hipGetDevice(&dev);
        if (n >= BIG) {
            hipError_t e = cached_dev_malloc(&p, n);
            if (e != hipSuccess) return e;
            blocks.push_back({p, n}); *out = p;
            return hipSuccess;
        }
        if (n > left) {                 // (the tail of the old chunk stays unused)
            hipError_t e = cached_dev_malloc(&p, CHUNK);
            if (e != hipSuccess) return e;
            blocks.push_back({p, (size_t)CHUNK}); cur = (char *)p; left = CHUNK;
        }
        *out = cur; cur += n; left -= n;
        return hipSuccess;
    }
    template <class T> hipError_t get(T **out, size_t n_bytes) { return get((void **)out, n_bytes); }
    void release() { for (auto &b : blocks) cached_dev_free(b.first, b.second, dev); blocks.clear(); cur = nullptr; left = 0; }
};

struct mcsas_plan {
    DevPool pool;
    mcsas_problem prob;
    ChainArgs args;
    SmearDev smear;                     // device copy of the smearing tables (empty when off)
    int qpl = 0, waves = 1, use_cache = 1, dev = 0;
    bool wide = false;                  // more than 1024 q-points, one workgroup per chain with the q-points split over its waves (chain_wide.h)
    size_t lds_bytes = 0;
    double *d_q = nullptr, *d_w = nullptr, *d_wI = nullptr, *d_I = nullptr, *d_q3inv = nullptr;
    double *d_rset = nullptr, *d_cache = nullptr, *d_fit = nullptr, *d_replay = nullptr;
    ChainOut *d_out = nullptr;
    int32_t *h_stop = nullptr;          // pinned + mapped: McSAS.stop as the kernels see it
    int32_t *d_stop_relay = nullptr;    // device memory, 16 bytes: the relayed stop word and the time stamp of the last look at h_stop (chain_common.h: stop_requested)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;
    bool launched = false;
    bool enqueue_failed = false;        // a launch returned an error after it had put work on `stream`: nothing marks where that work ends
    double last_ms = 0.;
    int64_t last_steps = 0;
    WgGeom wg{};
    // whole-chip pipeline (exec_mode 3)
    int mode = MCSAS_EXEC_WAVE;
    PipeArgs pipe{};
    PipeChain *d_chains = nullptr;
    double *d_ft = nullptr, *d_wft = nullptr, *d_dwin = nullptr, *d_gwin = nullptr, *d_scal = nullptr, *d_pval = nullptr;
    int32_t *d_slot_of = nullptr, *d_stage = nullptr, *d_povf = nullptr, *d_row_valid = nullptr, *d_rowq = nullptr;
    uint64_t *d_timeline = nullptr;
    int32_t *h_done = nullptr;          // pinned + mapped: the scan block of the last chain to finish writes the chain count into it
    int32_t *d_done_dev = nullptr;      // the device counter behind it
    PipeArgs *d_pipeargs = nullptr;     // the argument block the tick kernels read (device copy)
    PipeArgs *h_pipeargs = nullptr;     // ... and its pinned staging copy: the upload is a true asynchronous copy, so launch() of one plan does
                                        // not wait for another plan's work queued on the same stream
    hipStream_t sCopy = nullptr;        // fetch(): results come back on a non-blocking stream of their own (a blocking copy would
                                        // wait for whatever else is queued on the null stream — another plan's launch)
    hipFunction_t plugin_fn = nullptr;  // the chain kernel of a run-time model plug-in for this plan's mode and q count (this device's module), else null
    hipStream_t sP = nullptr, sS = nullptr;
    static constexpr int RING = 64;
    hipEvent_t evP[RING] = {}, evS[RING] = {};
    int ticks_launched = 0;
    // Result slots (mcsas_hip_plan_launch_slot / _fetch_slot): everything a finished analysis is read back from — parameter
    // sets, fits, per-chain outputs, the timing events, the host-visible "all chains done" word — exists MCSAS_PLAN_SLOTS times,
    // the workspaces (row cache, window buffers, chain records) once.  The plan's own members above are the view of the slot
    // that was last activated; the others are parked here.
    struct Slot {
        double *d_rset = nullptr, *d_fit = nullptr;
        ChainOut *d_out = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        int32_t *h_done = nullptr, *d_done_map = nullptr;
        PipeArgs *h_pipeargs = nullptr;
        bool launched = false, made = false;
        int ticks_launched = 0;
        double last_ms = 0.;
        int64_t last_steps = 0;
    };
    Slot slots[MCSAS_PLAN_SLOTS];
    int cur_slot = 0;
};

// make slot k the plan's active view (allocating it on first use)
static int plan_activate_slot(mcsas_plan *pl, int k) {
    if (k < 0 || k >= MCSAS_PLAN_SLOTS) return fail(MCSAS_EINVAL, "slot %d (0..%d)", k, MCSAS_PLAN_SLOTS - 1);
    if (k == pl->cur_slot) return MCSAS_OK;
    mcsas_plan::Slot &o = pl->slots[pl->cur_slot], &n = pl->slots[k];
    o.d_rset = pl->d_rset; o.d_fit = pl->d_fit; o.d_out = pl->d_out; o.ev0 = pl->ev0; o.ev1 = pl->ev1; o.h_done = pl->h_done;
    o.d_done_map = pl->pipe.n_done; o.h_pipeargs = pl->h_pipeargs; o.launched = pl->launched; o.ticks_launched = pl->ticks_launched;
    o.last_ms = pl->last_ms; o.last_steps = pl->last_steps; o.made = true;
    if (!n.made) {
        const size_t R = pl->prob.n_reps, N = pl->prob.n_contrib, P = pl->prob.n_active, qpad = pl->args.qpad;
        HIPCHK(pl->pool.get(&n.d_rset, sizeof(double) * R * N * P));
        HIPCHK(pl->pool.get(&n.d_fit, sizeof(double) * R * qpad));
        HIPCHK(pl->pool.get(&n.d_out, sizeof(ChainOut) * R));
        HIPCHK(hipMemset(n.d_out, 0, sizeof(ChainOut) * R));
        HIPCHK(hipEventCreate(&n.ev0)); HIPCHK(hipEventCreate(&n.ev1));
        if (pl->mode == MCSAS_EXEC_PIPELINE) {
            HIPCHK(cached_host_malloc((void **)&n.h_done, sizeof(int32_t), hipHostMallocMapped));
            *n.h_done = 0;
            HIPCHK(hipHostGetDevicePointer((void **)&n.d_done_map, n.h_done, 0));
        }
        n.made = true;
    }
    pl->d_rset = n.d_rset; pl->d_fit = n.d_fit; pl->d_out = n.d_out; pl->ev0 = n.ev0; pl->ev1 = n.ev1; pl->h_done = n.h_done;
    pl->pipe.n_done = n.d_done_map; pl->h_pipeargs = n.h_pipeargs; pl->launched = n.launched; pl->ticks_launched = n.ticks_launched;
    pl->last_ms = n.last_ms; pl->last_steps = n.last_steps;
    pl->args.rset = pl->d_rset; pl->args.fit = pl->d_fit; pl->args.out = pl->d_out;
    pl->cur_slot = k;
    return MCSAS_OK;
}

// kernel lookups, one translation unit per model (kern_wave.hip / kern_wg.hip)
#define DECL_K(m) void *mcsas_wave_kernel_m##m(int, bool); void *mcsas_wg_kernel_m##m(int); void *mcsas_wide_kernel_m##m(int); void *mcsas_pipe_tick_kernel_m##m(int, bool);
MCSAS_FOR_MODELS(DECL_K)
#undef DECL_K
void *mcsas_pipe_reset_kernel();

static void *pipe_tick_kernel_for(int model, int qpl, bool rowq) {
    switch (model) {
#define CASE_K(m) case m: return mcsas_pipe_tick_kernel_m##m(qpl, rowq);
        MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
        default: return nullptr;
    }
}

static void *wave_kernel_for(int model, int qpl, bool cache) {
    switch (model) {
#define CASE_K(m) case m: return mcsas_wave_kernel_m##m(qpl, cache);
        MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
        default: return nullptr;
    }
}
static void *wide_kernel_for(int model, int qpl) {
    switch (model) {
#define CASE_K(m) case m: return mcsas_wide_kernel_m##m(qpl);
        MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
        default: return nullptr;
    }
}
static void *wg_kernel_for(int model, int qpl) {
    switch (model) {
#define CASE_K(m) case m: return mcsas_wg_kernel_m##m(qpl);
        MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
        default: return nullptr;
    }
}

extern "C" void mcsas_hip_plan_destroy(mcsas_plan *pl) {
    if (!pl) return;
    // (nothing of this plan may still be running when its memory goes back to the cache: hipFree used to wait, the cache does not)
    if (pl->enqueue_failed) {           // a launch that failed half-way left kernels behind no event of ours: wait for the stream (and the device)
        (void)hipSetDevice(pl->dev);
        (void)hipStreamSynchronize(pl->stream);
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    if (pl->launched && pl->ev1) (void)hipEventSynchronize(pl->ev1);
    for (int k = 0; k < MCSAS_PLAN_SLOTS; ++k)
        if (k != pl->cur_slot && pl->slots[k].made && pl->slots[k].launched && pl->slots[k].ev1) (void)hipEventSynchronize(pl->slots[k].ev1);
    pl->pool.release();                 // every device array of the plan
    cached_host_free(pl->h_stop, sizeof(int32_t), hipHostMallocMapped);
    cached_host_free(pl->h_done, sizeof(int32_t), hipHostMallocMapped);
    cached_host_free(pl->h_pipeargs, sizeof(PipeArgs), hipHostMallocDefault);
    for (int i = 0; i < mcsas_plan::RING; ++i) {
        if (pl->evP[i]) hipEventDestroy(pl->evP[i]);
        if (pl->evS[i]) hipEventDestroy(pl->evS[i]);
    }
    if (pl->sP) hipStreamDestroy(pl->sP);
    if (pl->sS) hipStreamDestroy(pl->sS);
    if (pl->ev0) hipEventDestroy(pl->ev0);
    if (pl->ev1) hipEventDestroy(pl->ev1);
    for (int k = 0; k < MCSAS_PLAN_SLOTS; ++k) {          // the parked result slots (the active one's members were freed above)
        if (k == pl->cur_slot || !pl->slots[k].made) continue;
        mcsas_plan::Slot &sl = pl->slots[k];
        cached_host_free(sl.h_done, sizeof(int32_t), hipHostMallocMapped);
        cached_host_free(sl.h_pipeargs, sizeof(PipeArgs), hipHostMallocDefault);
        if (sl.ev0) hipEventDestroy(sl.ev0);
        if (sl.ev1) hipEventDestroy(sl.ev1);
    }
    delete pl;
}

// MCSAS_EXEC_AUTO, rows without an integral: nearest swept shape in (log q slots, log contributions), rates interpolated
// linearly in the repetition count, modes that cannot run the shape left out
static int auto_mode_light(int qpad, int n_contrib, double reps, bool pipe_ok, bool wg_ok) {
    if (reps < (double)auto_reps[0]) return pipe_ok ? MCSAS_EXEC_PIPELINE : (wg_ok ? MCSAS_EXEC_WORKGROUP : MCSAS_EXEC_WAVE);
    int iq = 0, in = 0;
    for (int i = 1; i < AUTO_NQ; ++i) if (std::fabs(std::log((double)qpad / auto_qpad[i])) < std::fabs(std::log((double)qpad / auto_qpad[iq]))) iq = i;
    for (int i = 1; i < AUTO_NN; ++i) if (std::fabs(std::log((double)n_contrib / auto_ncontrib[i])) < std::fabs(std::log((double)n_contrib / auto_ncontrib[in]))) in = i;
    int ir = 0;
    while (ir + 2 < AUTO_NR && reps >= (double)auto_reps[ir + 1]) ++ir;
    double f = (reps - auto_reps[ir]) / (double)(auto_reps[ir + 1] - auto_reps[ir]);
    f = f < 0. ? 0. : (f > 1. ? 1. : f);
    const float (&a)[3] = auto_rate[iq][in][ir], (&b)[3] = auto_rate[iq][in][ir + 1];
    const bool ok[3] = {true, wg_ok, pipe_ok};
    static const int modes[3] = {MCSAS_EXEC_WAVE, MCSAS_EXEC_WORKGROUP, MCSAS_EXEC_PIPELINE};
    int best = 0; double best_rate = -1.;
    for (int m = 0; m < 3; ++m) {
        if (!ok[m] || a[m] <= 0.f || b[m] <= 0.f) continue;
        const double r = a[m] + f * (b[m] - a[m]);
        if (r > best_rate) { best_rate = r; best = m; }
    }
    return modes[best];
}

extern "C" int mcsas_hip_plan_create(const mcsas_problem *p, mcsas_plan **out) {
    if (!p || !out) return fail(MCSAS_EINVAL, "null argument");
    *out = nullptr;
    if (p->struct_size != sizeof(mcsas_problem))
        return fail(MCSAS_EINVAL, "mcsas_problem size %u, library expects %zu (ABI mismatch)", p->struct_size, sizeof(mcsas_problem));
#ifndef MCSAS_TUNING
    if (p->reserved0 != 0) return fail(MCSAS_EINVAL, "mcsas_problem.reserved0 must be 0 (it is %d)", p->reserved0);
#endif
    if (p->nq < 1 || !p->q || !p->intensity || !p->sigma) return fail(MCSAS_EINVAL, "nq/q/intensity/sigma missing");
    if (p->n_contrib < 1 || p->n_reps < 1) return fail(MCSAS_EINVAL, "n_contrib and n_reps must be >= 1");
    if (p->n_active < 1) return fail(MCSAS_EINVAL, "a plan needs an active parameter (mcsas_hip_analyse answers the no-active-parameter case itself)");
    if (p->max_iter < 0 || p->max_retries < 0) return fail(MCSAS_EINVAL, "max_iter/max_retries negative");
    if (p->replay_stream && p->replay_len < 1) return fail(MCSAS_EINVAL, "replay_len must be >= 1");
    ModelArgs margs;
    int rc = fill_model_args(p, &margs);
    if (rc) return rc;
    for (int c = 0; c < p->n_active; ++c)
        if (p->gen_kind[c] < 0 || p->gen_kind[c] > 3) return fail(MCSAS_EINVAL, "gen_kind[%d]=%d", c, p->gen_kind[c]);
    DeviceGuard dev_guard;
    rc = select_device(p->device);
    if (rc) return rc;

    mcsas_plan *pl = new mcsas_plan();
    pl->prob = *p;
    pl->prob.q = pl->prob.intensity = pl->prob.sigma = nullptr;     // host arrays are not retained
    pl->prob.replay_stream = nullptr;
    pl->prob.smear_locs = pl->prob.smear_q_offset = pl->prob.smear_weights = nullptr;
    hipGetDevice(&pl->dev);
#define PCHK(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            int c_ = fail(e_ == hipErrorOutOfMemory ? MCSAS_ENOMEM : MCSAS_EHIP, "%s: %s (%s:%d)", #expr, \
                          hipGetErrorString(e_), __FILE__, __LINE__);                                \
            mcsas_hip_plan_destroy(pl);                                                              \
            return c_;                                                                               \
        }                                                                                            \
    } while (0)

    // q slots per lane: power of two so the kernels are fully unrolled
    int qpl = 1;
    while (qpl * WAVE < p->nq && qpl < 64) qpl *= 2;
    // up to 1024 q-points every execution mode has kernels (q slots per lane 1..16).  More (un-binned data, nBin = 0):
    // one workgroup per chain with the q-points split over up to 8 waves (chain_wide.h; 8 / 16 / 32 slots per lane:
    // up to 4096 / 8192 / 16384 q-points) — MCSAS_EXEC_WORKGROUP, and what MCSAS_EXEC_AUTO picks; up to 4096 q-points
    // MCSAS_EXEC_WAVE still runs one wavefront per chain with 32 / 64 slots per lane
    const bool wide_q = p->nq > 16 * WAVE;
    bool wide = false;
    int wide_waves = 0;
    if (wide_q) {
        const bool wave_asked = p->exec_mode == MCSAS_EXEC_WAVE || (p->exec_mode == MCSAS_EXEC_AUTO && p->waves_per_chain == 1);
        if (p->exec_mode == MCSAS_EXEC_PIPELINE) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "nq %d > 1024: the pipeline has no kernels for it (exec_mode 0, 1 or 2)", p->nq); }
        if (p->nq > WIDE_MAX_WAVES * WAVE * 32) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "nq %d > 16384 is not supported", p->nq); }
        if (wave_asked && p->nq > 64 * WAVE) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "nq %d > 4096 runs one workgroup per chain only (exec_mode 0 or 2)", p->nq); }
        if (!wave_asked) {
            wide = true;
            qpl = p->nq <= 8 * WAVE * WIDE_MAX_WAVES ? 8 : (p->nq <= 16 * WAVE * WIDE_MAX_WAVES ? 16 : 32);
            wide_waves = (p->nq + qpl * WAVE - 1) / (qpl * WAVE);
        }
    }
    const int qpad = wide ? qpl * WAVE * wide_waves : qpl * WAVE;
    rc = pl->smear.upload(p, qpad, &margs);
    if (rc) { mcsas_hip_plan_destroy(pl); return rc; }
    const int tab_shared = table_doubles_host(p->model_id, margs.int_div), tab_row = rowtab_doubles_host(p->model_id, margs.int_div);
    // rows that cost a numerical integration each (2: and whose cost varies with the parameter set — chain_pipe.h, pipe_geometry)
    const int heavy_rows = std::max(model_traits(p->model_id).row_class, margs.smear_nk > 0 ? 1 : 0);
    int n_cus = 256;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, pl->dev) == hipSuccess && v > 0) n_cus = v; }
#ifdef MCSAS_TUNING
    // measurement build (make tuning): mcsas_problem.reserved0 is the tuning / ablation word of the pipeline mode —
    // bits 8-11 rows per producer wave (0 = automatic), 12-15 cap on the scan sub-window in units of 8 steps, 16 every
    // proposal's `new` row stored and row slots swapped on acceptance (no lazy re-evaluation of stale rows), 18 the overlapped
    // producer (Gram MFMAs of sub-window s between the rows of s + 1, operands from HBM/L2), 19-20 row shares of a SIMD's two
    // producer waves, 7 XCD-aware block map; bits 0-6 switch stages OFF and give invalid results
    const int rpw_req = (p->reserved0 >> 8) & 15, eager_req = (p->reserved0 >> 16) & 1, gram_global_req = (p->reserved0 >> 18) & 1,
              sub_req = (p->reserved0 >> 12) & 15;
#else
    const int rpw_req = 0, eager_req = 0, gram_global_req = 0, sub_req = 0;
#endif
#define TABD(waves_per_block) (tab_shared + (waves_per_block) * tab_row)
    // execution mode (results do not depend on it)
    int mode = p->exec_mode;
    int waves = p->waves_per_chain;
    const bool plugin = is_plugin_model(p->model_id);
    if (plugin && wide_q && !wide) {
        mcsas_hip_plan_destroy(pl);
        return fail(MCSAS_EINVAL, "model plug-ins: more than 1024 q-points run one workgroup per chain only (exec_mode 0 or 2; nq %d, exec_mode %d asked)", p->nq, p->exec_mode);
    }
    if (mode == MCSAS_EXEC_AUTO) {
        if (waves == 1) mode = MCSAS_EXEC_WAVE;
        else if (waves > 1) mode = MCSAS_EXEC_WORKGROUP;
        else {
            PipeGeom pg; WgGeom wgm;
            bool pipe_ok = pipe_geometry(p->nq, p->n_contrib, TABD(PIPE_BLOCK / 64), heavy_rows, rpw_req, sub_req, gram_global_req, eager_req, p->n_reps, n_cus, &pg, model_traits(p->model_id).contrib_doubles) == 0;
            bool wg_ok = !wide_q && wg_geometry(p->nq, p->n_contrib, TABD(WG_MAX_WAVES), WG_MAX_WAVES, &wgm) == 0;
            // MCSAS_EXEC_AUTO never picks a mode whose working set does not fit the device: the speculative modes
            // NEED their row cache (N + 2 windows of rows per chain) and the pipeline its window buffers on top (d rows, Gram blocks,
            // scalars, proposals: 2 Kb (qpad + W + 4 + P) doubles per chain) — against half of what is free, the same rule the
            // allocation below applies; one wavefront per chain runs without a cache when it must.  An explicitly requested mode
            // still fails with MCSAS_ENOMEM.
            {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
                    const double Rd = (double)p->n_reps, row = 8.0 * (double)qpad;
                    if (pipe_ok) {
                        const double need = Rd * (((double)p->n_contrib + 2.0 * pg.kb + 1.0) * row
                                                  + 2.0 * pg.kb * (row + 8.0 * (pg.w + 4 + MCSAS_MAX_ACTIVE) + 4.0));
                        if (need >= 0.5 * (double)fr) pipe_ok = false;
                    }
                    if (wg_ok && Rd * ((double)p->n_contrib + 2.0 * wgm.window + 1.0) * row >= 0.5 * (double)fr) wg_ok = false;
                }
            }
            if (!heavy_rows) {
                // rows without an integral: the mode with the highest MEASURED rate at the nearest swept shape
                // (auto_table.h: 128..1024 q slots x 200..1000 contributions x 64..3000 repetitions on a 256-CU MI355X;
                // profiles/r03_mode_sweep*.jsonl).  Fewer than 64 chains: the pipeline wins at every swept shape (1, 10, 50
                // repetitions).  A device with another CU count is asked at the repetition count that loads it the same.
                mode = auto_mode_light(qpad, p->n_contrib, (double)p->n_reps * 256.0 / (double)n_cus, pipe_ok, wg_ok);
            } else {
                // Rows that cost an integral each: the pipeline at every chain count, since its producer waves pull the rows from a
                // queue (round 4, tools/sweep_heavy_modes2.sh, profiles/r04_heavy_modes.txt; steps/s pipeline | one wavefront per chain:
                // cylinders 5.2e6 | 1.3e6 at 300 chains, 6.0e6 | 3.6e6 at 2048, 5.7e6 | 4.4e6 at 4096; Kholodenko 4.4e6 | 0.9e6 at 300,
                // 5.2e6 | 4.8e6 at 2048, 4.7e6 | 4.5e6 at 4096; core-shell ellipsoids 2.5e6 | 0.7e6 at 300 chains x 1000 steps,
                // 3.8e6 | 3.6e6 at 2048 x 3000; the workgroup mode in between at best).  Where the pipeline's geometry does not fit
                // (fewer than 16 contributions), the older rule: one wavefront per chain from where the chains alone fill the
                // SIMDs, a workgroup per chain below.  Units of a 256-CU device.
                const double r_eff = (double)p->n_reps * 256.0 / (double)n_cus;
                if (pipe_ok) mode = MCSAS_EXEC_PIPELINE;
                else if (r_eff >= 1024.) mode = MCSAS_EXEC_WAVE;
                else if (wg_ok) mode = MCSAS_EXEC_WORKGROUP;
                else mode = MCSAS_EXEC_WAVE;
            }
        }
    }
    if (wide) { mode = MCSAS_EXEC_WORKGROUP; waves = wide_waves; }
    else if (wide_q) mode = MCSAS_EXEC_WAVE;
    if (mode == MCSAS_EXEC_WORKGROUP && waves < 2) waves = WG_MAX_WAVES;
    if (mode != MCSAS_EXEC_WORKGROUP) waves = 1;
    pl->wide = wide;
    if (mode < MCSAS_EXEC_WAVE || mode > MCSAS_EXEC_PIPELINE) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "exec_mode %d", mode); }
    pl->qpl = qpl; pl->waves = waves; pl->mode = mode;

    // padded data vectors; sigma == 0 -> 1 (backgroundscalingfit.py:117)
    std::vector<double> hq(qpad), hw(qpad, 0.), hwI(qpad, 0.), hI(qpad, 0.);
    double Sw = 0, SI = 0, SII = 0, Ss2 = 0;
    for (int i = 0; i < qpad; ++i) hq[i] = p->q[i < p->nq ? i : 0];
    for (int i = 0; i < p->nq; ++i) {
        double e = p->sigma[i] == 0.0 ? 1.0 : p->sigma[i];
        double w = 1.0 / (e * e);
        hw[i] = w; hwI[i] = w * p->intensity[i]; hI[i] = p->intensity[i];
        Sw += w; SI += w * p->intensity[i]; SII += w * p->intensity[i] * p->intensity[i]; Ss2 += e * e;
    }
    const size_t vb = sizeof(double) * qpad;
    PCHK(pl->pool.get(&pl->d_q, vb)); PCHK(pl->pool.get(&pl->d_w, vb)); PCHK(pl->pool.get(&pl->d_wI, vb)); PCHK(pl->pool.get(&pl->d_I, vb));
    PCHK(hipMemcpy(pl->d_q, hq.data(), vb, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(pl->d_w, hw.data(), vb, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(pl->d_wI, hwI.data(), vb, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(pl->d_I, hI.data(), vb, hipMemcpyHostToDevice));
    {   // 1 / q^3 (the branch-free sphere kernel's second reciprocal): the same two multiplications and the same correctly
        // rounded division the kernels used to make per block
        std::vector<double> hq3(qpad);
        for (int i = 0; i < qpad; ++i) hq3[i] = 1.0 / (hq[i] * hq[i] * hq[i]);
        PCHK(pl->pool.get(&pl->d_q3inv, vb));
        PCHK(hipMemcpy(pl->d_q3inv, hq3.data(), vb, hipMemcpyHostToDevice));
    }

    const size_t R = p->n_reps, N = p->n_contrib, P = p->n_active;
    PCHK(pl->pool.get(&pl->d_rset, sizeof(double) * R * N * P));
    PCHK(pl->pool.get(&pl->d_fit, sizeof(double) * R * qpad));
    PCHK(pl->pool.get(&pl->d_out, sizeof(ChainOut) * R));
    PCHK(hipMemset(pl->d_out, 0, sizeof(ChainOut) * R));

    // per-contribution intensity rows: the speculative kernels add two windows of spare row slots
    int cache_rows = (int)N;
    if (wide) {
        // q-split workgroup: no speculation, rows [N][qpad] like the wave kernel
    } else if (mode == MCSAS_EXEC_WORKGROUP) {
        int rcg = wg_geometry(p->nq, (int)N, TABD(waves), waves, &pl->wg);
        if (rcg) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "workgroup kernel: needs 2*window <= n_contrib and the window in LDS (nq=%d, n_contrib=%d, waves=%d)", p->nq, (int)N, waves); }
        cache_rows = (int)N + 2 * pl->wg.window;
    } else if (mode == MCSAS_EXEC_PIPELINE) {
        if (const int rcg = pipe_geometry(p->nq, (int)N, TABD(PIPE_BLOCK / 64), heavy_rows, rpw_req, sub_req, gram_global_req, eager_req, p->n_reps, n_cus, &pl->pipe.g, model_traits(p->model_id).contrib_doubles)) {
            mcsas_hip_plan_destroy(pl);
            if (rcg == 2) return fail(MCSAS_EINVAL, "pipeline: the window's row buffers / proposal records do not fit the 160 KB of LDS (nq=%d, n_contrib=%d)", p->nq, (int)N);
            return fail(MCSAS_EINVAL, "pipeline: needs n_contrib >= 16 and nq <= 1024 (nq=%d, n_contrib=%d)", p->nq, (int)N);
        }
        cache_rows = (int)N + 2 * pl->pipe.g.kb;
    }
    // The chains' row blocks must not all start at the same offset modulo the memory system's interleave: with N = 400 rows of
    // 4 KB a chain's block is 50 x 32 KB, every chain's row r sits at the same place in the channel pattern, and thousands of
    // chains walking their rows in step queue up on the same channels (measured: identical launches of 8192 chains at 3.7e8 or
    // 7.5e8 steps/s depending on whether the chains happened to run in step).  Spare rows make the block an odd number of rows.
    {
        int pad_rows = (cache_rows & 1) ? 0 : 1;
        if (const char *e = getenv("MCSAS_HIP_CACHE_PAD_ROWS")) pad_rows = atoi(e) < 0 ? 0 : atoi(e);     // (measurement knob)
        cache_rows += pad_rows;
    }
    size_t cache_bytes = sizeof(double) * R * (size_t)cache_rows * qpad;
    int use_cache = p->cache_intensities;
    if (use_cache < 0 || mode != MCSAS_EXEC_WAVE || wide_q) {
        size_t fr = 0, tot = 0;
        PCHK(hipMemGetInfo(&fr, &tot));
        use_cache = cache_bytes < fr / 2;
        if ((mode != MCSAS_EXEC_WAVE || wide_q) && !use_cache) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_ENOMEM, "intensity cache (%zu MB) does not fit", cache_bytes >> 20); }
    }
    pl->use_cache = use_cache;
    if (use_cache) PCHK(pl->pool.get(&pl->d_cache, cache_bytes));

    if (p->replay_stream) {
        size_t rb = sizeof(double) * R * (size_t)p->replay_len;
        PCHK(pl->pool.get(&pl->d_replay, rb));
        PCHK(hipMemcpy(pl->d_replay, p->replay_stream, rb, hipMemcpyHostToDevice));
    }
    PCHK(cached_host_malloc((void **)&pl->h_stop, sizeof(int32_t), hipHostMallocMapped));
    *pl->h_stop = 0;
    int32_t *d_stop = nullptr;
    PCHK(hipHostGetDevicePointer((void **)&d_stop, pl->h_stop, 0));
    PCHK(hipEventCreate(&pl->ev0)); PCHK(hipEventCreate(&pl->ev1));

    ChainArgs &a = pl->args;
    memset(&a, 0, sizeof a);
    a.model = margs;
    a.nq = p->nq; a.qpad = qpad;
    a.q = pl->d_q; a.w = pl->d_w; a.wI = pl->d_wI; a.I = pl->d_I;
    a.Sw = Sw; a.SI = SI; a.SII = SII; a.Ssig2 = Ss2;
    a.n_contrib = p->n_contrib; a.n_reps = p->n_reps;
    a.find_bg = p->find_background != 0; a.pos_bg = p->positive_background != 0;
    a.start_from_min = p->start_from_minimum != 0; a.max_retries = p->max_retries;
    a.max_iter = p->max_iter; a.conv_crit = p->conv_crit;
    for (int c = 0; c < MCSAS_MAX_ACTIVE; ++c) {
        a.gen_lo[c] = p->gen_lo[c]; a.gen_hi[c] = p->gen_hi[c]; a.start_value[c] = p->start_value[c];
        a.gen_kind[c] = p->gen_kind[c];
    }
    a.seed = p->seed; a.rep_offset = p->rep_offset; a.pad0 = p->reserved0;   // (0 in the release library: checked above)
    a.replay = pl->d_replay; a.replay_len = p->replay_len;
    a.stop_flag = d_stop;
    PCHK(pl->pool.get(&pl->d_stop_relay, 16));
    a.stop_relay = pl->d_stop_relay;
    a.rset = pl->d_rset; a.cache = pl->d_cache; a.cache_rows = cache_rows; a.fit = pl->d_fit; a.out = pl->d_out;

    if (mode == MCSAS_EXEC_WAVE) {
        pl->lds_bytes = sizeof(double) * (4 * (size_t)qpad + TABD(1));
        if (plugin) {
            rc = plugin_wave_function(p->model_id, qpl, use_cache != 0, &pl->plugin_fn);     // (compiled on first use of this q count)
            if (rc) { mcsas_hip_plan_destroy(pl); return rc; }
        } else if (!wave_kernel_for(p->model_id, qpl, use_cache)) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "no kernel for model %d qpl %d", p->model_id, qpl); }
    } else if (wide) {
        // partial sums, the model's tables and one row scratch per wave; q and 1/q^3 as well when they fit (ChainArgs::pad1)
        const size_t base = sizeof(double) * ((size_t)WIDE_PART_DOUBLES + TABD(waves));
        const size_t with_q = base + sizeof(double) * 2 * (size_t)qpad;
        a.pad1 = with_q <= 150 * 1024 ? 1 : 0;
        pl->lds_bytes = a.pad1 ? with_q : base;
        if (plugin) {
            rc = plugin_wide_function(p->model_id, qpl, &pl->plugin_fn);
            if (rc) { mcsas_hip_plan_destroy(pl); return rc; }
        } else if (!wide_kernel_for(p->model_id, qpl)) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "no q-split kernel for model %d qpl %d", p->model_id, qpl); }
    } else if (mode == MCSAS_EXEC_WORKGROUP) {
        pl->lds_bytes = pl->wg.lds_bytes;
        if (plugin) {
            rc = plugin_wg_function(p->model_id, qpl, &pl->plugin_fn);
            if (rc) { mcsas_hip_plan_destroy(pl); return rc; }
        }
    } else {
        if (plugin) {
            rc = plugin_pipe_function(p->model_id, qpl, pl->pipe.g.rowq != 0, &pl->plugin_fn);
            if (rc) { mcsas_hip_plan_destroy(pl); return rc; }
        }
        PipeArgs &pa = pl->pipe;
        const size_t Kb = pa.g.kb;
        PCHK(pl->pool.get(&pl->d_chains, sizeof(PipeChain) * R));
        PCHK(pl->pool.get(&pl->d_pipeargs, sizeof(PipeArgs)));
        PCHK(pl->pool.get(&pl->d_ft, sizeof(double) * R * qpad)); PCHK(pl->pool.get(&pl->d_wft, sizeof(double) * R * qpad));
        PCHK(pl->pool.get(&pl->d_slot_of, sizeof(int32_t) * R * N)); PCHK(pl->pool.get(&pl->d_stage, sizeof(int32_t) * R * 2 * Kb));
        PCHK(pl->pool.get(&pl->d_dwin, sizeof(double) * R * 2 * Kb * qpad));
        PCHK(pl->pool.get(&pl->d_gwin, sizeof(double) * R * 2 * Kb * pa.g.w));
        PCHK(pl->pool.get(&pl->d_scal, sizeof(double) * R * 2 * Kb * 4));
        PCHK(pl->pool.get(&pl->d_row_valid, sizeof(int32_t) * R * N));
        PCHK(hipMemset(pl->d_row_valid, 0, sizeof(int32_t) * R * N));
        PCHK(pl->pool.get(&pl->d_rowq, sizeof(int32_t) * R * 2));
        PCHK(hipMemset(pl->d_rowq, 0, sizeof(int32_t) * R * 2));
        PCHK(pl->pool.get(&pl->d_pval, sizeof(double) * R * 2 * Kb * MCSAS_MAX_ACTIVE));
        PCHK(pl->pool.get(&pl->d_povf, sizeof(int32_t) * R * 2 * Kb));
        PCHK(hipMemset(pl->d_povf, 0, sizeof(int32_t) * R * 2 * Kb));
        PCHK(cached_host_malloc((void **)&pl->h_done, sizeof(int32_t), hipHostMallocMapped));
        *pl->h_done = 0;
        int32_t *d_done = nullptr;
        PCHK(hipHostGetDevicePointer((void **)&d_done, pl->h_done, 0));
        for (int i = 0; i < mcsas_plan::RING; ++i)
            PCHK(hipEventCreateWithFlags(&pl->evS[i], hipEventDisableTiming));
        pa.c = a;
        pa.chains = pl->d_chains; pa.ft = pl->d_ft; pa.wft = pl->d_wft; pa.slot_of = pl->d_slot_of;
        pa.stage_slot = pl->d_stage; pa.dwin = pl->d_dwin; pa.gwin = pl->d_gwin; pa.scal = pl->d_scal; pa.row_valid = pl->d_row_valid; pa.pval = pl->d_pval; pa.rowq = pl->d_rowq;
        PCHK(pl->pool.get(&pl->d_done_dev, sizeof(int32_t)));
        pa.povf = pl->d_povf; pa.n_done = d_done; pa.n_done_dev = pl->d_done_dev; pa.tick = 0;
        pa.timeline = nullptr; pa.timeline_tick = -100;
#ifdef MCSAS_STAMPS
        if (const char *e = getenv("MCSAS_TIMELINE_TICK")) {
            pa.timeline_tick = atoi(e);
            PCHK(pl->pool.get(&pl->d_timeline, sizeof(uint64_t) * 8 * PIPE_TL_WORDS * (R + R * pa.g.prod_blocks_y)));
            PCHK(hipMemset(pl->d_timeline, 0, sizeof(uint64_t) * 8 * PIPE_TL_WORDS * (R + R * pa.g.prod_blocks_y)));
            pa.timeline = pl->d_timeline;
        }
#endif
        pl->lds_bytes = std::max(pa.g.prod_lds, pa.g.scan_lds);
    }
    if (pl->lds_bytes > 160 * 1024) { mcsas_hip_plan_destroy(pl); return fail(MCSAS_EINVAL, "LDS need %zu B > 160 KiB", pl->lds_bytes); }
    *out = pl;
    return MCSAS_OK;
#undef PCHK
#undef TABD
}

extern "C" int mcsas_hip_plan_reseed(mcsas_plan *pl, uint64_t seed, int32_t rep_offset) {
    if (!pl) return fail(MCSAS_EINVAL, "null plan");
    pl->args.seed = seed; pl->args.rep_offset = rep_offset;
    return MCSAS_OK;
}

static int pipeline_launch(mcsas_plan *pl, hipStream_t st) {
    PipeArgs &pa = pl->pipe;
    pa.c = pl->args;                                     // picks up reseed()
    pa.c.cache_rows = pl->args.cache_rows;
    const int R = pl->prob.n_reps, Kb = pa.g.kb;
    void *tick = pl->plugin_fn ? nullptr : pipe_tick_kernel_for(pl->prob.model_id, pl->qpl, pa.g.rowq != 0), *reset = mcsas_pipe_reset_kernel();
    if (!tick && !pl->plugin_fn) return fail(MCSAS_EINVAL, "no pipeline kernel for model %d qpl %d", pl->prob.model_id, pl->qpl);
    const size_t lds = std::max(pa.g.prod_lds, pa.g.scan_lds);
    if (tick && lds > 64 * 1024) HIPCHK(hipFuncSetAttribute(tick, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    *pl->h_done = 0;
    HIPCHK(hipMemsetAsync(pl->d_done_dev, 0, sizeof(int32_t), st));
    pa.tick = 0;
    if (!pl->h_pipeargs) HIPCHK(cached_host_malloc((void **)&pl->h_pipeargs, sizeof(PipeArgs), hipHostMallocDefault));
    *pl->h_pipeargs = pa;                                // (the previous launch's upload has completed: fetch() or the caller waited for it)
    HIPCHK(hipMemcpyAsync(pl->d_pipeargs, pl->h_pipeargs, sizeof(PipeArgs), hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(pl->ev0, st));
    {
        void *ka[] = {(void *)&pl->d_pipeargs};
        HIPCHK(hipLaunchKernel(reset, dim3((R + 63) / 64), dim3(64), ka, 0, st));
    }
    // worst case: every attempt runs to max_iter
    // (saturating: max_iter may be 2^62 and a product of two such numbers must not wrap to a negative budget)
    const long long TICK_CAP = 2000000000LL;
    const long long win_per_attempt = (long long)(pl->prob.max_iter / Kb) + 4;
    const long long attempts = (long long)pl->prob.max_retries + 1;
    // one attempt: its last window (index ceil(max_iter / Kb) - 1) is scanned at tick ceil(max_iter / Kb), exactly; with
    // retries the schedule depends on when attempts end, so the bound is generous and the host leaves when all chains are done
    const long long one_attempt = std::min((long long)((pl->prob.max_iter + Kb - 1) / Kb) + 1, TICK_CAP);
    const long long max_ticks = attempts == 1 ? one_attempt
                              : ((win_per_attempt >= TICK_CAP / attempts) ? TICK_CAP : std::min(attempts * win_per_attempt + 4, TICK_CAP));
    // scan blocks + producer blocks (chain-major; 8 XCD classes of ceil(R/8) chains each with diagnostic bit 128)
    const dim3 grid((MCSAS_TUNE_BITS(pl->args) & 128) ? R + 8 * ((R + 7) / 8) * pa.g.prod_blocks_y : R + R * pa.g.prod_blocks_y);
    PipeHot hot{};
    hot.q = pa.c.q; hot.w = pa.c.w; hot.wI = pa.c.wI; hot.chains = pa.chains;
    hot.n_reps = pa.c.n_reps; hot.n_contrib = pa.c.n_contrib; hot.n_active = pa.c.model.n_active; hot.qpad = pa.c.qpad;
    hot.kb = pa.g.kb; hot.prod_blocks_y = pa.g.prod_blocks_y; hot.w_sub = pa.g.w; hot.max_iter = pa.c.max_iter;
    hot.q3inv = pl->d_q3inv;
    long long t = -1;                                    // launch t = {SCAN(t), PROD(t+1)}
    for (; t < max_ticks; ++t) {
        if (t >= mcsas_plan::RING && (t % 16) == 0) {
            // throttle: stay at most ~RING ticks ahead of the GPU, forward the stop word, leave when all chains are done
            HIPCHK(hipEventSynchronize(pl->evS[((t - mcsas_plan::RING) / 16) % mcsas_plan::RING]));
            if (pl->prob.stop && *pl->prob.stop) *pl->h_stop = 1;
            if (*(volatile int32_t *)pl->h_done >= R) break;
        }
        int32_t tk = (int32_t)t, stop_now = (pl->prob.stop && *(volatile int32_t *)pl->prob.stop) ? 1 : 0;
        void *ka[] = {(void *)&pl->d_pipeargs, (void *)&tk, (void *)&stop_now, (void *)&hot};
        if (tick) HIPCHK(hipLaunchKernel(tick, grid, dim3(PIPE_BLOCK), ka, lds, st));
        else HIPCHK(hipModuleLaunchKernel(pl->plugin_fn, grid.x, 1, 1, PIPE_BLOCK, 1, 1, (unsigned)lds, st, ka, nullptr));
        if (t >= 0 && (t % 16) == 0) HIPCHK(hipEventRecord(pl->evS[(t / 16) % mcsas_plan::RING], st));
    }
    pl->ticks_launched = (int)t;
    HIPCHK(hipEventRecord(pl->ev1, st));
    return MCSAS_OK;
}

extern "C" int mcsas_hip_plan_launch_slot(mcsas_plan *pl, void *hip_stream, int32_t slot) {
    if (!pl) return fail(MCSAS_EINVAL, "null plan");
    DeviceGuard dev_guard;
    HIPCHK(hipSetDevice(pl->dev));
    {
        int rcs = plan_activate_slot(pl, slot);
        if (rcs) return rcs;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // The slots of a plan share its workspaces (row cache, chain records, window buffers, argument block): an analysis must not start
    // before the other slots' analyses have ended.  On one stream that is the stream's order; launched on ANOTHER stream this one waits
    // for their end events (a no-op when they are complete).  This slot's own previous analysis must be over before its pinned
    // argument block is rewritten.
    for (int k = 0; k < MCSAS_PLAN_SLOTS; ++k)
        if (k != pl->cur_slot && pl->slots[k].made && pl->slots[k].launched && pl->slots[k].ev1) HIPCHK(hipStreamWaitEvent(st, pl->slots[k].ev1, 0));
    if (pl->launched && pl->ev1) HIPCHK(hipEventSynchronize(pl->ev1));
    *pl->h_stop = (pl->prob.stop && *pl->prob.stop) ? 1 : 0;
    pl->stream = st;
    if (pl->mode != MCSAS_EXEC_PIPELINE) HIPCHK(hipMemsetAsync(pl->d_stop_relay, 0, 16, st));   // (the pipeline's ticks get McSAS.stop as a kernel argument)
    struct FailGuard {                                    // (any error return below leaves work on `st` that no end event covers)
        mcsas_plan *pl; bool ok = false;
        ~FailGuard() { if (!ok) pl->enqueue_failed = true; }
    } guard{pl};
    if (pl->mode == MCSAS_EXEC_PIPELINE) {
        int rc = pipeline_launch(pl, st);
        if (rc) return rc;
        pl->launched = true; guard.ok = true;
        return MCSAS_OK;
    }
    void *kargs[] = {(void *)&pl->args};
    void *fn;
    dim3 grid(pl->prob.n_reps), block;
    if (pl->plugin_fn) {                                  // a run-time model plug-in: the same arguments, through its code-object module
        void *kargs2[] = {(void *)&pl->args, pl->wide ? (void *)&pl->d_q3inv : (void *)&pl->wg};
        const bool wgm = pl->mode == MCSAS_EXEC_WORKGROUP;
        HIPCHK(hipEventRecord(pl->ev0, st));
        HIPCHK(hipModuleLaunchKernel(pl->plugin_fn, grid.x, 1, 1, wgm ? WAVE * pl->waves : WAVE, 1, 1, (unsigned)pl->lds_bytes, st,
                                     wgm ? kargs2 : kargs, nullptr));
        HIPCHK(hipEventRecord(pl->ev1, st));
        pl->launched = true; guard.ok = true;
        return MCSAS_OK;
    }
    if (pl->mode == MCSAS_EXEC_WAVE) {
        fn = wave_kernel_for(pl->prob.model_id, pl->qpl, pl->use_cache);
        block = dim3(WAVE);
    } else if (pl->wide) {
        fn = wide_kernel_for(pl->prob.model_id, pl->qpl);
        block = dim3(WAVE * pl->waves);
    } else {
        fn = wg_kernel_for(pl->prob.model_id, pl->qpl);
        block = dim3(WAVE * pl->waves);
    }
    if (!fn) return fail(MCSAS_EINVAL, "no kernel for model %d qpl %d", pl->prob.model_id, pl->qpl);
    if (pl->lds_bytes > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    HIPCHK(hipEventRecord(pl->ev0, st));
    if (pl->mode == MCSAS_EXEC_WAVE) {
        HIPCHK(hipLaunchKernel(fn, grid, block, kargs, pl->lds_bytes, st));
    } else if (pl->wide) {
        void *kargs3[] = {(void *)&pl->args, (void *)&pl->d_q3inv};
        HIPCHK(hipLaunchKernel(fn, grid, block, kargs3, pl->lds_bytes, st));
    } else {
        void *kargs2[] = {(void *)&pl->args, (void *)&pl->wg};
        HIPCHK(hipLaunchKernel(fn, grid, block, kargs2, pl->lds_bytes, st));
    }
    HIPCHK(hipEventRecord(pl->ev1, st));
    pl->launched = true; guard.ok = true;
    return MCSAS_OK;
}

extern "C" int mcsas_hip_plan_launch(mcsas_plan *pl, void *hip_stream) { return mcsas_hip_plan_launch_slot(pl, hip_stream, 0); }

extern "C" int mcsas_hip_plan_fetch_slot(mcsas_plan *pl, int32_t slot, mcsas_result *res) {
    if (!pl) return fail(MCSAS_EINVAL, "null plan");
    {
        DeviceGuard dg;
        HIPCHK(hipSetDevice(pl->dev));
        int rcs = plan_activate_slot(pl, slot);
        if (rcs) return rcs;
    }
    if (!pl->launched) return fail(MCSAS_EINVAL, "plan was not launched");
    if (res && res->struct_size != sizeof(mcsas_result))
        return fail(MCSAS_EINVAL, "mcsas_result size %u, library expects %zu", res->struct_size, sizeof(mcsas_result));
    DeviceGuard dev_guard;
    HIPCHK(hipSetDevice(pl->dev));
    // wait, forwarding the caller's stop word to the device-visible one (McSAS.stop, mcsas.py:357)
    while (pl->prob.stop) {                              // (no stop word: nothing to forward, plain wait below)
        hipError_t q = hipEventQuery(pl->ev1);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(MCSAS_EHIP, "kernel failed: %s", hipGetErrorString(q));
        if (*pl->prob.stop) *pl->h_stop = 1;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    HIPCHK(hipEventSynchronize(pl->ev1));
    if (!pl->sCopy) HIPCHK(cached_copy_stream(&pl->sCopy));      // (one per device, shared by every plan)
    // device -> host on the copy stream (the data is complete: ev1 has passed)
    auto d2h = [&](void *dst, const void *src, size_t n) -> hipError_t {
        hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, pl->sCopy);
        return e != hipSuccess ? e : hipStreamSynchronize(pl->sCopy);
    };
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
    pl->last_ms = ms;
    const size_t R = pl->prob.n_reps, N = pl->prob.n_contrib, P = pl->prob.n_active, Q = pl->prob.nq, qpad = pl->args.qpad;
    if (pl->mode == MCSAS_EXEC_PIPELINE && *(volatile int32_t *)pl->h_done < (int32_t)R)
        return fail(MCSAS_EHIP, "pipeline: %d of %zu chains finished within the %d ticks that were launched",
                    (int)*(volatile int32_t *)pl->h_done, R, pl->ticks_launched + 1);
    std::vector<ChainOut> ho(R);
    HIPCHK(d2h(ho.data(), pl->d_out, sizeof(ChainOut) * R));
    int64_t steps = 0; int ovf = 0;
    for (size_t r = 0; r < R; ++r) { steps += ho[r].total_steps; ovf |= ho[r].stream_overflow; }
    pl->last_steps = steps;
#ifdef MCSAS_STAMPS
    {
        static const char *names[20] = {"rows+dot+reduce", "B1 wait", "decide", "B2 wait", "apply", "sub-windows", "accepted", "-",
                                        "prod rows", "prod barrier", "prod gram", "prod blocks",
                                        "prologue", "epilogue", "window ticks", "block total", "gap between blocks (10ns)", "gaps", "in block (10ns)", "-"};
        for (size_t r = 0; r < R && r < 2; ++r) {
            fprintf(stderr, "[mcsas stamps] rep %zu (wave 0 cycles):", r);
            for (int i = 0; i < 20; ++i) fprintf(stderr, " %s=%lld", names[i], (long long)ho[r].dbg[i]);
            fprintf(stderr, "\n");
        }
        if (pl->d_timeline) {
            const size_t nb = R + R * pl->pipe.g.prod_blocks_y;
            std::vector<uint64_t> tl(nb * 8 * PIPE_TL_WORDS);
            HIPCHK(d2h(tl.data(), pl->d_timeline, tl.size() * 8));
            uint64_t t0 = ~0ull;
            const size_t TW = PIPE_TL_WORDS;
            for (size_t i = 0; i < nb * 8; ++i) if (tl[i * TW] && tl[i * TW] < t0) t0 = tl[i * TW];
            fprintf(stderr, "[mcsas timeline] tick %d: block wave start_us end_us hw_id xcc\n", pl->pipe.timeline_tick);
            for (size_t i = 0; i < nb * 8; ++i)
                if (tl[i * TW]) {
                    fprintf(stderr, "[mcsas timeline] %zu %zu %.2f %.2f %llx %llu", i / 8, i % 8, (tl[i * TW] - t0) * 0.01, (tl[i * TW + 1] - t0) * 0.01,
                            (unsigned long long)tl[i * TW + 2], (unsigned long long)tl[i * TW + 3]);
                    for (size_t m = 4; m < TW; ++m) fprintf(stderr, " %.2f", tl[i * TW + m] ? (tl[i * TW + m] - t0) * 0.01 : 0.0);
                    fprintf(stderr, "\n");
                }
        }
        fprintf(stderr, "[mcsas stamps] per rep: in-block us / gap us:");
        for (size_t r = 0; r < R; ++r) fprintf(stderr, " %.0f/%.0f", ho[r].dbg[18] * 0.01, ho[r].dbg[16] * 0.01);
        fprintf(stderr, "\n");
    }
#endif
    if (res) {
        if (res->contribs) {
            std::vector<double> hr(R * N * P);
            HIPCHK(d2h(hr.data(), pl->d_rset, sizeof(double) * hr.size()));
            for (size_t r = 0; r < R; ++r)
                for (size_t n = 0; n < N; ++n)
                    for (size_t p = 0; p < P; ++p) res->contribs[(n * P + p) * R + r] = hr[(r * N + n) * P + p];
        }
        if (res->fit) {
            std::vector<double> hf(R * qpad);
            HIPCHK(d2h(hf.data(), pl->d_fit, sizeof(double) * hf.size()));
            for (size_t r = 0; r < R; ++r)
                for (size_t k = 0; k < Q; ++k) res->fit[k * R + r] = hf[r * qpad + k];
        }
        for (size_t r = 0; r < R; ++r) {
            if (res->chisq) res->chisq[r] = ho[r].chisq;
            if (res->scaling) res->scaling[r] = ho[r].scaling;
            if (res->background) res->background[r] = ho[r].background;
            if (res->num_iter) res->num_iter[r] = ho[r].num_iter;
            if (res->num_moves) res->num_moves[r] = ho[r].num_moves;
            if (res->attempts) res->attempts[r] = ho[r].attempts;
            if (res->converged) res->converged[r] = ho[r].converged;
            if (res->seconds) res->seconds[r] = ho[r].seconds;
            if (res->draws) res->draws[r] = ho[r].draws;
        }
    }
    if (ovf) return fail(MCSAS_ESTREAM, "replay stream exhausted (replay_len=%lld)", (long long)pl->prob.replay_len);
    return MCSAS_OK;
}

extern "C" int mcsas_hip_plan_fetch(mcsas_plan *pl, mcsas_result *res) { return mcsas_hip_plan_fetch_slot(pl, 0, res); }

extern "C" int mcsas_hip_plan_info(mcsas_plan *pl, int32_t info[8]) {
    if (!pl || !info) return fail(MCSAS_EINVAL, "null argument");
    memset(info, 0, sizeof(int32_t) * 8);
    info[0] = pl->mode; info[1] = pl->waves; info[2] = pl->qpl;
    info[3] = pl->mode == MCSAS_EXEC_PIPELINE ? pl->pipe.g.kb : (pl->mode == MCSAS_EXEC_WORKGROUP && !pl->wide ? pl->wg.window : 1);
    info[4] = pl->mode == MCSAS_EXEC_PIPELINE ? pl->ticks_launched + 2 : 1;
    info[5] = pl->use_cache;
    return MCSAS_OK;
}

extern "C" int mcsas_hip_plan_last_ms(mcsas_plan *pl, double *ms) {
    if (!pl || !ms) return fail(MCSAS_EINVAL, "null argument");
    *ms = pl->last_ms;
    return MCSAS_OK;
}
extern "C" int mcsas_hip_plan_total_steps(mcsas_plan *pl, int64_t *steps) {
    if (!pl || !steps) return fail(MCSAS_EINVAL, "null argument");
    *steps = pl->last_steps;
    return MCSAS_OK;
}

static int analyse_one(const mcsas_problem *p, mcsas_result *res) {
    mcsas_plan *pl = nullptr;
    int rc = mcsas_hip_plan_create(p, &pl);
    if (rc) return rc;
    rc = mcsas_hip_plan_launch(pl, nullptr);
    if (!rc) rc = mcsas_hip_plan_fetch(pl, res);
    mcsas_hip_plan_destroy(pl);
    return rc;
}

// repetitions [first, first + count) of device-list entry `index` (contiguous blocks, sizes differ by at most one)
extern "C" int mcsas_hip_shard(int32_t n_reps, int32_t n_devices, int32_t index, int32_t *first, int32_t *count) {
    if (n_reps < 0 || n_devices < 1 || index < 0 || index >= n_devices || !first || !count) return fail(MCSAS_EINVAL, "bad argument");
    const int32_t base = n_reps / n_devices, extra = n_reps % n_devices;
    *count = base + (index < extra ? 1 : 0);
    *first = index * base + (index < extra ? index : extra);
    return MCSAS_OK;
}

// McSAS.analyse's repetition loop (mcsas.py:214-262) over several GPUs: one host thread, one plan and one stream per
// device, each running a contiguous block of repetitions under its global chain ids; every block lands in the caller's
// (…, numReps) arrays at its place.  No data-path exchange between the devices: the blocks share read-only inputs only.
static int analyse_sharded(const mcsas_problem *p, mcsas_result *res) {
    if (res->struct_size != sizeof(mcsas_result)) return fail(MCSAS_EINVAL, "mcsas_result size mismatch");
    const int G = p->n_devices;
    if (G > MCSAS_MAX_DEVICES) return fail(MCSAS_EINVAL, "n_devices %d > %d", G, MCSAS_MAX_DEVICES);
    if (p->n_reps < 1 || p->n_contrib < 1 || p->nq < 1) return fail(MCSAS_EINVAL, "n_reps, n_contrib and nq must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MCSAS_ENODEV, "no HIP device available");
    for (int g = 0; g < G; ++g)
        if (p->devices[g] < 0 || p->devices[g] >= ndev) return fail(MCSAS_ENODEV, "devices[%d] = %d, %d present", g, p->devices[g], ndev);
    const size_t R = p->n_reps, N = p->n_contrib, P = p->n_active, Q = p->nq;
    struct Shard {
        mcsas_problem prob; mcsas_result out;
        int32_t first = 0, count = 0; int rc = 0; std::string err;
        std::vector<double> contribs, fit, chisq, scaling, background, seconds;
        std::vector<int64_t> num_iter, num_moves, draws;
        std::vector<int32_t> attempts, converged;
    };
    std::vector<Shard> sh(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        Shard &s = sh[g];
        mcsas_hip_shard((int32_t)R, G, g, &s.first, &s.count);
        if (s.count == 0) continue;
        s.prob = *p;
        s.prob.n_devices = 0; s.prob.device = p->devices[g];
        s.prob.n_reps = s.count; s.prob.rep_offset = p->rep_offset + s.first;
        if (p->replay_stream) s.prob.replay_stream = p->replay_stream + (size_t)s.first * p->replay_len;
        const size_t Rg = s.count;
        memset(&s.out, 0, sizeof s.out);
        s.out.struct_size = sizeof(mcsas_result);
#define SHBUF(f) do { if (res->f) { s.f.resize(Rg); s.out.f = s.f.data(); } } while (0)
        if (res->contribs) { s.contribs.resize(N * P * Rg); s.out.contribs = s.contribs.data(); }
        if (res->fit) { s.fit.resize(Q * Rg); s.out.fit = s.fit.data(); }
        SHBUF(chisq); SHBUF(scaling); SHBUF(background); SHBUF(seconds); SHBUF(num_iter); SHBUF(num_moves); SHBUF(draws);
        SHBUF(attempts); SHBUF(converged);
#undef SHBUF
        th.emplace_back([&s]() {
            s.rc = analyse_one(&s.prob, &s.out);
            if (s.rc) s.err = g_err;                          // (thread-local in the worker)
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G; ++g)
        if (sh[g].rc) return fail(sh[g].rc, "device %d (repetitions %d..%d): %s", p->devices[g], sh[g].first, sh[g].first + sh[g].count - 1, sh[g].err.c_str());
    for (int g = 0; g < G; ++g) {
        const Shard &s = sh[g];
        const size_t Rg = s.count, f0 = s.first;
        if (res->contribs) for (size_t i = 0; i < N * P; ++i) for (size_t r = 0; r < Rg; ++r) res->contribs[i * R + f0 + r] = s.contribs[i * Rg + r];
        if (res->fit) for (size_t k = 0; k < Q; ++k) for (size_t r = 0; r < Rg; ++r) res->fit[k * R + f0 + r] = s.fit[k * Rg + r];
        for (size_t r = 0; r < Rg; ++r) {
#define SHCOPY(f) do { if (res->f) res->f[f0 + r] = s.f[r]; } while (0)
            SHCOPY(chisq); SHCOPY(scaling); SHCOPY(background); SHCOPY(seconds); SHCOPY(num_iter); SHCOPY(num_moves); SHCOPY(draws);
            SHCOPY(attempts); SHCOPY(converged);
#undef SHCOPY
        }
    }
    return MCSAS_OK;
}

extern "C" int mcsas_hip_analyse(const mcsas_problem *p, mcsas_result *res) {
    if (!res) return fail(MCSAS_EINVAL, "null result");
    if (p && p->n_active == 0) {
        // No active fit parameter (mcsas.py:198-201, 238-239, 322-323): analyse() runs ONE repetition of ONE
        // contribution, mcFit returns the model's intensity at its fixed parameter values with conval = -1,
        // scaling 1, background 0, and the repetition loop leaves without retrying.  Result arrays are sized
        // for n_contrib = n_reps = 1 whatever the problem says.
        if (res->struct_size != sizeof(mcsas_result)) return fail(MCSAS_EINVAL, "mcsas_result size mismatch");
        if (p->struct_size != sizeof(mcsas_problem)) return fail(MCSAS_EINVAL, "mcsas_problem size mismatch (ABI)");
        if (p->nq < 1 || !p->q) return fail(MCSAS_EINVAL, "nq/q missing");
        std::vector<double> cum(p->nq);
        int rc0 = mcsas_hip_model_calc(p, nullptr, 1, cum.data(), nullptr, nullptr, nullptr, nullptr);
        if (rc0) return rc0;
        if (res->fit) for (int k = 0; k < p->nq; ++k) res->fit[k] = cum[k];
        if (res->chisq) res->chisq[0] = -1.0;
        if (res->scaling) res->scaling[0] = 1.0;
        if (res->background) res->background[0] = 0.0;
        if (res->num_iter) res->num_iter[0] = 0;
        if (res->num_moves) res->num_moves[0] = 0;
        if (res->attempts) res->attempts[0] = 1;
        if (res->converged) res->converged[0] = 1;
        if (res->seconds) res->seconds[0] = 0.0;
        if (res->draws) res->draws[0] = 0;
        return MCSAS_OK;
    }
    if (p && p->struct_size == sizeof(mcsas_problem) && p->n_devices > 1) return analyse_sharded(p, res);
    return analyse_one(p, res);
}

// ------------------------------------------------------------------------------ model.calc
// scratch arrays of the one-call entry points below: from / back to the process-level cache (a histogram() makes a dozen of them)
template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    ~DevBuf() {
        if (!p) return;
        (void)hipStreamSynchronize(nullptr);       // (hipFree used to wait for the kernels that read it — they run on the null stream —; the cache does not)
        cached_dev_free(p, bytes, dev);
    }
    hipError_t alloc(size_t n) {
        bytes = (sizeof(T) * (n ? n : 1) + 255) / 256 * 256;
        (void)hipGetDevice(&dev);
        return cached_dev_malloc((void **)&p, bytes);
    }
};

extern "C" int mcsas_hip_model_calc(const mcsas_problem *p, const double *pset, int32_t n, double *cum_int,
                                    double *vset, double *wset, double *sset, double *rows) {
    if (!p || n < 1 || !p->q || p->nq < 1 || (!pset && p->n_active > 0)) return fail(MCSAS_EINVAL, "bad argument");
    ModelArgs m;
    int rc = fill_model_args(p, &m);
    if (rc) return rc;
    DeviceGuard dev_guard;
    rc = select_device(p->device);
    if (rc) return rc;
    const size_t Q = p->nq, P = p->n_active;                 // P == 0: every row is the model at its fixed values
    DevBuf<double> dq, dp, dr, dv, dw, ds, dc;
    HIPCHK(dq.alloc(Q)); HIPCHK(dp.alloc(n * P)); HIPCHK(dr.alloc((size_t)n * Q));
    HIPCHK(dv.alloc(n)); HIPCHK(dw.alloc(n)); HIPCHK(ds.alloc(n)); HIPCHK(dc.alloc(Q));
    HIPCHK(hipMemcpy(dq.p, p->q, sizeof(double) * Q, hipMemcpyHostToDevice));
    if (P > 0) HIPCHK(hipMemcpy(dp.p, pset, sizeof(double) * n * P, hipMemcpyHostToDevice));
    SmearDev smear;
    rc = smear.upload(p, p->nq, &m);
    if (rc) return rc;
    size_t lds = sizeof(double) * table_doubles_host(p->model_id, m.int_div);
    switch (p->model_id) {
#define CASE_K(mm) case mm: model_rows_kernel<mm><<<n, WAVE, lds>>>(m, p->nq, dq.p, dp.p, n, dr.p, dv.p, dw.p, ds.p); break;
        MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
        default:
            if (!is_plugin_model(p->model_id)) return fail(MCSAS_EINVAL, "model %d", p->model_id);
            rc = plugin_small_launch(p->model_id, 0, dim3(n), lds, m, (int)p->nq, (const double *)dq.p, (const double *)dp.p, (int)n, dr.p, dv.p, dw.p, ds.p);
            if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
    rows_cumsum_kernel<<<(p->nq + 255) / 256, 256>>>(p->nq, n, dr.p, dc.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if (cum_int) HIPCHK(hipMemcpy(cum_int, dc.p, sizeof(double) * Q, hipMemcpyDeviceToHost));
    if (vset) HIPCHK(hipMemcpy(vset, dv.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (wset) HIPCHK(hipMemcpy(wset, dw.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (sset) HIPCHK(hipMemcpy(sset, ds.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (rows) HIPCHK(hipMemcpy(rows, dr.p, sizeof(double) * n * Q, hipMemcpyDeviceToHost));
    return MCSAS_OK;
}

extern "C" int mcsas_hip_bgfit(int32_t nq, const double *I, const double *sigma, const double *C, int32_t find_bg,
                               int32_t pos_bg, int32_t num_params, int32_t device, double out[4]) {
    if (nq < 1 || !I || !sigma || !C || !out) return fail(MCSAS_EINVAL, "bad argument");
    DeviceGuard dev_guard;
    int rc = select_device(device);
    if (rc) return rc;
    DevBuf<double> dI, dS, dC, dO;
    HIPCHK(dI.alloc(nq)); HIPCHK(dS.alloc(nq)); HIPCHK(dC.alloc(nq)); HIPCHK(dO.alloc(4));
    HIPCHK(hipMemcpy(dI.p, I, sizeof(double) * nq, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dS.p, sigma, sizeof(double) * nq, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dC.p, C, sizeof(double) * nq, hipMemcpyHostToDevice));
    bgfit_kernel<<<1, WAVE>>>(nq, dI.p, dS.p, dC.p, find_bg != 0, pos_bg != 0, num_params, dO.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dO.p, sizeof(double) * 4, hipMemcpyDeviceToHost));
    return MCSAS_OK;
}

extern "C" int mcsas_hip_observability(const mcsas_problem *p, const double *contribs, const double *scaling,
                                       const double *vol_frac, double *min_req_vol) {
    if (!p || !contribs || !scaling || !vol_frac || !min_req_vol || !p->q || !p->sigma) return fail(MCSAS_EINVAL, "bad argument");
    ModelArgs m;
    int rc = fill_model_args(p, &m);
    if (rc) return rc;
    DeviceGuard dev_guard;
    rc = select_device(p->device);
    if (rc) return rc;
    const size_t Q = p->nq, P = p->n_active, N = p->n_contrib, R = p->n_reps;
    DevBuf<double> dq, dsg, dc, dsc, dvf, dm;
    HIPCHK(dq.alloc(Q)); HIPCHK(dsg.alloc(Q)); HIPCHK(dc.alloc(N * P * R)); HIPCHK(dsc.alloc(R));
    HIPCHK(dvf.alloc(N * R)); HIPCHK(dm.alloc(N * R));
    HIPCHK(hipMemcpy(dq.p, p->q, sizeof(double) * Q, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsg.p, p->sigma, sizeof(double) * Q, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc.p, contribs, sizeof(double) * N * P * R, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsc.p, scaling, sizeof(double) * R, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dvf.p, vol_frac, sizeof(double) * N * R, hipMemcpyHostToDevice));
    SmearDev smear;
    rc = smear.upload(p, p->nq, &m);
    if (rc) return rc;
    size_t lds = sizeof(double) * table_doubles_host(p->model_id, m.int_div);
    dim3 grid((unsigned)N, (unsigned)R);
    switch (p->model_id) {
#define CASE_K(mm) case mm: observability_kernel<mm><<<grid, WAVE, lds>>>(m, p->nq, dq.p, dsg.p, (int)N, (int)R, dc.p, dsc.p, dvf.p, dm.p); break;
        MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
        default: {
            if (!is_plugin_model(p->model_id)) return fail(MCSAS_EINVAL, "model %d", p->model_id);
            int rcp = plugin_small_launch(p->model_id, 1, grid, lds, m, (int)p->nq, (const double *)dq.p, (const double *)dsg.p, (int)N, (int)R,
                                          (const double *)dc.p, (const double *)dsc.p, (const double *)dvf.p, dm.p);
            if (rcp) return rcp;
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(min_req_vol, dm.p, sizeof(double) * N * R, hipMemcpyDeviceToHost));
    return MCSAS_OK;
}

extern "C" int mcsas_hip_histogram_prep(const mcsas_problem *p, const double *contribs, double *scaling, double *vset,
                                        double *wset, double *sset, double *min_req_vol) {
    if (!p || !contribs || !scaling || !vset || !wset || !sset || !min_req_vol || !p->q || !p->intensity || !p->sigma ||
        p->nq < 1 || p->n_contrib < 1 || p->n_reps < 1)
        return fail(MCSAS_EINVAL, "bad argument");
    ModelArgs m;
    int rc = fill_model_args(p, &m);
    if (rc) return rc;
    DeviceGuard dev_guard;
    rc = select_device(p->device);
    if (rc) return rc;
    const size_t Q = p->nq, P = p->n_active, N = p->n_contrib, R = p->n_reps;
    // the rows of a block of repetitions stay in HBM between the three kernels; blocks of at most ~8 GB
    const size_t per_rep = N * Q * sizeof(double);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(R, ((size_t)8 << 30) / per_rep));
    DevBuf<double> dq, dI, dsg, dc, drows, dsc, dv, dw, ds, dm;
    HIPCHK(dq.alloc(Q)); HIPCHK(dI.alloc(Q)); HIPCHK(dsg.alloc(Q)); HIPCHK(dc.alloc(N * P * R)); HIPCHK(drows.alloc(chunk * N * Q));
    HIPCHK(dsc.alloc(2 * R)); HIPCHK(dv.alloc(N * R)); HIPCHK(dw.alloc(N * R)); HIPCHK(ds.alloc(N * R)); HIPCHK(dm.alloc(N * R));
    HIPCHK(hipMemcpy(dq.p, p->q, sizeof(double) * Q, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dI.p, p->intensity, sizeof(double) * Q, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsg.p, p->sigma, sizeof(double) * Q, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc.p, contribs, sizeof(double) * N * P * R, hipMemcpyHostToDevice));
    SmearDev smear;
    rc = smear.upload(p, p->nq, &m);
    if (rc) return rc;
    const size_t lds = sizeof(double) * table_doubles_host(p->model_id, m.int_div);
    for (size_t r0 = 0; r0 < R; r0 += chunk) {
        const unsigned nr = (unsigned)std::min(chunk, R - r0);
        const dim3 grid((unsigned)N, nr);
        switch (p->model_id) {
#define CASE_K(mm) case mm: hist_rows_kernel<mm><<<grid, WAVE, lds>>>(m, p->nq, dq.p, (int)N, (int)R, (int)r0, dc.p, drows.p, dv.p, dw.p, ds.p); break;
            MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
            default: {
                if (!is_plugin_model(p->model_id)) return fail(MCSAS_EINVAL, "model %d", p->model_id);
                int rcp = plugin_small_launch(p->model_id, 2, grid, lds, m, (int)p->nq, (const double *)dq.p, (int)N, (int)R, (int)r0, (const double *)dc.p,
                                              drows.p, dv.p, dw.p, ds.p);
                if (rcp) return rcp;
            }
        }
        HIPCHK(hipGetLastError());
        hist_fit_kernel<<<nr, WAVE>>>(p->nq, dI.p, dsg.p, (int)N, (int)R, (int)r0, drows.p, p->find_background != 0,
                                      p->positive_background != 0, dsc.p);
        HIPCHK(hipGetLastError());
        hist_obs_kernel<<<grid, WAVE>>>(p->nq, dsg.p, (int)N, (int)R, (int)r0, drows.p, dsc.p, dv.p, dw.p, dm.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpy(scaling, dsc.p, sizeof(double) * 2 * R, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vset, dv.p, sizeof(double) * N * R, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(wset, dw.p, sizeof(double) * N * R, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sset, ds.p, sizeof(double) * N * R, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(min_req_vol, dm.p, sizeof(double) * N * R, hipMemcpyDeviceToHost));
    return MCSAS_OK;
}


// McSAS.histogram() complete (include/mcsas_hip.h): ONE packed upload (data vectors, parameter sets, bin edges, histogram records) from
// a pinned staging block, the kernels back to back on the null stream, ONE packed download — a call costs two transfers and no
// allocation once its block sizes have been seen (MemCache).
extern "C" int mcsas_hip_histogram(const mcsas_problem *p, const double *contribs, int32_t n_hist, const mcsas_histogram_spec *specs,
                                   double *scaling, double *fractions, double *out) {
    if (!p || !contribs || !scaling || !p->q || !p->intensity || !p->sigma || p->nq < 1 || p->n_contrib < 1 || p->n_reps < 1 || n_hist < 0 ||
        (n_hist > 0 && (!specs || !out)))
        return fail(MCSAS_EINVAL, "bad argument");
    if (p->n_contrib > 4096) return fail(MCSAS_EINVAL, "mcsas_hip_histogram stages a repetition's contributions in LDS: n_contrib %d > 4096 (use mcsas_hip_histogram_prep)", p->n_contrib);
    ModelArgs m;
    int rc = fill_model_args(p, &m);
    if (rc) return rc;
    DeviceGuard dev_guard;
    rc = select_device(p->device);
    if (rc) return rc;
    const size_t Q = p->nq, P = p->n_active, N = p->n_contrib, R = p->n_reps, NR = N * R;
    size_t n_edges = 0, n_out = 0;
    std::vector<HistSpecDev> hs((size_t)n_hist);
    for (int h = 0; h < n_hist; ++h) {
        const mcsas_histogram_spec &sp = specs[h];
        if (sp.n_bin < 0 || sp.n_bin > (1 << 20) || sp.param_index < 0 || sp.param_index >= (int)std::max<size_t>(P, 1) || sp.weighting < 0 || sp.weighting > 3 ||
            (sp.n_bin > 0 && !sp.edges) || P == 0)
            return fail(MCSAS_EINVAL, "histogram %d: param_index %d, weighting %d, n_bin %d", h, sp.param_index, sp.weighting, sp.n_bin);
        hs[h] = HistSpecDev{sp.param_index, sp.weighting, sp.n_bin, 0, (int64_t)n_edges, (int64_t)n_out, sp.lower, sp.upper};
        n_edges += (size_t)sp.n_bin + 1;
        n_out += (size_t)3 * sp.n_bin * R + 5 * R;
    }
    // staging layout (doubles): in = [q | I | sigma | contribs | edges | specs], back = [scaling 2R | fractions 8NR | out]
    const size_t spec_d = (sizeof(HistSpecDev) * (size_t)n_hist + 7) / 8;
    const size_t n_in = 3 * Q + N * P * R + n_edges + spec_d, n_back = 2 * R + 8 * NR + n_out;
    const size_t per_rep = N * Q * sizeof(double);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(R, ((size_t)8 << 30) / per_rep));
    DevBuf<double> din, dback, drows, dcum, dvws;
    HIPCHK(din.alloc(n_in)); HIPCHK(dback.alloc(n_back)); HIPCHK(drows.alloc(chunk * N * Q)); HIPCHK(dcum.alloc(chunk * Q)); HIPCHK(dvws.alloc(4 * NR));
    struct Pinned {
        double *p = nullptr; size_t bytes = 0;
        ~Pinned() { cached_host_free(p, bytes, hipHostMallocDefault); }
    } hin, hback;
    hin.bytes = sizeof(double) * n_in; hback.bytes = sizeof(double) * n_back;
    HIPCHK(cached_host_malloc((void **)&hin.p, hin.bytes, hipHostMallocDefault));
    HIPCHK(cached_host_malloc((void **)&hback.p, hback.bytes, hipHostMallocDefault));
    // every early return below leaves copies / kernels queued on the null stream: wait for them before the staging blocks above
    // (destroyed after this guard) go back to the cache, where the next call would overwrite them under the DMA
    struct SyncGuard {
        bool armed = true;
        ~SyncGuard() { if (armed) (void)hipStreamSynchronize(nullptr); }
    } sync_guard;
    double *w = hin.p;
    memcpy(w, p->q, sizeof(double) * Q); w += Q;
    memcpy(w, p->intensity, sizeof(double) * Q); w += Q;
    memcpy(w, p->sigma, sizeof(double) * Q); w += Q;
    memcpy(w, contribs, sizeof(double) * N * P * R); w += N * P * R;
    for (int h = 0; h < n_hist; ++h) { memcpy(w, specs[h].edges, sizeof(double) * ((size_t)specs[h].n_bin + 1)); w += (size_t)specs[h].n_bin + 1; }
    if (n_hist) memcpy(w, hs.data(), sizeof(HistSpecDev) * (size_t)n_hist);
    HIPCHK(hipMemcpyAsync(din.p, hin.p, hin.bytes, hipMemcpyHostToDevice, nullptr));
    const double *dq = din.p, *dI = din.p + Q, *dsg = din.p + 2 * Q, *dc = din.p + 3 * Q, *dedges = dc + N * P * R;
    const HistSpecDev *dspecs = reinterpret_cast<const HistSpecDev *>(dedges + n_edges);
    double *dsc = dback.p, *dfrac = dback.p + 2 * R, *dout = dfrac + 8 * NR;
    double *dv = dvws.p, *dw = dvws.p + NR, *ds = dvws.p + 2 * NR, *dm = dvws.p + 3 * NR;
    SmearDev smear;
    rc = smear.upload(p, p->nq, &m);
    if (rc) return rc;
    const size_t lds = sizeof(double) * table_doubles_host(p->model_id, m.int_div);
    for (size_t r0 = 0; r0 < R; r0 += chunk) {
        const unsigned nr = (unsigned)std::min(chunk, R - r0);
        const dim3 grid((unsigned)N, nr);
        switch (p->model_id) {
#define CASE_K(mm) case mm: hist_rows_kernel<mm><<<grid, WAVE, lds>>>(m, p->nq, dq, (int)N, (int)R, (int)r0, dc, drows.p, dv, dw, ds); break;
            MCSAS_FOR_MODELS(CASE_K)
#undef CASE_K
            default: {
                if (!is_plugin_model(p->model_id)) return fail(MCSAS_EINVAL, "model %d", p->model_id);
                int rcp = plugin_small_launch(p->model_id, 2, grid, lds, m, (int)p->nq, (const double *)dq, (int)N, (int)R, (int)r0, (const double *)dc,
                                              drows.p, dv, dw, ds);
                if (rcp) return rcp;
            }
        }
        HIPCHK(hipGetLastError());
        hist_colsum_kernel<<<dim3((unsigned)((Q + 255) / 256), nr), 256>>>(p->nq, (int)N, drows.p, dcum.p);
        HIPCHK(hipGetLastError());
        hist_fit_cum_kernel<<<nr, WAVE>>>(p->nq, dI, dsg, (int)R, (int)r0, dcum.p, p->find_background != 0, p->positive_background != 0, dsc);
        HIPCHK(hipGetLastError());
        hist_obs_kernel<<<grid, WAVE>>>(p->nq, dsg, (int)N, (int)R, (int)r0, drows.p, dsc, dv, dw, dm);
        HIPCHK(hipGetLastError());
    }
    hist_fractions_kernel<<<(unsigned)R, WAVE>>>((int)N, (int)R, dsc, dv, dw, ds, dm, dfrac);
    HIPCHK(hipGetLastError());
    if (n_hist > 0) {
        if (sizeof(double) * 3 * N > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)hist_bins_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 3 * N)));
        hist_bins_kernel<<<dim3((unsigned)n_hist, (unsigned)R), WAVE, sizeof(double) * 3 * N>>>((int)N, (int)P, (int)R, dc, dfrac, dedges, dspecs, dout);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(hback.p, dback.p, hback.bytes, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    sync_guard.armed = false;
    memcpy(scaling, hback.p, sizeof(double) * 2 * R);
    if (fractions) memcpy(fractions, hback.p + 2 * R, sizeof(double) * 8 * NR);
    if (n_out) memcpy(out, hback.p + 2 * R + 8 * NR, sizeof(double) * n_out);
    return MCSAS_OK;
}

extern "C" int mcsas_hip_prepare_uncertainty(int32_t n, const double *intensity, const double *sigma_raw, double fu_min,
                                             int32_t device, double *sigma_out) {
    if (n < 1 || !intensity || !sigma_out) return fail(MCSAS_EINVAL, "bad argument");
    DeviceGuard dev_guard;
    int rc = select_device(device);
    if (rc) return rc;
    DevBuf<double> dI, dS, dO;
    HIPCHK(dI.alloc(n)); HIPCHK(dO.alloc(n));
    HIPCHK(hipMemcpy(dI.p, intensity, sizeof(double) * n, hipMemcpyHostToDevice));
    if (sigma_raw) { HIPCHK(dS.alloc(n)); HIPCHK(hipMemcpy(dS.p, sigma_raw, sizeof(double) * n, hipMemcpyHostToDevice)); }
    prepare_uncertainty_kernel<<<(n + 255) / 256, 256>>>(n, dI.p, sigma_raw ? dS.p : nullptr, fu_min, dO.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(sigma_out, dO.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return MCSAS_OK;
}

extern "C" int mcsas_hip_rebin(int32_t n, const double *x, const double *f, const double *fu, int32_t n_bin,
                               const double *edges, int32_t device, double *x_out, double *f_out, double *fu_out,
                               int32_t *n_out) {
    if (n < 1 || n_bin < 1 || n_bin > 1000000 || !x || !f || !fu || !edges || !x_out || !f_out || !fu_out || !n_out)
        return fail(MCSAS_EINVAL, "bad argument");
    DeviceGuard dev_guard;
    int rc = select_device(device);
    if (rc) return rc;
    DevBuf<double> dx, df, du, de, bx, bf, bu;
    DevBuf<int32_t> bc;
    HIPCHK(dx.alloc(n)); HIPCHK(df.alloc(n)); HIPCHK(du.alloc(n)); HIPCHK(de.alloc(n_bin + 1));
    HIPCHK(bx.alloc(n_bin)); HIPCHK(bf.alloc(n_bin)); HIPCHK(bu.alloc(n_bin)); HIPCHK(bc.alloc(n_bin));
    HIPCHK(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(df.p, f, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(du.p, fu, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(de.p, edges, sizeof(double) * (n_bin + 1), hipMemcpyHostToDevice));
    rebin_kernel<<<n_bin, WAVE>>>(n, dx.p, df.p, du.p, de.p, bx.p, bf.p, bu.p, bc.p);
    HIPCHK(hipGetLastError());
    std::vector<double> hx(n_bin), hf(n_bin), hu(n_bin);
    std::vector<int32_t> hc(n_bin);
    HIPCHK(hipMemcpy(hx.data(), bx.p, sizeof(double) * n_bin, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hf.data(), bf.p, sizeof(double) * n_bin, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hu.data(), bu.p, sizeof(double) * n_bin, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hc.data(), bc.p, sizeof(int32_t) * n_bin, hipMemcpyDeviceToHost));
    int32_t k = 0;
    for (int b = 0; b < n_bin; ++b)                               // empty bins (and NaN means) are dropped (:339-341)
        if (hc[b] > 0 && hf[b] == hf[b]) { x_out[k] = hx[b]; f_out[k] = hf[b]; fu_out[k] = hu[b]; ++k; }
    *n_out = k;
    return MCSAS_OK;
}

// streams for hosts without a HIP binding of their own (ctypes / cgo callers that want analyses on several streams)
// ------------------------------------------------------------------------------ rows evaluated by the caller (ABI 4)
// Philox4x32-10 on the host: the stream of device_util.h's philox_uniform (counter = (idx >> 1, chain, 0), key = seed; even draws
// take words 0,1, odd draws words 2,3; 53 bits)
static double philox_uniform_host(uint64_t seed, uint32_t chain, uint64_t idx) {
    const uint64_t blk = idx >> 1;
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = chain, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t a = (idx & 1) ? c2 : c0, b = (idx & 1) ? c3 : c1;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

extern "C" int mcsas_hip_analyse_host_rows(const mcsas_problem *p, mcsas_rows_callback rows_cb, void *user, int32_t window,
                                           mcsas_result *res) {
    if (!p || !rows_cb || !res) return fail(MCSAS_EINVAL, "null argument");
    if (p->struct_size != sizeof(mcsas_problem))
        return fail(MCSAS_EINVAL, "mcsas_problem size %u, library expects %zu (ABI mismatch)", p->struct_size, sizeof(mcsas_problem));
    if (p->nq < 1 || p->nq > 16384 || !p->q || !p->intensity || !p->sigma) return fail(MCSAS_EINVAL, "nq %d (1..16384) / data pointers", p->nq);
    if (p->n_active < 1 || p->n_active > MCSAS_MAX_ACTIVE) return fail(MCSAS_EINVAL, "n_active %d (1..%d)", p->n_active, MCSAS_MAX_ACTIVE);
    if (p->n_contrib < 1 || p->n_reps < 1 || p->max_iter < 0 || p->max_retries < 0) return fail(MCSAS_EINVAL, "n_contrib / n_reps / max_iter / max_retries");
    if (p->reserved0) return fail(MCSAS_EINVAL, "reserved0 must be 0");
    for (int c = 0; c < p->n_active; ++c)
        if (p->gen_kind[c] < MCSAS_GEN_UNIFORM || p->gen_kind[c] > MCSAS_GEN_EXP3) return fail(MCSAS_EINVAL, "gen_kind[%d] = %d", c, p->gen_kind[c]);
    if (window < 1) window = 64;
    DeviceGuard guard;
    { int rc = select_device(p->device); if (rc) return rc; }
    const size_t R = p->n_reps, N = p->n_contrib, P = p->n_active, Q = p->nq, qpad = (Q + 63) / 64 * 64;
    const auto t_begin = std::chrono::steady_clock::now();

    // data vectors (sigma == 0 -> 1, backgroundscalingfit.py:117) and their sums, as mcsas_hip_plan_create prepares them
    std::vector<double> hw(qpad, 0.), hwI(qpad, 0.), hI(qpad, 0.);
    double Sw = 0, SI = 0, SII = 0, Ss2 = 0;
    for (size_t i = 0; i < Q; ++i) {
        const double e = p->sigma[i] == 0.0 ? 1.0 : p->sigma[i], w = 1.0 / (e * e);
        hw[i] = w; hwI[i] = w * p->intensity[i]; hI[i] = p->intensity[i];
        Sw += w; SI += w * p->intensity[i]; SII += w * p->intensity[i] * p->intensity[i]; Ss2 += e * e;
    }
    DevBuf<double> dw, dwI, dI, drset, dcache, dft, dfit, drows, dpv;
    DevBuf<FeedState> dstate;
    DevBuf<int32_t> dmeta;
    // rows per round: every chain's block is at most max(N, window) rows; as many chains per round as fit ~256 MB of staging
    const size_t blk = std::max(N, (size_t)window);
    const size_t round_rows = std::max(blk, std::min(R * blk, (size_t)(256u << 20) / (qpad * sizeof(double))));
    HIPCHK(dw.alloc(qpad)); HIPCHK(dwI.alloc(qpad)); HIPCHK(dI.alloc(qpad));
    HIPCHK(drset.alloc(R * N * P)); HIPCHK(dcache.alloc(R * N * qpad)); HIPCHK(dft.alloc(R * qpad)); HIPCHK(dfit.alloc(R * qpad));
    HIPCHK(drows.alloc(round_rows * qpad)); HIPCHK(dpv.alloc(round_rows * P)); HIPCHK(dstate.alloc(R)); HIPCHK(dmeta.alloc(3 * R));
    HIPCHK(hipMemcpy(dw.p, hw.data(), sizeof(double) * qpad, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dwI.p, hwI.data(), sizeof(double) * qpad, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dI.p, hI.data(), sizeof(double) * qpad, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dstate.p, 0, sizeof(FeedState) * R));
    HIPCHK(hipMemset(dfit.p, 0, sizeof(double) * R * qpad));

    FeedArgs fa;
    memset(&fa, 0, sizeof fa);
    ChainArgs &a = fa.c;
    a.model.n_active = (int)P;
    a.nq = (int)Q; a.qpad = (int)qpad; a.w = dw.p; a.wI = dwI.p; a.I = dI.p; a.Sw = Sw; a.SI = SI; a.SII = SII; a.Ssig2 = Ss2;
    a.n_contrib = (int)N; a.n_reps = (int)R; a.find_bg = p->find_background; a.pos_bg = p->positive_background;
    a.start_from_min = p->start_from_minimum; a.max_retries = p->max_retries; a.max_iter = p->max_iter; a.conv_crit = p->conv_crit;
    a.rset = drset.p; a.cache = dcache.p; a.cache_rows = (int)N; a.fit = dfit.p;
    fa.ft = dft.p; fa.state = dstate.p; fa.rows = drows.p; fa.pvals = dpv.p;
    fa.first = dmeta.p; fa.count = dmeta.p + R; fa.kind = dmeta.p + 2 * R;

    // the chains as the host sees them: where each is in its random stream and in the retry loop of McSAS.analyse (mcsas.py:220-246)
    struct HostChain { uint64_t draw_pos = 0; int64_t num_iter = 0, total = 0; int attempt = 0, phase = 0 /*0 init due, 1 running, 2 done*/, converged = 0, overflow = 0, stopped = 0; double seconds = 0; };
    std::vector<HostChain> hc(R);
    std::vector<FeedState> hs(R);
    std::vector<int32_t> meta(3 * R);
    std::vector<double> pset(round_rows * P), hrows(round_rows * Q), staged(round_rows * qpad, 0.);
    auto uniform = [&](size_t r, uint64_t idx, int *ovf) -> double {
        if (p->replay_stream) {
            if ((int64_t)idx < p->replay_len) return p->replay_stream[r * (size_t)p->replay_len + idx];
            *ovf = 1;
            return 0.5;
        }
        return philox_uniform_host(p->seed, (uint32_t)(p->rep_offset + (int)r), idx);
    };
    auto generate = [&](int c, double u) -> double {                      // numbergenerator.py:28-31,168-191; parameter.py:66-84
        if (p->gen_kind[c] != MCSAS_GEN_UNIFORM) {
            const double up = (double)p->gen_kind[c];
            u = (std::pow(10.0, 0.0 + (up - 0.0) * u) - 1.0) / (p->gen_kind[c] == 1 ? 10.0 : (p->gen_kind[c] == 2 ? 100.0 : 1000.0));
        }
        return u * (p->gen_hi[c] - p->gen_lo[c]) + p->gen_lo[c];
    };
    size_t done = 0;
    int64_t rounds = 0;
    bool stop_seen = false;
    while (done < R) {
        if (p->stop && *p->stop) stop_seen = true;
        size_t nrows = 0;
        std::fill(meta.begin(), meta.end(), 0);
        for (size_t r = 0; r < R; ++r) {
            HostChain &h = hc[r];
            if (h.phase == 2) continue;
            if (stop_seen) {                                                    // McSAS.stop (mcsas.py:357): end every chain where it is
                if (h.phase == 1) { meta[2 * R + r] = FEED_END; h.stopped = 1; }
                else { h.phase = 2; h.stopped = 1; ++done; }
                continue;
            }
            const size_t want = h.phase == 0 ? N : (size_t)std::min<int64_t>(window, p->max_iter - h.num_iter);
            if (nrows + want > round_rows) continue;                            // next round
            meta[r] = (int32_t)nrows; meta[R + r] = (int32_t)want; meta[2 * R + r] = h.phase == 0 ? FEED_INIT : FEED_STEPS;
            for (size_t k = 0; k < want; ++k)
                for (size_t c = 0; c < P; ++c) {
                    double v;
                    if (h.phase == 0) v = p->start_from_minimum ? p->start_value[c]     // mcsas.py:310-317: N draws per parameter, parameter-major
                                                                : generate((int)c, uniform(r, h.draw_pos + c * N + k, &h.overflow));
                    else v = generate((int)c, uniform(r, h.draw_pos + (uint64_t)(h.num_iter + (int64_t)k) * P + c, &h.overflow));   // :358
                    pset[(nrows + k) * P + c] = v;
                }
            nrows += want;
        }
        if (nrows) {
            // ScatteringModel.calc's loop body for every row (scatteringmodel.py:90-99): the caller's calcIntensity
            const int rc = rows_cb(user, (int32_t)nrows, pset.data(), hrows.data());
            if (rc) return fail(MCSAS_ECALLBACK, "rows callback returned %d", rc);
            for (size_t i = 0; i < nrows; ++i) memcpy(&staged[i * qpad], &hrows[i * Q], sizeof(double) * Q);
            HIPCHK(hipMemcpy(drows.p, staged.data(), sizeof(double) * nrows * qpad, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(dpv.p, pset.data(), sizeof(double) * nrows * P, hipMemcpyHostToDevice));
        }
        HIPCHK(hipMemcpy(dmeta.p, meta.data(), sizeof(int32_t) * 3 * R, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(feed_rows_kernel, dim3((unsigned)R), dim3(64), 0, nullptr, fa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(hs.data(), dstate.p, sizeof(FeedState) * R, hipMemcpyDeviceToHost));
        ++rounds;
        const double now = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        for (size_t r = 0; r < R; ++r) {
            HostChain &h = hc[r];
            const int kind = meta[2 * R + r];
            if (h.phase == 2 || (meta[R + r] == 0 && kind != FEED_END)) continue;
            if (kind == FEED_INIT) { h.phase = 1; h.num_iter = 0; if (!p->start_from_minimum) h.draw_pos += (uint64_t)N * P; }
            h.num_iter = hs[r].num_iter;
            if (hs[r].ended) {                                                  // the attempt is over (mcsas.py:424-439)
                h.total += h.num_iter; h.draw_pos += (uint64_t)h.num_iter * P;
                h.converged = hs[r].converged;
                if (h.converged || h.stopped || h.attempt >= p->max_retries) { h.phase = 2; h.seconds = now; ++done; }
                else { ++h.attempt; h.phase = 0; }                              // :220-246: the next attempt goes on in the stream
            }
        }
    }
    // results in the layout of mcsas_result (mcsas.py:203-210,233-251)
    std::vector<double> hr(R * N * P), hf(R * qpad);
    HIPCHK(hipMemcpy(hr.data(), drset.p, sizeof(double) * hr.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hf.data(), dfit.p, sizeof(double) * hf.size(), hipMemcpyDeviceToHost));
    int ovf = 0;
    for (size_t r = 0; r < R; ++r) {
        if (res->contribs) for (size_t n = 0; n < N; ++n) for (size_t c = 0; c < P; ++c) res->contribs[(n * P + c) * R + r] = hr[(r * N + n) * P + c];
        if (res->fit) for (size_t k = 0; k < Q; ++k) res->fit[k * R + r] = hf[r * qpad + k];
        if (res->chisq) res->chisq[r] = hs[r].chi2;
        if (res->scaling) res->scaling[r] = hs[r].A;
        if (res->background) res->background[r] = hs[r].b;
        if (res->num_iter) res->num_iter[r] = hs[r].num_iter;
        if (res->num_moves) res->num_moves[r] = hs[r].num_moves;
        if (res->attempts) res->attempts[r] = hc[r].attempt + 1;
        if (res->converged) res->converged[r] = hc[r].converged;
        if (res->seconds) res->seconds[r] = hc[r].seconds;
        if (res->draws) res->draws[r] = (int64_t)hc[r].draw_pos;
        ovf |= hc[r].overflow;
    }
    if (ovf) return fail(MCSAS_ESTREAM, "replay stream exhausted (replay_len=%lld)", (long long)p->replay_len);
    return MCSAS_OK;
}

extern "C" int mcsas_hip_stream_create(int32_t device, void **stream) {
    if (!stream) return fail(MCSAS_EINVAL, "null argument");
    DeviceGuard dev_guard;
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return MCSAS_OK;
}
extern "C" void mcsas_hip_stream_destroy(void *stream) {
    if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}

extern "C" int mcsas_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int mcsas_hip_abi_version(void) { return MCSAS_ABI_VERSION; }
extern "C" int mcsas_hip_is_tuning_build(void) {
#ifdef MCSAS_TUNING
    return 1;
#else
    return 0;
#endif
}
extern "C" const char *mcsas_hip_last_error(void) { return g_err.c_str(); }
