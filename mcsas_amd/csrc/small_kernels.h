// small_kernels.h — the model-templated kernels behind ScatteringModel.calc / McSAS.histogram (mcsas_hip_model_calc,
// mcsas_hip_observability, mcsas_hip_histogram_prep).  A header so that the run-time compiler instantiates them for a
// plug-in model as well (plugin_model.h); the built-in models' instances live in mcsas_hip.hip.
#pragma once
#include "chain_common.h"

namespace mcsas {

// rows[n][q] = SASModel.calcIntensity()[0] for contribution n (sasmodel.py:46-79); one wave per row
template <int M>
__global__ __launch_bounds__(64) void model_rows_kernel(ModelArgs m, int nq, const double *q, const double *pset,
                                                        int n, double *rows, double *vset, double *wset, double *sset) {
    extern __shared__ double tab[];
    Contrib<M>::fill_table(m, tab, threadIdx.x, WAVE);
    __syncthreads();
    const int i = blockIdx.x;
    double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
    for (int p = 0; p < m.n_active; ++p) row[p] = pset[(size_t)i * m.n_active + p];
    Contrib<M> c;
    c.prepare(m, row);
    if (threadIdx.x == 0) { vset[i] = c.v; wset[i] = c.w; sset[i] = c.s; }
    for (int k = threadIdx.x; k < nq; k += WAVE) {
        double it;
        if (Contrib<M>::CAN_SMEAR && m.smear_nk > 0) it = smeared_intensity<M>(c, m.smear_locs_t, m.smear_cw, m.smear_nk, m.smear_stride, k, tab);
        else it = c.intensity(q[k], tab);
        rows[(size_t)i * nq + k] = it;
    }
}

// min over q of sigma*vf / (A*I_c(q)), I_c != 0 (mcsas.py:582-590); one wave per (contribution, rep)
template <int M>
__global__ __launch_bounds__(64) void observability_kernel(ModelArgs m, int nq, const double *q, const double *sigma,
                                                           int N, int R, const double *contribs, const double *scaling,
                                                           const double *vol_frac, double *min_req) {
    extern __shared__ double tab[];
    Contrib<M>::fill_table(m, tab, threadIdx.x, WAVE);
    __syncthreads();
    const int c = blockIdx.x, r = blockIdx.y, P = m.n_active;
    double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
    for (int p = 0; p < P; ++p) row[p] = contribs[((size_t)c * P + p) * R + r];
    Contrib<M> cc;
    cc.prepare(m, row);
    const double vf = vol_frac[(size_t)c * R + r], A = scaling[r];
    double best = __builtin_inf();
    for (int k = threadIdx.x; k < nq; k += WAVE) {
        double it;
        if (Contrib<M>::CAN_SMEAR && m.smear_nk > 0) it = smeared_intensity<M>(cc, m.smear_locs_t, m.smear_cw, m.smear_nk, m.smear_stride, k, tab);
        else it = cc.intensity(q[k], tab);
        double scaled = A * it;
        if (scaled != 0.) best = fmin(best, (sigma[k] * vf) / scaled);
    }
    best = wave_min(best);
    if (threadIdx.x == 0) min_req[(size_t)c * R + r] = best;
}

// rows[r][c][q] = calcIntensity of contribution c of rep r, plus its v/w/s; one wave per (c, r)
template <int M>
__global__ __launch_bounds__(64) void hist_rows_kernel(ModelArgs m, int nq, const double *q, int N, int R, int r0,
                                                       const double *contribs, double *rows, double *vset, double *wset,
                                                       double *sset) {
    extern __shared__ double tab[];
    Contrib<M>::fill_table(m, tab, threadIdx.x, WAVE);
    __syncthreads();
    const int c = blockIdx.x, rl = blockIdx.y, r = r0 + rl, P = m.n_active;
    double row[MCSAS_MAX_ACTIVE] = {0., 0., 0., 0.};
    for (int p = 0; p < P; ++p) row[p] = contribs[((size_t)c * P + p) * R + r];
    Contrib<M> cc;
    cc.prepare(m, row);
    if (threadIdx.x == 0) { vset[(size_t)c * R + r] = cc.v; wset[(size_t)c * R + r] = cc.w; sset[(size_t)c * R + r] = cc.s; }
    double *out = rows + ((size_t)rl * N + c) * nq;
    for (int k = threadIdx.x; k < nq; k += WAVE) {
        double it;
        if (Contrib<M>::CAN_SMEAR && m.smear_nk > 0) it = smeared_intensity<M>(cc, m.smear_locs_t, m.smear_cw, m.smear_nk, m.smear_stride, k, tab);
        else it = cc.intensity(q[k], tab);
        out[k] = it;
    }
}

}  // namespace mcsas
