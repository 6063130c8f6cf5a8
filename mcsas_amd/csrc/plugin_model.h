// plugin_model.h — Contrib<MCSAS_MODEL_PLUGIN>: a scattering model whose form factor arrives as SOURCE TEXT at run time
// (mcsas_hip_plugin_compile, include/mcsas_hip.h), the counterpart of the reference's model discovery: any models/*.py with a
// ScatteringModel subclass is usable there (utils/findmodels.py:120-186) through four methods (bases/model/
// scatteringmodel.py:15-58, sasmodel.py:36-79).  The plug-in text defines the same four, per q point and on the model's full
// parameter vector p[MCSAS_MAX_PARAMS] (active parameters substituted and clipped into their valueRange like
// Parameter.setValue does):
//
//     __device__ double mcsas_plugin_formfactor(double q, const double *p);   // ScatteringModel.formfactor(dataset) at one q
//     __device__ double mcsas_plugin_volume(const double *p);                  // .volume()
//     __device__ double mcsas_plugin_absvolume(const double *p);               // .absVolume()  (volume x contrast^2 where the model has one)
//     __device__ double mcsas_plugin_surface(const double *p);                 // .surface()
//
// Everything in fastmath.h / device_util.h is available to it (sincos_fast, div_fast, j1_fast, ...).  This header is compiled
// only by hiprtc, in a translation unit assembled by the library: chain_common.h, the plug-in text, this file, then the kernel
// templates (chain_wave.h, chain_wg.h, chain_pipe.h, small_kernels.h) of which the requested ones are instantiated.
#pragma once
#include "models.h"

// What a row of this model costs (models.h, ROW_CLASS), as the pipeline should treat it: 0 = a few hundred instructions per q
// point (the default), 1 = an integral per q point (the form factor loops over orientations / a contour).  The plug-in text
// may say `#define MCSAS_PLUGIN_ROW_CLASS 1`; results do not depend on it, only how the rows are spread over the chip.
#ifndef MCSAS_PLUGIN_ROW_CLASS
#define MCSAS_PLUGIN_ROW_CLASS 0
#endif
// The reference's canSmear flag of the model class (sasmodel.py:56-60): `#define MCSAS_PLUGIN_CAN_SMEAR 1` makes every
// intensity of this model 2 trapz(F(locs)^2 w weights, x = qOffset) when the data set's smearing is configured.
#ifndef MCSAS_PLUGIN_CAN_SMEAR
#define MCSAS_PLUGIN_CAN_SMEAR 0
#endif

// The library's host side read the two values off the text by itself (mcsas_hip_plugin_compile) and hands its reading in: the
// geometry the host picks and the kernel instantiated here must agree on them, so a text the host misreads is refused.
#ifdef MCSAS_HOST_ROW_CLASS
static_assert((MCSAS_PLUGIN_ROW_CLASS) == MCSAS_HOST_ROW_CLASS,
              "MCSAS_PLUGIN_ROW_CLASS: write it as a plain `#define MCSAS_PLUGIN_ROW_CLASS 0|1` line (the library's host side read another value off the text)");
#endif
#ifdef MCSAS_HOST_CAN_SMEAR
static_assert(((MCSAS_PLUGIN_CAN_SMEAR) != 0) == (MCSAS_HOST_CAN_SMEAR != 0),
              "MCSAS_PLUGIN_CAN_SMEAR: write it as a plain `#define MCSAS_PLUGIN_CAN_SMEAR 0|1` line (the library's host side read another value off the text)");
#endif

__device__ double mcsas_plugin_formfactor(double q, const double *p);
__device__ double mcsas_plugin_volume(const double *p);
__device__ double mcsas_plugin_absvolume(const double *p);
__device__ double mcsas_plugin_surface(const double *p);

namespace mcsas {

template <> struct Contrib<MCSAS_MODEL_PLUGIN> {
    static constexpr int ROWTAB = 0, INT_DIV_PARAM = -1, ROW_CLASS = MCSAS_PLUGIN_ROW_CLASS;
    static constexpr bool CAN_SMEAR = MCSAS_PLUGIN_CAN_SMEAR != 0;
    double p[MCSAS_MAX_PARAMS];
    double v, w, s;
    static __host__ __device__ __forceinline__ int table_doubles(int) { return 0; }
    static __device__ __forceinline__ void fill_table(const ModelArgs &, double *, int, int) {}
    __device__ __forceinline__ void prepare(const ModelArgs &a, const double *row) {
        full_params(a, row, p);
        v = mcsas_plugin_absvolume(p);                             // vset: _volume() forwards to absVolume() (scatteringmodel.py:27-31)
        s = mcsas_plugin_surface(p);
        w = pow(mcsas_plugin_volume(p), 2. * a.comp_exp);          // weight() = volume()^(2c) (sasmodel.py:36-44)
    }
    __device__ __forceinline__ Contrib bcast(int lane) const {
        Contrib o;
#pragma unroll
        for (int i = 0; i < MCSAS_MAX_PARAMS; ++i) o.p[i] = readlane_f64(p[i], lane);
        o.w = readlane_f64(w, lane); o.v = 0.; o.s = 0.;
        return o;
    }
    __device__ __forceinline__ double intensity(double q, const double *) const {
        const double f = mcsas_plugin_formfactor(q, p);
        return f * f * w;                                          // sasmodel.py:77
    }
};

}  // namespace mcsas
