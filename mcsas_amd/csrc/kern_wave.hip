// kern_wave.hip — instantiates the wave-per-chain kernels for ONE model (-DMCSAS_M=<id>), so the
// models build in parallel.  Exports a lookup the host code links against.
#include "chain_wave.h"
#ifndef MCSAS_M
#error "compile with -DMCSAS_M=<model id>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
using namespace mcsas;

template <int QPL> static void *pick(bool cache) {
    return cache ? (void *)chain_wave_kernel<MCSAS_M, QPL, true> : (void *)chain_wave_kernel<MCSAS_M, QPL, false>;
}
void *CAT(mcsas_wave_kernel_m, MCSAS_M)(int qpl, bool cache) {
    switch (qpl) {
        case 1: return pick<1>(cache);
        case 2: return pick<2>(cache);
        case 4: return pick<4>(cache);
        case 8: return pick<8>(cache);
        case 16: return pick<16>(cache);
        // nq up to 2048 / 4096 (un-binned data files, nBin = 0): cached rows only — the host forces the cache on
        case 32: return cache ? (void *)chain_wave_kernel<MCSAS_M, 32, true> : nullptr;
        case 64: return cache ? (void *)chain_wave_kernel<MCSAS_M, 64, true> : nullptr;
        default: return nullptr;
    }
}
