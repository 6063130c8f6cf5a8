// kern_wide.hip — instantiates the q-split workgroup kernels (more than 1024 q-points) for ONE model (-DMCSAS_M=<id>).
#include "chain_wide.h"
#ifndef MCSAS_M
#error "compile with -DMCSAS_M=<model id>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
using namespace mcsas;

void *CAT(mcsas_wide_kernel_m, MCSAS_M)(int qpl) {
    switch (qpl) {
        case 8: return (void *)chain_wide_kernel<MCSAS_M, 8>;       // up to 4096 q-points (8 waves x 64 lanes x 8)
        case 16: return (void *)chain_wide_kernel<MCSAS_M, 16>;     // up to 8192
        case 32: return (void *)chain_wide_kernel<MCSAS_M, 32>;     // up to 16384
        default: return nullptr;
    }
}
