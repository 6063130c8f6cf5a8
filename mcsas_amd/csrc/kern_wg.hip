// kern_wg.hip — instantiates the workgroup-per-chain kernels for ONE model (-DMCSAS_M=<id>).
#include "chain_wg.h"
#ifndef MCSAS_M
#error "compile with -DMCSAS_M=<model id>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
using namespace mcsas;

void *CAT(mcsas_wg_kernel_m, MCSAS_M)(int qpl) {
    switch (qpl) {
        case 1: return (void *)chain_wg_kernel<MCSAS_M, 1>;
        case 2: return (void *)chain_wg_kernel<MCSAS_M, 2>;
        case 4: return (void *)chain_wg_kernel<MCSAS_M, 4>;
        case 8: return (void *)chain_wg_kernel<MCSAS_M, 8>;
        case 16: return (void *)chain_wg_kernel<MCSAS_M, 16>;
        default: return nullptr;
    }
}
