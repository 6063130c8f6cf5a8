// kern_pipe.hip — instantiates the pipeline tick kernel for ONE model (-DMCSAS_M=<id>) and, for
// model 0 only, the model-independent reset kernel.
#include "chain_pipe.h"
#ifndef MCSAS_M
#error "compile with -DMCSAS_M=<model id>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
using namespace mcsas;

void *CAT(mcsas_pipe_tick_kernel_m, MCSAS_M)(int qpl) {
    switch (qpl) {
        case 1: return (void *)pipe_tick_kernel<MCSAS_M, 1>;
        case 2: return (void *)pipe_tick_kernel<MCSAS_M, 2>;
        case 4: return (void *)pipe_tick_kernel<MCSAS_M, 4>;
        case 8: return (void *)pipe_tick_kernel<MCSAS_M, 8>;
        case 16: return (void *)pipe_tick_kernel<MCSAS_M, 16>;
        default: return nullptr;
    }
}
#if MCSAS_M == 0
void *mcsas_pipe_reset_kernel() { return (void *)pipe_reset_kernel<0>; }
#endif
