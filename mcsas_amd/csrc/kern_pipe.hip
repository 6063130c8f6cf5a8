// kern_pipe.hip — instantiates the pipeline tick kernel for ONE model (-DMCSAS_M=<id>) and, for
// model 0 only, the model-independent reset kernel.
#include "chain_pipe.h"
#ifndef MCSAS_M
#error "compile with -DMCSAS_M=<model id>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
using namespace mcsas;

// rowq: the kernel whose producers pull rows from a queue (rows with an integral, smeared models); the models without an integral
// have both, the others only that one
template <int QPL> static void *pick(bool rowq) {
    if (rowq) return (void *)pipe_tick_kernel<MCSAS_M, QPL, true>;
    if constexpr (pipe_light_model_v<MCSAS_M>) return (void *)pipe_tick_kernel<MCSAS_M, QPL, false>;
    return nullptr;
}
void *CAT(mcsas_pipe_tick_kernel_m, MCSAS_M)(int qpl, bool rowq) {
    switch (qpl) {
        case 1: return pick<1>(rowq);
        case 2: return pick<2>(rowq);
        case 4: return pick<4>(rowq);
        case 8: return pick<8>(rowq);
        case 16: return pick<16>(rowq);
        default: return nullptr;
    }
}
#if MCSAS_M == 0
void *mcsas_pipe_reset_kernel() { return (void *)pipe_reset_kernel<0>; }
#endif
