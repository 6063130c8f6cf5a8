// chain_wg.h — workgroup-per-chain kernel (speculative proposal window); see DESIGN.md.
#pragma once
#include "chain_common.h"

namespace mcsas {

struct WgGeom {
    int32_t waves, window, qpl, tab_doubles;
    uint64_t lds_bytes;
};

// host side: pick the proposal window that fits LDS; nonzero = does not fit
static inline int wg_geometry(int nq, int n_contrib, int tab_doubles, int waves, WgGeom *g) {
    (void)nq; (void)n_contrib; (void)tab_doubles; (void)waves; (void)g;
    return 1;
}

template <int M>
__global__ void chain_wg_kernel(const ChainArgs a, const WgGeom g) {
    (void)a; (void)g;
}

}  // namespace mcsas
