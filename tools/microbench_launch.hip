// microbench_launch.hip — cost of a kernel boundary between DEPENDENT launches of a tick-shaped kernel
// (200 workgroups x 512 threads, 147 KB dynamic LDS, body = a short dependent load chain), issued
// (a) back to back on one stream, (b) as a hipGraph of the same nodes.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_launch.hip -o build/microbench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void tick(int *state, int spin) {
    extern __shared__ double lds[];
    if (threadIdx.x == 0) {
        int v = state[blockIdx.x];
        for (int i = 0; i < spin; ++i) v = v * 3 + 1;
        lds[0] = v;
        state[blockIdx.x] = (int)lds[0] + 1;
    }
}

int main() {
    int *state;
    CHK(hipMalloc(&state, 4096 * sizeof(int)));
    CHK(hipMemset(state, 0, 4096 * sizeof(int)));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int n = 200;
    for (size_t lds : {(size_t)0, (size_t)147 * 1024}) {
        if (lds > 64 * 1024) CHK(hipFuncSetAttribute((const void *)tick, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int grid : {1, 50, 200}) {
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(e0, st));
                for (int i = 0; i < n; ++i) tick<<<grid, 512, lds, st>>>(state, 10);
                CHK(hipEventRecord(e1, st));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("stream  lds=%6zu grid=%3d : %.2f us per launch\n", lds, grid, ms * 1e3 / n);
            }
            hipGraph_t g; hipGraphExec_t ge;
            CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < n; ++i) tick<<<grid, 512, lds, st>>>(state, 10);
            CHK(hipStreamEndCapture(st, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(e0, st));
                CHK(hipGraphLaunch(ge, st));
                CHK(hipEventRecord(e1, st));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("graph   lds=%6zu grid=%3d : %.2f us per launch\n", lds, grid, ms * 1e3 / n);
            }
            CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        }
    }
    return 0;
}
