// microbench_stream.hip — how fast ONE workgroup can stream 4 KB rows HBM/Infinity-Cache -> LDS with the
// scan block's LDS-DMA ring (8 waves x RING rows in flight, global_load_lds_dwordx4), for 1, 50 and 256
// workgroups at once.  The ceiling for the pipeline's scan blocks (chain_pipe.h).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_stream.hip -o tools/bin/microbench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int RING>
__global__ __launch_bounds__(512) void stream(const double *buf, size_t rows_per_block, int passes, double *sink) {
    extern __shared__ double lds[];
    typedef __attribute__((address_space(3))) void *lds_vp;
    typedef __attribute__((address_space(1))) const void *glb_vp;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double *base = buf + (size_t)blockIdx.x * rows_per_block * 512;
    double *ring = lds + (size_t)wave * RING * 512;
    const int my_rows = (int)(rows_per_block / 8);
    double acc = 0.;
    for (int p = 0; p < passes; ++p) {
        int issued = 0;
        auto issue = [&](int m) {
            const char *g = reinterpret_cast<const char *>(base + (size_t)(wave + 8 * m) * 512) + lane * 16;
            double *l = ring + (size_t)(m % RING) * 512;
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_amdgcn_global_load_lds((glb_vp)(g + c * 1024), (lds_vp)(l + c * 128), 16, 0, 0);
        };
        for (; issued < RING && issued < my_rows; ++issued) issue(issued);
        for (int m = 0; m < my_rows; ++m) {
            if (RING >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (RING == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += ring[(size_t)(m % RING) * 512 + lane];
            if (issued < my_rows) { issue(issued); ++issued; }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    const size_t rows_per_block = 192 * 2 * 4;              // ~6 MB per block: like two window buffers x 4
    const int max_blocks = 256;
    double *buf, *sink;
    CHK(hipMalloc(&buf, sizeof(double) * 512 * rows_per_block * max_blocks));
    CHK(hipMemset(buf, 0, sizeof(double) * 512 * rows_per_block * max_blocks));
    CHK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const size_t lds = 8 * 4 * 4096;
    CHK(hipFuncSetAttribute((const void *)stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int blocks : {1, 8, 50, 128, 256}) {
        const int passes = 4;
        stream<4><<<blocks, 512, lds>>>(buf, rows_per_block, 1, sink);
        CHK(hipEventRecord(e0));
        stream<4><<<blocks, 512, lds>>>(buf, rows_per_block, passes, sink);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)blocks * rows_per_block * 4096 * passes;
        printf("blocks %3d ring 4: %.1f GB/s per block, %.2f TB/s total, %.1f ns per 4 KB row per block\n", blocks,
               bytes / blocks / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e6 / (rows_per_block * passes));
    }
    return 0;
}
