// microbench_fp64.hip — what one gfx950 SIMD sustains on the instruction mix of the form-factor
// loops: v_fma_f64 throughput/latency at 1..4 waves per SIMD, DPP wave reduction latency, clock.
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench_fp64 microbench_fp64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../mcsas_amd/csrc/device_util.h"
#include "../mcsas_amd/csrc/fastmath.h"

template <int ILP>
__global__ void k_fma(double *out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(r1 - r0); }
}

__global__ void k_reduce(double *out, int iters) {
    double v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) v = mcsas::wave_sum(v) * 1e-3 + threadIdx.x;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + 64] = v;
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0);
}

__global__ void k_sincos(double *out, int iters) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.37 + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { double s, c; mcsas::sincos_core(x[i], &s, &c); x[i] = x[i] + s * 1e-3 + c * 1e-4; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 8] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

__global__ void k_reduce8(double *out, int iters) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * (i + 1);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double v = 0;
    for (int it = 0; it < iters; ++it) {
        v = mcsas::wave_sum8_transposed(acc, threadIdx.x);
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] * 1e-3 + v * 1e-6 + threadIdx.x * (i + 1);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // check: quantity c total = (c+1) * sum(lane) = (c+1)*2016 at iters == 1
    out[threadIdx.x + 64] = v;
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0);
}

int main() {
    double *d; hipMalloc(&d, 1 << 20);
    double h[2];
    const int iters = 2000;
    printf("v_fma_f64: cycles per wave-instruction on one SIMD (s_memtime), 1 block on 1 CU\n");
    for (int waves : {1, 4, 8, 16}) {   // waves per CU -> waves/4 per SIMD (>=4)
        auto run = [&](auto kern, int ilp) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d, iters, 1.0000001, 1e-9);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            double instr = (double)iters * 16 * ilp;               // per wave
            double per_simd = instr * (waves < 4 ? 1 : waves / 4); // waves sharing a SIMD
            printf("  waves/CU %2d ILP %d: %.2f cyc per instr per SIMD, shader clock %.0f MHz\n", waves, ilp,
                   h[0] / per_simd, h[0] / h[1] * 100.0);
        };
        run(k_fma<1>, 1); run(k_fma<4>, 4); run(k_fma<8>, 8);
    }
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(64), 0, 0, d, 1000); hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("wave_sum (6 DPP stages + readlane, dependent): %.1f cycles each\n", h[0] / 1000);
    hipLaunchKernelGGL(k_reduce8, dim3(1), dim3(64), 0, 0, d, 1); hipDeviceSynchronize();
    {
        std::vector<double> hv(128);
        hipMemcpy(hv.data(), d, 128 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) { int c = 4 * (l & 1) + 2 * ((l >> 1) & 1) + ((l >> 2) & 1); if (hv[64 + l] != (c + 1) * 2016.0) ++bad; }
        printf("wave_sum8_transposed: %d wrong lanes (expect 0)\n", bad);
    }
    hipLaunchKernelGGL(k_reduce8, dim3(1), dim3(64), 0, 0, d, 1000); hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("wave_sum8_transposed (8 sums, dependent): %.1f cycles each\n", h[0] / 1000);
    for (int waves : {1, 4, 8}) {
        hipLaunchKernelGGL(k_sincos, dim3(1), dim3(64 * waves), 0, 0, d, 1000); hipDeviceSynchronize();
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("sincos_core x8 interleaved, waves/CU %d: %.1f cycles per wave-level sincos (per SIMD: %.1f)\n", waves,
               h[0] / 8000, h[0] / 8000 / (waves < 4 ? 1 : waves / 4));
    }
    // all CUs busy: clock under load
    hipLaunchKernelGGL(k_fma<8>, dim3(1024), dim3(256), 0, 0, d, 20000, 1.0000001, 1e-9); hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("full chip (1024 blocks x 256 thr) ILP8: %.2f cyc/instr/SIMD, shader clock %.0f MHz\n",
           h[0] / (20000.0 * 16 * 8), h[0] / h[1] * 100.0);
    return 0;
}
