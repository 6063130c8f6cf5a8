// microbench_icache.hip — does a kernel's code arrive cold at every launch on gfx950?
//
// The tick kernels' per-row timeline shows a producer wave's FIRST row at 4.0 us and every later one at 1.3 us.
// This stand-in separates the instruction cache from everything else: one straight-line body of NBODY dependent-free fp64 FMAs
// (8 bytes each, no memory access, no LDS) compiled ONCE as a non-inlined function and run three times per wave and launch; s_memtime
// around each pass.  Launched back to back LAUNCHES times with the tick kernels' shape (256 workgroups of 512 threads).
//   pass 1 - pass 3 = what the first execution of the body costs beyond its issue time, per launch;
//   if that difference is there in every launch, code is fetched again after every kernel boundary;
//   argument 4 = 1 puts the same arithmetic from a rolled loop in front of pass 1: a start-of-kernel effect that is not the code's
//   arrival (clocks, power) would then have passed before pass 1 starts.
// Build: hipcc -O3 --offload-arch=gfx950 -DNBODY=1536 -o microbench_icache microbench_icache.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef NBODY
#define NBODY 1536            // FMAs in the body: 12 KB of code
#endif

#ifdef BODY_F32
typedef float real;           // -DBODY_F32: the same number of 8-byte instructions (v_fma_f32), a quarter of the arithmetic per instruction
#else
typedef double real;
#endif

// eight independent chains, so that the body is issue-bound once its code is there
__device__ __noinline__ void body(real (&x)[8], real a, real b) {
#pragma unroll
    for (int i = 0; i < NBODY / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], a, b);
    }
}

#define PIN8(x) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))

// warm: the same NBODY FMAs from a ROLLED loop (a few dozen bytes of code) in front of pass 1 — the chip is at full fp64 load (clocks,
// power) when pass 1 starts, but the body's code has not been executed in this launch
// stagger: the younger half of a workgroup's waves (the ones that share their SIMDs with the older half) start pass 1 when the older
// half is through its own pass 1 (LDS counter, bounded poll)
__global__ __launch_bounds__(512) void k_passes(double *out, unsigned long long *t, real a, real b, int warm, int stagger) {
    __shared__ int passed;
    real x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (real)(threadIdx.x * 1e-3 + j);
    const int wave = threadIdx.x >> 6, half = (int)(blockDim.x >> 7);
    if (threadIdx.x == 0) passed = 0;
    __syncthreads();
    const unsigned long long tw = __builtin_amdgcn_s_memtime();
    if (stagger && wave >= half) {
        for (int spin = 0; spin < 100000; ++spin) {
            if (__hip_atomic_load(&passed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= half) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (warm) {
#pragma unroll 1
        for (int i = 0; i < NBODY / 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], a, b);
            PIN8(x);
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    body(x, a, b);
    PIN8(x);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (stagger && wave < half && (threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&passed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    body(x, a, b);
    PIN8(x);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    body(x, a, b);
    PIN8(x);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    double s = 0.;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (double)x[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        t[4 * w] = t0 - tw; t[4 * w + 1] = t1 - t0; t[4 * w + 2] = t2 - t1; t[4 * w + 3] = t3 - t2;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 512, launches = argc > 3 ? atoi(argv[3]) : 12;
    const int warm = argc > 4 ? atoi(argv[4]) : 0, stagger = argc > 5 ? atoi(argv[5]) : 0;
    if (blocks < 1 || blocks > 4096 || threads < 64 || threads > 512 || threads % 64 || launches < 1 || launches > 64) { fprintf(stderr, "usage: %s [blocks<=4096] [threads 64..512] [launches<=64] [warm 0|1] [stagger 0|1]\n", argv[0]); return 2; }
    const size_t waves = (size_t)blocks * (threads / 64);
    double *out = nullptr;
    unsigned long long *t = nullptr;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    CK(hipMalloc(&t, sizeof(unsigned long long) * 4 * waves * launches));
    for (int l = 0; l < launches; ++l)                              // back to back on one stream, like the ticks of an analysis
        hipLaunchKernelGGL(k_passes, dim3(blocks), dim3(threads), 0, nullptr, out, t + 4 * waves * l, (real)0.999999, (real)1e-7, warm, stagger);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(4 * waves * launches);
    CK(hipMemcpy(h.data(), t, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    printf("body: %d v_fma_%s (%d KB of code), %d workgroups x %d threads, %d launches back to back%s%s; s_memtime ticks, medians over the waves (p90 of pass 1)\n",
           NBODY, sizeof(real) == 4 ? "f32" : "f64", NBODY * 8 / 1024, blocks, threads, launches, warm ? ", a rolled loop of the same FMAs in front of pass 1" : "",
           stagger ? ", the younger half of the waves starts when the older half is through pass 1" : "");
    printf("launch  waves     wait/rolled     pass 1 (p90)        pass 2     pass 3     pass 1 - pass 3\n");
    const int wpb = threads / 64;
    for (int l = 0; l < launches; ++l)
        for (int part = 0; part < (stagger ? 2 : 1); ++part) {      // staggered: older and younger half apart
            std::vector<double> p[4];
            for (int k = 0; k < 4; ++k) {
                for (size_t w = 0; w < waves; ++w) {
                    const int wv = (int)(w % wpb);
                    if (stagger && (wv >= wpb / 2) != (part == 1)) continue;
                    p[k].push_back((double)h[4 * (waves * l + w) + k]);
                }
                std::sort(p[k].begin(), p[k].end());
            }
            const size_t n = p[0].size();
            printf("%4d   %s   %9.0f    %9.0f (%6.0f)    %9.0f  %9.0f    %9.0f\n", l, !stagger ? "all    " : (part ? "younger" : "older  "), p[0][n / 2], p[1][n / 2], p[1][n * 9 / 10],
                   p[2][n / 2], p[3][n / 2], p[1][n / 2] - p[3][n / 2]);
        }
    (void)hipFree(out); (void)hipFree(t);
    return 0;
}
