// microbench_persist.hip — what a kernel boundary per tick costs against flags in device memory, for a tick-shaped dataflow:
// R groups ("chains") of 1 consumer ("scan") + G producer workgroups of 512 threads with 147 KB of LDS each.  Tick t: the
// producers of a group write its window buffer [(t+1) & 1] (G x 96 KB) after reading a word the consumer wrote at tick t-1;
// the consumer reads and checks buffer [t & 1] (written at tick t-1) and writes that word.  Both sides burn `spin` dependent
// FMAs per thread on top (the rows / the decisions).
//   (a) one launch per tick on one stream — the dependency is the kernel boundary;
//   (b) one launch for all ticks — release / acquire at agent scope on per-group counters.
// Every value read is checked against what the writer must have written (stale data -> errors > 0).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_persist.hip -o build/microbench_persist && build/microbench_persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int T = 512, PER = 24;                          // doubles per producer thread and tick: 512 x 24 x 8 = 96 KB per block

struct Args {
    double *buf;                                          // [R][2][G][T*PER]
    int32_t *word;                                        // [R] written by the consumer at tick t: t + 1000 * group
    int32_t *scan_done;                                   // [R] ticks the consumer has finished (t + 1)
    int32_t *prod_cnt;                                    // [R][2] producers that have finished tick parity
    int32_t *errors, *abort_flag;
    int R, G, spin;
};

__device__ __forceinline__ double burn(double x, int spin) {
    for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);
    return x;
}

__device__ void produce(const Args &a, int r, int y, int t, double *lds) {          // PROD(t + 1)
    const int tid = threadIdx.x;
    int err = 0;
    if (t >= 1) { const int w = a.word[r]; if (w != (t - 1) + 1000 * r) err = 1; }   // what SCAN(t - 1) left
    double *dst = a.buf + (((size_t)r * 2 + ((t + 1) & 1)) * a.G + y) * (T * PER);
    const double z = burn((double)tid, a.spin) * 0.;      // (0: keeps the values checkable, the loop is not removed)
    lds[tid] = z;
    for (int i = 0; i < PER; ++i) dst[i * T + tid] = (double)(t + 1) * 4096. + (double)(i * T + tid) + z;
    if (err) atomicAdd(a.errors, 1);
}

__device__ void consume(const Args &a, int r, int t, double *lds) {                 // SCAN(t)
    const int tid = threadIdx.x;
    int err = 0;
    if (t >= 0) {
        const double *src = a.buf + ((size_t)r * 2 + (t & 1)) * a.G * (T * PER);
        for (int y = 0; y < a.G; ++y)
            for (int i = 0; i < PER; ++i)
                if (src[(size_t)y * T * PER + i * T + tid] != (double)t * 4096. + (double)(i * T + tid)) err = 1;
    }
    lds[tid] = burn((double)tid, a.spin);
    __syncthreads();
    if (tid == 0) a.word[r] = t + 1000 * r;
    if (err) atomicAdd(a.errors, 1);
}

__global__ __launch_bounds__(T) void tick_kernel(const Args a, int t) {
    extern __shared__ double lds[];
    const int b = blockIdx.x;
    if (b < a.R) consume(a, b, t, lds);
    else produce(a, (b - a.R) / a.G, (b - a.R) % a.G, t, lds);
}

__device__ __forceinline__ bool wait_ge(int32_t *p, int target, int32_t *abort_flag) {
    // one thread polls; the acquire fence behind the barrier makes the block's later loads see what the signaller released
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 1;
        long spins = 0;
        while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > 2000000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { good = 0; break; }
        }
        if (!good) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = good;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}

__global__ __launch_bounds__(T) void persist_kernel(const Args a, int t0, int nt) {
    extern __shared__ double lds[];
    // chain-major block order: a group's blocks are neighbours in the dispatch order
    const int b = blockIdx.x, r = b / (1 + a.G), role = b % (1 + a.G);
    for (int t = t0; t < t0 + nt; ++t) {
        if (role == 0) {
            if (t > t0) { if (!wait_ge(a.prod_cnt + r * 2 + (t & 1), a.G, a.abort_flag)) return; }   // (t == t0: the kernel boundary)
            if (threadIdx.x == 0) a.prod_cnt[r * 2 + (t & 1)] = 0;                  // consumed; the producers of t + 2 start behind scan_done = t + 1
            consume(a, r, t, lds);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(a.scan_done + r, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (t > t0) { if (!wait_ge(a.scan_done + r, t, a.abort_flag)) return; }
            produce(a, r, role - 1, t, lds);
            __syncthreads();                              // every thread's stores are issued ...
            if (threadIdx.x == 0) __hip_atomic_fetch_add(a.prod_cnt + r * 2 + ((t + 1) & 1), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 50, G = argc > 2 ? atoi(argv[2]) : 4, NT = 200;
    Args a{};
    a.R = R; a.G = G;
    // argv[3] = 1: the shared buffers in uncached device memory (MTYPE UC: no dirty L2 lines for the release to write back)
    const bool uncached = argc > 3 && atoi(argv[3]) != 0;
    auto alloc = [&](void **p, size_t n) { return uncached ? hipExtMallocWithFlags(p, n, hipDeviceMallocUncached) : hipMalloc(p, n); };
    CHK(alloc((void **)&a.buf, sizeof(double) * R * 2 * G * T * PER));
    CHK(alloc((void **)&a.word, 4 * R)); CHK(alloc((void **)&a.scan_done, 4 * R)); CHK(alloc((void **)&a.prod_cnt, 8 * R));
    CHK(alloc((void **)&a.errors, 4)); CHK(alloc((void **)&a.abort_flag, 4));
    printf("shared buffers: %s\n", uncached ? "uncached (hipDeviceMallocUncached)" : "hipMalloc");
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const size_t lds = 147 * 1024;
    CHK(hipFuncSetAttribute((const void *)tick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void *)persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nblk = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void *)persist_kernel, T, lds));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("%d groups x (1 + %d) blocks = %d blocks, %d CUs x %d block(s) per CU\n", R, G, R * (1 + G), prop.multiProcessorCount, nblk);
    if (R * (1 + G) > prop.multiProcessorCount * nblk) { printf("grid does not fit: the persistent variant needs every block resident\n"); return 1; }
    auto reset = [&]() {
        hipMemsetAsync(a.word, 0, 4 * R, st); hipMemsetAsync(a.scan_done, 0, 4 * R, st); hipMemsetAsync(a.prod_cnt, 0, 8 * R, st);
        hipMemsetAsync(a.errors, 0, 4, st); hipMemsetAsync(a.abort_flag, 0, 4, st);
    };
    for (int spin : {0, 400, 1600, 4000}) {
        a.spin = spin;
        float ms_a = 0, ms_b[3] = {0, 0, 0};
        int err_a = 0, err_b[3] = {0, 0, 0}, ab[3] = {0, 0, 0};
        for (int rep = 0; rep < 2; ++rep) {
            reset();
            CHK(hipEventRecord(e0, st));
            for (int t = -1; t < NT; ++t) tick_kernel<<<R * (1 + G), T, lds, st>>>(a, t);
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms_a, e0, e1));
            CHK(hipMemcpy(&err_a, a.errors, 4, hipMemcpyDeviceToHost));
        }
        const int per[3] = {NT + 1, 32, 8};                // ticks per launch
        for (int v = 0; v < 3; ++v)
            for (int rep = 0; rep < 2; ++rep) {
                reset();
                CHK(hipEventRecord(e0, st));
                for (int t = -1; t < NT; t += per[v]) {
                    const int n = (t + per[v] <= NT) ? per[v] : NT - t;
                    persist_kernel<<<R * (1 + G), T, lds, st>>>(a, t, n);
                }
                CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms_b[v], e0, e1));
                CHK(hipMemcpy(&err_b[v], a.errors, 4, hipMemcpyDeviceToHost));
                CHK(hipMemcpy(&ab[v], a.abort_flag, 4, hipMemcpyDeviceToHost));
            }
        printf("spin %4d: launch per tick %.2f us/tick (errors %d) | persistent: all ticks %.2f us/tick (errors %d, abort %d), 32 per launch %.2f (%d, %d), 8 per launch %.2f (%d, %d)\n",
               spin, ms_a * 1e3 / (NT + 1), err_a, ms_b[0] * 1e3 / (NT + 1), err_b[0], ab[0], ms_b[1] * 1e3 / (NT + 1), err_b[1], ab[1],
               ms_b[2] * 1e3 / (NT + 1), err_b[2], ab[2]);
    }
    return 0;
}
