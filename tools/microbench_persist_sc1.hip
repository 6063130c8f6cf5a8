// microbench_persist_sc1.hip — round 5, stage A of "take the kernel boundary out of the pipeline": the tick-shaped dataflow of
// tools/microbench_persist.hip (R groups of 1 consumer + G producer workgroups, 512 threads, 147 KB of LDS each = one per CU;
// tick t: the producers of a group write window buffer [(t+1) & 1] after reading what the consumer left at tick t-1, the
// consumer reads and checks buffer [t & 1] and leaves its word), handed over the way MI355X_MICROARCH.md's visibility table
// prescribes INSTEAD of agent-scope fences:
//   * every handed-off byte is stored with a 16-byte `buffer_store_dwordx4 ... sc1` (write-through) and loaded with a 16-byte
//     `buffer_load_dwordx4 ... sc1` (served by L2, never by this CU's L1);
//   * every storing wave drains its stores (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane signals — an
//     agent-scope relaxed atomic add on a monotonic per-group counter (producers) or an `sc1` flag store (consumer);
//   * the waiting side: ONE lane polls the counter / flag with an `sc1` load and `s_sleep`, BOUNDED (a time-out raises the
//     abort word, every block leaves), then a workgroup barrier, then the `sc1` loads.
// Three variants are timed in one process on the same buffers, every value checked:
//   (a) one launch per tick (kernel boundary = the hand-off; plain loads / stores)       — what the product does today
//   (b) one launch, release / acquire fences at agent scope (round 4's measurement)
//   (c) one launch, sc1 stores + drained counter, sc1 poll, sc1 loads                     — this file's question
// PER = doubles per producer thread and tick (48: 192 KB per producer block, 768 KB per group and tick = config 2's `d` rows).
//   hipcc -O3 --offload-arch=gfx950 -DPER=48 tools/microbench_persist_sc1.hip -o build/microbench_persist_sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#ifndef PER
#define PER 48
#endif
constexpr int T = 512;
static_assert(PER % 48 == 0 || PER == 24, "16-byte accesses, load batches of 8 / 12 / 24");
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;                                // gfx940+: cache-policy bit 4 = sc1

struct Args {
    double *buf;                                          // [R][2][G][T*PER]
    int32_t *word;                                        // [R][32] (a line each) written by the consumer at tick t: t + 1000 * group
    int32_t *scan_done;                                   // [R][32] ticks the consumer has finished (t + 1)
    int32_t *prod_cnt;                                    // [R][2][32] fence variant: producers done per tick parity; sc1 variant: [R][0] monotonic
    int32_t *errors, *abort_flag;
    int R, G, spin, skew;
};

__device__ __forceinline__ double burn(double x, int spin) {
#pragma unroll 1
    for (int i = 0; i < spin; ++i) { x = fma(x, 1.0000001, 1e-9); asm volatile("" : "+v"(x)); }     // (the same loop in every kernel)
    return x;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
union Pair { u32x4 u; double d[2]; };

// ---------------------------------------------------------------------------------------------- the two bodies, one code shape
// AUX = 0: plain buffer loads / stores (variants a, b); AUX = AUX_SC1: write-through stores, L1-bypassing loads (variant c)
template <int AUX>
__device__ __forceinline__ int produce(const Args &a, int r, int y, int t, double *lds) {          // PROD(t + 1)
    const int tid = threadIdx.x;
    int err = 0;
    if (t >= 1) {
        const int w = AUX ? __hip_atomic_load(a.word + r * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.word[r * 32];
        if (w != (t - 1) + 1000 * r) err = 1;
    }
    double *dst = a.buf + (((size_t)r * 2 + ((t + 1) & 1)) * a.G + y) * (T * PER);
    const __amdgpu_buffer_rsrc_t rs = rsrc(dst, T * PER * 8);
    const double z = burn((double)tid, a.spin + ((y * 7 + r) % 5) * a.skew) * 0.;
    lds[tid] = z;
    for (int i = 0; i < PER / 2; ++i) {
        Pair v;
        v.d[0] = (double)(t + 1) * 65536. + (double)((i * T + tid) * 2) + z;
        v.d[1] = v.d[0] + 1.;
        __builtin_amdgcn_raw_buffer_store_b128(v.u, rs, (i * T + tid) * 16, 0, AUX);
    }
    return err;
}
template <int AUX, int DEPTH = (PER == 24 ? 6 : 8), bool READ = true>
__device__ __forceinline__ int consume(const Args &a, int r, int t, double *lds) {                 // SCAN(t)
    const int tid = threadIdx.x;
    int err = 0;
    if (READ && t >= 0) {
        const double *src = a.buf + ((size_t)r * 2 + (t & 1)) * a.G * (T * PER);
        for (int y = 0; y < a.G; ++y) {
            const __amdgpu_buffer_rsrc_t rs = rsrc(src + (size_t)y * T * PER, T * PER * 8);
            for (int i0 = 0; i0 < PER / 2; i0 += DEPTH) {          // DEPTH 16-byte loads per lane in flight
                Pair v[DEPTH];
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) v[k].u = __builtin_amdgcn_raw_buffer_load_b128(rs, ((i0 + k) * T + tid) * 16, 0, AUX);
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) {
                    const double want = (double)t * 65536. + (double)(((i0 + k) * T + tid) * 2);
                    if (v[k].d[0] != want || v[k].d[1] != want + 1.) err = 1;
                }
            }
        }
    }
    lds[tid] = burn((double)tid, a.spin + (r % 3) * a.skew);
    return err;
}
__global__ __launch_bounds__(T) void tick_kernel(const Args a, int t) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, r = b / (1 + a.G), role = b % (1 + a.G);
    int err;
    if (role == 0) {
        err = consume<0>(a, r, t, lds);
        __syncthreads();
        if (threadIdx.x == 0) a.word[r * 32] = t + 1000 * r;
    } else err = produce<0>(a, r, role - 1, t, lds);
    if (err) atomicAdd(a.errors, 1);
}

// bounded poll by one lane; returns false (and raises the abort word) on time-out
__device__ __forceinline__ bool poll_ge(const int32_t *p, int target, int32_t *abort_flag) {
    long spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {     // global_load_dword ... sc1
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 400000 || ((spins & 63) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- (b) fences
__device__ __forceinline__ bool wait_fence(int32_t *p, int target, int32_t *abort_flag) {
    __shared__ int ok;
    if (threadIdx.x == 0) ok = poll_ge(p, target, abort_flag) ? 1 : 0;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}
__global__ __launch_bounds__(T) void persist_fence_kernel(const Args a, int t0, int nt) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, r = b / (1 + a.G), role = b % (1 + a.G);
    for (int t = t0; t < t0 + nt; ++t) {
        if (role == 0) {
            if (t > t0) { if (!wait_fence(a.prod_cnt + (r * 2 + (t & 1)) * 32, a.G, a.abort_flag)) return; }
            if (threadIdx.x == 0) a.prod_cnt[(r * 2 + (t & 1)) * 32] = 0;
            if (consume<0>(a, r, t, lds)) atomicAdd(a.errors, 1);
            __syncthreads();
            if (threadIdx.x == 0) {
                a.word[r * 32] = t + 1000 * r;
                __hip_atomic_store(a.scan_done + r * 32, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (t > t0) { if (!wait_fence(a.scan_done + r * 32, t, a.abort_flag)) return; }
            if (produce<0>(a, r, role - 1, t, lds)) atomicAdd(a.errors, 1);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(a.prod_cnt + (r * 2 + ((t + 1) & 1)) * 32, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------- (c) sc1 write-through hand-off
__device__ __forceinline__ bool wait_sc1(const int32_t *p, int target, int32_t *abort_flag) {
    __shared__ int ok;
    if (threadIdx.x == 0) ok = poll_ge(p, target, abort_flag) ? 1 : 0;
    __syncthreads();                                       // the other waves load only behind this barrier
    return ok != 0;
}
// The producers may be ONE iteration ahead of each other (iteration t needs SCAN(t-1), which needs everybody's iteration t-2),
// so ONE monotonic counter per group is not enough (found by this file's own check: 512 errors under uneven load): a counter per
// tick parity, monotonic — the adds of iteration t+1 to the parity of iteration t-1 come behind SCAN(t).
template <int DEPTH, bool READ>
__global__ __launch_bounds__(T) void persist_sc1_kernel(const Args a, int t0, int nt) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, r = b / (1 + a.G), role = b % (1 + a.G), tid = threadIdx.x;
    for (int t = t0; t < t0 + nt; ++t) {
        int err;
        if (role == 0) {                                   // SCAN(t): needs PROD(t) = every producer's iteration t - 1
            if (t > t0) { if (!wait_sc1(a.prod_cnt + (r * 2 + (t & 1)) * 32, a.G * ((t - t0 + 1) / 2), a.abort_flag)) return; }
            err = consume<AUX_SC1, DEPTH, READ>(a, r, t, lds);
            if (tid == 0) __hip_atomic_store(a.word + r * 32, t + 1000 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // payload, sc1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.scan_done + r * 32, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // flag, sc1
        } else {                                           // PROD(t + 1): needs SCAN(t - 1)
            if (t > t0) { if (!wait_sc1(a.scan_done + r * 32, t, a.abort_flag)) return; }
            err = produce<AUX_SC1>(a, r, role - 1, t, lds);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains ...
            __syncthreads();                                         // ... the workgroup meets ...
            if (tid == 0) __hip_atomic_fetch_add(a.prod_cnt + (r * 2 + ((t + 1) & 1)) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... one lane signals
        }
        if (err) atomicAdd(a.errors, 1);
    }
}

// ---------------------------------------------------------------------------------------------- (d) uncached buffers, plain accesses
// The question behind it: could the product's tick kernels run several ticks per launch WITHOUT touching their memory accesses, only
// by putting the buffers the workgroups exchange into memory the caches do not keep (hipDeviceMallocUncached, MTYPE UC)?  Plain
// loads / stores, every storing wave drains, barrier, one lane signals (sc1 flag / agent atomic); the waiting side polls, barrier.
// NOT one of MI355X_MICROARCH.md's validated forms: this variant is here to be CHECKED (every value) and timed, nothing else.
__global__ __launch_bounds__(T) void persist_uc_kernel(const Args a, int t0, int nt) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, r = b / (1 + a.G), role = b % (1 + a.G), tid = threadIdx.x;
    for (int t = t0; t < t0 + nt; ++t) {
        int err;
        if (role == 0) {
            if (t > t0) { if (!wait_sc1(a.prod_cnt + (r * 2 + (t & 1)) * 32, a.G * ((t - t0 + 1) / 2), a.abort_flag)) return; }
            err = consume<0>(a, r, t, lds);
            if (tid == 0) a.word[r * 32] = t + 1000 * r;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.scan_done + r * 32, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (t > t0) { if (!wait_sc1(a.scan_done + r * 32, t, a.abort_flag)) return; }
            err = produce<0>(a, r, role - 1, t, lds);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.prod_cnt + (r * 2 + ((t + 1) & 1)) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (err) atomicAdd(a.errors, 1);
    }
}

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 50, G = argc > 2 ? atoi(argv[2]) : 4, NT = argc > 3 ? atoi(argv[3]) : 200;
    const bool uncached = argc > 4 && atoi(argv[4]) != 0;      // argv[4] = 1: payload and word in hipDeviceMallocUncached memory, variant (d) instead of (b)
    Args a{};
    a.R = R; a.G = G;
    if (uncached) {
        CHK(hipExtMallocWithFlags((void **)&a.buf, sizeof(double) * R * 2 * G * T * PER, hipDeviceMallocUncached));
        CHK(hipExtMallocWithFlags((void **)&a.word, 128 * R, hipDeviceMallocUncached));
        printf("payload buffers: hipDeviceMallocUncached; column (b) is variant (d): plain accesses, drained counter, NO fence\n");
    } else {
        CHK(hipMalloc((void **)&a.buf, sizeof(double) * R * 2 * G * T * PER));
        CHK(hipMalloc((void **)&a.word, 128 * R));
    }
    CHK(hipMalloc((void **)&a.scan_done, 128 * R)); CHK(hipMalloc((void **)&a.prod_cnt, 256 * R));
    CHK(hipMalloc((void **)&a.errors, 4)); CHK(hipMalloc((void **)&a.abort_flag, 4));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const size_t lds = 147 * 1024;
    CHK(hipFuncSetAttribute((const void *)tick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void *)persist_fence_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void *)persist_uc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    constexpr int DEEP = PER == 24 ? 12 : 24, D0 = PER == 24 ? 6 : 8;
    CHK(hipFuncSetAttribute((const void *)persist_sc1_kernel<D0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void *)persist_sc1_kernel<DEEP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHK(hipFuncSetAttribute((const void *)persist_sc1_kernel<D0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nblk = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void *)persist_sc1_kernel<D0, true>, T, lds));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("%d groups x (1 + %d) blocks = %d blocks of %d threads, %d CUs x %d block(s) per CU; %d KB per producer block and tick, %d KB per group\n",
           R, G, R * (1 + G), T, prop.multiProcessorCount, nblk, T * PER * 8 / 1024, G * T * PER * 8 / 1024);
    if (R * (1 + G) > prop.multiProcessorCount * nblk) { printf("grid does not fit: the persistent variants need every block resident\n"); return 1; }
    auto reset = [&]() {
        (void)hipMemsetAsync(a.word, 0, 128 * R, st); (void)hipMemsetAsync(a.scan_done, 0, 128 * R, st); (void)hipMemsetAsync(a.prod_cnt, 0, 256 * R, st);
        (void)hipMemsetAsync(a.errors, 0, 4, st); (void)hipMemsetAsync(a.abort_flag, 0, 4, st);
    };
    const int spins[4] = {0, 400, 1600, 4000};
    for (int skewed = 0; skewed < 2; ++skewed)
        for (int si = 0; si < 4; ++si) {
            a.spin = spins[si];
            a.skew = skewed ? a.spin / 4 : 0;              // uneven load: producers / consumers of different groups burn up to 2x
            float ms[5] = {0, 0, 0, 0, 0};
            int err[5] = {0, 0, 0, 0, 0}, ab[5] = {0, 0, 0, 0, 0};
            for (int v = 0; v < 5; ++v)
                for (int rep = 0; rep < 3; ++rep) {
                    reset();
                    CHK(hipEventRecord(e0, st));
                    if (v == 0) for (int t = -1; t < NT; ++t) tick_kernel<<<R * (1 + G), T, lds, st>>>(a, t);
                    if (v == 1 && !uncached) persist_fence_kernel<<<R * (1 + G), T, lds, st>>>(a, -1, NT + 1);
                    if (v == 1 && uncached) persist_uc_kernel<<<R * (1 + G), T, lds, st>>>(a, -1, NT + 1);
                    if (v == 2) persist_sc1_kernel<D0, true><<<R * (1 + G), T, lds, st>>>(a, -1, NT + 1);
                    if (v == 3) persist_sc1_kernel<DEEP, true><<<R * (1 + G), T, lds, st>>>(a, -1, NT + 1);      // (c') deeper load queue
                    if (v == 4) persist_sc1_kernel<D0, false><<<R * (1 + G), T, lds, st>>>(a, -1, NT + 1);        // (c0) flags only: the consumer reads no payload
                    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
                    CHK(hipEventElapsedTime(&ms[v], e0, e1));
                    CHK(hipMemcpy(&err[v], a.errors, 4, hipMemcpyDeviceToHost));
                    CHK(hipMemcpy(&ab[v], a.abort_flag, 4, hipMemcpyDeviceToHost));
                }
            printf("spin %4d skew %4d: (a) launch per tick %7.2f us/tick (errors %d) | (b) fences %7.2f (errors %d, abort %d) | (c) sc1 hand-off %7.2f (errors %d, abort %d), %d loads in flight %7.2f (%d, %d), payload not read %7.2f (%d, %d)\n",
                   a.spin, a.skew, ms[0] * 1e3 / (NT + 1), err[0], ms[1] * 1e3 / (NT + 1), err[1], ab[1], ms[2] * 1e3 / (NT + 1), err[2], ab[2],
                   DEEP, ms[3] * 1e3 / (NT + 1), err[3], ab[3], ms[4] * 1e3 / (NT + 1), err[4], ab[4]);
            fflush(stdout);
        }
    return 0;
}
