// microbench_mfma64.hip — v_mfma_f64_16x16x4_f64 on gfx950: operand/result layout (checked against the
// host), issue rate on one SIMD, chip-wide rate, and how an MFMA wave and an fp64 VALU wave share a SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench_mfma64 microbench_mfma64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C: lane l holds A[l % 16][l / 16] and B[l / 16][l % 16]; register r of the
// result holds D[4 * r + l / 16][l % 16]  (NOT 4 * (l / 16) + r as for the f32 16x16x4 form: measured)
__global__ void k_layout(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    v4f64 acc = {0., 0., 0., 0.};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * r + l / 16) * 16 + l % 16] = acc[r];
}

// role 0: MFMA loop, role 1: v_fma_f64 loop; waves [0, mf_waves) of a block take role 0, the rest role 1
template <int NACC>
__global__ void k_rate(double *out, int iters, int mf_waves, double a, double b) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0, t1;
    double sink = 0.;
    if (wave < mf_waves) {
        v4f64 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0., 0., 0., 0.};
        double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NACC; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) sink += x[i];
    }
    out[1024 + blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = (double)(t1 - t0);
}

int main() {
    double *d; hipMalloc(&d, 64 << 20);
    // ---- layout
    std::vector<double> A(64), B(64), D(256), ref(256, 0.);
    for (int i = 0; i < 64; ++i) { A[i] = (rand() % 17) - 8; B[i] = (rand() % 13) - 6; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA = d, *dB = d + 64, *dD = d + 128;
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != ref[i];
    printf("layout check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);

    std::vector<double> h(16);
    const int iters = 400;
    auto run = [&](int blocks, int waves, int mf_waves, int nacc) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (nacc == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64 * waves), 0, 0, d, iters, mf_waves, 1.0000001, 1e-9);
            else hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(64 * waves), 0, 0, d, iters, mf_waves, 1.0000001, 1e-9);
            hipEventRecord(e1); hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, 128, hipMemcpyDeviceToHost);
        const double n_mf = (double)iters * 8 * nacc, n_fma = (double)iters * 64;
        printf("blocks %4d waves/block %2d (mfma waves %d, %d accumulators): ", blocks, waves, mf_waves, nacc);
        if (mf_waves > 0) printf("mfma wave: %.1f cycles per MFMA; ", h[0] / n_mf);
        if (mf_waves < waves) printf("valu wave: %.2f cycles per v_fma_f64; ", h[mf_waves] / n_fma);
        const double flops = (double)blocks * mf_waves * n_mf * 2048.;
        printf("kernel %.3f ms", ms);
        if (mf_waves > 0) printf(" -> %.1f TFLOP/s fp64 matrix", flops / ms * 1e-9);
        printf("\n");
    };
    printf("-- one block on one CU\n");
    run(1, 1, 1, 1); run(1, 1, 1, 4);            // dependent chain / 4 independent accumulators, one wave
    run(1, 4, 4, 4);                             // one MFMA wave per SIMD
    run(1, 8, 8, 4);                             // two per SIMD
    run(1, 4, 0, 4);                             // VALU only, one wave per SIMD
    run(1, 8, 4, 4);                             // one MFMA + one VALU wave per SIMD
    run(1, 8, 0, 4);                             // two VALU waves per SIMD
    printf("-- every CU\n");
    run(256, 4, 4, 4); run(256, 8, 4, 4); run(256, 8, 8, 4); run(256, 8, 0, 4);
    return bad != 0;
}
