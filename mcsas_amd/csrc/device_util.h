// device_util.h — wave64 primitives for the McSAS chain kernels (gfx950 / CDNA4 only).
#pragma once
#ifndef __HIPCC_RTC__          /* hiprtc (run-time model plug-ins) brings the device runtime and the fixed-width types itself */
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace mcsas {

constexpr int WAVE = 64;

// Global-address-space view of a pointer that was read out of an argument block in memory.  The compiler can
// only prove "global" for pointers that are kernel arguments themselves; everything else becomes flat_load /
// flat_store, which count in vmcnt AND lgkmcnt, return out of order and therefore force `s_waitcnt vmcnt(0)
// lgkmcnt(0)` before the first use — no counted waits, no load running ahead of its consumer.
#define MCSAS_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ MCSAS_GLOBAL T *glb(T *p) { return (MCSAS_GLOBAL T *)p; }
template <class T> __device__ __forceinline__ const MCSAS_GLOBAL T *glb(const T *p) { return (const MCSAS_GLOBAL T *)p; }

// ---- cross-lane (DPP) -----------------------------------------------------------------------
// DPP controls (CDNA ISA): quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // old = 0 so rows masked off contribute +0.0
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes; every lane gets the total (wave-uniform, lives in SGPRs).
// xor-1, xor-2 inside quads, half-mirror (8), mirror (16), then the two row broadcasts.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);           // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);          // row_half_mirror
    v += dpp_f64<0x140>(v);          // row_mirror  -> each 16-lane row holds its total
    v += dpp_f64<0x142, 0xA>(v);     // row_bcast15 into rows 1,3
    v += dpp_f64<0x143, 0xC>(v);     // row_bcast31 into rows 2,3 -> row 3 holds the total
    return readlane_f64(v, 63);
}

// two sums at once
__device__ __forceinline__ void wave_sum2(double &a, double &b) {
    a += dpp_f64<0xB1>(a); b += dpp_f64<0xB1>(b);
    a += dpp_f64<0x4E>(a); b += dpp_f64<0x4E>(b);
    a += dpp_f64<0x141>(a); b += dpp_f64<0x141>(b);
    a += dpp_f64<0x140>(a); b += dpp_f64<0x140>(b);
    a += dpp_f64<0x142, 0xA>(a); b += dpp_f64<0x142, 0xA>(b);
    a += dpp_f64<0x143, 0xC>(a); b += dpp_f64<0x143, 0xC>(b);
    a = readlane_f64(a, 63); b = readlane_f64(b, 63);
}

// three sums at once: the chains are independent so the DPP latencies interleave
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c) {
    a += dpp_f64<0xB1>(a); b += dpp_f64<0xB1>(b); c += dpp_f64<0xB1>(c);
    a += dpp_f64<0x4E>(a); b += dpp_f64<0x4E>(b); c += dpp_f64<0x4E>(c);
    a += dpp_f64<0x141>(a); b += dpp_f64<0x141>(b); c += dpp_f64<0x141>(c);
    a += dpp_f64<0x140>(a); b += dpp_f64<0x140>(b); c += dpp_f64<0x140>(c);
    a += dpp_f64<0x142, 0xA>(a); b += dpp_f64<0x142, 0xA>(b); c += dpp_f64<0x142, 0xA>(c);
    a += dpp_f64<0x143, 0xC>(a); b += dpp_f64<0x143, 0xC>(b); c += dpp_f64<0x143, 0xC>(c);
    a = readlane_f64(a, 63); b = readlane_f64(b, 63); c = readlane_f64(c, 63);
}

// ---- transposed reduction: eight sums for the price of ~1.5 -------------------------------------
// In: acc[c], c = 0..7, per-lane partial sums of eight different quantities.  Out: every lane whose
// low three lane-index bits spell (b0,b1,b2) holds the wave total of quantity c = 4*b0 + 2*b1 + b2.
// Three halving exchanges (lane xor 1, 2, 4: half of the values travel, half stay) then three plain
// exchanges (xor 8 by row rotation, xor 16 / 32 with the gfx950 permlane swaps): 70 instructions
// instead of 8 x 20.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_f64_banks(double old, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_xor4(double v) {      // row_shl:4 into banks 0,2; row_shr:4 into banks 1,3
    double d = dpp_f64_banks<0x104, 0x5>(v, v);
    return dpp_f64_banks<0x114, 0xA>(d, v);
}
__device__ __forceinline__ double sum_xor16(double v) {      // v[i] + v[i^16]
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double sum_xor32(double v) {      // v[i] + v[i^32]
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double wave_sum8_transposed(const double (&acc)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    double a4[4], a2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = b0 ? acc[i + 4] : acc[i], send = b0 ? acc[i] : acc[i + 4];
        a4[i] = keep + dpp_f64<0xB1>(send);                  // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b1 ? a4[i + 2] : a4[i], send = b1 ? a4[i] : a4[i + 2];
        a2[i] = keep + dpp_f64<0x4E>(send);                  // quad_perm [2,3,0,1]
    }
    const double keep = b2 ? a2[1] : a2[0], send = b2 ? a2[0] : a2[1];
    double a1 = keep + lane_xor4(send);
    a1 += dpp_f64<0x128>(a1);                                // row_ror:8 == lane xor 8
    a1 = sum_xor16(a1);
    return sum_xor32(a1);
}

// Two sums for the price of ~1.2: the first exchange hands every even lane its neighbour's a and every odd lane its
// neighbour's b, the five exchanges behind it stay within a parity.  Out: the wave total of a in every even lane, of b in
// every odd lane.
__device__ __forceinline__ double wave_sum2_split(double a, double b, int lane) {
    const bool b0 = lane & 1;
    const double keep = b0 ? b : a, send = b0 ? a : b;
    double t = keep + dpp_f64<0xB1>(send);                   // quad_perm [1,0,3,2]
    t += dpp_f64<0x4E>(t);                                   // quad_perm [2,3,0,1]
    t += lane_xor4(t);
    t += dpp_f64<0x128>(t);                                  // row_ror:8 == lane xor 8
    t = sum_xor16(t);
    return sum_xor32(t);
}

__device__ __forceinline__ double wave_min(double v) {
    v = fmin(v, __shfl_xor(v, 1)); v = fmin(v, __shfl_xor(v, 2));
    v = fmin(v, __shfl_xor(v, 4)); v = fmin(v, __shfl_xor(v, 8));
    v = fmin(v, __shfl_xor(v, 16)); v = fmin(v, __shfl_xor(v, 32));
    return v;
}

// ---- Philox4x32-10 (Salmon et al., SC'11) ----------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1;
        c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// draw `idx` of chain `chain`: counter (idx>>1, chain, 0), even idx -> words 0,1, odd -> 2,3;
// 53-bit double ((a>>5)*2^26 + (b>>6)) / 2^53, the construction numpy's legacy generator uses.
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t chain, uint64_t idx) {
    uint64_t blk = idx >> 1;
    uint32_t c[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), chain, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t a = (idx & 1) ? c[2] : c[0], b = (idx & 1) ? c[3] : c[1];
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// NumberGenerator.get transforms (numbergenerator.py:28-31,168-191)
__device__ __forceinline__ double gen_transform(int kind, double u) {
    if (kind == 0) return u;
    double up = (double)kind;
    double rs = pow(10.0, 0.0 + (up - 0.0) * u);
    double den = (kind == 1) ? 10.0 : (kind == 2 ? 100.0 : 1000.0);
    return (rs - 1.0) / den;
}

}  // namespace mcsas
