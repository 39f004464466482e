// Device-side helpers shared by the MVIN gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace mvin {

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}) -- for software
// pipelines whose slot / wait-count arithmetic must be constants (a runtime loop's back-edge makes hipcc's waitcnt pass
// drain the loads in flight)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kWave = 64;    // CDNA wavefront
constexpr int kBlock = 256;  // 4 waves per workgroup
constexpr int kTM = 32;      // node tasks (rows) per workgroup tile


// ---- DPP cross-lane reductions (no LDS traffic) --------------------------------------------
// A DPP "row" is 16 lanes.  quad_perm [1,0,3,2] (0xB1) = lane^1, quad_perm [2,3,0,1] (0x4E) =
// lane^2, row_half_mirror (0x141) pairs lane i with 7-i inside each 8, row_mirror (0x140) pairs
// i with 15-i: applied in that order with a commutative op every lane of an aligned 2/4/8/16
// lane group ends up with the group's reduction.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// lane l <-> lane l^16 and l^32 without the LDS crossbar: v_permlane16_swap / v_permlane32_swap (gfx950) exchange the
// odd 16-lane rows (32-lane halves) of one operand with the even ones of the other; fed the same value twice they return
// {value of the even partner, value of the odd partner} in every lane
__device__ __forceinline__ float xor16_sum(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}

// sum over aligned groups of 2^l2 lanes (l2 wave-uniform, 0..6); every lane gets the result
__device__ __forceinline__ float group_sum(float v, int l2) {
    if (l2 >= 1) v += dpp_mov<0xB1>(v);
    if (l2 >= 2) v += dpp_mov<0x4E>(v);
    if (l2 >= 3) v += dpp_mov<0x141>(v);
    if (l2 >= 4) v += dpp_mov<0x140>(v);
    if (l2 >= 5) v = xor16_sum(v);
    if (l2 >= 6) v = xor32_sum(v);
    return v;
}

__device__ __forceinline__ float group_max(float v, int l2) {
    if (l2 >= 1) v = fmaxf(v, dpp_mov<0xB1>(v));
    if (l2 >= 2) v = fmaxf(v, dpp_mov<0x4E>(v));
    if (l2 >= 3) v = fmaxf(v, dpp_mov<0x141>(v));
    if (l2 >= 4) v = fmaxf(v, dpp_mov<0x140>(v));
    if (l2 >= 5) v = xor16_max(v);
    if (l2 >= 6) v = xor32_max(v);
    return v;
}

// ---- whole-wave reductions without the LDS crossbar: DPP inside the 16-lane rows, then the gfx950 lane swaps
// (v_permlane16_swap / v_permlane32_swap) across rows; every lane gets the result ----
__device__ __forceinline__ float rows_combine_sum(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned yy = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(yy, yy, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ float rows_combine_max(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned yy = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(yy, yy, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ float wave_sum_fast(float v) { return rows_combine_sum(group_sum(v, 4)); }
__device__ __forceinline__ float wave_max_fast(float v) { return rows_combine_max(group_max(v, 4)); }
// the plain names are the same reductions (they used to be six __shfl_xor = ds_bpermute round trips through the LDS
// crossbar, ~300 cycles of dependent latency per reduction in every wave-per-node / wave-per-pair kernel)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_fast(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_max_fast(v); }

// exp(x) for x <= 0 (logit - max): the library's argument reduction (product error folded back in by fma) without its
// range selects; ~1 ulp
__device__ __forceinline__ float lean_exp(float x) {
    const float t = x * 1.44269502162933349609375f;              // float(log2 e)
    const float n = rintf(t);
    float f = fmaf(x, 1.44269502162933349609375f, -n);
    f = fmaf(x, 1.925963033500011e-8f, f);                       // log2 e - float(log2 e)
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// ---- entity-table rows: fp32 (16-byte lane loads) or bf16 (8-byte lane loads, widened to fp32
// exactly: bf16 -> f32 is a 16-bit shift) ------------------------------------------------------
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}

// chunk c (4 elements) of row `row` of a [*, D] table; bf16 is wave-uniform
__device__ __forceinline__ float4 load_row4(const void* table, int bf16, int64_t row, int D, int c) {
    if (bf16) return bf16x4_to_f32(reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(table) + row * D)[c]);
    return reinterpret_cast<const float4*>(reinterpret_cast<const float*>(table) + row * D)[c];
}

__device__ __forceinline__ float4 f4_fma(float s, float4 v, float4 a) {
    a.x = fmaf(s, v.x, a.x);
    a.y = fmaf(s, v.y, a.y);
    a.z = fmaf(s, v.z, a.z);
    a.w = fmaf(s, v.w, a.w);
    return a;
}

// Sum a float4 over the lane groups of a wave: lanes l and l^off for off = lpr, 2*lpr, ... < 64.
__device__ __forceinline__ float4 group_xor_sum(float4 a, int lpr) {
    for (int o = lpr; o < 16; o <<= 1) {       // inside a 16-lane row: the LDS crossbar (D < 64 only)
        a.x += __shfl_xor(a.x, o, kWave);
        a.y += __shfl_xor(a.y, o, kWave);
        a.z += __shfl_xor(a.z, o, kWave);
        a.w += __shfl_xor(a.w, o, kWave);
    }
    if (lpr <= 16) a = make_float4(xor16_sum(a.x), xor16_sum(a.y), xor16_sum(a.z), xor16_sum(a.w));
    if (lpr <= 32) a = make_float4(xor32_sum(a.x), xor32_sum(a.y), xor32_sum(a.z), xor32_sum(a.w));
    return a;
}

// acc[i] += sum_k sX[(rg + RP*i)*ldx + k] * W[k*Dout + j], i < NR, RP = kTM / NR.
// sX rows are read as broadcast float4 (every lane of a row group reads the same address);
// W columns are read coalesced across j from global memory (L1/L2 resident, <= 256 KB).
template <int NR>
__device__ __forceinline__ void tile_matvec(const float* __restrict__ sX, int ldx, int Din,
                                            const float* __restrict__ W, int Dout, int j, int rg,
                                            float (&acc)[NR]) {
    constexpr int RP = kTM / NR;
    const float* wp = W + j;
    for (int k = 0; k < Din; k += 4) {
        const float w0 = wp[(size_t)(k + 0) * Dout];
        const float w1 = wp[(size_t)(k + 1) * Dout];
        const float w2 = wp[(size_t)(k + 2) * Dout];
        const float w3 = wp[(size_t)(k + 3) * Dout];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float4 x = *reinterpret_cast<const float4*>(sX + (rg + RP * i) * ldx + k);
            acc[i] = fmaf(x.x, w0, acc[i]);
            acc[i] = fmaf(x.y, w1, acc[i]);
            acc[i] = fmaf(x.z, w2, acc[i]);
            acc[i] = fmaf(x.w, w3, acc[i]);
        }
    }
}

}  // namespace mvin
