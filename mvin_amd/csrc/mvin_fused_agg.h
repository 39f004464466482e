// Shared device helpers of the per-entity aggregates kernels (mvin_fused_agg.hip: dim 64; mvin_fused_agg32.hip: dim 32).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kAggWaves = 4;
constexpr int kAggUvLd = 132;         // floats per parent of the u1 | v block in LDS (128 + 4: sixteen lanes, sixteen bank groups)
constexpr int kAggPad = 4;            // list entries of padding behind a group's K slots (the half round issued ahead of the last one)
constexpr unsigned kAggOob = 0xFFFFFFF0u;       // a byte offset beyond every buffer: the load returns zeros, no memory access
constexpr unsigned kAggPadRow = 0xFFFFFE00u;    // ... that stays beyond them (and below 2^32) with a lane's column offset added
constexpr int agg_list_words(int K) { return 4 * 2 * (K + kAggPad); }      // per wave: 4 groups x (K + padding) x (offset, weight)
size_t fused_agg_lds_bytes(int nR, int K);

__device__ __forceinline__ int agg_xor16_imax(int v) {
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)a[0], (int)a[1]);
}
__device__ __forceinline__ int agg_xor32_imax(int v) {
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)a[0], (int)a[1]);
}

// The relation logits `t` [nR] (or NULL: no attention) as the table the softmaxes read: exp(t[r] - max over ALL relations) when the
// logits allow it -- softmax is shift invariant, so one table serves every row, without a maximum per row or an exp per slot; a row
// whose own logits all lie far below the global maximum would lose its weights to underflow, so a spread above 60 (exp(-60) = 9e-27:
// sums of K of them stay normal) keeps the logits and the per-row form.  Returns whether the table holds exponentials; the caller
// puts the workgroup barrier behind it.
__device__ __forceinline__ bool agg_logit_table(const float* t, int nR, float* sT, int tid, int lane) {
    float mx = -INFINITY, mn = INFINITY;
    if (t)
        for (int i = lane; i < nR; i += 64) {
            const float l = t[i];
            mx = fmaxf(mx, l), mn = fminf(mn, l);
        }
    else
        mx = mn = 0.f;
    mx = wave_max(mx), mn = -wave_max(-mn);
    const bool fast = __builtin_amdgcn_readfirstlane((mx - mn <= 60.f) ? 1 : 0) != 0;      // (NaN logits: per-row form)
    for (int i = tid; i < nR; i += kAggWaves * 64) {
        const float l = t ? t[i] : 0.f;
        sT[i] = fast ? lean_exp(fminf(l - mx, 0.f)) : l;
    }
    return fast;
}

// weights of a row's slots (SPL per lane of the 2^LG-lane group that holds it; cr = relation | multiplicity << 16 | ...): multiplicity x
// softmax over the distinct slots, over K (aggregators.py:118-146); a padding slot (multiplicity 0) weighs 0
template <int SPL, bool FAST, int LG = 4>
__device__ __forceinline__ void agg_row_weights(const unsigned (&cr)[SPL], bool att, const float* sT, float invK, float (&wk)[SPL]) {
    float lg[SPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const float mu = (float)((cr[i] >> 16) & 0xFFu);
        const float l = att ? sT[cr[i] & 0xFFFFu] : (FAST ? 1.f : 0.f);
        if constexpr (FAST) {
            wk[i] = mu * l;                              // l = exp(logit - global max)
        } else {
            wk[i] = mu;
            mx = fmaxf(mx, mu > 0.f ? l : -INFINITY);
            lg[i] = l;
        }
    }
    if (att) {
        float z = 0.f;
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < SPL; ++i) z += wk[i];
        } else {
            mx = group_max(mx, LG);
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
                wk[i] *= lean_exp(fminf(lg[i] - mx, 0.f));
                z += wk[i];
            }
        }
        z = group_sum(z, LG);
        const float rz = z > 0.f ? invK * __builtin_amdgcn_rcpf(z) : 0.f;
#pragma unroll
        for (int i = 0; i < SPL; ++i) wk[i] *= rz;
    } else {
#pragma unroll
        for (int i = 0; i < SPL; ++i) wk[i] *= invK;
    }
}

template <int SPL>
__device__ __forceinline__ void agg_load_slots(__amdgpu_buffer_rsrc_t adjE, __amdgpu_buffer_rsrc_t adjR, unsigned co, unsigned (&ce)[SPL],
                                               unsigned (&cr)[SPL]) {
    if constexpr (SPL == 1) {
        ce[0] = __builtin_amdgcn_raw_buffer_load_b32(adjE, co, 0, 0);
        cr[0] = __builtin_amdgcn_raw_buffer_load_b32(adjR, co, 0, 0);
    } else if constexpr (SPL == 2) {
        const u32x2 e2 = __builtin_amdgcn_raw_buffer_load_b64(adjE, co, 0, 0);
        const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(adjR, co, 0, 0);
        ce[0] = e2[0], ce[1] = e2[1], cr[0] = r2[0], cr[1] = r2[1];
    } else {
        static_assert(SPL == 4, "SPL");
        const u32x4 e4 = __builtin_amdgcn_raw_buffer_load_b128(adjE, co, 0, 0);
        const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(adjR, co, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) ce[i] = e4[i], cr[i] = r4[i];
    }
}

__device__ __forceinline__ float4 agg_row4(__amdgpu_buffer_rsrc_t tab, unsigned off) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(tab, off, 0, 0);
    return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
}


// the folded-tail kernels' arguments (mvin_score_l2_folded_fwd)
struct FoldArgs {
    const float* agg;            // [2][nE][D] H0 | G
    const float* M0;             // [nE][D]
    const int32_t* adj_e;        // duplicate-slot encoding
    const int32_t* adj_r;
    const int32_t* items;        // [B] (stride pid_stride words)
    const float* t1;             // [nR] relation logits of aggregator (1,.) or NULL
    const float* q;              // [B][D]
    const float* user_o;         // [B][D]
    const float *Wq, *bq, *Wv, *bv, *Wqm, *A1, *a1, *Wm1, *Wm2, *bm;      // the six blocks: regrouped copies (Wperm)
    float* item_emb;             // [B][D] or NULL
    float* scores;
    float* sig;                  // or NULL
    int64_t B;
    uint64_t table_bytes, adj_bytes;
    int K, nR, pid_stride;
    unsigned max_id;
    int dbg;                     // MVIN_FOLD_DBG (measurement only; results wrong): 1 no G rows, 2 no products behind the gather, 4 none before it, 8 no gather steps
    // the gather form (mvin_fused_wpp_fold.hip): no aggregates -- every pair walks its own children and grandchildren
    const float* tables;         // [3][nE][64] TA1 | TA2 | T0A
    const float* t0;             // [nR] relation logits of aggregator (0,.) or NULL
    const int32_t* order;        // [B] or NULL: slot i of the launch works on pair order[i]
};

hipError_t launch_entity_aggregates_d32(const EntityAggArgs& a, hipStream_t st);      // mvin_fused_agg32.hip
hipError_t launch_score_l2_folded_d32(const FoldArgs& f, hipStream_t st);
hipError_t launch_score_l2_folded_gather(const FoldArgs& f, hipStream_t st);           // mvin_fused_wpp_fold.hip
size_t fused_wppfold_lds_bytes(int nR, int K);

}  // namespace mvin
