// The two deepest tree levels over PER-ENTITY AGGREGATES (mvin_entity_aggregates + mvin_gather_attn_l2_agg_fwd; formulas:
// include/mvin_hip.h, reference model.py:251-305 + aggregators.py:98-146), dim 64, duplicate-slot encoding of the adjacency.
//
// In the projected-tables form (mvin_fused_wpp.hip) a parent x with query terms u1, v computes
//     out1[c]  = relu(TA1[x_c] + sum_k w(x_c)_k TA2[y_ck] + v)          nagg1 = sum_c (p1_c / K) out1[c]
//     nagg0    = sum_c (p0_c / K) T1[x_c] + c0 u1
// where w(e)_k -- the softmax of entity e's slots under aggregator (0,.)'s relation logits, over K -- is a property of the ENTITY e:
// the relation logits do not depend on the user (User_orient_rela: one logit per relation), so neither does the bracket's first part,
//     G[e]  = TA1[e] + sum_k w(e)_k TA2[y_ek]             and, with the SAME weights (p0_c / K = w(x)_c),
//     S0[e] = sum_k w(e)_k T1[y_ek].
// Both are tables over the entities, built once per call from the current parameters like the projected tables they are made of:
//     nagg0 = S0[x] + c0 u1                                nagg1 = sum_c (p1_c / K) relu(G[x_c] + v)
// -- a parent gathers its ~cnt distinct children's G rows (and one S0 row) instead of ~cnt + sum_c cnt_c rows of three tables: at
// BASELINE C3 (fan-out 32, ~7.4 distinct slots per row) 12 rows instead of 120, and the wave-per-parent kernel was bound by exactly
// those row requests (its texture-address pipes: 16 cycles per 64-lane 16-byte load, busy the whole launch).  The tables cost ~17 rows
// per ENTITY, so the form pays when the launch has more than ~n_entity / 6 parents; the rule of the projected tables (their
// products cost ~n_entity rows of MFMA work) is the stricter one and decides.
//
// Lanes = 4 groups x 16 column chunks in both kernels; a group owns one entity (its row: K / 16 slots per lane, softmax by DPP inside
// the 16-lane row; the (row offset, weight) list of its distinct slots through LDS, read back as broadcasts) -- no exchange between
// groups, full-wave stores.
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

// weights of a row's slots (SPL per lane of the 16-lane group that holds it; cr = relation | multiplicity << 16 | ...): multiplicity x
// softmax over the distinct slots, over K (aggregators.py:118-146); a padding slot (multiplicity 0) weighs 0
template <int SPL, bool FAST>
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
            mx = group_max(mx, 4);
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
                wk[i] *= lean_exp(fminf(lg[i] - mx, 0.f));
                z += wk[i];
            }
        }
        z = group_sum(z, 4);
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
    } else {
        const u32x2 e2 = __builtin_amdgcn_raw_buffer_load_b64(adjE, co, 0, 0);
        const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(adjR, co, 0, 0);
        ce[0] = e2[0], ce[1] = e2[1], cr[0] = r2[0], cr[1] = r2[1];
    }
}

__device__ __forceinline__ float4 agg_row4(__amdgpu_buffer_rsrc_t tab, unsigned off) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(tab, off, 0, 0);
    return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
}

// ---- S0 | G over all entities: a.table = T1 | TA1 | TA2, a.agg = S0 | G ([2][nE][64]); four entities per wave and step ----
template <int K>
__global__ __launch_bounds__(kAggWaves * 64) void entity_aggregates_kernel(FusedL2Args a) {
    constexpr int D = 64, SPL = K / 16;
    static_assert(K == 16 || K == 32, "K");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 3) & ~3;
    float* sT = smem;                                    // [nRp] relation logits of aggregator (0,.), or exp(logit - max) of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    unsigned* sLo = reinterpret_cast<unsigned*>(sT + nRp + wave * agg_list_words(K)) + g * (K + kAggPad);
    float* sLw = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(sT + nRp + wave * agg_list_words(K)) + 4 * (K + kAggPad)) + g * (K + kAggPad);
    const bool att = a.t0 != nullptr;
    const float invK = 1.f / (float)K;
    const bool fast = agg_logit_table(a.t0, a.nR, sT, tid, lane);
    __syncthreads();

    const unsigned tbytes = (unsigned)a.table_bytes;
    const char* tb = reinterpret_cast<const char*>(a.table);
    const __amdgpu_buffer_rsrc_t tabT1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb), 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t tabTA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb + a.table_bytes), 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t tabTA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb + 2 * a.table_bytes), 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t outS = __builtin_amdgcn_make_buffer_rsrc(a.agg, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t outG = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.agg) + a.table_bytes, 0, (int)tbytes, 0x00020000);
    const unsigned c16 = (unsigned)c * 16u;
    const unsigned n_entity = a.max_id + 1u;
    const unsigned nquad = (n_entity + 3u) >> 2;

    auto run = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        for (unsigned quad = blockIdx.x * kAggWaves + wave; quad < nquad; quad += gridDim.x * kAggWaves) {
            const unsigned e = quad * 4u + (unsigned)g;
            const bool valid = e < n_entity;
            unsigned ce[SPL], cr[SPL];
            agg_load_slots<SPL>(adjE, adjR, valid ? (e * (unsigned)K + (unsigned)(SPL * c)) * 4u : kAggOob, ce, cr);
            const float4 ta1 = agg_row4(tabTA1, valid ? e * (unsigned)(D * 4) + c16 : kAggPadRow);
            float wk[SPL];
            agg_row_weights<SPL, FAST>(cr, att, sT, invK, wk);
            int cc = (int)(cr[0] >> 24);                 // the row's distinct-slot count (in every slot word)
            cc = valid ? (cc < 1 ? 1 : (cc > K ? K : cc)) : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the previous step's reads of the lists are done)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
                sLo[SPL * c + i] = ((cr[i] >> 16) & 0xFFu) ? (ce[i] & 0xFFFFFFu) * (unsigned)(D * 4) : kAggPadRow;
                sLw[SPL * c + i] = wk[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int kmax = __builtin_amdgcn_readfirstlane(agg_xor32_imax(agg_xor16_imax(cc)));
            f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, g01 = {0.f, 0.f}, g23 = {0.f, 0.f};
            for (int k0 = 0; k0 < kmax; k0 += 4) {       // (slots behind a group's own count point beyond the buffers)
                const uint4 o4 = *reinterpret_cast<const uint4*>(sLo + k0);
                const float4 w4 = *reinterpret_cast<const float4*>(sLw + k0);
                const unsigned off[4] = {o4.x, o4.y, o4.z, o4.w};
                const float ws_[4] = {w4.x, w4.y, w4.z, w4.w};
                float4 r1[4], r2[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    r1[t] = agg_row4(tabT1, off[t] + c16);
                    r2[t] = agg_row4(tabTA2, off[t] + c16);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x2 w2 = {ws_[t], ws_[t]};
                    s01 = __builtin_elementwise_fma(w2, f32x2{r1[t].x, r1[t].y}, s01);
                    s23 = __builtin_elementwise_fma(w2, f32x2{r1[t].z, r1[t].w}, s23);
                    g01 = __builtin_elementwise_fma(w2, f32x2{r2[t].x, r2[t].y}, g01);
                    g23 = __builtin_elementwise_fma(w2, f32x2{r2[t].z, r2[t].w}, g23);
                }
            }
            const unsigned oo = valid ? e * (unsigned)(D * 4) + c16 : kAggOob;      // (a store beyond the buffer is dropped)
            const u32x4 vs = {__float_as_uint(s01[0]), __float_as_uint(s01[1]), __float_as_uint(s23[0]), __float_as_uint(s23[1])};
            const u32x4 vg = {__float_as_uint(g01[0] + ta1.x), __float_as_uint(g01[1] + ta1.y), __float_as_uint(g23[0] + ta1.z),
                              __float_as_uint(g23[1] + ta1.w)};
            __builtin_amdgcn_raw_buffer_store_b128(vs, outS, oo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(vg, outG, oo, 0, 0);
        }
    };
    if (fast) run(std::true_type{});
    else run(std::false_type{});
}

// ---- nagg0 = S0[x] + c0 u1, nagg1 = sum_c (p1_c / K) relu(G[x_c] + v) per parent: sixteen parents' query terms per wave and batch
//      on the matrix cores, then four parents at a time, one per 16-lane group ----
template <int K>
__global__ __launch_bounds__(kAggWaves * 64, 4) void gather_attn_l2_agg_kernel(FusedL2Args a) {
    constexpr int D = 64, SPL = K / 16;
    static_assert(K == 16 || K == 32, "K");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 3) & ~3;
    float* sT = smem;                                    // [nRp] relation logits of aggregator (1,.), or exp(logit - max) of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* sUV = sT + nRp + wave * (16 * kAggUvLd + agg_list_words(K));      // this wave's [16 parents][u1 (64) | v (64) | pad]
    unsigned* sLo = reinterpret_cast<unsigned*>(sUV + 16 * kAggUvLd) + g * (K + kAggPad);
    float* sLw = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(sUV + 16 * kAggUvLd) + 4 * (K + kAggPad)) + g * (K + kAggPad);
    const bool att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)K;
    const float c0 = a.t0 != nullptr ? invK : 1.f;       // sum of the parent's slot weights under aggregator (0,.), over K
    const bool fast = agg_logit_table(a.t1, a.nR, sT, tid, lane);
    __syncthreads();                                     // the only workgroup barrier

    const unsigned tbytes = (unsigned)a.table_bytes;
    const __amdgpu_buffer_rsrc_t aggS = __builtin_amdgcn_make_buffer_rsrc(a.agg, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t aggG = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.agg) + a.table_bytes, 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out0 = __builtin_amdgcn_make_buffer_rsrc(a.nagg0, 0, (int)(a.P * D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t out1 = __builtin_amdgcn_make_buffer_rsrc(a.nagg1, 0, (int)(a.P * D * 4), 0x00020000);
    const unsigned c16 = (unsigned)c * 16u;
    if (c < kAggPad) {                                   // the padding behind a group's K slots: beyond the buffer, no weight
        sLo[K + c] = kAggPadRow;
        sLw[K + c] = 0.f;
    }

    const int64_t nbatch = (a.P + 15) >> 4;
    const int64_t nwaves = (int64_t)gridDim.x * kAggWaves;
    auto run = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        for (int64_t batch = (int64_t)blockIdx.x * kAggWaves + wave; batch < nbatch; batch += nwaves) {
            const int64_t p_base = batch << 4;
            // ---- u1 = q W1 + b1, v = q Wv + bv of the batch's 16 parents: (u1 | v)^T[n, parent] = sum_k W[k][n] q[parent][k] on the
            //      matrix cores (A = the two 64 x 64 blocks straight from L2; B = the parents' query rows, lane (g, c = parent): 4 x 16
            //      bytes of its row; accumulator register r of column tile ntp <-> n = 16 ntp + 4 g + r).  Every address = a uniform
            //      base + ONE 32-bit lane offset + a constant ----
            {
                int64_t pr = min(p_base + c, a.P - 1);
                if (a.order) pr = a.order[pr];
                const unsigned qoff = (unsigned)(pr / a.parents_per_pair) * (unsigned)(D * 4) + (unsigned)g * 16u;
                const char* qbase = reinterpret_cast<const char*>(a.q);
                float4 qb[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) qb[nt] = *reinterpret_cast<const float4*>(qbase + 64 * nt + (size_t)qoff);
                unsigned woff = ((unsigned)(4 * g) * (unsigned)D + (unsigned)c) * 4u;      // W[4 g][c]
                unsigned boff = (unsigned)g * 16u;
                asm volatile("" : "+v"(woff), "+v"(boff));   // (loop-invariant loads are not to be hoisted out of the batch loop: 144 registers)
#pragma unroll
                for (int mat = 0; mat < 2; ++mat) {
                    const char* W = reinterpret_cast<const char*>(mat == 0 ? a.W1 : a.W2);
                    const char* bias = reinterpret_cast<const char*>(mat == 0 ? a.b1 : a.b2);
                    f32x4 acc[4];
#pragma unroll
                    for (int ntp = 0; ntp < 4; ++ntp) {
                        const float4 b = bias ? *reinterpret_cast<const float4*>(bias + 64 * ntp + (size_t)boff) : make_float4(0.f, 0.f, 0.f, 0.f);
                        acc[ntp] = f32x4{b.x, b.y, b.z, b.w};
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float bv[4] = {qb[nt].x, qb[nt].y, qb[nt].z, qb[nt].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int ntp = 0; ntp < 4; ++ntp) {      // W[16 nt + 4 g + r][16 ntp + c]
                                const float w = *reinterpret_cast<const float*>(W + ((16 * nt + r) * D + 16 * ntp) * 4 + (size_t)woff);
                                acc[ntp] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, bv[r], acc[ntp], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int ntp = 0; ntp < 4; ++ntp)
                        *reinterpret_cast<float4*>(sUV + c * kAggUvLd + mat * D + 16 * ntp + 4 * g) = make_float4(acc[ntp][0], acc[ntp][1], acc[ntp][2], acc[ntp][3]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const int nquad = (int)((min((int64_t)16, a.P - p_base) + 3) >> 2);
            for (int it = 0; it < nquad; ++it) {
                const int j = 4 * it + g;                // this group's parent of the batch
                const bool pvalid = p_base + j < a.P;
                int64_t p = min(p_base + j, a.P - 1);
                if (a.order) p = a.order[p];
                const unsigned x0 = (unsigned)fused_parent_id(a, p);
                unsigned ce[SPL], cr[SPL];
                agg_load_slots<SPL>(adjE, adjR, (x0 * (unsigned)K + (unsigned)(SPL * c)) * 4u, ce, cr);
                const float4 s0 = agg_row4(aggS, x0 * (unsigned)(D * 4) + c16);
                const float4 u1 = *reinterpret_cast<const float4*>(sUV + j * kAggUvLd + 4 * c);
                const float4 vv = *reinterpret_cast<const float4*>(sUV + j * kAggUvLd + D + 4 * c);
                float wk[SPL];
                agg_row_weights<SPL, FAST>(cr, att1, sT, invK, wk);
                int cc = (int)(cr[0] >> 24);
                cc = pvalid ? (cc < 1 ? 1 : (cc > K ? K : cc)) : 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the previous step's reads of the lists are done)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < SPL; ++i) {
                    sLo[SPL * c + i] = ((cr[i] >> 16) & 0xFFu) ? (ce[i] & 0xFFFFFFu) * (unsigned)(D * 4) : kAggPadRow;
                    sLw[SPL * c + i] = wk[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int kmax = __builtin_amdgcn_readfirstlane(agg_xor32_imax(agg_xor16_imax(cc)));
                const f32x2 v01 = {vv.x, vv.y}, v23 = {vv.z, vv.w};
                f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
                float4 ra[4], rb[4];
                float4 wa, wb;
                auto issue = [&](int k, float4 (&r)[4], float4& w) {
                    const uint4 o4 = *reinterpret_cast<const uint4*>(sLo + k);
                    w = *reinterpret_cast<const float4*>(sLw + k);
                    r[0] = agg_row4(aggG, o4.x + c16), r[1] = agg_row4(aggG, o4.y + c16);
                    r[2] = agg_row4(aggG, o4.z + c16), r[3] = agg_row4(aggG, o4.w + c16);
                };
                auto sum4 = [&](const float4 (&r)[4], const float4& w) {
                    const float ws_[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x2 o01 = f32x2{r[t].x, r[t].y} + v01, o23 = f32x2{r[t].z, r[t].w} + v23;
                        const f32x2 w2 = {ws_[t], ws_[t]};
                        a01 = __builtin_elementwise_fma(w2, f32x2{fmaxf(o01[0], 0.f), fmaxf(o01[1], 0.f)}, a01);
                        a23 = __builtin_elementwise_fma(w2, f32x2{fmaxf(o23[0], 0.f), fmaxf(o23[1], 0.f)}, a23);
                    }
                };
                // (half rounds of four children, one in flight while the other is summed; both are issued unconditionally: slots
                //  behind a parent's distinct count -- and the four list entries behind the K-th -- point beyond the buffer)
                issue(0, ra, wa);
                for (int k0 = 0; k0 < kmax; k0 += 8) {
                    issue(k0 + 4, rb, wb);
                    __builtin_amdgcn_sched_barrier(0);
                    sum4(ra, wa);
                    __builtin_amdgcn_sched_barrier(0);
                    issue(k0 + 8, ra, wa);
                    __builtin_amdgcn_sched_barrier(0);
                    sum4(rb, wb);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const unsigned oo = pvalid ? (unsigned)p * (unsigned)(D * 4) + c16 : kAggOob;      // (a store beyond the buffer is dropped)
                const u32x4 o0 = {__float_as_uint(fmaf(c0, u1.x, s0.x)), __float_as_uint(fmaf(c0, u1.y, s0.y)),
                                  __float_as_uint(fmaf(c0, u1.z, s0.z)), __float_as_uint(fmaf(c0, u1.w, s0.w))};
                const u32x4 o1 = {__float_as_uint(a01[0]), __float_as_uint(a01[1]), __float_as_uint(a23[0]), __float_as_uint(a23[1])};
                __builtin_amdgcn_raw_buffer_store_b128(o0, out0, oo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, out1, oo, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next batch's u1 | v block waits for this batch's reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    if (fast) run(std::true_type{});
    else run(std::false_type{});
}

bool fused_agg_supported(int D, int K) { return D == 64 && (K == 16 || K == 32); }

size_t fused_agg_lds_bytes(int nR, int K) {
    return ((size_t)((nR + 3) & ~3) + (size_t)kAggWaves * (16 * kAggUvLd + agg_list_words(K))) * sizeof(float);
}

// the projected-tables form over the ENCODED adjacency with every buffer addressable by 32-bit byte offsets, aggregates given
bool fused_agg_applies(const FusedL2Args& a, int D) {
    return a.prj && fused_agg_supported(D, a.K) && !a.probs_parent && !a.probs_child && a.adj_r && a.adj_bytes > 0 &&
           a.adj_bytes < (1ull << 31) && a.table_bytes > 0 && a.table_bytes < (1ull << 30) && (uint64_t)a.P * D * 4 < (1ull << 31) &&
           a.max_id < (1u << 24) && a.nR > 0 && fused_agg_lds_bytes(a.nR, a.K) <= 64 * 1024;
}

template <int K>
static hipError_t launch_entity_aggregates_k(const FusedL2Args& a, hipStream_t st) {
    const size_t lds = ((size_t)((a.nR + 3) & ~3) + (size_t)kAggWaves * agg_list_words(K)) * sizeof(float);
    const int64_t nquad = ((int64_t)a.max_id + 4) >> 2;
    const int64_t want = (nquad + kAggWaves - 1) / kAggWaves;
    const int64_t cap = 256 * 8;
    entity_aggregates_kernel<K><<<(int)(want < cap ? want : cap), kAggWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_entity_aggregates(const FusedL2Args& a, hipStream_t st) {
    switch (a.K) {
        case 16: return launch_entity_aggregates_k<16>(a, st);
        case 32: return launch_entity_aggregates_k<32>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int K>
static hipError_t launch_agg_k(const FusedL2Args& a, hipStream_t st) {
    const size_t lds = fused_agg_lds_bytes(a.nR, K);
    static thread_local int per_cu = 0;
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(gather_attn_l2_agg_kernel<K>), kAggWaves * 64, lds) != hipSuccess || v < 1)
            v = 4;
        per_cu = v > 8 ? 8 : v;
    }
    const int64_t nbatch = (a.P + 15) >> 4;
    const int64_t want = (nbatch + kAggWaves - 1) / kAggWaves;
    const int64_t cap = 256 * (int64_t)per_cu;           // persistent grid
    gather_attn_l2_agg_kernel<K><<<(int)(want < cap ? want : cap), kAggWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_gather_attn_l2_agg(const FusedL2Args& a, hipStream_t st) {
    switch (a.K) {
        case 16: return launch_agg_k<16>(a, st);
        case 32: return launch_agg_k<32>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
