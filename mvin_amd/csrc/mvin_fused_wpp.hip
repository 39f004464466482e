// The two deepest tree levels over PROJECTED tables (mvin_gather_attn_l2_prj_fwd; formulas: include/mvin_hip.h, reference
// model.py:251-305 + aggregators.py:98-146) at dim 64 as a WAVE-PER-PARENT kernel over the duplicate-slot encoding of the adjacency.
//
// The packed-tile kernel (mvin_fused_packed.hip) is bound by its own instruction issue, not by where its rows come from (the same
// launch on a 6 000-entity KG whose tables fit every L2 takes as long per row: scripts/probe_packed_bound.py): 744 vector + 242
// scalar instructions per pair for ~100 gathered rows -- most of them the machinery that keeps 32-row MFMA tiles full (rank the rows
// by list length, deal them to waves in a snake, (id, weight) lists in LDS, segment tables, a barrier per tile).  In the
// projected-tables form nothing of a child is a matrix product any more:
//     out1[c]  = relu(TA1[x_c] + sum_k w_ck TA2[y_ck] + v)          nagg1 = sum_c (p1_c / K) out1[c]
//     nagg0    = sum_c (p0_c / K) T1[x_c] + c0 u1                   (u1 = q W1 + b1, v = q Wv + bv per parent; c0 = sum_c p0_c / K)
// so a wave can walk a parent alone, and the encoding makes the walk compact by construction: the DISTINCT slots of a row come first
// and every slot word carries the neighbour's own distinct count.  Lanes = 4 groups x 16 column chunks (lane (g, c) holds floats
// [4c, 4c + 4) of whatever row its group works on):
//   * the parent's row: one slot per lane -> both softmaxes over its slots (multiplicity-weighted) by DPP / lane swaps;
//   * passes of FOUR children, one per 16-lane group: the child's row as K / 16 slots per lane -> softmax over its distinct slots
//     inside the 16-lane DPP row -> the group gathers its child's distinct TA2 rows (slot id and weight broadcast inside the group
//     by ds_bpermute, 16 bytes per lane = a whole 256-byte row per group and load) -> + TA1 + v -> relu -> the parent's two sums;
//     a padding slot's row offset lies beyond the buffer (zeros, no memory access), a padding child weighs 0;
//   * ONE cross-group reduction per parent.
// The parents' two query terms come sixteen at a time as transposed MFMA products into 8 KB of LDS per wave; no workgroup barrier
// after the prologue, four waves per SIMD hide the dependent loads (parent row -> child rows -> grandchild rows).
#include <cstdlib>
#include <type_traits>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWppWaves = 4;
constexpr int kWppUvLd = 132;         // floats per parent of the u1 | v block in LDS (128 + 4: sixteen lanes, sixteen bank groups)
constexpr int kWppRound = 4;          // list entries of padding behind a group's K slots (the half round issued ahead of the last one)
constexpr int wpp_list_words(int K) { return 4 * 2 * (K + kWppRound); }      // per wave: 4 groups x (K + a round of padding) x (offset, weight)

__device__ __forceinline__ float wpp_bperm(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int wpp_bperm(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

__device__ __forceinline__ int wpp_xor16_imax(int v) {
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)a[0], (int)a[1]);
}
__device__ __forceinline__ int wpp_xor32_imax(int v) {
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)a[0], (int)a[1]);
}

template <int K>
__global__ __launch_bounds__(kWppWaves * 64, 3) void gather_attn_l2_wpp_kernel(FusedL2Args a) {
    constexpr int D = 64, SPL = K / 16;                  // slots of a child's row per lane of its group
    static_assert(K == 16 || K == 32, "K");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 3) & ~3;
    float* sT0 = smem;                                   // [nRp] relation logits of aggregator (0,.), or exp(logit - max) of them
    float* sT1 = sT0 + nRp;                              // [nRp] ... of aggregator (1,.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sUV = sT1 + nRp + wave * (16 * kWppUvLd + wpp_list_words(K));     // this wave's [16 parents][u1 (64) | v (64) | pad]
    // ... and its four groups' slot lists: the (row offset, weight) of every slot of the group's child, read back as BROADCASTS (all
    // sixteen lanes of a group gather the same row).  Through ds_bpermute the loads of a round went out one at a time, each behind
    // its own lane exchange and the wait for it
    unsigned* sLo = reinterpret_cast<unsigned*>(sUV + 16 * kWppUvLd) + (lane >> 4) * (K + kWppRound);      // [K + round] offsets of group g
    float* sLw = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(sUV + 16 * kWppUvLd) + 4 * (K + kWppRound)) + (lane >> 4) * (K + kWppRound);
    const int g = lane >> 4, c = lane & 15;
    const bool att0 = a.t0 != nullptr, att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)K;
    // The softmaxes run over EXP TABLES when the logits allow it: softmax is shift invariant, so exp(t[r] - max over ALL relations) serves
    // every row -- no per-row maximum (a reduction per child and two per parent) and no exp per slot.  A row whose own logits all lie far
    // below the global maximum would lose its weights to underflow: a spread above 60 (exp(-60) = 9e-27, sums of K of them stay normal)
    // takes the per-row form instead.  Every wave reads the nR logits itself (no second barrier).
    float mx0 = -INFINITY, mn0 = INFINITY, mx1 = -INFINITY, mn1 = INFINITY;
    for (int i = lane; i < a.nR; i += 64) {
        const float l0 = att0 ? a.t0[i] : 0.f, l1 = att1 ? a.t1[i] : 0.f;
        mx0 = fmaxf(mx0, l0), mn0 = fminf(mn0, l0), mx1 = fmaxf(mx1, l1), mn1 = fminf(mn1, l1);
    }
    mx0 = wave_max(mx0), mn0 = -wave_max(-mn0), mx1 = wave_max(mx1), mn1 = -wave_max(-mn1);
    const bool fast = __builtin_amdgcn_readfirstlane((mx0 - mn0 <= 60.f && mx1 - mn1 <= 60.f) ? 1 : 0) != 0;      // (NaN logits: per-row form)
    for (int i = tid; i < a.nR; i += kWppWaves * 64) {
        const float l0 = att0 ? a.t0[i] : 0.f, l1 = att1 ? a.t1[i] : 0.f;
        sT0[i] = fast ? lean_exp(fminf(l0 - mx0, 0.f)) : l0;
        sT1[i] = fast ? lean_exp(fminf(l1 - mx1, 0.f)) : l1;
    }
    __syncthreads();                                     // the only workgroup barrier: the shared logit tables

    const unsigned tbytes = (unsigned)a.table_bytes;
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.table), 0, (int)(3u * tbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out0 = __builtin_amdgcn_make_buffer_rsrc(a.nagg0, 0, (int)(a.P * D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t out1 = __builtin_amdgcn_make_buffer_rsrc(a.nagg1, 0, (int)(a.P * D * 4), 0x00020000);
    constexpr unsigned kOob = 0xFFFFFFF0u;               // a byte offset beyond every buffer: the load returns zeros, no memory access
    constexpr unsigned kPadRow = 0xFFFFFE00u;            // ... that stays beyond them (and below 2^32) with a lane's column offset added
    auto row4 = [&](unsigned off) -> float4 {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(tab, off, 0, 0);
        return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
    };
    const unsigned c16 = (unsigned)c * 16u;
    if (c < kWppRound) {                                 // the padding behind a group's K slots: beyond the buffer, no weight
        sLo[K + c] = kPadRow;
        sLw[K + c] = 0.f;
    }
    const float c0 = att0 ? invK : 1.f;                  // sum of the parent's slot weights / K

    const int64_t nbatch = (a.P + 15) >> 4;
    const int64_t nwaves = (int64_t)gridDim.x * kWppWaves;
    auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    for (int64_t batch = (int64_t)blockIdx.x * kWppWaves + wave; batch < nbatch; batch += nwaves) {
        const int64_t p_base = batch << 4;
        // ---- u1 = q W1 + b1, v = q Wv + bv of the batch's 16 parents: (u1 | v)^T[n, parent] = sum_k W[k][n] q[parent][k] on the
        //      matrix cores (A = the two 64 x 64 blocks straight from L2: a few loads per parent; B = the parents' query rows,
        //      lane (g, c = parent): 4 x 16 bytes of its row; accumulator register r of column tile ntp <-> n = 16 ntp + 4 g + r) ----
        {
            // (every address = a uniform base + ONE 32-bit lane offset + a constant: per-lane 64-bit row pointers of the two matrices,
            //  loop invariant, were hoisted out of the batch loop and held 40 registers through the walk below)
            int64_t pr = min(p_base + c, a.P - 1);
            if (a.order) pr = a.order[pr];
            const unsigned qoff = (unsigned)(pr / a.parents_per_pair) * (unsigned)(D * 4) + (unsigned)g * 16u;
            const char* qbase = reinterpret_cast<const char*>(a.q);
            float4 qb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) qb[nt] = *reinterpret_cast<const float4*>(qbase + 64 * nt + (size_t)qoff);
            unsigned woff = ((unsigned)(4 * g) * (unsigned)D + (unsigned)c) * 4u;      // W[4 g][c]
            unsigned boff = (unsigned)g * 16u;
            asm volatile("" : "+v"(woff), "+v"(boff));   // (... and the loop-invariant LOADS would be hoisted next: 144 registers)
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
                    *reinterpret_cast<float4*>(sUV + c * kWppUvLd + mat * D + 16 * ntp + 4 * g) = make_float4(acc[ntp][0], acc[ntp][1], acc[ntp][2], acc[ntp][3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int npar = (int)min((int64_t)16, a.P - p_base);
        // The walk is a chain of DEPENDENT loads (parent id -> its row -> the children's rows -> the grandchildren's table rows), ~1 us
        // each when the line comes from beyond the L2, with three waves per SIMD to hide them: the counters of the first version showed
        // the waves waiting 70 % of their time and the vector ALU 30 % busy.  So every load is issued one stage early: the NEXT parent's
        // row while this parent is walked, the NEXT pass's four child rows (adjacency + T1 + TA1) before this pass's table rows, and the
        // table rows in half rounds of four, one half in flight while the other is summed.
        auto parent_slot = [&](int j) -> unsigned {      // byte offset of slot `lane` of parent j's row
            const int64_t p = a.order ? (int64_t)__builtin_amdgcn_readfirstlane(a.order[p_base + j]) : p_base + j;
            const unsigned x0 = (unsigned)__builtin_amdgcn_readfirstlane(fused_parent_id(a, p));
            return lane < K ? (x0 * (unsigned)K + (unsigned)lane) * 4u : kOob;
        };
        unsigned npe, npr;
        {
            const unsigned so = parent_slot(0);
            npe = __builtin_amdgcn_raw_buffer_load_b32(adjE, so, 0, 0);
            npr = __builtin_amdgcn_raw_buffer_load_b32(adjR, so, 0, 0);
        }
        for (int j = 0; j < npar; ++j) {
            const int64_t p = a.order ? (int64_t)__builtin_amdgcn_readfirstlane(a.order[p_base + j]) : p_base + j;
            // ---- the parent's row: slot `lane` (distinct slots first; a padding slot has multiplicity 0) ----
            const unsigned pe = npe, pr_ = npr;
            if (j + 1 < npar) {
                const unsigned so = parent_slot(j + 1);
                npe = __builtin_amdgcn_raw_buffer_load_b32(adjE, so, 0, 0);
                npr = __builtin_amdgcn_raw_buffer_load_b32(adjR, so, 0, 0);
            }
            const int sid = (int)(pe & 0xFFFFFFu);
            const int scnt = (int)(pe >> 24);            // the neighbour's own distinct count
            const int srel = (int)(pr_ & 0xFFFFu);
            const float smul = (float)((pr_ >> 16) & 0xFFu);
            int cnt0 = __builtin_amdgcn_readfirstlane((int)(pr_ >> 24));
            cnt0 = cnt0 < 1 ? 1 : (cnt0 > K ? K : cnt0);
            float p0 = smul, p1 = smul;
            if constexpr (FAST) {
                if (att0) {
                    const float e = smul * sT0[srel];
                    p0 = e * __builtin_amdgcn_rcpf(wave_sum(e));
                }
                if (att1) {
                    const float e = smul * sT1[srel];
                    p1 = e * __builtin_amdgcn_rcpf(wave_sum(e));
                }
            } else {
                if (att0) {
                    const float l = sT0[srel];
                    const float mx = wave_max(smul > 0.f ? l : -INFINITY);
                    const float e = smul * lean_exp(fminf(l - mx, 0.f));
                    p0 = e * __builtin_amdgcn_rcpf(wave_sum(e));
                }
                if (att1) {
                    const float l = sT1[srel];
                    const float mx = wave_max(smul > 0.f ? l : -INFINITY);
                    const float e = smul * lean_exp(fminf(l - mx, 0.f));
                    p1 = e * __builtin_amdgcn_rcpf(wave_sum(e));
                }
            }
            p0 *= invK;
            p1 *= invK;
            // ---- a pass's four children, one per 16-lane group: slot ci of the parent's row -> the child's row (SPL slots per lane of
            //      the group) and its own two table rows ----
            struct Child {
                int xc, cc;
                float w0, w1;
                unsigned ce[SPL], cr[SPL];
                float4 t1row, ta1row;
            };
            auto child_load = [&](int pass) -> Child {
                Child ch;
                const int ci = 4 * pass + g;
                const bool valid = ci < cnt0;
                // (the four exchanges run with EVERY lane active and are masked by arithmetic afterwards: ds_bpermute returns 0 for a
                //  source lane that EXEC has switched off, and hipcc moves an exchange whose result is selected by `valid` into the
                //  branch of the valid lanes -- a parent's 17th child, read by group 0 from lane 16 of (invalid) group 1, weighed 0)
                ch.xc = wpp_bperm(sid, ci & 63);
                ch.cc = wpp_bperm(scnt, ci & 63);
                ch.w0 = wpp_bperm(p0, ci & 63), ch.w1 = wpp_bperm(p1, ci & 63);
                asm volatile("" : "+v"(ch.w0), "+v"(ch.w1), "+v"(ch.cc));
                const float vm = valid ? 1.f : 0.f;
                ch.w0 *= vm;
                ch.w1 *= vm;
                ch.cc = valid ? (ch.cc < 1 ? 1 : (ch.cc > K ? K : ch.cc)) : 0;
                const unsigned co = valid ? ((unsigned)ch.xc * (unsigned)K + (unsigned)(SPL * c)) * 4u : kOob;
                if constexpr (SPL == 1) {
                    ch.ce[0] = __builtin_amdgcn_raw_buffer_load_b32(adjE, co, 0, 0);
                    ch.cr[0] = __builtin_amdgcn_raw_buffer_load_b32(adjR, co, 0, 0);
                } else {
                    const u32x2 e2 = __builtin_amdgcn_raw_buffer_load_b64(adjE, co, 0, 0);
                    const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(adjR, co, 0, 0);
                    ch.ce[0] = e2[0], ch.ce[1] = e2[1], ch.cr[0] = r2[0], ch.cr[1] = r2[1];
                }
                const unsigned xo = (unsigned)ch.xc * (unsigned)(D * 4) + c16;
                ch.t1row = row4(valid ? xo : kPadRow);                // T1[x_c]   (an invalid group: beyond the buffer, zeros)
                ch.ta1row = row4(valid ? xo + tbytes : kPadRow);      // TA1[x_c]
                return ch;
            };
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            const int npass = (cnt0 + 3) >> 2;
            Child nx = child_load(0);
            for (int pass = 0; pass < npass; ++pass) {
                const Child ch = nx;
                // ---- softmax over the child's distinct slots (aggregators.py:118-146), inside the 16-lane row ----
                float wk[SPL];
                unsigned yo[SPL];
                unsigned lg[SPL];
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < SPL; ++i) {
                    const float mu = (float)((ch.cr[i] >> 16) & 0xFFu);
                    const float l = att0 ? sT0[ch.cr[i] & 0xFFFFu] : (FAST ? 1.f : 0.f);
                    // byte offset of TA2[y]; a padding slot weighs 0 and its offset lies beyond the buffer (zeros, no memory access)
                    yo[i] = mu > 0.f ? (ch.ce[i] & 0xFFFFFFu) * (unsigned)(D * 4) + 2u * tbytes : kPadRow;
                    if constexpr (FAST) {
                        wk[i] = mu * l;                  // l = exp(logit - global max)
                    } else {
                        wk[i] = mu;
                        mx = fmaxf(mx, mu > 0.f ? l : -INFINITY);
                        lg[i] = __float_as_uint(l);
                    }
                }
                if (att0) {
                    float z = 0.f;
                    if constexpr (FAST) {
#pragma unroll
                        for (int i = 0; i < SPL; ++i) z += wk[i];
                    } else {
                        mx = group_max(mx, 4);
#pragma unroll
                        for (int i = 0; i < SPL; ++i) {
                            wk[i] *= lean_exp(fminf(__uint_as_float(lg[i]) - mx, 0.f));
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
                // ---- the group's rows: (offset, weight) of slot k at list position k ----
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the previous pass's reads of the lists are done)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < SPL; ++i) {
                    sLo[SPL * c + i] = yo[i];
                    sLw[SPL * c + i] = wk[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (pass + 1 < npass) nx = child_load(pass + 1);
                // (cc is uniform inside a 16-lane group: the maximum over the four groups is two lane swaps)
                const int kmax = __builtin_amdgcn_readfirstlane(wpp_xor32_imax(wpp_xor16_imax(ch.cc)));
                f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};            // (two floats per FMA: v_pk_fma_f32)
                float4 ra[4], rb[4];
                float4 wa, wb;
                auto issue = [&](int k, float4 (&r)[4], float4& w) {
                    const uint4 o4 = *reinterpret_cast<const uint4*>(sLo + k);
                    w = *reinterpret_cast<const float4*>(sLw + k);
                    r[0] = row4(o4.x + c16), r[1] = row4(o4.y + c16), r[2] = row4(o4.z + c16), r[3] = row4(o4.w + c16);
                };
                auto sum4 = [&](const float4 (&r)[4], const float4& w) {
                    const float ws_[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x2 w2 = {ws_[t], ws_[t]};
                        s01 = __builtin_elementwise_fma(w2, f32x2{r[t].x, r[t].y}, s01);
                        s23 = __builtin_elementwise_fma(w2, f32x2{r[t].z, r[t].w}, s23);
                    }
                };
                // (both halves are issued unconditionally: slots behind a child's distinct count -- and the four list entries behind the
                //  K-th -- point beyond the buffer, and a conditional issue costs the double buffer: register copies where the paths merge
                //  and a wait for EVERY load in flight)
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
                const float4 s = make_float4(s01[0], s01[1], s23[0], s23[1]);
                // ---- out1 = relu(TA1[x_c] + sum + v); the parent's sums ----
                const float4 vv = *reinterpret_cast<const float4*>(sUV + j * kWppUvLd + D + 4 * c);
                const float4 o1 = make_float4(fmaxf(s.x + ch.ta1row.x + vv.x, 0.f), fmaxf(s.y + ch.ta1row.y + vv.y, 0.f),
                                              fmaxf(s.z + ch.ta1row.z + vv.z, 0.f), fmaxf(s.w + ch.ta1row.w + vv.w, 0.f));
                acc1 = f4_fma(ch.w1, o1, acc1);
                acc0 = f4_fma(ch.w0, ch.t1row, acc0);
            }
            acc0 = make_float4(xor32_sum(xor16_sum(acc0.x)), xor32_sum(xor16_sum(acc0.y)), xor32_sum(xor16_sum(acc0.z)), xor32_sum(xor16_sum(acc0.w)));
            acc1 = make_float4(xor32_sum(xor16_sum(acc1.x)), xor32_sum(xor16_sum(acc1.y)), xor32_sum(xor16_sum(acc1.z)), xor32_sum(xor16_sum(acc1.w)));
            if (g == 0) {
                const float4 u1 = *reinterpret_cast<const float4*>(sUV + j * kWppUvLd + 4 * c);
                const unsigned oo = (unsigned)p * (unsigned)(D * 4) + c16;
                const u32x4 v0 = {__float_as_uint(fmaf(c0, u1.x, acc0.x)), __float_as_uint(fmaf(c0, u1.y, acc0.y)),
                                  __float_as_uint(fmaf(c0, u1.z, acc0.z)), __float_as_uint(fmaf(c0, u1.w, acc0.w))};
                const u32x4 v1 = {__float_as_uint(acc1.x), __float_as_uint(acc1.y), __float_as_uint(acc1.z), __float_as_uint(acc1.w)};
                __builtin_amdgcn_raw_buffer_store_b128(v0, out0, oo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(v1, out1, oo, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next batch's u1 | v block waits for this batch's reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    };
    if (fast) run(std::true_type{});
    else run(std::false_type{});
}

bool fused_wpp_supported(int D, int K) { return D == 64 && (K == 16 || K == 32); }

size_t fused_wpp_lds_bytes(int nR, int K) {
    return ((size_t)2 * ((nR + 3) & ~3) + (size_t)kWppWaves * (16 * kWppUvLd + wpp_list_words(K))) * sizeof(float);
}

// the projected-tables form over the ENCODED adjacency with every buffer addressable by 32-bit byte offsets
bool fused_wpp_applies(const FusedL2Args& a, int D) {
    static const char* e = getenv("MVIN_L2_WPP");
    if (e && e[0] == '0') return false;                  // A/B: the packed-tile kernel
    return a.prj && fused_wpp_supported(D, a.K) && !a.probs_parent && !a.probs_child && a.adj_r && a.adj_bytes > 0 &&
           a.adj_bytes < (1ull << 31) && a.table_bytes > 0 && a.table_bytes < (1ull << 30) && (uint64_t)a.P * D * 4 < (1ull << 31) &&
           a.max_id < (1u << 24) && a.W1 && a.W2 && a.q && fused_wpp_lds_bytes(a.nR, a.K) <= 48 * 1024;      // (dynamic LDS above 48 KB would need a function attribute)
}

template <int K>
static hipError_t launch_wpp_k(const FusedL2Args& a, hipStream_t st) {
    const size_t lds = fused_wpp_lds_bytes(a.nR, K);
    static thread_local int per_cu = 0;
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(gather_attn_l2_wpp_kernel<K>), kWppWaves * 64, lds) != hipSuccess || v < 1)
            v = 4;
        per_cu = v > 8 ? 8 : v;
    }
    const int64_t nbatch = (a.P + 15) >> 4;
    const int64_t want = (nbatch + kWppWaves - 1) / kWppWaves;
    const int64_t cap = 256 * (int64_t)per_cu;           // persistent grid
    gather_attn_l2_wpp_kernel<K><<<(int)(want < cap ? want : cap), kWppWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_gather_attn_l2_wpp(const FusedL2Args& a, hipStream_t st) {
    switch (a.K) {
        case 16: return launch_wpp_k<16>(a, st);
        case 32: return launch_wpp_k<32>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
