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
#include "mvin_fused_agg.h"

namespace mvin {

// ---- outS | outG over all entities (four entities per wave and step): the aggregates form S0 | G (tabS = T1, no selfS), the
//      folded-tail form H0 | G (tabS = TA1, selfS = T0A) ----
template <int K>
__global__ __launch_bounds__(kAggWaves * 64) void entity_aggregates_kernel(EntityAggArgs a) {
    constexpr int D = 64, SPL = K / 16;
    static_assert(K == 16 || K == 32 || K == 64, "K");
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

    const int tbytes = (int)a.table_bytes;
    const __amdgpu_buffer_rsrc_t tabS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.tabS), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t tabG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.tabG), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t selfS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.selfS), 0, a.selfS ? tbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t selfG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.selfG), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t outS = __builtin_amdgcn_make_buffer_rsrc(a.outS, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t outG = __builtin_amdgcn_make_buffer_rsrc(a.outG, 0, tbytes, 0x00020000);
    const unsigned c16 = (unsigned)c * 16u;
    const unsigned n_entity = (unsigned)a.n_entity;
    const unsigned nquad = (n_entity + 3u) >> 2;

    auto run = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        for (unsigned quad = blockIdx.x * kAggWaves + wave; quad < nquad; quad += gridDim.x * kAggWaves) {
            const unsigned e = quad * 4u + (unsigned)g;
            const bool valid = e < n_entity;
            unsigned ce[SPL], cr[SPL];
            agg_load_slots<SPL>(adjE, adjR, valid ? (e * (unsigned)K + (unsigned)(SPL * c)) * 4u : kAggOob, ce, cr);
            const unsigned so = valid ? e * (unsigned)(D * 4) + c16 : kAggPadRow;
            const float4 sg = agg_row4(selfG, so);
            const float4 ss = agg_row4(selfS, so);       // (no selfS: an empty buffer, zeros)
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
                    r1[t] = agg_row4(tabS, off[t] + c16);
                    r2[t] = agg_row4(tabG, off[t] + c16);
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
            const u32x4 vs = {__float_as_uint(s01[0] + ss.x), __float_as_uint(s01[1] + ss.y), __float_as_uint(s23[0] + ss.z),
                              __float_as_uint(s23[1] + ss.w)};
            const u32x4 vg = {__float_as_uint(g01[0] + sg.x), __float_as_uint(g01[1] + sg.y), __float_as_uint(g23[0] + sg.z),
                              __float_as_uint(g23[1] + sg.w)};
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
__global__ __launch_bounds__(kAggWaves * 64, K == 64 ? 3 : 4) void gather_attn_l2_agg_kernel(FusedL2Args a) {
    constexpr int D = 64, SPL = K / 16;
    static_assert(K == 16 || K == 32 || K == 64, "K");
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
    const bool fold = a.fold != 0;
    const float c0 = fold ? 1.f : a.t0 != nullptr ? invK : 1.f;      // sum of the parent's slot weights under aggregator (0,.), over K (folded: inside Wq)
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
            // (parent c of the batch in every group: its slot of the launch and its entity id, known before the products -- the four
            //  steps below take theirs by a lane exchange, and each step's adjacency row is requested one step early)
            int64_t pr = min(p_base + c, a.P - 1);
            if (a.order) pr = a.order[pr];
            const int x0c = fused_parent_id(a, pr);
            {
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
            struct Quad {
                unsigned p;
                unsigned ce[SPL], cr[SPL];
                float4 s0;
            };
            auto quad_load = [&](int it) -> Quad {       // group g's parent of step `it`: j = 4 it + g
                Quad qd;
                const int src = ((lane & 48) + 4 * it + g) << 2;
                qd.p = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)pr);
                const unsigned x0 = (unsigned)__builtin_amdgcn_ds_bpermute(src, x0c);
                agg_load_slots<SPL>(adjE, adjR, (x0 * (unsigned)K + (unsigned)(SPL * c)) * 4u, qd.ce, qd.cr);
                qd.s0 = agg_row4(aggS, x0 * (unsigned)(D * 4) + c16);
                return qd;
            };
            Quad nx = quad_load(0);
            for (int it = 0; it < nquad; ++it) {
                const int j = 4 * it + g;                // this group's parent of the batch
                const bool pvalid = p_base + j < a.P;
                const Quad qd = nx;
                if (it + 1 < nquad) nx = quad_load(it + 1);
                const unsigned p = qd.p;
                const unsigned (&ce)[SPL] = qd.ce;
                const unsigned (&cr)[SPL] = qd.cr;
                const float4 s0 = qd.s0;
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
                const unsigned oo = pvalid ? p * (unsigned)(D * 4) + c16 : kAggOob;      // (a store beyond the buffer is dropped)
                float4 r0 = make_float4(fmaf(c0, u1.x, s0.x), fmaf(c0, u1.y, s0.y), fmaf(c0, u1.z, s0.z), fmaf(c0, u1.w, s0.w));
                float4 r1 = make_float4(a01[0], a01[1], a23[0], a23[1]);
                if (fold) {      // folded-tail form: out0 = relu(H0[x] + q Wq + bq) and Z2 = out0 + nagg1 leave the kernel
                    r0 = make_float4(fmaxf(r0.x, 0.f), fmaxf(r0.y, 0.f), fmaxf(r0.z, 0.f), fmaxf(r0.w, 0.f));
                    r1 = make_float4(r1.x + r0.x, r1.y + r0.y, r1.z + r0.z, r1.w + r0.w);
                }
                const u32x4 o0 = {__float_as_uint(r0.x), __float_as_uint(r0.y), __float_as_uint(r0.z), __float_as_uint(r0.w)};
                const u32x4 o1 = {__float_as_uint(r1.x), __float_as_uint(r1.y), __float_as_uint(r1.z), __float_as_uint(r1.w)};
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

// ---- folded-tail form in ONE launch (mvin_score_l2_folded_fwd): everything above key addressing for a pair, from its item id and
//      query row to its score.  The pair kernel above with the tail's products on its own matrix-core scheme: a batch of 16 pairs per
//      wave, every product TRANSPOSED -- (X W)^T[n, pair] = sum_k W[k][n] X[pair][k], A = the 64 x 64 block straight from L2, B = the
//      pairs' rows, lane (g, c = pair) holding floats [16 nt + 4 g, + 4) of its row -- so an accumulator (lane (g, c): n = 16 ntp + 4 g + r)
//      IS the next product's B operand, and the rows a group leaves in LDS (out0, Z2: row-major per pair) are read back as B operands
//      by one 16-byte read per k tile.  Nothing of a pair but its score (and item embedding) is written:
//        t = q Wq + bq ; v = q Wv + bv ; m = q Wqm                                        (matrix cores; t, v -> LDS, m stays in registers)
//        out0 = relu(H0[x] + t) ; Z2 = out0 + sum_c (p1_c / K) relu(G[x_c] + v)           (four pairs at a time, one per 16-lane group)
//        out2 = relu(Z2 A1 + a1) ; item = M0[x] + m + out0 Wm1 + out2 Wm2 + bm ; score = <user_o, item>       (matrix cores)
template <int K>
__global__ __launch_bounds__(kAggWaves * 64, 3) void score_l2_folded_kernel(FoldArgs a) {
    constexpr int D = 64, SPL = K / 16;
    static_assert(K == 16 || K == 32 || K == 64, "K");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 3) & ~3;
    float* sT = smem;                                    // [nRp] relation logits of aggregator (1,.), or exp(logit - max) of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* sUV = sT + nRp + wave * (16 * kAggUvLd + agg_list_words(K));      // this wave's [16 pairs][t -> out0 (64) | v -> Z2 (64) | pad]
    unsigned* sLo = reinterpret_cast<unsigned*>(sUV + 16 * kAggUvLd) + g * (K + kAggPad);
    float* sLw = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(sUV + 16 * kAggUvLd) + 4 * (K + kAggPad)) + g * (K + kAggPad);
    const bool att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)K;
    const bool fast = agg_logit_table(a.t1, a.nR, sT, tid, lane);
    __syncthreads();                                     // the only workgroup barrier

    const int tbytes = (int)a.table_bytes;
    const __amdgpu_buffer_rsrc_t aggS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.agg), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t aggG = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.agg)) + a.table_bytes, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
    const unsigned c16 = (unsigned)c * 16u;
    if (c < kAggPad) {                                   // the padding behind a group's K slots: beyond the buffer, no weight
        sLo[K + c] = kAggPadRow;
        sLw[K + c] = 0.f;
    }

    const int64_t nbatch = (a.B + 15) >> 4;
    const int64_t nwaves = (int64_t)gridDim.x * kAggWaves;
    auto run = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        for (int64_t batch = (int64_t)blockIdx.x * kAggWaves + wave; batch < nbatch; batch += nwaves) {
            const int64_t p_base = batch << 4;
            // pair c of the batch in every group: its row of the launch and its entity id
            const int64_t pr = min(p_base + c, a.B - 1);
            const bool cvalid = p_base + c < a.B;
            unsigned x0u = (unsigned)a.items[pr * a.pid_stride];
            x0u = x0u < a.max_id ? x0u : a.max_id;
            const int x0c = (int)x0u;
            unsigned woff = ((unsigned)(4 * g) * (unsigned)D + (unsigned)c * 4u) * 4u;      // Wperm[4 g][c][0]
            unsigned boff = (unsigned)g * 16u;
            unsigned roff = (unsigned)pr * (unsigned)(D * 4) + (unsigned)g * 16u;     // floats [4 g, 4 g + 4) of row `pr` of a [B][64] array
            asm volatile("" : "+v"(woff), "+v"(boff));   // (loop-invariant loads are not to be hoisted out of the batch loop)
            // THREE products as one chain of 12 k tiles, acc_j[ntp] += sum over k of W_j[k][16 ntp + c] b_j[nt] (64 MFMAs each).  A from the
            // regrouped copies of the blocks (Wperm: the four column tiles' values of a lane's k in one 16-byte load -- buffer loads, so that
            // they are known not to alias the LDS traffic around them) through a ring of three register buffers: the loads of k tiles
            // s + 1 and s + 2 are in flight under the MFMAs of tile s, across the products' boundaries.  `between(j)` runs behind product j.
            auto chain3 = [&](const float* W0p, const float* W1p, const float* W2p, const f32x4 (&b0)[4], const f32x4 (&b1)[4], const f32x4 (&b2)[4],
                              f32x4 (&acc0)[4], f32x4 (&acc1)[4], f32x4 (&acc2)[4], auto&& between) {
                asm volatile("" : "+s"(W0p), "+s"(W1p), "+s"(W2p));      // (descriptors and k-tile bases are not to live in SGPRs across the batch loop: 1 100 spills)
                const __amdgpu_buffer_rsrc_t wr[3] = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W0p), 0, D * D * 4, 0x00020000),
                                                      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1p), 0, D * D * 4, 0x00020000),
                                                      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2p), 0, D * D * 4, 0x00020000)};
                f32x4 ring[3][4];
                auto load = [&](auto s_) {
                    constexpr int s = decltype(s_)::value, j = s / 4, nt = s % 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr[j], woff + (unsigned)(r * D * 4), nt * 16 * D * 4, 0);
                        ring[s % 3][r] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
                    }
                };
                load(std::integral_constant<int, 0>{});
                load(std::integral_constant<int, 1>{});
                static_for<12>([&](auto s_) {
                    constexpr int s = decltype(s_)::value, j = s / 4, nt = s % 4;
                    if constexpr (s + 2 < 12) load(std::integral_constant<int, s + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 (&b)[4] = j == 0 ? b0 : j == 1 ? b1 : b2;
                    f32x4 (&acc)[4] = j == 0 ? acc0 : j == 1 ? acc1 : acc2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int ntp = 0; ntp < 4; ++ntp) acc[ntp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s % 3][r][ntp], b[nt][r], acc[ntp], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (nt == 3) between(std::integral_constant<int, j>{});
                });
            };
            auto bias4 = [&](const float* bp, f32x4 (&acc)[4]) {
                const char* bias = reinterpret_cast<const char*>(bp);
#pragma unroll
                for (int ntp = 0; ntp < 4; ++ntp)
                    acc[ntp] = bias ? *reinterpret_cast<const f32x4*>(bias + 64 * ntp + (size_t)boff) : f32x4{0.f, 0.f, 0.f, 0.f};
            };
            f32x4 qm[4], qb[4];
            {
                const char* qbase = reinterpret_cast<const char*>(a.q);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) qb[nt] = *reinterpret_cast<const f32x4*>(qbase + 64 * nt + (size_t)roff);
                f32x4 at[4], av[4];
                bias4(a.bq, at);
                bias4(a.bv, av);
                bias4(a.bm, qm);
                if (!(a.dbg & 4))
                    chain3(a.Wq, a.Wv, a.Wqm, qb, qb, qb, at, av, qm, [&](auto j_) {
                        constexpr int j = decltype(j_)::value;
                        if constexpr (j < 2) {           // t, then v: into the pairs' LDS rows
                            const f32x4 (&acc)[4] = j == 0 ? at : av;
#pragma unroll
                            for (int ntp = 0; ntp < 4; ++ntp) *reinterpret_cast<f32x4*>(sUV + c * kAggUvLd + j * D + 16 * ntp + 4 * g) = acc[ntp];
                        }
                    });
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const int nquad = (a.dbg & 8) ? 0 : (int)((min((int64_t)16, a.B - p_base) + 3) >> 2);
            struct Quad {
                unsigned ce[SPL], cr[SPL];
                float4 s0;
            };
            auto quad_load = [&](int it) -> Quad {       // group g's pair of step `it`: j = 4 it + g
                Quad qd;
                const unsigned x0 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane & 48) + 4 * it + g) << 2, x0c);
                agg_load_slots<SPL>(adjE, adjR, (x0 * (unsigned)K + (unsigned)(SPL * c)) * 4u, qd.ce, qd.cr);
                qd.s0 = agg_row4(aggS, x0 * (unsigned)(D * 4) + c16);
                return qd;
            };
            Quad nx = quad_load(0);
            for (int it = 0; it < nquad; ++it) {
                const int j = 4 * it + g;                // this group's pair of the batch
                const bool pvalid = p_base + j < a.B;
                const Quad qd = nx;
                if (it + 1 < nquad) nx = quad_load(it + 1);
                const unsigned (&ce)[SPL] = qd.ce;
                const unsigned (&cr)[SPL] = qd.cr;
                float* rowT = sUV + j * kAggUvLd + 4 * c;
                const float4 tt = *reinterpret_cast<const float4*>(rowT);
                const float4 vv = *reinterpret_cast<const float4*>(rowT + D);
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
                const int kmax = (a.dbg & 1) ? 0 : __builtin_amdgcn_readfirstlane(agg_xor32_imax(agg_xor16_imax(cc)));
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
                // out0 = relu(H0[x] + t) ; Z2 = out0 + nagg1: this pair's two rows of the block, in place of t and v
                const float4 o0 = make_float4(fmaxf(qd.s0.x + tt.x, 0.f), fmaxf(qd.s0.y + tt.y, 0.f), fmaxf(qd.s0.z + tt.z, 0.f),
                                              fmaxf(qd.s0.w + tt.w, 0.f));
                *reinterpret_cast<float4*>(rowT) = o0;
                *reinterpret_cast<float4*>(rowT + D) = make_float4(a01[0] + o0.x, a01[1] + o0.y, a23[0] + o0.z, a23[1] + o0.w);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- out2 = relu(Z2 A1 + a1) ; item = M0[x] + m + out0 Wm1 + out2 Wm2 (+ bm, in m) ; score ----
            {
                f32x4 zb[4], ob[4], o2[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) zb[nt] = *reinterpret_cast<const f32x4*>(sUV + c * kAggUvLd + D + 16 * nt + 4 * g);
                bias4(a.a1, o2);
                // Z2 A1 -> out2 (its accumulators, through the ReLU, are the third product's B operand); out0 Wm1 and out2 Wm2 into the item row
                if (!(a.dbg & 2))
                    chain3(a.A1, a.Wm1, a.Wm2, zb, ob, o2, o2, qm, qm, [&](auto j_) {
                        if constexpr (decltype(j_)::value == 0) {
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                o2[nt] = f32x4{fmaxf(o2[nt][0], 0.f), fmaxf(o2[nt][1], 0.f), fmaxf(o2[nt][2], 0.f), fmaxf(o2[nt][3], 0.f)};
                                ob[nt] = *reinterpret_cast<const f32x4*>(sUV + c * kAggUvLd + 16 * nt + 4 * g);      // out0: the second product's B operand
                            }
                        }
                    });
                const bool uo_is_q = a.user_o == a.q;     // (default wiring: the query IS user_o -- its rows are in registers since the batch's top)
                float part = 0.f;
#pragma unroll
                for (int ntp = 0; ntp < 4; ++ntp) {
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.M0) + (size_t)x0u * (D * 4) + 64 * ntp + (size_t)boff);
                    const f32x4 uo = uo_is_q ? qb[ntp] : *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.user_o) + 64 * ntp + (size_t)roff);
                    const f32x4 it4 = qm[ntp] + m0;
                    if (a.item_emb && cvalid) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.item_emb) + 64 * ntp + (size_t)roff) = it4;
                    part += it4[0] * uo[0] + it4[1] * uo[1] + it4[2] * uo[2] + it4[3] * uo[3];
                }
                part = xor32_sum(xor16_sum(part));       // the four groups' quarters of pair c's row
                if (g == 0 && cvalid) {
                    a.scores[pr] = part;
                    if (a.sig) a.sig[pr] = 1.f / (1.f + expf(-part));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next batch's block waits for this batch's reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    if (fast) run(std::true_type{});
    else run(std::false_type{});
}

template <int K>
static hipError_t launch_fold_k(const FoldArgs& a, hipStream_t st) {
    const size_t lds = fused_agg_lds_bytes(a.nR, K);
    static thread_local int per_cu = 0;
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(score_l2_folded_kernel<K>), kAggWaves * 64, lds) != hipSuccess || v < 1)
            v = 4;
        per_cu = v > 8 ? 8 : v;
        if (const char* e = getenv("MVIN_FOLD_WGS")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;      // (measurement: workgroups per CU)
    }
    const int64_t nbatch = (a.B + 15) >> 4;
    const int64_t want = (nbatch + kAggWaves - 1) / kAggWaves;
    const int64_t cap = 256 * (int64_t)per_cu;           // persistent grid
    score_l2_folded_kernel<K><<<(int)(want < cap ? want : cap), kAggWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_score_l2_folded(const float* agg, const float* M0, const int32_t* adj_e, const int32_t* adj_r, const int32_t* items, int pid_stride,
                                  const float* t1, const float* q, const float* user_o, const float* Wq, const float* bq, const float* Wv,
                                  const float* bv, const float* Wqm, const float* A1, const float* a1, const float* Wm1, const float* Wm2,
                                  const float* bm, float* item_emb, float* scores, float* sig, int64_t B, int K, int D, int nR, int n_entity,
                                  hipStream_t st) {
    FoldArgs f{};
    f.agg = agg, f.M0 = M0, f.adj_e = adj_e, f.adj_r = adj_r, f.items = items, f.pid_stride = pid_stride, f.t1 = t1, f.q = q, f.user_o = user_o;
    f.Wq = Wq, f.bq = bq, f.Wv = Wv, f.bv = bv, f.Wqm = Wqm, f.A1 = A1, f.a1 = a1, f.Wm1 = Wm1, f.Wm2 = Wm2, f.bm = bm;
    f.item_emb = item_emb, f.scores = scores, f.sig = sig, f.B = B, f.K = K, f.nR = nR, f.max_id = (unsigned)(n_entity - 1);
    f.table_bytes = (uint64_t)n_entity * (uint64_t)D * 4, f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    static const char* dbg = getenv("MVIN_FOLD_DBG");
    f.dbg = dbg ? atoi(dbg) : 0;
    if (D == 32) return launch_score_l2_folded_d32(f, st);
    switch (K) {
        case 16: return launch_fold_k<16>(f, st);
        case 32: return launch_fold_k<32>(f, st);
        case 64: return launch_fold_k<64>(f, st);
        default: return hipErrorInvalidValue;
    }
}

bool fused_agg_supported(int D, int K) { return D == 64 && (K == 16 || K == 32 || K == 64); }
bool fused_fold_supported(int D, int K) { return fused_agg_supported(D, K) || (D == 32 && (K == 16 || K == 32)); }
size_t fused_fold_d32_lds_bytes(int nR, int K);
size_t fused_fold_lds_bytes(int D, int nR, int K) { return D == 32 ? fused_fold_d32_lds_bytes(nR, K) : fused_agg_lds_bytes(nR, K); }

size_t fused_agg_lds_bytes(int nR, int K) {
    return ((size_t)((nR + 3) & ~3) + (size_t)kAggWaves * (16 * kAggUvLd + agg_list_words(K))) * sizeof(float);
}

// the projected-tables form over the ENCODED adjacency with every buffer addressable by 32-bit byte offsets, aggregates given
bool fused_agg_applies(const FusedL2Args& a, int D) {
    return a.prj && fused_agg_supported(D, a.K) && !a.probs_parent && !a.probs_child && a.adj_r && a.adj_bytes > 0 &&
           a.adj_bytes < (1ull << 31) && a.table_bytes > 0 && a.table_bytes < (1ull << 30) && (uint64_t)a.P * D * 4 < (1ull << 31) &&
           a.max_id < (1u << 24) && a.nR > 0 && fused_agg_lds_bytes(a.nR, a.K) <= 48 * 1024;      // (dynamic LDS above 48 KB needs a function attribute: such relation counts keep the other kernels)
}

template <int K>
static hipError_t launch_entity_aggregates_k(const EntityAggArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)((a.nR + 3) & ~3) + (size_t)kAggWaves * agg_list_words(K)) * sizeof(float);
    const int64_t nquad = ((int64_t)a.n_entity + 3) >> 2;
    const int64_t want = (nquad + kAggWaves - 1) / kAggWaves;
    const int64_t cap = 256 * 8;
    entity_aggregates_kernel<K><<<(int)(want < cap ? want : cap), kAggWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_entity_aggregates(const EntityAggArgs& a, hipStream_t st) {
    if (a.D == 32) return launch_entity_aggregates_d32(a, st);
    switch (a.K) {
        case 16: return launch_entity_aggregates_k<16>(a, st);
        case 32: return launch_entity_aggregates_k<32>(a, st);
        case 64: return launch_entity_aggregates_k<64>(a, st);
        default: return hipErrorInvalidValue;
    }
}

// ---- folded-tail form: the per-call parameter block behind its four tables and two aggregates (mvin_fold_tables) ----
//   Wstack [4][D][D] = W1.A0 | W2.A0 | W0.A0 | W0.Wm0     (the B operands of the table build: TA1 | TA2 | T0A | M0)
//   Wv [D][D] = (W1 + c W2).A0      bv [D] = (b1 + c b2).A0 + a0        (the children's query term, as in the projected-tables form)
//   Wq [D][D] = (W0 + c W1).A0      bq [D] = (b0 + c b1).A0 + a0        ((ev0 + nagg0) A0 + a0 = H0[x] + q Wq + bq)
//   bm [D]    = bmix + b0.Wm0                                           (ev0 Wm0 = M0[x] + q W0.Wm0 + b0.Wm0)
// Wm0 = the first D rows of the mix-hop combiner Wmix [3 D, D].  Block i < D: row i of the five products; block D: the biases.
// Behind them, for the single-launch kernels (D = 64 / 32): Wperm [6][D][D] = Wq | Wv | Wqm | A1 | Wm1 | Wm2 with the columns of a row
// regrouped, Wperm[k][c][ntp] = W[k][16 ntp + c] -- the A values a lane needs for one k (one per 16-column tile) in ONE 16- / 8-byte load.
__global__ void fold_prepare_kernel(const float* __restrict__ W0, const float* __restrict__ b0, const float* __restrict__ W1,
                                    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                                    const float* __restrict__ A0, const float* __restrict__ a0, const float* __restrict__ Wmix,
                                    const float* __restrict__ bmix, const float* __restrict__ A1, float c, int D, float* __restrict__ blk) {
    const int i = blockIdx.x, j = threadIdx.x;
    float* Wstack = blk;
    float* Wv = blk + (size_t)4 * D * D;
    float* Wq = Wv + (size_t)D * D;
    float* bv = Wq + (size_t)D * D;
    float* bq = bv + D;
    float* bm = bq + D;
    if (j >= D) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, sm = 0.f;
    for (int k0 = 0; k0 < D; k0 += 8) {                  // (eight steps' operands loaded before their FMAs)
        float av[8], mv[8], x0[8], x1[8], x2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            av[k] = A0[(size_t)(k0 + k) * D + j];
            mv[k] = Wmix[(size_t)(k0 + k) * D + j];
            x0[k] = i < D ? W0[(size_t)i * D + k0 + k] : (b0 ? b0[k0 + k] : 0.f);
            x1[k] = i < D ? W1[(size_t)i * D + k0 + k] : (b1 ? b1[k0 + k] : 0.f);
            x2[k] = i < D ? W2[(size_t)i * D + k0 + k] : (b2 ? b2[k0 + k] : 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s0 = fmaf(x0[k], av[k], s0);
            s1 = fmaf(x1[k], av[k], s1);
            s2 = fmaf(x2[k], av[k], s2);
            sm = fmaf(x0[k], mv[k], sm);
        }
    }
    if (i < D) {
        Wstack[(size_t)i * D + j] = s1;
        Wstack[(size_t)D * D + (size_t)i * D + j] = s2;
        Wstack[(size_t)2 * D * D + (size_t)i * D + j] = s0;
        Wstack[(size_t)3 * D * D + (size_t)i * D + j] = sm;
        Wv[(size_t)i * D + j] = fmaf(c, s2, s1);
        Wq[(size_t)i * D + j] = fmaf(c, s1, s0);
        if (D == 64 || D == 32) {
            float* Wperm = bm + D;
            const size_t o = (size_t)i * D + (size_t)(j & 15) * (D / 16) + (size_t)(j >> 4);
            const size_t DD = (size_t)D * D;
            Wperm[o] = fmaf(c, s1, s0);
            Wperm[DD + o] = fmaf(c, s2, s1);
            Wperm[2 * DD + o] = sm;
            Wperm[3 * DD + o] = A1[(size_t)i * D + j];
            Wperm[4 * DD + o] = Wmix[DD + (size_t)i * D + j];
            Wperm[5 * DD + o] = Wmix[2 * DD + (size_t)i * D + j];
        }
    } else {
        const float av0 = a0 ? a0[j] : 0.f;
        bv[j] = fmaf(c, s2, s1) + av0;
        bq[j] = fmaf(c, s1, s0) + av0;
        bm[j] = (bmix ? bmix[j] : 0.f) + sm;
    }
}

hipError_t launch_fold_prepare(const float* W0, const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* A0,
                               const float* a0, const float* Wmix, const float* bmix, const float* A1, float c, int D, float* blk, hipStream_t st) {
    fold_prepare_kernel<<<D + 1, D < 64 ? 64 : D, 0, st>>>(W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, c, D, blk);
    return hipGetLastError();
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
        case 64: return launch_agg_k<64>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
