// The per-entity aggregates H0 | G and the folded-tail launch (mvin_fold_tables / mvin_score_l2_folded_fwd; formulas: include/mvin_hip.h,
// algebra and kernel design: mvin_fused_agg.hip) at DIM 32, K in {16, 32} -- BASELINE config C2's shape.  Same scheme on another lane
// geometry: a row is 128 bytes = 8 lanes x 16 bytes, so a wave holds EIGHT rows at a time (8 groups x 8 column chunks: eight entities per
// step of the aggregates kernel, eight pairs per gather step -- two steps per batch of 16 pairs), a row's K slots are K / 8 per lane and
// its softmax runs inside 8 lanes; the products have two 16-column tiles and two k tiles (8 MFMAs per k tile, a chain of three products
// = 6 k tiles through the ring of three register buffers; a lane's A values of one k = ONE 8-byte load of the regrouped block).
#include "mvin_fused_agg.h"

namespace mvin {

constexpr int kA32Ld = 68;            // floats per pair of the t | v block in LDS (64 + 4)
constexpr int a32_list_words(int K) { return 8 * 2 * (K + kAggPad); }      // per wave: 8 groups x (K + padding) x (offset, weight)

// max over the eight 8-lane groups of a value that is uniform inside each
__device__ __forceinline__ int a32_groups_imax(int v) {
    const int m = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);      // row_mirror: lane i <-> 15 - i, the other group of the row
    return agg_xor32_imax(agg_xor16_imax(max(v, m)));
}

template <int K>
__global__ __launch_bounds__(kAggWaves * 64) void entity_aggregates_d32_kernel(EntityAggArgs a) {
    constexpr int D = 32, SPL = K / 8;
    static_assert(K == 16 || K == 32, "K");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 3) & ~3;
    float* sT = smem;                                    // [nRp] relation logits of aggregator (0,.), or exp(logit - max) of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 3, c = lane & 7;
    unsigned* sLo = reinterpret_cast<unsigned*>(sT + nRp + wave * a32_list_words(K)) + g * (K + kAggPad);
    float* sLw = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(sT + nRp + wave * a32_list_words(K)) + 8 * (K + kAggPad)) + g * (K + kAggPad);
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
    const unsigned noct = (n_entity + 7u) >> 3;

    auto run = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        for (unsigned oct = blockIdx.x * kAggWaves + wave; oct < noct; oct += gridDim.x * kAggWaves) {
            const unsigned e = oct * 8u + (unsigned)g;
            const bool valid = e < n_entity;
            unsigned ce[SPL], cr[SPL];
            agg_load_slots<SPL>(adjE, adjR, valid ? (e * (unsigned)K + (unsigned)(SPL * c)) * 4u : kAggOob, ce, cr);
            const unsigned so = valid ? e * (unsigned)(D * 4) + c16 : kAggPadRow;
            const float4 sg = agg_row4(selfG, so);
            const float4 ss = agg_row4(selfS, so);       // (no selfS: an empty buffer, zeros)
            float wk[SPL];
            agg_row_weights<SPL, FAST, 3>(cr, att, sT, invK, wk);
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
            const int kmax = __builtin_amdgcn_readfirstlane(a32_groups_imax(cc));
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

template <int K>
__global__ __launch_bounds__(kAggWaves * 64, 3) void score_l2_folded_d32_kernel(FoldArgs a) {
    constexpr int D = 32, SPL = K / 8, NT = 2;
    static_assert(K == 16 || K == 32, "K");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 3) & ~3;
    float* sT = smem;                                    // [nRp] relation logits of aggregator (1,.), or exp(logit - max) of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;              // the products' lane roles: k / column quarter g, pair c of the batch
    const int g8 = lane >> 3, c8 = lane & 7;             // the gather's: group g8 holds a pair's row, column chunk c8
    float* sUV = sT + nRp + wave * (16 * kA32Ld + a32_list_words(K));      // this wave's [16 pairs][t -> out0 (32) | v -> Z2 (32) | pad]
    unsigned* sLo = reinterpret_cast<unsigned*>(sUV + 16 * kA32Ld) + g8 * (K + kAggPad);
    float* sLw = reinterpret_cast<float*>(reinterpret_cast<unsigned*>(sUV + 16 * kA32Ld) + 8 * (K + kAggPad)) + g8 * (K + kAggPad);
    const bool att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)K;
    const bool fast = agg_logit_table(a.t1, a.nR, sT, tid, lane);
    __syncthreads();                                     // the only workgroup barrier

    const int tbytes = (int)a.table_bytes;
    const __amdgpu_buffer_rsrc_t aggS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.agg), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t aggG = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.agg)) + a.table_bytes, 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
    const unsigned c16 = (unsigned)c8 * 16u;
    if (c8 < kAggPad) {                                  // the padding behind a group's K slots: beyond the buffer, no weight
        sLo[K + c8] = kAggPadRow;
        sLw[K + c8] = 0.f;
    }

    const int64_t nbatch = (a.B + 15) >> 4;
    const int64_t nwaves = (int64_t)gridDim.x * kAggWaves;
    auto run = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        for (int64_t batch = (int64_t)blockIdx.x * kAggWaves + wave; batch < nbatch; batch += nwaves) {
            const int64_t p_base = batch << 4;
            const int64_t pr = min(p_base + c, a.B - 1);
            const bool cvalid = p_base + c < a.B;
            unsigned x0u = (unsigned)a.items[pr * a.pid_stride];
            x0u = x0u < a.max_id ? x0u : a.max_id;
            const int x0c = (int)x0u;
            unsigned woff = ((unsigned)(4 * g) * 16u + (unsigned)c) * (unsigned)(NT * 4);      // Wperm[4 g][c][0]
            unsigned boff = (unsigned)g * 16u;
            unsigned roff = (unsigned)pr * (unsigned)(D * 4) + (unsigned)g * 16u;     // floats [4 g, 4 g + 4) of row `pr` of a [B][32] array
            asm volatile("" : "+v"(woff), "+v"(boff));   // (loop-invariant loads are not to be hoisted out of the batch loop)
            // THREE products as one chain of 6 k tiles, acc_j[ntp] += sum over k of W_j[k][16 ntp + c] b_j[nt] (16 MFMAs each), A through
            // a ring of three register buffers (mvin_fused_agg.hip)
            auto chain3 = [&](const float* W0p, const float* W1p, const float* W2p, const f32x4 (&b0)[NT], const f32x4 (&b1)[NT], const f32x4 (&b2)[NT],
                              f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], f32x4 (&acc2)[NT], auto&& between) {
                asm volatile("" : "+s"(W0p), "+s"(W1p), "+s"(W2p));
                const __amdgpu_buffer_rsrc_t wr[3] = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W0p), 0, D * D * 4, 0x00020000),
                                                      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1p), 0, D * D * 4, 0x00020000),
                                                      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2p), 0, D * D * 4, 0x00020000)};
                f32x2 ring[3][4];
                auto load = [&](auto s_) {
                    constexpr int s = decltype(s_)::value, j = s / NT, nt = s % NT;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(wr[j], woff + (unsigned)(r * D * 4), nt * 16 * D * 4, 0);
                        ring[s % 3][r] = f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
                    }
                };
                load(std::integral_constant<int, 0>{});
                load(std::integral_constant<int, 1>{});
                static_for<3 * NT>([&](auto s_) {
                    constexpr int s = decltype(s_)::value, j = s / NT, nt = s % NT;
                    if constexpr (s + 2 < 3 * NT) load(std::integral_constant<int, s + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 (&b)[NT] = j == 0 ? b0 : j == 1 ? b1 : b2;
                    f32x4 (&acc)[NT] = j == 0 ? acc0 : j == 1 ? acc1 : acc2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int ntp = 0; ntp < NT; ++ntp) acc[ntp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s % 3][r][ntp], b[nt][r], acc[ntp], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (nt == NT - 1) between(std::integral_constant<int, j>{});
                });
            };
            auto bias4 = [&](const float* bp, f32x4 (&acc)[NT]) {
                const char* bias = reinterpret_cast<const char*>(bp);
#pragma unroll
                for (int ntp = 0; ntp < NT; ++ntp)
                    acc[ntp] = bias ? *reinterpret_cast<const f32x4*>(bias + 64 * ntp + (size_t)boff) : f32x4{0.f, 0.f, 0.f, 0.f};
            };
            f32x4 qm[NT], qb[NT];
            {
                const char* qbase = reinterpret_cast<const char*>(a.q);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) qb[nt] = *reinterpret_cast<const f32x4*>(qbase + 64 * nt + (size_t)roff);
                f32x4 at[NT], av[NT];
                bias4(a.bq, at);
                bias4(a.bv, av);
                bias4(a.bm, qm);
                chain3(a.Wq, a.Wv, a.Wqm, qb, qb, qb, at, av, qm, [&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    if constexpr (j < 2) {               // t, then v: into the pairs' LDS rows
                        const f32x4 (&acc)[NT] = j == 0 ? at : av;
#pragma unroll
                        for (int ntp = 0; ntp < NT; ++ntp) *reinterpret_cast<f32x4*>(sUV + c * kA32Ld + j * D + 16 * ntp + 4 * g) = acc[ntp];
                    }
                });
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const int noct = (int)((min((int64_t)16, a.B - p_base) + 7) >> 3);
            struct Oct {
                unsigned ce[SPL], cr[SPL];
                float4 s0;
            };
            auto oct_load = [&](int it) -> Oct {         // group g8's pair of step `it`: j = 8 it + g8
                Oct qd;
                const unsigned x0 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane & 48) + 8 * it + g8) << 2, x0c);
                agg_load_slots<SPL>(adjE, adjR, (x0 * (unsigned)K + (unsigned)(SPL * c8)) * 4u, qd.ce, qd.cr);
                qd.s0 = agg_row4(aggS, x0 * (unsigned)(D * 4) + c16);
                return qd;
            };
            Oct nx = oct_load(0);
            for (int it = 0; it < noct; ++it) {
                const int j = 8 * it + g8;               // this group's pair of the batch
                const bool pvalid = p_base + j < a.B;
                const Oct qd = nx;
                if (it + 1 < noct) nx = oct_load(it + 1);
                const unsigned (&ce)[SPL] = qd.ce;
                const unsigned (&cr)[SPL] = qd.cr;
                float* rowT = sUV + j * kA32Ld + 4 * c8;
                const float4 tt = *reinterpret_cast<const float4*>(rowT);
                const float4 vv = *reinterpret_cast<const float4*>(rowT + D);
                float wk[SPL];
                agg_row_weights<SPL, FAST, 3>(cr, att1, sT, invK, wk);
                int cc = (int)(cr[0] >> 24);
                cc = pvalid ? (cc < 1 ? 1 : (cc > K ? K : cc)) : 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the previous step's reads of the lists are done)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < SPL; ++i) {
                    sLo[SPL * c8 + i] = ((cr[i] >> 16) & 0xFFu) ? (ce[i] & 0xFFFFFFu) * (unsigned)(D * 4) : kAggPadRow;
                    sLw[SPL * c8 + i] = wk[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int kmax = __builtin_amdgcn_readfirstlane(a32_groups_imax(cc));
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
                f32x4 zb[NT], ob[NT], o2[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) zb[nt] = *reinterpret_cast<const f32x4*>(sUV + c * kA32Ld + D + 16 * nt + 4 * g);
                bias4(a.a1, o2);
                chain3(a.A1, a.Wm1, a.Wm2, zb, ob, o2, o2, qm, qm, [&](auto j_) {
                    if constexpr (decltype(j_)::value == 0) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            o2[nt] = f32x4{fmaxf(o2[nt][0], 0.f), fmaxf(o2[nt][1], 0.f), fmaxf(o2[nt][2], 0.f), fmaxf(o2[nt][3], 0.f)};
                            ob[nt] = *reinterpret_cast<const f32x4*>(sUV + c * kA32Ld + 16 * nt + 4 * g);      // out0: the second product's B operand
                        }
                    }
                });
                const bool uo_is_q = a.user_o == a.q;
                float part = 0.f;
#pragma unroll
                for (int ntp = 0; ntp < NT; ++ntp) {
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.M0) + (size_t)x0u * (D * 4) + 64 * ntp + (size_t)boff);
                    const f32x4 uo = uo_is_q ? qb[ntp] : *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.user_o) + 64 * ntp + (size_t)roff);
                    const f32x4 it4 = qm[ntp] + m0;
                    if (a.item_emb && cvalid) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.item_emb) + 64 * ntp + (size_t)roff) = it4;
                    part += it4[0] * uo[0] + it4[1] * uo[1] + it4[2] * uo[2] + it4[3] * uo[3];
                }
                part = xor32_sum(xor16_sum(part));       // the four lane groups' quarters of pair c's row
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

size_t fused_fold_d32_lds_bytes(int nR, int K) {
    return ((size_t)((nR + 3) & ~3) + (size_t)kAggWaves * (16 * kA32Ld + a32_list_words(K))) * sizeof(float);
}

template <int K>
static hipError_t launch_ea32_k(const EntityAggArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)((a.nR + 3) & ~3) + (size_t)kAggWaves * a32_list_words(K)) * sizeof(float);
    const int64_t noct = ((int64_t)a.n_entity + 7) >> 3;
    const int64_t want = (noct + kAggWaves - 1) / kAggWaves;
    const int64_t cap = 256 * 8;
    entity_aggregates_d32_kernel<K><<<(int)(want < cap ? want : cap), kAggWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_entity_aggregates_d32(const EntityAggArgs& a, hipStream_t st) {
    switch (a.K) {
        case 16: return launch_ea32_k<16>(a, st);
        case 32: return launch_ea32_k<32>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int K>
static hipError_t launch_fold32_k(const FoldArgs& a, hipStream_t st) {
    const size_t lds = fused_fold_d32_lds_bytes(a.nR, K);
    static thread_local int per_cu = 0;
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(score_l2_folded_d32_kernel<K>), kAggWaves * 64, lds) != hipSuccess || v < 1)
            v = 3;
        per_cu = v > 8 ? 8 : v;
    }
    const int64_t nbatch = (a.B + 15) >> 4;
    const int64_t want = (nbatch + kAggWaves - 1) / kAggWaves;
    const int64_t cap = 256 * (int64_t)per_cu;           // persistent grid
    score_l2_folded_d32_kernel<K><<<(int)(want < cap ? want : cap), kAggWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_score_l2_folded_d32(const FoldArgs& f, hipStream_t st) {
    switch (f.K) {
        case 16: return launch_fold32_k<16>(f, st);
        case 32: return launch_fold32_k<32>(f, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
