// The folded tail WITHOUT per-entity sums: every pair gathers the rows of its own distinct children and grandchildren (the wave-per-parent
// kernel of mvin_fused_wpp.hip) and finishes in the same launch (the tail of mvin_fused_agg.hip's folded kernel) -- what
// mvin_score_l2_folded_gather_fwd runs; MVIN.agg = False / MVIN_L2_AGG=0 ("every pair gathers its own rows": SURVEY 7.3-c keeps the per-entity
// route a separate mode, this is the other one).  Per-ROW hoists only (SURVEY 7.3-c route 2: the matrices moved to the tables, same rows
// gathered): TA1 = E W1 A0, TA2 = E W2 A0, T0A = E W0 A0, M0 = E W0 Wm0 (mvin_fold_tables_ex with aggregates = 0).  Per pair x:
//     out1[c] = relu(TA1[x_c] + sum_k w_ck TA2[y_ck] + q Wv + bv)        nagg1 = sum_c (p1_c / K) out1[c]
//     out0    = relu(T0A[x] + sum_c (p0_c / K) TA1[x_c] + q Wq + bq)     ((ev0 + nagg0) A0 + a0 of mvin_l2_tail_fwd: nagg0 A0 is a sum of TA1 rows)
//     out2    = relu((out0 + nagg1) A1 + a1) ;  item = M0[x] + q W0 Wm0 + out0 Wm1 + out2 Wm2 + bm ;  score = <user_o, item>
// One T1 row per child less than the projected-tables kernel gathers, no nagg0 / nagg1 round trip, no tail launch; the six products ride on
// the gather kernel's idle matrix cores (it is bound by its texture-address pipes).
#include "mvin_fused_agg.h"

namespace mvin {

constexpr int kWppWaves = 4;
constexpr int kWppUvLd = 132;
constexpr int kWppRound = 4;
constexpr int wpp_list_words(int K) { return 4 * 2 * (K + kWppRound); }

__device__ __forceinline__ float wppf_bperm(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int wppf_bperm(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

template <int K>
__global__ __launch_bounds__(kWppWaves * 64, 4) void score_l2_wppfold_kernel(FoldArgs a) {
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
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.tables), 0, (int)(3u * tbytes), 0x00020000);      // TA1 | TA2 | T0A
    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);
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

    const int64_t nbatch = (a.B + 15) >> 4;
    const int64_t nwaves = (int64_t)gridDim.x * kWppWaves;
    auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    for (int64_t batch = (int64_t)blockIdx.x * kWppWaves + wave; batch < nbatch; batch += nwaves) {
        const int64_t p_base = batch << 4;
        // ---- t = q Wq + bq, v = q Wv + bv (-> the pairs' LDS rows), m = q W0Wm0 + bm (registers) of the batch's 16 pairs: one chain of
        //      three transposed products (mvin_fused_agg.hip) ----
        int64_t pr = min(p_base + c, a.B - 1);
        const bool cvalid = p_base + c < a.B;
        if (a.order) pr = a.order[pr];
        unsigned x0u = (unsigned)a.items[pr * a.pid_stride];
        x0u = x0u < a.max_id ? x0u : a.max_id;
        unsigned woff = ((unsigned)(4 * g) * 16u + (unsigned)c) * 16u;      // Wperm[4 g][c][0]
        unsigned boff = (unsigned)g * 16u;
        unsigned roff = (unsigned)pr * (unsigned)(D * 4) + (unsigned)g * 16u;     // floats [4 g, 4 g + 4) of row `pr` of a [B][64] array
        asm volatile("" : "+v"(woff), "+v"(boff));       // (loop-invariant loads are not to be hoisted out of the batch loop)
        auto chain3 = [&](const float* W0p, const float* W1p, const float* W2p, const f32x4 (&b0)[4], const f32x4 (&b1)[4], const f32x4 (&b2)[4],
                          f32x4 (&acc0)[4], f32x4 (&acc1)[4], f32x4 (&acc2)[4], auto&& between) {
            asm volatile("" : "+s"(W0p), "+s"(W1p), "+s"(W2p));
            const __amdgpu_buffer_rsrc_t wr[3] = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W0p), 0, D * D * 4, 0x00020000),
                                                  __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1p), 0, D * D * 4, 0x00020000),
                                                  __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2p), 0, D * D * 4, 0x00020000)};
            f32x4 ring[3][4];
            auto load = [&](auto s_) {
                constexpr int s = decltype(s_)::value, jm = s / 4, nt = s % 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr[jm], woff + (unsigned)(r * D * 4), nt * 16 * D * 4, 0);
                    ring[s % 3][r] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
                }
            };
            load(std::integral_constant<int, 0>{});
            load(std::integral_constant<int, 1>{});
            static_for<12>([&](auto s_) {
                constexpr int s = decltype(s_)::value, jm = s / 4, nt = s % 4;
                if constexpr (s + 2 < 12) load(std::integral_constant<int, s + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 (&b)[4] = jm == 0 ? b0 : jm == 1 ? b1 : b2;
                f32x4 (&acc)[4] = jm == 0 ? acc0 : jm == 1 ? acc1 : acc2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int ntp = 0; ntp < 4; ++ntp) acc[ntp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s % 3][r][ntp], b[nt][r], acc[ntp], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (nt == 3) between(std::integral_constant<int, jm>{});
            });
        };
        auto bias4 = [&](const float* bp, f32x4 (&acc)[4]) {
            const char* bias = reinterpret_cast<const char*>(bp);
#pragma unroll
            for (int ntp = 0; ntp < 4; ++ntp)
                acc[ntp] = bias ? *reinterpret_cast<const f32x4*>(bias + 64 * ntp + (size_t)boff) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        f32x4 qm[4];
        {
            f32x4 qb[4], at[4], av[4];
            const char* qbase = reinterpret_cast<const char*>(a.q);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) qb[nt] = *reinterpret_cast<const f32x4*>(qbase + 64 * nt + (size_t)roff);
            bias4(a.bq, at);
            bias4(a.bv, av);
            bias4(a.bm, qm);
            chain3(a.Wq, a.Wv, a.Wqm, qb, qb, qb, at, av, qm, [&](auto j_) {
                constexpr int jm = decltype(j_)::value;
                if constexpr (jm < 2) {
                    const f32x4 (&acc)[4] = jm == 0 ? at : av;
#pragma unroll
                    for (int ntp = 0; ntp < 4; ++ntp) *reinterpret_cast<f32x4*>(sUV + c * kWppUvLd + jm * D + 16 * ntp + 4 * g) = acc[ntp];
                }
            });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int npar = (int)min((int64_t)16, a.B - p_base);
        // The walk is a chain of DEPENDENT loads (parent id -> its row -> the children's rows -> the grandchildren's table rows), ~1 us
        // each when the line comes from beyond the L2, with three waves per SIMD to hide them: the counters of the first version showed
        // the waves waiting 70 % of their time and the vector ALU 30 % busy.  So every load is issued one stage early: the NEXT parent's
        // row while this parent is walked, the NEXT pass's four child rows (adjacency + T1 + TA1) before this pass's table rows, and the
        // table rows in half rounds of four, one half in flight while the other is summed.
        auto parent_slot = [&](int j) -> unsigned {      // byte offset of slot `lane` of parent j's row
            const int64_t p = a.order ? (int64_t)__builtin_amdgcn_readfirstlane(a.order[p_base + j]) : p_base + j;
            unsigned x0 = (unsigned)a.items[p * a.pid_stride];
            x0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(x0 < a.max_id ? x0 : a.max_id));
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
            unsigned xp = (unsigned)a.items[p * a.pid_stride];
            xp = (unsigned)__builtin_amdgcn_readfirstlane((int)(xp < a.max_id ? xp : a.max_id));
            const float4 t0a = row4(2u * tbytes + xp * (unsigned)(D * 4) + c16);      // T0A[x]: needed behind the passes
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
                float4 ta1row;
            };
            auto child_load = [&](int pass) -> Child {
                Child ch;
                const int ci = 4 * pass + g;
                const bool valid = ci < cnt0;
                // (the four exchanges run with EVERY lane active and are masked by arithmetic afterwards: ds_bpermute returns 0 for a
                //  source lane that EXEC has switched off, and hipcc moves an exchange whose result is selected by `valid` into the
                //  branch of the valid lanes -- a parent's 17th child, read by group 0 from lane 16 of (invalid) group 1, weighed 0)
                ch.xc = wppf_bperm(sid, ci & 63);
                ch.cc = wppf_bperm(scnt, ci & 63);
                ch.w0 = wppf_bperm(p0, ci & 63), ch.w1 = wppf_bperm(p1, ci & 63);
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
                ch.ta1row = row4(valid ? xo : kPadRow);               // TA1[x_c]  (an invalid group: beyond the buffer, zeros)
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
                    yo[i] = mu > 0.f ? (ch.ce[i] & 0xFFFFFFu) * (unsigned)(D * 4) + tbytes : kPadRow;
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
                const int kmax = __builtin_amdgcn_readfirstlane(agg_xor32_imax(agg_xor16_imax(ch.cc)));
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
                acc0 = f4_fma(ch.w0, ch.ta1row, acc0);      // nagg0 A0 = sum_c (p0_c / K) TA1[x_c]
            }
            acc0 = make_float4(xor32_sum(xor16_sum(acc0.x)), xor32_sum(xor16_sum(acc0.y)), xor32_sum(xor16_sum(acc0.z)), xor32_sum(xor16_sum(acc0.w)));
            acc1 = make_float4(xor32_sum(xor16_sum(acc1.x)), xor32_sum(xor16_sum(acc1.y)), xor32_sum(xor16_sum(acc1.z)), xor32_sum(xor16_sum(acc1.w)));
            {   // out0 = relu(T0A[x] + nagg0 A0 + t) ; Z2 = out0 + nagg1: into the pair's two LDS rows, in place of t and v
                float* rowT = sUV + j * kWppUvLd + 4 * c;
                const float4 tt = *reinterpret_cast<const float4*>(rowT);
                const float4 o0 = make_float4(fmaxf(t0a.x + acc0.x + tt.x, 0.f), fmaxf(t0a.y + acc0.y + tt.y, 0.f), fmaxf(t0a.z + acc0.z + tt.z, 0.f),
                                              fmaxf(t0a.w + acc0.w + tt.w, 0.f));
                if (g == 0) {
                    *reinterpret_cast<float4*>(rowT) = o0;
                    *reinterpret_cast<float4*>(rowT + D) = make_float4(acc1.x + o0.x, acc1.y + o0.y, acc1.z + o0.z, acc1.w + o0.w);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- out2 = relu(Z2 A1 + a1) ; item = M0[x] + m + out0 Wm1 + out2 Wm2 (+ bm, in m) ; score ----
        {
            f32x4 zb[4], ob[4], o2[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) zb[nt] = *reinterpret_cast<const f32x4*>(sUV + c * kWppUvLd + D + 16 * nt + 4 * g);
            bias4(a.a1, o2);
            chain3(a.A1, a.Wm1, a.Wm2, zb, ob, o2, o2, qm, qm, [&](auto j_) {
                if constexpr (decltype(j_)::value == 0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        o2[nt] = f32x4{fmaxf(o2[nt][0], 0.f), fmaxf(o2[nt][1], 0.f), fmaxf(o2[nt][2], 0.f), fmaxf(o2[nt][3], 0.f)};
                        ob[nt] = *reinterpret_cast<const f32x4*>(sUV + c * kWppUvLd + 16 * nt + 4 * g);
                    }
                }
            });
            float part = 0.f;
#pragma unroll
            for (int ntp = 0; ntp < 4; ++ntp) {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.M0) + (size_t)x0u * (D * 4) + 64 * ntp + (size_t)boff);
                const f32x4 uo = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.user_o) + 64 * ntp + (size_t)roff);
                const f32x4 it4 = qm[ntp] + m0;
                if (a.item_emb && cvalid) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.item_emb) + 64 * ntp + (size_t)roff) = it4;
                part += it4[0] * uo[0] + it4[1] * uo[1] + it4[2] * uo[2] + it4[3] * uo[3];
            }
            part = xor32_sum(xor16_sum(part));
            if (g == 0 && cvalid) {
                a.scores[pr] = part;
                if (a.sig) a.sig[pr] = 1.f / (1.f + expf(-part));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next batch's t | v block waits for this batch's reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    };
    if (fast) run(std::true_type{});
    else run(std::false_type{});
}


size_t fused_wppfold_lds_bytes(int nR, int K) {
    return ((size_t)2 * ((nR + 3) & ~3) + (size_t)kWppWaves * (16 * kWppUvLd + wpp_list_words(K))) * sizeof(float);
}

template <int K>
static hipError_t launch_wppfold_k(const FoldArgs& a, hipStream_t st) {
    const size_t lds = fused_wppfold_lds_bytes(a.nR, K);
    static thread_local int per_cu = 0;
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(score_l2_wppfold_kernel<K>), kWppWaves * 64, lds) != hipSuccess || v < 1)
            v = 3;
        per_cu = v > 8 ? 8 : v;
    }
    const int64_t nbatch = (a.B + 15) >> 4;
    const int64_t want = (nbatch + kWppWaves - 1) / kWppWaves;
    const int64_t cap = 256 * (int64_t)per_cu;           // persistent grid
    score_l2_wppfold_kernel<K><<<(int)(want < cap ? want : cap), kWppWaves * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_score_l2_folded_gather(const FoldArgs& f, hipStream_t st) {
    switch (f.K) {
        case 16: return launch_wppfold_k<16>(f, st);
        case 32: return launch_wppfold_k<32>(f, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
