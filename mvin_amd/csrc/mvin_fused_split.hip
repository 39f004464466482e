// Role-split variant of the fused two-level gather + attention kernel (gfx950) for D >= 32.
//
// Same arithmetic, arguments and outputs as gather_attn_l2_kernel (mvin_fused.hip; reference
// model.py:251-305, aggregators.py:98-146) -- what changes is WHO does what inside a workgroup.
// In the symmetric kernel every wave gathers rows, then multiplies (226 VGPRs: 48 resident weight
// registers + 32 load registers -> 2 waves per SIMD, and the row loads stop while the wave multiplies).
// Here a workgroup is NG "gather" waves + NM = D/16 "dense" waves and a software pipeline over the
// tiles t (min(K, 32) children) of its parents; at step s
//
//   gather waves : tile s   : for each child, its K grandchild rows as 16-byte lane loads (one lane
//                             group per child, 16 loads in flight per lane), S' = (1/K) sum_k p_k E[y_k]
//                             and the raw child row  -> LDS tile sA[s & 1]
//   dense waves  : tile s-1 : MFMA phases B and C of mvin_fused.hip on sA[(s-1) & 1] (weights resident
//                             as B fragments, one 16-column tile per wave), nagg0 / nagg1 in registers
//                  tile s+1 : its id work -- the children's adjacency rows (issued before the MFMAs,
//                             consumed after them), softmax over K -> (id, weight) list sYP[(s+1) & 1]
//                  parent of tile s+2: adjacency row -> child ids + attention weights p0 / p1 (ring of 4)
//
// so the row gathers never wait for a multiply or for an id fetch, and neither role carries the other's
// registers: the kernel fits 128 VGPRs = 4 waves per SIMD (two 8-wave workgroups per CU at D = 64).
// One workgroup barrier per step; the dense waves order phase B -> phase C among themselves through
// an LDS counter (the gather waves do not take part).
//
// Supported: D in {32, 64, 128}; K in {32, 64, 128}, and K = 16 at D = 32 (BASELINE config C2: one parent per
// 16-child tile, 2 + 2 waves); fp32 or bf16 table.  Everything else stays on gather_attn_l2_kernel.
#include <cstdlib>
#include <type_traits>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Development aid (FusedL2Args::dbg & 4, env MVIN_SPLIT_DBG): workgroup 0 stamps s_memtime at the phase
// boundaries of its first steps; read back with mvin_debug_read_trace (scripts/trace_split.py).
constexpr int kTraceSteps = 64, kTraceSlots = 8;
__device__ long long g_split_trace[2 * kTraceSteps * kTraceSlots];

// Sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), every lane gets the total: two VALU lane
// swaps (gfx950 v_permlane32_swap / v_permlane16_swap) instead of two ds_bpermute round trips through the LDS.
__device__ __forceinline__ float rows_sum(float v) {
    const unsigned x = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // {[lo,lo], [hi,hi]}
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned yy = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane16_swap(yy, yy, false, false);    // {[r0,r0,r2,r2], [r1,r1,r3,r3]}
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// waves per SIMD the register budget is cut for: two 8-wave workgroups per CU (128 VGPRs), or one of 12 (168)
constexpr int split_minw(int D, int NG) { return ((NG + D / 16) * 64 <= 512) ? 4 : 3; }

template <int D, int KT, bool BF, int NG>
struct SplitGeom {
    static constexpr int TM = KT < 32 ? KT : 32;        // children per tile (one parent per tile at K = 16)
    static constexpr int RT = TM / 16;                  // 16-row MFMA tiles per tile
    static constexpr int NT = D / 16;                   // 16-column MFMA tiles = dense waves
    static constexpr int NM = NT;
    static constexpr int NW = NG + NM;
    static constexpr int KS = D / 4;                    // MFMA k-steps per DxD matrix
    static constexpr int LDA = 2 * D + 2;               // conflict-free A-fragment reads
    static constexpr int LDZ = D + 2;
    static constexpr int YLD = KT + 1;                  // (id, weight) row stride: lane groups read different rows
    static constexpr bool WIDE = BF && D == 128;        // 8 bf16 per lane (16-byte loads)
    static constexpr int EPL = WIDE ? 8 : 4;
    static constexpr int LPRX = D / EPL;                // lanes per table row
    static constexpr int RPWX = 64 / LPRX;              // lane groups (children) per gather wave-round
    static constexpr int NPW = TM / NG;                 // children per gather wave per tile
    static constexpr int NTILE = KT / TM;
    static constexpr int NPL = (KT + 63) / 64;          // parent-row ids per lane (dense wave 0)
    static constexpr int NCH = TM * KT / 4;             // int4 adjacency chunks per tile
    static constexpr int CPL = (NCH + NM * 64 - 1) / (NM * 64);   // ... per dense lane
    static constexpr int LPN = KT / 4;                  // lanes per child adjacency row
    static constexpr int LPN_L2 = (LPN == 4) ? 2 : (LPN == 8) ? 3 : (LPN == 16) ? 4 : 5;
    static constexpr int MINW = split_minw(D, NG);
    static_assert(NPW % RPWX == 0, "children per gather wave must be a multiple of its lane groups");
    static_assert(KT == 16 || KT == 32 || KT == 64 || KT == 128, "K");
};

size_t fused_split_lds_bytes(int D, int K, int nR) {
    const size_t tm = K < 32 ? K : 32;
    const size_t words = 2 * tm * (size_t)(2 * D + 2) + tm * (size_t)(D + 2) + 12 * (size_t)K + 2 * (size_t)((nR + 1) & ~1) + 2
                         + 4 * (size_t)D + 4 * (size_t)D;
    return words * 4 + 2 * tm * (size_t)(K + 1) * sizeof(int2);
}

template <int D, int KT, bool BF, int NG, int UNR, bool TRACE>
__global__ __launch_bounds__((NG + D / 16) * 64, split_minw(D, NG)) void gather_attn_l2_split_kernel(FusedL2Args a) {
    using G = SplitGeom<D, KT, BF, NG>;
    constexpr int TM = G::TM, NM = G::NM, KS = G::KS, LDA = G::LDA, LDZ = G::LDZ, YLD = G::YLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nRp = (a.nR + 1) & ~1;
    float* sA = smem;                                   // [2][TM][LDA]  {E[x1] raw | S'}
    float* sZ = sA + 2 * TM * LDA;                      // [TM][LDZ]
    float* sP0 = sZ + TM * LDZ;                         // [4][KT]  ring over parents
    float* sP1 = sP0 + 4 * KT;                          // [4][KT]
    float* sT0 = sP1 + 4 * KT;                          // [nRp]
    float* sT1 = sT0 + nRp;                             // [nRp]
    int* sX1 = reinterpret_cast<int*>(sT1 + nRp);       // [4][KT]
    int* sCnt = sX1 + 4 * KT;                           // [2]
    float* sQ = reinterpret_cast<float*>(sCnt + 2);     // [4][D]  query vector of the parent's pair (ring)
    float* sBias = sQ + 4 * D;                          // [3][D]  a0 | b1 | b2 (LDS, not registers: a spilled
                                                        //         register's reload would wait for every id load)
    int2* sYP = reinterpret_cast<int2*>(sBias + 4 * D); // [2][TM][YLD]   (even word offset: 8-byte aligned)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool is_dense = wave < NM;
    const bool has_proj = a.W1 != nullptr;
    const bool has_att0 = a.t0 != nullptr, has_att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)KT;
    const int64_t nloc = (a.P - blockIdx.x + gridDim.x - 1) / gridDim.x;    // parents of this workgroup
    const int64_t S = nloc * G::NTILE;                                      // its tiles
    auto stamp = [&](int64_t s, int slot) {
        if constexpr (TRACE) {
            if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == NM) && s >= 8 && s < 8 + kTraceSteps)
                g_split_trace[((wave == 0 ? 0 : 1) * kTraceSteps + (s - 8)) * kTraceSlots + slot] = __builtin_readcyclecounter();
        }
    };
    const int dbg = TRACE ? a.dbg : 0;                  // the skip-work knobs exist in the traced build only

    for (int i = tid; i < a.nR; i += G::NW * 64) {
        sT0[i] = has_att0 ? a.t0[i] : 0.f;
        sT1[i] = has_att1 ? a.t1[i] : 0.f;
    }
    if (tid == 0) sCnt[0] = 0;
    for (int i = tid; i < D; i += G::NW * 64) {
        sBias[i] = a.a0 ? a.a0[i] : 0.f;
        sBias[D + i] = (has_proj && a.b1) ? a.b1[i] : 0.f;
        sBias[2 * D + i] = (has_proj && a.b2) ? a.b2[i] : 0.f;
    }
    __syncthreads();

    if (is_dense) {
        // =====================================================================================
        // dense waves: MFMA phases for tile s-1, id work for tile s+1 and the parent of tile s+2
        // =====================================================================================
        const int q16 = lane >> 4, l16 = lane & 15;
        const int nt = wave;
        const int col = 16 * nt + l16;
        const int mlane = wave * 64 + lane;
        const float c2scale = has_att0 ? invK : 1.f;    // (sum_k p_k)/K
        float bW1[KS], bW2[KS], bA0[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk = 4 * s + q16;
            bW1[s] = has_proj ? a.W1[kk * D + col] : 0.f;
            bW2[s] = has_proj ? a.W2[kk * D + col] : 0.f;
            bA0[s] = a.A0[kk * D + col];
        }

        auto parent_of = [&](int64_t i) -> int64_t { return blockIdx.x + i * gridDim.x; };
        // adjacency rows and the two output rows through buffer descriptors: 32-bit per-lane offsets instead of
        // loop-invariant 64-bit per-lane pointers (those were what the register allocator spilled, and a spill
        // reload's vmcnt(0) waits for every id load in flight).  The launcher guarantees both are < 4 GiB.
        const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(a.adj_r), 0, a.adj_r ? (int)a.adj_bytes : 0, 0x00020000);   // none: ids read as 0
        const __amdgpu_buffer_rsrc_t out0 = __builtin_amdgcn_make_buffer_rsrc(a.nagg0, 0, (int)(a.P * D * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t out1 = __builtin_amdgcn_make_buffer_rsrc(a.nagg1, 0, (int)(a.P * D * 4), 0x00020000);
        // parent adjacency row -> registers (dense wave 0: lane n handles children n, n+64, ...)
        // (no projection: zero records -> every load returns 0 without touching memory)
        const __amdgpu_buffer_rsrc_t qsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.q), 0, has_proj ? (int)((a.P / a.parents_per_pair) * D * 4) : 0, 0x00020000);
        auto parent_load = [&](int64_t pp, int64_t x0, int (&xs)[G::NPL], int (&rr)[G::NPL], float (&qr)[(D + 63) / 64]) {
            // no branch around any load or its use (see the note at the step loop): lane n holds child n % KT, so at
            // KT = 32 the upper half-wave duplicates the lower one
#pragma unroll
            for (int i = 0; i < G::NPL; ++i) {
                const unsigned off = ((unsigned)x0 * KT + ((lane + 64 * i) & (KT - 1))) * 4u;
                xs[i] = __builtin_amdgcn_raw_buffer_load_b32(adjE, off, 0, 0);
                rr[i] = __builtin_amdgcn_raw_buffer_load_b32(adjR, off, 0, 0);
            }
            const unsigned qoff = ((unsigned)pp / (unsigned)a.parents_per_pair) * (unsigned)D;
#pragma unroll
            for (int i = 0; i < (D + 63) / 64; ++i)
                qr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(qsrc, (qoff + lane + 64 * i) * 4u, 0, 0));
        };
        // ... -> child ids + attention weights of aggregator (0,.) / (1,.) over the K children
        auto parent_store = [&](const int (&xs)[G::NPL], const int (&rr)[G::NPL], const float (&qr)[(D + 63) / 64], int slot,
                                bool commit) {
            if (commit) {
#pragma unroll
                for (int i = 0; i < (D + 63) / 64; ++i)
                    if (lane + 64 * i < D) sQ[slot * D + lane + 64 * i] = qr[i];
            }
            float s0[G::NPL], s1[G::NPL];
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int i = 0; i < G::NPL; ++i) {          // sT0 / sT1 hold zeros without attention: uniform weights
                s0[i] = sT0[rr[i]];
                s1[i] = sT1[rr[i]];
                m0 = fmaxf(m0, s0[i]);
                m1 = fmaxf(m1, s1[i]);
            }
            constexpr int PL2 = (KT >= 64) ? 6 : (KT == 32) ? 5 : 4;     // lanes holding distinct children: 64, 32 or 16
            m0 = group_max(m0, PL2);
            m1 = group_max(m1, PL2);
            float z0 = 0.f, z1 = 0.f;
#pragma unroll
            for (int i = 0; i < G::NPL; ++i) {
                s0[i] = has_att0 ? expf(s0[i] - m0) : 1.f;
                s1[i] = has_att1 ? expf(s1[i] - m1) : 1.f;
                z0 += s0[i];
                z1 += s1[i];
            }
            z0 = group_sum(z0, PL2);
            z1 = group_sum(z1, PL2);
#pragma unroll
            for (int i = 0; i < G::NPL; ++i) {
                const int n = (lane + 64 * i) & (KT - 1);
                if (commit) {
                    const float p0 = has_att0 ? s0[i] / z0 : 1.f;
                    const float p1 = has_att1 ? s1[i] / z1 : 1.f;
                    sX1[slot * KT + n] = xs[i];
                    sP0[slot * KT + n] = p0;
                    sP1[slot * KT + n] = p1;
                }
            }
        };
        // int4 chunk `it` of the adjacency rows of the children of tile (slot, tile)
        auto chunk_load = [&](int slot, int tile, int it, int4& ye, int4& re) {
            // unconditional (lanes beyond the tile's chunk count repeat earlier chunks and drop the result)
            const int item = (it * (NM * 64) + mlane) & (G::NCH - 1);
            const int nl = item >> G::LPN_L2, ch = item & (G::LPN - 1);
            const unsigned off = ((unsigned)sX1[slot * KT + tile * TM + nl] * KT + 4 * ch) * 4u;
            const u32x4 e4 = __builtin_amdgcn_raw_buffer_load_b128(adjE, off, 0, 0);
            ye = make_int4((int)e4[0], (int)e4[1], (int)e4[2], (int)e4[3]);
            const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(adjR, off, 0, 0);
            re = make_int4((int)r4[0], (int)r4[1], (int)r4[2], (int)r4[3]);
        };
        // ... -> softmax over K inside the child's lane group -> (grandchild id, p_k / K) list
        auto chunk_finish = [&](int tile, int it, int buf, const int4& ye, const int4& re) {
            const int item = it * (NM * 64) + mlane;
            const bool valid = item < G::NCH;
            const int nl = item >> G::LPN_L2, ch = item & (G::LPN - 1);
            // sT0 holds zeros without attention; the duplicate lanes hold valid relation ids too
            const float sc0 = sT0[re.x], sc1 = sT0[re.y], sc2 = sT0[re.z], sc3 = sT0[re.w];
            const float m = group_max(fmaxf(fmaxf(sc0, sc1), fmaxf(sc2, sc3)), G::LPN_L2);
            float e0 = 1.f, e1 = 1.f, e2 = 1.f, e3 = 1.f;
            if (has_att0) {
                e0 = expf(sc0 - m);
                e1 = expf(sc1 - m);
                e2 = expf(sc2 - m);
                e3 = expf(sc3 - m);
            }
            const float z = group_sum((e0 + e1) + (e2 + e3), G::LPN_L2);
            if (valid) {
                const float r = has_att0 ? invK / z : invK;     // p_k / K = e_k * (1 / (K z)): one division per lane
                int2* dst = sYP + ((size_t)buf * TM + nl) * YLD + 4 * ch;
                dst[0] = make_int2(ye.x, __float_as_int(e0 * r));
                dst[1] = make_int2(ye.y, __float_as_int(e1 * r));
                dst[2] = make_int2(ye.z, __float_as_int(e2 * r));
                dst[3] = make_int2(ye.w, __float_as_int(e3 * r));
            }
        };

        // ---- pipeline fill: parents 0 and 1, id list of tile 0 ----
        if (wave == 0) {
            int xs[G::NPL], rr[G::NPL];
            float qr[(D + 63) / 64];
            parent_load(parent_of(0), fused_parent_id(a, parent_of(0)), xs, rr, qr);
            parent_store(xs, rr, qr, 0, true);
            if (nloc > 1) {
                parent_load(parent_of(1), fused_parent_id(a, parent_of(1)), xs, rr, qr);
                parent_store(xs, rr, qr, 1, true);
            }
        }
        __syncthreads();
        {
            int4 ye[G::CPL], re[G::CPL];
#pragma unroll
            for (int it = 0; it < G::CPL; ++it) chunk_load(0, 0, it, ye[it], re[it]);
#pragma unroll
            for (int it = 0; it < G::CPL; ++it) chunk_finish(0, it, 0, ye[it], re[it]);
        }
        __syncthreads();

        float nacc0 = 0.f, nacc1 = 0.f;
        // The id loads of a step are issued and consumed UNCONDITIONALLY, by every dense wave, on clamped indices
        // (results are dropped where a step has nothing to prepare; only wave 0 commits the parent data).  With
        // conditional issue / consume pairs the compiler's waitcnt pass must assume a load may still be pending at
        // the loop back-edge and puts a vmcnt(0) in front of the next step's first load -- which then waits for
        // whatever was issued just before it (measured: ~3000 cycles per step).
        // The entity id of the parent to prepare is a wave-uniform scalar load taken one step ahead.
        auto clampi = [&](int64_t i) -> int64_t { return i < nloc ? i : nloc - 1; };
        int x0n = fused_parent_id(a, parent_of(clampi(2 / G::NTILE)));
        int dense_iter = 0;
        for (int64_t s = 0; s <= S; ++s) {
            stamp(s, 0);
            // ---------------- issue this step's id loads (they land under the MFMAs) ----------------
            const int64_t i2r = (s + 2) / G::NTILE;
            const int64_t i2 = clampi(i2r);
            const bool commit_parent = wave == 0 && (s + 2) % G::NTILE == 0 && i2r >= 2 && i2r < nloc;
            const int64_t t1 = s + 1 < S ? s + 1 : S - 1;
            const int64_t i1 = t1 / G::NTILE;
            const int tile1 = (int)(t1 % G::NTILE);
            int4 ye[G::CPL], re[G::CPL];
            constexpr bool CHUNK_EARLY = G::CPL <= 2;   // few registers: issue before phase B (more cover)
            if constexpr (CHUNK_EARLY) {
#pragma unroll
                for (int it = 0; it < G::CPL; ++it) chunk_load((int)(i1 & 3), tile1, it, ye[it], re[it]);
            }
            int nxs[G::NPL], nrr[G::NPL];
            float nq[(D + 63) / 64];
            parent_load(parent_of(i2), x0n, nxs, nrr, nq);
            x0n = fused_parent_id(a, parent_of(clampi((s + 3) / G::NTILE)));
            // ---------------- dense phases of tile s-1 ----------------
            if (s >= 1) {
                const int64_t td = s - 1;
                const int64_t id_ = td / G::NTILE;
                const int tile = (int)(td % G::NTILE);
                const int slot = (int)(id_ & 3);
                const float* tA = sA + (td & 1) * TM * LDA;
                const float* tP0 = sP0 + slot * KT + tile * TM;
                if (tile == 0) {
                    nacc0 = 0.f;
                    nacc1 = 0.f;
                }
                // the query is already in the tile rows (the gather waves store E[x1] + q and S' + (sum_k p_k / K) q,
                // model.py:277: (entity_vectors + transfer_o) . W + b), so only the biases remain
                const float c1v = sBias[D + col];
                const float c2v = sBias[2 * D + col] * c2scale;
                // phase B: self1 = E[x1] W1 + c1 ; Z = self1 + S' W2 + c2 (model.py:277-283 applied after the sum)
                f32x4 accE[G::RT], accS[G::RT];
#pragma unroll
                for (int m = 0; m < G::RT; ++m) {
                    accE[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    accS[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (has_proj && !(dbg & 1)) {
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
#pragma unroll
                        for (int m = 0; m < G::RT; ++m) {
                            const float* ar = tA + (16 * m + l16) * LDA + 4 * k + q16;
                            accE[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], bW1[k], accE[m], 0, 0, 0);
                            accS[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[D], bW2[k], accS[m], 0, 0, 0);
                        }
                    }
                }
                float part = 0.f;
#pragma unroll
                for (int m = 0; m < G::RT; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * m + 4 * q16 + r;
                        float s1v, zv;
                        if (has_proj) {
                            s1v = accE[m][r] + c1v;
                            zv = s1v + (accS[m][r] + c2v);
                        } else {
                            s1v = tA[row * LDA + col];
                            zv = s1v + tA[row * LDA + D + col];
                        }
                        part = fmaf(tP0[row], s1v, part);
                        sZ[row * LDZ + col] = zv;
                    }
                }
                nacc0 += rows_sum(part);
                stamp(s, 1);
            }
            // many chunks per lane: issued between the phases -- their registers are live only while phase C's 8
            // accumulators are (phase B holds 16), and they land under its MFMAs
            if constexpr (!CHUNK_EARLY) {
#pragma unroll
                for (int it = 0; it < G::CPL; ++it) chunk_load((int)(i1 & 3), tile1, it, ye[it], re[it]);
            }
            if (s >= 1) {
                const int64_t td = s - 1;
                const int64_t id_ = td / G::NTILE;
                const int tile = (int)(td % G::NTILE);
                const int slot = (int)(id_ & 3);
                const float* tP1 = sP1 + slot * KT + tile * TM;
                // every dense wave's columns of Z must be in LDS before any of them starts phase C
                ++dense_iter;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_fetch_add(sCnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(sCnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NM * dense_iter)
                    __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                stamp(s, 2);
                // phase C: out1 = relu(Z A0 + a0) (aggregators.py:108-116) ; nagg1 += sum_n p1[n] out1[n]
                f32x4 acc2[G::RT];
#pragma unroll
                for (int m = 0; m < G::RT; ++m) acc2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!(dbg & 1))
#pragma unroll
                for (int k = 0; k < KS; ++k) {
#pragma unroll
                    for (int m = 0; m < G::RT; ++m) {
                        const float az = sZ[(16 * m + l16) * LDZ + 4 * k + q16];
                        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bA0[k], acc2[m], 0, 0, 0);
                    }
                }
                float part = 0.f;
                const float a0v = sBias[col];
#pragma unroll
                for (int m = 0; m < G::RT; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * m + 4 * q16 + r;
                        const float o = fmaxf(acc2[m][r] + a0v, 0.f);
                        part = fmaf(tP1[row], o, part);
                    }
                }
                nacc1 += rows_sum(part);
                if (tile == G::NTILE - 1 && q16 == 0) {
                    const int64_t p = parent_of(id_);
                    const unsigned off = ((unsigned)p * D + col) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nacc0 * invK), out0, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nacc1 * invK), out1, off, 0, 0);
                }
            }
            stamp(s, 3);
            // ---------------- finish the id work ----------------
#pragma unroll
            for (int it = 0; it < G::CPL; ++it) chunk_finish(tile1, it, (int)((s + 1) & 1), ye[it], re[it]);
            stamp(s, 4);
            parent_store(nxs, nrr, nq, (int)(i2r & 3), commit_parent);
            stamp(s, 5);
            __syncthreads();
            stamp(s, 6);
        }
    } else {
        // =====================================================================================
        // gather waves: tile s -> sA[s & 1]
        // =====================================================================================
        const int gw = wave - NM;
        const float c2scale = has_att0 ? invK : 1.f;    // (sum_k p_k) / K
        const int g = lane / G::LPRX, c = lane % G::LPRX;
        const bool buf32 = a.table_bytes < (1ull << 32);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(a.table), 0, buf32 ? (int)a.table_bytes : 0, 0x00020000);
        const unsigned c16 = (unsigned)c * 16u;
        // addressing mode resolved OUTSIDE the loops (a per-load wave-uniform branch keeps every load in its
        // own basic block): 0 = bf16 rows, 64-bit addresses; 1 = fp32 through the buffer descriptor; 2 = fp32, 64-bit
        // addresses; 3 = bf16 rows through the buffer descriptor (one 32-bit offset per load instead of a 64-bit
        // pointer and its shift / add chain: fewer address registers per load in flight)
        auto run = [&](auto mode_c) {
            constexpr int MODE = decltype(mode_c)::value;
            auto row4 = [&](int id) -> float4 {
                if constexpr (MODE == 0) {
                    return bf16x4_to_f32(reinterpret_cast<const uint2*>(
                        reinterpret_cast<const uint16_t*>(a.table) + (int64_t)id * D)[c]);
                } else if constexpr (MODE == 3) {
                    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((unsigned)id * (unsigned)(D * 2)) + (unsigned)c * 8u, 0, 0);
                    return bf16x4_to_f32(make_uint2(raw[0], raw[1]));
                } else if constexpr (MODE == 1) {
                    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((unsigned)id * (unsigned)(D * 4)) + c16, 0, 0);
                    return make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]),
                                       __uint_as_float(raw[3]));
                } else {
                    return reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.table) + (int64_t)id * D)[c];
                }
            };
            auto load8 = [&](int id, float4& lo, float4& hi) {
                if constexpr (MODE == 3) {
                    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((unsigned)id * (unsigned)(D * 2)) + c16, 0, 0);
                    lo = bf16x4_to_f32(make_uint2(raw[0], raw[1]));
                    hi = bf16x4_to_f32(make_uint2(raw[2], raw[3]));
                } else {
                    const uint4 raw = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.table) + (int64_t)id * D)[c];
                    lo = bf16x4_to_f32(make_uint2(raw.x, raw.y));
                    hi = bf16x4_to_f32(make_uint2(raw.z, raw.w));
                }
            };
            auto put = [&](float* dst, float4 lo, float4 hi) {
                float* q = dst + G::EPL * c;
                *reinterpret_cast<float2*>(q) = make_float2(lo.x, lo.y);
                *reinterpret_cast<float2*>(q + 2) = make_float2(lo.z, lo.w);
                if constexpr (G::WIDE) {
                    *reinterpret_cast<float2*>(q + 4) = make_float2(hi.x, hi.y);
                    *reinterpret_cast<float2*>(q + 6) = make_float2(hi.z, hi.w);
                }
            };
            for (int64_t s = 0; s <= S; ++s) {
                stamp(s, 0);
                if (s < S && !(dbg & 2)) {
                    const int slot = (int)((s / G::NTILE) & 3);
                    const int tile = (int)(s % G::NTILE);
                    const int buf = (int)(s & 1);
#pragma unroll
                    for (int j = 0; j < G::NPW / G::RPWX; ++j) {
                        const int nl = gw * G::NPW + j * G::RPWX + g;
                        const int2* yp = sYP + ((size_t)buf * TM + nl) * YLD;
                        float* arow = sA + ((size_t)buf * TM + nl) * LDA;
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc;
                        float4 sv, sv1 = acc;
                        if constexpr (G::WIDE) {
                            load8(sX1[slot * KT + tile * TM + nl], sv, sv1);
                            auto body = [&](int k) {
                                const int2 e = yp[k];
                                float4 lo, hi;
                                load8(e.x, lo, hi);
                                acc = f4_fma(__int_as_float(e.y), lo, acc);
                                acc1 = f4_fma(__int_as_float(e.y), hi, acc1);
                            };
                            if constexpr (UNR == 48 && MODE == 3) {
                                // two register batches of HB rows in rotation: HB .. 2*HB loads in flight per lane, no drain
                                // between batches.  Named arrays + a runtime loop: left to the unroller hipcc serialises
                                // load -> use beyond 8 rows, in one fully unrolled block it sinks every load to its use,
                                // and an indexed ring of batches goes to scratch (all measured).
                                constexpr int HB = 8;
                                static_assert(KT % (2 * HB) == 0, "K must be a multiple of two batches");
                                auto issue = [&](u32x4 (&r)[HB], int k0) {
#pragma unroll
                                    for (int i = 0; i < HB; ++i)
                                        r[i] = __builtin_amdgcn_raw_buffer_load_b128(
                                            rsrc, ((unsigned)yp[k0 + i].x * (unsigned)(D * 2)) + c16, 0, 0);
                                };
                                auto consume = [&](const u32x4 (&r)[HB], int k0) {
#pragma unroll
                                    for (int i = 0; i < HB; ++i) {
                                        const float w = __int_as_float(yp[k0 + i].y);
                                        acc = f4_fma(w, bf16x4_to_f32(make_uint2(r[i][0], r[i][1])), acc);
                                        acc1 = f4_fma(w, bf16x4_to_f32(make_uint2(r[i][2], r[i][3])), acc1);
                                    }
                                };
                                u32x4 ra[HB], rb[HB];
                                issue(ra, 0);
                                for (int k0 = 0; k0 < KT; k0 += 2 * HB) {
                                    issue(rb, k0 + HB);
                                    consume(ra, k0);
                                    if (k0 + 2 * HB < KT) issue(ra, k0 + 2 * HB);
                                    consume(rb, k0 + HB);
                                }
                            } else {
#pragma unroll 8
                                for (int k = 0; k < KT; ++k) body(k);
                            }
                        } else {
                            sv = row4(sX1[slot * KT + tile * TM + nl]);     // first: lands under the K row loads
#pragma unroll UNR
                            for (int k = 0; k < KT; ++k) {
                                const int2 e = yp[k];
                                acc = f4_fma(__int_as_float(e.y), row4(e.x), acc);
                            }
                        }
                        // this lane's elements of the pair's query (zeros without the projection), read after the row
                        // loop so that they are not live across it
                        const float* qrow = sQ + slot * D + G::EPL * c;
                        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 q0 = has_proj ? *reinterpret_cast<const float4*>(qrow) : zero4;
                        const float4 q1 = (has_proj && G::WIDE) ? *reinterpret_cast<const float4*>(qrow + 4) : zero4;
                        put(arow, f4_fma(1.f, q0, sv), f4_fma(1.f, q1, sv1));                      // E[x1] + q
                        put(arow + D, f4_fma(c2scale, q0, acc), f4_fma(c2scale, q1, acc1));      // S' + (sum p / K) q
                        stamp(s, 1 + j);
                    }
                }
                stamp(s, 5);
                __syncthreads();
                stamp(s, 6);
            }
        };
        __syncthreads();   // parents 0 / 1 in the ring
        __syncthreads();   // id list of tile 0
        if constexpr (BF) {
            if (buf32) run(std::integral_constant<int, 3>{});
            else run(std::integral_constant<int, 0>{});
        } else {
            if (buf32) run(std::integral_constant<int, 1>{});
            else run(std::integral_constant<int, 2>{});
        }
    }
}

template <int D, int KT, bool BF, int NG, int UNR = 16, bool TRACE = false>
static hipError_t launch_split(const FusedL2Args& a, hipStream_t st) {
    using G = SplitGeom<D, KT, BF, NG>;
    const size_t lds = fused_split_lds_bytes(D, KT, a.nR);
    auto kern = gather_attn_l2_split_kernel<D, KT, BF, NG, UNR, TRACE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int per_cu = G::MINW == 4 ? 16 / G::NW : 1;         // 4 waves per SIMD: 16 waves per CU
    while (per_cu > 1 && per_cu * lds > 160 * 1024) --per_cu;
    const int64_t cap = 256 * per_cu;                   // persistent: one pipeline per resident workgroup
    const int grid = (int)(a.P < cap ? a.P : cap);
    kern<<<grid, G::NW * 64, lds, st>>>(a);
    return hipGetLastError();
}

bool fused_split_supported(int D, int K) {
    if (D == 32) return K == 16 || K == 32 || K == 64 || K == 128;
    return (D == 64 || D == 128) && (K == 32 || K == 64 || K == 128);
}

// ... and for these arguments: no attention outputs requested (eval_case_study keeps the symmetric kernel),
// adjacency tables and output rows addressable with 32-bit byte offsets
bool fused_split_applies(const FusedL2Args& a, int D) {
    return fused_split_supported(D, a.K) && !a.probs_parent && !a.probs_child && a.adj_bytes > 0 &&
           a.adj_bytes < (1ull << 31) && (uint64_t)a.P * D * 4 < (1ull << 31);
}

// rows in flight per lane in the gather loop: 16 (the unroller's batches) everywhere, except at D = 128 on a bf16 table
// (168-VGPR budget, one workgroup per CU, so only 4 gather waves feed the texture path): two register batches of 8 in
// rotation for K >= 64 (UNR = 48; C5, K = 128: 10.9 -> 13.3 TB/s; K = 64: 9.2 -> 10.5).  The same rotation spills in the 128-VGPR kernels (C3 / C4: 5-10x slower)
// and three batches / batches of 16 spill at D = 128 too.  MVIN_SPLIT_UNR = 8 / 16 selects the older variants (A/B).
template <int D, bool BF, int NG>
static hipError_t launch_split_k(const FusedL2Args& a, hipStream_t st) {
    static const char* u = getenv("MVIN_SPLIT_UNR");
    const int unr = u ? atoi(u) : 0;
    constexpr bool ROT = D == 128 && BF;
    switch (a.K) {
        case 32:
            if (unr == 8) return launch_split<D, 32, BF, NG, 8>(a, st);
            if constexpr (D == 64 && !BF) {
                if (a.dbg) return launch_split<D, 32, BF, NG, 16, true>(a, st);     // MVIN_SPLIT_DBG: traced build
            }
            return launch_split<D, 32, BF, NG>(a, st);      // K = 32: the rotation is slower (2.9 -> 4.5 ms, D = 128 bf16)
        case 64:
            if constexpr (ROT) {
                if (unr != 16) return launch_split<D, 64, BF, NG, 48>(a, st);
            }
            return launch_split<D, 64, BF, NG>(a, st);
        case 128:
            if constexpr (ROT) {
                if (unr != 16) return launch_split<D, 128, BF, NG, 48>(a, st);
            }
            return launch_split<D, 128, BF, NG>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t split_read_trace(long long* host_dst, size_t n) {
    const size_t have = sizeof(g_split_trace) / sizeof(long long);
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_split_trace), (n < have ? n : have) * sizeof(long long));
}

hipError_t launch_gather_attn_l2_split(const FusedL2Args& a, int D, int table_bf16, hipStream_t st) {
    if (D == 32) {
        if (a.K == 16) return table_bf16 ? launch_split<32, 16, true, 2>(a, st) : launch_split<32, 16, false, 2>(a, st);
        return table_bf16 ? launch_split_k<32, true, 4>(a, st) : launch_split_k<32, false, 4>(a, st);
    }
    if (D == 64) return table_bf16 ? launch_split_k<64, true, 4>(a, st) : launch_split_k<64, false, 4>(a, st);
    if (D == 128) return table_bf16 ? launch_split_k<128, true, 4>(a, st) : launch_split_k<128, false, 4>(a, st);
    return hipErrorInvalidValue;
}

}  // namespace mvin
