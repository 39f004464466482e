// Packed-tile variant of the fused two-level gather + attention kernel (gfx950), for an adjacency in the duplicate-slot
// encoding of mvin_prep.hip (mvin_encode_adjacency).
//
// Same arithmetic, arguments and outputs as gather_attn_l2_split_kernel (reference model.py:251-305,
// aggregators.py:98-146).  The reference's sampler repeats (neighbour, relation) slots whenever an entity has fewer than K
// edges (data_loader_user_set.py:383-384); on the KGs the reference ships most entities do (7.4 distinct slots of 32 on
// last-fm).  Equal slots have equal logits and equal rows, so
//     sum_k p_k E[y_k] = sum_{distinct s} (m_s exp(t[r_s] - max) / Z) E[y_s],     Z = sum_s m_s exp(t[r_s] - max)
// is exact (a re-association of the softmax-weighted sum, like project-after-sum), and a repeated CHILD contributes one
// row of the dense tile with weight m p instead of m rows.  The role-split kernel walks K children x K grandchildren per
// parent whatever the repeats; this one walks the distinct slots only:
//
//   * a TILE is 32 child rows taken from CONSECUTIVE parents (a parent's distinct children may straddle two tiles), so
//     the MFMA tiles stay full when a parent has 7 distinct children;
//   * the per-parent sums  nagg0 = (1/K) sum_n p0[n] self1[n],  nagg1 = (1/K) sum_n p1[n] out1[n]  become one more
//     MFMA per accumulator register: A = [segment x row] weights (0 outside the parent's rows), B = the accumulators
//     themselves (contraction index = row, in the order the accumulator layout holds them);
//   * roles: NG = 4 "front" waves + D / 16 "dense" waves per workgroup, ONE workgroup barrier per tile, two workgroups per CU.
//     A front wave owns 32 / NG rows of a tile end to end -- child ids, child adjacency chunks -> softmax over the distinct
//     slots -> (grandchild id, weight) lists in LDS that only it (and its helper, below) reads -> row gathers bounded by the
//     longest list of the round -- software-pipelined over tiles: the child words of tile s+2 and the adjacency chunks of
//     tile s+1 are in flight while tile s is gathered.  Which rows a wave owns is decided per tile: the 32 rows are ranked by
//     list length and dealt to waves / lane groups in a snake.  The dense waves run the MFMA phases of tile s-1 and then,
//     where it pays (D = 64, K <= 32), use the time they would wait for the front: they gather the SECOND round (the short
//     lists) of tile s and compute the parent softmax -> segment tables of tile s+1.
//   * what hipcc needs for this shape of kernel is in the comments at wave_lds_sync(), at the dense loop's lane laundering and
//     in HISTORY.md (round 4): spilled registers and conditional id loads each cost a vmcnt(0) behind freshly issued loads.
//
// Supported: D in {32, 64, 128}; K in {16, 32, 64, 128}; fp32 or bf16 table smaller than 4 GiB, at most 2^24 entities.
#include <cstdlib>
#include <type_traits>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kPackCH = 256;        // most parents per workgroup (ids, query rows and distinct-child counts staged in LDS)
#ifndef MVIN_PACK_MAXB
#define MVIN_PACK_MAXB 16
#endif
constexpr int kPackMaxB = MVIN_PACK_MAXB;

template <int N>
__device__ __forceinline__ int dpp_row_shr(int v) {      // lane l of a 16-lane row gets lane l-N (0 below the row start)
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned row_or16(unsigned v) {     // OR over a 16-lane row, every lane gets it
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    return v;
}

// LDS hand-off between LANES of one wave (no workgroup barrier): the hardware runs a wave's DS operations in order, but
// to the compiler every lane is a thread of its own -- without a fence it may forward a lane's own (conditional) store
// to its later load and hoist the other lanes' loads above the store (seen at D = 32, K = 32, where writer and reader
// index with the same expression).  Wavefront-scope fences emit no instructions.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int pack_minw(int D, int NG) { return ((NG + D / 16) * 64 <= 512) ? 4 : 3; }

template <int D, int KT, bool BF, int NG>
struct PackGeom {
    static constexpr int TM = 32;                       // child rows per tile
    static constexpr int RT = 2;                        // 16-row MFMA tiles per tile
    static constexpr int NM = D / 16;                   // dense waves (one 16-column tile each)
    static constexpr int NW = NG + NM;
    static constexpr int KS = D / 4;                    // MFMA k-steps per DxD matrix
    static constexpr int LDA = 2 * D + 2;               // conflict-free A-fragment reads
    static constexpr int LDZ = D + 2;
    static constexpr int YLD = KT + 1;                  // (id, weight) row stride
    static constexpr bool WIDE = BF && D == 128;        // 8 bf16 per lane (16-byte loads)
    static constexpr int EPL = WIDE ? 8 : 4;
    static constexpr int LPRX = D / EPL;                // lanes per table row
    static constexpr int RPWX = 64 / LPRX;              // rows per gather wave-round
    static constexpr int RPW = TM / NG;                 // rows per front wave per tile
    static constexpr int NRND = RPW / RPWX;             // gather rounds per wave
    static constexpr int LPN = KT / 4;                  // lanes per child adjacency row (int4 chunks)
    static constexpr int LPN_L2 = (LPN == 4) ? 2 : (LPN == 8) ? 3 : (LPN == 16) ? 4 : 5;
    static constexpr int RPP = 64 / LPN;                // child rows per id pass
    static constexpr int NP3 = (RPW + RPP - 1) / RPP;   // id passes per wave
    static constexpr int SEGW = 16 / NG;                // parent segments per front wave
    static constexpr int NP2 = SEGW / 4;                // ... in passes of four (16 lanes per parent row)
    static constexpr int SPL = KT / 16;                 // parent-row slots per lane
    static constexpr int MINW = pack_minw(D, NG);
    static_assert(RPW % RPWX == 0 && NRND >= 1, "rows per front wave must be whole gather rounds");
    static_assert(SEGW % 4 == 0, "segments per front wave");
    static_assert(KT == 16 || KT == 32 || KT == 64 || KT == 128, "K");
};

// LDS layout (words unless noted); the same function sizes the launch
struct PackLds {
    size_t sA, sZ, sYP, sW0, sW1, sSeg, sSegP, sSegF, sMeta, sPR, sSt, sCarry, sCnt, sT0, sT1, sBias, sPid, sPq, sPcnt, sU, total;
};
// the dense waves gather the second round of a tile (see HELP in the kernel): its lists are double-buffered
__host__ __device__ constexpr bool pack_help(int D, int K, bool bf, int NG) {
    const int epl = (bf && D == 128) ? 8 : 4, rpwx = 64 / (D / epl), nrnd = (32 / NG) / rpwx;
    return nrnd == 2 && !(bf && D == 128) && K <= 32;
}
constexpr int kPackUR = 16;         // projected-tables form: parents whose projected queries a dense wave keeps (one MFMA batch)
__host__ __device__ inline PackLds pack_lds(int D, int K, int nR, int NG, bool bf, bool prj = false) {
    PackLds l{};
    const size_t NM = D / 16, nRp = (nR + 1) & ~1;
    size_t o = 0;
    auto take = [&](size_t words) { const size_t at = o; o += (words + 3) & ~(size_t)3; return at; };   // 16-byte aligned pieces
    l.sA = take(2 * 32 * (size_t)(2 * D + 2));
    l.sZ = take(prj ? 0 : 32 * (size_t)(D + 2));        // (no Z tile in the projected-tables form)
    const size_t lrows = pack_help(D, K, bf, NG) ? 48 : 32;
    l.sYP = take(lrows * (size_t)(K + 1) * 2);          // ids [lrows][K+1], then weights likewise (48: the lists of the second
                                                        // gather round are double-buffered, see list_base)
    l.sW0 = take(3 * 32);                               // the segment tables are a ring of three tiles
    l.sW1 = take(3 * 32);
    l.sSeg = take(3 * 32);
    l.sSegP = take(3 * 16);
    l.sSegF = take(3 * 16);
    l.sMeta = take(3 * 2);
    l.sPR = take((size_t)NG * 2 * 96);                  // per front wave, two tiles: rows by rank | child entity | query row
    l.sSt = take((size_t)(NG + NM) * 2 * 16);           // per wave (front and dense), two tiles
    l.sCarry = take(NM * 2 * 16);
    l.sCnt = take(2);
    l.sT0 = take(nRp);
    l.sT1 = take(nRp);
    l.sBias = take(3 * (size_t)D);
    l.sPid = take(kPackCH);
    l.sPq = take(kPackCH);
    l.sPcnt = take(kPackCH / 4);                        // bytes
    l.sU = take(prj ? NM * kPackUR * 32 : 0);           // per dense wave: [kPackUR][2][16]  (q.W1 + b1 | q.W2 + b2), its 16 columns
    l.total = o * 4;
    return l;
}

// Development aid (FusedL2Args::dbg & 8, env MVIN_SPLIT_DBG=8): cycles per phase summed over every workgroup's first
// dense and first front wave (s_memtime), read back with mvin_debug_read_trace under MVIN_PACK_TRACE (scripts/trace_packed.py)
__device__ unsigned long long g_pack_prof[2 * 8];

// PRJ: the projected-tables form (mvin_gather_attn_l2_prj_fwd).  Everything the kernel applies to a gathered row before the
// ReLU is linear and the attention weights are scalars, so the matrices move from the rows to the TABLE:
//     self1 = (E[x1] + q) W1 + b1                                        = T1[x1] + u1
//     Z A0 + a0 = (self1 + (S' + c q) W2 + c b2) A0 + a0                 = TA1[x1] + sum_k w_k TA2[y_k] + v
// with T1 = E.W1, TA1 = E.W1.A0, TA2 = E.W2.A0 taken once per ENTITY (mvin_project_tables: a.table = [T1 ; TA1 ; TA2], rebuilt
// by every call) and u1 = q.W1 + b1, v = q.(W1 + c W2).A0 + (b1 + c b2).A0 + a0 once per PARENT (a.W1 / a.b1 and a.W2 / a.b2 =
// the combined matrix / bias, built by the same call).  The kernel gathers the same ids and the same number of grandchild rows
// (one more self row per distinct child) and has NO D x D product per distinct child left: a dense wave's work per tile is
// out1 = relu(row + v), the per-parent sums (16 MFMAs) and, every ~4 tiles, the projected queries of the next 16 parents.
// The front loads no query rows at all.
template <int D, int KT, bool BF, int NG, bool PROF = false, bool PRJ = false>
__global__ __launch_bounds__((NG + D / 16) * 64, pack_minw(D, NG)) void gather_attn_l2_packed_kernel(FusedL2Args a, int ppw) {
    static_assert(!(PRJ && BF), "projected tables are fp32");
    using G = PackGeom<D, KT, BF, NG>;
    constexpr int TM = G::TM, NM = G::NM, KS = G::KS, LDA = G::LDA, LDZ = G::LDZ, YLD = G::YLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PackLds L = pack_lds(D, KT, a.nR, NG, BF, PRJ);
    float* sA = smem + L.sA;                            // [2][TM][LDA]  {E[x1] + q | S' + (sum p / K) q}
    float* sZ = smem + L.sZ;                            // [TM][LDZ]
    int* sYI = reinterpret_cast<int*>(smem + L.sYP);    // [TM][YLD]  grandchild ids; rows private to a front wave
    float* sYW = smem + L.sYP + (pack_help(D, KT, BF, NG) ? 48 : 32) * YLD;     // their weights
    float* sW0 = smem + L.sW0;                          // [2][TM]  weight of the row in its parent's nagg0 (p0 m / K)
    float* sW1 = smem + L.sW1;                          // [2][TM]  ... nagg1
    int* sSeg = reinterpret_cast<int*>(smem + L.sSeg);  // [2][TM]  segment (parent of the tile) the row belongs to
    int* sSegP = reinterpret_cast<int*>(smem + L.sSegP);   // [2][16] global parent index of the segment or -1
    int* sSegF = reinterpret_cast<int*>(smem + L.sSegF);   // [2][16] bit 0: continues from the previous tile; bit 1: continues in the next
    int* sMeta = reinterpret_cast<int*>(smem + L.sMeta);   // [2][2]  segments (0 = no more tiles), rows
    int* sPR = reinterpret_cast<int*>(smem + L.sPR);    // [NG][2][3][TM] by rank: row | list length << 8, child entity, query row
    int* sSt = reinterpret_cast<int*>(smem + L.sSt);    // [NG][2][16] first row of every segment (front-wave scratch)
    float* sCarry = smem + L.sCarry;                    // [NM][2][16] partial sums of the parent that straddles a tile boundary
    int* sCnt = reinterpret_cast<int*>(smem + L.sCnt);
    float* sT0 = smem + L.sT0;
    float* sT1 = smem + L.sT1;
    float* sBias = smem + L.sBias;                      // [3][D]  a0 | b1 | b2
    int* sPid = reinterpret_cast<int*>(smem + L.sPid);  // [ppw] entity id of the workgroup's parents
    int* sPq = reinterpret_cast<int*>(smem + L.sPq);    // [ppw] their query row (pair)
    unsigned char* sPcnt = reinterpret_cast<unsigned char*>(smem + L.sPcnt);   // [ppw] distinct children

    const int tid = threadIdx.x;
    int lane = tid & 63;                                // (not const: the dense loop re-defines it, see there)
    const int wave = tid >> 6;
    const bool is_dense = wave < NM;
    const bool has_proj = PRJ ? false : a.W1 != nullptr;
    const bool has_att0 = a.t0 != nullptr, has_att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)KT;
    long long prof_last = 0;
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int slot) {
        if constexpr (PROF) {
            const long long t = __builtin_readcyclecounter();
            prof_acc[slot] += (unsigned long long)(t - prof_last);
            prof_last = t;
        }
    };
    auto prof_flush = [&](int role) {
        if constexpr (PROF) {
            if (lane == 0)
                for (int i = 0; i < 8; ++i) atomicAdd(&g_pack_prof[role * 8 + i], prof_acc[i]);
        }
    };
    const int64_t p_base = (int64_t)blockIdx.x * ppw;
    const int n_loc = (int)((a.P - p_base) < ppw ? (a.P - p_base) : ppw);     // parents of this workgroup (contiguous)

    const __amdgpu_buffer_rsrc_t adjE = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(a.adj_e), 0, (int)a.adj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t adjR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(a.adj_r), 0, (int)a.adj_bytes, 0x00020000);

    for (int i = tid; i < a.nR; i += G::NW * 64) {
        sT0[i] = has_att0 ? a.t0[i] : 0.f;
        sT1[i] = has_att1 ? a.t1[i] : 0.f;
    }
    if (tid == 0) sCnt[0] = 0;
    for (int i = tid; i < D; i += G::NW * 64) {
        sBias[i] = a.a0 ? a.a0[i] : 0.f;
        sBias[D + i] = ((has_proj || PRJ) && a.b1) ? a.b1[i] : 0.f;       // (PRJ: added to the projected queries, not in phase B)
        sBias[2 * D + i] = ((has_proj || PRJ) && a.b2) ? a.b2[i] : 0.f;
    }
    for (int i = tid; i < n_loc; i += G::NW * 64) {
        const int pid = fused_parent_id(a, p_base + i);
        const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(adjR, (unsigned)pid * (unsigned)(KT * 4), 0, 0);
        sPid[i] = pid;
        sPq[i] = (int)((unsigned)(p_base + i) / (unsigned)a.parents_per_pair);
        const unsigned cn = w >> 24;
        sPcnt[i] = (unsigned char)(cn < 1u ? 1u : (cn > (unsigned)KT ? (unsigned)KT : cn));    // (a plain adjacency has 0 here)
    }
    __syncthreads();

    // =====================================================================================
    // what both roles use: table rows, list slots, the row gather
    // =====================================================================================
    const float c2scale = has_att0 ? invK : 1.f;        // (sum_k p_k) / K
    int g = lane / G::LPRX, c = lane % G::LPRX;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.table), 0, (int)(PRJ ? 3 * a.table_bytes : a.table_bytes), 0x00020000);
    const unsigned q_bytes = (unsigned)((a.P / a.parents_per_pair) * D * 4);
    const __amdgpu_buffer_rsrc_t qsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.q), 0, has_proj ? (int)q_bytes : 0, 0x00020000);
    const unsigned ta1_off = PRJ ? (unsigned)a.table_bytes : 0u;     // PRJ: a child's second self row (TA1) ...
    const unsigned t2_off = PRJ ? 2u * (unsigned)a.table_bytes : 0u; // ... and the grandchild rows (TA2)
    unsigned c16 = (unsigned)c * 16u;
    // a table row as this lane's EPL elements
    auto rowload = [&](int id, float4& lo, float4& hi, unsigned toff = 0u) {
        if constexpr (G::WIDE) {
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((unsigned)id * (unsigned)(D * 2)) + c16, 0, 0);
            lo = bf16x4_to_f32(make_uint2(raw[0], raw[1]));
            hi = bf16x4_to_f32(make_uint2(raw[2], raw[3]));
        } else if constexpr (BF) {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((unsigned)id * (unsigned)(D * 2)) + (unsigned)c * 8u, 0, 0);
            lo = bf16x4_to_f32(make_uint2(raw[0], raw[1]));
        } else {
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((unsigned)id * (unsigned)(D * 4)) + c16 + toff, 0, 0);
            lo = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3]));
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
    // ---- which rows a front wave owns: the tile's 32 rows are ranked by the length of their lists (classes of 1 << SH
    // entries, the granularity of a load batch; longest first), rank p goes to slot p (p < NS) or, on the way back, to slot
    // 2 NS - 1 - p, ... (a snake over NS = 32 / NRND slots), slot i to wave i % NG, lane group i / NG: round h of a wave
    // gathers the h-th rows of its lane groups, of similar length, and the waves of a tile issue about the same number of
    // row loads.  HELP (two rounds per wave: D = 64, or D = 128 on a bf16 table): the SECOND round -- the short lists -- is
    // gathered by dense wave gw, in the time it would otherwise wait for the front (the front is the critical path: 19 k
    // cycles per tile against 8 k of MFMA phases); its lists are double-buffered by tile parity, because the front wave
    // rewrites them for the next tile while the dense wave may still walk them. ----
    constexpr int NRND = G::NRND, NS = TM / NRND;
    constexpr bool HELP = pack_help(D, KT, BF, NG);      // (NRND == 2, not the 8-per-lane bf16 form, K <= 32: measured C4 (K = 64)
                                                          //  1.10 -> 1.15 ms (round 4), 1.107 -> 1.107 (round 5: the dense waves idle 25 k
                                                          //  of a tile's 32 k cycles there and helping still moves nothing -- the row rate,
                                                          //  not the front's issue, bounds that shape), C5 (bf16, K = 128) 1.72 -> 2.53 ms)
    static_assert(!HELP || NRND == 2, "HELP needs two gather rounds");
    constexpr int SH = KT >= 32 ? (KT == 32 ? 2 : KT == 64 ? 3 : 4) : 1;
    auto rank_of = [&](int gwx, int h, int grp) -> int {    // rank of the h-th row of lane group grp of front wave gwx
        const int slot = grp * NG + gwx;
        return (h & 1) ? (h + 1) * NS - 1 - slot : h * NS + slot;
    };
    auto list_base = [&](int gwx, int h, int grp, int par) -> int {     // first entry of that row's (id, weight) list
        if constexpr (HELP) return (h == 0 ? gwx * G::RPWX + grp : (1 + par) * NG * G::RPWX + gwx * G::RPWX + grp) * YLD;
        else return (gwx * G::RPW + h * G::RPWX + grp) * YLD;
    };
    // gather rounds [h0, h1) of front wave gwx's rows of the tile with parity `par` -> sA[buf]; MAXB list rows in flight
    auto gather_rounds = [&](int gwx, int par, int buf, int h0, int h1, auto maxb_c) {
        constexpr int MAXB = decltype(maxb_c)::value;
        const int* src = sPR + (gwx * 2 + par) * 96;
        int rr[NRND], xx[NRND], qq[NRND], cn[NRND], cmx[NRND];
#pragma unroll
        for (int h = 0; h < NRND; ++h) {
            const int p = rank_of(gwx, h, g);
            const int v = src[p];
            rr[h] = v & 31;
            cn[h] = v >> 8;
            xx[h] = src[32 + p];
            qq[h] = src[64 + p];
        }
#pragma unroll
        for (int h = 0; h < NRND; ++h) {                 // longest list of the round (wave-uniform)
            cmx[h] = __builtin_amdgcn_readfirstlane((int)wave_max((float)cn[h]));
            if constexpr (PROF) {
                if (a.dbg & 2) cmx[h] = 0;               // floor without the row gathers
            }
        }
#pragma unroll
        for (int h = 0; h < NRND; ++h) {
            if (h < h0 || h >= h1) continue;
            const int cmax = cmx[h];
            const int lb = list_base(gwx, h, g, par);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc;
            float* arow = sA + ((size_t)buf * TM + rr[h]) * LDA;
            // NB list rows in flight (entries past the list have weight 0: children() pads to K).  FIRST: behind the
            // child row and the query row, which are finished (E[x1] + q stored, (sum p / K) q added to the sum) as soon
            // as they land
            auto batch = [&](auto nb_c, auto first_c, int k0) {
                constexpr int NB = decltype(nb_c)::value;
                constexpr bool FIRST = decltype(first_c)::value;
                float4 sv, sv1 = make_float4(0.f, 0.f, 0.f, 0.f);
                u32x4 qa = (u32x4){0u, 0u, 0u, 0u}, qb = qa;
                float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sa1 = sa;
                if constexpr (FIRST) {
                    rowload(xx[h], sv, sv1);
                    if constexpr (PRJ) rowload(xx[h], sa, sa1, ta1_off);
                    // this lane's elements of the pair's query (zero records without the projection: the loads return 0)
                    const unsigned qoff = ((unsigned)qq[h] * (unsigned)D + (unsigned)(G::EPL * c)) * 4u;
                    if constexpr (!PRJ) {
                        qa = __builtin_amdgcn_raw_buffer_load_b128(qsrc, qoff, 0, 0);
                        if constexpr (G::WIDE) qb = __builtin_amdgcn_raw_buffer_load_b128(qsrc, qoff + 16u, 0, 0);
                    }
                }
                float4 lo[NB], hi[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) rowload(sYI[lb + k0 + i], lo[i], hi[i], t2_off);
                if constexpr (FIRST) {
                    const float4 q0 = make_float4(__uint_as_float(qa[0]), __uint_as_float(qa[1]), __uint_as_float(qa[2]), __uint_as_float(qa[3]));
                    const float4 q1 = make_float4(__uint_as_float(qb[0]), __uint_as_float(qb[1]), __uint_as_float(qb[2]), __uint_as_float(qb[3]));
                    if constexpr (PRJ) {
                        put(arow, sv, sv1);                                                // T1[x1]; the parent's u1 / v join on the dense side
                        acc = sa;                                                          // TA1[x1] + sum_k w_k TA2[y_k]
                    } else {
                        put(arow, f4_fma(1.f, q0, sv), f4_fma(1.f, q1, sv1));              // E[x1] + q
                        acc = f4_fma(c2scale, q0, acc);                                    // S' + (sum p / K) q
                        if constexpr (G::WIDE) acc1 = f4_fma(c2scale, q1, acc1);
                    }
                }
                // the weights are read when the rows are consumed: not live while the loads are in flight
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const float w = sYW[lb + k0 + i];
                    acc = f4_fma(w, lo[i], acc);
                    if constexpr (G::WIDE) acc1 = f4_fma(w, hi[i], acc1);
                }
            };
            using std::integral_constant;
            using T_ = std::true_type;
            using F_ = std::false_type;
            constexpr int NBF = MAXB >= 16 ? 12 : 4;             // list rows in the first batch (2 more loads ride in it)
            int k0;
            if (NBF >= 12 && cmax > 8) {
                batch(integral_constant<int, NBF>{}, T_{}, 0);
                k0 = NBF;
            } else if (NBF >= 12 && cmax > 4) {
                batch(integral_constant<int, (NBF >= 12 ? 8 : 4)>{}, T_{}, 0);
                k0 = 8;
            } else {
                batch(integral_constant<int, 4>{}, T_{}, 0);
                k0 = 4;
            }
            for (; k0 + MAXB <= cmax; k0 += MAXB) batch(integral_constant<int, MAXB>{}, F_{}, k0);
            const int rem = cmax - k0;
            if (rem > 8) batch(integral_constant<int, MAXB>{}, F_{}, k0);
            else if (rem > 4) batch(integral_constant<int, 8>{}, F_{}, k0);
            else if (rem > 0) batch(integral_constant<int, 4>{}, F_{}, k0);
            put(arow + D, acc, acc1);
        }
    };

    // DPAR: the parent softmax (segment tables of the next tile) runs on the dense waves, in their slack, not on the front
    constexpr bool DPAR = HELP && NM >= NG;         // (PRJ with the parent softmax left on the front: 1.495 vs 1.492 ms -- the same)
    struct Tile {
        int i0, c0;          // first parent (local index), children of it placed in earlier tiles
        int nseg, rows;      // parents in the tile (0: no tile), child rows
        int st_last;         // first row of the last parent
        bool open;           // the last parent continues in the next tile
        unsigned Em;         // bit (end row - 1) of every parent
    };
    // pack: which parents fill the tile that starts at (i0, c0) (every wave that needs it computes the same thing);
    // segment starts -> sSt[slot]
    auto pack = [&](int i0, int c0, int slot) -> Tile {
        Tile t{i0, c0, 0, 0, 0, false, 0u};
        const int j16 = lane & 15;
        int cj = (i0 + j16 < n_loc) ? (int)sPcnt[i0 + j16] : 0;
        if (j16 == 0) cj -= c0;
        int e = cj;
        e += dpp_row_shr<1>(e);
        e += dpp_row_shr<2>(e);
        e += dpp_row_shr<4>(e);
        e += dpp_row_shr<8>(e);
        const int st = e - cj;                       // first row of segment j16
        const bool in = (i0 + j16 < n_loc) && st < TM;
        t.nseg = __popcll(__ballot(in) & 0xFFFFull);
        const int last = t.nseg > 0 ? t.nseg - 1 : 0;    // (no parents left: zero rows)
        const int e_last = __builtin_amdgcn_readlane(e, last);
        t.st_last = __builtin_amdgcn_readlane(st, last);
        t.rows = t.nseg == 0 ? 0 : (e_last < TM ? e_last : TM);
        t.open = t.nseg != 0 && e_last > TM;
        t.Em = __builtin_amdgcn_readfirstlane(row_or16(in ? (1u << ((e < TM ? e : TM) - 1)) : 0u));
        if (lane < 16) sSt[(wave * 2 + slot) * 16 + lane] = st;
        return t;
    };
    auto advance = [&](const Tile& t, int& i0, int& c0) {
        if (t.nseg == 0) {
            c0 = 0;
            i0 = t.i0;
        } else if (t.open) {
            c0 = (t.nseg == 1 ? t.c0 : 0) + (TM - t.st_last);
            i0 = t.i0 + t.nseg - 1;
        } else {
            c0 = 0;
            i0 = t.i0 + t.nseg;
        }
    };
    // row r -> segment, slot of the parent's encoded adjacency row
    auto row_seg = [&](const Tile& t, int r, int& seg, int& slot) {
        const unsigned below = t.Em & ((1u << r) - 1u);     // segment ends before r
        seg = __popc(below);
        const int st_r = below ? 32 - __clz((int)below) : 0;
        slot = r - st_r + (seg == 0 ? t.c0 : 0);
    };
    // issue: the parent rows (relation words) of the segments of quarter wv
    auto issue_prw = [&](const Tile& t, int (&prw)[G::NP2][G::SPL], int wv) {
    const int j16 = lane & 15;
#pragma unroll
    for (int ps = 0; ps < G::NP2; ++ps) {
        const int sg = wv * G::SEGW + 4 * ps + (lane >> 4);
        const int pi = t.i0 + sg < n_loc ? t.i0 + sg : n_loc - 1;
        const unsigned off = ((unsigned)sPid[pi] * KT + (unsigned)j16 * G::SPL) * 4u;
        if constexpr (G::SPL == 1) {
            prw[ps][0] = __builtin_amdgcn_raw_buffer_load_b32(adjR, off, 0, 0);
        } else if constexpr (G::SPL == 2) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(adjR, off, 0, 0);
            prw[ps][0] = (int)v[0];
            prw[ps][1] = (int)v[1];
        } else {
#pragma unroll
            for (int h = 0; h < G::SPL / 4; ++h) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(adjR, off + 16u * h, 0, 0);
                prw[ps][4 * h + 0] = (int)v[0];
                prw[ps][4 * h + 1] = (int)v[1];
                prw[ps][4 * h + 2] = (int)v[2];
                prw[ps][4 * h + 3] = (int)v[3];
            }
        }
    }
    };
    // parents: softmax over the distinct slots -> row weights and segment tables of the tile, ring slot mb
    auto parents = [&](const Tile& t, const int (&prw)[G::NP2][G::SPL], int slot, int mb, int wv) {
        const int j16 = lane & 15;
#pragma unroll
        for (int ps = 0; ps < G::NP2; ++ps) {
            const int sg = wv * G::SEGW + 4 * ps + (lane >> 4);
            const bool sv = sg < t.nseg;
            float s0[G::SPL], s1[G::SPL], mu[G::SPL];
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int i = 0; i < G::SPL; ++i) {
                const unsigned w = (unsigned)prw[ps][i];
                const int rel = (int)(w & 0xFFFFu) < a.nR ? (int)(w & 0xFFFFu) : 0;
                mu[i] = (float)((w >> 16) & 0xFFu);
                s0[i] = sT0[rel];
                s1[i] = sT1[rel];
                m0 = fmaxf(m0, mu[i] > 0.f ? s0[i] : -INFINITY);
                m1 = fmaxf(m1, mu[i] > 0.f ? s1[i] : -INFINITY);
            }
            m0 = group_max(m0, 4);
            m1 = group_max(m1, 4);
            float z0 = 0.f, z1 = 0.f;
#pragma unroll
            for (int i = 0; i < G::SPL; ++i) {
                s0[i] = has_att0 ? mu[i] * lean_exp(s0[i] - m0) : mu[i];
                s1[i] = has_att1 ? mu[i] * lean_exp(s1[i] - m1) : mu[i];
                z0 += s0[i];
                z1 += s1[i];
            }
            z0 = group_sum(z0, 4);
            z1 = group_sum(z1, 4);
            const float r0 = has_att0 ? invK / z0 : invK, r1 = has_att1 ? invK / z1 : invK;
            const int stg = sSt[(wave * 2 + slot) * 16 + (sg & 15)];
            const int c0s = sg == 0 ? t.c0 : 0;
#pragma unroll
            for (int i = 0; i < G::SPL; ++i) {
                const int sl = j16 * G::SPL + i;
                const int rowi = stg + sl - c0s;
                if (sv && mu[i] > 0.f && sl >= c0s && rowi < TM) {
                    sW0[mb * TM + rowi] = s0[i] * r0;
                    sW1[mb * TM + rowi] = s1[i] * r1;
                    sSeg[mb * TM + rowi] = sg;
                }
            }
            if (j16 == 0) {
                sSegP[mb * 16 + sg] = sv ? (int)(p_base + t.i0 + sg) : -1;      // P * D * 4 < 2^31: fits an int
                sSegF[mb * 16 + sg] = ((sg == 0 && t.c0 > 0) ? 1 : 0) | ((sg == t.nseg - 1 && t.open) ? 2 : 0);
            }
        }
        if (wv == 0) {
            if (lane < TM && lane >= t.rows) {       // padding rows of a last, partial tile
                sW0[mb * TM + lane] = 0.f;
                sW1[mb * TM + lane] = 0.f;
                sSeg[mb * TM + lane] = 0;
            }
            if (lane == 0) {
                sMeta[2 * mb] = t.nseg;
                sMeta[2 * mb + 1] = t.rows;
            }
        }
    };

    if (is_dense) {
        // =====================================================================================
        // dense waves: MFMA phases of tile s-1
        // =====================================================================================
        int q16 = lane >> 4, l16 = lane & 15;
        int col = 16 * wave + l16;
        float bW1[KS], bW2[KS], bA0[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk = 4 * s + q16;
            bW1[s] = has_proj ? a.W1[kk * D + col] : 0.f;
            bW2[s] = has_proj ? a.W2[kk * D + col] : 0.f;
            bA0[s] = PRJ ? 0.f : a.A0[kk * D + col];
        }
        float* carry = sCarry + wave * 32;
        // PRJ: the projected queries u1 = q.W1 + b1, u2 = q.W2 + b2 of the workgroup's parents, sixteen parents per batch, this
        // wave's 16 columns of each; lane group q16 contracts k = KS q16 .. KS q16 + KS - 1 (its A values are one contiguous run
        // of the query row).  A tile spans at most 16 parents: the batch starts at the first parent of the tile that ran past the
        // previous one (a straddling parent is projected twice: ~20 batches per 256 parents instead of 16, for half the LDS of a
        // two-batch ring -- which cost the K = 64 instances their second workgroup per CU: 1.09 -> 1.32 ms at C4)
        float* sUw = smem + L.sU + (PRJ ? wave * (kPackUR * 32) : 0);
        int ucount = -(1 << 20);                         // first parent (local index) of the batch at hand
        // (buffer loads: ONE offset register per operand and the step in the instruction's immediate -- with flat loads hipcc
        //  precomputed 2 KS 64-bit addresses, ran out of registers and issued load -> vmcnt(0) -> MFMA 2 KS times per batch)
        const __amdgpu_buffer_rsrc_t w1src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W1), 0, PRJ ? D * D * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t w2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2), 0, PRJ ? D * D * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t uqsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q), 0, PRJ ? (int)q_bytes : 0, 0x00020000);
        constexpr int UKC = KS < 16 ? KS : 16;       // MFMA steps per round of a batch (D = 128: two rounds)
        struct UQ {
            u32x4 v[UKC / 4];                            // a lane's run of the parent's query row (one round)
        };
        auto u_issue_q = [&](UQ& o, int k0) {
            if constexpr (PRJ) {
                const int pi = ucount + l16 < n_loc ? ucount + l16 : n_loc - 1;
                const unsigned qo = ((unsigned)sPq[pi] * (unsigned)D + (unsigned)(KS * q16)) * 4u;
#pragma unroll
                for (int k4 = 0; k4 < UKC / 4; ++k4) o.v[k4] = __builtin_amdgcn_raw_buffer_load_b128(uqsrc, qo + 4u * (unsigned)k0 + 16u * (unsigned)k4, 0, 0);
            }
        };
        // one batch: the next 16 parents (prefetching the query rows a tile ahead was measured: no gain, 1.52 vs 1.49 ms)
        auto u_batch = [&] {
            if constexpr (PRJ) {
                UQ o;
                u_issue_q(o, 0);
                const unsigned wo = ((unsigned)(KS * q16) * (unsigned)D + (unsigned)col) * 4u;
                f32x4 u1 = (f32x4){0.f, 0.f, 0.f, 0.f}, u2 = u1;
#pragma unroll
                for (int k0 = 0; k0 < KS; k0 += UKC) {
                    unsigned w1v[UKC], w2v[UKC];
                    if (k0 > 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        u_issue_q(o, k0);
                    }
#pragma unroll
                    for (int k = 0; k < UKC; ++k) {
                        w1v[k] = __builtin_amdgcn_raw_buffer_load_b32(w1src, wo, (k0 + k) * D * 4, 0);
                        w2v[k] = __builtin_amdgcn_raw_buffer_load_b32(w2src, wo, (k0 + k) * D * 4, 0);
                    }
#pragma unroll
                    for (int k = 0; k < UKC; ++k) {
                        const float av = __uint_as_float(o.v[k >> 2][k & 3]);
                        u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float(w1v[k]), u1, 0, 0, 0);
                        u2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float(w2v[k]), u2, 0, 0, 0);
                    }
                }
                const float b1v = sBias[D + col], b2v = sBias[2 * D + col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = sUw + (4 * q16 + r) * 32 + l16;
                    d[0] = u1[r] + b1v;
                    d[16] = u2[r] + b2v;
                }
                wave_lds_sync();
            }
        };
        int dense_iter = 0;
        constexpr int DMAXB = 8;                         // list rows in flight in the helping gather (the weights stay resident)
        // DPAR: dense wave w < NG walks the tiles itself (same packing as the front), one tile ahead of the front's gather, and
        // writes the segment tables of quarter w.  Parent rows are issued AFTER the MFMA phases and consumed behind the helping
        // gather: nothing of it is live across the phases (their accumulators + the resident weights fill the budget)
        int di0 = 0, dc0 = 0;
        const bool dpar = DPAR && wave < NG;
        auto dense_parents = [&](int slot, int mb, auto&& between) {
            Tile dt = pack(di0, dc0, slot);
            int dprw[G::NP2][G::SPL];
            issue_prw(dt, dprw, wave);
            advance(dt, di0, dc0);
            between();
            wave_lds_sync();
            parents(dt, dprw, slot, mb, wave);
        };
        if (dpar) dense_parents(0, 0, [] {});            // tile 0 -> ring slot 0
        __syncthreads();                                 // the front's prologue: lists and rank tables of tile 0
        {
            auto help = [&] {
                if constexpr (HELP) {
                    if (wave < NG) gather_rounds(wave, 0, 0, 1, 2, std::integral_constant<int, DMAXB>{});
                }
            };
            if (dpar) dense_parents(1, 1, help);         // tile 1 -> ring slot 1 (zero segments: the end mark)
            else help();
        }
        if constexpr (PROF) prof_last = __builtin_readcyclecounter();
        for (int64_t s = 1;; ++s) {
            __syncthreads();                            // tile s-1 is in sA[(s-1) & 1]
            // Everything derived from the lane id is RE-derived per tile from an opaque copy: left loop-invariant, hipcc keeps
            // the address constants of the MFMA phases, of the helping gather and of the parent softmax live together across
            // the whole loop and spills 20 of them (each reload a vmcnt(0) in front of phase B); recomputing costs ~20 VALU
            asm volatile("" : "+v"(lane));
            q16 = lane >> 4;
            l16 = lane & 15;
            col = 16 * wave + l16;
            g = lane / G::LPRX;
            c = lane % G::LPRX;
            c16 = (unsigned)c * 16u;
            tick(0);
            const int buf = (int)((s - 1) & 1);        // sA half
            const int mb = (int)((s - 1) % 3);         // segment tables: ring of three
            const int nseg = sMeta[2 * mb], rows = sMeta[2 * mb + 1];
            if (nseg == 0) break;
            int i0t = 0;                                // PRJ: local index of the tile's first parent
            if constexpr (PRJ) {
                i0t = __builtin_amdgcn_readfirstlane(sSegP[mb * 16]) - (int)p_base;
                if (i0t + nseg > ucount + kPackUR) {
                    wave_lds_sync();                    // (the previous tile's reads of the batch are done)
                    ucount = i0t;
                    u_batch();
                }
            }
            const float* tA = sA + buf * TM * LDA;
            const bool two = rows > 16;                 // second 16-row MFMA tile in use
            // weights of this lane's rows in the per-parent sums: A operand of the segment products, row 16 m + 4 q16 + r
            float wa0[G::RT][4], wa1[G::RT][4];
            int sgi[G::RT][4];                          // (PRJ: the rows' segments pick the parent's projected query)
#pragma unroll
            for (int m = 0; m < G::RT; ++m) {
                const int4 sg = *reinterpret_cast<const int4*>(sSeg + mb * TM + 16 * m + 4 * q16);
                sgi[m][0] = sg.x, sgi[m][1] = sg.y, sgi[m][2] = sg.z, sgi[m][3] = sg.w;
                const float4 w0 = *reinterpret_cast<const float4*>(sW0 + mb * TM + 16 * m + 4 * q16);
                const float4 w1 = *reinterpret_cast<const float4*>(sW1 + mb * TM + 16 * m + 4 * q16);
                wa0[m][0] = sg.x == l16 ? w0.x : 0.f;
                wa0[m][1] = sg.y == l16 ? w0.y : 0.f;
                wa0[m][2] = sg.z == l16 ? w0.z : 0.f;
                wa0[m][3] = sg.w == l16 ? w0.w : 0.f;
                wa1[m][0] = sg.x == l16 ? w1.x : 0.f;
                wa1[m][1] = sg.y == l16 ? w1.y : 0.f;
                wa1[m][2] = sg.z == l16 ? w1.z : 0.f;
                wa1[m][3] = sg.w == l16 ? w1.w : 0.f;
            }
            f32x4 accN0 = (f32x4){0.f, 0.f, 0.f, 0.f}, accN1 = accN0;
            if constexpr (PRJ) {
                // self1 = T1[x1] + u1 ; out1 = relu(TA1[x1] + sum_k w_k TA2[y_k] + v) ; the per-parent sums.  No phase C, no Z tile.
#pragma unroll
                for (int m = 0; m < G::RT; ++m) {
                    if (m == 0 || two) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * m + 4 * q16 + r;
                            const float* up = sUw + ((i0t - ucount + sgi[m][r]) & (kPackUR - 1)) * 32 + l16;
                            const float s1v = tA[row * LDA + col] + up[0];
                            const float o = fmaxf(tA[row * LDA + D + col] + up[16], 0.f);
                            accN0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[m][r], s1v, accN0, 0, 0, 0);
                            accN1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[m][r], o, accN1, 0, 0, 0);
                        }
                    }
                }
                tick(1);
                tick(2);
            } else {
            const float c1v = sBias[D + col];
            const float c2v = sBias[2 * D + col] * c2scale;
            // phase B: self1 = (E[x1] + q) W1 + b1 ; Z = self1 + (S' + (sum p / K) q) W2 + (sum p / K) b2   (model.py:277-283)
            f32x4 accE[G::RT], accS[G::RT];
#pragma unroll
            for (int m = 0; m < G::RT; ++m) {
                accE[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                accS[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            bool do_mfma = true;
            if constexpr (PROF) do_mfma = !(a.dbg & 1);  // floor without the MFMAs
            if (has_proj && do_mfma) {
                if (two) {
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
#pragma unroll
                        for (int m = 0; m < G::RT; ++m) {
                            const float* ar = tA + (16 * m + l16) * LDA + 4 * k + q16;
                            accE[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], bW1[k], accE[m], 0, 0, 0);
                            accS[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[D], bW2[k], accS[m], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const float* ar = tA + l16 * LDA + 4 * k + q16;
                        accE[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], bW1[k], accE[0], 0, 0, 0);
                        accS[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[D], bW2[k], accS[0], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < G::RT; ++m) {
                if (m == 0 || two) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * m + 4 * q16 + r;
                        float s1v, zv;
                        if (has_proj) {
                            s1v = accE[m][r] + c1v;
                            zv = s1v + (accS[m][r] + c2v);
                        } else {
                            s1v = tA[row * LDA + col];
                            zv = tA[row * LDA + D + col];
                            zv += s1v;
                        }
                        // nagg0[segment] += w0[row] self1[row]: contraction over this accumulator register's four rows
                        accN0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[m][r], s1v, accN0, 0, 0, 0);
                        sZ[row * LDZ + col] = zv;
                    }
                }
            }
            tick(1);
            // every dense wave's columns of Z must be in LDS before any of them starts phase C
            ++dense_iter;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_fetch_add(sCnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(sCnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NM * dense_iter)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            tick(2);
            // phase C: out1 = relu(Z A0 + a0) (aggregators.py:108-116) ; nagg1[segment] += w1[row] out1[row]
            f32x4 acc2[G::RT];
#pragma unroll
            for (int m = 0; m < G::RT; ++m) acc2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!do_mfma) {
            } else if (two) {
#pragma unroll
                for (int k = 0; k < KS; ++k) {
#pragma unroll
                    for (int m = 0; m < G::RT; ++m) {
                        const float az = sZ[(16 * m + l16) * LDZ + 4 * k + q16];
                        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bA0[k], acc2[m], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const float az = sZ[l16 * LDZ + 4 * k + q16];
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bA0[k], acc2[0], 0, 0, 0);
                }
            }
            const float a0v = sBias[col];
#pragma unroll
            for (int m = 0; m < G::RT; ++m) {
                if (m == 0 || two) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float o = fmaxf(acc2[m][r] + a0v, 0.f);
                        accN1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[m][r], o, accN1, 0, 0, 0);
                    }
                }
            }
            }
            // accN0 / accN1 register i = segment 4 q16 + i, column col
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sg = 4 * q16 + i;
                const int pidx = sSegP[mb * 16 + sg];
                const int fl = sSegF[mb * 16 + sg];
                if (pidx >= 0) {
                    float v0 = accN0[i], v1 = accN1[i];
                    if (fl & 1) {
                        v0 += carry[l16];
                        v1 += carry[16 + l16];
                    }
                    if (fl & 2) {
                        carry[l16] = v0;
                        carry[16 + l16] = v1;
                    } else {
                        a.nagg0[(int64_t)pidx * D + col] = v0;
                        a.nagg1[(int64_t)pidx * D + col] = v1;
                    }
                }
            }
            tick(3);
            {
                auto help = [&] {                        // the second gather round of tile s (if there is one)
                    if constexpr (HELP) {
                        if (wave < NG && sMeta[2 * (int)(s % 3)] != 0)
                            gather_rounds(wave, (int)(s & 1), (int)(s & 1), 1, 2, std::integral_constant<int, DMAXB>{});
                    }
                };
                if (dpar) dense_parents((int)((s + 1) & 1), (int)((s + 1) % 3), help);     // + the segment tables of tile s+1
                else help();
            }
            tick(4);
            if constexpr (PROF) prof_acc[7] += 1;
        }
        if (wave == 0) prof_flush(0);
    } else {
        // =====================================================================================
        // front waves: tile s -> sA[s & 1] and its segment tables
        // =====================================================================================
        const int gw = wave - NM;
        struct Ids {
            int prw[G::NP2][G::SPL];     // relation words of this wave's parent rows
            int xw;                      // encoded child id (id | list length << 24) of tile row lane & 31
            int rq;                      // query row of its parent
            bool valid;
        };
        // issue: parent rows (relation words) of this wave's segments; the child word of EVERY tile row (each wave ranks all rows)
        auto issue_ids = [&](const Tile& t, Ids& d) {
            if constexpr (!DPAR) issue_prw(t, d.prw, gw);
            const int r = lane & 31;
            int seg, slot;
            row_seg(t, r, seg, slot);
            d.valid = r < t.rows;
            int pi = d.valid ? t.i0 + seg : t.i0;
            pi = pi < n_loc ? pi : n_loc - 1;
            d.rq = sPq[pi];
            d.xw = __builtin_amdgcn_raw_buffer_load_b32(adjE, ((unsigned)sPid[pi] * KT + (unsigned)(d.valid ? slot : 0)) * 4u, 0, 0);
        };
        // rank the rows -> sPR / sPX / sPQ [par][rank] = (row | list length << 8, child entity, query row), private to the wave
        auto rank_rows = [&](const Ids& d, int par) {
            const unsigned w = (unsigned)d.xw;
            const int cnt = d.valid ? (int)(w >> 24) : 0;
            const unsigned xid = (w & 0xFFFFFFu) < a.max_id ? (w & 0xFFFFFFu) : a.max_id;
            const int cls = (cnt + (1 << SH) - 1) >> SH;     // 0 .. 8
            int rank = 0, base = 0;
#pragma unroll
            for (int cc = 8; cc >= 0; --cc) {
                const unsigned m = (unsigned)(__ballot(cls == cc) & 0xFFFFFFFFull);
                if (cls == cc) rank = base + (int)__builtin_amdgcn_mbcnt_lo(m, 0u);
                base += __popc(m);
            }
            if (lane < 32) {
                int* dst = sPR + (gw * 2 + par) * 96;
                dst[rank] = lane | (cnt << 8);
                dst[32 + rank] = (int)xid;
                dst[64 + rank] = d.rq;
            }
        };
        // issue: the adjacency chunks of this wave's child rows (local row lr = h * RPWX + group)
        auto issue_chunks = [&](int par, int4 (&ye)[G::NP3], int4 (&re)[G::NP3]) {
            const int* src = sPR + (gw * 2 + par) * 96;
#pragma unroll
            for (int ps = 0; ps < G::NP3; ++ps) {
                int lr = ps * G::RPP + lane / G::LPN;
                lr = lr < G::RPW ? lr : G::RPW - 1;
                const int ch = lane % G::LPN;
                const unsigned xid = (unsigned)src[32 + rank_of(gw, lr / G::RPWX, lr % G::RPWX)];
                const unsigned off = (xid * KT + 4u * ch) * 4u;
                const u32x4 e4 = __builtin_amdgcn_raw_buffer_load_b128(adjE, off, 0, 0);
                ye[ps] = make_int4((int)e4[0], (int)e4[1], (int)e4[2], (int)e4[3]);
                const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(adjR, off, 0, 0);
                re[ps] = make_int4((int)r4[0], (int)r4[1], (int)r4[2], (int)r4[3]);
            }
        };
        // children: softmax over the distinct slots -> (grandchild id, weight) lists of this wave's rows
        auto children = [&](int par, const int4 (&ye)[G::NP3], const int4 (&re)[G::NP3]) {
            const int* src = sPR + (gw * 2 + par) * 96;
#pragma unroll
            for (int ps = 0; ps < G::NP3; ++ps) {
                const int lr0 = ps * G::RPP + lane / G::LPN;
                const bool lv = lr0 < G::RPW;
                const int lr = lv ? lr0 : G::RPW - 1;
                const int ch = lane % G::LPN;
                const unsigned w4[4] = {(unsigned)re[ps].x, (unsigned)re[ps].y, (unsigned)re[ps].z, (unsigned)re[ps].w};
                float sc[4], mu[4];
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rel = (int)(w4[i] & 0xFFFFu) < a.nR ? (int)(w4[i] & 0xFFFFu) : 0;
                    mu[i] = (float)((w4[i] >> 16) & 0xFFu);
                    sc[i] = sT0[rel];
                    m = fmaxf(m, mu[i] > 0.f ? sc[i] : -INFINITY);
                }
                m = group_max(m, G::LPN_L2);
                float z = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i] = has_att0 ? mu[i] * lean_exp(sc[i] - m) : mu[i];
                    z += sc[i];
                }
                z = group_sum(z, G::LPN_L2);
                if (lv) {
                    const float rr = has_att0 ? invK / z : invK;
                    // list slot = (wave, local row), not the tile row: a tile row changes owner from tile to tile, and another
                    // wave may still be walking the previous tile's list of that row
                    const int lb = list_base(gw, lr / G::RPWX, lr % G::RPWX, par) + 4 * ch;
                    int* di = sYI + lb;
                    float* dw = sYW + lb;
                    const unsigned idm = a.max_id;
                    di[0] = (int)(((unsigned)ye[ps].x & 0xFFFFFFu) < idm ? ((unsigned)ye[ps].x & 0xFFFFFFu) : idm);
                    di[1] = (int)(((unsigned)ye[ps].y & 0xFFFFFFu) < idm ? ((unsigned)ye[ps].y & 0xFFFFFFu) : idm);
                    di[2] = (int)(((unsigned)ye[ps].z & 0xFFFFFFu) < idm ? ((unsigned)ye[ps].z & 0xFFFFFFu) : idm);
                    di[3] = (int)(((unsigned)ye[ps].w & 0xFFFFFFu) < idm ? ((unsigned)ye[ps].w & 0xFFFFFFu) : idm);
                    dw[0] = sc[0] * rr;
                    dw[1] = sc[1] * rr;
                    dw[2] = sc[2] * rr;
                    dw[3] = sc[3] * rr;
                }
            }
        };
        // ---- software pipeline over the tiles: while tile s is gathered, the adjacency chunks of tile s+1 and the parent
        // rows / child words of tile s+2 are in flight, so no step waits for an id fetch it has just issued.  Every id load
        // and its use is unconditional (a tile past the end packs to zero rows and loads clamped addresses), and the step
        // starts from a provably empty load queue: with conditional issue / use pairs or loads pending over the loop edge
        // the compiler's waitcnt pass falls back to vmcnt(0) right behind the loads it has just issued. ----
        int i0 = 0, c0 = 0;
        Tile tb = pack(i0, c0, 0);                       // tile s+1 (slot (s+1) & 1 of sSt)
        Ids db;
        int4 yeb[G::NP3], reb[G::NP3];
        issue_ids(tb, db);
        advance(tb, i0, c0);
        rank_rows(db, 0);
        wave_lds_sync();
        issue_chunks(0, yeb, reb);
        if constexpr (!DPAR) parents(tb, db.prw, 0, 0, gw);
        children(0, yeb, reb);
        Tile tc = pack(i0, c0, 1);                       // tile s+2
        Ids dc;
        issue_ids(tc, dc);
        advance(tc, i0, c0);
        wave_lds_sync();
        __syncthreads();                                 // tile 0's lists and rank tables are visible to the helping dense waves
        if constexpr (PROF) prof_last = __builtin_readcyclecounter();
        // here: tile 0's lists are in LDS; (tc, dc) = tile 1 with its ids in flight
        bool have = true;                                // tile s exists
        for (int64_t s = 0; have; ++s) {
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the ids of tile s+1 (issued a step ago)
            // lane-derived constants re-derived per tile from an opaque copy, as in the dense loop (fewer values live across the loop)
            asm volatile("" : "+v"(lane));
            g = lane / G::LPRX;
            c = lane % G::LPRX;
            c16 = (unsigned)c * 16u;
            const int par = (int)(s & 1), par1 = par ^ 1;
            tb = tc;
            db = dc;
            rank_rows(db, par1);
            wave_lds_sync();
            issue_chunks(par1, yeb, reb);
            tc = pack(i0, c0, par);                      // tile s+2 -> the sSt slot tile s used
            issue_ids(tc, dc);
            advance(tc, i0, c0);
            tick(0);
            gather_rounds(gw, par, par, 0, HELP ? 1 : NRND, std::integral_constant<int, (G::WIDE || KT >= 128) ? 8 : kPackMaxB>{});     // (K = 128: the id pipeline holds 40 registers)
            tick(1);
            wave_lds_sync();                             // the lists of tile s are consumed; sSt of tile s+2 is written
            if constexpr (!DPAR) parents(tb, db.prw, par1, (int)((s + 1) % 3), gw);   // (no tile s+1: zero segments = the end mark)
            tick(2);
            children(par1, yeb, reb);
            wave_lds_sync();
            tick(3);
            have = tb.nseg != 0;
            __syncthreads();
            tick(4);
            if constexpr (PROF) prof_acc[7] += 1;
        }
        __syncthreads();                                 // the dense waves' last step (they read the end mark behind it)
        if (gw == 0) prof_flush(1);
    }
}

hipError_t pack_read_prof(long long* host_dst, size_t n) {
    const size_t have = sizeof(g_pack_prof) / sizeof(long long);
    hipError_t e = hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pack_prof), (n < have ? n : have) * sizeof(long long));
    if (e != hipSuccess) return e;
    static const unsigned long long zeros[2 * 8] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pack_prof), zeros, sizeof(zeros));
}

template <int D, int KT, bool BF, int NG, bool PROF = false, bool PRJ = false>
static hipError_t launch_packed(const FusedL2Args& a, hipStream_t st) {
    using G = PackGeom<D, KT, BF, NG>;
    const size_t lds = pack_lds(D, KT, a.nR, NG, BF, PRJ).total;
    auto kern = gather_attn_l2_packed_kernel<D, KT, BF, NG, PROF, PRJ>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // parents per workgroup: enough workgroups to balance the tail (tiles per parent vary with the distinct counts),
    // few enough that the weight fragments and the id prologue are amortised
    static const char* env = getenv("MVIN_PACK_PPW");
    int64_t ppw = env ? atoi(env) : 0;
    if (ppw <= 0) {          // measured (MVIN_PACK_PPW sweeps at 512 / 4 096 / 65 536 / 524 288 parents): ~512 workgroups, 2 .. 256 parents each
        ppw = (a.P + 511) / 512;
        if (ppw < 2) ppw = 2;
        if (ppw > 256) ppw = 256;
    }
    if (ppw > kPackCH) ppw = kPackCH;
    const int64_t grid = (a.P + ppw - 1) / ppw;
    kern<<<(unsigned)grid, G::NW * 64, lds, st>>>(a, (int)ppw);
    return hipGetLastError();
}

// ---- projected-tables form: the per-call parameter block behind the three tables (see the PRJ comment at the kernel) ----
//   Wstack [3][D][D] = W1 | W1.A0 | W2.A0   (the B operands of the table build: mvin_linear_fwd, nz = 3)
//   Wv [D][D] = (W1 + c W2).A0 ;  b1c [D] = b1 ;  bv [D] = (b1 + c b2).A0 + a0
__global__ void prj_prepare_kernel(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ b1,
                                   const float* __restrict__ b2, const float* __restrict__ A0, const float* __restrict__ a0, float c, int D,
                                   float* __restrict__ blk) {
    const int i = blockIdx.x, j = threadIdx.x;
    float* Wstack = blk;
    float* Wv = blk + (size_t)3 * D * D;
    float* b1c = Wv + (size_t)D * D;
    float* bv = b1c + D;
    if (j >= D) return;
    if (i < D) {
        // eight steps' operands loaded before their FMAs (a rolled loop was a chain of 2 D dependent load latencies: 23 us)
        float s1 = 0.f, s2 = 0.f;
        for (int k0 = 0; k0 < D; k0 += 8) {
            float av[8], w1[8], w2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                av[k] = A0[(size_t)(k0 + k) * D + j];
                w1[k] = W1[(size_t)i * D + k0 + k];
                w2[k] = W2[(size_t)i * D + k0 + k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1 = fmaf(w1[k], av[k], s1);
                s2 = fmaf(w2[k], av[k], s2);
            }
        }
        Wstack[(size_t)i * D + j] = W1[(size_t)i * D + j];
        Wstack[(size_t)D * D + (size_t)i * D + j] = s1;
        Wstack[(size_t)2 * D * D + (size_t)i * D + j] = s2;
        Wv[(size_t)i * D + j] = fmaf(c, s2, s1);
    } else {
        float s = a0 ? a0[j] : 0.f;
        if (b1) {
            for (int k0 = 0; k0 < D; k0 += 8) {
                float av[8], bc[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    av[k] = A0[(size_t)(k0 + k) * D + j];
                    bc[k] = fmaf(c, b2[k0 + k], b1[k0 + k]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) s = fmaf(bc[k], av[k], s);
            }
        }
        b1c[j] = b1 ? b1[j] : 0.f;
        bv[j] = s;
    }
}

hipError_t launch_prj_prepare(const float* W1, const float* W2, const float* b1, const float* b2, const float* A0, const float* a0,
                              float c, int D, float* blk, hipStream_t st) {
    prj_prepare_kernel<<<D + 1, D < 64 ? 64 : D, 0, st>>>(W1, W2, b1, b2, A0, a0, c, D, blk);
    return hipGetLastError();
}

bool fused_packed_supported(int D, int K) {
    return (D == 32 || D == 64 || D == 128) && (K == 16 || K == 32 || K == 64 || K == 128);
}

// ... and for these arguments: no attention outputs (they are per slot: the plain adjacency), tables addressable with
// 32-bit byte offsets, output rows indexable with an int
bool fused_packed_applies(const FusedL2Args& a, int D) {
    return fused_packed_supported(D, a.K) && !a.probs_parent && !a.probs_child && a.adj_r && a.adj_bytes > 0 &&
           a.adj_bytes < (1ull << 31) && (uint64_t)a.P * D * 4 < (1ull << 31) && a.table_bytes < (a.prj ? (1ull << 30) : (1ull << 32)) &&
           a.max_id < (1u << 24) &&
           (uint64_t)a.P / (uint64_t)a.parents_per_pair * D * 4 < (a.prj ? (1ull << 30) : (1ull << 31));
}

template <int D, bool BF>
static hipError_t launch_packed_k(const FusedL2Args& a, hipStream_t st) {
    if constexpr (!BF) {
        if (a.prj) {
            switch (a.K) {
                case 16: return launch_packed<D, 16, BF, 4, false, true>(a, st);
                case 32:
                    if constexpr (D == 64) {
                        if (a.dbg & 8) return launch_packed<D, 32, BF, 4, true, true>(a, st);      // MVIN_SPLIT_DBG=8: profiled build
                    }
                    return launch_packed<D, 32, BF, 4, false, true>(a, st);
                case 64: return launch_packed<D, 64, BF, 4, false, true>(a, st);
                case 128: return launch_packed<D, 128, BF, 4, false, true>(a, st);
                default: return hipErrorInvalidValue;
            }
        }
    } else if (a.prj) return hipErrorInvalidValue;
    switch (a.K) {
        case 16: return launch_packed<D, 16, BF, 4>(a, st);
        case 32:
            if constexpr (D == 64 && !BF) {
                if (a.dbg & 8) return launch_packed<D, 32, BF, 4, true>(a, st);      // MVIN_SPLIT_DBG=8: profiled build
            }
            return launch_packed<D, 32, BF, 4>(a, st);
        case 64:
            if constexpr (D == 64 && !BF) {
                if (a.dbg & 8) return launch_packed<D, 64, BF, 4, true>(a, st);      // MVIN_SPLIT_DBG=8: profiled build
            }
            return launch_packed<D, 64, BF, 4>(a, st);
        case 128: return launch_packed<D, 128, BF, 4>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gather_attn_l2_packed(const FusedL2Args& a, int D, int table_bf16, hipStream_t st) {
    switch (D) {
        case 32: return table_bf16 ? launch_packed_k<32, true>(a, st) : launch_packed_k<32, false>(a, st);
        case 64: return table_bf16 ? launch_packed_k<64, true>(a, st) : launch_packed_k<64, false>(a, st);
        case 128: return table_bf16 ? launch_packed_k<128, true>(a, st) : launch_packed_k<128, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
