// MVIN._key_addressing (model.py:161-240) for pairs grouped by user, dense form (mvin_keyaddr_dense.hip), over STATIC
// per-user records.  A user's ripple sets (data_loader_user_set.py: user_triplet_set, built once per data set) never change
// between batches, and neither does anything the dense kernel derives from the ids alone -- the relation buckets of the
// memories (which rows share an R_KGE[r] and hence an MFMA tile), the tile table, the clamped head / tail ids.  The dense
// kernel re-derives them for every user segment of every batch: zeroing, LDS atomics, a scan, three workgroup barriers,
// 3.4 k of its 31 k cycles per segment, with nothing else of the segment able to start.  Here they are one record per user,
// written once (user_records_kernel below, like mvin_encode_adjacency for the adjacency) and landed in LDS by the LDS-DMA
// path a segment ahead: one wave, len / 64 instructions, no registers, no barrier.
//
// D = 64, fp32 table, 64 <= P * NmP <= 128 rows per user.
//
// Per user segment s (one workgroup of 12 waves; waves 0..10 hold the R_KGE fragments, wave 11 does the h-set read; the four
// "side" waves 8..11 take no logits / softmax / reads tile and carry the loads of the tile phases instead):
//   top     : [barrier]  wave 11: record of s+1 -> LDS (LDS-DMA);  every wave: 12 of the segment's tail rows -> registers
//   U       : U_m = R_KGE[r_m] . h_m as 16-row tiles of memories that share a relation (bucket table of the record),
//             h-set read (wave 11); the head rows were landed during segment s-1; behind its own U work a wave writes its
//             tail rows to sT (their memory latency, ~6 k cycles for the burst, has passed under the MFMA tiles)
//   per tile of 16 pairs, THREE barriers:  logits | softmax | reads -> out      (main waves 0..7)
//             side waves: the NEXT tile's item rows -- pair index -> item id -> row, one step per phase -- into the other
//             sEi buffer; under a segment's first tile also the next segment's head rows -> sH, in two pieces of 16 rows per
//             side wave, each issued one phase and written the next
// What was measured on the way and not kept (LDS-DMA for the rows, one relation per wave on v_mfma_f32_4x4x1, ...): HISTORY.md.
// Results are those of key_addr_dense_kernel up to the order of the h-set read's sum (four rows per step instead of one); the
// attention reads are the same products in the same order, bit for bit.
#include <cstdlib>
#include <utility>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float kas_exp(float x) {          // exp for x <= 0, as kad_exp (mvin_keyaddr_dense.hip)
    const float t = x * 1.44269502162933349609375f;
    const float n = rintf(t);
    float f = fmaf(x, 1.44269502162933349609375f, -n);
    f = fmaf(x, 1.925963033500011e-8f, f);
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

__device__ long long g_kas_trace[64 * 16];    // MVIN_KA_TRACE=1: workgroup 0 stamps its phase boundaries (scripts/trace_keyaddr.py)

constexpr int kSW = 12;       // waves per workgroup (168 VGPRs each: the resident R_KGE fragments; one workgroup per CU)
constexpr int kSSide = 4;     // the last kSSide waves: the loads of the tile phases (item chain, the next segment's head rows)
constexpr int kRT = 2;        // 16-row MFMA row tiles per tile of pairs: they share every B operand read and every barrier
constexpr int kST = 16 * kRT; // pairs per tile (a C3 user has ~22 pairs: one pass of the three tile phases instead of two)
constexpr int kSMaxRows = 128;  // rows per user (P * NmP) the staging registers hold: 4 side waves x 2 x 4 loads x 4 rows (heads); 12 waves x 3 x 4 (tails)

// ---- the record of one user (int32 words; every section starts on a multiple of 4, the record is whole 256-byte lines) ----
__host__ __device__ constexpr int kas_pad4(int v) { return (v + 3) & ~3; }
__host__ __device__ constexpr KaRecLayout ka_rec_layout_c(int P, int Nm, int nR) {
    KaRecLayout R{};
    if (P < 1 || P > 8 || Nm < 1 || Nm > 256 || nR < 1 || nR > 4096) return R;
    R.NmP = (Nm + 15) & ~15;
    R.rows = P * R.NmP;
    const int nrl = nR < P * Nm ? nR : P * Nm;
    R.maxtiles = R.rows / 16 + nrl;                        // every bucket wastes less than one tile
    R.o_cnt = 4;                                           // [0] = number of tiles; [1..3] = 0
    R.o_off = R.o_cnt + kas_pad4(nR);
    R.o_trel = R.o_off + kas_pad4(nR);
    R.o_bidx = R.o_trel + kas_pad4(R.maxtiles);
    R.o_head = R.o_bidx + R.maxtiles * 16;
    R.o_tail = R.o_head + R.rows;
    R.o_hr = R.o_tail + R.rows;                            // row of [nR, nE, D] (mvin_project_relations): relation * n_entity + head
    R.len = (R.o_hr + R.rows + 63) & ~63;
    return R;
}
KaRecLayout ka_rec_layout(int P, int Nm, int nR) { return ka_rec_layout_c(P, Nm, nR); }

// One wave per user.  Inside a bucket the rows keep their (hop, m) order, so the record is a function of the ids alone.
__global__ __launch_bounds__(64) void user_records_kernel(const int32_t* __restrict__ uts, int n_user, int P, int Nm, int nR, int n_entity,
                                                           KaRecLayout RL, int32_t* __restrict__ out) {
    extern __shared__ int s_rec[];
    int* sCnt = s_rec;            // [nR] members of the bucket, then: members placed so far
    int* sOff = s_rec + nR;       // [nR] first row of the bucket
    const int u = blockIdx.x, lane = threadIdx.x;
    if (u >= n_user) return;
    int32_t* rec = out + (size_t)u * RL.len;
    const int32_t* ub = uts + (int64_t)u * P * 3 * Nm;
    const unsigned emax = (unsigned)(n_entity - 1);
    if (lane >= 1 && lane < 4) rec[lane] = 0;                // the rest of the record was set to -1 by the launcher
    for (int r = lane; r < nR; r += 64) sCnt[r] = 0;
    __syncthreads();
    auto row_rel = [&](int i) -> int {                       // relation of row i = hop * NmP + m, or -1 for a padding row
        const int hop = i / RL.NmP, m = i - hop * RL.NmP;
        if (i >= RL.rows || m >= Nm) return -1;
        return (int)min((unsigned)ub[(hop * 3 + 1) * Nm + m], (unsigned)(nR - 1));
    };
    for (int i0 = 0; i0 < RL.rows; i0 += 64) {
        const int r = row_rel(i0 + lane);
        if (r >= 0) atomicAdd(&sCnt[r], 1);
    }
    __syncthreads();
    int ntile = 0;
    for (int r0 = 0; r0 < nR; r0 += 64) {
        const int r = r0 + lane;
        const int cnt = r < nR ? sCnt[r] : 0;
        const int tiles = (cnt + 15) >> 4;
        int incl = tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int first = ntile + incl - tiles;
        if (r < nR) {
            rec[RL.o_cnt + r] = cnt;
            rec[RL.o_off + r] = first * 16;
            sOff[r] = first * 16;
            for (int j = 0; j < tiles; ++j) rec[RL.o_trel + first + j] = r;
        }
        ntile += __shfl(incl, 63, 64);
    }
    if (lane == 0) rec[0] = ntile;
    __syncthreads();
    for (int r = lane; r < nR; r += 64) sCnt[r] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < RL.rows; i0 += 64) {
        const int i = i0 + lane;
        const int r = row_rel(i);
        int before = 0;
        bool last = true;                                    // no later lane of this round shares the relation
        for (int l = 0; l < 64; ++l) {
            const int rl = __shfl(r, l, 64);
            if (rl == r && l < lane) ++before;
            if (rl == r && l > lane) last = false;
        }
        if (r >= 0) {
            const int at = sCnt[r] + before;
            rec[RL.o_bidx + sOff[r] + at] = i;
            const int hop = i / RL.NmP, m = i - hop * RL.NmP;
            const int hd = (int)min((unsigned)ub[(hop * 3 + 0) * Nm + m], emax);
            rec[RL.o_head + i] = hd;
            rec[RL.o_tail + i] = (int)min((unsigned)ub[(hop * 3 + 2) * Nm + m], emax);
            const long long hr = (long long)r * n_entity + hd;
            rec[RL.o_hr + i] = hr < (1ll << 31) ? (int)hr : 0;       // (tables that large never take the gathered form)
        }
        __syncthreads();
        if (r >= 0 && last) sCnt[r] += before + 1;
        __syncthreads();
    }
}

hipError_t launch_user_records(const int32_t* uts, int n_user, int P, int Nm, int nR, int n_entity, int32_t* out, hipStream_t st) {
    const KaRecLayout RL = ka_rec_layout(P, Nm, nR);
    if (RL.len == 0) return hipErrorInvalidValue;
    // every word the kernel does not write (unused bucket slots, padding rows, section padding) reads -1
    hipError_t e = hipMemsetAsync(out, 0xFF, (size_t)n_user * RL.len * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    user_records_kernel<<<n_user, 64, (size_t)2 * nR * sizeof(int), st>>>(uts, n_user, P, Nm, nR, n_entity, RL, out);
    return hipGetLastError();
}

struct KaStaticLds {
    int h, u, t, ei, l, z, hset, rec, orig, desc, total;     // word offsets; total in bytes
};

__host__ __device__ constexpr KaStaticLds ka_static_layout(int P, const KaRecLayout& RL) {
    KaStaticLds L{};
    constexpr int D = 64;
    int o = 0;
    auto take = [&](int words) { const int at = o; o += (words + 3) & ~3; return at; };
    L.rec = take(2 * RL.len);
    L.h = take(RL.rows * (D + 4));
    L.u = take((RL.rows + 1) * (D + 4));                     // + a spare row: where the U tiles' padding rows are written
    L.t = take(RL.rows * D);                                 // rows unpadded (16-byte chunks swizzled)
    L.ei = take(2 * kST * (D + 4));                          // this tile's item rows and the next tile's
    L.l = take(kST * (RL.rows + 2));
    L.z = take(kST * P);
    L.hset = take(D);
    L.orig = take(2 * kST);
    L.desc = take(4);
    L.total = o * 4;
    return L;
}

// n (<= 16) consecutive 256-byte lines, global -> LDS, same offset on both sides
template <int J>
__device__ __forceinline__ void kas_dma_line(const char* p, int lane4, int n) {
    if (J < n) asm volatile("global_load_lds_dword %0, %1 offset:%2" ::"v"(lane4), "s"(p), "n"(J * 256) : "memory");
}
template <int... J>
__device__ __forceinline__ void kas_dma_lines(const char* p, int lane4, unsigned m0, int n, std::integer_sequence<int, J...>) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0) : "memory");
    (kas_dma_line<J>(p, lane4, n), ...);
}

// CP > 0: an instance for ONE shape (P = CP hops of Nm = CNM memories, CNR relations): every offset of the record and of the LDS
// layout, every row stride and trip count is a constant -- the generic instance keeps ~60 of them in scalar registers, 130-180
// scalar registers spilled to vector lanes, a v_readlane in front of most address computations
// ER: the U rows are GATHERED -- U_m = R_KGE[r_m] . E[h_m] depends on (entity, relation) alone, and mvin_project_relations
// has written it for every such pair ([nR, nE, D], once per call) -- and the h-set read takes its logits E[h] . w from a per-entity
// table of the same call: no U tiles, no resident R_KGE fragments, the h-set read is a softmax over Nm table values + one weighted
// sum of the head rows at hand.
template <bool TRACE, int CP, int CNM, int CNR, bool ER = false>
__global__ __launch_bounds__(kSW * 64, 1) void key_addr_static_kernel(KeyAddrGroupedArgs a_, KaRecLayout RL_, KaStaticLds L_) {
    const KaRecLayout RL = CP > 0 ? ka_rec_layout_c(CP, CNM, CNR) : RL_;
    const KaStaticLds L = CP > 0 ? ka_static_layout(CP, ka_rec_layout_c(CP > 0 ? CP : 1, CNM > 0 ? CNM : 16, CNR > 0 ? CNR : 1)) : L_;
    KeyAddrGroupedArgs a = a_;
    if (CP > 0) a.P = CP, a.Nm = CNM, a.nR = CNR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = 64, LPR = 16, RPW = 4, NT = 4, KS = 16, LDH = D + 4, LDT = D, NTHR = kSW * 64;
    const int P = a.P, Nm = a.Nm, NmP = RL.NmP, PN = RL.rows, LDL = PN + 2;
    float* sH = smem + L.h;
    float* sU = smem + L.u;
    float* sT = smem + L.t;                                  // [PN][D] tail rows
    float* sEi = smem + L.ei;                                // [2][kST][LDH]
    float* sL = smem + L.l;
    float* sZ = smem + L.z;
    float* sHset = smem + L.hset;
    int* sRec = reinterpret_cast<int*>(smem + L.rec);        // [2][RL.len]
    int* sOrig = reinterpret_cast<int*>(smem + L.orig);      // [2][kST] original pair index (-1: padding)
    int* sDesc = reinterpret_cast<int*>(smem + L.desc);      // {user, first pair, end} of the segment after the next

    const int wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // The lane number is LAUNDERED at the top of every segment and tile (relane): left alone, the compiler hoists every per-lane
    // address of every phase out of the segment loop as its invariants and then spills them -- and with them parts of the
    // resident R_KGE fragments, reloaded from scratch in the middle of the U tiles' MFMA chains (142 spilled registers)
    // (and nothing derived from it is a variable: KAS_Q16 / KAS_L16 / KAS_TID / KAS_ITID are recomputed where they are used)
    int lane = threadIdx.x & 63;
    auto relane = [&]() { asm volatile("" : "+v"(lane)); };
#define KAS_Q16 (lane >> 4)
#define KAS_L16 (lane & 15)
#define KAS_TID (wave_u * 64 + lane)
#define KAS_ITID (wave_u * 64 + lane - (NTHR - 16 * LPR))    /* >= 0: a thread of the side waves (16 rows x 16 lanes) */
    const bool has_set = a.w != nullptr;
    const bool side = wave_u >= kSW - kSSide;
    const int G = (int)gridDim.x;
    const unsigned emax = (unsigned)__builtin_amdgcn_readfirstlane(a.n_entity > 0 ? a.n_entity - 1 : 0x7fffffff);
    auto item_id = [&](int o) -> unsigned {
        const unsigned v = a.items64 ? reinterpret_cast<const unsigned*>(a.items64)[2 * (int64_t)o] : (unsigned)a.items32[o];
        return min(v, emax);
    };
    const int slot0 = has_set ? 1 : 0;
    const unsigned ldo32 = (unsigned)a.ldo;                  // output offsets in 32-bit arithmetic (key_addr_static_applies: B * ldo < 2^31)
    const int nseg = a.nseg_dev ? *a.nseg_dev : a.nseg;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- the item row of a pair hangs on three dependent loads (pair index -> item id -> E row): fetched one tile ahead by the
    //      side waves, one step per tile phase, so that every wait falls where these waves would stand at a barrier anyway ----
    int st_o[kRT];
    unsigned st_item[kRT];
    bool st_act[kRT], st_valid[kRT];                         // a side thread carries row (its own) of every row tile
    auto chain_a = [&](int t0, int p1) {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            st_act[rt] = KAS_ITID >= 0 && t0 + 16 * rt < p1; // (a row tile past the segment's pairs is not fetched at all)
            st_valid[rt] = false;
            st_o[rt] = 0, st_item[rt] = 0;                   // (assigned on every path: not carried from tile to tile)
            if (st_act[rt]) {
                const int p = t0 + 16 * rt + KAS_ITID / LPR;
                st_valid[rt] = p < p1;
                st_o[rt] = a.pair_index[st_valid[rt] ? p : p1 - 1];
            }
        }
    };
    auto chain_b = [&]() {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt)
            if (st_act[rt]) st_item[rt] = item_id(st_o[rt]);
    };
    auto chain_c = [&](int buf) {                            // rows -> sEi[buf] (the wait for them stands here: the side waves are idle)
        float4 e[kRT];
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
            e[rt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st_act[rt]) e[rt] = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.E) + (size_t)st_item[rt] * D)[KAS_ITID % LPR];
        }
        if (KAS_ITID >= 0) {
#pragma unroll
            for (int rt = 0; rt < kRT; ++rt) {
                const int i = 16 * rt + KAS_ITID / LPR, cc = KAS_ITID % LPR;
                float* dst = sEi + (buf * kST + i) * LDH + 4 * cc;
                *reinterpret_cast<float2*>(dst) = make_float2(e[rt].x, e[rt].y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(e[rt].z, e[rt].w);
                if (cc == 0) sOrig[buf * kST + i] = st_act[rt] && st_valid[rt] ? st_o[rt] : -1;
            }
        }
    };

    // ---- segment descriptors: this one, the next, and (read during the U phase) the one after ----
    auto desc = [&](int seg, int& u, int& p0, int& p1) {
        u = 0, p0 = 0, p1 = 0;
        if (seg < nseg) {
            u = a.seg_user[seg];
            p0 = a.seg_ptr[seg];
            p1 = a.seg_ptr[seg + 1];
        }
    };
    int u0, p00, p10, u1, p01, p11;
    desc((int)blockIdx.x, u0, p00, p10);
    desc((int)blockIdx.x + G, u1, p01, p11);

    // ---- R_KGE fragments resident in registers (as key_addr_dense_kernel), in PAIRS: task t = (relation r = t / 2, column tiles
    //      2 (t % 2) and 2 (t % 2) + 1) belongs to wave t % 11, task slot t / 11 (two task slots = four fragments per wave).  One
    //      read of a row tile's index and A operand then feeds two accumulator chains of 16 MFMAs (a task of one column tile
    //      spent ~0.6 k cycles on those reads and its loop per 512 of MFMA issue).  Contraction index permuted so that a lane's
    //      values are contiguous. ----
    constexpr int RES = 4;
    const bool resident = !ER && a.nR * NT <= RES * (kSW - 1);
    float rb[RES][KS];
    auto load_bfrag = [&](int r, int nt, float (&bf)[KS]) {
        const float* Rr = a.R + (size_t)r * D * D + (size_t)(16 * nt + KAS_L16) * D + KS * KAS_Q16;   // R[r][n][k]
#pragma unroll
        for (int k = 0; k < KS; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(Rr + k);
            bf[k] = v.x;
            bf[k + 1] = v.y;
            bf[k + 2] = v.z;
            bf[k + 3] = v.w;
        }
    };
    if (resident && wave < kSW - 1) {
#pragma unroll
        for (int ts = 0; ts < RES / 2; ++ts) {
            const int t = wave_u + (kSW - 1) * ts;           // task: relation t / 2, column tiles 2 (t % 2) + {0, 1}
            if (t < a.nR * NT / 2) {
                load_bfrag(t / 2, 2 * (t % 2), rb[2 * ts]);
                load_bfrag(t / 2, 2 * (t % 2) + 1, rb[2 * ts + 1]);
            }
        }
    }
    // ---- a user's 2 * PN rows, staged through registers: 16 bytes per lane, four rows per load.  A burst of 128 rows (32 KB)
    //      takes 2.5 - 6 k cycles to come back whoever issues it and however (what the CU keeps in flight bounds it; LDS-DMA, one
    //      256-byte row per instruction and no registers, is no faster and costs the issuing wave ~85 cycles per row --
    //      _ubench/lds_dma.hip), which is longer than a tile phase: so the rows travel in pieces, issued early and written late --
    //        tail rows of s   : every wave 12 rows (three loads), issued at the segment's top, written behind the wave's U work;
    //        head rows of s+1 : the side waves, 2 x 16 rows each, under tile 0's logits -> softmax and softmax -> reads phases.
    //      (the registers are local to a piece: one array at function scope was live across the whole segment loop) ----
    const int sj = wave_u - (kSW - kSSide);                  // side wave number (< 0: not a side wave)
    auto stage_issue = [&](auto& stg, const int* ids, int base, const float* tab = nullptr) {   // rows base + 4 b + (lane >> 4); ids: clamped row ids in LDS (-1: padding); tab: the table (default E)
        constexpr int N = sizeof(stg) / sizeof(float4);
        const int g_ = KAS_Q16, c_ = KAS_L16;
        // all ids, then all loads, no branch in between: one LDS and one memory latency per piece (a padding row or a row past PN
        // reads row 0 of the table; stage_write drops it)
        int idr[N];
#pragma unroll
        for (int b = 0; b < N; ++b) idr[b] = ids[min(base + 4 * b + g_, PN - 1)];
#pragma unroll
        for (int b = 0; b < N; ++b)
            stg[b] = reinterpret_cast<const float4*>((tab ? tab : reinterpret_cast<const float*>(a.E)) + (size_t)(unsigned)max(idr[b], 0) * D)[c_];
    };
    // sH rows are padded (LDH = D + 4).  sT rows are not: the 16-byte chunk c of row m sits at position c ^ ((m & 3) << 2), which
    // puts the four rows of a reads-phase B operand (m = q16 + 4 j) into four different 16-bank groups
    auto stage_write = [&](const auto& stg, const int* ids, int base, float* dst, int ld, bool swz) {
        constexpr int N = sizeof(stg) / sizeof(float4);
        const int g_ = KAS_Q16, c_ = swz ? KAS_L16 ^ (KAS_Q16 << 2) : KAS_L16;      // (row & 3 == lane >> 4: base is a multiple of 4)
#pragma unroll
        for (int b = 0; b < N; ++b) {
            const int row = base + 4 * b + g_;
            if (row < PN) {
                const bool real = ids[row] >= 0;             // padding rows read as zero
                float* d = dst + (size_t)row * ld + 4 * c_;
                *reinterpret_cast<float2*>(d) = real ? make_float2(stg[b].x, stg[b].y) : make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(d + 2) = real ? make_float2(stg[b].z, stg[b].w) : make_float2(0.f, 0.f);
            }
        }
    };
    constexpr int kHalf = kSMaxRows / (2 * kSSide);          // head rows per side wave and piece (16)
    auto heads_now = [&](const int* rec_) {                  // both pieces at once (prologue; a segment without pairs)
        if (sj >= 0) {
            float4 stg[kSMaxRows / (4 * kSSide)];
            stage_issue(stg, rec_ + RL.o_head, sj * 2 * kHalf);
            stage_write(stg, rec_ + RL.o_head, sj * 2 * kHalf, sH, LDH, false);
        }
    };
    auto dma_record = [&](int user, int par) {
        const char* src = reinterpret_cast<const char*>(a.records + (size_t)user * RL.len);
        const int lines = RL.len >> 6;
        for (int b = 0; b < lines; b += 16)
            kas_dma_lines(src + (size_t)b * 256, lane * 4, lds0 + (unsigned)((L.rec + par * RL.len) * 4) + (unsigned)(b * 256), lines - b,
                          std::make_integer_sequence<int, 16>{});
    };
    // U tile: rows bidx[row0 .. row0 + 15] of sH times the fragment -> sU
    auto u_tile = [&](const int* sBidx, int row0, int nt, const float (&bf)[KS]) {
        const int ia = sBidx[row0 + KAS_L16];
        const float4* ar = reinterpret_cast<const float4*>(sH + (ia >= 0 ? ia : 0) * LDH + KS * KAS_Q16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; k += 4) {
            const float4 av = ar[k >> 2];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bf[k], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bf[k + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bf[k + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bf[k + 3], acc, 0, 0, 0);
        }
        const int4 io4 = *reinterpret_cast<const int4*>(sBidx + row0 + 4 * KAS_Q16);
        const int io[4] = {io4.x, io4.y, io4.z, io4.w};
        // unconditional: the rows of an empty bucket slot (-1) go to a spare row behind sU (predicated, each of the four writes
        // was a compare, an exec-mask save / restore and a branch: 270 cycles per tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) sU[(io[i] >= 0 ? io[i] : PN) * LDH + 16 * nt + KAS_L16] = acc[i];
    };

    // two column tiles (nt0, nt0 + 1) of one row tile: one index / A operand read, two accumulator chains
    auto u_tile2 = [&](const int* sBidx, int row0, int nt0, const float (&bfa)[KS], const float (&bfb)[KS]) {
        const int ia = sBidx[row0 + KAS_L16];
        const float4* ar = reinterpret_cast<const float4*>(sH + (ia >= 0 ? ia : 0) * LDH + KS * KAS_Q16);
        f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; k += 4) {
            const float4 av = ar[k >> 2];
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bfa[k], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bfb[k], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bfa[k + 1], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bfb[k + 1], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bfa[k + 2], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bfb[k + 2], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bfa[k + 3], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bfb[k + 3], cb, 0, 0, 0);
        }
        const int4 io4 = *reinterpret_cast<const int4*>(sBidx + row0 + 4 * KAS_Q16);
        const int io[4] = {io4.x, io4.y, io4.z, io4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* d = sU + (io[i] >= 0 ? io[i] : PN) * LDH + 16 * nt0 + KAS_L16;      // (-1: the spare row, see u_tile)
            d[0] = ca[i];
            d[16] = cb[i];
        }
    };

    // ---- prologue: padding rows are never landed (zero for good); record + head rows of the first segment ----
    for (int i = KAS_TID; i < PN * LDH; i += NTHR) sH[i] = 0.f;
    for (int i = KAS_TID; i < PN * LDT; i += NTHR) sT[i] = 0.f;
    const bool any = (int)blockIdx.x < nseg;
    if (any && wave_u == kSW - 1) {
        dma_record(u0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (any) heads_now(sRec);                                // (its tail rows: under the first U phase like every segment's)
    int ep = 0;                                              // sEi / sOrig buffer of the tile at hand
    chain_a(p00, p10);
    chain_b();
    chain_c(0);

    int iter = 0;
    auto stamp = [&](int slot) {
        if constexpr (TRACE) {
            if (blockIdx.x == 0 && KAS_TID == 64 * a.dbg && iter >= 4 && iter < 68) g_kas_trace[(iter - 4) * 16 + slot] = __builtin_readcyclecounter();
        }
    };
    for (int seg = blockIdx.x; seg < nseg; seg += G, ++iter) {
        relane();
        stamp(0);
        const int p0 = p00, p1 = p10;
        const int par = iter & 1;
        const bool has_next = seg + G < nseg;
        const int* rec = sRec + par * RL.len;                // this segment's record (landed a segment ago)
        const int* recn = sRec + (par ^ 1) * RL.len;
        const int* sBidx = rec + RL.o_bidx;
        __syncthreads();                                     // previous segment fully consumed
        stamp(13);
        // this segment's tail rows (first needed by tile 0's reads): 12 rows per wave, in flight under the wave's U work and
        // written behind it (three loads: the memory latency, ~2.5 k cycles here, is paid once and under the MFMA tiles)
        float4 tl[3];
        const bool has_tl = wave_u * 12 < PN;               // (128 rows: waves 0 .. 10; wave 11, the U phase's longest, loads none)
        if (has_tl) stage_issue(tl, rec + RL.o_tail, wave_u * 12);
        float4 ul[3];                                        // ER: this segment's U rows, 12 per wave like the tail rows
        if constexpr (ER) {
            if (has_tl) stage_issue(ul, rec + RL.o_hr, wave_u * 12, a.ER);
        }
        // fire and forget: the record of the next segment (waited for behind this wave's h-set read)
        if (wave_u == kSW - 1 && has_next) dma_record(u1, par ^ 1);
        // the descriptor of the segment after the next: read by wave 11 only (three registers held across the U tiles by every
        // wave cost a quarter of a resident fragment its registers), handed on through LDS behind the phase's barrier
        const bool d_have = seg + 2 * G < nseg;
        auto desc_load = [&](int& d_u, int& d_p0, int& d_p1) {
            d_u = 0, d_p0 = 0, d_p1 = 0;
            if (d_have) {                                    // vector loads through a lane-dependent zero (see key_addr_dense_kernel)
                int zv = 0;
                asm volatile("" : "+v"(zv));
                const int i = seg + 2 * G + zv;
                d_u = a.seg_user[i];
                d_p0 = a.seg_ptr[i];
                d_p1 = a.seg_ptr[i + 1];
            }
        };
        auto desc_put = [&](int d_u, int d_p0, int d_p1) {
            if (lane == 0) {
                sDesc[0] = d_u;
                sDesc[1] = d_p0;
                sDesc[2] = d_p1;
            }
        };
        stamp(3);
        // ---- h-set read (wave 11) next to the U tiles (waves 0..10) ----
        if (has_set && wave == kSW - 1) {
            // online softmax over the head rows of hop 0, FOUR rows per step and lane group: their loads, dot products and
            // exponentials are independent, the running maximum is touched once per step
            int d_u, d_p0, d_p1;
            desc_load(d_u, d_p0, d_p1);
            if constexpr (ER) {
                // logits from the per-entity table (one memory per lane: NmP <= 64), softmax across the wave, then ONE weighted
                // sum of the head rows at hand, four rows per step
                const int hid = lane < NmP ? rec[RL.o_head + lane] : -1;
                const float v = (lane < Nm && hid >= 0) ? a.hs[hid] : -INFINITY;
                const float M = wave_max(v);
                const float e = lane < Nm ? kas_exp(v - M) : 0.f;
                const float p = e / wave_sum(e);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int j = 0; j < NmP / RPW; ++j) {
                    const int m = RPW * j + KAS_Q16;
                    const float pm = __shfl(p, m, 64);
                    acc = f4_fma(pm, *reinterpret_cast<const float4*>(sH + (size_t)m * LDH + 4 * KAS_L16), acc);
                }
                acc = group_xor_sum(acc, LPR);
                if (KAS_Q16 == 0) *reinterpret_cast<float4*>(sHset + 4 * KAS_L16) = acc;
            } else {
            const float4 w4 = reinterpret_cast<const float4*>(a.w)[KAS_L16];
            constexpr int HB = 4;
            float mx = -INFINITY, z = 0.f;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int m0 = 0; m0 < NmP; m0 += HB * RPW) {
                float4 h4[HB];
                float d[HB];
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int m = m0 + RPW * j + KAS_Q16;
                    h4[j] = m < NmP ? *reinterpret_cast<const float4*>(sH + (size_t)m * LDH + 4 * KAS_L16) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float cm = -INFINITY;
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int m = m0 + RPW * j + KAS_Q16;
                    float v = fmaf(h4[j].x, w4.x, fmaf(h4[j].y, w4.y, fmaf(h4[j].z, w4.z, h4[j].w * w4.w)));
                    v = group_sum(v, 4);
                    d[j] = m < Nm ? v : -INFINITY;           // (the padding rows m >= Nm take no part)
                    cm = fmaxf(cm, d[j]);
                }
                const float nm = fmaxf(mx, cm);
                if (nm > -INFINITY) {
                    const float sc = mx == -INFINITY ? 0.f : kas_exp(mx - nm);
                    z *= sc;
                    acc = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        const float e = d[j] == -INFINITY ? 0.f : kas_exp(d[j] - nm);
                        z += e;
                        acc = f4_fma(e, h4[j], acc);
                    }
                    mx = nm;
                }
            }
            // merge the lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48: the same column chunk of different row groups)
            float M = xor32_max(xor16_max(mx));
            const float f = mx == -INFINITY ? 0.f : kas_exp(mx - M);
            z *= f;
            acc = make_float4(acc.x * f, acc.y * f, acc.z * f, acc.w * f);
            acc = group_xor_sum(acc, LPR);
            z = group_xor_sum(make_float4(z, 0.f, 0.f, 0.f), LPR).x;
            const float inv = 1.f / z;
            if (KAS_Q16 == 0) *reinterpret_cast<float4*>(sHset + 4 * KAS_L16) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
            }
            desc_put(d_u, d_p0, d_p1);
        } else if constexpr (!ER) {
            if (resident) {
                if (wave < kSW - 1) {
                    // wave-uniform task table first (scalars)
                    int ntl[RES / 2], rw[RES / 2];
#pragma unroll
                    for (int ts = 0; ts < RES / 2; ++ts) {
                        const int t = wave_u + (kSW - 1) * ts;
                        const bool ok = t < a.nR * NT / 2;
                        const int r = ok ? t / 2 : 0;
                        ntl[ts] = ok ? (rec[RL.o_cnt + r] + 15) >> 4 : 0;
                        rw[ts] = rec[RL.o_off + r];
                    }
#pragma unroll
                    for (int ts = 0; ts < RES / 2; ++ts) {
                        const int nt0 = 2 * ((wave_u + (kSW - 1) * ts) % 2);
                        const int tiles = __builtin_amdgcn_readfirstlane(ntl[ts]), row0 = __builtin_amdgcn_readfirstlane(rw[ts]);
                        for (int j = 0; j < tiles; ++j) u_tile2(sBidx, row0 + 16 * j, nt0, rb[2 * ts], rb[2 * ts + 1]);
                    }
                }
            } else {
                const int nw = has_set ? kSW - 1 : kSW;
                const int ntile = __builtin_amdgcn_readfirstlane(rec[0]);
                for (int task = wave_u; task < ntile * NT; task += nw) {
                    const int tl = task / NT, nt = task - tl * NT;
                    float bfrag[KS];
                    load_bfrag(rec[RL.o_trel + tl], nt, bfrag);
                    u_tile(sBidx, 16 * tl, nt, bfrag);
                }
            }
            stamp(9);
            // padding memories: U rows never written by a tile must read as zero
            if (Nm < NmP) {
                for (int i = KAS_TID; i < PN * NT; i += (has_set ? (kSW - 1) : kSW) * 64) {
                    const int row = i / NT, nt = i - row * NT;
                    if (rec[RL.o_tail + row] < 0) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) sU[(size_t)row * LDH + 16 * nt + j] = 0.f;
                    }
                }
            }
        }
        if (!has_set && wave_u == kSW - 1) {
            int d_u, d_p0, d_p1;
            desc_load(d_u, d_p0, d_p1);
            desc_put(d_u, d_p0, d_p1);
        }
        if (has_tl) stage_write(tl, rec + RL.o_tail, wave_u * 12, sT, LDT, true);
        if constexpr (ER) {
            if (has_tl) stage_write(ul, rec + RL.o_hr, wave_u * 12, sU, LDH, false);     // (padding rows: zeros)
        }
        // wave 11: the next segment's record has landed before the tiles' first barrier
        if (wave_u == kSW - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // behind the phase's barrier: the descriptors move up by one
        auto rotate = [&]() {
            u0 = u1, p00 = p01, p10 = p11;
            u1 = __builtin_amdgcn_readfirstlane(sDesc[0]);
            p01 = __builtin_amdgcn_readfirstlane(sDesc[1]);
            p11 = __builtin_amdgcn_readfirstlane(sDesc[2]);
        };
        stamp(4);
        // ---- the user's pairs, 16 at a time: three barriers per tile (logits | softmax | reads).  Every phase is an if (side) /
        //      else: the side waves' staging registers are then not live inside the main waves' loops (one code path for both cost
        //      two of the four resident fragments their registers).  Side waves per tile: the NEXT tile's item rows, one step of
        //      the chain per phase, into the other sEi buffer; on a segment's first tile also the next segment's head and tail
        //      rows, eight 16-byte loads per lane, issued one phase and written the next. ----
        constexpr int kMW = kSW - kSSide;                    // main waves: every logits / softmax / reads task
        for (int t0 = p0; t0 < p1; t0 += kST, ep ^= 1) {
            relane();
            const bool last = t0 + kST >= p1;                // p00 / p10: by now the NEXT segment's (none: an empty range)
            const bool heads = t0 == p0 && has_next;         // this tile carries the next segment's rows
            const int nrt = p1 - t0 > 16 ? kRT : 1;          // row tiles of this tile that hold pairs
            float4 stg[kHalf / 4];                           // (side waves, under `heads`; live across one barrier each time)
            const float* sEc = sEi + ep * kST * LDH;
            const int* sOc = sOrig + ep * kST;
            __syncthreads();                                 // sU / sHset complete; previous tile consumed, this tile's rows in sEi
            relane();
            if (t0 == p0) rotate();
            if (t0 == p0) stamp(5);
            if (side) {
                chain_a(last ? p00 : t0 + kST, last ? p10 : p1);
                // the next segment's head rows: sH is free since this barrier, the record landed during the U phase
                if (heads) stage_issue(stg, recn + RL.o_head, sj * 2 * kHalf);
            } else {
                // logits L[pair, m] = E[item_pair] . U_m : one 16-memory tile per task
                for (int mt = wave; mt < PN / 16; mt += kMW) {
                    f32x4 acc[kRT];
                    const float4* ar = reinterpret_cast<const float4*>(sEc + KAS_L16 * LDH + KS * KAS_Q16);
                    const float4* br = reinterpret_cast<const float4*>(sU + (16 * mt + KAS_L16) * LDH + KS * KAS_Q16);
#pragma unroll
                    for (int rt = 0; rt < kRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (nrt == kRT) {                        // the row tiles share the B operand; their accumulator chains interleave
#pragma unroll
                        for (int k = 0; k < KS / 4; ++k) {
                            const float4 bv = br[k];
                            float4 av[kRT];
#pragma unroll
                            for (int rt = 0; rt < kRT; ++rt) av[rt] = ar[rt * 16 * LDH / 4 + k];
#pragma unroll
                            for (int rt = 0; rt < kRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].x, bv.x, acc[rt], 0, 0, 0);
#pragma unroll
                            for (int rt = 0; rt < kRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].y, bv.y, acc[rt], 0, 0, 0);
#pragma unroll
                            for (int rt = 0; rt < kRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].z, bv.z, acc[rt], 0, 0, 0);
#pragma unroll
                            for (int rt = 0; rt < kRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].w, bv.w, acc[rt], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < KS / 4; ++k) {
                            const float4 av = ar[k], bv = br[k];
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[0], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[0], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[0], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[0], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < kRT; ++rt) {
                        if (rt < nrt) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) sL[(size_t)(16 * rt + 4 * KAS_Q16 + i) * LDL + 16 * mt + KAS_L16] = acc[rt][i];
                        }
                    }
                }
            }
            if (t0 == p0) stamp(10);
            __syncthreads();
            relane();
            if (t0 == p0) stamp(6);
            if (side) {
                chain_b();
                if (heads) {
                    stage_write(stg, recn + RL.o_head, sj * 2 * kHalf, sH, LDH, false);
                    stage_issue(stg, recn + RL.o_head, sj * 2 * kHalf + kHalf);      // the second piece
                }
            } else {
                // softmax over the Nm memories of every (pair, hop) (:223): un-normalised weights back to sL, 1/sum to sZ; four rows
                // per wave pass, a 16-lane DPP row per (pair, hop)
                const int rg = lane >> 4, cl = lane & 15;
                for (int base = wave * 4; base < 16 * nrt * P; base += kMW * 4) {
                    const int task = base + rg;
                    const bool ok = task < 16 * nrt * P;
                    const int tk = ok ? task : base;
                    const int pi = tk / P, hop = tk - pi * P;
                    float* row = sL + (size_t)pi * LDL + hop * NmP;
                    float mx = -INFINITY, z = 0.f;
                    if (NmP <= 64) {
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = (cl + 16 * u < Nm) ? row[cl + 16 * u] : -INFINITY;
                        mx = group_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), 4);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float e = (cl + 16 * u < Nm) ? kas_exp(v[u] - mx) : 0.f;
                            if (ok && cl + 16 * u < NmP) row[cl + 16 * u] = e;
                            z += e;
                        }
                    } else {
                        for (int m = cl; m < Nm; m += 16) mx = fmaxf(mx, row[m]);
                        mx = group_max(mx, 4);
                        for (int m = cl; m < NmP; m += 16) {
                            const float e = m < Nm ? kas_exp(row[m] - mx) : 0.f;
                            if (ok) row[m] = e;
                            z += e;
                        }
                    }
                    z = group_sum(z, 4);
                    if (ok && cl == 0) sZ[task] = 1.f / z;      // sZ[pi * P + hop]
                }
            }
            if (t0 == p0) stamp(11);
            __syncthreads();
            relane();
            if (t0 == p0) stamp(7);
            if (side) {
                if (heads) stage_write(stg, recn + RL.o_head, sj * 2 * kHalf + kHalf, sH, LDH, false);
                chain_c(ep ^ 1);
            } else {
                // reads o[pair, hop, :] = sum_m p[pair, m] t_m : one (hop, 16-column tile) per task
                for (int task = wave; task < P * NT; task += kMW) {
                    const int hop = task / NT, nt = task - hop * NT;
                    f32x4 acc[kRT];
#pragma unroll
                    for (int rt = 0; rt < kRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float* ar = sL + (size_t)KAS_L16 * LDL + hop * NmP + KAS_Q16;
                    const float* br = sT + (size_t)(hop * NmP + KAS_Q16) * LDT + 16 * (nt ^ KAS_Q16) + KAS_L16;     // (swizzled chunks: row & 3 == KAS_Q16)
                    if (nrt == kRT) {
                        for (int k = 0; k < NmP / 4; k += 4) {
                            float av[kRT][4], bv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                bv[u] = br[(size_t)4 * (k + u) * LDT];
#pragma unroll
                                for (int rt = 0; rt < kRT; ++rt) av[rt][u] = ar[(size_t)16 * rt * LDL + 4 * (k + u)];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
#pragma unroll
                                for (int rt = 0; rt < kRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][u], bv[u], acc[rt], 0, 0, 0);
                            }
                        }
                    } else {
                        for (int k = 0; k < NmP / 4; k += 4) {
                            float av[4], bv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                av[u] = ar[4 * (k + u)];
                                bv[u] = br[(size_t)4 * (k + u) * LDT];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc[0], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < kRT; ++rt) {
                        if (rt < nrt) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int pi = 16 * rt + 4 * KAS_Q16 + i;
                                const int orig = sOc[pi];
                                if (orig >= 0)
                                    a.out[(size_t)((unsigned)orig * ldo32 + (unsigned)((slot0 + hop) * D + 16 * nt + KAS_L16))] = acc[rt][i] * sZ[pi * P + hop];
                            }
                        }
                    }
                }
                if (has_set) {
                    for (int i = KAS_TID; i < 16 * nrt * LPR; i += kMW * 64) {
                        const int pi = i / LPR, cc = i - pi * LPR;
                        const int orig = sOc[pi];
                        if (orig >= 0)
                            *reinterpret_cast<float4*>(a.out + (size_t)((unsigned)orig * ldo32 + (unsigned)(4 * cc))) = *reinterpret_cast<const float4*>(sHset + 4 * cc);
                    }
                }
            }
            if (t0 == p0) stamp(8);
        }
        if (p0 >= p1) {                                      // a segment without pairs still hands sH and the item chain on
            __syncthreads();
            rotate();
            if (has_next) heads_now(recn);
            chain_a(p00, p10);
            chain_b();
            chain_c(ep);
        }
    }
}
#undef KAS_Q16
#undef KAS_L16
#undef KAS_TID
#undef KAS_ITID

bool key_addr_static_supported(int D, int P, int Nm, int nR) {
    if (D != 64) return false;
    const KaRecLayout RL = ka_rec_layout(P, Nm, nR);
    if (RL.len == 0 || RL.rows < 64 || RL.rows > kSMaxRows) return false;
    return ka_static_layout(P, RL).total <= 160 * 1024;
}

bool key_addr_static_applies(const KeyAddrGroupedArgs& a, int table_bf16) {
    static const bool off = getenv("MVIN_KA_STATIC") && atoi(getenv("MVIN_KA_STATIC")) == 0;
    return !off && a.records != nullptr && !table_bf16 && key_addr_static_supported(a.D, a.P, a.Nm, a.nR) &&
           (int64_t)a.nseg * a.ldo < (int64_t(1) << 31);       // (nseg = the batch size: the bound the caller gives for the segments)
}

// the gathered form (KeyAddrGroupedArgs::ER): one memory per lane in the h-set read, 32-bit row numbers in the records
bool key_addr_static_er_ok(int P, int Nm, int nR, int n_entity, bool has_set) {
    const KaRecLayout RL = ka_rec_layout(P, Nm, nR);
    return RL.len != 0 && (!has_set || RL.NmP <= 64) && (long long)nR * n_entity < (1ll << 31);
}

// s[e] = E[e] . w for every entity (the h-set read's logits), one wave per 4 rows
__global__ __launch_bounds__(256) void entity_dot_kernel(const float* __restrict__ E, const float* __restrict__ w, int n, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + g;
    float s = 0.f;
    if (row < n)
        for (int d = 4 * c; d < D; d += 64) {
            const float4 e4 = *reinterpret_cast<const float4*>(E + row * D + d), w4 = *reinterpret_cast<const float4*>(w + d);
            s = fmaf(e4.x, w4.x, fmaf(e4.y, w4.y, fmaf(e4.z, w4.z, fmaf(e4.w, w4.w, s))));
        }
    s = group_sum(s, 4);
    if (row < n && c == 0) out[row] = s;
}

// RT[r][k][n] = R[r][n][k]: the B operand of the table build (mvin_linear_fwd multiplies rows by W[k][n])
__global__ void transpose_blocks_kernel(const float* __restrict__ R, int D, float* __restrict__ RT) {
    const int r = blockIdx.x / D, k = blockIdx.x % D, n = threadIdx.x;
    if (n < D) RT[((size_t)r * D + k) * D + n] = R[((size_t)r * D + n) * D + k];
}

hipError_t launch_transpose_blocks(const float* R, int nR, int D, float* RT, hipStream_t st) {
    transpose_blocks_kernel<<<nR * D, D < 64 ? 64 : D, 0, st>>>(R, D, RT);
    return hipGetLastError();
}

hipError_t launch_entity_dot(const float* E, const float* w, int n, int D, float* out, hipStream_t st) {
    entity_dot_kernel<<<(n + 15) / 16, 256, 0, st>>>(E, w, n, D, out);
    return hipGetLastError();
}

hipError_t kas_read_trace(long long* host_dst, size_t n) {
    const size_t have = sizeof(g_kas_trace) / sizeof(long long);
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_kas_trace), (n < have ? n : have) * sizeof(long long));
}

hipError_t launch_key_addr_static(const KeyAddrGroupedArgs& a, hipStream_t st) {
    const KaRecLayout RL = ka_rec_layout(a.P, a.Nm, a.nR);
    const KaStaticLds L = ka_static_layout(a.P, RL);
    static const bool trace = getenv("MVIN_KA_TRACE") != nullptr;
    // BASELINE.json's metric config (last-fm: 2 hops of 64 memories, 9 relations) has its own instance
    const bool c3 = a.P == 2 && a.Nm == 64 && a.nR == 9 && !(getenv("MVIN_KAS_GENERIC") && atoi(getenv("MVIN_KAS_GENERIC")));
    const bool er = a.ER != nullptr;
    auto k = er ? (c3 ? (trace ? key_addr_static_kernel<true, 2, 64, 9, true> : key_addr_static_kernel<false, 2, 64, 9, true>)
                      : key_addr_static_kernel<false, 0, 0, 0, true>)
                : c3 ? (trace ? key_addr_static_kernel<true, 2, 64, 9> : key_addr_static_kernel<false, 2, 64, 9>)
                     : (trace ? key_addr_static_kernel<true, 0, 0, 0> : key_addr_static_kernel<false, 0, 0, 0>);
    hipError_t e = hipSuccess;
    if (L.total > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
    if (e != hipSuccess) return e;
    KeyAddrGroupedArgs b = a;
    b.dbg = 0;
    if (trace) {
        const char* tw = getenv("MVIN_KA_TRACE_WAVE");
        b.dbg = tw ? atoi(tw) % kSW : 0;
    }
    const int grid = a.nseg < 256 ? a.nseg : 256;            // persistent: one workgroup per CU
    k<<<grid, kSW * 64, L.total, st>>>(b, RL, L);
    return hipGetLastError();
}

}  // namespace mvin
