// MVIN._key_addressing (model.py:161-240) for pairs grouped by user, in its DENSE form: with a user's ripple
// sets fixed for all of the user's pairs, every step of the attention reads is a small matrix product, and all
// of them run on v_mfma_f32_16x16x4_f32 (exact fp32).  Same interface as key_addr_grouped_kernel
// (mvin_keyaddr_grouped.hip); this is the variant the launcher prefers whenever its LDS footprint fits.
//
// Per user segment (one workgroup of 16 waves):
//   stage   : uts[u] ids; head rows h_m and tail rows t_m of every hop -> LDS (sH, sT), once
//   U       : U_m = R_KGE[r_m] . h_m for every memory m -- the reference's own association, model.py:214-216
//             (tf.matmul(r_emb, h_expanded)) -- as row tiles of 16 memories that share a relation (the memories
//             are bucketed by relation in LDS; a bucket is padded to whole tiles): A = h rows, B = R_KGE[r]^T
//             fragments straight from L1/L2, result -> sU in (hop, m) order
//   h-set   : o_hset = sum_m softmax_m(h0_m . w_h) h0_m (:162-197), once per user (one wave, VALU)
//   per tile of 16 pairs:
//     logits : L[pair, m] = E[item_pair] . U_m                       (:219-220)   MFMA, A = item rows, B = sU^T
//     softmax: over the Nm memories of each hop (:223)               one wave per pair
//     reads  : o[pair, hop] = sum_m p[pair, m] t_m                   (:229)       MFMA, A = p, B = sT
// Nothing of size [B, nR, D] or [B, Nm, D, D] exists; a user's 2*P*Nm rows are read once per batch.
#include <cstdlib>
#include <utility>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// exp for arguments <= 0 (logit - max): the library's argument reduction without its range selects (as d16_exp / kw_exp)
__device__ __forceinline__ float kad_exp(float x) {
    const float t = x * 1.44269502162933349609375f;              // float(log2 e)
    const float n = rintf(t);
    float f = fmaf(x, 1.44269502162933349609375f, -n);
    f = fmaf(x, 1.925963033500011e-8f, f);                       // log2 e - float(log2 e)
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// development aid (MVIN_KA_TRACE=1, scripts/trace_keyaddr.py): workgroup 0 stamps s_memtime at its phase boundaries
__device__ long long g_ka_trace[64 * 16];
// waves per workgroup: 12 (168 VGPRs each: room for the resident R_KGE fragments; one workgroup per CU), or 4 at D = 16,
// where a segment's tiles are a quarter of the work and its LDS a quarter of the space: four workgroups per CU instead of
// one, i.e. four user segments' dependent phases in flight per CU (the kernel is latency-bound per segment)
constexpr int ka_dense_waves(int D) { return D == 16 ? 4 : D == 32 ? 8 : 12; }
constexpr int ka_dense_minw(int D) { return D == 32 ? 4 : 1; }       // waves per SIMD the register budget is cut for
// pairs per tile: one 16-row MFMA tile.  Two at D = 64 (a user of the C3 batch has ~22 pairs: one pass of the
// four barrier-separated tile phases instead of two) were measured and dropped: every tile phase took exactly twice
// as long (logits 1.55 -> 3.3 k cycles, softmax 2.6 -> 4.5, reads 2.2 -> 4.3: they are throughput-, not barrier-bound) and the
// two accumulator sets next to the resident R_KGE fragments cost 16-22 spilled registers (U phase 11.4 -> 13.7 k cycles).
constexpr int kDT = 16;      // pairs per tile
constexpr int kKaDmaRows = 12;   // rows a wave lands per LDS-DMA batch (below)

struct KaDenseLds {
    size_t h, u, t, ei, l, z, hset, lg, idh, idt, rel, rank, bidx, cnt, off, tile_rel, tile_row, orig, idn, total;
    int NmP, PN, maxtiles;
};

static KaDenseLds ka_dense_layout(int D, int P, int Nm, int nR) {
    KaDenseLds L{};
    const int Ph = P > 0 ? P : 1;
    L.NmP = (Nm + 15) & ~15;
    L.PN = P * L.NmP;
    const int nrl = nR < P * Nm ? nR : P * Nm;
    L.maxtiles = L.PN / 16 + (nrl > 0 ? nrl : 1);          // every bucket wastes less than one tile
    size_t o = 0;
    auto take = [&](size_t words) { const size_t at = o; o += (words + 3) & ~(size_t)3; return at; };
    L.h = take((size_t)Ph * L.NmP * (D + 4));
    L.u = take((size_t)L.PN * (D + 4));
    L.t = take((size_t)L.PN * (D + 16));
    L.ei = take((size_t)kDT * (D + 4));
    L.l = take((size_t)kDT * (L.PN + 2));
    L.z = take((size_t)kDT * (P > 0 ? P : 1));
    L.hset = take(D);
    L.lg = take(L.NmP);
    L.idh = take((size_t)Ph * L.NmP);
    L.idt = take((size_t)Ph * L.NmP);
    L.rel = take((size_t)Ph * L.NmP);
    L.rank = take((size_t)Ph * L.NmP);
    L.bidx = take((size_t)L.maxtiles * 16);
    L.cnt = take(nR);
    L.off = take(nR);
    L.tile_rel = take(L.maxtiles);
    L.tile_row = take(L.maxtiles);
    L.orig = take(kDT + 4);
    // LDS-DMA form (ka_dense_dma_applies): {head, tail, relation} ids of this and of the next segment
    L.idn = take((D == 64 && P > 0 && L.PN >= 64 && L.PN <= kKaDmaRows * ka_dense_waves(64)) ? (size_t)2 * 3 * L.PN : 0);
    L.total = o * 4;
    return L;
}

// LDS-DMA form (D = 64, fp32 table): a table row is 256 bytes = one global_load_lds_dword of a wave (4 bytes per lane ->
// M0 + offset + 4 * lane), the row address is wave-uniform (SGPR pair), so a row costs one instruction and no VGPR.
// A wave lands kKaDmaRows consecutive rows of the padded LDS array behind ONE M0 write (the 13-bit signed instruction offset
// reaches 4095 bytes; it also moves the global address, so the pointer is pre-biased).
template <int LDB, int J>
__device__ __forceinline__ void ka_dma_row(const char* E, int idv, int lane4) {
    const int id = __builtin_amdgcn_readlane(idv, J);       // < 0: past this wave's rows (padding rows land row 0: weight 0)
    if (id >= 0) {
        const char* p = E + ((size_t)(unsigned)id << 8) - J * LDB;
        asm volatile("global_load_lds_dword %0, %1 offset:%2" ::"v"(lane4), "s"(p), "n"(J * LDB) : "memory");
    }
}
template <int LDB, int... J>
__device__ __forceinline__ void ka_dma_rows(const char* E, int idv, int lane4, unsigned m0, std::integer_sequence<int, J...>) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0) : "memory");
    (ka_dma_row<LDB, J>(E, idv, lane4), ...);
}
bool ka_dense_dma_applies(int D, int table_bf16, int P, int Nm) {
    static const bool off = getenv("MVIN_KA_DMA") && atoi(getenv("MVIN_KA_DMA")) == 0;
    const int NmP = (Nm + 15) & ~15;
    // few rows (amazon-book: one hop of 16 memories) would be landed by one or two of the 12 waves: measured slower (3.13 -> 3.23 ms)
    return !off && D == 64 && !table_bf16 && P > 0 && P * NmP >= 64 && P * NmP <= kKaDmaRows * ka_dense_waves(64);
}

template <int D, bool BF, bool TRACE, bool DMA>
__global__ __launch_bounds__(ka_dense_waves(D) * 64, ka_dense_minw(D)) void key_addr_dense_kernel(KeyAddrGroupedArgs a, KaDenseLds L) {
    constexpr int kDW = ka_dense_waves(D);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LPR = D / 4, RPW = 64 / LPR, NT = D / 16, KS = D / 4, LDH = D + 4, LDT = D + 16, NTHR = kDW * 64;
    constexpr int LPR_L2 = (LPR == 4) ? 2 : (LPR == 8) ? 3 : (LPR == 16) ? 4 : 5;
    const int P = a.P, Nm = a.Nm, Ph = P > 0 ? P : 1, NmP = L.NmP, PN = L.PN, LDL = PN + 2;
    float* sH = smem + L.h;
    float* sU = smem + L.u;
    float* sT = smem + L.t;
    float* sEi = smem + L.ei;
    float* sL = smem + L.l;
    float* sZ = smem + L.z;
    float* sHset = smem + L.hset;
    float* sLg = smem + L.lg;
    int* sIdH = reinterpret_cast<int*>(smem + L.idh);
    int* sIdT = reinterpret_cast<int*>(smem + L.idt);
    int* sRel = reinterpret_cast<int*>(smem + L.rel);
    int* sRank = reinterpret_cast<int*>(smem + L.rank);
    int* sBidx = reinterpret_cast<int*>(smem + L.bidx);
    int* sCnt = reinterpret_cast<int*>(smem + L.cnt);
    int* sOff = reinterpret_cast<int*>(smem + L.off);
    int* sTileRel = reinterpret_cast<int*>(smem + L.tile_rel);
    int* sTileRow = reinterpret_cast<int*>(smem + L.tile_row);
    int* sOrig = reinterpret_cast<int*>(smem + L.orig);      // [kDT] original pair index (-1: padding), [kDT] = #tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // the same number, known to be wave-uniform (scalar registers)
    const int g = lane / LPR, c = lane % LPR;
    const int q16 = lane >> 4, l16 = lane & 15;
    const bool has_set = a.w != nullptr;
    // last row of E: every id that indexes the table is clamped to it.  32-bit throughout (ids < 2^31; of an int64 id only the
    // low word is read): the 64-bit spelling of the clamp cost this kernel 4 spilled registers and 18 % (1.29 -> 1.53 ms)
    const unsigned emax = (unsigned)__builtin_amdgcn_readfirstlane(a.n_entity > 0 ? a.n_entity - 1 : 0x7fffffff);
    auto item_id = [&](int o) -> unsigned {
        const unsigned v = a.items64 ? reinterpret_cast<const unsigned*>(a.items64)[2 * (int64_t)o] : (unsigned)a.items32[o];
        return min(v, emax);
    };
    const int slot0 = has_set ? 1 : 0;

    const int nseg = a.nseg_dev ? *a.nseg_dev : a.nseg;
    // The item row of a pair hangs on three dependent loads (pair_index -> items -> E row).  A tile's rows are
    // therefore fetched one tile AHEAD into registers (tile 0: before the segment's own id -> row chain starts),
    // and the next segment's descriptor is read while the current one is processed.
    // (D = 64: by the LAST kDT * LPR threads and after the tile's first barrier -- the logits tiles go to the first 8 of the 12
    // waves, so the chain's waits overlap that phase instead of standing in front of it: 1.8 k cycles per tile)
    constexpr bool LATE = D == 64;
    const int itid = LATE ? tid - (NTHR - kDT * LPR) : (tid < kDT * LPR ? tid : -1);
    auto item_row = [&](int t0, int p1, float4& e, int& orig) {
        e = make_float4(0.f, 0.f, 0.f, 0.f);
        orig = -1;
        if (itid >= 0 && t0 < p1) {
            const int p = t0 + itid / LPR;
            const int o = a.pair_index[p < p1 ? p : p1 - 1];
            const unsigned item = item_id(o);
            e = load_row4(a.E, BF, item, D, itid % LPR);
            orig = p < p1 ? o : -1;
        }
    };
    // The same chain in three stages, one per tile phase (D = 64, register-staged form): pair index under the tile's first barrier, item
    // id under the logits phase (these waves have no logits tile), row under softmax + reads -- every wait falls where the wave would
    // wait anyway: logits phase 2.5 k -> 1.6 k cycles, 1.340 -> 1.295 ms at C3, amazon-book 3.08 -> 2.97 ms.  (LDS-DMA form: measured
    // slower, 1.233 -> 1.266 ms -- the stage-2 wait also covers the head-row batch issued in front of stage 1; it keeps the one-piece chain.)
    constexpr bool STAGED = LATE && !DMA;
    int st_o = 0;
    unsigned st_item = 0;
    bool st_act = false, st_valid = false;
    auto chain_a = [&](int t0, int p1) {
        st_act = itid >= 0 && t0 < p1;
        st_valid = false;
        if (st_act) {
            const int p = t0 + itid / LPR;
            st_valid = p < p1;
            st_o = a.pair_index[st_valid ? p : p1 - 1];
        }
    };
    auto chain_b = [&]() {
        if (st_act) st_item = item_id(st_o);
    };
    auto chain_c = [&](float4& e, int& orig) {
        e = make_float4(0.f, 0.f, 0.f, 0.f);
        orig = -1;
        if (st_act) {
            e = load_row4(a.E, BF, st_item, D, itid % LPR);
            orig = st_valid ? st_o : -1;
        }
    };
    int nu = 0, np0 = 0, np1 = 0;
    if ((int)blockIdx.x < nseg) {
        nu = a.seg_user[blockIdx.x];
        np0 = a.seg_ptr[blockIdx.x];
        np1 = a.seg_ptr[blockIdx.x + 1];
    }
    // R_KGE fragments RESIDENT in registers when the model has few enough relations: (relation r, column tile nt)
    // number q = r*NT + nt belongs to wave q % (kDW-1), slot q / (kDW-1) (the last wave does the h-set read).  Every workgroup
    // of the chip needs all of R_KGE (147 KB at C3) for every user; fetched per tile those loads ran at ~6 k
    // cycles each (hot lines of a tiny table behind 4 000 waves, L1 too small to hold it next to the row traffic).
    // The contraction index is permuted (MFMA step s, slot q16 stands for k = KS*q16 + s; A and B agree) so that a
    // lane's KS fragment values of R[r][n][.] are contiguous: KS/4 16-byte loads.
    constexpr int RES = (KS <= 16) ? 4 : 0;
    const bool resident = RES > 0 && a.nR * NT <= RES * (kDW - 1) && P > 0;
    float rb[RES > 0 ? RES : 1][KS];
    auto load_bfrag = [&](int r, int nt, float (&bf)[KS]) {
        const float* Rr = a.R + (size_t)r * D * D + (size_t)(16 * nt + l16) * D + KS * q16;   // R[r][n][k]
#pragma unroll
        for (int k = 0; k < KS; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(Rr + k);
            bf[k] = v.x;
            bf[k + 1] = v.y;
            bf[k + 2] = v.z;
            bf[k + 3] = v.w;
        }
    };
    if (resident && wave < kDW - 1) {
#pragma unroll
        for (int sl = 0; sl < RES; ++sl) {
            const int q = wave_u + (kDW - 1) * sl;
            if (q < a.nR * NT) load_bfrag(q / NT, q % NT, rb[sl]);
        }
    }
    // U tile: rows sBidx[row0 .. row0+15] of sH times the fragment -> sU
    auto u_tile = [&](int row0, int nt, const float (&bf)[KS]) {
        const int ia = sBidx[row0 + l16];
        // rows are 16-byte aligned (LDH = D + 4): 16-byte reads of the A values, and the four output rows' indices in one read
        const float4* ar = reinterpret_cast<const float4*>(sH + (ia >= 0 ? ia : 0) * LDH + KS * q16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; k += 4) {
            const float4 av = ar[k >> 2];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bf[k], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bf[k + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bf[k + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bf[k + 3], acc, 0, 0, 0);
        }
        const int4 io4 = *reinterpret_cast<const int4*>(sBidx + row0 + 4 * q16);
        const int io[4] = {io4.x, io4.y, io4.z, io4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (io[i] >= 0) sU[io[i] * LDH + 16 * nt + l16] = acc[i];
        }
    };
    // ---- LDS-DMA form: ids two segments ahead in registers, one ahead in LDS; head rows one segment ahead in sH ----
    const int rows = Ph * NmP;
    int* sIdN = reinterpret_cast<int*>(smem + L.idn);        // [parity][head | tail | relation][rows]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int nu2 = 0, np02 = 0, np12 = 0;                         // descriptor of the segment after the next
    int pid_h = -1, pid_t = -1, pid_r = 0;                   // thread i < rows: ids of row i of the NEXT segment
    auto load_ids = [&](int user, int& h, int& t, int& r) {
        h = -1;
        t = -1;
        r = 0;
        if (tid < rows) {
            const int hop = tid / NmP, m = tid - hop * NmP;
            if (m < Nm) {
                const int32_t* ub = a.uts + (int64_t)user * Ph * 3 * Nm;
                // RAW here: three independent loads in flight, consumed a phase later (a clamp at this point made each of them
                // wait for its own data: three memory latencies in a row in front of the segment's first barrier, 3 k cycles)
                h = ub[(hop * 3 + 0) * Nm + m];
                t = ub[(hop * 3 + 2) * Nm + m];
                r = ub[(hop * 3 + 1) * Nm + m];
            }
        }
    };
    auto put_ids = [&](int par, int h, int t, int r) {
        if (tid < rows) {
            int* b = sIdN + par * 3 * rows;
            // device-resident ids are not validated per launch: clamped into the table (a fault would kill the process);
            // padding rows (m >= Nm) keep their -1 / -1 / 0
            const bool real = tid - (tid / NmP) * NmP < Nm;
            b[tid] = real ? (int)min((unsigned)h, emax) : -1;
            b[rows + tid] = real ? (int)min((unsigned)t, emax) : -1;
            b[2 * rows + tid] = real ? (int)min((unsigned)r, (unsigned)(a.nR - 1)) : 0;
        }
    };
    // this wave's kKaDmaRows rows of sH (LDB = row stride in bytes) or sT, ids from LDS
    auto dma_batch = [&](const int* ids, size_t word_off, auto ldb_c) {
        constexpr int LDB = decltype(ldb_c)::value;
        const int row0 = ((wave_u + kDW - 1) % kDW) * kKaDmaRows;      // wave 0 (the scan wave) takes the last, usually empty, block
        const int idv = (lane < kKaDmaRows && row0 + lane < rows) ? ids[row0 + lane] : -1;
        ka_dma_rows<LDB>(reinterpret_cast<const char*>(a.E), idv, lane * 4, lds0 + (unsigned)(word_off * 4) + (unsigned)(row0 * LDB),
                         std::make_integer_sequence<int, kKaDmaRows>{});
    };
    if constexpr (DMA) {
        for (int i = tid; i < rows * LDH; i += NTHR) sH[i] = 0.f;        // padding rows (m >= Nm) are never landed: zero for good
        for (int i = tid; i < PN * LDT; i += NTHR) sT[i] = 0.f;
        if ((int)blockIdx.x < nseg) {
            load_ids(nu, pid_h, pid_t, pid_r);
            put_ids(0, pid_h, pid_t, pid_r);
        }
        if ((int)(blockIdx.x + gridDim.x) < nseg) {
            nu2 = a.seg_user[blockIdx.x + gridDim.x];
            np02 = a.seg_ptr[blockIdx.x + gridDim.x];
            np12 = a.seg_ptr[blockIdx.x + gridDim.x + 1];
        }
        __syncthreads();
        if ((int)blockIdx.x < nseg) dma_batch(sIdN, L.h, std::integral_constant<int, LDH * 4>{});
    }
    // first tile of the first segment; every later tile -- also a later SEGMENT's first one -- is fetched under the tile before
    // it (at a segment's top the three dependent loads stood in the open: 4.4 k of its 37.5 k cycles)
    float4 e_next;
    int orig_next;
    if constexpr (LATE) item_row(np0, np1, e_next, orig_next);
    int iter = 0;
    auto stamp = [&](int slot) {
        if constexpr (TRACE) {
            if (blockIdx.x == 0 && tid == 64 * a.dbg && iter >= 4 && iter < 68) g_ka_trace[(iter - 4) * 16 + slot] = __builtin_readcyclecounter();
        }
    };
    for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x, ++iter) {
        stamp(0);
        const int u = nu, p0 = np0, p1 = np1;
        const int par = iter & 1;
        const bool has_next = seg + (int)gridDim.x < nseg;
        int* idc = sIdN + par * 3 * rows;                    // DMA form: this segment's ids
        const int* idt_cur = DMA ? idc + rows : sIdT;
        if constexpr (DMA) {
            nu = nu2;                                        // read during the previous segment's U phase (below)
            np0 = np02;
            np1 = np12;
            if (has_next) load_ids(nu, pid_h, pid_t, pid_r);           // consumed after the bucket scan
        } else if (has_next) {
            nu = a.seg_user[seg + gridDim.x];
            np0 = a.seg_ptr[seg + gridDim.x];
            np1 = a.seg_ptr[seg + gridDim.x + 1];
        }
        if constexpr (!LATE) item_row(p0, p1, e_next, orig_next);
        __syncthreads();                                     // previous segment fully consumed
        stamp(13);
        for (int i = tid; i < a.nR; i += NTHR) sCnt[i] = 0;
        for (int i = tid; i < L.maxtiles * 16; i += NTHR) sBidx[i] = -1;
        stamp(14);
        __syncthreads();
      if constexpr (DMA) {
        // ---- rank of every memory inside its relation (ids are in LDS since the previous segment) ----
        if (tid < rows) {
            int rk = 0;
            if (idc[rows + tid] >= 0) rk = atomicAdd(&sCnt[idc[2 * rows + tid]], 1);
            sRank[tid] = rk;
        }
      } else {
        // ---- ids (row i = hop * NmP + m; padding rows m >= Nm stay zero), rank of every memory inside its relation ----
        const int32_t* ub = a.uts + (int64_t)u * Ph * 3 * Nm;
        for (int i = tid; i < Ph * NmP; i += NTHR) {
            const int hop = i / NmP, m = i - hop * NmP;
            int idh = -1, idt = -1, r = 0, rk = 0;
            if (m < Nm) {
                idh = (int)min((unsigned)ub[(hop * 3 + 0) * Nm + m], emax);      // clamped into the table, like every device id
                if (hop < P) {
                    idt = (int)min((unsigned)ub[(hop * 3 + 2) * Nm + m], emax);
                    r = min((unsigned)ub[(hop * 3 + 1) * Nm + m], (unsigned)(a.nR - 1));            // indexes LDS
                    rk = atomicAdd(&sCnt[r], 1);
                }
            }
            sIdH[i] = idh;
            sIdT[i] = idt;
            sRel[i] = r;
            sRank[i] = rk;
        }
      }
        __syncthreads();
        stamp(1);
        // ---- buckets padded to whole 16-row tiles: offsets + the tile table (wave 0) ----
        if (wave == 0) {
            int base = 0, ntile = 0;
            for (int r0 = 0; r0 < a.nR; r0 += 64) {
                const int r = r0 + lane;
                const int cnt = r < a.nR ? sCnt[r] : 0;
                const int tiles = (cnt + 15) >> 4;
                int incl = tiles;                            // inclusive scan of `tiles` over the 64 lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                const int first = ntile + incl - tiles;
                if (r < a.nR) sOff[r] = first * 16;
                for (int j = 0; j < tiles; ++j) {
                    sTileRel[first + j] = r;
                    sTileRow[first + j] = (first + j) * 16;
                }
                ntile += __shfl(incl, 63, 64);
                base += 0;
            }
            if (lane == 0) sOrig[kDT] = ntile;
        }
        if constexpr (DMA) {
            // tail rows (needed by the first reads phase), landed next to wave 0's scan.  The M0 write in front of a batch waits for
            // EVERY vector memory operation the wave has in flight -- at the segment's top that was the previous tile's stores, 2.5 k
            // cycles; here they and the id loads of the top are done.  This segment's head rows landed a segment ago.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dma_batch(idc + rows, L.t, std::integral_constant<int, LDT * 4>{});
        }
        __syncthreads();
        const int ntile = sOrig[kDT];
        stamp(2);
      if constexpr (DMA) {
        // ---- bucket index table; the next segment's ids -> LDS; this segment's rows have landed ----
        if (tid < rows && idc[rows + tid] >= 0) sBidx[sOff[idc[2 * rows + tid]] + sRank[tid]] = tid;
        if (has_next) put_ids(par ^ 1, pid_h, pid_t, pid_r);
      } else {
        // ---- rows -> LDS; bucket index table ----
        for (int i = wave * RPW + g; i < Ph * NmP; i += kDW * RPW) {
            const int idh = sIdH[i], idt = sIdT[i];
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f), t = h;
            if (idh >= 0) h = load_row4(a.E, BF, idh, D, c);
            if (idt >= 0) t = load_row4(a.E, BF, idt, D, c);
            float* dh = sH + (size_t)i * LDH + 4 * c;
            *reinterpret_cast<float2*>(dh) = make_float2(h.x, h.y);
            *reinterpret_cast<float2*>(dh + 2) = make_float2(h.z, h.w);
            if (i < PN) {
                *reinterpret_cast<float4*>(sT + (size_t)i * LDT + 4 * c) = t;
                if (c == 0 && idt >= 0) sBidx[sOff[sRel[i]] + sRank[i]] = i;
            }
        }
      }
        __syncthreads();
        stamp(3);
        // the descriptor of the segment after the next: scalar loads, which EVERY barrier waits for (lgkmcnt) -- issued here, at the
        // start of the longest phase, they cost nothing; at the segment's top they stood in front of its first barrier (~1 k cycles).
        // (Register-staged form: measured slower with the move, 1.336 -> 1.374 ms; it keeps its load at the top.)
        // (as VECTOR loads through a lane-dependent zero, turned into scalars after the U tiles: left to itself the compiler reads
        // each with a vector load + readfirstlane + its own vmcnt(0) -- the kernel's stores make them "clobberable", so no s_load --
        // three memory latencies in a row in front of every wave's U tiles)
        int d_u = 0, d_p0 = 0, d_p1 = 0;
        const bool d_have = DMA && seg + 2 * (int)gridDim.x < nseg;
        if constexpr (DMA) {
            if (d_have) {
                int zv = 0;
                asm volatile("" : "+v"(zv));
                const int i = seg + 2 * (int)gridDim.x + zv;
                d_u = a.seg_user[i];
                d_p0 = a.seg_ptr[i];
                d_p1 = a.seg_ptr[i + 1];
            }
        }
        // ---- h-set read (wave 15) next to the U tiles (waves 0..14) ----
        if (has_set && wave == kDW - 1) {
            // ONE pass over the head rows of hop 0 (online softmax per lane group, the RPW groups merged at the end): the
            // three-pass form (logits -> LDS, softmax, weighted sum) made this wave the last of the phase by ~1.8 k cycles
            const float4 w4 = reinterpret_cast<const float4*>(a.w)[c];
            float mx = -INFINITY, z = 0.f;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int m0 = 0; m0 < NmP; m0 += RPW) {
                const int m = m0 + g;
                const float4 h4 = *reinterpret_cast<const float4*>(sH + (size_t)m * LDH + 4 * c);
                float d = fmaf(h4.x, w4.x, fmaf(h4.y, w4.y, fmaf(h4.z, w4.z, h4.w * w4.w)));
                d = group_sum(d, LPR_L2);
                if (m < Nm) {                            // (the padding rows m >= Nm take no part)
                    const float nm = fmaxf(mx, d);
                    const float sc = mx == -INFINITY ? 0.f : kad_exp(mx - nm), e = kad_exp(d - nm);   // (kad_exp has no range selects)
                    z = fmaf(z, sc, e);
                    acc = make_float4(fmaf(acc.x, sc, e * h4.x), fmaf(acc.y, sc, e * h4.y), fmaf(acc.z, sc, e * h4.z),
                                      fmaf(acc.w, sc, e * h4.w));
                    mx = nm;
                }
            }
            // merge the lane groups (lanes l, l ^ LPR, l ^ 2 LPR, ...: the same column chunk c of different row groups)
            float M = mx;
            if (LPR <= 16) M = xor16_max(M);
            if (LPR <= 32) M = xor32_max(M);
            if (LPR < 16) {
                for (int o = LPR; o < 16; o <<= 1) M = fmaxf(M, __shfl_xor(M, o, kWave));
            }
            const float f = mx == -INFINITY ? 0.f : kad_exp(mx - M);      // a group that saw no row: factor 0
            z *= f;
            acc = make_float4(acc.x * f, acc.y * f, acc.z * f, acc.w * f);
            acc = group_xor_sum(acc, LPR);
            z = group_xor_sum(make_float4(z, 0.f, 0.f, 0.f), LPR).x;
            const float inv = 1.f / z;
            if (g == 0) *reinterpret_cast<float4*>(sHset + 4 * c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        } else if (P > 0) {
            const int nw = has_set ? kDW - 1 : kDW;
            if (resident) {
                if (wave < kDW - 1) {
                    // wave-uniform task table first (scalar: with per-lane copies of these the kernel spilled the LDS addresses
                    // and reloaded them from scratch in front of every slot's tiles)
                    int tl[RES > 0 ? RES : 1], rw[RES > 0 ? RES : 1];
#pragma unroll
                    for (int sl = 0; sl < RES; ++sl) {
                        const int q = wave_u + (kDW - 1) * sl;
                        const bool ok = q < a.nR * NT;
                        const int r = ok ? q / NT : 0;
                        tl[sl] = ok ? (sCnt[r] + 15) >> 4 : 0;
                        rw[sl] = sOff[r];
                    }
#pragma unroll
                    for (int sl = 0; sl < RES; ++sl) {
                        const int nt = (wave_u + (kDW - 1) * sl) % NT;
                        const int tiles = __builtin_amdgcn_readfirstlane(tl[sl]), row0 = __builtin_amdgcn_readfirstlane(rw[sl]);
                        for (int j = 0; j < tiles; ++j) u_tile(row0 + 16 * j, nt, rb[sl]);
                    }
                }
            } else {
                for (int task = wave; task < ntile * NT; task += nw) {
                    const int tl = task / NT, nt = task - tl * NT;
                    float bfrag[KS];
                    load_bfrag(sTileRel[tl], nt, bfrag);
                    u_tile(sTileRow[tl], nt, bfrag);
                }
            }
            stamp(9);
            // padding memories: U rows never written by a tile must read as zero
            for (int i = tid; i < PN * NT; i += (has_set ? (kDW - 1) : kDW) * 64) {
                const int row = i / NT, nt = i - row * NT;
                if (idt_cur[row] < 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) sU[(size_t)row * LDH + 16 * nt + j] = 0.f;
                }
            }
        }
        if constexpr (DMA) {
            if (d_have) {
                nu2 = __builtin_amdgcn_readfirstlane(d_u);
                np02 = __builtin_amdgcn_readfirstlane(d_p0);
                np12 = __builtin_amdgcn_readfirstlane(d_p1);
            }
        }
        stamp(4);
        // ---- the user's pairs, 16 at a time ----
        for (int t0 = p0; t0 < p1; t0 += kDT) {
            __syncthreads();                                 // sU / sHset complete; previous tile consumed
            if (itid >= 0) {
                const int i = itid / LPR, cc = itid % LPR;
                float* dst = sEi + i * LDH + 4 * cc;
                *reinterpret_cast<float2*>(dst) = make_float2(e_next.x, e_next.y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(e_next.z, e_next.w);
                if (cc == 0) sOrig[i] = orig_next;
            }
            if (t0 == p0) stamp(10);
            if constexpr (DMA) {                             // sH is free from here on: the next segment's head rows
                if (t0 == p0 && has_next) dma_batch(sIdN + (par ^ 1) * 3 * rows, L.h, std::integral_constant<int, LDH * 4>{});
            }
            if (t0 == p0) stamp(11);
            const bool last = t0 + kDT >= p1;                // np0 / np1: the next segment's (no next one: a harmless re-read)
            if constexpr (!LATE) item_row(t0 + kDT, p1, e_next, orig_next);
            if constexpr (STAGED) chain_a(last ? np0 : t0 + kDT, last ? np1 : p1);
            __syncthreads();
            if (t0 == p0) stamp(5);
            if constexpr (LATE && !STAGED) item_row(last ? np0 : t0 + kDT, last ? np1 : p1, e_next, orig_next);
            if constexpr (STAGED) chain_b();
            // logits L[pair, m] = E[item_pair] . U_m : one 16-memory tile per task
            for (int mt = wave; mt < PN / 16; mt += kDW) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // contraction index permuted as in u_tile (step k of slot q16 stands for KS * q16 + k on both operands): a lane's
                // values are contiguous, four steps per 16-byte read (with one 4-byte read per step and operand the loop was a
                // chain of eight LDS latencies: 1.55 k cycles for 16 MFMAs)
                const float4* ar = reinterpret_cast<const float4*>(sEi + l16 * LDH + KS * q16);
                const float4* br = reinterpret_cast<const float4*>(sU + (16 * mt + l16) * LDH + KS * q16);
#pragma unroll
                for (int k = 0; k < KS / 4; ++k) {
                    const float4 av = ar[k], bv = br[k];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) sL[(size_t)(4 * q16 + i) * LDL + 16 * mt + l16] = acc[i];
            }
            __syncthreads();
            if (t0 == p0) stamp(6);
            if constexpr (STAGED) chain_c(e_next, orig_next);
            // softmax over the Nm memories of every (pair, hop) (:223): un-normalised weights back to sL, 1/sum to sZ
            // FOUR rows per wave pass, a 16-lane DPP row per (pair, hop) and NmP/16 values per lane: with a whole wave
            // per row the phase was VALU-issue-bound on two 64-wide reductions per 64 values (3.7 k cycles per tile,
            // three rounds over the 12 waves; interleaving three rows per wave changed nothing)
            {
                const int rg = lane >> 4, cl = lane & 15;
                for (int base = wave * 4; base < kDT * P; base += kDW * 4) {
                    const int task = base + rg;
                    const bool ok = task < kDT * P;
                    const int tk = ok ? task : base;
                    const int pi = tk / P, hop = tk - pi * P;
                    float* row = sL + (size_t)pi * LDL + hop * NmP;
                    float mx = -INFINITY, z = 0.f;
                    if (NmP <= 64) {
                        // at most four values per lane: read ONCE, together (the two run-time loops below were chains of LDS
                        // latencies: 2.6 k cycles per tile for 64 values per row)
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = (cl + 16 * u < Nm) ? row[cl + 16 * u] : -INFINITY;
                        mx = group_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), 4);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float e = (cl + 16 * u < Nm) ? kad_exp(v[u] - mx) : 0.f;
                            if (ok && cl + 16 * u < NmP) row[cl + 16 * u] = e;
                            z += e;
                        }
                    } else {
                        for (int m = cl; m < Nm; m += 16) mx = fmaxf(mx, row[m]);
                        mx = group_max(mx, 4);
                        for (int m = cl; m < NmP; m += 16) {
                            const float e = m < Nm ? kad_exp(row[m] - mx) : 0.f;
                            if (ok) row[m] = e;
                            z += e;
                        }
                    }
                    z = group_sum(z, 4);
                    if (ok && cl == 0) sZ[task] = 1.f / z;      // sZ[pi * P + hop]
                }
            }
            if constexpr (DMA) {
                if (t0 == p0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tail rows have landed
            }
            __syncthreads();
            if (t0 == p0) stamp(7);
            // reads o[pair, hop, :] = sum_m p[pair, m] t_m : one (hop, 16-column tile) per task
            for (int task = wave; task < P * NT; task += kDW) {
                const int hop = task / NT, nt = task - hop * NT;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* ar = sL + (size_t)l16 * LDL + hop * NmP + q16;
                const float* br = sT + (size_t)(hop * NmP + q16) * LDT + 16 * nt + l16;
                // NmP is a multiple of 16: four steps per round, their eight LDS reads issued before the first MFMA (one
                // step at a time the loop was a chain of LDS latencies: 2.7 k cycles for 16 MFMAs)
                for (int k = 0; k < NmP / 4; k += 4) {
                    float av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        av[u] = ar[4 * (k + u)];
                        bv[u] = br[(size_t)4 * (k + u) * LDT];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pi = 4 * q16 + i;
                    const int orig = sOrig[pi];
                    if (orig >= 0)
                        a.out[(int64_t)orig * a.ldo + (size_t)(slot0 + hop) * D + 16 * nt + l16] = acc[i] * sZ[pi * P + hop];
                }
            }
            if (t0 == p0) stamp(8);
            if (has_set) {
                for (int i = tid; i < kDT * LPR; i += NTHR) {
                    const int pi = i / LPR, cc = i - pi * LPR;
                    const int orig = sOrig[pi];
                    if (orig >= 0)
                        *reinterpret_cast<float4*>(a.out + (int64_t)orig * a.ldo + 4 * cc) = *reinterpret_cast<const float4*>(sHset + 4 * cc);
                }
            }
        }
        if (LATE && p0 >= p1) item_row(np0, np1, e_next, orig_next);         // a segment without pairs
        if constexpr (DMA) {
            if (p0 >= p1 && has_next) {                      // a segment without pairs still hands sH on
                __syncthreads();
                dma_batch(sIdN + (par ^ 1) * 3 * rows, L.h, std::integral_constant<int, LDH * 4>{});
            }
        }
    }
}

bool key_addr_dense_supported(int D, int P, int Nm, int nR) {
    const bool dok = D == 16 || D == 32 || D == 64 || D == 128;
    return dok && Nm >= 1 && Nm <= 256 && P >= 0 && P <= 8 && nR >= 1 && nR <= 4096 &&
           ka_dense_layout(D, P, Nm, nR).total <= 160 * 1024;
}

template <int D>
static hipError_t launch_kad(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st) {
    const KaDenseLds L = ka_dense_layout(D, a.P, a.Nm, a.nR);
    constexpr int kDW = ka_dense_waves(D);
    hipError_t e = hipSuccess;
    // persistent grid: as many workgroups as the CUs hold (registers, LDS and wave slots decide: 1 per CU at D >= 32)
    auto grid_for = [&](const void* k) {
        static thread_local const void* last_k = nullptr;      // the query is host-side arithmetic, but not free
        static thread_local size_t last_lds = 0;
        static thread_local int last_per_cu = 1;
        if (k != last_k || L.total != last_lds) {
            int per_cu = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kDW * 64, L.total) != hipSuccess || per_cu < 1) per_cu = 1;
            last_k = k;
            last_lds = L.total;
            last_per_cu = per_cu;
        }
        const int cap = 256 * last_per_cu;
        return a.nseg < cap ? a.nseg : cap;
    };
    if (table_bf16) {
        auto k = key_addr_dense_kernel<D, true, false, false>;
        if (L.total > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
        if (e != hipSuccess) return e;
        k<<<grid_for(reinterpret_cast<const void*>(k)), kDW * 64, L.total, st>>>(a, L);
    } else {
        static const bool trace = getenv("MVIN_KA_TRACE") != nullptr;
        const bool dma = D == 64 && ka_dense_dma_applies(D, 0, a.P, a.Nm);
        auto k = dma ? ((D == 64 && trace) ? key_addr_dense_kernel<D, false, D == 64, D == 64> : key_addr_dense_kernel<D, false, false, D == 64>)
                     : ((D == 64 && trace) ? key_addr_dense_kernel<D, false, true, false> : key_addr_dense_kernel<D, false, false, false>);
        if (L.total > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
        if (e != hipSuccess) return e;
        KeyAddrGroupedArgs b = a;
        b.dbg = 0;
        if (trace) {                                         // which wave of workgroup 0 stamps (scripts/trace_keyaddr.py)
            const char* tw = getenv("MVIN_KA_TRACE_WAVE");
            b.dbg = tw ? atoi(tw) % kDW : 0;
        }
        k<<<grid_for(reinterpret_cast<const void*>(k)), kDW * 64, L.total, st>>>(b, L);
    }
    return hipGetLastError();
}

hipError_t ka_read_trace(long long* host_dst, size_t n) {
    const size_t have = sizeof(g_ka_trace) / sizeof(long long);
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_ka_trace), (n < have ? n : have) * sizeof(long long));
}

hipError_t launch_key_addr_dense(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st) {
    switch (a.D) {
        case 16: return launch_kad<16>(a, table_bf16, st);
        case 32: return launch_kad<32>(a, table_bf16, st);
        case 64: return launch_kad<64>(a, table_bf16, st);
        case 128: return launch_kad<128>(a, table_bf16, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
