// MVIN._key_addressing (model.py:161-240) for pairs grouped by user at dim 16 -- the dimension of every run script the
// reference ships (src/bash/mvin_*.sh) -- and dim 32 (BASELINE config C2), with ONE WAVE per user segment.  Same interface
// and arithmetic as key_addr_dense_kernel (mvin_keyaddr_dense.hip); the mapping is the opposite.  At these dimensions a
// user's 2*P*Nm ripple-set rows are 16-32 KB and every product is a handful of 16x16x4 MFMA steps, so the workgroup-wide
// phases of the dense kernel (stage -> barrier -> U -> barrier -> logits -> barrier -> softmax -> barrier -> reads, 48 k
// cycles per segment with four segments in flight per CU at D = 16) are mostly barrier and latency.  Here nothing is
// shared between waves but a read-only LDS copy of R_KGE, and there is no workgroup barrier after the prologue:
//   * U_m = R_KGE[r_m] . h_m (model.py:214-216) and the tail rows t_m live in REGISTERS, already in MFMA B-fragment
//     layout: lane (q, j) = (lane / 16, lane % 16) holds U[m = 16t + j][n = 4s + q] (t < NT, s < D/4) and
//     T[m = 4s + q][n = 16c + j] (s < 4 NT, c < D/16) -- 32 + 32 VGPRs per hop of 64 memories at D = 32, half of that at
//     D = 16, where both hops stay resident; at D = 32 the wave walks its pairs once per hop.  U is computed in that
//     layout on the VALU (v_pk_fma_f32) from head rows staged through LDS and R_KGE rows read from LDS (relation stride
//     D*D + 4 words: lanes of different relations land on different banks, lanes of the same relation read one address);
//   * per tile of 16 pairs: logits L = E[items] . U^T (:219-220) as NT * D/4 MFMA steps per hop -> accumulators hold
//     L[pair 4q + r][m = 16t + j]; softmax over the memories (:223) = in-lane over t, then a 16-lane DPP row reduction;
//     the un-normalised weights cross from accumulator layout to A-fragment layout through a 16 x 68-word LDS tile
//     private to the wave (conflict-free both ways); reads o = P . T (:229) as 4 NT * D/16 MFMA steps per hop; 64-byte
//     row pieces go out straight from the accumulators;
//   * the h-set read (:162-197) once per user: lane groups hold the head rows of hop 0, 16-lane DPP reductions.
// 16 waves per CU at D = 16 (8 at D = 32), each on its own user, hide each other's dependent loads.
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kKwLdP = 68;                   // row stride (words) of the wave's 16 x 64 weight tile
constexpr int kKwIds = 2 * 3 * 64;           // the user's ids: [hop][h | r | t][64], padded with -1

template <int D>
struct KwCfg {
    static constexpr int KS = D / 4;         // MFMA steps of the logits (contraction over the D components)
    static constexpr int NC = D / 16;        // 16-column tiles of an output row
    static constexpr int CH = D / 4;         // 4-element chunks of a row
    static constexpr int LdR = D * D + 4;    // words per relation matrix in LDS
    static constexpr int LdH = D + 4;        // row stride (words) of a staged 16-row head tile: 16-byte reads of 16 rows hit 16 bank groups
    static constexpr int Waves = D == 16 ? 4 : 8;      // waves per workgroup (they only share the LDS copy of R_KGE)
    static constexpr int MinW = D == 16 ? 4 : 2;       // waves per SIMD the register budget is cut for
    // staged head tiles: two at D = 32; ONE at D = 16, where the second one cost the fourth workgroup per CU (44 KB of LDS per
    // workgroup -> 3 per CU; 39 KB -> 4 = 16 waves) -- the LDS queue of a wave is in order, so a single tile only needs the
    // compiler fences it already has
    static constexpr int NBuf = D == 16 ? 1 : 2;
    static constexpr int PerWave = 16 * kKwLdP + kKwIds + D + 16 + NBuf * 16 * LdH;   // + h-set read + pair indices + head tiles
};

// exp(x) for the softmax arguments (x = logit - max <= 0, or discarded by a select): the argument reduction of the
// library routine -- x * log2(e) split into an integer and a fraction with the product's rounding error folded back in by
// fma -- without its overflow / underflow selects (ldexp flushes to 0 on its own): 7 instructions instead of 12
__device__ __forceinline__ float kw_exp(float x) {
    const float t = x * 1.44269502162933349609375f;              // float(log2 e)
    const float n = rintf(t);
    float f = fmaf(x, 1.44269502162933349609375f, -n);
    f = fmaf(x, 1.925963033500011e-8f, f);                       // log2 e - float(log2 e)
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of ONE wave execute in order; this only stops the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Table rows are addressed as  base (SGPR pair) + 32-bit byte offset (one VGPR): tables up to 4 GB (the launcher checks), and
// no 64-bit per-lane pointer arithmetic -- it was a third of the per-user instructions.
template <bool BF, int D>
__device__ __forceinline__ float kw_elem(const void* E, unsigned row, int n) {
    const char* b = reinterpret_cast<const char*>(E);
    if (BF) return __uint_as_float((unsigned)*reinterpret_cast<const uint16_t*>(b + (row * (2u * D) + 2u * n)) << 16);
    return *reinterpret_cast<const float*>(b + (row * (4u * D) + 4u * n));
}

// elements 4c .. 4c+3 of a row
template <bool BF, int D>
__device__ __forceinline__ float4 kw_chunk(const void* E, unsigned row, int c) {
    const char* b = reinterpret_cast<const char*>(E);
    if (BF) {
        const uint2 v = *reinterpret_cast<const uint2*>(b + (row * (2u * D) + 8u * c));
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(b + (row * (4u * D) + 16u * c));
}

// P (1 or 2 hops), the presence of the h-set read, NT (memory tiles of 16 per hop: 1 for n_memory <= 16 -- amazon-book's
// shipped setting --, else 4) and FULL (n_memory == 16 * NT: no padding memories to mask) are compile-time: as run-time
// branches they cut the per-user section into dozens of basic blocks that hipcc could neither schedule nor allocate
// (300 spills).
template <int D, bool BF, int P, bool HAS_SET, int NT, bool FULL>
__global__ __launch_bounds__(KwCfg<D>::Waves * 64, KwCfg<D>::MinW) void key_addr_wave_kernel(KeyAddrGroupedArgs a) {
    using C = KwCfg<D>;
    constexpr int KS = C::KS, NC = C::NC, CH = C::CH, LdR = C::LdR, LdH = C::LdH, NWV = C::Waves;
    constexpr int PR = D == 16 ? P : 1;                          // hops whose fragments are resident at a time
    constexpr int CPL = CH / 4;                                  // head-row chunks a lane stages per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int Nm = a.Nm;
    float* sR = smem;                                            // [nR][LdR]
    float* sW = smem + (size_t)a.nR * LdR + (size_t)wave * C::PerWave;
    float* sP = sW;                                              // [16][68]
    int* sIds = reinterpret_cast<int*>(sW + 16 * kKwLdP);        // [2][3][64]
    float* sHset = sW + 16 * kKwLdP + kKwIds;                    // [D]
    int* sOrig = reinterpret_cast<int*>(sHset + D);              // [16]
    float* sHt = sHset + D + 16;                                 // [NBuf][16][LdH]
    for (int i = tid; i < a.nR * D * D; i += NWV * 64) sR[(i / (D * D)) * LdR + (i % (D * D))] = a.R[i];
    __syncthreads();

    constexpr int slot0 = HAS_SET ? 1 : 0;
    const unsigned max_id = (unsigned)(a.n_entity - 1);
    const int nseg = a.nseg_dev ? *a.nseg_dev : a.nseg;
    const int nw = gridDim.x * NWV;
    for (int seg = blockIdx.x * NWV + wave; seg < nseg; seg += nw) {
        const int u = __builtin_amdgcn_readfirstlane(a.seg_user[seg]);
        const int p0 = __builtin_amdgcn_readfirstlane(a.seg_ptr[seg]);
        const int p1 = __builtin_amdgcn_readfirstlane(a.seg_ptr[seg + 1]);
        // a tile's item rows hang on three dependent loads (pair_index -> items -> E row): they are started a tile ahead
        auto item_frag = [&](int t0, float (&av)[KS], int& orig) {
            const int p = t0 + j;
            const int o = a.pair_index[p < p1 ? p : p1 - 1];
            const int64_t item = a.items64 ? a.items64[o] : (int64_t)a.items32[o];
            const unsigned row = min((unsigned)item, max_id);
#pragma unroll
            for (int s = 0; s < KS; ++s) av[s] = kw_elem<BF, D>(a.E, row, 4 * s + q);
            orig = p < p1 ? o : -1;
        };
        float av[KS];
        int orig;
        item_frag(p0, av, orig);
        // ---- the user's ids -> LDS ([hop][h | r | t][64], -1 beyond Nm) ----
        wave_lds_sync();                                         // the previous segment's reads of sIds / sP are done
        const int32_t* ub = a.uts + (int64_t)u * P * 3 * Nm;
        for (int i = lane; i < P * 3 * 64; i += 64) {
            const int hx = i >> 6, m = i & 63;
            const int v = m < Nm ? ub[hx * Nm + m] : -1;          // head / tail ids clamped into the table, like every device id
            sIds[i] = (m < Nm && hx % 3 != 1) ? (int)min((unsigned)v, max_id) : v;
        }
        wave_lds_sync();
        // Head rows reach the lanes through LDS: a tile of 16 rows (memories m = 16t + j of one hop) is ONE coalesced
        // wave-load per 64 bytes of row -- lane (q, j) fetches chunks q, q + 4, .. of row j -- written to a double-buffered
        // 16 x (D + 4)-word tile, from which every lane reads the chunks of ITS row (the four q lanes the same address).
        // Loading the rows straight into the lanes that need them took four times the wave-loads, each fetching every row
        // four times over: the texture addresser was busy 74 % of the kernel (rocprofv3 TA_TA_BUSY).  The load of tile
        // i + 1 is in flight while tile i is used.
        int stage_i = 0;                                         // tiles staged so far (buffer = parity)
        auto head_chunks = [&](int hop, int t, float4 (&v)[CPL]) {
            const int idh = sIds[(hop * 3 + 0) * 64 + 16 * t + j];
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc)                      // padding memories read row 0 (finite; masked by position)
                v[cc] = kw_chunk<BF, D>(a.E, idh >= 0 ? idh : 0, q + 4 * cc);
        };
        auto stage = [&](const float4 (&v)[CPL]) {
            float* dst = sHt + (stage_i & (C::NBuf - 1)) * 16 * LdH;
            wave_lds_sync();                                     // the reads of this buffer two tiles ago are done
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) *reinterpret_cast<float4*>(dst + j * LdH + 4 * (q + 4 * cc)) = v[cc];
            wave_lds_sync();
            ++stage_i;
            return dst + j * LdH;                                // this lane's row
        };
        if (HAS_SET) {
            // o_hset = sum_m softmax_m(h0_m . w) h0_m (:162-197), BEFORE the fragments occupy their registers: this lane's
            // rows are m = 16t + j (the four q rows hold copies); two passes over the NT tiles of hop 0
            float lg[NT];
            float4 cur[CPL], nxt[CPL];
            head_chunks(0, 0, cur);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + 1 < NT) head_chunks(0, t + 1, nxt);
                const float* hr = stage(cur);
                float dl = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * c);
                    dl = fmaf(hv.x, a.w[4 * c], fmaf(hv.y, a.w[4 * c + 1], fmaf(hv.z, a.w[4 * c + 2], fmaf(hv.w, a.w[4 * c + 3], dl))));
                }
                lg[t] = (16 * t + j) < Nm ? dl : -INFINITY;
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) cur[cc] = nxt[cc];
            }
            float mx = lg[0];
#pragma unroll
            for (int t = 1; t < NT; ++t) mx = fmaxf(mx, lg[t]);
            mx = group_max(mx, 4);
            float e[NT], z = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                e[t] = (16 * t + j) < Nm ? kw_exp(lg[t] - mx) : 0.f;
                z += e[t];
            }
            z = group_sum(z, 4);
            float part[D];
#pragma unroll
            for (int k = 0; k < D; ++k) part[k] = 0.f;
            head_chunks(0, 0, cur);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + 1 < NT) head_chunks(0, t + 1, nxt);
                const float* hr = stage(cur);                    // second pass (e = 0 on padding)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * c);
                    part[4 * c] = fmaf(e[t], hv.x, part[4 * c]);
                    part[4 * c + 1] = fmaf(e[t], hv.y, part[4 * c + 1]);
                    part[4 * c + 2] = fmaf(e[t], hv.z, part[4 * c + 2]);
                    part[4 * c + 3] = fmaf(e[t], hv.w, part[4 * c + 3]);
                }
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) cur[cc] = nxt[cc];
            }
            const float inv = 1.f / z;
            float mine[NC];                                      // lane j keeps sums j, 16 + j, .. (selects, no divergent branches)
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2) mine[c2] = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float sum = group_sum(part[k], 4);
                mine[k / 16] = j == (k % 16) ? sum : mine[k / 16];
            }
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2) sHset[16 * c2 + j] = mine[c2] * inv;     // the four q rows hold copies: same value, same address
        }
        // ---- hop group by hop group (D = 16: both hops at once; D = 32: one hop, the pairs walked once per hop) ----
#pragma unroll
        for (int h0 = 0; h0 < P; h0 += PR) {
            if (h0 > 0) item_frag(p0, av, orig);
            // ---- U and T fragments -> registers ----
            float Bl[PR][NT][KS], Tf[PR][4 * NT][NC];
            {
                float4 cur[CPL], nxt[CPL];
                head_chunks(h0, 0, cur);
#pragma unroll
                for (int i = 0; i < NT * PR; ++i) {
                    if (i + 1 < NT * PR) head_chunks(h0 + (i + 1) / NT, (i + 1) % NT, nxt);
                    const float* hr = stage(cur);
                    const int hl = i / NT, t = i % NT;
                    int r = min((unsigned)sIds[((h0 + hl) * 3 + 1) * 64 + 16 * t + j], (unsigned)(a.nR - 1));
                    if (a.dbg & 1) r = 0;                        // MVIN_KA_WAVE_DBG bit 0 (measurement only): every lane reads R_KGE[0]
                    const float* Rr = sR + (size_t)r * LdR + q * D;
                    f32x2 d[KS];                                 // even / odd k apart: v_pk_fma_f32
#pragma unroll
                    for (int s = 0; s < KS; ++s) d[s] = f32x2{0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * c);
                        const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            const float4 v = *reinterpret_cast<const float4*>(Rr + 4 * s * D + 4 * c);   // R[r][n = 4s + q][4c ..]
                            d[s] = __builtin_elementwise_fma(f32x2{v.x, v.y}, h01, d[s]);
                            d[s] = __builtin_elementwise_fma(f32x2{v.z, v.w}, h23, d[s]);
                        }
                        // KS + 1 LDS reads at a time: left alone hipcc issues all of a row's up front and spills
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < KS; ++s) Bl[hl][t][s] = d[s].x + d[s].y;     // U[m = 16t + j][n = 4s + q]
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) cur[cc] = nxt[cc];
                }
            }
#pragma unroll
            for (int hl = 0; hl < PR; ++hl) {
#pragma unroll
                for (int s = 0; s < 4 * NT; ++s) {
                    const int idt = sIds[((h0 + hl) * 3 + 2) * 64 + 4 * s + q];
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2)
                        Tf[hl][s][c2] = kw_elem<BF, D>(a.E, idt >= 0 ? idt : 0, 16 * c2 + j);           // T[m = 4s + q][n = 16c + j]
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // a do-while: segments are never empty, and with a guarded loop LLVM sinks the whole fragment arithmetic above
            // into the guarded block, away from its loads -- every loaded value then lives across the branch (580 spills)
            int t0 = p0;
            do {
                float avn[KS];
                int orig_n;
                item_frag(t0 + 16 < p1 ? t0 + 16 : t0, avn, orig_n);     // the next tile's rows land under this tile's work
                wave_lds_sync();                                 // previous tile's reads of sOrig / sHset writes visible
                if (q == 0) sOrig[j] = orig;
                wave_lds_sync();
                int og[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) og[r] = sOrig[4 * q + r];
#pragma unroll
                for (int hl = 0; hl < PR; ++hl) {
                    f32x4 acc[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < KS; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Bl[hl][t][s], acc[t], 0, 0, 0);
                    }
                    // softmax over the memories of pair 4q + r (:223): in-lane over t, then across the 16 lanes of the row
                    float zinv[4];
                    wave_lds_sync();                             // the previous hop's A reads of sP are done
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mx = -INFINITY;
#pragma unroll
                        for (int t = 0; t < NT; ++t) mx = fmaxf(mx, (FULL || (16 * t + j) < Nm) ? acc[t][r] : -INFINITY);
                        mx = group_max(mx, 4);
                        float z = 0.f;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float e = (FULL || (16 * t + j) < Nm) ? kw_exp(acc[t][r] - mx) : 0.f;
                            sP[(4 * q + r) * kKwLdP + 16 * t + j] = e;
                            z += e;
                        }
                        zinv[r] = 1.f / group_sum(z, 4);
                    }
                    wave_lds_sync();
                    // reads o = P . T (:229): A[i = j][k = q] = P[pair j][m = 4s + q]
                    f32x4 o[NC];
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2) o[c2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float* ap = sP + j * kKwLdP + q;
#pragma unroll
                    for (int s = 0; s < 4 * NT; ++s) {
                        const float pa = ap[4 * s];
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) o[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, Tf[hl][s][c2], o[c2], 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (og[r] >= 0) {
#pragma unroll
                            for (int c2 = 0; c2 < NC; ++c2)
                                a.out[(int64_t)og[r] * a.ldo + (size_t)(slot0 + h0 + hl) * D + 16 * c2 + j] = o[c2][r] * zinv[r];
                        }
                }
                if (HAS_SET && h0 == 0) {
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2) {
                        const float hs = sHset[16 * c2 + j];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (og[r] >= 0) a.out[(int64_t)og[r] * a.ldo + 16 * c2 + j] = hs;
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) av[s] = avn[s];
                orig = orig_n;
                t0 += 16;
            } while (t0 < p1);
        }
    }
}

template <int D>
static size_t kw_lds_bytes(int nR) { return ((size_t)nR * KwCfg<D>::LdR + (size_t)KwCfg<D>::Waves * KwCfg<D>::PerWave) * sizeof(float); }

bool key_addr_wave16_supported(int D, int P, int Nm, int nR) {
    static const bool off = (getenv("MVIN_KA_WAVE16") != nullptr && getenv("MVIN_KA_WAVE16")[0] == '0') ||
                            (getenv("MVIN_KA_WAVE") != nullptr && getenv("MVIN_KA_WAVE")[0] == '0');
    static const bool off32 = getenv("MVIN_KA_WAVE32") != nullptr && getenv("MVIN_KA_WAVE32")[0] == '0';
    if (off || !(P == 1 || P == 2) || Nm < 1 || Nm > 64 || nR < 1) return false;
    if (D == 16) return kw_lds_bytes<16>(nR) <= 80 * 1024;       // two workgroups of four waves per CU at least
    if (D == 32) return !off32 && kw_lds_bytes<32>(nR) <= 156 * 1024;                 // one workgroup of eight waves per CU
    return false;
}

// rows are addressed by 32-bit byte offsets
bool key_addr_wave16_applies(const KeyAddrGroupedArgs& a) {
    return key_addr_wave16_supported(a.D, a.P, a.Nm, a.nR) && a.n_entity > 0 && (uint64_t)a.n_entity * 4u * a.D < (1ull << 32);
}

template <int D>
static hipError_t launch_kw(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st) {
    constexpr int NWV = KwCfg<D>::Waves;
    const size_t lds = kw_lds_bytes<D>(a.nR);
    hipError_t err = hipSuccess;
    auto launch = [&](auto kernel) {
        // persistent grid: as many workgroups as the CUs hold (LDS decides: D = 16: 39 KB at nR = 9 -> 4 per CU = 16 waves,
        // 75 KB at nR = 39 -> 2; D = 32: one workgroup of 8 waves)
        static thread_local const void* last_k = nullptr;
        static thread_local size_t last_lds = 0;
        static thread_local int last_per_cu = 1;
        const void* k = reinterpret_cast<const void*>(kernel);
        if (lds > 64 * 1024) {
            err = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (err != hipSuccess) return;
        }
        if (k != last_k || lds != last_lds) {
            int per_cu = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NWV * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            last_k = k;
            last_lds = lds;
            last_per_cu = per_cu;
        }
        const int need = (a.nseg + NWV - 1) / NWV;
        static const int dbg = getenv("MVIN_KA_WAVE_DBG") ? atoi(getenv("MVIN_KA_WAVE_DBG")) : 0;      // measurement knobs
        KeyAddrGroupedArgs b = a;
        b.dbg = dbg;
        const int per_cu = (dbg & 4) ? 1 : (dbg & 2) ? 2 : last_per_cu;
        const int cap = 256 * (per_cu < last_per_cu ? per_cu : last_per_cu);
        kernel<<<need < cap ? need : cap, NWV * 64, lds, st>>>(b);
    };
    const bool hs = a.w != nullptr;
    const int nt = a.Nm <= 16 ? 1 : 4;
    const bool full = a.Nm == 16 * nt;
#define MVIN_KW(BFV, PV, HV, NTV, FV) \
    if ((table_bf16 != 0) == BFV && a.P == PV && hs == HV && nt == NTV && full == FV) launch(key_addr_wave_kernel<D, BFV, PV, HV, NTV, FV>);
#define MVIN_KWB(BFV, FV) \
    MVIN_KW(BFV, 1, false, 1, FV) MVIN_KW(BFV, 1, true, 1, FV) MVIN_KW(BFV, 2, false, 1, FV) MVIN_KW(BFV, 2, true, 1, FV) \
    MVIN_KW(BFV, 1, false, 4, FV) MVIN_KW(BFV, 1, true, 4, FV) MVIN_KW(BFV, 2, false, 4, FV) MVIN_KW(BFV, 2, true, 4, FV)
    MVIN_KWB(false, false) MVIN_KWB(true, false) MVIN_KWB(false, true) MVIN_KWB(true, true)
#undef MVIN_KWB
#undef MVIN_KW
    return err != hipSuccess ? err : hipGetLastError();
}

hipError_t launch_key_addr_wave16(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st) {
    return a.D == 16 ? launch_kw<16>(a, table_bf16, st) : launch_kw<32>(a, table_bf16, st);
}

}  // namespace mvin
