// MVIN._key_addressing (model.py:161-240) for pairs grouped by user at dim 16 -- the dimension of every run script the
// reference ships (src/bash/mvin_*.sh) -- with ONE WAVE per user segment.  Same interface and arithmetic as
// key_addr_dense_kernel (mvin_keyaddr_dense.hip); the mapping is the opposite.  At D = 16 a user's 2*P*Nm ripple-set
// rows are 16 KB and every product is a handful of 16x16x4 MFMA steps, so the workgroup-wide phases of the dense kernel
// (stage -> barrier -> U -> barrier -> logits -> barrier -> softmax -> barrier -> reads, 48 k cycles per segment with four
// segments in flight per CU) are mostly barrier and latency.  Here nothing is shared between waves and there is no
// workgroup barrier after the prologue:
//   * U_m = R_KGE[r_m] . h_m (model.py:214-216) and the tail rows t_m live in REGISTERS, already in MFMA B-fragment
//     layout (32 + 32 VGPRs for two hops of 64 memories): lane (q, j) = (lane / 16, lane % 16) holds
//     U[m = 16t + j][n = 4s + q] (t, s < 4) and T[m = 4s + q][n = j] (s < 16).  U is computed in that layout on the VALU
//     from the head rows and R_KGE rows read from LDS (relation stride 260 words: lanes of different relations land on
//     different banks, lanes of the same relation read the same address);
//   * per tile of 16 pairs: logits L = E[items] . U^T (:219-220) as 16 MFMA steps per hop -> accumulators hold
//     L[pair 4q + r][m = 16t + j]; softmax over the memories (:223) = in-lane over t, then a 16-lane DPP row reduction;
//     the un-normalised weights cross from accumulator layout to A-fragment layout through a 16 x 68-word LDS tile
//     private to the wave (conflict-free both ways); reads o = P . T (:229) as 16 MFMA steps per hop; rows of 64 bytes
//     go out straight from the accumulators;
//   * the h-set read (:162-197) once per user: lane groups hold the head rows of hop 0, 16-lane DPP reductions.
// 16 waves per CU, each on its own user, hide each other's dependent loads (segment -> ids -> rows -> item ids -> item rows).
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kW16Waves = 4;                 // waves per workgroup (they only share the LDS copy of R_KGE)
constexpr int kW16LdP = 68;                  // row stride (words) of the wave's 16 x 64 weight tile
constexpr int kW16LdR = 260;                 // words per relation matrix in LDS
constexpr int kW16Ids = 2 * 3 * 64;          // the user's ids: [hop][h | r | t][64], padded with -1
constexpr int kW16LdH = 20;                  // row stride (words) of a staged 16-row head tile: 16-byte reads of 16 rows hit 16 bank groups
constexpr int kW16PerWave = 16 * kW16LdP + kW16Ids + 16 + 16 + 2 * 16 * kW16LdH;   // + h-set read + pair indices + two head tiles

// exp(x) for the softmax arguments (x = logit - max <= 0, or discarded by a select): the argument reduction of the
// library routine -- x * log2(e) split into an integer and a fraction with the product's rounding error folded back in by
// fma -- without its overflow / underflow selects (ldexp flushes to 0 on its own): 7 instructions instead of 12
__device__ __forceinline__ float w16_exp(float x) {
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
template <bool BF>
__device__ __forceinline__ float w16_elem(const void* E, unsigned row, int n) {
    const char* b = reinterpret_cast<const char*>(E);
    if (BF) return __uint_as_float((unsigned)*reinterpret_cast<const uint16_t*>(b + (row * 32u + 2u * n)) << 16);
    return *reinterpret_cast<const float*>(b + (row * 64u + 4u * n));
}

// elements 4c .. 4c+3 of a 16-wide row
template <bool BF>
__device__ __forceinline__ float4 w16_chunk(const void* E, unsigned row, int c) {
    const char* b = reinterpret_cast<const char*>(E);
    if (BF) {
        const uint2 v = *reinterpret_cast<const uint2*>(b + (row * 32u + 8u * c));
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(b + (row * 64u + 16u * c));
}

// P (1 or 2 hops) and the presence of the h-set read are compile-time: as run-time branches they cut the per-user section
// into dozens of basic blocks that hipcc could neither schedule nor allocate (300 spills)
// NT = memory tiles of 16 per hop (1 for n_memory <= 16 -- amazon-book's shipped setting --, else 4)
// FULL: n_memory == 16 * NT, no padding memories to mask
template <bool BF, int P, bool HAS_SET, int NT, bool FULL>
__global__ __launch_bounds__(kW16Waves * 64, 4) void key_addr_wave16_kernel(KeyAddrGroupedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int Nm = a.Nm;
    constexpr int Ph = P;
    constexpr bool has_set = HAS_SET;
    float* sR = smem;                                            // [nR][260]
    float* sW = smem + (size_t)a.nR * kW16LdR + (size_t)wave * kW16PerWave;
    float* sP = sW;                                              // [16][68]
    int* sIds = reinterpret_cast<int*>(sW + 16 * kW16LdP);       // [2][3][64]
    float* sHset = sW + 16 * kW16LdP + kW16Ids;                  // [16]
    int* sOrig = reinterpret_cast<int*>(sHset + 16);             // [16]
    float* sHt = sHset + 32;                                     // [2][16][20]
    for (int i = tid; i < a.nR * 256; i += kW16Waves * 64) sR[(i >> 8) * kW16LdR + (i & 255)] = a.R[i];
    __syncthreads();

    constexpr int slot0 = HAS_SET ? 1 : 0;
    const unsigned max_id = (unsigned)(a.n_entity - 1);
    const int nseg = a.nseg_dev ? *a.nseg_dev : a.nseg;
    const int nw = gridDim.x * kW16Waves;
    for (int seg = blockIdx.x * kW16Waves + wave; seg < nseg; seg += nw) {
        const int u = __builtin_amdgcn_readfirstlane(a.seg_user[seg]);
        const int p0 = __builtin_amdgcn_readfirstlane(a.seg_ptr[seg]);
        const int p1 = __builtin_amdgcn_readfirstlane(a.seg_ptr[seg + 1]);
        // the first tile's item rows hang on three dependent loads (pair_index -> items -> E row): start them now
        auto item_frag = [&](int t0, float (&av)[4], int& orig) {
            const int p = t0 + j;
            const int o = a.pair_index[p < p1 ? p : p1 - 1];
            const int64_t item = a.items64 ? a.items64[o] : (int64_t)a.items32[o];
            const unsigned row = min((unsigned)item, max_id);
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = w16_elem<BF>(a.E, row, 4 * s + q);
            orig = p < p1 ? o : -1;
        };
        float av[4];
        int orig;
        item_frag(p0, av, orig);
        // ---- the user's ids -> LDS ([hop][h | r | t][64], -1 beyond Nm) ----
        wave_lds_sync();                                         // the previous segment's reads of sIds / sP are done
        const int32_t* ub = a.uts + (int64_t)u * Ph * 3 * Nm;
        for (int i = lane; i < Ph * 3 * 64; i += 64) {
            const int hx = i >> 6, m = i & 63;
            sIds[i] = m < Nm ? ub[hx * Nm + m] : -1;
        }
        wave_lds_sync();
        // Head rows reach the lanes through LDS: a tile of 16 rows (memories m = 16t + j of one hop) is ONE coalesced
        // wave-load -- lane (q, j) fetches chunk q of row j -- written to a double-buffered 16 x 20-word tile, from which
        // every lane reads the four chunks of ITS row (the four q lanes the same address).  Loading the rows straight into
        // the lanes that need them took four 16-byte wave-loads per tile, each fetching every row four times over: the
        // texture addresser was busy 74 % of the kernel (rocprofv3 TA_TA_BUSY), now NT * (2 + P) wave-loads per user
        // instead of 4 * NT * (2 + P).  The load of tile i + 1 is in flight while tile i is used.
        int stage_i = 0;                                         // tiles staged so far (buffer = parity)
        auto head_chunk = [&](int hop, int t) {
            const int idh = sIds[(hop * 3 + 0) * 64 + 16 * t + j];
            return w16_chunk<BF>(a.E, idh >= 0 ? idh : 0, q);     // padding memories read row 0 (finite; masked by position)
        };
        auto stage = [&](const float4& v) {
            float* dst = sHt + (stage_i & 1) * 16 * kW16LdH;
            wave_lds_sync();                                     // the reads of this buffer two tiles ago are done
            *reinterpret_cast<float4*>(dst + j * kW16LdH + 4 * q) = v;
            wave_lds_sync();
            ++stage_i;
            return dst + j * kW16LdH;                            // this lane's row
        };
        if (has_set) {
            // o_hset = sum_m softmax_m(h0_m . w) h0_m (:162-197), BEFORE the fragments occupy 64 registers: this lane's rows
            // are m = 16t + j (the four q rows hold copies); two passes over the NT tiles of hop 0
            float lg[NT];
            float4 cur = head_chunk(0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float4 nxt = cur;
                if (t + 1 < NT) nxt = head_chunk(0, t + 1);
                const float* hr = stage(cur);
                float dl = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * c);
                    dl = fmaf(hv.x, a.w[4 * c], fmaf(hv.y, a.w[4 * c + 1], fmaf(hv.z, a.w[4 * c + 2], fmaf(hv.w, a.w[4 * c + 3], dl))));
                }
                lg[t] = (16 * t + j) < Nm ? dl : -INFINITY;
                cur = nxt;
            }
            float mx = lg[0];
#pragma unroll
            for (int t = 1; t < NT; ++t) mx = fmaxf(mx, lg[t]);
            mx = group_max(mx, 4);
            float e[NT], z = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                e[t] = (16 * t + j) < Nm ? w16_exp(lg[t] - mx) : 0.f;
                z += e[t];
            }
            z = group_sum(z, 4);
            float part[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) part[k] = 0.f;
            cur = head_chunk(0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float4 nxt = cur;
                if (t + 1 < NT) nxt = head_chunk(0, t + 1);
                const float* hr = stage(cur);                    // second pass (e = 0 on padding)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * c);
                    part[4 * c] = fmaf(e[t], hv.x, part[4 * c]);
                    part[4 * c + 1] = fmaf(e[t], hv.y, part[4 * c + 1]);
                    part[4 * c + 2] = fmaf(e[t], hv.z, part[4 * c + 2]);
                    part[4 * c + 3] = fmaf(e[t], hv.w, part[4 * c + 3]);
                }
                cur = nxt;
            }
            const float inv = 1.f / z;
            float mine = 0.f;                                    // lane k keeps sum k (selects, no divergent branches)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float sum = group_sum(part[k], 4);
                mine = j == k ? sum : mine;
            }
            sHset[j] = mine * inv;                               // the four q rows hold copies: same value to the same address
        }
        // ---- U and T fragments -> registers ----
        float Bl[2][NT][4], Tf[2][4 * NT];
        {
            float4 cur = head_chunk(0, 0);
#pragma unroll
            for (int i = 0; i < NT * P; ++i) {
                float4 nxt = cur;
                if (i + 1 < NT * P) nxt = head_chunk((i + 1) / NT, (i + 1) % NT);
                const float* hr = stage(cur);
                const int hop = i / NT, t = i % NT;
                int r = min((unsigned)sIds[(hop * 3 + 1) * 64 + 16 * t + j], (unsigned)(a.nR - 1));
                if (a.NRL & 1) r = 0;                            // MVIN_W16_DBG bit 0 (measurement only): every lane reads R_KGE[0]
                const float* Rr = sR + (size_t)r * kW16LdR + q * 16;
                f32x2 d[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};       // even / odd k apart: v_pk_fma_f32
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * c);
                    const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float4 v = *reinterpret_cast<const float4*>(Rr + 64 * s + 4 * c);          // R[r][n = 4s + q][4c ..]
                        d[s] = __builtin_elementwise_fma(f32x2{v.x, v.y}, h01, d[s]);
                        d[s] = __builtin_elementwise_fma(f32x2{v.z, v.w}, h23, d[s]);
                    }
                    // five LDS reads at a time: left alone hipcc issues all of a row's up front (80 registers) and spills
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) Bl[hop][t][s] = d[s].x + d[s].y;         // U[m = 16t + j][n = 4s + q]
                cur = nxt;
            }
        }
#pragma unroll
        for (int hop = 0; hop < P; ++hop) {
#pragma unroll
            for (int s = 0; s < 4 * NT; ++s) {
                const int idt = sIds[(hop * 3 + 2) * 64 + 4 * s + q];
                Tf[hop][s] = w16_elem<BF>(a.E, idt >= 0 ? idt : 0, j);          // T[m = 4s + q][n = j]
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // a do-while: segments are never empty, and with a guarded loop LLVM sinks the whole fragment arithmetic above into
        // the guarded block, away from its loads -- every loaded value then lives across the branch (580 spills)
        int t0 = p0;
        do {
            float avn[4];
            int orig_n;
            item_frag(t0 + 16 < p1 ? t0 + 16 : t0, avn, orig_n);     // the next tile's rows land under this tile's work
            wave_lds_sync();                                     // previous tile's reads of sOrig / sHset writes visible
            if (q == 0) sOrig[j] = orig;
            wave_lds_sync();
            int og[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) og[r] = sOrig[4 * q + r];
#pragma unroll
            for (int hop = 0; hop < 2; ++hop) {
                if (hop < P) {
                    f32x4 acc[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Bl[hop][t][s], acc[t], 0, 0, 0);
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
                            const float e = (FULL || (16 * t + j) < Nm) ? w16_exp(acc[t][r] - mx) : 0.f;
                            sP[(4 * q + r) * kW16LdP + 16 * t + j] = e;
                            z += e;
                        }
                        zinv[r] = 1.f / group_sum(z, 4);
                    }
                    wave_lds_sync();
                    // reads o = P . T (:229): A[i = j][k = q] = P[pair j][m = 4s + q]
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    const float* ap = sP + j * kW16LdP + q;
#pragma unroll
                    for (int s = 0; s < 4 * NT; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s], Tf[hop][s], o, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (og[r] >= 0) a.out[(int64_t)og[r] * a.ldo + (size_t)(slot0 + hop) * 16 + j] = o[r] * zinv[r];
                }
            }
            if (has_set) {
                const float hs = sHset[j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (og[r] >= 0) a.out[(int64_t)og[r] * a.ldo + j] = hs;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = avn[s];
            orig = orig_n;
            t0 += 16;
        } while (t0 < p1);
    }
}

static size_t w16_lds_bytes(int nR) { return ((size_t)nR * kW16LdR + (size_t)kW16Waves * kW16PerWave) * sizeof(float); }

bool key_addr_wave16_supported(int D, int P, int Nm, int nR) {
    static const bool off = getenv("MVIN_KA_WAVE16") != nullptr && getenv("MVIN_KA_WAVE16")[0] == '0';
    return !off && D == 16 && (P == 1 || P == 2) && Nm >= 1 && Nm <= 64 && nR >= 1 && w16_lds_bytes(nR) <= 80 * 1024;      // two workgroups per CU at least
}

// rows are addressed by 32-bit byte offsets
bool key_addr_wave16_applies(const KeyAddrGroupedArgs& a) {
    return key_addr_wave16_supported(a.D, a.P, a.Nm, a.nR) && a.n_entity > 0 && (uint64_t)a.n_entity * 64u < (1ull << 32);
}

hipError_t launch_key_addr_wave16(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st) {
    const size_t lds = w16_lds_bytes(a.nR);
    hipError_t err = hipSuccess;
    auto launch = [&](auto kernel) {
        // persistent grid: as many workgroups as the CUs hold (LDS decides: 44 KB at nR = 9 -> 3 per CU = 12 waves; 75 KB at nR = 39 -> 2)
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
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kW16Waves * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            last_k = k;
            last_lds = lds;
            last_per_cu = per_cu;
        }
        const int need = (a.nseg + kW16Waves - 1) / kW16Waves;
        static const int dbg = getenv("MVIN_W16_DBG") ? atoi(getenv("MVIN_W16_DBG")) : 0;      // measurement knobs
        KeyAddrGroupedArgs b = a;
        b.NRL = dbg;
        const int per_cu = (dbg & 4) ? 1 : (dbg & 2) ? 2 : last_per_cu;
        const int cap = 256 * per_cu;
        kernel<<<need < cap ? need : cap, kW16Waves * 64, lds, st>>>(b);
    };
    const bool hs = a.w != nullptr;
    const int nt = a.Nm <= 16 ? 1 : 4;
    const bool full = a.Nm == 16 * nt;
#define MVIN_W16(BFV, PV, HV, NTV, FV) \
    if ((table_bf16 != 0) == BFV && a.P == PV && hs == HV && nt == NTV && full == FV) launch(key_addr_wave16_kernel<BFV, PV, HV, NTV, FV>);
#define MVIN_W16B(BFV, FV) \
    MVIN_W16(BFV, 1, false, 1, FV) MVIN_W16(BFV, 1, true, 1, FV) MVIN_W16(BFV, 2, false, 1, FV) MVIN_W16(BFV, 2, true, 1, FV) \
    MVIN_W16(BFV, 1, false, 4, FV) MVIN_W16(BFV, 1, true, 4, FV) MVIN_W16(BFV, 2, false, 4, FV) MVIN_W16(BFV, 2, true, 4, FV)
    MVIN_W16B(false, false) MVIN_W16B(true, false) MVIN_W16B(false, true) MVIN_W16B(true, true)
#undef MVIN_W16B
#undef MVIN_W16
    return err != hipSuccess ? err : hipGetLastError();
}

}  // namespace mvin
