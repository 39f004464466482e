// MVIN._key_addressing (model.py:161-240) for pairs GROUPED BY USER.
//
// The ripple sets a pair reads are its user's (train.py:117-120: memories_x[i] = user_triplet_set[user][i][x]),
// so every pair of one user gathers the same 2*P*Nm entity rows.  With the batch's pairs sorted by user, one
// workgroup owns one user segment:
//   stage   : the user's ripple-set ids (uts[u], [P,3,Nm] int32), its h / t rows -> LDS once; the relations that
//             occur in the user's memories are compacted to a local list (<= min(nR, P*Nm) entries)
//   h-set   : o_hset = sum_m softmax_m(h0_m . w_h) h0_m  (:162-197) -- does not depend on the item: once per user
//   per tile of 16 pairs:
//     V     : V[pair, r, :] = E[item_pair] . R_KGE[r] for the user's relations, v_mfma_f32_16x16x4_f32
//             (16 pairs = one MFMA row tile; B fragments of R_KGE[r] straight from L2) -> LDS.  This is the
//             (R h).v = h.(v R) re-association of mvin_keyaddr.hip, but the [B, nR, D] tensor never exists.
//     pairs : one wave per pair and hop: logits s_m = h_m . V[pair, r_m, :] (rows and V from LDS), softmax over
//             Nm (:223), o = sum_m p_m t_m (:229) -> out[pair, slot*D ...]
// HBM / L2 traffic per USER: 2*P*Nm rows; per pair: one item row + the output.  Same arithmetic per pair as
// key_addr_kernel up to fp32 summation order.
//
// Supported: D in {16, 32, 64, 128}, Nm <= 256, rows + V tile within the 160 KB LDS (launcher checks).
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGW = 16;      // waves per workgroup: one pair of a 16-pair tile each
constexpr int kGT = 16;      // pairs per tile (one MFMA row tile)

struct KaGroupedLds {
    size_t h, t, ei, v, idh, idt, rl, map, list, lg, hset, orig, total;
};

static KaGroupedLds ka_grouped_layout(int D, int P, int Nm, int nR, int NRL) {
    const int Ph = P > 0 ? P : 1;
    KaGroupedLds L{};
    size_t o = 0;
    L.h = o; o += (size_t)Ph * Nm * D;
    L.t = o; o += (size_t)P * Nm * D;
    L.v = o; o += (size_t)kGT * NRL * D;
    L.ei = o; o += (size_t)kGT * (D + 2);
    L.lg = o; o += (size_t)kGW * Nm;
    L.hset = o; o += D;
    L.idh = o; o += (size_t)Ph * Nm;
    L.idt = o; o += (size_t)Ph * Nm;
    L.rl = o; o += (size_t)Ph * Nm;
    L.map = o; o += nR;
    L.list = o; o += NRL;
    L.orig = o; o += kGT + 2;
    L.total = o * 4;
    return L;
}

template <int D, bool BF>
__global__ __launch_bounds__(kGW * 64) void key_addr_grouped_kernel(KeyAddrGroupedArgs a, KaGroupedLds L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LPR = D / 4, RPW = 64 / LPR, NT = D / 16, KS = D / 4, LDE = D + 2;
    constexpr int LPR_L2 = (LPR == 4) ? 2 : (LPR == 8) ? 3 : (LPR == 16) ? 4 : 5;
    const int P = a.P, Nm = a.Nm, Ph = P > 0 ? P : 1, NRL = a.NRL;
    float* sH = smem + L.h;
    float* sT = smem + L.t;
    float* sV = smem + L.v;
    float* sEi = smem + L.ei;
    float* sLg = smem + L.lg;
    float* sHset = smem + L.hset;
    int* sIdH = reinterpret_cast<int*>(smem + L.idh);
    int* sIdT = reinterpret_cast<int*>(smem + L.idt);
    int* sRl = reinterpret_cast<int*>(smem + L.rl);
    int* sMap = reinterpret_cast<int*>(smem + L.map);
    int* sList = reinterpret_cast<int*>(smem + L.list);
    int* sOrig = reinterpret_cast<int*>(smem + L.orig);      // [kGT] original pair index (-1: padding), [kGT] = nrl

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPR, c = lane % LPR;
    const int q16 = lane >> 4, l16 = lane & 15;
    const bool has_set = a.w != nullptr;
    const unsigned emax = (unsigned)(a.n_entity > 0 ? a.n_entity - 1 : 0x7fffffff);    // last row of E
    const int slot0 = has_set ? 1 : 0;
    auto row4 = [&](int id) -> float4 { return load_row4(a.E, BF, id, D, c); };

    const int nseg = a.nseg_dev ? *a.nseg_dev : a.nseg;      // the segment count may live on the device (no host sync)
    for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const int u = a.seg_user[seg];
        const int p0 = a.seg_ptr[seg], p1 = a.seg_ptr[seg + 1];
        __syncthreads();                                     // previous segment fully consumed
        for (int i = tid; i < a.nR; i += kGW * 64) sMap[i] = 0;
        __syncthreads();
        // ---- the user's ripple-set ids; mark the relations in use ----
        const int32_t* ub = a.uts + (int64_t)u * Ph * 3 * Nm;
        for (int i = tid; i < Ph * Nm; i += kGW * 64) {
            const int hop = i / Nm, m = i - hop * Nm;
            sIdH[i] = (int)min((unsigned)ub[(hop * 3 + 0) * Nm + m], emax);     // clamped into the table, like every device id
            sIdT[i] = (int)min((unsigned)ub[(hop * 3 + 2) * Nm + m], emax);
            const int r = min((unsigned)ub[(hop * 3 + 1) * Nm + m], (unsigned)(a.nR - 1));   // indexes LDS below
            sRl[i] = r;
            if (hop < P) sMap[r] = 1;
        }
        __syncthreads();
        // ---- compact the relations in use: sMap[r] -> local index, sList[local] = r ----
        if (wave == 0) {
            int base = 0;
            for (int r0 = 0; r0 < a.nR; r0 += 64) {
                const int r = r0 + lane;
                const bool f = r < a.nR && sMap[r] != 0;
                const unsigned long long bal = __ballot(f);
                const int idx = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (f) {
                    sMap[r] = idx;
                    sList[idx] = r;
                }
                base += __popcll(bal);
            }
            if (lane == 0) sOrig[kGT] = base;
        }
        __syncthreads();
        const int nrl = sOrig[kGT];
        // ---- stage the rows: RPW rows per wave-instruction ----
        for (int i = wave * RPW + g; i < Ph * Nm; i += kGW * RPW) {
            const float4 h = row4(sIdH[i]);
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool hop_row = i < P * Nm;
            if (hop_row) t = row4(sIdT[i]);
            *reinterpret_cast<float4*>(sH + (size_t)i * D + 4 * c) = h;
            if (hop_row) {
                *reinterpret_cast<float4*>(sT + (size_t)i * D + 4 * c) = t;
                if (c == 0) sRl[i] = sMap[sRl[i]];
            }
        }
        __syncthreads();
        // one attention read over rows in LDS: logits (rows `keys` . per-row vector) -> softmax -> sum p_m vals[m]
        // `vsel(m)` gives the LDS address of the D-vector multiplying key row m
        float* lg = sLg + (size_t)wave * Nm;
        auto attend = [&](const float* keys, const float* vals, auto vsel) -> float4 {
#pragma unroll 4
            for (int m0 = 0; m0 < Nm; m0 += RPW) {
                const int m = m0 + g;
                float d = 0.f;
                if (m < Nm) {
                    const float4 k4 = *reinterpret_cast<const float4*>(keys + (size_t)m * D + 4 * c);
                    const float4 v4 = *reinterpret_cast<const float4*>(vsel(m) + 4 * c);
                    d = fmaf(k4.x, v4.x, fmaf(k4.y, v4.y, fmaf(k4.z, v4.z, k4.w * v4.w)));
                }
                d = group_sum(d, LPR_L2);
                if (c == 0 && m < Nm) lg[m] = d;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // this wave's LDS writes before its reads
            float mx = -INFINITY;
            for (int m = lane; m < Nm; m += 64) mx = fmaxf(mx, lg[m]);
            mx = wave_max(mx);
            float z = 0.f;
            for (int m = lane; m < Nm; m += 64) {
                const float e = expf(lg[m] - mx);
                lg[m] = e;
                z += e;
            }
            z = wave_sum(z);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int m0 = 0; m0 < Nm; m0 += RPW) {
                const int m = m0 + g;
                if (m < Nm) acc = f4_fma(lg[m], *reinterpret_cast<const float4*>(vals + (size_t)m * D + 4 * c), acc);
            }
            acc = group_xor_sum(acc, LPR);
            const float inv = 1.f / z;
            return make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        };
        // ---- h-set read (:162-197): the same for every pair of the user ----
        if (has_set && wave == 0) {
            const float4 o = attend(sH, sH, [&](int) { return a.w; });
            if (g == 0) *reinterpret_cast<float4*>(sHset + 4 * c) = o;
        }
        // ---- the user's pairs, 16 at a time ----
        for (int t0 = p0; t0 < p1; t0 += kGT) {
            __syncthreads();                                 // previous tile's sEi / sV consumed; sHset written
            if (tid < kGT * LPR) {
                const int i = tid / LPR, cc = tid % LPR;
                const int p = t0 + i;
                const int orig = a.pair_index[p < p1 ? p : p1 - 1];
                const int64_t item = (int64_t)min((uint64_t)(a.items64 ? a.items64[orig] : (int64_t)a.items32[orig]), (uint64_t)emax);
                const float4 e = load_row4(a.E, BF, item, D, cc);
                float* dst = sEi + i * LDE + 4 * cc;
                *reinterpret_cast<float2*>(dst) = make_float2(e.x, e.y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(e.z, e.w);
                if (cc == 0) sOrig[i] = p < p1 ? orig : -1;
            }
            __syncthreads();
            // V[pair, rl, :] = E[item_pair] . R_KGE[sList[rl]]: wave -> (column tile, relations rl0, rl0+step, ...)
            if (P > 0) {
                constexpr int STEP = kGW / NT > 0 ? kGW / NT : 1;
                constexpr int RB = (KS <= 16) ? 2 : 1;          // relations whose B fragments are in flight together
                const int nt = wave % NT;
                for (int rl0 = wave / NT; rl0 < nrl; rl0 += RB * STEP) {
                    float bfrag[RB][KS];
#pragma unroll
                    for (int j = 0; j < RB; ++j) {
                        const int rl = rl0 + j * STEP;
                        const float* Rr = a.R + (size_t)sList[rl < nrl ? rl : rl0] * D * D + 16 * nt + l16;
#pragma unroll
                        for (int k = 0; k < KS; ++k) bfrag[j][k] = Rr[(size_t)(4 * k + q16) * D];
                    }
#pragma unroll
                    for (int j = 0; j < RB; ++j) {
                        const int rl = rl0 + j * STEP;
                        if (rl >= nrl) break;
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < KS; ++k)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sEi[l16 * LDE + 4 * k + q16], bfrag[j][k], acc, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) sV[((size_t)(4 * q16 + i) * NRL + rl) * D + 16 * nt + l16] = acc[i];
                    }
                }
            }
            __syncthreads();
            for (int pi = wave; pi < kGT; pi += kGW) {
                const int orig = sOrig[pi];
                if (orig < 0) continue;
                float* orow = a.out + (int64_t)orig * a.ldo;
                if (has_set && g == 0) *reinterpret_cast<float4*>(orow + 4 * c) = *reinterpret_cast<const float4*>(sHset + 4 * c);
                for (int hop = 0; hop < P; ++hop) {
                    const int* rl = sRl + hop * Nm;
                    const float* vb = sV + (size_t)pi * NRL * D;
                    const float4 o = attend(sH + (size_t)hop * Nm * D, sT + (size_t)hop * Nm * D,
                                            [&](int m) { return vb + (size_t)rl[m] * D; });
                    if (g == 0) *reinterpret_cast<float4*>(orow + (size_t)(slot0 + hop) * D + 4 * c) = o;
                }
            }
        }
    }
}

size_t key_addr_grouped_lds_bytes(int D, int P, int Nm, int nR) {
    const int nrl = nR < P * Nm ? nR : P * Nm;
    return ka_grouped_layout(D, P, Nm, nR, nrl > 0 ? nrl : 1).total;
}

bool key_addr_grouped_supported(int D, int P, int Nm, int nR) {
    const bool dok = D == 16 || D == 32 || D == 64 || D == 128;
    if (key_addr_dense_supported(D, P, Nm, nR)) return true;
    return dok && Nm >= 1 && Nm <= 256 && P >= 0 && P <= 8 && nR >= 1 && nR <= 4096 &&
           key_addr_grouped_lds_bytes(D, P, Nm, nR) <= 160 * 1024;
}

template <int D>
static hipError_t launch_kag(KeyAddrGroupedArgs a, int table_bf16, hipStream_t st) {
    const int nrl = a.nR < a.P * a.Nm ? a.nR : a.P * a.Nm;
    a.NRL = nrl > 0 ? nrl : 1;
    const KaGroupedLds L = ka_grouped_layout(D, a.P, a.Nm, a.nR, a.NRL);
    const int per_cu = (int)((160 * 1024) / L.total) < 2 ? 1 : 2;
    const int cap = 256 * per_cu;
    const int grid = a.nseg < cap ? a.nseg : cap;
    hipError_t e = hipSuccess;
    if (table_bf16) {
        auto k = key_addr_grouped_kernel<D, true>;
        if (L.total > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
        if (e != hipSuccess) return e;
        k<<<grid, kGW * 64, L.total, st>>>(a, L);
    } else {
        auto k = key_addr_grouped_kernel<D, false>;
        if (L.total > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
        if (e != hipSuccess) return e;
        k<<<grid, kGW * 64, L.total, st>>>(a, L);
    }
    return hipGetLastError();
}

hipError_t launch_key_addr_grouped(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st) {
    // D = 16 (the reference's shipped dimension) and 32, one or two hops of <= 64 memories: one wave per user
    // (MVIN_KA_WAVE=0 / MVIN_KA_WAVE32=0: A/B)
    if (key_addr_wave16_applies(a)) return launch_key_addr_wave16(a, table_bf16, st);
    // the dense (all-MFMA) form whenever its LDS footprint fits; MVIN_KA_DENSE=0 keeps this file's kernel (A/B)
    static const char* dense_env = getenv("MVIN_KA_DENSE");
    if (!(dense_env && dense_env[0] == '0') && key_addr_dense_supported(a.D, a.P, a.Nm, a.nR)) {
        // the same kernel over static per-user records, where the caller built them (MVIN_KA_STATIC=0: A/B)
        if (key_addr_static_applies(a, table_bf16)) return launch_key_addr_static(a, st);
        return launch_key_addr_dense(a, table_bf16, st);
    }
    switch (a.D) {
        case 16: return launch_kag<16>(a, table_bf16, st);
        case 32: return launch_kag<32>(a, table_bf16, st);
        case 64: return launch_kag<64>(a, table_bf16, st);
        case 128: return launch_kag<128>(a, table_bf16, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
