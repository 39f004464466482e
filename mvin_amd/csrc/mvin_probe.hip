// Measurement aid: the row gathers of mvin_gather_attn_l2_fwd with everything else removed, written the plain way.
// For every parent the fused kernel must read the K child rows and the K*K grandchild rows, once each
// (no cross-pair or intra-tree reuse is assumed).  This kernel reads exactly those rows with the
// same 16-byte lane loads -- one wave per parent, 8 loads in flight per lane, the ids from the flat level lists
// mvin_expand_ids writes, fetched one round ahead -- adds the elements up and writes one float per parent: no
// adjacency chase, no softmax, no projection, no MFMA, no LDS tile, 32 waves per CU.  bench.py times it on the
// timed region's own table and pairs as a REFERENCE POINT for the fused kernel's row rate (a straightforward
// gather-and-sum over the same stream), next to the nominal L2 / HBM peaks and the idealised gather ceilings of
// scripts/micro/dma_probe.hip.  It is not an upper bound: on the metric workload the fused kernel's role-split
// gather loop (two parents in flight per CU, so a parent's repeated rows stay in L1) moves the same rows faster.
#include "mvin_kernels.h"

namespace mvin {

template <int RB, bool BF>      // RB: row bytes
__global__ __launch_bounds__(256) void gather_probe_l2_kernel(const void* __restrict__ table, const int32_t* __restrict__ ids1,
                                                              const int32_t* __restrict__ ids2, int64_t n_parents, int K,
                                                              float* __restrict__ sums) {
    constexpr int LPR = RB / 16, RPI = 64 / LPR;                 // lanes per row, rows per wave-instruction
    constexpr int INF = RPI <= 8 ? 8 : 64 / RPI;                 // loads in flight per lane
    constexpr int RPR = RPI * INF;                               // rows per round (<= 64: one id per lane)
    const int lane = threadIdx.x & 63;
    const int g = lane / LPR, c = lane % LPR;
    const int64_t nwave = (int64_t)gridDim.x * 4, wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const char* tab = reinterpret_cast<const char*>(table);
    const int R = K + K * K;                                     // rows per parent
    auto row_sum = [&](int id) -> float {
        const char* p = tab + (size_t)(unsigned)id * RB + c * 16;
        if constexpr (BF) {
            const uint4 r = *reinterpret_cast<const uint4*>(p);
            const float4 a = bf16x4_to_f32(make_uint2(r.x, r.y)), b = bf16x4_to_f32(make_uint2(r.z, r.w));
            return (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
        } else {
            const float4 a = *reinterpret_cast<const float4*>(p);
            return (a.x + a.y) + (a.z + a.w);
        }
    };
    for (int64_t p = wave0; p < n_parents; p += nwave) {        // one parent per wave and round
        const int32_t* l1 = ids1 + p * K;
        const int32_t* l2 = ids2 + p * (int64_t)K * K;
        auto load_ids = [&](int r0) -> int {                     // one coalesced load: lane l holds the id of row r0 + l
            const int r = r0 + lane;
            return (lane < RPR && r < R) ? (r < K ? l1[r] : l2[r - K]) : -1;
        };
        float acc = 0.f;
        int nid = load_ids(0);
        for (int r0 = 0; r0 < R; r0 += RPR) {
            const int idv = nid;
            nid = load_ids(r0 + RPR);                            // next round's ids: in flight under this round's rows
            float v[INF];
#pragma unroll
            for (int i = 0; i < INF; ++i) {
                const int id = __shfl(idv, i * RPI + g, kWave);
                v[i] = id >= 0 ? row_sum(id) : 0.f;
            }
#pragma unroll
            for (int i = 0; i < INF; ++i) acc += v[i];
        }
        acc = wave_sum_fast(acc);
        if (lane == 0) sums[p] = acc;
    }
}

hipError_t launch_gather_probe_l2(const void* table, const int32_t* ids1, const int32_t* ids2, int64_t n_parents, int K,
                                  int D, int table_bf16, float* sums, hipStream_t st) {
    const int rb = D * (table_bf16 ? 2 : 4);
    const int64_t cap = 256 * 8;                                 // 8 workgroups of 4 waves per CU
    const int grid = (int)((n_parents + 3) / 4 < cap ? (n_parents + 3) / 4 : cap);
#define MVIN_PROBE(RBV, BFV) gather_probe_l2_kernel<RBV, BFV><<<grid, 256, 0, st>>>(table, ids1, ids2, n_parents, K, sums)
    if (table_bf16) {
        switch (rb) {
            case 64: MVIN_PROBE(64, true); break;
            case 128: MVIN_PROBE(128, true); break;
            case 256: MVIN_PROBE(256, true); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (rb) {
            case 64: MVIN_PROBE(64, false); break;
            case 128: MVIN_PROBE(128, false); break;
            case 256: MVIN_PROBE(256, false); break;
            case 512: MVIN_PROBE(512, false); break;
            default: return hipErrorInvalidValue;
        }
    }
#undef MVIN_PROBE
    return hipGetLastError();
}

}  // namespace mvin
