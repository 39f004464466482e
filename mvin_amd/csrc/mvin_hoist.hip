// Entity-table ("hoisted") mode of the two deepest levels -- SURVEY.md 7.3-c route 2b.
// Everything aggregator (0,.) does below level L-2 before its ReLU is linear in entity-table
// rows and its attention weights do not depend on the pair (7.3-a), so
//     S[e]  = (1/K) sum_k softmax_k(t0[adj_r[e,k]]) * E[adj_e[e,k]]
// is a per-ENTITY vector, and the per-pair work of model.py:295-305 at hop L-2 becomes
//     nagg1[x] = (1/K) sum_k softmax_k(t1[adj_r[x,k]]) * relu(R1[adj_e[x,k]] + d_pair)
// with R1 = (E.W1 + S.W2).A0 another per-entity table (host side: mvin_amd/model.py).
// Both are the same primitive: a fixed-fan-out attention mix of table rows,
//     out[i] = (1/K) sum_k w_k(x_i) * f(T[adj_e[x_i,k]] + rowbias[i / npg]),   f = id | relu
// One wave per node: the node's K child ids and relation logits arrive in one coalesced load
// per 64 children, the softmax over K is a wave reduction, then 16-byte lane loads of the rows
// (D/4 lanes per row, 64/(D/4) rows per wave-instruction, 8 loads in flight per lane).
#include "mvin_kernels.h"

namespace mvin {

template <int KC, bool BF>
__global__ __launch_bounds__(kBlock) void gather_mix_kernel(GatherMixArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D, K = a.K;
    const int lpr = 1 << a.lpr_log2, rpw = kWave >> a.lpr_log2;
    const int g = lane >> a.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    const float invK = 1.f / (float)K;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < a.nodes; i += (int64_t)gridDim.x * 4) {
        const int64_t x = a.node_ids ? (int64_t)a.node_ids[i] : i;
        const int32_t* ae = a.adj_e + x * K;
        const int32_t* ar = a.adj_r + x * K;
        int yid[KC];
        float wgt[KC];
        float mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < KC; ++q) {
            const int k = q * kWave + lane;
            const bool v = k < K;
            yid[q] = v ? ae[k] : 0;
            float s = v ? 0.f : -INFINITY;
            if (v && a.rel_score) s = a.rel_score[ar[k]];
            wgt[q] = s;
            mx = fmaxf(mx, s);
        }
        if (a.rel_score) {   // tf.nn.softmax over the K neighbors (aggregators.py:139), then 1/K (:144)
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < KC; ++q) {
                const float e = (q * kWave + lane < K) ? expf(wgt[q] - mx) : 0.f;
                wgt[q] = e;
                sum += e;
            }
            sum = wave_sum(sum);
#pragma unroll
            for (int q = 0; q < KC; ++q) wgt[q] = (wgt[q] / sum) * invK;
        } else {             // plain mean (aggregators.py:148-152)
#pragma unroll
            for (int q = 0; q < KC; ++q) wgt[q] = (q * kWave + lane < K) ? invK : 0.f;
        }
        float4 bias = z4;
        if (a.rowbias && cact) bias = reinterpret_cast<const float4*>(a.rowbias + (i / a.npg) * D)[c];
        float4 acc = z4;
#pragma unroll
        for (int q = 0; q < KC; ++q) {
            const int kc = K - q * kWave < kWave ? K - q * kWave : kWave;   // wave-uniform
            for (int j0 = 0; j0 < kc; j0 += rpw * 8) {
                float4 v[8];
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kl = j0 + u * rpw + g;
                    const bool ok = kl < kc;
                    const int src = ok ? kl : 0;
                    const int id = __shfl(yid[q], src, kWave);
                    const float wk = __shfl(wgt[q], src, kWave);
                    w[u] = ok ? wk : 0.f;
                    v[u] = z4;
                    if (ok && cact) v[u] = load_row4(a.table, BF, id, D, c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float4 r = make_float4(v[u].x + bias.x, v[u].y + bias.y, v[u].z + bias.z, v[u].w + bias.w);
                    if (a.relu) r = make_float4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
                    acc = f4_fma(w[u], r, acc);
                }
            }
        }
        acc = group_xor_sum(acc, lpr);
        if (cact && g == 0) *reinterpret_cast<float4*>(a.out + i * D + (c << 2)) = acc;
    }
}

hipError_t launch_gather_mix(const GatherMixArgs& a, hipStream_t st) {
    const int64_t nblk = (a.nodes + 3) / 4;
    const int64_t cap = 256 * 8;   // 8 workgroups per CU: the kernel needs few registers
    const int grid = (int)(nblk < cap ? nblk : cap);
    const int kc = (a.K + kWave - 1) / kWave;
#define MVIN_GM(KCV)                                                                   \
    if (a.table_bf16) gather_mix_kernel<KCV, true><<<grid, kBlock, 0, st>>>(a);        \
    else gather_mix_kernel<KCV, false><<<grid, kBlock, 0, st>>>(a);
    if (kc <= 1) { MVIN_GM(1) }
    else if (kc <= 2) { MVIN_GM(2) }
    else if (kc <= 4) { MVIN_GM(4) }
    else return hipErrorInvalidValue;
#undef MVIN_GM
    return hipGetLastError();
}

}  // namespace mvin
