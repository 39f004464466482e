// MVIN._key_addressing attention reads (model.py:161-240), all preference hops of a pair in
// ONE pass: a wave owns a pair, keeps every ripple-set head row in registers between the
// logit pass and the weighted-sum pass (each table row is read exactly once, with all of a
// hop's 2*Nm row loads in flight together), and writes the concatenated o-vectors
// [o_hset | o_hop0 | o_hop1 ...] (model.py:204-232) that feed the user MLP.
//   hop logits    s_m = h_m . V[b, r_m, :]   with V[b,r,:] = E[item_b] . R_KGE[r]  ((R h).v == h.(v R))
//   h-set logits  s_m = h0_m . w_h           (user term and bias cancel in the softmax, :171-189)
#include "mvin_kernels.h"

namespace mvin {

__device__ __forceinline__ float dot4(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// BUF: fp32 table below 4 GiB read through a buffer descriptor (32-bit byte offsets: one VALU per
// row address instead of a 64-bit multiply-add chain, and half the address registers)
template <int NJ, bool OWN, bool BF, bool BUF>
__global__ __launch_bounds__(kBlock) void key_addr_kernel(KeyAddrArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D, Nm = a.Nm;
    const int lpr = 1 << a.lpr_log2, rpw = kWave >> a.lpr_log2;
    const int g = lane >> a.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    const int slot0 = a.w ? 1 : 0;
    const int nhop = a.P > 0 ? a.P : 1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.E), 0, BUF ? (int)a.table_bytes : 0, 0x00020000);
    const unsigned c16 = (unsigned)c * 16u;
    const unsigned emax = (unsigned)(a.n_entity > 0 ? a.n_entity - 1 : 0x7fffffff);      // last row of E
    auto row4 = [&](int id) -> float4 {
        id = (int)min((unsigned)id, emax);                   // device-resident ids are clamped into the table
        if constexpr (BUF) {
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)id * (unsigned)(D * 4) + c16, 0, 0);
            return make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]),
                               __uint_as_float(raw[3]));
        } else {
            return load_row4(a.E, BF, id, D, c);
        }
    };

    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < a.B; b += (int64_t)gridDim.x * 4) {
        for (int hop = 0; hop < nhop; ++hop) {
            const bool do_hop = hop < a.P;
            const bool do_set = hop == 0 && a.w != nullptr;
            if (!do_hop && !do_set) continue;
            const char* vb = reinterpret_cast<const char*>(a.V + b * a.nR * (int64_t)D);   // wave-uniform base
            const KeyAddrLists lists = key_addr_lists(a, b, hop);
            const int32_t* mh = lists.h;
            const int32_t* mr = do_hop ? lists.r : nullptr;
            const int32_t* mt = do_hop ? lists.t : nullptr;
            // ids: one coalesced load per list, then distributed to the row groups by bpermute
            const int lm = lane < Nm ? lane : Nm - 1;
            const int idh = mh[lm];
            const int idt = do_hop ? mt[lm] : 0;
            const int idr = do_hop ? mr[lm] : 0;
            int hid[NJ], tix[NJ], rid[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int m = j * rpw + g;
                if (Nm <= kWave) {
                    m = m < Nm ? m : Nm - 1;
                    hid[j] = __shfl(idh, m, kWave);
                    tix[j] = __shfl(idt, m, kWave);
                    rid[j] = __shfl(idr, m, kWave);
                } else {  // more memories than lanes: per-row loads
                    const bool v = m < Nm;
                    hid[j] = v ? mh[m] : 0;
                    tix[j] = (v && do_hop) ? mt[m] : 0;
                    rid[j] = (v && do_hop) ? mr[m] : 0;
                }
            }
            float4 hrow[NJ], trow[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                hrow[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cact && j * rpw + g < Nm) hrow[j] = row4(hid[j]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                trow[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (do_hop && cact && j * rpw + g < Nm)
                    trow[j] = row4(tix[j]);
            }
            const float4 wv = (do_set && cact) ? reinterpret_cast<const float4*>(a.w)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 acc_s = make_float4(0.f, 0.f, 0.f, 0.f), acc_h = make_float4(0.f, 0.f, 0.f, 0.f);
            float zh = 1.f, zs = 1.f;
            if constexpr (OWN) {
                // ---- logits; lane c of a row group keeps the logit of row j == c ----
                float oh = -INFINITY, os = -INFINITY;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float ph = 0.f, ps = 0.f;
                    if (do_hop && cact) {
                        const float4 v = *reinterpret_cast<const float4*>(vb + ((unsigned)rid[j] * (unsigned)(D * 4) + c16));
                        ph = dot4(hrow[j], v);
                    }
                    if (do_set) ps = dot4(hrow[j], wv);
                    if (do_hop) ph = group_sum(ph, a.lpr_log2);   // DPP: no LDS round trips
                    if (do_set) ps = group_sum(ps, a.lpr_log2);
                    const bool mine = (c == j) && (j * rpw + g < Nm);
                    oh = mine ? ph : oh;
                    os = mine ? ps : os;
                }
                // ---- softmax over the Nm memories (tf.nn.softmax, model.py:189 / :223): ONE exp
                // per lane; the un-normalised weights return to the group's lanes by bpermute and
                // 1/sum is applied once to the accumulated row sum ----
                const float mxh = wave_max(oh), mxs = wave_max(os);
                const bool own = oh != -INFINITY || os != -INFINITY;
                const float xh = (own && do_hop) ? expf(oh - mxh) : 0.f;
                const float xs = (own && do_set) ? expf(os - mxs) : 0.f;
                zh = wave_sum(xh);
                zs = wave_sum(xs);
                const int gbase = lane & ~(lpr - 1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {   // weighted sums (model.py:195 / :229)
                    if (do_hop) acc_h = f4_fma(__shfl(xh, gbase | j, kWave), trow[j], acc_h);
                    if (do_set) acc_s = f4_fma(__shfl(xs, gbase | j, kWave), hrow[j], acc_s);
                }
            } else {
                // more rows per lane than lanes per row (Nm > 64): every lane exponentiates its rows
                float sh[NJ], ss[NJ];
                float mxh = -INFINITY, mxs = -INFINITY;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float ph = 0.f, ps = 0.f;
                    if (do_hop && cact) {
                        const float4 v = *reinterpret_cast<const float4*>(vb + ((unsigned)rid[j] * (unsigned)(D * 4) + c16));
                        ph = dot4(hrow[j], v);
                    }
                    if (do_set) ps = dot4(hrow[j], wv);
                    ph = group_sum(ph, a.lpr_log2);
                    ps = group_sum(ps, a.lpr_log2);
                    const bool v = j * rpw + g < Nm;
                    sh[j] = v ? ph : -INFINITY;
                    ss[j] = v ? ps : -INFINITY;
                    mxh = fmaxf(mxh, sh[j]);
                    mxs = fmaxf(mxs, ss[j]);
                }
                for (int o = lpr; o < kWave; o <<= 1) {
                    mxh = fmaxf(mxh, __shfl_xor(mxh, o, kWave));
                    mxs = fmaxf(mxs, __shfl_xor(mxs, o, kWave));
                }
                zh = zs = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bool v = j * rpw + g < Nm;
                    const float eh = (v && do_hop) ? expf(sh[j] - mxh) : 0.f;
                    const float es = (v && do_set) ? expf(ss[j] - mxs) : 0.f;
                    zh += eh;
                    zs += es;
                    acc_h = f4_fma(eh, trow[j], acc_h);
                    acc_s = f4_fma(es, hrow[j], acc_s);
                }
                for (int o = lpr; o < kWave; o <<= 1) {
                    zh += __shfl_xor(zh, o, kWave);
                    zs += __shfl_xor(zs, o, kWave);
                }
            }
            if (do_set) {
                acc_s = group_xor_sum(acc_s, lpr);
                const float inv = 1.f / zs;
                if (cact && g == 0)
                    *reinterpret_cast<float4*>(a.out + b * a.ldo + (c << 2)) =
                        make_float4(acc_s.x * inv, acc_s.y * inv, acc_s.z * inv, acc_s.w * inv);
            }
            if (do_hop) {
                acc_h = group_xor_sum(acc_h, lpr);
                const float inv = 1.f / zh;
                if (cact && g == 0)
                    *reinterpret_cast<float4*>(a.out + b * a.ldo + (int64_t)(slot0 + hop) * D + (c << 2)) =
                        make_float4(acc_h.x * inv, acc_h.y * inv, acc_h.z * inv, acc_h.w * inv);
            }
        }
    }
}

// Row softmax (tf.nn.softmax over the last axis, model.py:189 / :223) for the shared-user form of
// key addressing (mvin_amd/model.py:_key_addressing_shared): out[r, :] = softmax(x[r, :]), one wave
// per row, n <= 4096.
__global__ __launch_bounds__(kBlock) void row_softmax_kernel(const float* __restrict__ x, int64_t rows, int n,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const float* xr = x + r * n;
        float mx = -INFINITY;
        for (int i = lane; i < n; i += kWave) mx = fmaxf(mx, xr[i]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int i = lane; i < n; i += kWave) sum += expf(xr[i] - mx);
        sum = wave_sum(sum);
        for (int i = lane; i < n; i += kWave) out[r * n + i] = expf(xr[i] - mx) / sum;
    }
}

// Aggregator._mix_neighbor_vectors / _mix_neighbor_vectors_urv (aggregators.py:37-77; KGCN's mixer, defined by the reference and
// never called by MVIN) on materialised tensors:
//   s[b,n,k] = mean_d(user[b,d] * rel[b,n,k,d]) ; p = softmax_k(s) ; out[b,n,:] = mean_k(p[b,n,k] * neigh[b,n,k,:])
// With `logits` [nodes, K] given the scores are taken from there (SumAggregator_urh_matrix._mix_neighbor_vectors_urh,
// aggregators.py:118-146: the caller computes relation . urh_weights[D:2D]; the user / self terms cancel in the softmax); with
// neither logits nor rel / user every weight is 1 (_mix_neighbor_vectors_no_ur, :148-152: the plain mean).
// One wave per node (b, n): a lane owns D/64-strided columns; the K logits are wave reductions.  K <= 64.
__global__ __launch_bounds__(kBlock) void mix_urv_kernel(const float* __restrict__ neigh, const float* __restrict__ rel,
                                                         const float* __restrict__ user, const float* __restrict__ logits,
                                                         int64_t nodes, int N, int K, int D,
                                                         float* __restrict__ out, float* __restrict__ probs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < nodes; t += (int64_t)gridDim.x * 4) {
        const float* nv = neigh + t * K * D;
        float p = lane < K ? 1.f : 0.f;
        if (logits || (rel && user)) {
            float logit = -INFINITY;                     // lane k keeps the logit of child k
            if (logits) {
                if (lane < K) logit = logits[t * K + lane];
            } else {
                const float* u = user + (t / N) * D;
                const float* rv = rel + t * K * D;
                for (int k = 0; k < K; ++k) {
                    float part = 0.f;
                    for (int d = lane; d < D; d += kWave) part = fmaf(u[d], rv[(int64_t)k * D + d], part);
                    part = wave_sum(part) / (float)D;    // tf.reduce_mean(user * relation, axis=-1)
                    logit = lane == k ? part : logit;
                }
            }
            const float m = wave_max(logit);
            const float e = lane < K ? expf(logit - m) : 0.f;
            p = e / wave_sum(e);
        }
        if (probs && lane < K) probs[t * K + lane] = p;
        for (int d0 = 0; d0 < D; d0 += kWave) {          // (wave-uniform trip count: the broadcasts below need every lane awake)
            const int d = d0 + lane;
            const bool live = d < D;
            float acc = 0.f;
            for (int k = 0; k < K; ++k) {
                const float pk = __shfl(p, k, kWave);
                const float x = nv[(int64_t)k * D + (live ? d : 0)];
                acc = fmaf(pk, live ? x : 0.f, acc);
            }
            if (live) out[t * D + d] = acc / (float)K;   // tf.reduce_mean(p * neighbor_vectors, axis=2)
        }
    }
}

hipError_t launch_mix_urv(const float* neigh, const float* rel, const float* user, const float* logits, int64_t nodes, int N, int K,
                          int D, float* out, float* probs, hipStream_t st) {
    const int64_t nblk = (nodes + 3) / 4;
    const int64_t cap = 256 * 16;
    mix_urv_kernel<<<(int)(nblk < cap ? nblk : cap), kBlock, 0, st>>>(neigh, rel, user, logits, nodes, N, K, D, out, probs);
    return hipGetLastError();
}

hipError_t launch_row_softmax(const float* x, int64_t rows, int n, float* out, hipStream_t st) {
    const int64_t nblk = (rows + 3) / 4;
    const int64_t cap = 256 * 16;
    row_softmax_kernel<<<(int)(nblk < cap ? nblk : cap), kBlock, 0, st>>>(x, rows, n, out);
    return hipGetLastError();
}

int key_addr_nj(int Nm, int D) {
    const int rpw = kWave >> lpr_log2_for(D);
    const int need = (Nm + rpw - 1) / rpw;
    int nj = 1;
    while (nj < need) nj *= 2;
    return nj;  // > 16 means: not supported by the register-resident kernel
}

hipError_t launch_key_addr(const KeyAddrArgs& a, int table_bf16, hipStream_t st) {
    // per-pair ripple sets with P >= 1 hops, Nm <= 64 and a small V block: the streaming (LDS-DMA) pipeline;
    // MVIN_KA_STREAM=0 keeps the register-resident kernel below for A/B measurements
    if (key_addr_stream_supported(a, table_bf16)) return launch_key_addr_stream(a, table_bf16, st);
    const int nj = key_addr_nj(a.Nm, a.D);
    const int64_t nblk = (a.B + 3) / 4;
    const int64_t cap = 256 * 8;
    const int grid = (int)(nblk < cap ? nblk : cap);
    const bool own = nj <= (1 << a.lpr_log2);   // rows per lane <= lanes per row: one exp per lane
    const bool buf = !table_bf16 && a.table_bytes > 0 && a.table_bytes < (1ull << 32);
#define MVIN_KA2(NJV, OWNV)                                                                        \
    if (table_bf16) key_addr_kernel<NJV, OWNV, true, false><<<grid, kBlock, 0, st>>>(a);            \
    else if (buf) key_addr_kernel<NJV, OWNV, false, true><<<grid, kBlock, 0, st>>>(a);              \
    else key_addr_kernel<NJV, OWNV, false, false><<<grid, kBlock, 0, st>>>(a);
#define MVIN_KA(NJV)                          \
    if (own) { MVIN_KA2(NJV, true) }          \
    else { MVIN_KA2(NJV, false) }             \
    break;
    switch (nj) {
        case 1: MVIN_KA(1)
        case 2: MVIN_KA(2)
        case 4: MVIN_KA(4)
        case 8: MVIN_KA(8)
        case 16: MVIN_KA(16)
        default: return hipErrorInvalidValue;
    }
#undef MVIN_KA
#undef MVIN_KA2
    return hipGetLastError();
}

}  // namespace mvin
