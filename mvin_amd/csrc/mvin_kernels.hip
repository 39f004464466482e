// MVIN scoring path -- gfx950 (MI355X) kernels, generic VALU versions.
//
// Every kernel cites the reference op sequence it replaces (paths relative to
// johnnyjana730/MVIN src/model/MVIN/).  Layout: all tensors dense row-major fp32; ids int32.
// Work decomposition: a "node task" t = (pair b, node n) owns the K children of one tree
// node; a wave64 gathers the K child rows of a task with 16-byte loads (D/4 lanes per row,
// 64/(D/4) rows per wave-instruction), the softmax over K is a wave-level reduction, and the
// small dense epilogues run on tiles of kTM = 32 tasks staged in LDS.
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

// --------------------------------------------------------------------------------------
// model.py:243-256  MVIN.get_neighbors
// --------------------------------------------------------------------------------------
__global__ void expand_level0_kernel(const int64_t* __restrict__ items64,
                                     const int32_t* __restrict__ items32, int B, int n_entity,
                                     int32_t* __restrict__ ent0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    long long v = items64 ? (long long)items64[b] : (long long)items32[b];
    // The reference's tf.gather raises on out-of-range ids; a GPU fault would kill the
    // process instead, so clamp (the Python boundary validates host inputs).
    v = v < 0 ? 0 : (v >= n_entity ? n_entity - 1 : v);
    ent0[b] = (int32_t)v;
}

__global__ void expand_level_kernel(const int32_t* __restrict__ adj_e,
                                    const int32_t* __restrict__ adj_r,
                                    const int32_t* __restrict__ parent, int64_t n_out, int K,
                                    int32_t* __restrict__ ent_next, int32_t* __restrict__ rel_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_out; idx += stride) {
        const int64_t p = idx / K;
        const int k = (int)(idx - p * K);
        const int64_t x = parent[p];
        ent_next[idx] = adj_e[x * K + k];
        rel_out[idx] = adj_r[x * K + k];
    }
}

// --------------------------------------------------------------------------------------
// aggregators.py:130-133 restricted to the k-dependent term: t[r] = Rel[r,:] . urh_w[D:2D]
// --------------------------------------------------------------------------------------
__global__ void rel_score_kernel(const float* __restrict__ rel, const float* __restrict__ urh_w,
                                 int nR, int D, float* __restrict__ t) {
    const int r = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    if (r >= nR) return;
    float s = 0.f;
    for (int d = lane; d < D; d += kWave) s = fmaf(rel[(size_t)r * D + d], urh_w[D + d], s);
    s = wave_sum(s);
    if (lane == 0) t[r] = s;
}

// --------------------------------------------------------------------------------------
// rows x small dense (tf.matmul sites: model.py:279, :312, :234; per-relation item
// projection of :214-220) with optional gathered sources (tf.nn.embedding_lookup) and the
// fused score epilogue of model.py:158-159.
// --------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(kBlock) void linear_kernel(mvin_linear_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RP = kTM / NR;
    const int Din = a.sum_sources ? a.Dsrc : a.nsrc * a.Dsrc;
    const int ldx = Din + 4;
    float* sX = smem;
    const int tid = threadIdx.x;
    const int rg = tid / a.Dout;
    const int j = tid - rg * a.Dout;
    const bool active = rg < RP;
    const int z = blockIdx.y;
    const float* W = a.W ? a.W + (size_t)z * a.w_zstride : nullptr;
    const float* bias = a.bias ? a.bias + (size_t)z * a.bias_zstride : nullptr;
    float* out = a.out + (size_t)z * a.out_zstride;
    const int c4 = a.Dsrc >> 2;
    const int64_t ntiles = (a.rows + kTM - 1) / kTM;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * kTM;
        for (int s = 0; s < a.nsrc; ++s) {
            const float* src = a.src[s];
            const int32_t* ids = a.ids[s];
            for (int idx = tid; idx < kTM * c4; idx += kBlock) {
                const int row = idx / c4, c = idx - row * c4;
                const int64_t r = r0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < a.rows) {
                    int64_t srow = !ids ? r : (a.ids64 ? reinterpret_cast<const int64_t*>(ids)[r] : (int64_t)ids[r]);
                    if (ids && a.src_rows > 0) srow = (int64_t)min((uint64_t)srow, (uint64_t)(a.src_rows - 1));   // clamped into the table
                    v = load_row4(src, (a.src_bf16 >> s) & 1, srow, a.Dsrc, c);
                }
                float4* dst = reinterpret_cast<float4*>(sX + row * ldx + (a.sum_sources ? 0 : s * a.Dsrc) + c * 4);
                if (a.sum_sources && s > 0) {  // same thread wrote this slot for s-1
                    const float4 o = *dst;
                    v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                }
                *dst = v;
            }
        }
        __syncthreads();
        float acc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[i] = 0.f;
        if (active) {
            if (W) {
                tile_matvec<NR>(sX, ldx, Din, W, a.Dout, j, rg, acc);
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) acc[i] = sX[(rg + RP * i) * ldx + j];
            }
            const float bj = bias ? bias[j] : 0.f;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int64_t r = r0 + rg + RP * i;
                if (r < a.rows) {
                    float v = acc[i] + bj;
                    if (a.rowbias) v += a.rowbias[(r / a.rows_per_group) * a.Dout + j];
                    if (a.relu) v = fmaxf(v, 0.f);
                    acc[i] = v;
                    out[r * a.ldo + j] = v;
                }
            }
        }
        if (a.score_u) {
            __syncthreads();  // everyone is done reading sX
            if (active) {
#pragma unroll
                for (int i = 0; i < NR; ++i) sX[(rg + RP * i) * ldx + j] = acc[i];
            }
            __syncthreads();
            if (tid < kTM) {
                const int64_t r = r0 + tid;
                if (r < a.rows) {
                    float s = 0.f;
                    for (int d = 0; d < a.Dout; ++d)
                        s = fmaf(sX[tid * ldx + d], a.score_u[r * a.Dout + d], s);
                    if (a.score_out) a.score_out[r] = s;
                    if (a.sigmoid_out) a.sigmoid_out[r] = 1.f / (1.f + expf(-s));
                }
            }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------
// SumAggregator_urh_matrix (aggregators.py:98-152) fused with the child lookups of
// MVIN.aggregate_delta_whole (model.py:267-268, :295-305).
//   gather mode : children = table rows addressed through the adjacency of node_ids[t]
//                 (the deepest hop; K^L rows per pair never materialised)
//   dense mode  : children = neigh[t*K + k, :] (materialised upper levels)
// Phase A (per wave, one task at a time): ids -> softmax over K (wave reduction) ->
//   weighted sum of the K child rows (16-byte loads, D/4 lanes per row) -> S in LDS.
// Phase B (tile of kTM tasks): Z = self + (S.Wc + psum*c_child[b]) / K   (or self + S/K).
// Phase C: out = relu(Z.Wagg + bagg).
// --------------------------------------------------------------------------------------
// PRE: child-row loads of the first PRE steps issued ahead of the softmax (latency-bound launches of few tiles:
// 43 -> 36 us at 512 nodes, 52 -> 43 at 16 384; the 24 extra VGPRs cost occupancy when the grid fills the chip:
// 228 -> 311 us at 131 072 nodes, hence PRE = 0 there)
template <int NR, int PRE>
__global__ __launch_bounds__(kBlock) void gather_attn_kernel(GatherAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RP = kTM / NR;
    const int D = a.D, K = a.K;
    const int ldx = D + 4;
    float* sS = smem;                 // [kTM][ldx]
    float* sZ = sS + kTM * ldx;       // [kTM][ldx]
    int2* sYP = reinterpret_cast<int2*>(sZ + kTM * ldx);  // [4 waves][K]  {child row id, p bits}

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    const int lpr = 1 << a.lpr_log2;          // lanes per row (power of two >= D/4)
    const int rpw = kWave >> a.lpr_log2;      // rows per wave-instruction
    const int g = lane >> a.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    int2* yp = sYP + wave * K;

    const int rg = tid / D;
    const int j = tid - rg * D;
    const bool active = rg < RP;
    const float invK_den = (float)K;
    const float psum = a.rel_score ? 1.f : (float)K;

    // TR node tasks per tile (rows TR.. of the LDS tiles are unused: computed on, never stored): launches of a few
    // hundred nodes spread over 4x the workgroups, a wave walks 2 tasks instead of 8
    const int TR = a.tile_rows;
    const int64_t ntiles = (a.T + TR - 1) / TR;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t t0 = tile * TR;
        // ---------------- phase A ----------------
        for (int tt = 0; tt < TR / 4; ++tt) {
            const int trow = wave * (TR / 4) + tt;
            const int64_t t = t0 + trow;
            if (t >= a.T) {
                if (cact && g == 0)
                    *reinterpret_cast<float4*>(sS + trow * ldx + (c << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            int64_t xbase = 0;
            if (a.gather) xbase = (int64_t)a.node_ids[t] * K;
            // ids + logits
            float mx = -INFINITY;
            for (int k = lane; k < K; k += kWave) {
                int y = 0;        // dense rows are addressed by t*K + k below
                int64_t r = 0;    // index into rel_score: relation id, or child index (per-child logits)
                if (a.gather) {
                    y = a.adj_e[xbase + k];
                    if (a.rel_score) r = a.adj_r[xbase + k];
                } else if (a.rel_score) {
                    r = a.rel_ids ? (int64_t)a.rel_ids[t * K + k] : t * K + k;
                }
                yp[k].x = y;      // the id first: the row loads below need it, not the score that is still in flight
                const float sc = a.rel_score ? a.rel_score[r] : 0.f;
                yp[k].y = __float_as_int(sc);
                mx = fmaxf(mx, sc);
            }
            // the child rows do not depend on the softmax: the first kPre steps' loads are issued BEFORE the three
            // wave reductions (ids -> {scores -> softmax | rows} instead of ids -> scores -> softmax -> rows)
            constexpr int kPre = PRE;
            float4 pre[PRE > 0 ? PRE : 1];
            const float* dbase = a.gather ? nullptr : a.neigh + t * K * (int64_t)D;
#pragma unroll
            for (int i = 0; i < kPre; ++i) {
                const int k = g + i * rpw;
                pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K && cact)
                    pre[i] = a.gather ? load_row4(a.table, a.table_bf16, yp[k].x, D, c)
                                      : reinterpret_cast<const float4*>(dbase + (int64_t)k * D)[c];
            }
            mx = wave_max(mx);
            float sum = 0.f;
            for (int k = lane; k < K; k += kWave) {
                const float e = a.rel_score ? expf(__int_as_float(yp[k].y) - mx) : 1.f;
                yp[k].y = __float_as_int(e);
                sum += e;
            }
            sum = wave_sum(sum);
            for (int k = lane; k < K; k += kWave) {
                const float p = a.rel_score ? __int_as_float(yp[k].y) / sum : 1.f;
                yp[k].y = __float_as_int(p);
                if (a.probs) a.probs[t * K + k] = p;
            }
            // weighted sum of the K child rows
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < kPre; ++i) {
                const int k = g + i * rpw;
                if (k < K) acc = f4_fma(__int_as_float(yp[k].y), pre[i], acc);
            }
            if (a.gather) {
#pragma unroll 8
                for (int k = g + kPre * rpw; k < K; k += rpw) {
                    const int2 e = yp[k];
                    if (cact) {
                        const float4 v = load_row4(a.table, a.table_bf16, e.x, D, c);
                        acc = f4_fma(__int_as_float(e.y), v, acc);
                    }
                }
            } else {
#pragma unroll 8
                for (int k = g + kPre * rpw; k < K; k += rpw) {
                    const float p = __int_as_float(yp[k].y);
                    if (cact) {
                        const float4 v = reinterpret_cast<const float4*>(dbase + (int64_t)k * D)[c];
                        acc = f4_fma(p, v, acc);
                    }
                }
            }
            acc = group_xor_sum(acc, lpr);
            if (cact && g == 0) {
                *reinterpret_cast<float4*>(sS + trow * ldx + (c << 2)) = acc;
                if (a.s_out) {
                    const float ik = 1.f / (float)K;
                    *reinterpret_cast<float4*>(a.s_out + t * D + (c << 2)) =
                        make_float4(acc.x * ik, acc.y * ik, acc.z * ik, acc.w * ik);
                }
            }
        }
        __syncthreads();
        // ---------------- phase B ----------------
        float acc[NR];
        if (active) {
            if (a.Wc) {
#pragma unroll
                for (int i = 0; i < NR; ++i) acc[i] = 0.f;
                tile_matvec<NR>(sS, ldx, D, a.Wc, D, j, rg, acc);
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) acc[i] = sS[(rg + RP * i) * ldx + j];
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = rg + RP * i;
                const int64_t t = t0 + row;
                float zv = 0.f;
                if (t < a.T && row < TR) {
                    float v = acc[i];
                    if (a.Wc && a.c_child) v = fmaf(psum, a.c_child[(t / a.N) * D + j], v);
                    zv = a.self_vec[t * D + j] + v / invK_den;
                    if (a.z_out) a.z_out[t * D + j] = zv;
                }
                sZ[row * ldx + j] = zv;
            }
        }
        __syncthreads();
        // ---------------- phase C ----------------
        if (active) {
#pragma unroll
            for (int i = 0; i < NR; ++i) acc[i] = 0.f;
            tile_matvec<NR>(sZ, ldx, D, a.Wagg, D, j, rg, acc);
            const float bj = a.bagg ? a.bagg[j] : 0.f;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int64_t t = t0 + rg + RP * i;
                if (t < a.T && rg + RP * i < TR) a.out[t * D + j] = fmaxf(acc[i] + bj, 0.f);
            }
        }
        // next tile's phase A writes sS/sYP only; its phase B (after a barrier) writes sZ.
    }
}

// --------------------------------------------------------------------------------------
// MVIN._key_addressing attention reads (model.py:162-197 and :210-230).  One wave per pair.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ripple_attn_kernel(RippleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    const int D = a.D, Nm = a.Nm;
    const int lpr = 1 << a.lpr_log2;
    const int rpw = kWave >> a.lpr_log2;
    const int g = lane >> a.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    float* sc = smem + wave * Nm;

    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < a.B; b += (int64_t)gridDim.x * 4) {
        const int32_t* sid = a.score_ids + b * Nm;
        const int32_t* vid = a.value_ids + b * Nm;
        // pass 1: logits
        for (int m0 = 0; m0 < Nm; m0 += rpw) {
            const int m = m0 + g;
            const bool valid = m < Nm;
            float part = 0.f;
            if (valid && cact) {
                const float4 h = load_row4(a.E, a.table_bf16, sid[m], D, c);
                const float* vp = a.mode == 0 ? a.V + ((b * a.nR + a.rel_ids[b * Nm + m]) * (int64_t)D) : a.w;
                const float4 v = reinterpret_cast<const float4*>(vp)[c];
                part = fmaf(h.x, v.x, fmaf(h.y, v.y, fmaf(h.z, v.z, h.w * v.w)));
            }
            for (int o = 1; o < lpr; o <<= 1) part += __shfl_xor(part, o, kWave);
            if (valid && c == 0) sc[m] = part;
        }
        // softmax over the Nm memories (tf.nn.softmax, model.py:189 / :223)
        float mx = -INFINITY;
        for (int m = lane; m < Nm; m += kWave) mx = fmaxf(mx, sc[m]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int m = lane; m < Nm; m += kWave) {
            const float e = expf(sc[m] - mx);
            sc[m] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        // pass 2: weighted sum of the value rows (model.py:195 / :229)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int m = g; m < Nm; m += rpw) {
            const float p = sc[m] / sum;
            if (cact) {
                const float4 v = load_row4(a.E, a.table_bf16, vid[m], D, c);
                acc = f4_fma(p, v, acc);
            }
        }
        acc = group_xor_sum(acc, lpr);
        if (cact && g == 0) *reinterpret_cast<float4*>(a.out + b * a.ldo + (c << 2)) = acc;
    }
}

// --------------------------------------------------------------------------------------
// launch helpers
// --------------------------------------------------------------------------------------
static int nr_for(int Dout) {
    int rp = kBlock / Dout;      // rows that fit side by side
    int p = 1;
    while (p * 2 <= rp && p * 2 <= kTM) p *= 2;
    return kTM / p;              // NR in {1,2,4,8,16,32}
}

static int grid_for(int64_t ntiles) {
    const int64_t cap = 256 * 8;  // 256 CUs x 8 workgroups
    return (int)(ntiles < cap ? (ntiles < 1 ? 1 : ntiles) : cap);
}

template <typename KernelT>
static hipError_t ensure_lds(KernelT kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define MVIN_DISPATCH_NR(nr, CALL)            \
    switch (nr) {                             \
        case 1: { CALL(1); break; }           \
        case 2: { CALL(2); break; }           \
        case 4: { CALL(4); break; }           \
        case 8: { CALL(8); break; }           \
        case 16: { CALL(16); break; }         \
        default: { CALL(32); break; }         \
    }

hipError_t launch_expand(const int32_t* adj_e, const int32_t* adj_r, const int64_t* items64,
                         const int32_t* items32, int B, int K, int levels, int n_entity,
                         int32_t* ent_out, int32_t* rel_out, hipStream_t st) {
    expand_level0_kernel<<<(B + 255) / 256, 256, 0, st>>>(items64, items32, B, n_entity, ent_out);
    int64_t n_parent = B;
    int32_t* parent = ent_out;
    int32_t* rel = rel_out;
    for (int e = 0; e < levels; ++e) {
        const int64_t n_out = n_parent * K;
        int32_t* next = parent + n_parent;
        const int64_t blocks = (n_out + 255) / 256;
        expand_level_kernel<<<(int)(blocks < 65536 ? blocks : 65536), 256, 0, st>>>(
            adj_e, adj_r, parent, n_out, K, next, rel);
        rel += n_out;
        parent = next;
        n_parent = n_out;
    }
    return hipGetLastError();
}

hipError_t launch_rel_score(const float* rel, const float* urh_w, int nR, int D, float* t,
                            hipStream_t st) {
    rel_score_kernel<<<(nR + 3) / 4, kBlock, 0, st>>>(rel, urh_w, nR, D, t);
    return hipGetLastError();
}

hipError_t launch_linear(const mvin_linear_args& a, hipStream_t st) {
    static const bool no_mfma = getenv("MVIN_LINEAR_VALU") != nullptr;
    if (!no_mfma && linear_mfma_supported(a)) return launch_linear_mfma(a, st);
    const int nr = nr_for(a.Dout);
    const size_t lds = (size_t)kTM * ((a.sum_sources ? 1 : a.nsrc) * a.Dsrc + 4) * sizeof(float);
    const int64_t ntiles = (a.rows + kTM - 1) / kTM;
    dim3 grid(grid_for(ntiles), a.nz > 0 ? a.nz : 1);
#define CALL(NRV)                                                         \
    {                                                                     \
        hipError_t e = ensure_lds(linear_kernel<NRV>, lds);               \
        if (e != hipSuccess) return e;                                    \
        linear_kernel<NRV><<<grid, kBlock, lds, st>>>(a);                 \
    }
    MVIN_DISPATCH_NR(nr, CALL)
#undef CALL
    return hipGetLastError();
}

hipError_t launch_gather_attn(const GatherAttnArgs& a0, hipStream_t st) {
    const int nr = nr_for(a0.D);
    GatherAttnArgs a = a0;
    const size_t lds = (size_t)2 * kTM * (a.D + 4) * sizeof(float) + (size_t)4 * a.K * sizeof(int2);
    a.tile_rows = (a.T + kTM - 1) / kTM < 256 ? 8 : kTM;
    const int64_t ntiles = (a.T + a.tile_rows - 1) / a.tile_rows;
    dim3 grid(grid_for(ntiles));
    const bool pre = ntiles <= 1024;
#define CALL(NRV)                                                         \
    {                                                                     \
        hipError_t e = pre ? ensure_lds(gather_attn_kernel<NRV, 8>, lds)  \
                           : ensure_lds(gather_attn_kernel<NRV, 0>, lds); \
        if (e != hipSuccess) return e;                                    \
        if (pre) gather_attn_kernel<NRV, 8><<<grid, kBlock, lds, st>>>(a);\
        else gather_attn_kernel<NRV, 0><<<grid, kBlock, lds, st>>>(a);    \
    }
    MVIN_DISPATCH_NR(nr, CALL)
#undef CALL
    return hipGetLastError();
}

hipError_t launch_ripple(const RippleArgs& a, hipStream_t st) {
    const size_t lds = (size_t)4 * a.Nm * sizeof(float);
    const int64_t nblk = (a.B + 3) / 4;
    ripple_attn_kernel<<<grid_for(nblk), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

// ---- raw row movers (multi-GPU exchange): 4-byte words, one row per lane group ----
template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void move_rows_kernel(uint32_t* __restrict__ table, const int32_t* __restrict__ ids,
                                                           int64_t n, int words, uint32_t* __restrict__ rows) {
    const int64_t total = n * words;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / words;
        const int w = (int)(i - r * words);
        const int64_t t = (int64_t)ids[r] * words + w;
        if (SCATTER) table[t] = rows[i];
        else rows[i] = table[t];
    }
}

hipError_t launch_move_rows(void* table, const int32_t* ids, int64_t n, int row_bytes, void* rows, bool scatter,
                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int words = row_bytes / 4;
    const int64_t nblk = (n * words + kBlock - 1) / kBlock;
    const int grid = (int)(nblk < 256 * 32 ? nblk : 256 * 32);
    if (scatter) move_rows_kernel<true><<<grid, kBlock, 0, st>>>((uint32_t*)table, ids, n, words, (uint32_t*)rows);
    else move_rows_kernel<false><<<grid, kBlock, 0, st>>>((uint32_t*)table, ids, n, words, (uint32_t*)rows);
    return hipGetLastError();
}


// ---- entity ids -> shard space (multi-GPU layer, mvin_amd/dist.py): pi(x) = (x mod W) * n_local + x div W ----
template <typename T>
__global__ __launch_bounds__(kBlock) void shard_space_ids_kernel(const T* __restrict__ ids, int64_t n, int world, int n_local,
                                                                 T* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t x = (int64_t)ids[i];
        const int64_t q = x / world;
        out[i] = (T)((x - q * world) * n_local + q);
    }
}

hipError_t launch_shard_space_ids(const void* ids, bool is64, int64_t n, int world, int n_local, void* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t nblk = (n + kBlock - 1) / kBlock;
    const int grid = (int)(nblk < 256 * 8 ? nblk : 256 * 8);
    if (is64) shard_space_ids_kernel<int64_t><<<grid, kBlock, 0, st>>>((const int64_t*)ids, n, world, n_local, (int64_t*)out);
    else shard_space_ids_kernel<int32_t><<<grid, kBlock, 0, st>>>((const int32_t*)ids, n, world, n_local, (int32_t*)out);
    return hipGetLastError();
}

}  // namespace mvin
