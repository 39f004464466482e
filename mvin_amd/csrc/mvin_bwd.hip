// Backward kernels of the MVIN path (scope row f-2): gradients of the loss of
// src/model/MVIN/model.py:378-412 with respect to every table and weight, so that MVIN.train
// (model.py:416-417: forward + backward + Adam in one sess.run) runs on the GPU.
// They mirror the forward kernels: what the forward gathers, the backward scatter-adds
// (atomic float adds on table rows); softmax backward is a wave-level reduction; weight
// gradients are row-reductions of outer products.  Correctness-first versions.
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

// ------------------------------------------------------------------------------------------
// element-wise family
// ------------------------------------------------------------------------------------------
// *accum += sum over the block of `local`: ONE atomic per workgroup (a per-wave atomic on a single
// address serialises: 32 k atomics cost ~100 us)
__device__ __forceinline__ void block_accumulate(float* accum, float local) {
    __shared__ float s_part[16];
    local = wave_sum(local);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_part[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < nw; ++i) t += s_part[i];
        if (t != 0.f) atomicAdd(accum, t);
    }
}

__global__ void eltwise_kernel(EltArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float local = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        switch (a.mode) {
            case 0:  // y = alpha * x + beta * y
                a.y[i] = a.alpha * a.x[i] + (a.beta != 0.f ? a.beta * a.y[i] : 0.f);
                break;
            case 1: {  // sigmoid cross entropy (tf.nn.sigmoid_cross_entropy_with_logits, model.py:379):
                       // y = (sigmoid(s) - label) * alpha ; loss += max(s,0) - s*label + log1p(exp(-|s|))
                const float s = a.x[i], lab = a.z[i];
                a.y[i] = (1.f / (1.f + expf(-s)) - lab) * a.alpha;
                local += (fmaxf(s, 0.f) - s * lab + log1pf(expf(-fabsf(s)))) * a.beta;
                break;
            }
            case 2:  // relu backward: y = z > 0 ? x : 0   (z = forward output)
                a.y[i] = a.z[i] > 0.f ? a.x[i] : 0.f;
                break;
            case 3:  // sum of squares: accum += alpha * x^2
                local += a.alpha * a.x[i] * a.x[i];
                break;
            case 4: {  // Adam (tf.train.AdamOptimizer, model.py:414): x = param, y = grad, z = m, w = v
                const float g = a.y[i];
                const float m = a.beta1 * a.z[i] + (1.f - a.beta1) * g;
                const float v = a.beta2 * a.w[i] + (1.f - a.beta2) * g * g;
                a.z[i] = m;
                a.w[i] = v;
                a.x[i] -= a.alpha * m / (sqrtf(v) + a.eps);
                break;
            }
            case 5: {  // row scale: y[r, :] = beta * y[r, :] + alpha * z[r] * x[r, :]   (n = rows * D)
                const int64_t r = i / a.D;
                a.y[i] = (a.beta != 0.f ? a.beta * a.y[i] : 0.f) + a.alpha * a.z[r] * a.x[i];
                break;
            }
            case 7: {  // row-weighted sum of squares: accum += alpha * z[r] * x[r, :]^2   (n = rows * D)
                const float xv = a.x[i];
                local += a.alpha * a.z[i / a.D] * xv * xv;
                break;
            }
            case 8: {  // sum of squares of GATHERED rows: accum += alpha * sum_r x[ids[r], :]^2, ids = (int32*) z
                       // (the regulariser of model.py:383-385 on the ripple-set rows, without materialising them)
                const int64_t r = i / a.D;
                const int64_t row = reinterpret_cast<const int32_t*>(a.z)[r];
                const float xv = a.x[row * a.D + (i - r * a.D)];
                local += a.alpha * xv * xv;
                break;
            }
            case 6: {  // group row sum: y[g, j] = alpha * sum_{n < N} x[(g*N + n), j]   (n = groups * D)
                const int64_t gidx = i / a.D;
                const int j = (int)(i - gidx * a.D);
                const float* col = a.x + gidx * a.N * (int64_t)a.D + j;
                float sgm = 0.f;
                int q = 0;
                for (; q + 8 <= a.N; q += 8) {          // eight independent loads in flight, summed in the same order
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = col[(int64_t)(q + e) * a.D];
#pragma unroll
                    for (int e = 0; e < 8; ++e) sgm += v[e];
                }
                for (; q < a.N; ++q) sgm += col[(int64_t)q * a.D];
                a.y[i] = a.alpha * sgm;
                break;
            }
        }
    }
    if (a.accum) block_accumulate(a.accum, local);
}

// ------------------------------------------------------------------------------------------
// out[b] += number of i with ids[i] == b  (bins <= 4096: per-workgroup LDS histogram, one global atomic per bin and
// workgroup; 262 k atomics straight onto 9 global addresses took 62 us)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_ids_kernel(const int32_t* __restrict__ ids, int64_t n, int nbins,
                                                        float* __restrict__ out) {
    __shared__ int bins[4096];
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = ids[i];
        if (b >= 0 && b < nbins) atomicAdd(&bins[b], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x)
        if (bins[i]) atomicAdd(out + i, (float)bins[i]);
}

hipError_t launch_count_ids(const int32_t* ids, int64_t n, int nbins, float* out, hipStream_t st) {
    const int64_t nb = (n + 4095) / 4096;
    count_ids_kernel<<<(int)(nb < 1 ? 1 : nb < 256 ? nb : 256), 256, 0, st>>>(ids, n, nbins, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// dTable[ids[r], :] += alpha * X[r, :]      (backward of tf.nn.embedding_lookup)
// ------------------------------------------------------------------------------------------
// One float per lane: a wave instruction adds 64 CONSECUTIVE floats (one 256-byte row at D = 64).  With a float4 chunk per
// lane the four component atomics of a wave each touched every 16-byte-strided word of four rows -- 8 cache lines per
// instruction, a quarter of each -- and the L2 atomic units work per line.
__global__ void scatter_add_rows_kernel(float* __restrict__ dtable, const int32_t* __restrict__ ids, int ids64,
                                        const float* __restrict__ x, int64_t rows, int D, float alpha) {
    const int64_t n = rows * D, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        const int64_t row = ids64 ? reinterpret_cast<const int64_t*>(ids)[r] : (int64_t)ids[r];
        atomicAdd(dtable + row * D + c, alpha * x[i]);
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient of the rows x dense operator:  dW[z] += X^T . dY[z],  db[z] += sum_r dY[z][r]
// X = concat / sum of sources exactly as mvin_linear_fwd stages them; dY optionally masked by the
// forward output (relu).  grid = (row blocks, Din blocks, nz); each thread owns up to 16 entries
// of the [IB x Dout] slab of dW and walks the block's row tiles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void linear_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Din = a.lin.sum_sources ? a.lin.Dsrc : a.lin.nsrc * a.lin.Dsrc;
    const int Dout = a.lin.Dout;
    const int ldx = Din + 4, ldy = Dout + 1;
    float* sX = smem;                 // [32][ldx]
    float* sY = sX + kTM * ldx;       // [32][ldy]
    const int tid = threadIdx.x;
    const int z = blockIdx.z;
    const int IB = a.IB;
    const int i0 = blockIdx.y * IB;
    const int nib = (Din - i0) < IB ? (Din - i0) : IB;
    const int nent = nib * Dout;      // entries of this block's dW slab
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float accb = 0.f;
    const float* dY = a.dY + (size_t)z * a.dy_zstride;
    const int c4 = a.lin.Dsrc >> 2;
    const int64_t ntiles = (a.lin.rows + kTM - 1) / kTM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * kTM;
        for (int s = 0; s < a.lin.nsrc; ++s) {
            const float* src = a.lin.src[s];
            const int32_t* ids = a.lin.ids[s];
            for (int idx = tid; idx < kTM * c4; idx += kBlock) {
                const int row = idx / c4, c = idx - row * c4;
                const int64_t r = r0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < a.lin.rows) {
                    const int64_t srow = !ids ? r : (a.lin.ids64 ? reinterpret_cast<const int64_t*>(ids)[r] : (int64_t)ids[r]);
                    v = reinterpret_cast<const float4*>(src + srow * a.lin.Dsrc)[c];
                }
                float4* dst = reinterpret_cast<float4*>(sX + row * ldx + (a.lin.sum_sources ? 0 : s * a.lin.Dsrc) + c * 4);
                if (a.lin.sum_sources && s > 0) {
                    const float4 o = *dst;
                    v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                }
                *dst = v;
            }
        }
        for (int idx = tid; idx < kTM * Dout; idx += kBlock) {
            const int row = idx / Dout, j = idx - row * Dout;
            const int64_t r = r0 + row;
            float v = 0.f;
            if (r < a.lin.rows) {
                v = dY[r * a.ldy + j];
                if (a.mask) v = a.mask[(size_t)z * a.mask_zstride + r * a.ldm + j] > 0.f ? v : 0.f;
            }
            sY[row * ldy + j] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + kBlock * q;
            if (e < nent) {
                const int i = i0 + e / Dout, j = e % Dout;
                float s = acc[q];
                for (int row = 0; row < kTM; ++row) s = fmaf(sX[row * ldx + i], sY[row * ldy + j], s);
                acc[q] = s;
            }
        }
        if (a.db && blockIdx.y == 0 && tid < Dout)
            for (int row = 0; row < kTM; ++row) accb += sY[row * ldy + tid];
        __syncthreads();
    }
    float* dW = a.dW + (size_t)z * a.dw_zstride;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + kBlock * q;
        if (e < nent && acc[q] != 0.f) atomicAdd(dW + (size_t)(i0 + e / Dout) * Dout + e % Dout, acc[q]);
    }
    if (a.db && blockIdx.y == 0 && tid < Dout && accb != 0.f) atomicAdd(a.db + (size_t)z * a.db_zstride + tid, accb);
}

// The same weight gradient on the matrix cores, for Din = TI*16 and Dout = TJ*16 with TI*TJ <= 64 (every D x D and
// concat-of-3 shape of the path at D in {16, 32, 64}): dW = X^T . dY as v_mfma_f32_16x16x4_f32 with A = X^T (a 32-row
// tile of X in LDS, read column-wise: 16 consecutive floats per row group), B = dY, the [TI x TJ] output tiles dealt to
// the four waves and kept in registers over the block's row tiles, one atomic per entry at the end.  The VALU kernel
// above reads two LDS words per fma (LDS-bound) and runs on at most 64 workgroups per weight slab.
typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int TI, int TJ>
__device__ __forceinline__ void wgrad_mfma_body(const WgradArgs& a, const int bx, const int gx, const int z) {
    constexpr int Din = TI * 16, Dout = TJ * 16, NT = TI * TJ, NTW = (NT + 3) / 4;
    constexpr int ldx = (Din % 32 == 0) ? Din + 16 : Din;      // row stride = 16 mod 32 words: the four row groups of an
    constexpr int ldy = (Dout % 32 == 0) ? Dout + 16 : Dout;   // A / B fragment read hit disjoint banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                 // [32][ldx]
    float* sY = sX + kTM * ldx;       // [32][ldy]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q16 = lane >> 4;
    f32x4w acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    const float* dY = a.dY + (size_t)z * a.dy_zstride;
    const int c4 = a.lin.Dsrc >> 2;
    const int64_t ntiles = (a.lin.rows + kTM - 1) / kTM;
    for (int64_t tile = bx; tile < ntiles; tile += gx) {
        const int64_t r0 = tile * kTM;
        for (int s = 0; s < a.lin.nsrc; ++s) {
            const float* src = a.lin.src[s];
            const int32_t* ids = a.lin.ids[s];
            for (int idx = tid; idx < kTM * c4; idx += kBlock) {
                const int row = idx / c4, c = idx - row * c4;
                const int64_t r = r0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < a.lin.rows) {
                    const int64_t srow = !ids ? r : (a.lin.ids64 ? reinterpret_cast<const int64_t*>(ids)[r] : (int64_t)ids[r]);
                    v = reinterpret_cast<const float4*>(src + srow * a.lin.Dsrc)[c];
                }
                float4* dst = reinterpret_cast<float4*>(sX + row * ldx + (a.lin.sum_sources ? 0 : s * a.lin.Dsrc) + c * 4);
                if (a.lin.sum_sources && s > 0) {
                    const float4 o = *dst;
                    v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                }
                *dst = v;
            }
            if (a.lin.sum_sources && s + 1 < a.lin.nsrc) __syncthreads();    // (same thread -> same cell: ordering only)
        }
        for (int idx = tid; idx < kTM * Dout; idx += kBlock) {
            const int row = idx / Dout, j = idx - row * Dout;
            const int64_t r = r0 + row;
            float v = 0.f;
            if (r < a.lin.rows) {
                v = dY[r * a.ldy + j];
                if (a.mask) v = a.mask[(size_t)z * a.mask_zstride + r * a.ldm + j] > 0.f ? v : 0.f;
            }
            sY[row * ldy + j] = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tt = wave + 4 * t;
            if (tt < NT) {
                const int ti = tt / TJ, tj = tt - ti * TJ;
#pragma unroll
                for (int ks = 0; ks < kTM / 4; ++ks)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sX[(4 * ks + q16) * ldx + 16 * ti + l16],
                                                                 sY[(4 * ks + q16) * ldy + 16 * tj + l16], acc[t], 0, 0, 0);
            }
        }
        if (a.db && tid < Dout)
            for (int row = 0; row < kTM; ++row) accb += sY[row * ldy + tid];
        __syncthreads();
    }
    float* dW = a.dW + (size_t)z * a.dw_zstride;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tt = wave + 4 * t;
        if (tt < NT) {
            const int ti = tt / TJ, tj = tt - ti * TJ;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (acc[t][r] != 0.f) atomicAdd(dW + (size_t)(16 * ti + 4 * q16 + r) * Dout + 16 * tj + l16, acc[t][r]);
        }
    }
    if (a.db && tid < Dout && accb != 0.f) atomicAdd(a.db + (size_t)z * a.db_zstride + tid, accb);
}

template <int TI, int TJ>
__global__ __launch_bounds__(kBlock) void linear_wgrad_mfma_kernel(WgradArgs a) {
    wgrad_mfma_body<TI, TJ>(a, blockIdx.x, gridDim.x, blockIdx.z);
}

// Several weight-gradient problems of ONE (Din, Dout) tile shape in one launch (blockIdx.y = problem): at the reference's
// batch sizes every such launch is a few microseconds of work behind ~10 us of launch + ramp, and a training step has
// eleven of them (mvin_linear_wgrad_multi; mvin_amd/training.py queues them and flushes before the optimizer).
constexpr int kWgradMulti = 8;
struct WgradMultiArgs {
    WgradArgs p[kWgradMulti];
    int gx[kWgradMulti];
    int n;
};
template <int TI, int TJ>
__global__ __launch_bounds__(kBlock) void linear_wgrad_mfma_multi_kernel(WgradMultiArgs m) {
    const int pi = blockIdx.y;
    const WgradArgs& a = m.p[pi];
    const int nz = a.lin.nz > 0 ? a.lin.nz : 1;
    if ((int)blockIdx.x >= m.gx[pi] || (int)blockIdx.z >= nz) return;
    wgrad_mfma_body<TI, TJ>(a, blockIdx.x, m.gx[pi], blockIdx.z);
}

// ------------------------------------------------------------------------------------------
// backward of the neighbor mix (aggregators.py:118-152): agg[t] = (1/K) sum_k p[t,k] c[t,k],
// p = softmax_k(logit[rel[t,k]]).  Given dvec[t] = dL/d agg[t]:
//     g_k = dvec . c_k ;  dc_k = (p_k / K) dvec ;  dlogit_k = (p_k / K) (g_k - sum_j p_j g_j)
// children c_k: dense rows (child[t*K+k]) -> dchild written; or table rows through the adjacency
// (deepest hop) -> dc_k atomically added to dtable[y_k].  dT[rel] accumulates the logit gradients
// (block-level LDS table, flushed with one atomic per relation).  One wave per task.
// ------------------------------------------------------------------------------------------
// "By entity" form (gather form with node_ids == NULL and rel_score given): task t IS entity t, dvec
// [nE, D] holds the gradient rows already summed over every tree node that carries that entity
// (everything here is linear in dvec and depends on the node only through its entity), the attention
// weights are recomputed from rel_score, and tasks whose row is all zero are skipped.
__global__ __launch_bounds__(kBlock) void agg_bwd_kernel(AggBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sT = smem;                          // [nR]
    float* sG = sT + a.nR;                     // [4 waves][K]
    float* sP = sG + 4 * a.K;                  // [4 waves][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, K = a.K;
    const int lpr = 1 << a.lpr_log2, rpw = kWave >> a.lpr_log2;
    const int g = lane >> a.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    float* gk = sG + wave * K;
    float* pk = sP + wave * K;
    const bool att = a.probs != nullptr || a.rel_score != nullptr;
    for (int i = tid; i < a.nR; i += kBlock) sT[i] = 0.f;
    __syncthreads();
    const float invK = 1.f / (float)K;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < a.T; t += (int64_t)gridDim.x * 4) {
        const int64_t xbase = a.gather ? (a.node_ids ? (int64_t)a.node_ids[t] : t) * K : 0;
        const float4 dv = cact ? reinterpret_cast<const float4*>(a.dvec + t * D)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.skip_zero && !__any(dv.x != 0.f || dv.y != 0.f || dv.z != 0.f || dv.w != 0.f)) continue;
        // attention weights of this task -> pk
        if (a.probs) {
            for (int k = lane; k < K; k += kWave) pk[k] = a.probs[t * K + k];
        } else if (a.rel_score) {   // softmax_k(rel_score[adj_r[x, k]]), as the forward computes it
            float mx = -INFINITY;
            for (int k = lane; k < K; k += kWave) {
                const float sc = a.rel_score[a.adj_r[xbase + k]];
                pk[k] = sc;
                mx = fmaxf(mx, sc);
            }
            mx = wave_max(mx);
            float sum = 0.f;
            for (int k = lane; k < K; k += kWave) {
                const float e = expf(pk[k] - mx);
                pk[k] = e;
                sum += e;
            }
            sum = wave_sum(sum);
            for (int k = lane; k < K; k += kWave) pk[k] = pk[k] / sum;
        } else {
            for (int k = lane; k < K; k += kWave) pk[k] = 1.f;
        }
        // pass 1: g_k = dvec . c_k
        for (int k0 = 0; k0 < K; k0 += rpw) {
            const int k = k0 + g;
            float part = 0.f;
            if (k < K && cact) {
                const float* row = a.gather ? a.table + (int64_t)a.adj_e[xbase + k] * D : a.child + (t * K + k) * (int64_t)D;
                const float4 v = reinterpret_cast<const float4*>(row)[c];
                part = fmaf(dv.x, v.x, fmaf(dv.y, v.y, fmaf(dv.z, v.z, dv.w * v.w)));
            }
            part = group_sum(part, a.lpr_log2);
            if (k < K && c == 0) gk[k] = part;
        }
        __builtin_amdgcn_wave_barrier();
        float pg = 0.f;  // sum_j p_j g_j
        for (int k = lane; k < K; k += kWave) pg += pk[k] * gk[k];
        pg = wave_sum(pg);
        // logit gradients -> relation table
        if (att) {
            for (int k = lane; k < K; k += kWave) {
                const float dl = pk[k] * invK * (gk[k] - pg);
                const int r = a.gather ? a.adj_r[xbase + k] : a.rel_ids[t * K + k];
                atomicAdd(&sT[r], dl);
            }
        }
        // pass 2: child gradients
        if (a.gather) {
            // row-contiguous atomics (see scatter_add_rows_kernel): a lane owns ONE element of dvec, a wave
            // instruction adds 64 consecutive floats = 64 / dpad whole rows
            const int dpad = 4 * lpr;                      // row slot, a power of two >= D
            if (dpad <= kWave) {
                const int kk = lane / dpad, e = lane - kk * dpad, rp = kWave / dpad;
                const float dve = e < D ? a.dvec[t * D + e] : 0.f;
                for (int k0 = 0; k0 < K; k0 += rp) {
                    const int k = k0 + kk;
                    if (k < K && e < D) atomicAdd(a.dtable + (int64_t)a.adj_e[xbase + k] * D + e, pk[k] * invK * dve);
                }
            } else {
                for (int k = 0; k < K; ++k) {
                    const float w = pk[k] * invK;
                    float* d = a.dtable + (int64_t)a.adj_e[xbase + k] * D;
                    for (int e = lane; e < D; e += kWave) atomicAdd(d + e, w * a.dvec[t * D + e]);
                }
            }
        } else {
            for (int k0 = 0; k0 < K; k0 += rpw) {
                const int k = k0 + g;
                if (k < K && cact) {
                    const float w = pk[k] * invK;
                    reinterpret_cast<float4*>(a.dchild + (t * K + k) * (int64_t)D)[c] =
                        make_float4(w * dv.x, w * dv.y, w * dv.z, w * dv.w);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (a.dT)
        for (int i = tid; i < a.nR; i += kBlock)
            if (sT[i] != 0.f) atomicAdd(a.dT + i, sT[i]);
}

// ------------------------------------------------------------------------------------------
// backward of mvin_rel_score: t[r] = Rel[r] . w[D:2D]
// ------------------------------------------------------------------------------------------
__global__ void rel_score_bwd_kernel(const float* __restrict__ rel, const float* __restrict__ urh_w,
                                     const float* __restrict__ dT, int nR, int D, float* __restrict__ drel,
                                     float* __restrict__ durh) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float dw = 0.f;
    const float w = urh_w[D + d];
    for (int r = 0; r < nR; ++r) {
        const float g = dT[r];
        atomicAdd(drel + (size_t)r * D + d, g * w);
        dw = fmaf(g, rel[(size_t)r * D + d], dw);
    }
    atomicAdd(durh + D + d, dw);
}

// ------------------------------------------------------------------------------------------
// backward of the key-addressing reads (model.py:161-240), one wave per pair.  Recomputes the
// logits and softmaxes of the forward, then for every memory m of every hop:
//   hop : do = dL/d o_hop ;  g_m = do . t_m ;  dl_m = p_m (g_m - sum p g)
//         dE[t_m] += p_m do (+ 2 l2 t_m) ; dE[h_m] += dl_m V[b,r_m] (+ 2 l2 h_m) ; dV[b,r_m] += dl_m h_m
//   set : ds = dL/d o_hset ; g'_m = ds . h0_m ; dl'_m = p'_m (g'_m - sum p' g')
//         dE[h0_m] += p'_m ds + dl'_m w_h ;  dw_h += dl'_m h0_m
// (the 2 l2 terms are the sum(h_emb^2) + sum(t_emb^2) regularisers of model.py:383-385).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void key_addr_bwd_kernel(KeyAddrBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.f.D, Nm = a.f.Nm;
    const int lpr = 1 << a.f.lpr_log2, rpw = kWave >> a.f.lpr_log2;
    const int g = lane >> a.f.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    const int nm2 = (2 * Nm + 3) & ~3;   // 16-byte aligned tiles behind the two [Nm] arrays
    float* sL = smem;                    // [Nm] logits of the read          (workgroup-shared)
    float* sGm = sL + Nm;                // [Nm] g_m
    // row-contiguous atomics: the 4-floats-per-lane results of a step (rpw rows x 4*lpr floats = 256 words per target)
    // go through a wave-private tile so that one atomic instruction adds 64 CONSECUTIVE floats -- the L2 atomic
    // units work per cache line, and a float4-per-lane atomic touches a quarter of eight lines (scatter_add_rows: 4x)
    float* sTr = smem + nm2 + wave * (3 * 256 + 3 * 64);       // [3 targets][256]
    int* sRowI = reinterpret_cast<int*>(sTr + 3 * 256);        // [3 targets][64] row numbers (-1: none); rpw <= 64
    // dV[b, r, :] of one (pair, hop) collects Nm contributions on only nR rows: summed in LDS (ds_add_f32) and flushed
    // once -- nR row atomics per read instead of Nm
    float* sDV = a.dv_lds ? smem + nm2 + 4 * (3 * 256 + 3 * 64) : nullptr;     // [nR][D]   (workgroup-shared)
    const int slot0 = a.f.w ? 1 : 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Ef = reinterpret_cast<const float*>(a.f.E);   // training keeps the tables in fp32

    // one WORKGROUP per (pair, read), its four waves striding over the memories: the P hop reads and the h-set read of
    // a pair are independent, and at the reference's batch sizes (512 / 1024 pairs) the kernel is one long dependent
    // chain per task (ids -> rows -> logits -> softmax -> rows again -> atomics) with most of the chip idle: a wave per
    // PAIR took 3x, a wave per (pair, read) 1.9x the time of this form at B = 512
    const int nread = a.f.P + (a.f.w ? 1 : 0);
    const int mstep = 4 * rpw;
    float reg = 0.f;                     // this lane's share of sum(h^2) + sum(t^2) over the hop rows it reads
    for (int64_t u = blockIdx.x; u < a.f.B * nread; u += gridDim.x) {
        const int64_t b = u / nread;
        const int rd = (int)(u - b * nread);
        const bool is_set = rd == a.f.P;          // reads 0..P-1: hops; read P: the h-set read (rows of hop 0)
        const int hop = is_set ? 0 : rd;
        const int32_t* mh = a.f.mem_h[hop] + b * Nm;
        const int32_t* mr = is_set ? nullptr : a.f.mem_r[hop] + b * Nm;
        const int32_t* mv = is_set ? mh : a.f.mem_t[hop] + b * Nm;   // value rows
        const float4 dvo = cact ? reinterpret_cast<const float4*>(
                                      a.dout + b * a.f.ldo + (int64_t)(is_set ? 0 : slot0 + hop) * D)[c] : z4;
        // logits and g_m
        for (int m0 = wave * rpw; m0 < Nm; m0 += mstep) {
            const int m = m0 + g;
            float pl = 0.f, pg = 0.f;
            if (m < Nm && cact) {
                const float4 h = reinterpret_cast<const float4*>(Ef + (int64_t)mh[m] * D)[c];
                const float4 sv = is_set ? reinterpret_cast<const float4*>(a.f.w)[c]
                                         : reinterpret_cast<const float4*>(a.f.V + (b * a.f.nR + mr[m]) * (int64_t)D)[c];
                pl = fmaf(h.x, sv.x, fmaf(h.y, sv.y, fmaf(h.z, sv.z, h.w * sv.w)));
                const float4 val = reinterpret_cast<const float4*>(Ef + (int64_t)mv[m] * D)[c];
                pg = fmaf(dvo.x, val.x, fmaf(dvo.y, val.y, fmaf(dvo.z, val.z, dvo.w * val.w)));
            }
            pl = group_sum(pl, a.f.lpr_log2);
            pg = group_sum(pg, a.f.lpr_log2);
            if (m < Nm && c == 0) {
                sL[m] = pl;
                sGm[m] = pg;
            }
        }
        if (sDV && !is_set)
            for (int e = tid; e < a.f.nR * D; e += kBlock) sDV[e] = 0.f;
        __syncthreads();
        // softmax statistics over all Nm logits: every wave computes them for itself (read-only on sL / sGm)
        float mx = -INFINITY;
        for (int m = lane; m < Nm; m += kWave) mx = fmaxf(mx, sL[m]);
        mx = wave_max(mx);
        float zs = 0.f;
        for (int m = lane; m < Nm; m += kWave) zs += expf(sL[m] - mx);
        zs = wave_sum(zs);
        float pgs = 0.f;
        for (int m = lane; m < Nm; m += kWave) pgs += expf(sL[m] - mx) / zs * sGm[m];
        pgs = wave_sum(pgs);
        // scatter
        float4 dwacc = z4;   // h-set logit-weight gradient: per-lane partial, ONE atomic set per wave
        const int rowf = 4 * lpr;            // floats per row slot of the transpose tile (>= D)
        for (int m0 = wave * rpw; m0 < Nm; m0 += mstep) {
            const int m = m0 + g;
            const bool act = m < Nm;
            float4 dh = z4, dval = z4, dvv = z4;
            int hrow = -1, vrow = -1, rrow = -1;
            if (act && cact) {
                const float p = expf(sL[m] - mx) / zs;
                const float dl = p * (sGm[m] - pgs);
                hrow = mh[m];
                vrow = mv[m];
                const float4 h = reinterpret_cast<const float4*>(Ef + (int64_t)hrow * D)[c];
                if (is_set) {
                    const float4 w4 = reinterpret_cast<const float4*>(a.f.w)[c];
                    // value row == head row
                    dh = make_float4(p * dvo.x + dl * w4.x, p * dvo.y + dl * w4.y, p * dvo.z + dl * w4.z,
                                     p * dvo.w + dl * w4.w);
                    dwacc = f4_fma(dl, h, dwacc);
                    vrow = -1;
                } else {
                    const int r = mr[m];
                    const float4 v4 = reinterpret_cast<const float4*>(a.f.V + (b * a.f.nR + r) * (int64_t)D)[c];
                    const float4 val = reinterpret_cast<const float4*>(Ef + (int64_t)vrow * D)[c];
                    const float l2 = 2.f * a.l2;
                    dh = make_float4(dl * v4.x + l2 * h.x, dl * v4.y + l2 * h.y, dl * v4.z + l2 * h.z,
                                     dl * v4.w + l2 * h.w);
                    dval = make_float4(p * dvo.x + l2 * val.x, p * dvo.y + l2 * val.y, p * dvo.z + l2 * val.z,
                                       p * dvo.w + l2 * val.w);
                    dvv = make_float4(dl * h.x, dl * h.y, dl * h.z, dl * h.w);
                    rrow = r;
                    reg = fmaf(h.x, h.x, fmaf(h.y, h.y, fmaf(h.z, h.z, fmaf(h.w, h.w, reg))));
                    reg = fmaf(val.x, val.x, fmaf(val.y, val.y, fmaf(val.z, val.z, fmaf(val.w, val.w, reg))));
                }
            }
            // transpose through LDS: tile[target][g][4c..4c+3]
            *reinterpret_cast<float4*>(sTr + 0 * 256 + g * rowf + 4 * c) = dh;
            *reinterpret_cast<float4*>(sTr + 1 * 256 + g * rowf + 4 * c) = dval;
            *reinterpret_cast<float4*>(sTr + 2 * 256 + g * rowf + 4 * c) = dvv;
            if (c == 0) {
                sRowI[0 * 64 + g] = hrow;
                sRowI[1 * 64 + g] = vrow;
                sRowI[2 * 64 + g] = rrow;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = k * 64 + lane;
                const int j = e / rowf, col = e - j * rowf;
                if (col < D) {
                    const int hr = sRowI[j], vr = sRowI[64 + j], rr = sRowI[128 + j];
                    if (hr >= 0) atomicAdd(a.dE + (int64_t)hr * D + col, sTr[e]);
                    if (vr >= 0) atomicAdd(a.dE + (int64_t)vr * D + col, sTr[256 + e]);
                    if (rr >= 0) {
                        if (sDV) atomicAdd(sDV + rr * D + col, sTr[512 + e]);      // ds_add_f32
                        else atomicAdd(a.dV + (b * a.f.nR + rr) * (int64_t)D + col, sTr[512 + e]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (is_set) {
            dwacc = group_xor_sum(dwacc, lpr);
            if (cact && g == 0) {
                // dw_rep replicas of the [D] accumulator, picked by the pair: 4 B atomic instructions onto the same two
                // cache lines serialise in the L2 atomic unit (~50 ns each: 100 us at B = 512); the caller sums the replicas
                float* dw = a.dw + (int)((b * 4 + wave) & (a.dw_rep - 1)) * D + 4 * c;
                atomicAdd(dw + 0, dwacc.x);
                atomicAdd(dw + 1, dwacc.y);
                atomicAdd(dw + 2, dwacc.z);
                atomicAdd(dw + 3, dwacc.w);
            }
        }
        __syncthreads();                 // every wave is done with sL / sGm and has added its share to sDV
        if (sDV && !is_set) {
            float* dvb = a.dV + b * a.f.nR * (int64_t)D;
            for (int e = tid; e < a.f.nR * D; e += kBlock) {
                const float v = sDV[e];
                if (v != 0.f) atomicAdd(dvb + e, v);
            }
            if (a.Rk && a.items && D <= kWave) {
                // the item's share, from the block that is in LDS anyway: V[b, r, :] = E[item_b] . R[r]  =>
                // dE[item_b, i] += sum_r sum_j dV[b, r, j] R[r, i, j].  Lane = component i, the four waves stride over
                // the relations; one row-contiguous atomic per wave (it was a [B, nR*D] x [nR*D, D] product on the VALU
                // tile kernel -- 49 us at every batch size -- plus a scatter-add launch)
                const int64_t item = a.items64 ? reinterpret_cast<const int64_t*>(a.items)[b]
                                               : (int64_t)reinterpret_cast<const int32_t*>(a.items)[b];
                if (lane < D) {
                    float acc = 0.f;
                    for (int r = wave; r < a.f.nR; r += 4) {
                        const float4* rr = reinterpret_cast<const float4*>(a.Rk + ((size_t)r * D + lane) * D);
                        const float4* dv = reinterpret_cast<const float4*>(sDV + r * D);
#pragma unroll 4
                        for (int j = 0; j < (D >> 2); ++j) {
                            const float4 w4 = rr[j], d4 = dv[j];
                            acc = fmaf(w4.x, d4.x, fmaf(w4.y, d4.y, fmaf(w4.z, d4.z, fmaf(w4.w, d4.w, acc))));
                        }
                    }
                    if (acc != 0.f) atomicAdd(a.dE + item * D + lane, acc);
                }
            }
            __syncthreads();             // flushed before the next task zeroes it
        }
    }
    if (a.reg_accum) block_accumulate(a.reg_accum, a.l2 * reg);
}

// ------------------------------------------------------------------------------------------
static int blocks_for(int64_t n, int per) {
    int64_t b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

// ------------------------------------------------------------------------------------------
// every parameter of the model in ONE launch: L2 terms of model.py:387-412 (loss += l2/2 sum x^2,
// g += l2 x) and, optionally, the tf.train.AdamOptimizer update (model.py:414).  Gradients and
// the Adam moments live in flat buffers; a static segment table maps flat ranges to parameters.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_adam_multi_kernel(const mvin_param_seg* __restrict__ segs, int nseg,
                                                            int64_t total, float* __restrict__ g,
                                                            float* __restrict__ mo, float* __restrict__ vo,
                                                            float* accum, int apply_adam, float lr_t,
                                                            const float* __restrict__ lr_dev, float b1,
                                                            float b2, float eps) {
    __shared__ int64_t s_off[257];
    if (lr_dev) lr_t = *lr_dev;              // step size kept on the device: a captured step replays unchanged
    for (int i = threadIdx.x; i < nseg; i += blockDim.x) s_off[i] = segs[i].off;
    if (threadIdx.x == 0) s_off[nseg] = total;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float local = 0.f;
    auto seg_of = [&](int64_t i) {           // last segment with off <= i
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    };
    auto one = [&](float x, float& gr, float& m, float& v, float l2) {   // -> updated x
        if (l2 != 0.f) {
            gr = fmaf(l2, x, gr);
            local = fmaf(0.5f * l2 * x, x, local);
        }
        if (apply_adam) {
            m = b1 * m + (1.f - b1) * gr;
            v = b2 * v + (1.f - b2) * gr * gr;
            x = x - lr_t * m / (sqrtf(v) + eps);
        }
        return x;
    };
    // four consecutive floats per thread as 16-byte accesses wherever they sit in one parameter and its storage is
    // 16-byte aligned (every segment of the model is: D % 4 == 0); element-wise across segment boundaries and the tail
    const int64_t nquad = (total + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
        const int64_t i = q << 2;
        const int lo = seg_of(i);
        const mvin_param_seg sg = segs[lo];
        float* xp = sg.x + (i - sg.off);
        if (i + 4 <= s_off[lo + 1] && (reinterpret_cast<uintptr_t>(xp) & 15) == 0) {
            float4 x = *reinterpret_cast<float4*>(xp);
            float4 gr = *reinterpret_cast<float4*>(g + i);
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), v = m;
            if (apply_adam) {
                m = *reinterpret_cast<float4*>(mo + i);
                v = *reinterpret_cast<float4*>(vo + i);
            }
            x.x = one(x.x, gr.x, m.x, v.x, sg.l2);
            x.y = one(x.y, gr.y, m.y, v.y, sg.l2);
            x.z = one(x.z, gr.z, m.z, v.z, sg.l2);
            x.w = one(x.w, gr.w, m.w, v.w, sg.l2);
            if (sg.l2 != 0.f) *reinterpret_cast<float4*>(g + i) = gr;
            if (apply_adam) {
                *reinterpret_cast<float4*>(mo + i) = m;
                *reinterpret_cast<float4*>(vo + i) = v;
                *reinterpret_cast<float4*>(xp) = x;
            }
        } else {
            for (int64_t e = i; e < i + 4 && e < total; ++e) {
                const mvin_param_seg se = segs[seg_of(e)];
                float* xe = se.x + (e - se.off);
                float gr = g[e], m = 0.f, v = 0.f;
                if (apply_adam) {
                    m = mo[e];
                    v = vo[e];
                }
                const float xn = one(*xe, gr, m, v, se.l2);
                if (se.l2 != 0.f) g[e] = gr;
                if (apply_adam) {
                    mo[e] = m;
                    vo[e] = v;
                    *xe = xn;
                }
            }
        }
    }
    if (accum) block_accumulate(accum, local);
}

hipError_t launch_l2_adam_multi(const mvin_param_seg* segs, int nseg, int64_t total, float* g, float* mo, float* vo,
                                float* accum, int apply_adam, float lr_t, const float* lr_dev, float b1, float b2,
                                float eps, hipStream_t st) {
    l2_adam_multi_kernel<<<blocks_for((total + 3) / 4, 4096), 256, 0, st>>>(segs, nseg, total, g, mo, vo, accum, apply_adam,
                                                                  lr_t, lr_dev, b1, b2, eps);
    return hipGetLastError();
}

hipError_t launch_eltwise(const EltArgs& a, hipStream_t st) {
    // modes whose elements are chains of dependent / strided loads (6: N rows per output, 8: id -> row) get one
    // element per thread; the streaming modes four
    eltwise_kernel<<<blocks_for(a.n, (a.mode == 6 || a.mode == 8) ? 256 : 1024), 256, 0, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_scatter_add_rows(float* dtable, const int32_t* ids, int ids64, const float* x, int64_t rows, int D,
                                   float alpha, hipStream_t st) {
    scatter_add_rows_kernel<<<blocks_for(rows * D, 1024), 256, 0, st>>>(dtable, ids, ids64, x, rows, D, alpha);
    return hipGetLastError();
}

hipError_t launch_linear_wgrad(WgradArgs a, hipStream_t st);

// workgroups per weight matrix: every workgroup ends with Din x Dout atomics onto the SAME 16 KB, so more
// workgroups = more same-line atomics, fewer = a longer serial tile loop.  Measured optimum ~ tiles / 8
// (512 tiles: 256 workgroups 11.9 us vs 14.5 at 512; 4 096 tiles: 512 -> 27 us vs 34 at 256 and 30 at 1 024;
// 16 384 tiles: 1 024 -> 66 us vs 89 at 512)
static int wgrad_grid_x(const WgradArgs& a) {
    const int nz = a.lin.nz > 0 ? a.lin.nz : 1;
    const int64_t ntiles = (a.lin.rows + kTM - 1) / kTM;
    int64_t cap = ntiles / 8;
    cap = cap < 256 ? 256 : cap > 1024 ? 1024 : cap;
    cap /= nz;
    if (cap < 1) cap = 1;
    return (int)(ntiles < 1 ? 1 : ntiles < cap ? ntiles : cap);
}

template <int TI, int TJ>
static hipError_t launch_wgrad_mfma(const WgradArgs& a, hipStream_t st) {
    constexpr int Din = TI * 16, Dout = TJ * 16;
    constexpr int ldx = (Din % 32 == 0) ? Din + 16 : Din, ldy = (Dout % 32 == 0) ? Dout + 16 : Dout;
    const int nz = a.lin.nz > 0 ? a.lin.nz : 1;
    const size_t lds = (size_t)kTM * (ldx + ldy) * sizeof(float);
    linear_wgrad_mfma_kernel<TI, TJ><<<dim3(wgrad_grid_x(a), 1, nz), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

template <int TI, int TJ>
static hipError_t launch_wgrad_mfma_multi(const WgradArgs* const* probs, int n, hipStream_t st) {
    constexpr int Din = TI * 16, Dout = TJ * 16;
    constexpr int ldx = (Din % 32 == 0) ? Din + 16 : Din, ldy = (Dout % 32 == 0) ? Dout + 16 : Dout;
    const size_t lds = (size_t)kTM * (ldx + ldy) * sizeof(float);
    for (int i0 = 0; i0 < n; i0 += kWgradMulti) {
        WgradMultiArgs m{};
        m.n = n - i0 < kWgradMulti ? n - i0 : kWgradMulti;
        int gx = 1, gz = 1;
        for (int i = 0; i < m.n; ++i) {
            m.p[i] = *probs[i0 + i];
            m.gx[i] = wgrad_grid_x(m.p[i]);
            gx = m.gx[i] > gx ? m.gx[i] : gx;
            const int nz = m.p[i].lin.nz > 0 ? m.p[i].lin.nz : 1;
            gz = nz > gz ? nz : gz;
        }
        linear_wgrad_mfma_multi_kernel<TI, TJ><<<dim3(gx, m.n, gz), kBlock, lds, st>>>(m);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static bool wgrad_mfma_shape(const WgradArgs& a, int& ti, int& tj) {
    static const bool no_mfma = getenv("MVIN_WGRAD_VALU") != nullptr;
    const int Din = a.lin.sum_sources ? a.lin.Dsrc : a.lin.nsrc * a.lin.Dsrc;
    if (no_mfma || (a.lin.Dsrc & 3) || Din % 16 || a.lin.Dout % 16) return false;
    ti = Din / 16;
    tj = a.lin.Dout / 16;
#define MVIN_WG(TIV, TJV) if (ti == TIV && tj == TJV) return true;
    MVIN_WG(1, 1) MVIN_WG(2, 1) MVIN_WG(3, 1) MVIN_WG(4, 1)
    MVIN_WG(2, 2) MVIN_WG(4, 2) MVIN_WG(6, 2) MVIN_WG(8, 2)
    MVIN_WG(4, 4) MVIN_WG(8, 4) MVIN_WG(12, 4) MVIN_WG(16, 4)
    MVIN_WG(8, 8)
#undef MVIN_WG
    return false;
}

// n problems: those of one MFMA tile shape go out together (up to kWgradMulti per launch), the rest one by one
hipError_t launch_linear_wgrad_multi(const WgradArgs* probs, int n, hipStream_t st) {
    static const bool one_by_one = getenv("MVIN_WGRAD_MULTI") != nullptr && atoi(getenv("MVIN_WGRAD_MULTI")) == 0;
    bool done[64] = {};
    if (n > 64) return hipErrorInvalidValue;
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        int ti = 0, tj = 0;
        if (one_by_one || !wgrad_mfma_shape(probs[i], ti, tj)) {
            hipError_t e = launch_linear_wgrad(probs[i], st);
            if (e != hipSuccess) return e;
            done[i] = true;
            continue;
        }
        const WgradArgs* grp[64];
        int ng = 0;
        for (int j = i; j < n; ++j) {
            int tij = 0, tjj = 0;
            if (!done[j] && wgrad_mfma_shape(probs[j], tij, tjj) && tij == ti && tjj == tj) {
                grp[ng++] = &probs[j];
                done[j] = true;
            }
        }
        hipError_t e = hipErrorInvalidValue;
#define MVIN_WG(TIV, TJV) if (ti == TIV && tj == TJV) e = launch_wgrad_mfma_multi<TIV, TJV>(grp, ng, st);
        MVIN_WG(1, 1) MVIN_WG(2, 1) MVIN_WG(3, 1) MVIN_WG(4, 1)
        MVIN_WG(2, 2) MVIN_WG(4, 2) MVIN_WG(6, 2) MVIN_WG(8, 2)
        MVIN_WG(4, 4) MVIN_WG(8, 4) MVIN_WG(12, 4) MVIN_WG(16, 4)
        MVIN_WG(8, 8)
#undef MVIN_WG
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_linear_wgrad(WgradArgs a, hipStream_t st) {
    const int Din = a.lin.sum_sources ? a.lin.Dsrc : a.lin.nsrc * a.lin.Dsrc;
    static const bool no_mfma = getenv("MVIN_WGRAD_VALU") != nullptr;
    if (!no_mfma && (a.lin.Dsrc & 3) == 0) {
        const int ti = Din / 16, tj = a.lin.Dout / 16;
        if (Din % 16 == 0 && a.lin.Dout % 16 == 0) {
#define MVIN_WG(TIV, TJV) if (ti == TIV && tj == TJV) return launch_wgrad_mfma<TIV, TJV>(a, st);
            MVIN_WG(1, 1) MVIN_WG(2, 1) MVIN_WG(3, 1) MVIN_WG(4, 1)
            MVIN_WG(2, 2) MVIN_WG(4, 2) MVIN_WG(6, 2) MVIN_WG(8, 2)
            MVIN_WG(4, 4) MVIN_WG(8, 4) MVIN_WG(12, 4) MVIN_WG(16, 4)
            MVIN_WG(8, 8)
#undef MVIN_WG
        }
    }
    a.IB = 4096 / a.lin.Dout;
    if (a.IB < 1) a.IB = 1;
    if (a.IB > Din) a.IB = Din;
    const int ny = (Din + a.IB - 1) / a.IB;
    const int64_t ntiles = (a.lin.rows + kTM - 1) / kTM;
    int gx = (int)(ntiles < 64 ? (ntiles < 1 ? 1 : ntiles) : 64);
    const size_t lds = ((size_t)kTM * (Din + 4) + (size_t)kTM * (a.lin.Dout + 1)) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_wgrad_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    linear_wgrad_kernel<<<dim3(gx, ny, a.lin.nz > 0 ? a.lin.nz : 1), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_agg_bwd(const AggBwdArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)a.nR + 8 * (size_t)a.K) * sizeof(float);
    agg_bwd_kernel<<<blocks_for(a.T, 4), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_rel_score_bwd(const float* rel, const float* urh_w, const float* dT, int nR, int D, float* drel,
                                float* durh, hipStream_t st) {
    rel_score_bwd_kernel<<<(D + 63) / 64, 64, 0, st>>>(rel, urh_w, dT, nR, D, drel, durh);
    return hipGetLastError();
}

hipError_t launch_key_addr_bwd(const KeyAddrBwdArgs& a0, hipStream_t st) {
    KeyAddrBwdArgs a = a0;
    size_t lds = ((size_t)((2 * a.f.Nm + 3) & ~3) + 4 * (3 * 256 + 3 * 64)) * sizeof(float);
    const size_t dv = (size_t)a.f.nR * a.f.D * sizeof(float);
    a.dv_lds = (a.f.P > 0 && a.dV && lds + dv <= 48 * 1024 && getenv("MVIN_KAB_DV_GLOBAL") == nullptr) ? 1 : 0;
    if (a.dv_lds) lds += dv;
    if (a.Rk && !a.dv_lds) return hipErrorInvalidValue;      // the in-kernel item gradient reads the LDS copy of dV
    if (lds > 64 * 1024) {       // n_memory beyond ~6 000: raise the dynamic-LDS limit (the ABI caps Nm at 8 192 = 80 KB)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(key_addr_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    key_addr_bwd_kernel<<<blocks_for(a.f.B * (a.f.P + (a.f.w ? 1 : 0)), 1), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

}  // namespace mvin
