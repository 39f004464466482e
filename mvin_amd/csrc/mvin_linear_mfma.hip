// rows x small dense on the matrix cores: out[z][r,:] = act(X[r,:] . W[z] + bias[z] + rowbias)
// for Dout = D in {16, 32, 64} and Din = NE * D (NE = 1..4) -- and D = 128 with NE <= 2, two 16-column slabs per wave
// (BASELINE config C5's projections and aggregator epilogues) --, the shapes of every tf.matmul
// site of the path at those dims (model.py:279 projection, :312 mix-hop combiner, :234 user MLP,
// the per-relation item projection of :214-220, aggregators.py:110).  Same contract as the VALU
// linear_kernel (gathered / concatenated / summed sources, int32 or int64 ids, z-batched weights,
// per-pair row bias, ReLU, fused score + sigmoid of model.py:158-159).
//
// One workgroup (4 waves) owns a weight matrix W[z] for its whole life: the B fragments of
// v_mfma_f32_16x16x4_f32 stay in VGPRs (Din/4 registers per wave) while it walks row tiles of
// 32: rows are staged in LDS with stride Din+2 (conflict-free A-fragment ds_read_b32), each wave
// produces one 16-column slab of the tile.
#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D, int NE>
__global__ __launch_bounds__(kBlock) void linear_mfma_kernel(mvin_linear_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = D / 16;
    constexpr int MTW = (NT >= 4) ? 2 : 1;
    constexpr int SL = (NT == 8) ? 2 : 1;      // 16-column slabs per wave (slab nt + 4 * sl)
    constexpr int DIN = NE * D;
    constexpr int KS = DIN / 4;
    constexpr int LDX = DIN + 2;
    float* sX = smem;                 // [32][LDX]
    float* sScore = sX + kTM * LDX;   // [NT][32] per-slab partial dot products (summed in a fixed order)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q16 = lane >> 4, l16 = lane & 15;
    const int nt = (NT >= 4) ? wave : (wave % NT);
    const int mt0 = (NT >= 4) ? 0 : (wave / NT);
    const bool dense = mt0 < 2;
    const int z = blockIdx.y;
    const float* W = a.W + (size_t)z * a.w_zstride;
    const float* bias = a.bias ? a.bias + (size_t)z * a.bias_zstride : nullptr;
    float* out = a.out + (size_t)z * a.out_zstride;

    float bW[SL][KS], bj[SL];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
        const int col = 16 * (nt + 4 * sl) + l16;
#pragma unroll
        for (int s = 0; s < KS; ++s) bW[sl][s] = dense ? W[(size_t)(4 * s + q16) * D + col] : 0.f;
        bj[sl] = (dense && bias) ? bias[col] : 0.f;
    }

    const int c4 = a.Dsrc >> 2;
    const int64_t ntiles = (a.rows + kTM - 1) / kTM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * kTM;
        // ---- stage the row tile (concat or sum of the sources) ----
        // one contiguous fp32 source of the full input width (the user MLP over o_cat, model.py:232-236; aggregator epilogues): every
        // 16-byte load of the tile in flight at once.  (The general loop below has run-time bounds: hipcc leaves it rolled, one load
        // -> wait -> LDS store per trip, six memory latencies in a row per 32-row tile at Din = 192 -- the kernel ran at half its
        // MFMA rate with four workgroups per CU taking turns to wait.)
        constexpr int C4 = DIN / 4, PER = kTM * C4 >= kBlock ? kTM * C4 / kBlock : 1;
        const bool plain = a.nsrc == 1 && !a.ids[0] && !a.sum_sources && !(a.src_bf16 & 1) && a.Dsrc == DIN && PER * kBlock == kTM * C4;
        if (plain) {
            const float4* src = reinterpret_cast<const float4*>(a.src[0]);
            int tl = tid;                                    // laundered: the per-thread (row, chunk) pairs of the loads are recomputed per
            asm volatile("" : "+v"(tl));                     // tile -- kept across the tile loop they cost the kernel its fourth wave per SIMD
            // (in rounds of at most three loads per thread: all six at once cost the kernel its fourth wave per SIMD -- 140 registers)
            constexpr int RND = PER > 3 ? (PER % 3 == 0 ? 3 : 2) : PER;
#pragma unroll
            for (int i0 = 0; i0 < PER; i0 += RND) {
                float4 v[RND];
#pragma unroll
                for (int i = 0; i < RND; ++i) {
                    const int idx = tl + (i0 + i) * kBlock, row = idx / C4, c = idx - row * C4;
                    const int64_t r = r0 + row;
                    v[i] = (i0 + i < PER && r < a.rows) ? src[r * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < RND; ++i) {
                    if (i0 + i < PER) {
                        const int idx = tl + (i0 + i) * kBlock, row = idx / C4, c = idx - row * C4;
                        float* dst = sX + row * LDX + c * 4;
                        *reinterpret_cast<float2*>(dst) = make_float2(v[i].x, v[i].y);
                        *reinterpret_cast<float2*>(dst + 2) = make_float2(v[i].z, v[i].w);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);           // (or the scheduler hoists the next round's loads up here)
            }
        } else
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
                float* dst = sX + row * LDX + (a.sum_sources ? 0 : s * a.Dsrc) + c * 4;
                if (a.sum_sources && s > 0) {  // same thread wrote this slot for s-1
                    v.x += dst[0];
                    v.y += dst[1];
                    v.z += dst[2];
                    v.w += dst[3];
                }
                *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
            }
        }
        __syncthreads();
        if (dense) {
            f32x4 acc[SL][MTW];
#pragma unroll
            for (int sl = 0; sl < SL; ++sl)
#pragma unroll
                for (int m = 0; m < MTW; ++m) acc[sl][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const float av = sX[(16 * (mt0 + m) + l16) * LDX + 4 * s + q16];
#pragma unroll
                    for (int sl = 0; sl < SL; ++sl)
                        acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bW[sl][s], acc[sl][m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) {
                const int col = 16 * (nt + 4 * sl) + l16;
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = 16 * (mt0 + m) + 4 * q16 + r4;
                        const int64_t r = r0 + row;
                        float part = 0.f;
                        if (r < a.rows) {
                            float v = acc[sl][m][r4] + bj[sl];
                            if (a.rowbias) v += a.rowbias[(r / a.rows_per_group) * D + col];
                            if (a.relu) v = fmaxf(v, 0.f);
                            out[r * a.ldo + col] = v;
                            if (a.score_u) part = v * a.score_u[r * D + col];
                        }
                        if (a.score_u) {   // uniform branch: all lanes reduce over the 16 columns of the slab
                            part = group_sum(part, 4);
                            if (l16 == 0) sScore[(nt + 4 * sl) * kTM + row] = part;
                        }
                    }
                }
            }
        }
        if (a.score_u) {
            __syncthreads();
            if (tid < kTM) {
                const int64_t r = r0 + tid;
                if (r < a.rows) {
                    float sc = 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) sc += sScore[t * kTM + tid];
                    if (a.score_out) a.score_out[r] = sc;
                    if (a.sigmoid_out) a.sigmoid_out[r] = 1.f / (1.f + expf(-sc));
                }
            }
        }
        __syncthreads();
    }
}

bool linear_mfma_supported(const mvin_linear_args& a) {
    if (!a.W) return false;
    const int D = a.Dout;
    if (D != 16 && D != 32 && D != 64 && D != 128) return false;
    const int din = (a.sum_sources ? 1 : a.nsrc) * a.Dsrc;
    if (din % D) return false;
    const int ne = din / D;
    return ne >= 1 && ne <= (D == 128 ? 2 : 4);     // D = 128: 2 * Din / 4 fragment registers per lane
}

template <int D, int NE>
static hipError_t launch_lm(const mvin_linear_args& a, hipStream_t st) {
    const size_t lds = ((size_t)kTM * (NE * D + 2) + (D / 16) * kTM) * sizeof(float);
    const int64_t ntiles = (a.rows + kTM - 1) / kTM;
    const int nz = a.nz > 0 ? a.nz : 1;
    // workgroups per CU over all z, from a sweep at C3: one full round at the kernel's occupancy for the
    // wide-input forms (4 waves/SIMD at NE = 3), 12 for NE = 1 (7 waves/SIMD; a second round hides the tail)
    constexpr int kWgPerCu = NE == 1 ? 12 : 4;
    int64_t gx = (256 * kWgPerCu + nz - 1) / nz;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    linear_mfma_kernel<D, NE><<<dim3((unsigned)gx, (unsigned)nz), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_lm_ne(const mvin_linear_args& a, int ne, hipStream_t st) {
    switch (ne) {
        case 1: return launch_lm<D, 1>(a, st);
        case 2: return launch_lm<D, 2>(a, st);
        case 3: return launch_lm<D, 3>(a, st);
        default: return launch_lm<D, 4>(a, st);
    }
}

hipError_t launch_linear_mfma(const mvin_linear_args& a, hipStream_t st) {
    const int ne = (a.sum_sources ? 1 : a.nsrc) * a.Dsrc / a.Dout;
    switch (a.Dout) {
        case 16: return launch_lm_ne<16>(a, ne, st);
        case 32: return launch_lm_ne<32>(a, ne, st);
        case 128: return ne == 1 ? launch_lm<128, 1>(a, st) : launch_lm<128, 2>(a, st);
        default: return launch_lm_ne<64>(a, ne, st);
    }
}

}  // namespace mvin
