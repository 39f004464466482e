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

template <int NJ>
__global__ __launch_bounds__(kBlock) void key_addr_kernel(KeyAddrArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D, Nm = a.Nm;
    const int lpr = 1 << a.lpr_log2, rpw = kWave >> a.lpr_log2;
    const int g = lane >> a.lpr_log2, c = lane & (lpr - 1);
    const bool cact = (c << 2) < D;
    const int slot0 = a.w ? 1 : 0;
    const int nhop = a.P > 0 ? a.P : 1;

    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < a.B; b += (int64_t)gridDim.x * 4) {
        for (int hop = 0; hop < nhop; ++hop) {
            const bool do_hop = hop < a.P;
            const bool do_set = hop == 0 && a.w != nullptr;
            if (!do_hop && !do_set) continue;
            const int32_t* mh = a.mem_h[hop] + b * Nm;
            const int32_t* mr = do_hop ? a.mem_r[hop] + b * Nm : nullptr;
            const int32_t* mt = do_hop ? a.mem_t[hop] + b * Nm : nullptr;
            int hid[NJ], tix[NJ], rid[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int m = j * rpw + g;
                const bool v = m < Nm;
                hid[j] = v ? mh[m] : 0;
                tix[j] = (v && do_hop) ? mt[m] : 0;
                rid[j] = (v && do_hop) ? mr[m] : 0;
            }
            float4 hrow[NJ], trow[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                hrow[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cact && j * rpw + g < Nm) hrow[j] = reinterpret_cast<const float4*>(a.E + (int64_t)hid[j] * D)[c];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                trow[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (do_hop && cact && j * rpw + g < Nm)
                    trow[j] = reinterpret_cast<const float4*>(a.E + (int64_t)tix[j] * D)[c];
            }
            // ---- logits ----
            float sh[NJ], ss[NJ];
            const float4 wv = (do_set && cact) ? reinterpret_cast<const float4*>(a.w)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float ph = 0.f, ps = 0.f;
                if (do_hop && cact) {
                    const float4 v = reinterpret_cast<const float4*>(a.V + (b * a.nR + rid[j]) * (int64_t)D)[c];
                    ph = dot4(hrow[j], v);
                }
                if (do_set) ps = dot4(hrow[j], wv);
                for (int o = 1; o < lpr; o <<= 1) {
                    ph += __shfl_xor(ph, o, kWave);
                    ps += __shfl_xor(ps, o, kWave);
                }
                const bool v = j * rpw + g < Nm;
                sh[j] = v ? ph : -INFINITY;
                ss[j] = v ? ps : -INFINITY;
            }
            // ---- softmax over the Nm memories: in-lane over j, across row groups by xor ----
            float mxh = -INFINITY, mxs = -INFINITY;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                mxh = fmaxf(mxh, sh[j]);
                mxs = fmaxf(mxs, ss[j]);
            }
            for (int o = lpr; o < kWave; o <<= 1) {
                mxh = fmaxf(mxh, __shfl_xor(mxh, o, kWave));
                mxs = fmaxf(mxs, __shfl_xor(mxs, o, kWave));
            }
            float zh = 0.f, zs = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool v = j * rpw + g < Nm;
                sh[j] = (v && do_hop) ? expf(sh[j] - mxh) : 0.f;
                ss[j] = (v && do_set) ? expf(ss[j] - mxs) : 0.f;
                zh += sh[j];
                zs += ss[j];
            }
            for (int o = lpr; o < kWave; o <<= 1) {
                zh += __shfl_xor(zh, o, kWave);
                zs += __shfl_xor(zs, o, kWave);
            }
            // ---- weighted sums (model.py:195 / :229) ----
            if (do_set) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc = f4_fma(ss[j] / zs, hrow[j], acc);
                acc = group_xor_sum(acc, lpr);
                if (cact && g == 0) *reinterpret_cast<float4*>(a.out + b * a.ldo + (c << 2)) = acc;
            }
            if (do_hop) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc = f4_fma(sh[j] / zh, trow[j], acc);
                acc = group_xor_sum(acc, lpr);
                if (cact && g == 0)
                    *reinterpret_cast<float4*>(a.out + b * a.ldo + (int64_t)(slot0 + hop) * D + (c << 2)) = acc;
            }
        }
    }
}

int key_addr_nj(int Nm, int D) {
    const int rpw = kWave >> lpr_log2_for(D);
    const int need = (Nm + rpw - 1) / rpw;
    int nj = 1;
    while (nj < need) nj *= 2;
    return nj;  // > 16 means: not supported by the register-resident kernel
}

hipError_t launch_key_addr(const KeyAddrArgs& a, hipStream_t st) {
    const int nj = key_addr_nj(a.Nm, a.D);
    const int64_t nblk = (a.B + 3) / 4;
    const int64_t cap = 256 * 8;
    const int grid = (int)(nblk < cap ? nblk : cap);
    switch (nj) {
        case 1: key_addr_kernel<1><<<grid, kBlock, 0, st>>>(a); break;
        case 2: key_addr_kernel<2><<<grid, kBlock, 0, st>>>(a); break;
        case 4: key_addr_kernel<4><<<grid, kBlock, 0, st>>>(a); break;
        case 8: key_addr_kernel<8><<<grid, kBlock, 0, st>>>(a); break;
        case 16: key_addr_kernel<16><<<grid, kBlock, 0, st>>>(a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mvin
