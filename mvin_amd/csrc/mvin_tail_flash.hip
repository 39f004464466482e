// mvin_l2_tail_fwd at dim 64 without a workgroup phase: everything of MVIN.aggregate_delta_whole above the fused two-level kernel
// (mvin_tail.hip's header: ev0 -> out0 -> out2 -> mix-hop combiner -> score; model.py:270-317, aggregators.py:108-116, model.py:158-159)
// as a chain of TRANSPOSED products on v_mfma_f32_16x16x4_f32,
//     Y^T[n, pair] = sum_c W[c][n] X[pair][c]        A = W   (LDS copy, packed in operand order once per workgroup)
//                                                    B = X^T (registers)
// Accumulator register r of column tile nt of lane (q16, l16) is element [n = 16 nt + 4 q16 + r][pair = l16]; contraction step
// (nt, r) of lane group q16 stands for c = 16 nt + 4 q16 + r: the accumulators of one product, after its bias / ReLU / + nagg, ARE
// the B operand of the next.  A wave walks tiles of 32 pairs (two 16-pair column tiles share every A operand read) on its own:
// no LDS image of the activations, no barrier after the prologue (the tile-image kernel had five barrier-separated phases per
// 32-row tile and two waves per SIMD to cover them: 0.31 ms per 524 288 pairs against 0.17 ms of matrix-pipe time), the mix-hop
// combiner accumulated block by block as its inputs appear (ev0, out0, out2 never coexist).  Per-pair rows (E[item], q, nagg0,
// nagg1, user_o) are loaded straight into the accumulator layout, 16 bytes per lane, a product or a tile ahead of their use.
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTfWaves = 12;          // one workgroup per CU: three waves per SIMD share one 96 KB copy of the six weight blocks
constexpr int kTfRT = 1;              // 16-pair column tiles per wave tile (2: every A operand read shared by two tiles, 64 more registers)

__device__ __forceinline__ int tf_swz(int lane) { return ((lane >> 2) ^ (lane >> 3)) & 3; }   // (see flash_swz, mvin_keyaddr_flash.hip)

#define TF_FENCE() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(kTfWaves * 64, kTfWaves / 4) void l2_tail_flash_kernel(TailArgs a) {
    constexpr int D = 64;
    extern __shared__ __attribute__((aligned(16))) float sW[];      // [6 blocks][4 column tiles][64 lanes][16]
    const int lane = threadIdx.x & 63, l16 = lane & 15, q16 = lane >> 4;
    // ---- prologue: W0 | A0 | A1 | Wmix[0:D] | Wmix[D:2D] | Wmix[2D:3D] in A-operand order:
    //      element (block, ntp, lane (q, l), nt, r) = W_block[16 nt + 4 q + r][16 ntp + l], chunk nt at position nt ^ swz(lane)
    for (int idx = threadIdx.x; idx < 6 * D * D; idx += kTfWaves * 64) {
        const int blk = idx >> 12, c = (idx >> 6) & 63, n = idx & 63;
        const float* src = blk == 0 ? a.W0 : blk == 1 ? a.A0 : blk == 2 ? a.A1 : a.Wmix + (size_t)(blk - 3) * D * D;
        const int nt = c >> 4, q = (c >> 2) & 3, r = c & 3, ntp = n >> 4, l = n & 15, ln = 16 * q + l;
        sW[((blk * 4 + ntp) * 64 + ln) * 16 + 4 * (nt ^ tf_swz(ln)) + r] = src[c * D + n];
    }
    __syncthreads();
    const int swz = tf_swz(lane);
    const float4* sWl = reinterpret_cast<const float4*>(sW) + (size_t)lane * 4;      // + (blk * 4 + ntp) * 256 + (nt ^ swz)
    const unsigned emax = (unsigned)(a.n_entity > 0 ? a.n_entity - 1 : 0x7fffffff);
    const float* __restrict__ E = reinterpret_cast<const float*>(a.E);

    // acc[rt][ntp] += W_blk^T B[rt]: eight accumulator chains (4 column tiles x 2 pair tiles) per contraction step
    auto product = [&](int blk, const f32x4 (&B)[kTfRT][4], f32x4 (&acc)[kTfRT][4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float4 w[4];
#pragma unroll
            for (int ntp = 0; ntp < 4; ++ntp) w[ntp] = sWl[(blk * 4 + ntp) * 256 + (((a.dbg & 2) ? 0 : nt) ^ swz)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int ntp = 0; ntp < 4; ++ntp) {
                    const float wv = r == 0 ? w[ntp].x : r == 1 ? w[ntp].y : r == 2 ? w[ntp].z : w[ntp].w;
#pragma unroll
                    for (int rt = 0; rt < kTfRT; ++rt) acc[rt][ntp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, B[rt][nt][r], acc[rt][ntp], 0, 0, 0);
                }
            }
        }
    };
    // a [*, D] row-major per-pair tensor in accumulator layout: lane (q16, l16) of pair tile rt holds columns 16 nt + 4 q16 + [0, 4)
    auto load_rows = [&](const float* __restrict__ t, const int64_t (&row)[kTfRT], f32x4 (&v)[kTfRT][4]) {
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 x = reinterpret_cast<const float4*>(t + row[rt] * D)[4 * nt + q16];
                v[rt][nt] = f32x4{x.x, x.y, x.z, x.w};
            }
        }
    };
    auto bias4 = [&](const float* __restrict__ b, f32x4 (&v)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 x = b ? reinterpret_cast<const float4*>(b)[4 * nt + q16] : make_float4(0.f, 0.f, 0.f, 0.f);
            v[nt] = f32x4{x.x, x.y, x.z, x.w};
        }
    };
    const int64_t ntiles = (a.B + 16 * kTfRT - 1) / (16 * kTfRT);
    const int64_t stride = (int64_t)gridDim.x * kTfWaves;
    int64_t tile = (int64_t)blockIdx.x * kTfWaves + (threadIdx.x >> 6);
    // rows of a tile: pair r0 + 16 rt + l16 (clamped: the rows of a ragged last tile are computed and not stored)
    auto tile_rows = [&](int64_t t, int64_t (&row)[kTfRT], int64_t (&item)[kTfRT]) {
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
            const int64_t p = t * (16 * kTfRT) + 16 * rt + l16;
            row[rt] = p < a.B ? p : a.B - 1;
            const unsigned id = a.items64 ? reinterpret_cast<const unsigned*>(a.items64)[2 * row[rt]] : (unsigned)a.items32[row[rt]];
            item[rt] = (int64_t)min(id, emax);
        }
    };
    int64_t rowN[kTfRT], itemN[kTfRT];
    f32x4 xe[kTfRT][4], xq[kTfRT][4];
    if (tile < ntiles) {
        tile_rows(tile, rowN, itemN);
        load_rows(E, itemN, xe);
        load_rows(a.q, rowN, xq);
    }
    for (; tile < ntiles; tile += stride) {
        int64_t row[kTfRT];
        bool valid[kTfRT];
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) row[rt] = rowN[rt], valid[rt] = tile * (16 * kTfRT) + 16 * rt + l16 < a.B;
        f32x4 act[kTfRT][4], n0[kTfRT][4];
        if (!(a.dbg & 1)) load_rows(a.nagg0, row, n0);
        else {
#pragma unroll
            for (int rt = 0; rt < kTfRT; ++rt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) n0[rt][nt] = xe[rt][nt];
        }
        TF_FENCE();
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) act[rt][nt] = xe[rt][nt] + xq[rt][nt];                  // X = E[item] + q
        }
        const bool more = tile + stride < ntiles;
        if (more) tile_rows(tile + stride, rowN, itemN);
        TF_FENCE();
        f32x4 bv[4];
        // ---- ev0 = X W0 + b0 ----
        f32x4 ev[kTfRT][4];
        bias4(a.b0, bv);
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ev[rt][nt] = bv[nt];
        }
        product(0, act, ev);
        TF_FENCE();
        f32x4 n1[kTfRT][4];
        if (!(a.dbg & 1)) load_rows(a.nagg1, row, n1);
        else {
#pragma unroll
            for (int rt = 0; rt < kTfRT; ++rt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) n1[rt][nt] = n0[rt][nt];
        }
        TF_FENCE();
        // ---- item = bmix + ev0 Wmix[0:D] ... ; Z1 = ev0 + nagg0 ----
        f32x4 it[kTfRT][4];
        bias4(a.bmix, bv);
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) it[rt][nt] = bv[nt];
        }
        product(3, ev, it);
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) act[rt][nt] = ev[rt][nt] + n0[rt][nt];
        }
        // ---- out0 = relu(Z1 A0 + a0) ----
        bias4(a.a0, bv);
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ev[rt][nt] = bv[nt];
        }
        product(1, act, ev);
        TF_FENCE();
        f32x4 uo[kTfRT][4];
        if (!(a.dbg & 1)) load_rows(a.user_o, row, uo);
        else {
#pragma unroll
            for (int rt = 0; rt < kTfRT; ++rt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) uo[rt][nt] = n1[rt][nt];
        }
        TF_FENCE();
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[rt][nt][r] = fmaxf(ev[rt][nt][r], 0.f);
            }
        }
        product(4, ev, it);                                  // ... + out0 Wmix[D:2D]
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) act[rt][nt] = ev[rt][nt] + n1[rt][nt];       // Z2 = out0 + nagg1
        }
        TF_FENCE();
        // the next tile's rows: in flight under this tile's last two products (nagg0 / nagg1 have left their registers)
        if (more) {
            load_rows(E, itemN, xe);
            load_rows(a.q, rowN, xq);
        }
        TF_FENCE();
        // ---- out2 = relu(Z2 A1 + a1) ----
        bias4(a.a1, bv);
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ev[rt][nt] = bv[nt];
        }
        product(2, act, ev);
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[rt][nt][r] = fmaxf(ev[rt][nt][r], 0.f);
            }
        }
        product(5, ev, it);                                  // ... + out2 Wmix[2D:3D]
        // ---- score = user_o . item ; outputs ----
#pragma unroll
        for (int rt = 0; rt < kTfRT; ++rt) {
            float part = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(uo[rt][nt][r], it[rt][nt][r], part);
            }
            part = xor32_sum(xor16_sum(part));               // the four lane groups hold the four quarters of a pair's columns
            if (valid[rt] && !((a.dbg & 4) && part != 12345.f)) {
                if (a.item_emb) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        reinterpret_cast<float4*>(a.item_emb + row[rt] * D)[4 * nt + q16] = make_float4(it[rt][nt][0], it[rt][nt][1], it[rt][nt][2], it[rt][nt][3]);
                }
                if (q16 == 0) {
                    a.scores[row[rt]] = part;
                    if (a.sig) a.sig[row[rt]] = 1.f / (1.f + expf(-part));
                }
            }
        }
    }
}
#undef TF_FENCE

bool l2_tail_flash_applies(const TailArgs& a, int D) {
    static const bool on = getenv("MVIN_TAIL_FLASH") && atoi(getenv("MVIN_TAIL_FLASH")) == 1;
    const bool off = !on;      // opt-in: 0.305 vs 0.340 ms alone and -1.8 % on the one-stream line, but its 96 KB of LDS per CU shut the other stream's kernels out (two streams: 2.258 vs 2.245 ms)
    return !off && D == 64 && !a.table_bf16 && a.W0 && a.q && a.B > 0;
}

hipError_t launch_l2_tail_flash(const TailArgs& a, hipStream_t st) {
    const size_t lds = (size_t)6 * 64 * 64 * sizeof(float);
    static thread_local bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(l2_tail_flash_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    TailArgs b = a;
    {
        static const char* dbg = getenv("MVIN_TAIL_DBG");
        b.dbg = dbg ? atoi(dbg) : 0;
    }
    const int64_t ntiles = (a.B + 16 * kTfRT - 1) / (16 * kTfRT);
    const int64_t want = (ntiles + kTfWaves - 1) / kTfWaves;
    const int grid = (int)(want < 256 ? want : 256);
    l2_tail_flash_kernel<<<grid, kTfWaves * 64, lds, st>>>(b);
    return hipGetLastError();
}

}  // namespace mvin
