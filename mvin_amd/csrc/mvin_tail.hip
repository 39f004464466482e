// Everything of MVIN.aggregate_delta_whole above the fused two-level kernel, for the metric shape
// (wide_deep, n_mix_hop = 1, h_hop = 2: L = 2), in ONE launch per batch instead of four:
//   ev0    = (E[item] + q) W_0 + b_0                      user-oriented projection of level 0, model.py:270-283
//   out0   = relu((ev0  + nagg0) A_0 + a_0)               aggregator (0,0) at hop 0, aggregators.py:108-116
//   out2   = relu((out0 + nagg1) A_1 + a_1)               aggregator (1,0) at hop 0
//   item   = [ev0 | out0 | out2] Wmix + bmix              mix-hop combiner, model.py:310-315
//   score  = sum_d user_o[d] item[d] ; sigmoid            model.py:158-159
// nagg0 / nagg1 are the neighbor aggregates mvin_gather_attn_l2_fwd returns per pair.
// One workgroup of D/16 waves walks 32-row tiles; all six D x D weight blocks stay resident as B fragments of
// v_mfma_f32_16x16x4_f32 (6 * D/4 registers per wave), the five intermediates of a tile live in LDS
// (the contraction index is permuted so that a lane's A values of four steps are ONE 16-byte LDS read).  D in {16, 32, 64}.
//
// LDS image of a tile.  D < 64: row stride D + 4, MFMA step s of slot q16 stands for k = KS * q16 + s.
// D = 64 (round 5): ds_read_b128 is serviced in four NON-CONTIGUOUS 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, ... : MI355X_MICROARCH.md, LDS), so with the padded layout the A read of lanes 12-15 (slot q16 = 0, rows 12-15) and of
// lanes 24-27 (slot 1, rows 8-11) met on the same banks: one extra LDS cycle per group, four per read = ONE PER MFMA
// (rocprofv3 round 4: SQ_LDS_BANK_CONFLICT 12.58 M = the kernel's MFMA count, 38.6 % of its LDS cycles).  No padding fixes
// it (the two row sets of a group are an interval and its complement; a shift never maps the interval onto itself).  An
// XOR does: rows are 64 floats, the 16-byte slot `p` of row `r` lives at slot p ^ (r & 15), and the four slots lane group
// q16 reads are h(q16, j) = 12 (q16 & 1) ^ 4 (q16 >> 1) ^ j -- slots of q16 = 0 / 1 (and 2 / 3) differ by 12, and x ^ 12 maps
// the rows {4..11} onto themselves and {0-3, 12-15} onto themselves, so a service group's sixteen lanes hit sixteen different
// slots.  The B fragments take the same permutation (step s of slot q16 <-> k = 4 h(q16, s / 4) + s % 4).
#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__((D / 16) * 64) void l2_tail_kernel(TailArgs a) {
    constexpr bool SWZ = D == 64;
    constexpr int NT = D / 16, KS = D / 4, LD = SWZ ? D : D + 4, TM = 32, NTHR = NT * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                   // [TM][LD]  E[item] + q, then Z2 = out0 + nagg1
    float* sE0 = sX + TM * LD;          // ev0
    float* sZ1 = sE0 + TM * LD;         // ev0 + nagg0, then out2
    float* sO0 = sZ1 + TM * LD;         // out0
    float* sSc = sO0 + TM * LD;         // [NT][TM] per-slab partial scores

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q16 = lane >> 4, l16 = lane & 15;
    const int col = 16 * wave + l16;
    const bool proj = a.W0 != nullptr;

    // element (row, k) of a tile image; slot of the four values lane group q16 reads for steps 4j .. 4j+3
    auto at = [&](int row, int k) -> int {
        if constexpr (SWZ) return row * LD + ((((k >> 2) ^ row) & 15) << 2) + (k & 3);
        return row * LD + k;
    };
    auto hslot = [&](int j) -> int { return (12 * (q16 & 1)) ^ (4 * (q16 >> 1)) ^ j; };
    float bW0[KS], bA0[KS], bA1[KS], bC0[KS], bC1[KS], bC2[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = SWZ ? 4 * hslot(s >> 2) + (s & 3) : KS * q16 + s;
        const size_t o = (size_t)kk * D + col;
        bW0[s] = proj ? a.W0[o] : 0.f;
        bA0[s] = a.A0[o];
        bA1[s] = a.A1[o];
        bC0[s] = a.Wmix[o];
        bC1[s] = a.Wmix[(size_t)D * D + o];
        bC2[s] = a.Wmix[(size_t)2 * D * D + o];
    }
    const float b0v = (proj && a.b0) ? a.b0[col] : 0.f;
    const float a0v = a.a0 ? a.a0[col] : 0.f;
    const float a1v = a.a1 ? a.a1[col] : 0.f;
    const float bcv = a.bmix ? a.bmix[col] : 0.f;

    // acc[m] += A(tile rows of `src`) . B fragment
    auto mma = [&](const float* src, const float (&bf)[KS], f32x4 (&acc)[2]) {
#pragma unroll
        for (int s = 0; s < KS; s += 4) {
            float4 av[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                av[m] = *reinterpret_cast<const float4*>(src + (SWZ ? (16 * m + l16) * LD + ((hslot(s >> 2) ^ l16) << 2)
                                                                    : (16 * m + l16) * LD + KS * q16 + s));
            // the two row tiles' accumulator chains ALTERNATE: a dependent v_mfma_f32_16x16x4_f32 issues 40 cycles after its
            // predecessor, an independent one 32 (MI355X_MICROARCH.md) -- four in a row on one accumulator cost 24 cycles per step group
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].x, bf[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].x, bf[s], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].y, bf[s + 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].y, bf[s + 1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].z, bf[s + 2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].z, bf[s + 2], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].w, bf[s + 3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].w, bf[s + 3], acc[1], 0, 0, 0);
        }
    };
    const int64_t ntiles = (a.B + TM - 1) / TM;
    // Every global read of a tile is issued ONE TILE AHEAD, together (the rows were written by the previous kernels and
    // come from HBM / Infinity Cache: with the reads inside the phases each of the four phases exposed a full memory
    // latency per tile, 0.57 ms per 524 288 C3 pairs; hoisted to the tile start 0.31 ms).
    constexpr int XPT = TM * (D / 4) / NTHR;                // X chunks (4 floats) per thread: 2
    static_assert(TM * (D / 4) % NTHR == 0, "tile chunks must divide among the threads");
    struct TileIn {
        float4 x[XPT];                                      // E[item] + q
        float n0[2][4], n1[2][4], uo[2][4];
    };
    auto load_tile = [&](int64_t tile, TileIn& t) {
        const int64_t r0 = tile * TM;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / (D / 4), c = idx - row * (D / 4);
            const int64_t r = r0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.B) {
                int64_t it = a.items64 ? a.items64[r] : (int64_t)a.items32[r];
                if (a.n_entity > 0) it = (int64_t)min((uint64_t)it, (uint64_t)(a.n_entity - 1));      // clamped into the table
                v = load_row4(a.E, a.table_bf16, it, D, c);
                if (proj) {
                    const float4 qv = reinterpret_cast<const float4*>(a.q + r * D)[c];
                    v.x += qv.x;
                    v.y += qv.y;
                    v.z += qv.z;
                    v.w += qv.w;
                }
            }
            t.x[i] = v;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = r0 + 16 * m + 4 * q16 + r;
                const bool v = gr < a.B;
                t.n0[m][r] = v ? a.nagg0[gr * D + col] : 0.f;
                t.n1[m][r] = v ? a.nagg1[gr * D + col] : 0.f;
                t.uo[m][r] = v ? a.user_o[gr * D + col] : 0.f;
            }
        }
    };
    TileIn nxt;
    if ((int64_t)blockIdx.x < ntiles) load_tile(blockIdx.x, nxt);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TM;
        const TileIn cur = nxt;
        __syncthreads();                                    // previous tile's buffers consumed
        // ---- X = E[item] + q ----
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / (D / 4), c = idx - row * (D / 4);
            float* dst = sX + at(row, 4 * c);
            *reinterpret_cast<float4*>(dst) = cur.x[i];
        }
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x, nxt);     // in flight under this tile's phases
        __syncthreads();
        // ---- ev0 ; Z1 = ev0 + nagg0 ----
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (proj) mma(sX, bW0, acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * m + 4 * q16 + r;
                    const float e0 = proj ? acc[m][r] + b0v : sX[at(row, col)];
                    sE0[at(row, col)] = e0;
                    sZ1[at(row, col)] = e0 + cur.n0[m][r];
                }
            }
        }
        __syncthreads();
        // ---- out0 = relu(Z1 A0 + a0) ; Z2 = out0 + nagg1 (into sX) ----
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mma(sZ1, bA0, acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * m + 4 * q16 + r;
                    const float o0 = fmaxf(acc[m][r] + a0v, 0.f);
                    sO0[at(row, col)] = o0;
                    sX[at(row, col)] = o0 + cur.n1[m][r];
                }
            }
        }
        __syncthreads();
        // ---- out2 = relu(Z2 A1 + a1) (into sZ1) ----
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mma(sX, bA1, acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sZ1[at(16 * m + 4 * q16 + r, col)] = fmaxf(acc[m][r] + a1v, 0.f);
            }
        }
        __syncthreads();
        // ---- item = [ev0 | out0 | out2] Wmix + bmix ; score ----
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mma(sE0, bC0, acc);
            mma(sO0, bC1, acc);
            mma(sZ1, bC2, acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * m + 4 * q16 + r;
                    const int64_t gr = r0 + row;
                    const float v = acc[m][r] + bcv;
                    if (gr < a.B && a.item_emb) a.item_emb[gr * D + col] = v;
                    float part = cur.uo[m][r] * v;                  // rows past B hold zeros
                    part = group_sum(part, 4);              // the slab's 16 columns of this row
                    if (l16 == 0) sSc[wave * TM + row] = part;
                }
            }
        }
        __syncthreads();
        if (tid < TM && r0 + tid < a.B) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NT; ++w) s += sSc[w * TM + tid];      // fixed order: deterministic
            a.scores[r0 + tid] = s;
            if (a.sig) a.sig[r0 + tid] = 1.f / (1.f + expf(-s));
        }
    }
}

// ---- folded-tail form (mvin_score_l2_folded_fwd), dim 64: the pair kernel (mvin_fused_agg.hip) has already left out0 = relu((ev0 +
//      nagg0) A0 + a0) and Z2 = out0 + nagg1 -- through per-entity tables both are a gathered row plus a product of the query -- and
//      ev0 itself only enters the combiner, where it is a gathered row plus a product of the query too:
//        out2 = relu(Z2 A1 + a1) ;  item = M0[x] + q Wqm + out0 Wm1 + out2 Wm2 + bm            (M0 = E W0 Wm0, Wqm = W0 Wm0)
//      FOUR products per pair instead of six, two barriers per tile instead of four; the tile machinery is l2_tail_kernel<64>'s
//      (XOR-swizzled images, B fragments resident, every global read issued one tile ahead).
__global__ __launch_bounds__(256) void l2_tail_fold_kernel(TailFoldArgs a) {
    constexpr int D = 64, NT = 4, KS = 16, LD = 64, TM = 32, NTHR = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                   // [TM][LD]  q
    float* sZ2 = sQ + TM * LD;          // Z2
    float* sO0 = sZ2 + TM * LD;         // out0
    float* sO2 = sO0 + TM * LD;         // out2
    float* sSc = sO2 + TM * LD;         // [NT][TM] per-slab partial scores

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q16 = lane >> 4, l16 = lane & 15;
    const int col = 16 * wave + l16;
    auto at = [&](int row, int k) -> int { return row * LD + ((((k >> 2) ^ row) & 15) << 2) + (k & 3); };
    auto hslot = [&](int j) -> int { return (12 * (q16 & 1)) ^ (4 * (q16 >> 1)) ^ j; };
    float bQm[KS], bA1[KS], bC1[KS], bC2[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = 4 * hslot(s >> 2) + (s & 3);
        const size_t o = (size_t)kk * D + col;
        bQm[s] = a.Wqm[o];
        bA1[s] = a.A1[o];
        bC1[s] = a.Wmix[(size_t)D * D + o];
        bC2[s] = a.Wmix[(size_t)2 * D * D + o];
    }
    const float a1v = a.a1 ? a.a1[col] : 0.f;
    const float bmv = a.bm[col];
    auto mma = [&](const float* src, const float (&bf)[KS], f32x4 (&acc)[2]) {
#pragma unroll
        for (int s = 0; s < KS; s += 4) {
            float4 av[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) av[m] = *reinterpret_cast<const float4*>(src + (16 * m + l16) * LD + ((hslot(s >> 2) ^ l16) << 2));
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].x, bf[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].x, bf[s], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].y, bf[s + 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].y, bf[s + 1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].z, bf[s + 2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].z, bf[s + 2], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].w, bf[s + 3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1].w, bf[s + 3], acc[1], 0, 0, 0);
        }
    };
    const int64_t ntiles = (a.B + TM - 1) / TM;
    constexpr int XPT = TM * (D / 4) / NTHR;                // row chunks (4 floats) per thread and image: 2
    // (plain arrays, copied element by element: as members of one struct the three float4 arrays stayed in scratch memory)
    f32x4 nq[XPT], nz[XPT], no[XPT];      // (native vectors: arrays of HIP's float4 struct stayed in scratch memory here)
    float nm0[2][4], nuo[2][4];
    auto load_tile = [&](int64_t tile) {
        const int64_t r0 = tile * TM;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / (D / 4), c = idx - row * (D / 4);
            const int64_t r = min(r0 + row, a.B - 1);       // (rows past B: the last row again, never stored)
            nq[i] = reinterpret_cast<const f32x4*>(a.q + r * D)[c];
            nz[i] = reinterpret_cast<const f32x4*>(a.z2 + r * D)[c];
            no[i] = reinterpret_cast<const f32x4*>(a.out0 + r * D)[c];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = min(r0 + 16 * m + 4 * q16 + r, a.B - 1);
                int64_t it = a.items64 ? a.items64[gr] : (int64_t)a.items32[gr];
                it = (int64_t)min((uint64_t)it, (uint64_t)(a.n_entity - 1));      // clamped into the table
                nm0[m][r] = a.M0[it * D + col];
                nuo[m][r] = a.user_o[gr * D + col];
            }
        }
    };
    if ((int64_t)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TM;
        f32x4 cq[XPT], cz[XPT], co[XPT];
        float cm0[2][4], cuo[2][4];
#pragma unroll
        for (int i = 0; i < XPT; ++i) cq[i] = nq[i], cz[i] = nz[i], co[i] = no[i];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cm0[m][r] = nm0[m][r], cuo[m][r] = nuo[m][r];
        }
        __syncthreads();                                    // previous tile's buffers consumed
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / (D / 4), c = idx - row * (D / 4);
            *reinterpret_cast<f32x4*>(sQ + at(row, 4 * c)) = cq[i];
            *reinterpret_cast<f32x4*>(sZ2 + at(row, 4 * c)) = cz[i];
            *reinterpret_cast<f32x4*>(sO0 + at(row, 4 * c)) = co[i];
        }
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);          // in flight under this tile's phases
        __syncthreads();
        // ---- out2 = relu(Z2 A1 + a1) ; the first two blocks of the combiner meanwhile (they do not need it) ----
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        {
            f32x4 a2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mma(sZ2, bA1, a2);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sO2[at(16 * m + 4 * q16 + r, col)] = fmaxf(a2[m][r] + a1v, 0.f);
            }
            mma(sQ, bQm, acc);
            mma(sO0, bC1, acc);
        }
        __syncthreads();
        // ---- item = M0[x] + q Wqm + out0 Wm1 + out2 Wm2 + bm ; score ----
        mma(sO2, bC2, acc);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * m + 4 * q16 + r;
                const int64_t gr = r0 + row;
                const float v = acc[m][r] + cm0[m][r] + bmv;
                if (gr < a.B && a.item_emb) a.item_emb[gr * D + col] = v;
                float part = cuo[m][r] * v;                     // (rows past B: never stored)
                part = group_sum(part, 4);                      // the slab's 16 columns of this row
                if (l16 == 0) sSc[wave * TM + row] = part;
            }
        }
        __syncthreads();
        if (tid < TM && r0 + tid < a.B) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NT; ++w) s += sSc[w * TM + tid];      // fixed order: deterministic
            a.scores[r0 + tid] = s;
            if (a.sig) a.sig[r0 + tid] = 1.f / (1.f + expf(-s));
        }
    }
}

hipError_t launch_l2_tail_fold(const TailFoldArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(4 * 32 * 64 + 4 * 32) * 4;
    const int64_t ntiles = (a.B + 31) / 32;
    const int64_t cap = 256 * 4;
    l2_tail_fold_kernel<<<(int)(ntiles < cap ? ntiles : cap), 256, lds, st>>>(a);
    return hipGetLastError();
}

bool l2_tail_supported(int D) { return D == 16 || D == 32 || D == 64; }

template <int D>
static hipError_t launch_tail_d(const TailArgs& a, hipStream_t st) {
    constexpr int NT = D / 16;
    const size_t lds = (size_t)(4 * 32 * (D == 64 ? D : D + 4) + NT * 32) * 4;
    const int64_t ntiles = (a.B + 31) / 32;
    const int64_t cap = 256 * (D == 64 ? 4 : 8);
    l2_tail_kernel<D><<<(int)(ntiles < cap ? ntiles : cap), NT * 64, lds, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_l2_tail(const TailArgs& a, int D, hipStream_t st) {
    if (l2_tail_flash_applies(a, D)) return launch_l2_tail_flash(a, st);      // dim 64, fp32 table, projection on: mvin_tail_flash.hip
    switch (D) {
        case 16: return launch_tail_d<16>(a, st);
        case 32: return launch_tail_d<32>(a, st);
        case 64: return launch_tail_d<64>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
