// Fused two-level gather + attention kernel for gfx950 (MI355X) -- the hot kernel.
//
// Replaces, for the two deepest levels of the tree at once, the reference's
//   tf.gather(adj_*) (model.py:251-252) -> embedding_lookup (:267-268) -> user-oriented
//   projection (:270-283) -> SumAggregator_urh_matrix at hop L-1 (aggregators.py:98-146)
// and the neighbor mixes of the two aggregator applications that consume those levels at hop
// L-2 (i = 0: neighbors = projected level-(L-1) rows; i = 1: neighbors = hop-(L-1) outputs).
//
// One workgroup owns one PARENT node (a level-(L-2) node; for L = 2 the pair's item itself):
//   prologue  : parent's adjacency row -> child ids x1[n], attention weights p0 (aggregator
//               (0,.)) and p1 (aggregator (1,.)) over its K children           (wave 0)
//   per tile of 32 children (16 when K <= 16, with D/16 waves per workgroup):
//     phase A : each wave gathers the K grandchild rows of its children: adjacency rows as
//               int4 (K/4 lanes per child), softmax over K by xor-shuffles inside the lane
//               group, rows as 16-byte loads (D/4 lanes per row, 8 loads in flight per lane),
//               S' = (1/K) sum_k p_k E[y_k] and the raw child row E[x1] -> LDS tile.
//               Two lane-group mappings: one group per child (GPC: the sum stays inside the
//               group) or the groups of a wave striding over one child's k (xor-reduced);
//               launch_l2_pick chooses per shape from measurements
//     phase B : MFMA (v_mfma_f32_16x16x4_f32, weights resident in VGPRs as B fragments):
//               self1 = E[x1].W1 + c1 ; Z = self1 + S'.W2 + (psum/K) c2 ; Z -> LDS ;
//               (c_e = q_b.W_e + b_e is formed per pair from the same B fragments)
//               nagg0 += sum_n p0[n] self1[n]   (from the accumulator registers)
//     phase C : out1 = relu(Z.A0 + a0) (MFMA) ; nagg1 += sum_n p1[n] out1[n]
//   epilogue  : nagg0/K, nagg1/K -> HBM (2 x D floats per parent).
// Nothing of size K^L or K^(L-1) is ever written to memory.
//
// Supported: D in {16, 32, 64, 128}; K a power of two in [4, 256]; fp32.
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D, int NW, int TMV>
struct FusedGeom {
    static constexpr int TM = TMV;                      // children (rows) per tile: 32, or 16 when K <= 16
    static constexpr int RT = TM / 16;                  // 16-row MFMA tiles per tile
    static constexpr int NT = D / 16;                   // 16-column MFMA tiles
    static constexpr int MTW = (NW == NT) ? RT : 1;     // 16-row MFMA tiles per wave
    static constexpr int NPW = TM / NW;                 // children per wave per tile
    static constexpr int LPR = D / 4;                   // lanes per table row (float4 each)
    static constexpr int RPW = kWave / LPR;             // rows per wave-instruction
    static constexpr int LDA = 2 * D + 2;               // LDS row stride: conflict-free A-fragment reads
    static constexpr int LDZ = D + 2;
    static constexpr int KS = D / 4;                    // MFMA k-steps per DxD matrix
};

// KIT > 0: software-pipelined variant.  The three dependent id fetches that precede a parent's
// row gathers (parent id -> parent adjacency row -> its children's adjacency rows) are issued
// one parent AHEAD and land while the current parent is being gathered / multiplied:
//   top of iteration p        : wave 0 issues the adjacency-row loads of parent p+G (registers)
//   end of wave 0's phase A   : ... turns them into x1/p0/p1 of p+G in the OTHER LDS buffer
//   after the post-A barrier  : every wave issues the int4 adjacency chunks of its children of
//                               p+G (KIT x 2 int4 registers), consumed at the top of the next
//                               iteration without waiting
// KIT = number of int4 chunk iterations per wave per tile (= NPW*K/256 <= 2); KIT = 0 keeps the
// unpipelined flow (any K <= 256).
template <int D, int NW, int KIT, bool BF, int TMV, bool GPC>
__global__ __launch_bounds__(NW * 64) void gather_attn_l2_kernel(FusedL2Args a) {
    using G = FusedGeom<D, NW, TMV>;
    constexpr int kTM = G::TM;                          // shadows the 32-row default of mvin_common.h
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NBUF = KIT > 0 ? 2 : 1;
    const int K = a.K;
    const int ntile = (K + kTM - 1) / kTM;
    const int Kpad = ntile * kTM;
    float* sA = smem;                                   // [32][LDA]  {E[x1] raw | S'}
    float* sZ = sA + kTM * G::LDA;                      // [32][LDZ]
    float* sP0b = sZ + kTM * G::LDZ;                    // [NBUF][Kpad]
    float* sP1b = sP0b + NBUF * Kpad;                   // [NBUF][Kpad]
    float* sN0 = sP1b + NBUF * Kpad;                    // [D]
    float* sN1 = sN0 + D;                               // [D]
    float* sT0 = sN1 + D;                               // [nR]
    float* sT1 = sT0 + a.nR;                            // [nR]
    int* sX1b = reinterpret_cast<int*>(sT1 + a.nR);     // [NBUF][Kpad]
    int2* sYP = reinterpret_cast<int2*>(sX1b + NBUF * Kpad);  // [NW][NPW][K]; offset is even: 8-B aligned

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // elements of a table row per lane: 4 (16-byte fp32 / 8-byte bf16 loads), or 8 bf16 (16-byte loads)
    // at D = 128, where halving the load instructions pays (C5: +49 %); at D <= 64 the narrower rows
    // would leave too few rows per lane group in flight (measured slower)
    constexpr bool WIDE = BF && D == 128 && TMV == 32;
    constexpr int EPL = WIDE ? 8 : 4;
    constexpr int LPRX = D / EPL, RPWX = kWave / LPRX;
    const int g = lane / LPRX, c = lane % LPRX;
    const int q16 = lane >> 4, l16 = lane & 15;
    // dense-phase tile ownership
    const int nt = (NW == G::NT) ? wave : (wave % G::NT);
    const int mt0 = (NW == G::NT) ? 0 : (wave / G::NT);
    const bool dense = mt0 < G::RT;
    const int col = 16 * nt + l16;
    const bool has_proj = a.W1 != nullptr;
    const bool has_att0 = a.t0 != nullptr, has_att1 = a.t1 != nullptr;
    const float invK = 1.f / (float)K;
    const float c2scale = has_att0 ? invK : 1.f;        // (sum_k p_k)/K

    // ---- weights -> B fragments, resident for the whole (persistent) workgroup ----
    float bW1[G::KS], bW2[G::KS], bA0[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
        const int kk = 4 * s + q16;
        bW1[s] = (dense && has_proj) ? a.W1[kk * D + col] : 0.f;
        bW2[s] = (dense && has_proj) ? a.W2[kk * D + col] : 0.f;
        bA0[s] = dense ? a.A0[kk * D + col] : 0.f;
    }
    const float a0v = (dense && a.a0) ? a.a0[col] : 0.f;
    const float b1v = (dense && has_proj && a.b1) ? a.b1[col] : 0.f;
    const float b2v = (dense && has_proj && a.b2) ? a.b2[col] : 0.f;
    for (int i = tid; i < a.nR; i += NW * 64) {
        sT0[i] = has_att0 ? a.t0[i] : 0.f;
        sT1[i] = has_att1 ? a.t1[i] : 0.f;
    }

    // entity table through a buffer descriptor when it is < 4 GiB (32-bit byte offsets)
    const bool buf32 = !BF && a.table_bytes < (1ull << 32);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.table), 0, buf32 ? (int)a.table_bytes : 0, 0x00020000);
    const unsigned c16 = (unsigned)c * 16u;
    // bf16 row chunk of this lane: 8 elements, widened to fp32 exactly
    auto load8 = [&](int id, float4& lo, float4& hi) {
        const uint4 raw = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.table) + (int64_t)id * D)[c];
        lo = bf16x4_to_f32(make_uint2(raw.x, raw.y));
        hi = bf16x4_to_f32(make_uint2(raw.z, raw.w));
    };
    // this lane's EPL floats of an LDS tile row half (float2 stores: rows are 8-byte aligned)
    auto put = [&](float* dst, float4 lo, float4 hi) {
        float* q = dst + EPL * c;
        *reinterpret_cast<float2*>(q) = make_float2(lo.x, lo.y);
        *reinterpret_cast<float2*>(q + 2) = make_float2(lo.z, lo.w);
        if constexpr (WIDE) {
            *reinterpret_cast<float2*>(q + 4) = make_float2(hi.x, hi.y);
            *reinterpret_cast<float2*>(q + 6) = make_float2(hi.z, hi.w);
        }
    };
    const int ypld = GPC ? K + 1 : K;           // sYP row stride (odd in the 16-row variant: the lane
                                                        // groups read different rows at once)
    const int lpn = 1 << a.lpn_log2;                    // lanes per child adjacency row (K/4)
    const int npi = kWave >> a.lpn_log2;                // children per wave-instruction

    // parent adjacency row -> registers (lane n handles children n, n+64, ...; K <= 256)
    auto parent_load = [&](int64_t x0, int (&xs)[4], int (&rr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            xs[i] = 0;
            rr[i] = 0;
            if (n < K) {
                xs[i] = a.adj_e[x0 * K + n];
                if (has_att0 || has_att1) rr[i] = a.adj_r[x0 * K + n];
            }
        }
    };
    // ... -> child ids + attention weights of aggregator (0,.) / (1,.) over the K children
    auto parent_store = [&](int64_t pp, const int (&xs)[4], const int (&rr)[4], int* sX1w, float* sP0w, float* sP1w) {
        float s0[4], s1[4];
        float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            s0[i] = s1[i] = -INFINITY;
            if (n < K) {
                s0[i] = has_att0 ? sT0[rr[i]] : 0.f;
                s1[i] = has_att1 ? sT1[rr[i]] : 0.f;
                m0 = fmaxf(m0, s0[i]);
                m1 = fmaxf(m1, s1[i]);
            }
        }
        m0 = wave_max(m0);
        m1 = wave_max(m1);
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            if (n < K) {
                s0[i] = has_att0 ? expf(s0[i] - m0) : 1.f;
                s1[i] = has_att1 ? expf(s1[i] - m1) : 1.f;
                z0 += s0[i];
                z1 += s1[i];
            }
        }
        z0 = wave_sum(z0);
        z1 = wave_sum(z1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = lane + 64 * i;
            if (n < Kpad) {
                const bool in = n < K;
                const float p0 = in ? (has_att0 ? s0[i] / z0 : 1.f) : 0.f;
                const float p1 = in ? (has_att1 ? s1[i] / z1 : 1.f) : 0.f;
                sX1w[n] = xs[i];
                sP0w[n] = p0;
                sP1w[n] = p1;
                if (in && a.probs_parent && has_att0) a.probs_parent[pp * K + n] = p0;
            }
        }
    };
    // int4 chunk `it` of the adjacency rows of this wave's children in `tile`
    auto chunk_load = [&](const int* sX1r, int tile, int it, int4& ye, int4& re) {
        const int nl = it * npi + (lane >> a.lpn_log2);
        const int n = tile * kTM + wave * G::NPW + nl;
        ye = make_int4(0, 0, 0, 0);
        re = make_int4(0, 0, 0, 0);
        if (nl < G::NPW && n < K) {
            const int64_t xb = (int64_t)sX1r[n] * K + 4 * (lane & (lpn - 1));
            ye = *reinterpret_cast<const int4*>(a.adj_e + xb);
            if (has_att0) re = *reinterpret_cast<const int4*>(a.adj_r + xb);
        }
    };

    constexpr int KITR = KIT > 0 ? KIT : 1;
    int4 pye[KITR], pre[KITR];          // prefetched chunks of (next parent, tile 0)
    int64_t x0_next = 0;                // entity id of the parent after the current one
    int buf = 0;
    if (KIT > 0) {                      // pipeline fill for this workgroup's first parent
        const int64_t p0i = blockIdx.x;
        __syncthreads();                // sT0/sT1 visible
        if (wave == 0 && p0i < a.P) {
            int xs[4], rr[4];
            parent_load(fused_parent_id(a, p0i), xs, rr);
            parent_store(p0i, xs, rr, sX1b, sP0b, sP1b);
            if (p0i + gridDim.x < a.P) x0_next = fused_parent_id(a, p0i + gridDim.x);
        }
        __syncthreads();
        if (p0i < a.P) {
#pragma unroll
            for (int it = 0; it < KITR; ++it) chunk_load(sX1b, 0, it, pye[it], pre[it]);
        }
    }

    for (int64_t p = blockIdx.x; p < a.P; p += gridDim.x) {
        const int64_t b = p / a.parents_per_pair;
        const int64_t pn = p + gridDim.x;
        const bool has_next = KIT > 0 && pn < a.P;
        int* sX1 = sX1b + buf * Kpad;
        float* sP0 = sP0b + buf * Kpad;
        float* sP1 = sP1b + buf * Kpad;
        int nxs[4], nrr[4];             // next parent's adjacency row (wave 0, pipelined variant)
        if (KIT == 0) {
            __syncthreads();  // previous parent fully consumed (sP*, sN*, sX1)
            if (wave == 0) {
                int xs[4], rr[4];
                parent_load(fused_parent_id(a, p), xs, rr);
                parent_store(p, xs, rr, sX1, sP0, sP1);
            }
        } else if (wave == 0 && has_next) {
            parent_load(x0_next, nxs, nrr);                       // lands during this parent's phase A
            x0_next = pn + gridDim.x < a.P ? fused_parent_id(a, pn + gridDim.x) : 0;
        }
        if (tid < D) {
            sN0[tid] = 0.f;
            sN1[tid] = 0.f;
        }
        if (KIT == 0) __syncthreads();

        float nacc0 = 0.f, nacc1 = 0.f;
        // c_e[col] = q_b . W_e[:, col] + b_e[col] (model.py:277-279 applied to the broadcast query):
        // the W columns are already in registers as B fragments, so this is KS FMAs + 2 shuffles
        float c1v = 0.f, c2v = 0.f;
        if (dense && has_proj) {
            const float* qb = a.q + b * D;
#pragma unroll
            for (int s = 0; s < G::KS; ++s) {
                const float qv = qb[4 * s + q16];
                c1v = fmaf(qv, bW1[s], c1v);
                c2v = fmaf(qv, bW2[s], c2v);
            }
            c1v += __shfl_xor(c1v, 16, kWave);
            c1v += __shfl_xor(c1v, 32, kWave);
            c2v += __shfl_xor(c2v, 16, kWave);
            c2v += __shfl_xor(c2v, 32, kWave);
            c1v += b1v;
            c2v = (c2v + b2v) * c2scale;
        }

        for (int tile = 0; tile < ntile; ++tile) {
            // ---------------- phase A: ids, softmax, row gather ----------------
            const int node0 = tile * kTM + wave * G::NPW;
            int2* ypw = sYP + (size_t)wave * G::NPW * ypld;
#pragma unroll
            for (int it = 0; it < (KIT > 0 ? KITR : 8); ++it) {
                if (it * npi >= G::NPW) break;
                const int nl = it * npi + (lane >> a.lpn_log2);
                const int ch = lane & (lpn - 1);
                const int n = node0 + nl;
                const bool valid = nl < G::NPW && n < K;
                int4 ye, re;
                if (KIT > 0 && tile == 0 && it < KITR) {
                    ye = pye[it < KITR ? it : 0];
                    re = pre[it < KITR ? it : 0];
                } else {
                    chunk_load(sX1, tile, it, ye, re);
                }
                float sc0 = 0.f, sc1 = 0.f, sc2 = 0.f, sc3 = 0.f;
                if (has_att0 && valid) {
                    sc0 = sT0[re.x];
                    sc1 = sT0[re.y];
                    sc2 = sT0[re.z];
                    sc3 = sT0[re.w];
                }
                float m = group_max(fmaxf(fmaxf(sc0, sc1), fmaxf(sc2, sc3)), a.lpn_log2);
                float e0 = 1.f, e1 = 1.f, e2 = 1.f, e3 = 1.f;
                if (has_att0) {
                    e0 = expf(sc0 - m);
                    e1 = expf(sc1 - m);
                    e2 = expf(sc2 - m);
                    e3 = expf(sc3 - m);
                }
                const float z = group_sum((e0 + e1) + (e2 + e3), a.lpn_log2);
                if (valid) {
                    if (has_att0) {
                        e0 /= z;
                        e1 /= z;
                        e2 /= z;
                        e3 /= z;
                        if (a.probs_child)
                            *reinterpret_cast<float4*>(a.probs_child + ((p * K + n) * K + 4 * ch)) =
                                make_float4(e0, e1, e2, e3);
                    }
                    int2* dst = ypw + nl * ypld + 4 * ch;
                    dst[0] = make_int2(ye.x, __float_as_int(e0 * invK));
                    dst[1] = make_int2(ye.y, __float_as_int(e1 * invK));
                    dst[2] = make_int2(ye.z, __float_as_int(e2 * invK));
                    dst[3] = make_int2(ye.w, __float_as_int(e3 * invK));
                }
            }
            if constexpr (GPC) {
                // one lane group per child: its K rows are K independent loads in flight and the
                // weighted sum never leaves the group (no cross-group reduction).  In the 16-row
                // variant there are exactly as many lane groups per wave as children per wave (256/D).
                static_assert(G::NPW % RPWX == 0, "children per wave must be a multiple of the lane groups");
                auto row4 = [&](int id) -> float4 {
                    if (BF)
                        return bf16x4_to_f32(reinterpret_cast<const uint2*>(
                            reinterpret_cast<const uint16_t*>(a.table) + (int64_t)id * D)[c]);
                    if (buf32) {
                        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
                            rsrc, ((unsigned)id * (unsigned)(D * 4)) + c16, 0, 0);
                        return make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]),
                                           __uint_as_float(raw[2]), __uint_as_float(raw[3]));
                    }
                    return reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.table) + (int64_t)id * D)[c];
                };
#pragma unroll
                for (int j = 0; j < G::NPW / RPWX; ++j) {
                    const int nl = j * RPWX + g;
                    const int n = node0 + nl;
                    float* arow = sA + (wave * G::NPW + nl) * G::LDA;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc;
                    float4 sv = acc, sv1 = acc;
                    if (n < K) {
                        const int2* yp = ypw + nl * ypld;
                        if constexpr (WIDE) {
#pragma unroll 8
                            for (int k = 0; k < K; ++k) {
                                const int2 e = yp[k];
                                float4 lo, hi;
                                load8(e.x, lo, hi);
                                acc = f4_fma(__int_as_float(e.y), lo, acc);
                                acc1 = f4_fma(__int_as_float(e.y), hi, acc1);
                            }
                            load8(sX1[n], sv, sv1);
                        } else {
#pragma unroll 8
                            for (int k = 0; k < K; ++k) {
                                const int2 e = yp[k];
                                acc = f4_fma(__int_as_float(e.y), row4(e.x), acc);
                            }
                            sv = row4(sX1[n]);
                        }
                    }
                    put(arow, sv, sv1);
                    put(arow + D, acc, acc1);
                }
            } else
            for (int nl = 0; nl < G::NPW; ++nl) {
                const int n = node0 + nl;
                const int row = wave * G::NPW + nl;
                float* arow = sA + row * G::LDA;
                if (n >= K) {  // padding child: finite zeros (its p0/p1 are 0)
                    if (g == 0) {
                        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                        put(arow, z, z);
                        put(arow + D, z, z);
                    }
                    continue;
                }
                const int2* yp = ypw + nl * K;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc;
                float4 sv = acc, sv1 = acc;
                if constexpr (WIDE) {
                    // bf16 table, D = 128: 16-byte lane loads (8 elements), widened to fp32 exactly
#pragma unroll 8
                    for (int k = g; k < K; k += RPWX) {
                        const int2 e = yp[k];
                        float4 lo, hi;
                        load8(e.x, lo, hi);
                        acc = f4_fma(__int_as_float(e.y), lo, acc);
                        acc1 = f4_fma(__int_as_float(e.y), hi, acc1);
                    }
                    if (g == 0) load8(sX1[n], sv, sv1);
                    acc1 = group_xor_sum(acc1, LPRX);
                } else if (BF) {
                    // bf16 table: 8-byte lane loads (4 elements), widened to fp32 exactly
                    const uint16_t* tb = reinterpret_cast<const uint16_t*>(a.table);
#pragma unroll 8
                    for (int k = g; k < K; k += RPWX) {
                        const int2 e = yp[k];
                        const float4 v = bf16x4_to_f32(reinterpret_cast<const uint2*>(tb + (int64_t)e.x * D)[c]);
                        acc = f4_fma(__int_as_float(e.y), v, acc);
                    }
                    if (g == 0) sv = bf16x4_to_f32(reinterpret_cast<const uint2*>(tb + (int64_t)sX1[n] * D)[c]);
                } else if (buf32) {
                    // 32-bit row offsets through a buffer descriptor: one VALU per row address
                    // instead of a 64-bit shift+add chain; packed FMAs (2 f32 per issue)
                    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll 8
                    for (int k = g; k < K; k += RPWX) {
                        const int2 e = yp[k];
                        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
                            rsrc, ((unsigned)e.x * (unsigned)(D * 4)) + c16, 0, 0);
                        const float w = __int_as_float(e.y);
                        const f32x2 ww = {w, w};
                        const f32x2 v01 = {__uint_as_float(raw[0]), __uint_as_float(raw[1])};
                        const f32x2 v23 = {__uint_as_float(raw[2]), __uint_as_float(raw[3])};
                        a01 = __builtin_elementwise_fma(ww, v01, a01);
                        a23 = __builtin_elementwise_fma(ww, v23, a23);
                    }
                    acc = make_float4(a01[0], a01[1], a23[0], a23[1]);
                    if (g == 0) {
                        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(
                            rsrc, ((unsigned)sX1[n] * (unsigned)(D * 4)) + c16, 0, 0);
                        sv = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]),
                                         __uint_as_float(raw[3]));
                    }
                } else {   // tables >= 4 GiB: 64-bit global addressing
                    const float* tf = reinterpret_cast<const float*>(a.table);
#pragma unroll 8
                    for (int k = g; k < K; k += RPWX) {
                        const int2 e = yp[k];
                        const float4 v = reinterpret_cast<const float4*>(tf + (int64_t)e.x * D)[c];
                        acc = f4_fma(__int_as_float(e.y), v, acc);
                    }
                    if (g == 0) sv = reinterpret_cast<const float4*>(tf + (int64_t)sX1[n] * D)[c];
                }
                acc = group_xor_sum(acc, LPRX);
                if (g == 0) {
                    put(arow, sv, sv1);
                    put(arow + D, acc, acc1);
                }
            }
            if (KIT > 0 && tile == 0 && wave == 0 && has_next)
                parent_store(pn, nxs, nrr, sX1b + (buf ^ 1) * Kpad, sP0b + (buf ^ 1) * Kpad, sP1b + (buf ^ 1) * Kpad);
            __syncthreads();
            if (KIT > 0 && tile == ntile - 1 && has_next) {
#pragma unroll
                for (int it = 0; it < KITR; ++it) chunk_load(sX1b + (buf ^ 1) * Kpad, 0, it, pye[it], pre[it]);
            }
            // ---------------- phase B: projections (MFMA), Z -> LDS, nagg0 ----------------
            if (dense) {
                f32x4 accE[G::MTW], accS[G::MTW];
#pragma unroll
                for (int m = 0; m < G::MTW; ++m) {
                    accE[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    accS[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (has_proj) {
#pragma unroll
                    for (int s = 0; s < G::KS; ++s) {
#pragma unroll
                        for (int m = 0; m < G::MTW; ++m) {
                            const float* ar = sA + (16 * (mt0 + m) + l16) * G::LDA + 4 * s + q16;
                            accE[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[0], bW1[s], accE[m], 0, 0, 0);
                            accS[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[D], bW2[s], accS[m], 0, 0, 0);
                        }
                    }
                }
                float part = 0.f;
#pragma unroll
                for (int m = 0; m < G::MTW; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (mt0 + m) + 4 * q16 + r;
                        float s1v, zv;
                        if (has_proj) {
                            s1v = accE[m][r] + c1v;
                            zv = s1v + (accS[m][r] + c2v);
                        } else {
                            s1v = sA[row * G::LDA + col];
                            zv = s1v + sA[row * G::LDA + D + col];
                        }
                        part = fmaf(sP0[tile * kTM + row], s1v, part);
                        sZ[row * G::LDZ + col] = zv;
                    }
                }
                part += __shfl_xor(part, 16, kWave);
                part += __shfl_xor(part, 32, kWave);
                nacc0 += part;
            }
            __syncthreads();
            // ---------------- phase C: aggregator dense + relu (MFMA), nagg1 ----------------
            if (dense) {
                f32x4 acc2[G::MTW];
#pragma unroll
                for (int m = 0; m < G::MTW; ++m) acc2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < G::KS; ++s) {
#pragma unroll
                    for (int m = 0; m < G::MTW; ++m) {
                        const float az = sZ[(16 * (mt0 + m) + l16) * G::LDZ + 4 * s + q16];
                        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bA0[s], acc2[m], 0, 0, 0);
                    }
                }
                float part = 0.f;
#pragma unroll
                for (int m = 0; m < G::MTW; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (mt0 + m) + 4 * q16 + r;
                        const float o = fmaxf(acc2[m][r] + a0v, 0.f);
                        part = fmaf(sP1[tile * kTM + row], o, part);
                    }
                }
                part += __shfl_xor(part, 16, kWave);
                part += __shfl_xor(part, 32, kWave);
                nacc1 += part;
            }
            // the next tile's phase A touches sA / sYP only; sZ is rewritten after its barrier
        }
        // ---------------- epilogue ----------------
        if (dense && q16 == 0) {
            atomicAdd(&sN0[col], nacc0);
            atomicAdd(&sN1[col], nacc1);
        }
        __syncthreads();
        if (tid < D) {
            a.nagg0[p * D + tid] = sN0[tid] * invK;
            a.nagg1[p * D + tid] = sN1[tid] * invK;
        }
        if (KIT > 0) buf ^= 1;
    }
}

size_t fused_l2_lds_bytes(int D, int NW, int K, int nR, int nbuf, int kTM, bool gpc) {
    const int ntile = (K + kTM - 1) / kTM, Kpad = ntile * kTM;
    const size_t words = (size_t)kTM * (2 * D + 2) + (size_t)kTM * (D + 2) + 3 * (size_t)nbuf * Kpad + 2 * D + 2 * nR;
    return words * 4 + (size_t)NW * (kTM / NW) * (gpc ? K + 1 : K) * sizeof(int2);
}

template <int D, int NW, int KIT, bool BF, int TMV, bool GPC>
static hipError_t launch_l2(const FusedL2Args& a, hipStream_t st) {
    const size_t lds = fused_l2_lds_bytes(D, NW, a.K, a.nR, KIT > 0 ? 2 : 1, TMV, GPC);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_attn_l2_kernel<D, NW, KIT, BF, TMV, GPC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int64_t cap = 256 * 4 * (32 / TMV);  // persistent: 256 CUs x up to 4 resident 32-row workgroups
    const int grid = (int)(a.P < cap ? a.P : cap);
    gather_attn_l2_kernel<D, NW, KIT, BF, TMV, GPC><<<grid, NW * 64, lds, st>>>(a);
    return hipGetLastError();
}

// chunk iterations per wave per tile: NPW children x K/4 lanes each over 64 lanes
// The gather phase has two lane-group mappings.  GPC (one lane group per child, no cross-group
// reduction) is the faster one wherever the id-pipelined variants (KIT > 0) apply and a wave has at
// least as many children as lane groups (D >= 32): measured 1.08x (D=64,K=32) to 1.9x (D=32,K=32).
// The unpipelined variant (KIT = 0: K >= 64 at D >= 64) keeps the groups-stride-over-k mapping,
// which is the faster one there (K=64: 3.9 vs 6.2 ms).
template <int D, int NW, bool BF, int TMV>
static hipError_t launch_l2_pick(const FusedL2Args& a, hipStream_t st) {
    static const bool nopipe = getenv("MVIN_L2_NOPIPE") != nullptr;
    constexpr bool GPC = (TMV == 16) || D >= 32;
    const int kit = ((TMV / NW) * (a.K / 4) + 63) / 64;
    if (!nopipe && kit <= 1) return launch_l2<D, NW, 1, BF, TMV, GPC>(a, st);
    if constexpr (D <= 32) {   // at D >= 64 the second chunk pair costs a wave of occupancy
        if (!nopipe && kit == 2) return launch_l2<D, NW, 2, BF, TMV, GPC>(a, st);
    }
    return launch_l2<D, NW, 0, BF, TMV, TMV == 16>(a, st);
}

// K <= 16: a 32-row tile would be at most half full (idle waves in the gather phase, padded MFMA
// rows), so the parent's children go in ONE 16-row tile owned by a workgroup of D/16 waves.
template <int D, bool BF>
static hipError_t launch_l2_small(const FusedL2Args& a, hipStream_t st) {
    return launch_l2_pick<D, D / 16, BF, 16>(a, st);
}

bool fused_l2_supported(int D, int K) {
    const bool dok = D == 16 || D == 32 || D == 64 || D == 128;
    const bool kok = K >= 4 && K <= 256 && (K & (K - 1)) == 0;
    return dok && kok;
}

bool fused_l2_split_in_use() {
    static const char* split_env = getenv("MVIN_L2_SPLIT");
    static const bool use_split = !(split_env && split_env[0] == '0');
    return use_split;
}

hipError_t launch_gather_attn_l2(const FusedL2Args& a, int D, int table_bf16, hipStream_t st) {
    // D >= 32, K in {16 (D = 32), 32, 64, 128}: the role-split pipeline (gather waves + dense waves);
    // MVIN_L2_SPLIT=0 keeps the symmetric kernel below for A/B measurements
    // D = 32, K <= 16 (BASELINE config C2): one wave per parent -- tiles of 16 x 16 rows are too small for the pipeline's
    // per-step costs (MVIN_L2_D32=0: A/B)
    if (fused_d32_applies(a, D)) return launch_gather_attn_l2_d32(a, table_bf16, st);
    if (fused_l2_split_in_use() && fused_split_applies(a, D)) return launch_gather_attn_l2_split(a, D, table_bf16, st);
    // D = 16, K <= 16 (the reference's shipped settings): one wave per parent, no workgroup phases
    if (fused_d16_applies(a, D)) return launch_gather_attn_l2_d16(a, table_bf16, st);
    static const bool no_small = getenv("MVIN_L2_NOSMALL") != nullptr;
    const bool small = a.K <= 16 && !no_small;
    if (table_bf16) {
        switch (D) {
            case 16: return small ? launch_l2_small<16, true>(a, st) : launch_l2_pick<16, 4, true, 32>(a, st);
            case 32: return small ? launch_l2_small<32, true>(a, st) : launch_l2_pick<32, 4, true, 32>(a, st);
            case 64: return small ? launch_l2_small<64, true>(a, st) : launch_l2_pick<64, 4, true, 32>(a, st);
            case 128: return small ? launch_l2_small<128, true>(a, st) : launch_l2_pick<128, 8, true, 32>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (D) {
        case 16: return small ? launch_l2_small<16, false>(a, st) : launch_l2_pick<16, 4, false, 32>(a, st);
        case 32: return small ? launch_l2_small<32, false>(a, st) : launch_l2_pick<32, 4, false, 32>(a, st);
        case 64: return small ? launch_l2_small<64, false>(a, st) : launch_l2_pick<64, 4, false, 32>(a, st);
        case 128: return small ? launch_l2_small<128, false>(a, st) : launch_l2_pick<128, 8, false, 32>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
