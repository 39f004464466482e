// MVIN._key_addressing (model.py:161-240) + the user MLP (model.py:232-236) for pairs grouped by user, as ONE barrier-free kernel:
// every WAVE walks tiles of up to 32 pairs of one user on its own, nothing but registers between the stages ("flash" form).
//
// The reads of a hop are two matrix products with a softmax in between,
//     logits[pair, m] = E[item_pair] . U_m          U_m = R_KGE[r_m] . E[h_m]          (:214-220)
//     o[pair, :]      = sum_m softmax_m(logits)[pair, m] E[t_m]                        (:223-230)
// followed by user_o = concat(o_list) . user_mlp_matrix + bias (:232-236).  U_m depends on the (relation, entity) pair alone and
// the MLP is linear in the tail rows, so both move from the gathered rows to per-call TABLES built from the current parameters
// (mvin_key_addressing_flash_prepare; an exact re-association like the projected tables of the two deepest levels):
//     ER[r][e] = R_KGE[r] . E[e]  ([nR, nE, D]);   TW_j[e] = E[e] . Wmlp[64 j : 64 j + 64, :]  (one [nE, D] table per block j of o_list)
//     user_o[pair] = bias + sum_m p_hset[m] TW_0[h_m] + sum_hop sum_m p_hop[pair, m] TW_{1 + hop}[t_m]
// The static per-user records (mvin_build_user_records) hold the row numbers.  Both products of a hop run TRANSPOSED on
// v_mfma_f32_16x16x4_f32, so that the accumulator of the first is the B operand of the second without leaving the registers
// (accumulator register r of lane (q16, l16) = element [4 q16 + r][l16]; the B operand of contraction step k of lane (q16, l16)
// = element [k-th index of lane group q16][l16]):
//     logits^T[m, pair] = sum_d ER[hr_m][d] Ei[pair][d]     A = U rows    (lane (q16, l16 = m): 4 x 16 B of its row, from global)
//                                                           B = Ei        (lane (q16, l16 = pair): 4 x 16 B of its row, from global)
//     user_o^T[n, pair] += sum_m TW[t_m][n] p[m, pair]      A = TW rows   (lane (q16, l16): float4 TW[t_m(q16)][4 l16 ..], global)
//                                                           B = p         = the logits accumulators after the softmax
// Every gathered row is read straight into its MFMA operand layout as whole 256-byte rows (16 lanes x 16 B) ONCE per 32 pairs (two
// 16-pair column tiles share every A operand); nothing is staged in LDS, no workgroup barrier, no role split.  The row loads are
// software-pipelined by hand: the 16 registers a memory tile's U rows leave behind in the logits product take that tile's TW rows
// at once, and the registers the TW rows leave behind take the next stage's U rows -- a load is in flight for a whole product.
// The h-set read (:162-197) does not depend on the item: once per slot.
//
// Scheduling: a SLOT table over the batch in user order -- segment s = pairs [p0, p1) of one user owns the slots
// p0 / 64 + s ... (disjoint by construction, holes = -1), one slot = up to 64 pairs = two tiles -- and ONE atomic counter the
// persistent waves draw slots from (results do not depend on who draws what).  D = 64, fp32 tables, Nm <= 64, P >= 1.
#include <cstdlib>
#include <type_traits>

#include "mvin_kernels.h"

namespace mvin {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFlashChunk = 64;       // pairs per slot (two tiles of 32): the h-set read is taken once per slot
constexpr int kFlashWaves = 4;        // waves per workgroup (one per SIMD)

// slot_seg[p0 / 64 + s + c] = s for the chunks c of segment s; every other slot keeps the -1 of the launcher's memset
__global__ void ka_flash_slots_kernel(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ nseg_dev, int nseg_bound,
                                      int32_t* __restrict__ slot_seg) {
    const int nseg = nseg_dev ? min(*nseg_dev, nseg_bound) : nseg_bound;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) {
        const int p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
        const int first = p0 / kFlashChunk + s;
        const int nch = (p1 - p0 + kFlashChunk - 1) / kFlashChunk;
        for (int c = 0; c < nch; ++c) slot_seg[first + c] = s;
    }
}

#define KAF_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ long long g_kaf_trace[64 * 16];    // MVIN_KAF_TRACE=1: wave 0 of workgroup 0 stamps its stage boundaries (scripts/trace_flash.py)

template <int NMT, bool HAS_SET, bool TRACE = false>
__global__ __launch_bounds__(kFlashWaves * 64, 2) void key_addr_flash_kernel(KaFlashArgs a, KaRecLayout RL) {
    constexpr int D = 64;
    // the slot's constant part of user_o (bias + h-set block), parked per wave between the tiles of a slot: 16 registers more in
    // the tile loop were 18 spilled ones at four memory tiles per hop
    __shared__ f32x4 sUo0[kFlashWaves][4][64];
    const int lane = threadIdx.x & 63, l16 = lane & 15, q16 = lane >> 4;
    f32x4 (*myUo0)[64] = sUo0[threadIdx.x >> 6];
    const int P = a.P, NmP = RL.NmP;
    const unsigned emax = (unsigned)(a.n_entity - 1);
    const float* __restrict__ E = a.E;
    const size_t tw_stride = (size_t)a.n_entity * D;

    struct Rows {
        float4 v[NMT][4];
    };
    // U rows of one hop: memory tile mt, lane (q16, l16) -> row hr[mt] (of memory 16 mt + l16), bytes [64 nt + 16 q16, + 16)
    auto issue_u = [&](int mt, const int (&hr)[NMT], Rows& R) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.ER + (size_t)(unsigned)max(hr[mt], 0) * D) + q16;
        // (a row of the 245 MB table is used once per slot: streamed past the caches that hold the entity-sized tables)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 v = __builtin_nontemporal_load(src + 4 * nt);
            R.v[mt][nt] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    // projected tail rows: step (mt, i) <-> memory 16 mt + 4 q16 + i, lane (q16, l16) -> columns [4 l16, + 4) of that row
    auto issue_t = [&](int mt, const float* __restrict__ tab, const int4 (&ids)[NMT], Rows& R) {
        const int id[4] = {ids[mt].x, ids[mt].y, ids[mt].z, ids[mt].w};
#pragma unroll
        for (int i = 0; i < 4; ++i)          // (a padding memory, id -1, weighs 0: any finite row will do)
            R.v[mt][i] = reinterpret_cast<const float4*>(tab + (size_t)(unsigned)max(id[i], 0) * D)[l16];
    };
    // uo^T += rows^T p for one memory tile (both column tiles of pairs share the A operand)
    auto reads_tile = [&](bool two, int mt, const Rows& R, const f32x4 (&p0)[NMT], const f32x4 (&p1)[NMT], f32x4 (&u0)[4], f32x4 (&u1)[4]) {
        // (the flag is tested ONCE per memory tile: under a test per product every MFMA of the second column tile sat in a branch of
        //  its own, with the loads in flight waited for inside)
        if (two) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t[4] = {R.v[mt][i].x, R.v[mt][i].y, R.v[mt][i].z, R.v[mt][i].w};
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) u0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[nt], p0[mt][i], u0[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) u1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[nt], p1[mt][i], u1[nt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t[4] = {R.v[mt][i].x, R.v[mt][i].y, R.v[mt][i].z, R.v[mt][i].w};
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) u0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[nt], p0[mt][i], u0[nt], 0, 0, 0);
            }
        }
    };
    // softmax over the memories of one hop for the pair of this lane's column: lg[mt][i] <-> memory 16 mt + 4 q16 + i, spread over
    // the four lane groups; a padding memory (id < 0) weighs 0; normalised weights in place.  No branch: the exponentials of a
    // wave issue back to back beside the other wave's products
    auto softmax = [&](f32x4 (&lg)[NMT], const int4 (&ids)[NMT]) {
        float mx = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const int id[4] = {ids[mt].x, ids[mt].y, ids[mt].z, ids[mt].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, id[i] >= 0 ? lg[mt][i] : -INFINITY);
        }
        mx = xor32_max(xor16_max(mx));
        float z = 0.f;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const int id[4] = {ids[mt].x, ids[mt].y, ids[mt].z, ids[mt].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = lean_exp(fminf(lg[mt][i] - mx, 0.f));        // (evaluated for every memory: a select, not a branch)
                const float e = id[i] >= 0 ? x : 0.f;
                lg[mt][i] = e;
                z += e;
            }
        }
        z = xor32_sum(xor16_sum(z));
        const float inv = 1.f / z;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[mt][i] *= inv;
        }
    };

    int iter = 0;
    auto stamp = [&](int slot) {
        if constexpr (TRACE) {
            if (blockIdx.x == 0 && threadIdx.x == 0 && iter >= 2 && iter < 66) g_kaf_trace[(iter - 2) * 16 + slot] = __builtin_readcyclecounter();
        }
    };
    for (;; ++iter) {
        stamp(0);
        int k = 0;
        if (lane == 0) k = atomicAdd(a.counter, 1) + 1;       // (the launcher's memset leaves -1)
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= a.nslots) break;
        const int s = __builtin_amdgcn_readfirstlane(a.slot_seg[k]);
        if (s < 0) continue;
        const int u = __builtin_amdgcn_readfirstlane(a.seg_user[s]);
        const int p0 = __builtin_amdgcn_readfirstlane(a.seg_ptr[s]), p1 = __builtin_amdgcn_readfirstlane(a.seg_ptr[s + 1]);
        const int cbeg = p0 + (k - (p0 / kFlashChunk + s)) * kFlashChunk;
        const int cend = min(p1, cbeg + kFlashChunk);
        const int32_t* __restrict__ rec = a.records + (size_t)u * RL.len;
        stamp(1);
        auto load_hr = [&](int hop, int (&hr)[NMT]) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) hr[mt] = rec[RL.o_hr + hop * NmP + 16 * mt + l16];
        };
        auto load_tid = [&](int hop, int4 (&tid)[NMT]) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) tid[mt] = *reinterpret_cast<const int4*>(rec + RL.o_tail + hop * NmP + 16 * mt + 4 * q16);
        };
        // the pairs of a tile: column l16 of column tile rt <-> position t0 + 16 rt + l16 of the batch in user order
        auto load_orig = [&](int t0, int (&orig)[2], bool (&valid)[2]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int p = t0 + 16 * rt + l16;
                valid[rt] = p < cend;
                orig[rt] = a.pair_index[min(p, cend - 1)];
            }
        };
        auto load_item = [&](const int (&orig)[2], unsigned (&item)[2]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                item[rt] = min(a.items64 ? reinterpret_cast<const unsigned*>(a.items64)[2 * (int64_t)orig[rt]] : (unsigned)a.items32[orig[rt]], emax);
        };
        auto load_be = [&](const unsigned (&item)[2], float4 (&bE)[2][4]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bE[rt][nt] = reinterpret_cast<const float4*>(E + (size_t)item[rt] * D)[4 * nt + q16];
            }
        };

        // ---- first tile's operands on their way before anything is computed ----
        int orig[2], origN[2];
        bool valid[2], validN[2];
        unsigned item[2];
        float4 bE[2][4];
        int hrC[NMT], hrN[NMT];
        int4 tidC[NMT];
        Rows R;
        load_orig(cbeg, orig, valid);
        load_hr(0, hrC);
        load_tid(0, tidC);
        load_orig(cbeg + 32, origN, validN);                 // (past the slot's end: clamped, unused)
        // ---- per slot: bias + the h-set block (pair-independent: the same value in every column); its rows travel through the
        //      registers of the first stage's U rows, which follow them tile by tile ----
        f32x4 uo0[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {        // accumulator register r of column tile nt <-> output column 16 q16 + 4 r + nt
            const float4 b = a.bmlp ? reinterpret_cast<const float4*>(a.bmlp)[4 * q16 + r] : make_float4(0.f, 0.f, 0.f, 0.f);
            uo0[0][r] = b.x, uo0[1][r] = b.y, uo0[2][r] = b.z, uo0[3][r] = b.w;
        }
        if constexpr (HAS_SET) {
            int4 hid[NMT];
            f32x4 ph[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) hid[mt] = *reinterpret_cast<const int4*>(rec + RL.o_head + 16 * mt + 4 * q16);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) issue_t(mt, a.TW, hid, R);
            KAF_FENCE();
            stamp(11);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int id[4] = {hid[mt].x, hid[mt].y, hid[mt].z, hid[mt].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) ph[mt][i] = a.hs[max(id[i], 0)];
            }
            load_item(orig, item);
            KAF_FENCE();
            stamp(12);
            softmax(ph, hid);
            KAF_FENCE();
            stamp(13);
            f32x4 unused[4];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                reads_tile(false, mt, R, ph, ph, uo0, unused);
                KAF_FENCE();
                issue_u(mt, hrC, R);
                KAF_FENCE();
            }
            stamp(14);
        } else {
            load_item(orig, item);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) issue_u(mt, hrC, R);
        }
        load_be(item, bE);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) myUo0[nt][lane] = uo0[nt];      // (read back by this lane only)
        KAF_FENCE();
        stamp(2);

        const int ntile = (cend - cbeg + 31) >> 5;
        int t0 = cbeg;
        // one tile of up to 16 (two false) or up to 32 pairs
        auto run_tile = [&](const bool two, const bool more) {
            f32x4 uo[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) uo[rt][nt] = myUo0[nt][lane];
            }
            unsigned itemN[2];
            if (more) load_item(origN, itemN);
            for (int hop = 0; hop < P; ++hop) {
                const bool last_hop = hop + 1 == P;
                const bool next_stage = !last_hop || more;   // somebody will consume the U rows requested below
                const int hopN = last_hop ? 0 : hop + 1;
                const float* __restrict__ tw = a.TW + (size_t)(hop + (HAS_SET ? 1 : 0)) * tw_stride;
                if (next_stage) load_hr(hopN, hrN);
                KAF_FENCE();
                if (t0 == cbeg && hop < 3) stamp(3 + 4 * hop);
                // logits^T, one memory tile after the other; the registers of a tile's U rows take its projected tail rows at once
                f32x4 lg[2][NMT];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    lg[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    lg[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (two) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const float av[4] = {R.v[mt][nt].x, R.v[mt][nt].y, R.v[mt][nt].z, R.v[mt][nt].w};
                            const float e0[4] = {bE[0][nt].x, bE[0][nt].y, bE[0][nt].z, bE[0][nt].w};
                            const float e1[4] = {bE[1][nt].x, bE[1][nt].y, bE[1][nt].z, bE[1][nt].w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                lg[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], e0[i], lg[0][mt], 0, 0, 0);
                                lg[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], e1[i], lg[1][mt], 0, 0, 0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const float av[4] = {R.v[mt][nt].x, R.v[mt][nt].y, R.v[mt][nt].z, R.v[mt][nt].w};
                            const float e0[4] = {bE[0][nt].x, bE[0][nt].y, bE[0][nt].z, bE[0][nt].w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) lg[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], e0[i], lg[0][mt], 0, 0, 0);
                        }
                    }
                    KAF_FENCE();
                    issue_t(mt, tw, tidC, R);
                    KAF_FENCE();
                }
                // the item rows are done with after the tile's last logits: the next tile's take their registers
                if (last_hop && more) load_be(itemN, bE);
                KAF_FENCE();
                if (t0 == cbeg && hop < 3) stamp(4 + 4 * hop);
                softmax(lg[0], tidC);
                if (two) softmax(lg[1], tidC);
                KAF_FENCE();
                if (t0 == cbeg && hop < 3) stamp(5 + 4 * hop);
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    reads_tile(two, mt, R, lg[0], lg[1], uo[0], uo[1]);
                    KAF_FENCE();
                    if (next_stage) issue_u(mt, hrN, R);
                    KAF_FENCE();
                }
                if (next_stage) load_tid(hopN, tidC);
                if (t0 == cbeg && hop < 3) stamp(6 + 4 * hop);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (valid[rt] && (rt == 0 || two)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<float4*>(a.user_o + (size_t)orig[rt] * D + 16 * q16 + 4 * r) =
                            make_float4(uo[rt][0][r], uo[rt][1][r], uo[rt][2][r], uo[rt][3][r]);
                }
            }
        };
        for (int tile = 0; tile < ntile; ++tile, t0 += 32) {
            const bool more = tile + 1 < ntile;
            run_tile(cend - t0 > 16, more);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) orig[rt] = origN[rt], valid[rt] = validN[rt];
        }
        stamp(15);
    }
}

hipError_t kaf_read_trace(long long* host_dst, size_t n) {
    const size_t have = sizeof(g_kaf_trace) / sizeof(long long);
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_kaf_trace), (n < have ? n : have) * sizeof(long long));
}
#undef KAF_FENCE

bool key_addr_flash_supported(int D, int P, int Nm, int nR, int n_entity) {
    if (D != 64 || P < 1 || P > 8 || Nm < 1 || Nm > 64) return false;
    const KaRecLayout RL = ka_rec_layout(P, Nm, nR);
    if (RL.len == 0) return false;
    return (long long)nR * n_entity < (1ll << 31);                                    // 32-bit row numbers in the records
}

// scheduling workspace (int32 words): the slot table + the counter
size_t key_addr_flash_ws_elems(int64_t B, int nseg_bound) { return (size_t)(B / kFlashChunk + nseg_bound + 1) + 1; }

template <int NMT>
static hipError_t launch_flash_n(const KaFlashArgs& a, const KaRecLayout& RL, bool has_set, hipStream_t st) {
    auto k = has_set ? key_addr_flash_kernel<NMT, true> : key_addr_flash_kernel<NMT, false>;
    if constexpr (NMT == 4) {
        static const bool trace = getenv("MVIN_KAF_TRACE") != nullptr;
        if (trace && has_set) k = key_addr_flash_kernel<4, true, true>;
    }
    static thread_local int per_cu[2] = {0, 0};
    int& pc = per_cu[has_set ? 1 : 0];
    if (pc == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(k), kFlashWaves * 64, 0) != hipSuccess || v < 1) v = 2;
        pc = v > 8 ? 8 : v;
    }
    // persistent: every CU full; no more workgroups than slots' worth of waves
    const int64_t want = ((int64_t)a.nslots + kFlashWaves - 1) / kFlashWaves;
    const int grid = (int)(want < 256 * (int64_t)pc ? want : 256 * (int64_t)pc);
    k<<<grid, kFlashWaves * 64, 0, st>>>(a, RL);
    return hipGetLastError();
}

hipError_t launch_key_addr_flash(const KaFlashArgs& a_, int nseg_bound, bool has_set, int32_t* sched_ws, hipStream_t st) {
    KaFlashArgs a = a_;
    const KaRecLayout RL = ka_rec_layout(a.P, a.Nm, a.nR);
    if (RL.len == 0 || !sched_ws) return hipErrorInvalidValue;
    a.nslots = (int)(a.B / kFlashChunk + nseg_bound + 1);
    a.slot_seg = sched_ws;
    a.counter = sched_ws + a.nslots;
    hipError_t e = hipMemsetAsync(sched_ws, 0xFF, ((size_t)a.nslots + 1) * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    {
        const int blocks = nseg_bound < 256 * 64 ? (nseg_bound + 255) / 256 : 64;
        ka_flash_slots_kernel<<<blocks > 0 ? blocks : 1, 256, 0, st>>>(a.seg_ptr, a.nseg_dev, nseg_bound, sched_ws);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    switch (RL.NmP / 16) {
        case 1: return launch_flash_n<1>(a, RL, has_set, st);
        case 2: return launch_flash_n<2>(a, RL, has_set, st);
        case 3: return launch_flash_n<3>(a, RL, has_set, st);
        case 4: return launch_flash_n<4>(a, RL, has_set, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
