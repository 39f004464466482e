// Internal kernel-argument structs and launchers (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvin_hip.h"
#include "mvin_common.h"

namespace mvin {

struct GatherAttnArgs {
    int gather;               // 1: children through adjacency + table; 0: dense neigh rows
    int table_bf16;           // the table holds bf16 rows
    const void* table;        // [nE, D]
    const int32_t* adj_e;     // [nE, K]
    const int32_t* adj_r;     // [nE, K]
    const int32_t* node_ids;  // [T]
    const float* neigh;       // [T*K, D]
    const int32_t* rel_ids;   // [T*K]
    const float* rel_score;   // [nR] or NULL (uniform weights)
    const float* self_vec;    // [T, D]
    const float* Wc;          // [D, D] or NULL
    const float* c_child;     // [B, D] or NULL
    const float* Wagg;        // [D, D]
    const float* bagg;        // [D] or NULL
    float* out;               // [T, D]
    float* probs;             // [T, K] or NULL
    float* s_out;             // [T, D] or NULL: (1/K) sum_k p_k child_k before the projection (training)
    float* z_out;             // [T, D] or NULL: self + neighbors_agg, the input of the dense layer (training)
    int64_t T;
    int N, K, D, lpr_log2;
    int tile_rows;            // node tasks per workgroup tile: 32, or 8 when there are fewer tiles than CUs (set by the launcher)
};

struct RippleArgs {
    const void* E;              // [nE, D] fp32 or bf16 (table_bf16)
    int table_bf16;
    const int32_t* score_ids;
    const int32_t* rel_ids;
    const int32_t* value_ids;
    const float* V;
    const float* w;
    float* out;
    int64_t ldo;
    int64_t B;
    int mode, Nm, D, nR, lpr_log2;
};

struct RippleBuildArgs {
    const int64_t* indptr;      // [nE+1] CSR of the undirected KG
    const int32_t* dst;         // [nnz] neighbor entity
    const int32_t* rel;         // [nnz] relation
    const int64_t* hist_ptr;    // [nU+1] CSR of the users' positive train items
    const int32_t* hist_items;  // [.]
    int32_t* out;               // [nU, P, 3, Nm]
    int64_t n_user;
    uint64_t seed;
    int P, Nm, n_neighbor;
};

struct KeyAddrArgs {
    const void* E;             // [nE, D] fp32 (or bf16 when the kernel is instantiated with BF)
    const float* V;            // [B, nR, D] or NULL (p_hop == 0)
    const float* w;            // [D] h-set logit weights or NULL (PS_O_ft off)
    const int32_t* mem_h[8];   // per hop [B, Nm]
    const int32_t* mem_r[8];
    const int32_t* mem_t[8];
    float* out;                // [B, ldo]: [o_hset | o_hop0 | ...]
    int64_t ldo;
    int64_t B;
    int P, Nm, D, nR, lpr_log2;
    uint64_t table_bytes;      // size of E in bytes (0: unknown -> 64-bit addressing)
    // users feed (instead of mem_h / mem_r / mem_t): pair b reads the lists of user users[b] straight out of
    // user_triplet_set [nU, max(1,P), 3, Nm] (the feed assembly of train.py:117-120 inside the kernel)
    const int32_t* uts;
    const int64_t* users64;
    const int32_t* users32;
    int n_user;                // rows of uts: user ids are clamped to [0, n_user) (device feeds are not validated per launch)
    int n_entity;              // rows of E: head / tail ids are clamped to [0, n_entity) (0: unknown, no clamp)
};

// the three id lists (heads, relations, tails) of pair b at `hop`
struct KeyAddrLists {
    const int32_t *h, *r, *t;
};
__device__ __forceinline__ KeyAddrLists key_addr_lists(const KeyAddrArgs& a, int64_t b, int hop) {
    if (a.uts) {
        int64_t u = a.users64 ? a.users64[b] : (int64_t)a.users32[b];
        u = u < 0 ? 0 : (u >= a.n_user ? a.n_user - 1 : u);
        const int32_t* base = a.uts + ((u * (a.P > 0 ? a.P : 1) + hop) * 3) * (int64_t)a.Nm;
        return {base, base + a.Nm, base + 2 * a.Nm};
    }
    const int64_t o = b * a.Nm;
    return {a.mem_h[hop] + o, hop < a.P ? a.mem_r[hop] + o : nullptr, hop < a.P ? a.mem_t[hop] + o : nullptr};
}

struct KeyAddrGroupedArgs {
    const void* E;             // [nE, D]
    const float* R;            // [nR, D, D] relation_emb_KGE_matrix
    const float* w;            // [D] h-set logit weights or NULL
    const int32_t* uts;        // [nU, max(1,P), 3, Nm]
    const int32_t* seg_user;   // [nseg]
    const int32_t* seg_ptr;    // [nseg + 1]
    const int32_t* nseg_dev;   // [1] actual number of segments (<= nseg) or NULL
    const int32_t* pair_index; // [B]
    const int64_t* items64;    // [B] (or items32)
    const int32_t* items32;
    float* out;                // [B, ldo]
    int64_t ldo;
    int nseg, P, Nm, D, nR, NRL;
    int n_entity;              // rows of E: the wave-per-user kernel clamps item ids into the table
    int dbg;                   // wave-per-user kernel, measurement only (MVIN_KA_WAVE_DBG)
    const int32_t* records;    // [nU, ka_rec_layout(P, Nm, nR).len] static per-user records (mvin_keyaddr_static.hip) or NULL
    const float* ER;           // [nR, nE, D] R_KGE[r] . E[e] for every (relation, entity) (mvin_project_relations) or NULL: with the
                               // records, a user's U rows are GATHERED from it instead of multiplied (key_addr_static_kernel<..., ER>)
    const float* hs;           // [nE] E[e] . w (same call) -- the h-set read's logits; required with ER when w is given
};

// the barrier-free ("flash") form of the grouped key addressing + user MLP (mvin_keyaddr_flash.hip)
struct KaFlashArgs {
    const float* E;            // [nE, 64] fp32
    const float* ER;           // [nR, nE, 64] R_KGE[r] . E[e] (mvin_project_relations)
    const float* hs;           // [nE] E[e] . w (same call) or NULL (no h-set read)
    const int32_t* records;    // static per-user records
    const int32_t* seg_user;
    const int32_t* seg_ptr;
    const int32_t* nseg_dev;
    const int32_t* pair_index;
    const int64_t* items64;
    const int32_t* items32;
    const float* TW;           // [P + has_set][nE, 64] E . user_mlp_matrix[64 j : 64 j + 64, :] (mvin_key_addressing_flash_prepare)
    const float* bmlp;         // [64] or NULL
    float* user_o;             // [B, 64]
    int32_t* slot_seg;         // scheduling (set by the launcher)
    int32_t* counter;
    int nslots;
    int P, Nm, nR, n_entity;
    int64_t B;
};

// the static record of one user: offsets in int32 words (mvin_keyaddr_static.hip); len == 0: shape outside the record form
struct KaRecLayout {
    int NmP, rows, maxtiles, o_cnt, o_off, o_trel, o_bidx, o_head, o_tail, o_hr, len;
};

struct TailArgs {
    const void* E;             // [nE, D] fp32 or bf16
    const int64_t* items64;    // [B] (or items32)
    const int32_t* items32;
    const float* q;            // [B, D] query (transfer_o[0]); unused without the projection
    const float* user_o;       // [B, D]
    const float* nagg0;        // [B, D] from mvin_gather_attn_l2_fwd
    const float* nagg1;
    const float* W0;           // [D, D] level-0 projection or NULL (User_orient off)
    const float* b0;
    const float* A0;           // aggregator (0,0)
    const float* a0;
    const float* A1;           // aggregator (1,0)
    const float* a1;
    const float* Wmix;         // [3D, D] mix-hop combiner
    const float* bmix;
    float* item_emb;           // [B, D] or NULL
    float* scores;             // [B]
    float* sig;                // [B] or NULL
    int64_t B;
    int table_bf16;
    int n_entity;              // rows of E: item ids are clamped to [0, n_entity)
    int dbg;                   // MVIN_TAIL_DBG (measurement only; results wrong): 1 no nagg / user_o loads, 2 one LDS read per product, 4 no stores
};

// entity_aggregates_kernel (mvin_fused_agg.hip): outS[e] = (selfS[e] +) sum_k w(e)_k tabS[y_ek], outG[e] = selfG[e] + sum_k w(e)_k tabG[y_ek]
struct EntityAggArgs {
    const float* tabS;         // [nE, 64] rows summed into outS (T1; folded-tail form: TA1)
    const float* tabG;         // [nE, 64] rows summed into outG (TA2)
    const float* selfS;        // [nE, 64] or NULL (folded-tail form: T0A)
    const float* selfG;        // [nE, 64] (TA1)
    const int32_t* adj_e;      // duplicate-slot encoding
    const int32_t* adj_r;
    const float* t0;           // [nR] relation logits of aggregator (0,.) or NULL (plain mean)
    float* outS;               // [nE, 64]
    float* outG;               // [nE, 64]
    int n_entity, K, nR;
    uint64_t table_bytes, adj_bytes;
    int D;                     // 64 (0: 64) or 32 (mvin_fused_agg32.hip)
};

// l2_tail_fold_kernel (mvin_tail.hip): the part of aggregate_delta_whole above the folded pair kernel
//   out2 = relu(z2 A1 + a1) ;  item = M0[item] + q Wqm + out0 Wm1 + out2 Wm2 + bm ;  score = <user_o, item>
struct TailFoldArgs {
    const float* M0;           // [nE, 64] = E W0 Wm0 (per-call table)
    const int64_t* items64;
    const int32_t* items32;
    const float* q;            // [B, 64]
    const float* user_o;       // [B, 64]
    const float* out0;         // [B, 64] from the folded pair kernel
    const float* z2;           // [B, 64] out0 + nagg1
    const float* Wqm;          // [64, 64] = W0 Wm0
    const float* A1;
    const float* a1;
    const float* Wmix;         // [3 * 64, 64]: rows 64.. are Wm1 | Wm2
    const float* bm;           // [64] = bmix + b0 Wm0
    float* item_emb;           // [B, 64] or NULL
    float* scores;
    float* sig;
    int64_t B;
    int n_entity;
};

struct GatherMixArgs {
    const void* table;         // [nE, D] fp32 or bf16
    const int32_t* adj_e;      // [nE, K]
    const int32_t* adj_r;      // [nE, K]
    const int32_t* node_ids;   // [nodes] or NULL (node i = entity i)
    const float* rel_score;    // [nR] relation logits or NULL (plain mean)
    const float* rowbias;      // [ceil(nodes / npg), D] added to every child row, or NULL
    float* out;                // [nodes, D]
    int64_t nodes;
    int npg, K, D, lpr_log2, relu, table_bf16;
};

struct FusedL2Args {
    const void* table;           // [nE, D] fp32 (or bf16: BF instantiation)
    const int32_t* adj_e;        // [nE, K]
    const int32_t* adj_r;        // [nE, K]
    const int32_t* parent_ids;   // [P] entity id of every level-(L-2) node
    const float* t0;             // [nR] relation logits of aggregator (0,.) or NULL (uniform)
    const float* t1;             // [nR] relation logits of aggregator (1,.) or NULL
    const float* W1;             // [D, D] projection of level L-1 or NULL (User_orient off)
    const float* W2;             // [D, D] projection of level L
    const float* q;              // [B, D] query vector of every pair (transfer_o[0])
    const float* b1;             // [D] projection biases (or NULL)
    const float* b2;
    const float* A0;             // [D, D] aggregator (0,.) weights
    const float* a0;             // [D] bias or NULL
    float* nagg0;                // [P, D] (1/K) sum_n p0[n] self1[n]
    float* nagg1;                // [P, D] (1/K) sum_n p1[n] out1[n]
    float* probs_parent;         // [P, K] or NULL
    float* probs_child;          // [P*K, K] or NULL
    int64_t P;
    uint64_t table_bytes;        // nE * D * 4 (buffer descriptor range)
    uint64_t adj_bytes;          // nE * K * 4: size of adj_e / adj_r (0: unknown)
    int parents_per_pair, K, nR, lpn_log2;
    int pid_stride;              // 1: parent_ids is int32 [P]; 2: the low words of an int64 [P] array (little endian)
    unsigned max_id;             // n_entity - 1: parent ids are clamped (a fault would kill the process)
    int dbg;                     // timing experiments only (MVIN_SPLIT_DBG): 1 = skip the MFMAs, 2 = skip the row loads
    const int32_t* order;        // wave-per-parent kernel (mvin_fused_wpp.hip), parents_per_pair == 1 only, or NULL: slot i of the launch
                                 // works on parent order[i] (its id, its query row, its output rows) -- a permutation that puts parents
                                 // with the same entity next to each other, so that their (identical) rows are cache hits
    float* agg;                  // per-entity aggregates form (mvin_fused_agg.hip), or NULL: [2][nE][64] fp32, S0 | G -- written by
                                 // entity_aggregates_kernel from the projected tables (`table`) and the adjacency, read by
                                 // gather_attn_l2_agg_kernel in place of the tables
    int fold;                    // gather_attn_l2_agg_kernel, folded-tail form (mvin_score_l2_folded_fwd): agg = H0 | G, W1 / b1 = Wq / bq;
                                 // nagg0 <- out0 = relu(H0[x] + q Wq + bq), nagg1 <- out0 + sum_c (p1_c / K) relu(G[x_c] + v)
    int prj;                     // packed kernel over PROJECTED tables (mvin_gather_attn_l2_prj_fwd): `table` = [3][nE][D] fp32
                                 // (E.W1 | E.W1.A0 | E.W2.A0); W1 / b1 and W2 / b2 (= the combined (W1 + c W2).A0 and its bias)
                                 // project the PARENTS' queries only; A0 / a0 unused
};

__device__ __forceinline__ int fused_parent_id(const FusedL2Args& a, int64_t i) {
    const unsigned v = (unsigned)a.parent_ids[i * a.pid_stride];
    return (int)(v < a.max_id ? v : a.max_id);
}

// ---- backward (mvin_bwd.hip) ----
struct EltArgs {
    int mode;
    int64_t n;
    float* x;
    float* y;
    float* z;
    float* w;
    float* accum;
    float alpha, beta, beta1, beta2, eps;
    int D, N;
};

struct WgradArgs {
    mvin_linear_args lin;     // sources / rows / Dsrc / Dout / nz as in the forward
    const float* dY;          // [rows, ldy] (+ z * dy_zstride)
    int64_t ldy, dy_zstride;
    const float* mask;        // forward output for relu masking or NULL
    int64_t ldm, mask_zstride;
    float* dW;                // [Din, Dout] (+ z * dw_zstride), accumulated
    int64_t dw_zstride;
    float* db;                // [Dout] (+ z * db_zstride) or NULL, accumulated
    int64_t db_zstride;
    int IB;
};

struct AggBwdArgs {
    int gather;
    const float* table;       // gather: [nE, D]
    const int32_t* adj_e;
    const int32_t* adj_r;
    const int32_t* node_ids;  // [T]
    const float* child;       // dense: [T*K, D]
    const int32_t* rel_ids;   // dense: [T*K]
    const float* probs;       // [T, K] or NULL (uniform, unless rel_score is given)
    const float* rel_score;   // [nR]: recompute the softmax instead of reading probs (by-entity form)
    int skip_zero;            // skip tasks whose dvec row is all zero
    const float* dvec;        // [T, D]
    float* dtable;            // gather: [nE, D] accumulated atomically
    float* dchild;            // dense: [T*K, D] written
    float* dT;                // [nR] accumulated or NULL
    int64_t T;
    int K, D, nR, lpr_log2;
};

struct KeyAddrBwdArgs {
    KeyAddrArgs f;            // forward arguments (out unused)
    const float* dout;        // [B, ldo] gradient of [o_hset | o_hop0 | ...]
    float* dE;                // [nE, D] accumulated
    float* dV;                // [B, nR, D] accumulated (zero-initialised by the caller)
    float* dw;                // [D] accumulated (h-set logit weights) or NULL
    float l2;                 // l2_weight of the sum(h^2)+sum(t^2) regulariser
    const float* Rk;          // R_KGE [nR, D, D] and ...
    const void* items;        // ... the pairs' item ids [B] (int64 when items64): when both are given (and dV sits in
    int items64;              //     LDS) the kernel adds dE[item_b] += sum_r dV[b, r, :] . R[r]^T itself (V = E[item] . R)
    int dw_rep;               // dw is [dw_rep, D] (power of two >= 1): replicas spread the same-address atomics
    int dv_lds;               // set by the launcher: dV contributions summed per (pair, hop) in LDS first
    float* reg_accum;         // *reg_accum += l2 (sum h^2 + sum t^2) over the hop rows (model.py:383-385) or NULL
};

hipError_t launch_eltwise(const EltArgs& a, hipStream_t st);
hipError_t launch_l2_adam_multi(const mvin_param_seg* segs, int nseg, int64_t total, float* g, float* mo, float* vo,
                                float* accum, int apply_adam, float lr_t, const float* lr_dev, float b1, float b2,
                                float eps, hipStream_t st);
hipError_t launch_scatter_add_rows(float* dtable, const int32_t* ids, int ids64, const float* x, int64_t rows, int D,
                                   float alpha, hipStream_t st);
hipError_t launch_linear_wgrad(WgradArgs a, hipStream_t st);
hipError_t launch_linear_wgrad_multi(const WgradArgs* probs, int n, hipStream_t st);   // n <= 64
hipError_t launch_agg_bwd(const AggBwdArgs& a, hipStream_t st);
hipError_t launch_rel_score_bwd(const float* rel, const float* urh_w, const float* dT, int nR, int D, float* drel,
                                float* durh, hipStream_t st);
hipError_t launch_key_addr_bwd(const KeyAddrBwdArgs& a, hipStream_t st);

inline int lpr_log2_for(int D) {
    int l = 0;
    while ((4 << l) < D) ++l;
    return l;  // 2^l lanes x 4 floats >= D
}

hipError_t launch_expand(const int32_t* adj_e, const int32_t* adj_r, const int64_t* items64,
                         const int32_t* items32, int B, int K, int levels, int n_entity,
                         int32_t* ent_out, int32_t* rel_out, hipStream_t st);
hipError_t launch_rel_score(const float* rel, const float* urh_w, int nR, int D, float* t,
                            hipStream_t st);
hipError_t launch_linear(const mvin_linear_args& a, hipStream_t st);
bool linear_mfma_supported(const mvin_linear_args& a);
hipError_t launch_linear_mfma(const mvin_linear_args& a, hipStream_t st);
hipError_t launch_gather_attn(const GatherAttnArgs& a, hipStream_t st);
hipError_t launch_ripple(const RippleArgs& a, hipStream_t st);
hipError_t launch_sample_adjacency(const int64_t* indptr, const int32_t* dst, const int32_t* rel, int n_entity,
                                   int K, uint64_t seed, int32_t* adj_e, int32_t* adj_r, hipStream_t st);
hipError_t launch_ripple_build(const RippleBuildArgs& a, hipStream_t st);
int key_addr_nj(int Nm, int D);
bool key_addr_grouped_supported(int D, int P, int Nm, int nR);
hipError_t launch_key_addr_grouped(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st);
bool key_addr_wave16_supported(int D, int P, int Nm, int nR);    // wave-per-user variant for D = 16 / 32, mvin_keyaddr_wave.hip
bool key_addr_wave16_applies(const KeyAddrGroupedArgs& a);     // + table small enough for 32-bit row offsets
hipError_t launch_key_addr_wave16(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st);
bool key_addr_dense_supported(int D, int P, int Nm, int nR);     // dense (all-MFMA) variant, mvin_keyaddr_dense.hip
hipError_t launch_key_addr_dense(const KeyAddrGroupedArgs& a, int table_bf16, hipStream_t st);
KaRecLayout ka_rec_layout(int P, int Nm, int nR);               // static per-user records + the kernel over them, mvin_keyaddr_static.hip
hipError_t launch_user_records(const int32_t* uts, int n_user, int P, int Nm, int nR, int n_entity, int32_t* out, hipStream_t st);
bool key_addr_static_supported(int D, int P, int Nm, int nR);
bool key_addr_static_applies(const KeyAddrGroupedArgs& a, int table_bf16);
hipError_t launch_key_addr_static(const KeyAddrGroupedArgs& a, hipStream_t st);
bool key_addr_flash_supported(int D, int P, int Nm, int nR, int n_entity);
size_t key_addr_flash_ws_elems(int64_t B, int nseg_bound);
hipError_t launch_key_addr_flash(const KaFlashArgs& a, int nseg_bound, bool has_set, int32_t* sched_ws, hipStream_t st);
hipError_t kas_read_trace(long long* host_dst, size_t n);
hipError_t kaf_read_trace(long long* host_dst, size_t n);
hipError_t launch_key_addr(const KeyAddrArgs& a, int table_bf16, hipStream_t st);
bool key_addr_stream_supported(const KeyAddrArgs& a, int table_bf16);       // LDS-DMA streaming variant, mvin_keyaddr_stream.hip
hipError_t launch_key_addr_stream(const KeyAddrArgs& a, int table_bf16, hipStream_t st);
hipError_t launch_move_rows(void* table, const int32_t* ids, int64_t n, int row_bytes, void* rows, bool scatter,
                            hipStream_t st);
hipError_t launch_shard_space_ids(const void* ids, bool is64, int64_t n, int world, int n_local, void* out, hipStream_t st);
hipError_t launch_row_softmax(const float* x, int64_t rows, int n, float* out, hipStream_t st);
hipError_t launch_mix_urv(const float* neigh, const float* rel, const float* user, const float* logits, int64_t nodes, int N, int K,
                          int D, float* out, float* probs, hipStream_t st);
hipError_t launch_gather_mix(const GatherMixArgs& a, hipStream_t st);
// ---- the whole depth-2 pass in one launch (mvin_score_small.hip) ----
struct ScoreSmallArgs {
    const void* E;               // [nE, D] fp32 or bf16
    const int32_t* adj_e;        // [nE, K] plain adjacency, or its duplicate-slot encoding (enc != 0)
    const int32_t* adj_r;
    const float* R;              // [nR, D, D] relation_emb_KGE_matrix
    const float* w_h;            // [D] h-set logit weights or NULL (PS_O_ft off)
    const float* Wu;             // [(P + (w_h != NULL)) * D, D] user MLP
    const float* bu;
    const float* t0;             // [nR] relation logits of aggregator (0,0) / (1,0) or NULL (plain mean)
    const float* t1;
    const float *W0, *b0, *W1, *b1, *W2, *b2;      // projections of levels 0, 1, 2 or all NULL (User_orient off)
    const float *A0, *a0, *A1, *a1, *Wmix, *bmix;
    const int64_t* items;        // [B]
    const int32_t* mem_h[4];     // per hop [B, Nm] (per-pair feed) ...
    const int32_t* mem_r[4];
    const int32_t* mem_t[4];
    const int32_t* uts;          // ... or user_triplet_set [nU, max(1,P), 3, Nm] + users [B]
    const int64_t* users;
    float* user_o;               // out [B, D]
    float* item_emb;             // out [B, D] or NULL
    float* scores;               // out [B]
    float* sig;                  // out [B] or NULL
    int64_t B;
    int G;                       // pairs per workgroup (<= 0: chosen by the launcher)
    int K, P, Nm, nR, n_entity, n_user, enc, table_bf16;
    int depth1;                  // one-hop tree (h_hop = 1): no grandchildren, no aggregator (1,0)
    int dbg;                     // timing experiments only (MVIN_SMALL_DBG)
    uint64_t table_bytes;        // nE * D * 4 and nE * K * 4: buffer descriptor ranges (both below 4 GiB)
    uint64_t adj_bytes;
};
bool score_small_supported(int D, int K, int P, int Nm, int nR);
int score_small_group(int D, int K, int P, int Nm, int nR, int has_hset, int64_t B);
hipError_t launch_score_small(ScoreSmallArgs a, int D, hipStream_t st);
hipError_t small_read_trace(long long* host_dst, size_t n);

bool l2_tail_supported(int D);
hipError_t launch_l2_tail(const TailArgs& a, int D, hipStream_t st);
bool l2_tail_flash_applies(const TailArgs& a, int D);
hipError_t launch_l2_tail_flash(const TailArgs& a, hipStream_t st);
bool fused_l2_supported(int D, int K);
hipError_t launch_gather_attn_l2(const FusedL2Args& a, int D, int table_bf16, hipStream_t st);
bool fused_split_supported(int D, int K);      // role-split variant (mvin_fused_split.hip)
bool fused_split_applies(const FusedL2Args& a, int D);
hipError_t launch_gather_probe_l2(const void* table, const int32_t* ids1, const int32_t* ids2, int64_t n_parents, int K,
                                  int D, int table_bf16, float* sums, hipStream_t st);   // mvin_probe.hip
hipError_t launch_group_pairs(const int64_t* u64, const int32_t* u32, int64_t B, int n_user, int32_t* count, int32_t* offs,
                              int32_t* rank, int32_t* seg_user, int32_t* seg_ptr, int32_t* nseg, int32_t* pair_index, hipStream_t st);   // mvin_group.hip
hipError_t launch_count_ids(const int32_t* ids, int64_t n, int nbins, float* out, hipStream_t st);   // mvin_bwd.hip
bool fused_d32_supported(int D, int K);        // wave-per-parent variant for D = 32, K in {8, 16} (mvin_fused_d32.hip)
bool fused_d32_applies(const FusedL2Args& a, int D);
size_t order_ws_elems(int64_t B);                             // pairs in key order (mvin_order.hip)
hipError_t launch_order_by_key(const int64_t* k64, const int32_t* k32, int64_t B, int32_t* ws, int32_t* order, hipStream_t st);
bool fused_wpp_supported(int D, int K);                       // wave-per-parent kernel over projected tables, dim 64 (mvin_fused_wpp.hip)
bool fused_wpp_applies(const FusedL2Args& a, int D);
hipError_t launch_gather_attn_l2_wpp(const FusedL2Args& a, hipStream_t st);
bool fused_agg_supported(int D, int K);                       // per-entity aggregates S0 | G of the projected tables, dim 64 (mvin_fused_agg.hip)
bool fused_fold_supported(int D, int K);                      // folded-tail form: dim 64 (K 16 / 32 / 64) and dim 32 (K 16 / 32: mvin_fused_agg32.hip)
size_t fused_fold_lds_bytes(int D, int nR, int K);
bool fused_agg_applies(const FusedL2Args& a, int D);
hipError_t launch_entity_aggregates(const EntityAggArgs& a, hipStream_t st);
hipError_t launch_fold_prepare(const float* W0, const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* A0,
                               const float* a0, const float* Wmix, const float* bmix, const float* A1, float c, int D, float* blk, hipStream_t st);   // mvin_fused_agg.hip
hipError_t launch_l2_tail_fold(const TailFoldArgs& a, hipStream_t st);          // mvin_tail.hip (dim 64)
hipError_t launch_score_l2_folded(const float* agg, const float* M0, const int32_t* adj_e, const int32_t* adj_r, const int32_t* items, int pid_stride,
                                  const float* t1, const float* q, const float* user_o, const float* Wq, const float* bq, const float* Wv,
                                  const float* bv, const float* Wqm, const float* A1, const float* a1, const float* Wm1, const float* Wm2,
                                  const float* bm, float* item_emb, float* scores, float* sig, int64_t B, int K, int D, int nR, int n_entity,
                                  hipStream_t st);                              // mvin_fused_agg.hip / mvin_fused_agg32.hip: the folded-tail form in one launch
hipError_t launch_gather_attn_l2_agg(const FusedL2Args& a, hipStream_t st);     // a.agg, the encoding, a.t1, the query terms -> nagg0 / nagg1
hipError_t launch_gather_attn_l2_d32(const FusedL2Args& a, int table_bf16, hipStream_t st, bool encoded = false);   // encoded: adj_e / adj_r = the duplicate-slot encoding
bool fused_d16_supported(int D, int K);        // wave-per-parent variant for D = 16, K <= 16 (mvin_fused_d16.hip)
bool fused_d16_applies(const FusedL2Args& a, int D);
hipError_t launch_gather_attn_l2_d16(const FusedL2Args& a, int table_bf16, hipStream_t st);
bool fused_l2_split_in_use();                   // false under MVIN_L2_SPLIT=0
hipError_t launch_gather_attn_l2_split(const FusedL2Args& a, int D, int table_bf16, hipStream_t st);
hipError_t split_read_trace(long long* host_dst, size_t n);
bool fused_packed_supported(int D, int K);
bool key_addr_static_er_ok(int P, int Nm, int nR, int n_entity, bool has_set);
hipError_t launch_transpose_blocks(const float* R, int nR, int D, float* RT, hipStream_t st);
hipError_t launch_entity_dot(const float* E, const float* w, int n, int D, float* out, hipStream_t st);
hipError_t launch_prj_prepare(const float* W1, const float* W2, const float* b1, const float* b2, const float* A0, const float* a0,
                              float c, int D, float* blk, hipStream_t st);     // packed-tile variant over the duplicate-slot encoding (mvin_fused_packed.hip)
bool fused_packed_applies(const FusedL2Args& a, int D);
hipError_t pack_read_prof(long long* host_dst, size_t n);
hipError_t launch_gather_attn_l2_packed(const FusedL2Args& a, int D, int table_bf16, hipStream_t st);
hipError_t launch_encode_adjacency(const int32_t* adj_e, const int32_t* adj_r, int n_entity, int K, int32_t* cnt,
                                   int32_t* enc_e, int32_t* enc_r, hipStream_t st);   // mvin_prep.hip
hipError_t ka_read_trace(long long* host_dst, size_t n);   // development aid, see mvin_fused_split.hip

}  // namespace mvin
