/*
 * mvin_hip.h -- C ABI of libmvin_hip.so: the MI355X (gfx950) implementation of the MVIN
 * K-hop neighbor-attention aggregation + embedding-propagation scoring path.
 *
 * The reference (johnnyjana730/MVIN) is pure Python on TensorFlow 1.x and has no FFI of
 * its own; its boundary is the Python class surface `MVIN` / `SumAggregator_urh_matrix`
 * (src/model/MVIN/model.py:6-444, aggregators.py:17-152).  Every entry point below
 * replaces one stock-TF op sequence of that graph and cites it.  mvin_amd/model.py and
 * mvin_amd/aggregators.py bind these through ctypes and keep the reference's call surface.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch allocates; the library
 *    holds no state except a per-thread last-error string);
 *  - tensors are dense row-major; ids are int32 (int64 also at mvin_expand_ids and in
 *    mvin_linear_args, the reference's placeholder dtype, model.py:50-51); tables are fp32, the
 *    entity table optionally bf16 (table_bf16 / src_bf16) with fp32 arithmetic;
 *  - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); launches
 *    are asynchronous on it; the library never synchronises;
 *  - return 0 on success, <0 for argument/shape errors, >0 = hipError_t of the launch.
 *    mvin_last_error() describes the last failure on the calling thread;
 *  - B = pairs in the batch, K = neighbor_sample_size, D = dim, N = nodes per pair at the
 *    level being aggregated, T = B*N "node tasks".
 */
#ifndef MVIN_HIP_H
#define MVIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVIN_ABI_VERSION 12
/* The library is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table. */
#define MVIN_API __attribute__((visibility("default")))
#define MVIN_MAX_DIM 256      /* D % 4 == 0, 4 <= D <= 256 */
#define MVIN_MAX_SRC 8        /* concatenated sources of mvin_linear_fwd */

MVIN_API int mvin_abi_version(void);
MVIN_API const char* mvin_last_error(void);

/* Development aid, not part of the reference-facing path: with MVIN_SPLIT_DBG=4 in the environment workgroup 0
 * of the role-split fused kernel stamps s_memtime at its phase boundaries; this copies the stamps
 * ([2 roles][64 steps][8 slots] int64) to `host_dst` (synchronous).  scripts/trace_split.py prints them. */
MVIN_API int mvin_debug_read_trace(long long* host_dst, size_t n);

/* Number of int32 elements of the flattened id lists mvin_expand_ids writes:
 * entities levels 0..levels (B * sum_{e<=levels} K^e) and relations levels 0..levels-1
 * (B * sum_{1<=e<=levels} K^e).  Level e of `ent_out` starts at B*sum_{i<e}K^i and is
 * [B, K^e]; level e of `rel_out` starts at B*sum_{1<=i<=e}K^i and is [B, K^(e+1)]. */
MVIN_API size_t mvin_ent_elems(int B, int K, int levels);
MVIN_API size_t mvin_rel_elems(int B, int K, int levels);

/* MVIN.get_neighbors (model.py:243-256): level-by-level expansion of the fixed-fan-out
 * adjacency.  entities[0] = items; entities[e+1][b, j*K+k] = adj_entity[entities[e][b,j], k];
 * relations[e][b, j*K+k] = adj_relation[entities[e][b,j], k].  `levels` expansions.
 * items are int64 (items_i64) or int32 (items_i32); exactly one must be non-NULL. */
MVIN_API int mvin_expand_ids(const int32_t* adj_entity, const int32_t* adj_relation,
                    const int64_t* items_i64, const int32_t* items_i32,
                    int B, int K, int levels, int n_entity,
                    int32_t* ent_out, int32_t* rel_out, void* stream);

/* Attention logits of SumAggregator_urh_matrix._mix_neighbor_vectors_urh
 * (aggregators.py:118-139) reduced to their k-dependent term:
 * score[b,n,k] = [user; rel_k; self] . urh_weights = const(b,n) + rel_emb[r_k] . urh_w[D:2D];
 * the constant cancels in softmax_k, so t[r] = rel_emb[r,:] . urh_weights[D:2D] is all
 * the kernels need.  t_out is [nR]. */
MVIN_API int mvin_rel_score(const float* relation_emb, const float* urh_weights, int nR, int D,
                   float* t_out, void* stream);

/* Generic "rows x small dense" operator behind the stock tf.matmul sites of the path:
 *   out[z][r, :] = act( concat_s(X_s[r, :]) . W[z] + bias[z] + rowbias[r / rows_per_group, :] )
 * X_s is either dense rows (ids[s]==NULL: X_s[r] = src[s] + r*Dsrc) or a table gather
 * (X_s[r] = src[s] + ids[s][r]*Dsrc; tf.nn.embedding_lookup).  W is [nsrc*Dsrc, Dout]
 * (NULL = identity copy; needs nsrc==1, Dsrc==Dout).  nz batches share the sources and
 * step W/bias/out by the given element strides.  Optional fused scoring epilogue
 * (model.py:158-159): score[r] = sum_j out[r,j]*score_u[r,j]; sigmoid[r] = 1/(1+exp(-score)).
 * sum_sources = 1 feeds (X_0 + X_1 + ...) . W instead: the aggregator epilogue
 * relu((self + neighbors_agg) . weights + bias) of aggregators.py:108-116 when neighbors_agg
 * was produced by mvin_gather_attn_l2_fwd.
 * Covers: user-oriented projection (model.py:270-283), mix-hop combiner (:310-315),
 * user MLP (:232-236), the per-relation item projection of _key_addressing (:211-220). */
typedef struct {
    const float* src[MVIN_MAX_SRC];
    const int32_t* ids[MVIN_MAX_SRC];
    int nsrc;
    int Dsrc;
    int Dout;
    int64_t rows;
    const float* W;
    const float* bias;         /* [Dout] or NULL */
    const float* rowbias;      /* [ceil(rows/rows_per_group), Dout] or NULL */
    int rows_per_group;
    int relu;
    int ids64;                 /* 1: ids[] point to int64 row ids (the reference's placeholder dtype) */
    int sum_sources;           /* 1: X = sum_s X_s (Din = Dsrc) instead of the concatenation */
    float* out;
    int64_t ldo;                 /* out row stride in elements (>= Dout) */
    int nz;
    int64_t w_zstride, bias_zstride, out_zstride;
    const float* score_u;      /* [rows, Dout] or NULL */
    float* score_out;          /* [rows] or NULL */
    float* sigmoid_out;        /* [rows] or NULL */
    int src_bf16;              /* bit s set: src[s] is a bf16 table (read as bf16, widened to fp32) */
    int64_t src_rows;          /* rows of the gathered sources (those with ids[s] != NULL): ids are clamped to
                                  [0, src_rows) like every device-resident id; 0 = unknown, no clamp */
} mvin_linear_args;
MVIN_API int mvin_linear_fwd(const mvin_linear_args* args, void* stream);

/* Deepest hop of MVIN.aggregate_delta_whole (model.py:267-283 + :295-305 at hop = L-1,
 * i = 0, n = 0) fused with SumAggregator_urh_matrix._call (aggregators.py:98-146):
 * for node task t = (b, n) with entity x = node_ids[t]:
 *     y_k = adj_entity[x,k], r_k = adj_relation[x,k]                 (tf.gather, model.py:251-252)
 *     p   = softmax_k(rel_score[r_k])   (or p_k = 1 when rel_score == NULL: aggregators.py:148-152)
 *     S   = sum_k p_k * table[y_k, :]                                (embedding_lookup + weighted sum)
 *     agg = (S . Wc + (sum_k p_k) * c_child[b, :]) / K   (Wc == NULL: agg = S / K)
 *           == reduce_mean_k(p_k * ((table[y_k]+q_b) . W_L + b_L)) by linearity, c_child = q_b.W_L + b_L
 *     out[t, :] = relu((self_vec[t, :] + agg) . Wagg + bagg)
 * The K^L child rows are never materialised.  probs ([T, K]) is optional (model.py:294,304). */
MVIN_API int mvin_gather_attn_fwd(const float* table, const int32_t* adj_entity, const int32_t* adj_relation,
                         const int32_t* node_ids, const float* rel_score,
                         const float* self_vec, const float* Wc, const float* c_child,
                         const float* Wagg, const float* bagg,
                         int B, int N, int K, int D, int n_entity,
                         float* out, float* probs, void* stream);

/* Two deepest levels in one pass (hot kernel; MFMA dense phases).  For every PARENT node
 * (level L-2; P = B * parents_per_pair of them, entity ids in parent_ids) with children
 * x1[n] = adj_entity[parent, n] and grandchildren y[n,k] = adj_entity[x1[n], k]:
 *     p[n,:]  = softmax_k(t0[adj_relation[x1[n], k]])                       (or uniform, t0 NULL)
 *     c1 = q[b] . W1 + b1, c2 = q[b] . W2 + b2        (the broadcast query of model.py:277-279)
 *     self1[n] = table[x1[n]] . W1 + c1                                     (or table[x1[n]], W1 NULL)
 *     Z[n]    = self1[n] + ((sum_k p[n,k] table[y[n,k]]) . W2 + (sum_k p[n,k]) c2) / K
 *     out1[n] = relu(Z[n] . A0 + a0)            = aggregator (0,.) at hop L-1  (aggregators.py:98-146)
 *     p0 = softmax_n(t0[adj_relation[parent, n]]), p1 = softmax_n(t1[adj_relation[parent, n]])
 *     nagg0[parent] = (1/K) sum_n p0[n] self1[n]   -> neighbors_agg of aggregator (0,.) at hop L-2
 *     nagg1[parent] = (1/K) sum_n p1[n] out1[n]    -> neighbors_agg of aggregator (1,.) at hop L-2
 * (model.py:295-305 with i = 0 and i = 1).  probs_parent [P,K] = p0 and probs_child [P*K,K] = p are
 * optional (model.py:294,304).  table_bf16 = 1: the entity table holds bf16 rows (BASELINE config C5);
 * arithmetic stays fp32.  Returns -3 when (D, K) is outside the fused kernel's range
 * (D in {16,32,64,128}, K a power of two in [4,256]); callers then use the per-level entry points. */
MVIN_API int mvin_gather_attn_l2_fwd(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                            const int32_t* parent_ids, const float* t0, const float* t1,
                            const float* W1, const float* W2, const float* b1, const float* b2,
                            const float* q, const float* A0, const float* a0,
                            int B, int parents_per_pair, int K, int D, int n_entity, int nR,
                            float* nagg0, float* nagg1, float* probs_parent, float* probs_child,
                            int table_bf16, void* stream);
/* The same with the parents given as int64 ids read in place (the item ids of the reference's int64 placeholder,
 * model.py:50: at tree depth 2 the parents ARE the batch's items, so no id-conversion launch precedes the kernel). */
MVIN_API int mvin_gather_attn_l2_fwd_i64(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                                const int64_t* parent_ids, const float* t0, const float* t1,
                                const float* W1, const float* W2, const float* b1, const float* b2,
                                const float* q, const float* A0, const float* a0,
                                int B, int parents_per_pair, int K, int D, int n_entity, int nR,
                                float* nagg0, float* nagg1, float* probs_parent, float* probs_child,
                                int table_bf16, void* stream);
MVIN_API int mvin_gather_attn_l2_supported(int D, int K);
/* The same pass over an adjacency in the DUPLICATE-SLOT ENCODING of mvin_encode_adjacency (below): the reference's sampler
 * repeats (neighbour, relation) slots whenever an entity has fewer than K edges (data_loader_user_set.py:383-384); equal
 * slots have equal logits and equal rows, so their softmax weights are added up and every distinct row is gathered once,
 * and the distinct children of consecutive parents are packed into full MFMA tiles (mvin_fused_packed.hip).  Same
 * arithmetic up to the order of fp32 additions; no attention outputs (those are per slot: use the plain adjacency).
 * parent_ids: int32 [P], or int64 [P] read in place when parent_ids_i64 != 0.  D in {32, 64, 128}, K in {16, 32, 64, 128},
 * nR <= 4096, n_entity <= 2^24, tables below 4 GiB (-3 otherwise). */
MVIN_API int mvin_gather_attn_l2_enc_fwd(const void* table, const int32_t* enc_entity, const int32_t* enc_relation,
                                const void* parent_ids, int parent_ids_i64, const float* t0, const float* t1,
                                const float* W1, const float* W2, const float* b1, const float* b2,
                                const float* q, const float* A0, const float* a0,
                                int B, int parents_per_pair, int K, int D, int n_entity, int nR,
                                float* nagg0, float* nagg1, int table_bf16, void* stream);
MVIN_API int mvin_gather_attn_l2_enc_supported(int D, int K);
/* The encoded pass over PROJECTED tables.  Everything the pass applies to a gathered row before the ReLU of
 * aggregators.py:116 is linear -- the user-oriented projection (model.py:270-283) and the aggregator's matrix (aggregators.py:
 * 108-116) -- and the attention weights are scalars, so the matrices move from the gathered rows to the table:
 *     self1      = (E[x1] + q) W1 + b1                                  = T1[x1] + u1
 *     Z A0 + a0  = (self1 + (sum_k w_k E[y_k] + c q) W2 + c b2) A0 + a0 = TA1[x1] + sum_k w_k TA2[y_k] + v        (c = sum_k w_k)
 * T1 = E W1, TA1 = E W1 A0, TA2 = E W2 A0 (once per ENTITY);  u1 = q W1 + b1, v = q (W1 + c W2) A0 + (b1 + c b2) A0 + a0 (once
 * per PARENT, inside the kernel).  An exact re-association, like project-after-sum and the duplicate-slot encoding: the kernel
 * gathers the same ids and grandchild rows per pair (one more self row per distinct child) and has no D x D product per
 * distinct child left.  mvin_project_tables writes the three tables and the per-call parameter block into `ws`
 * (mvin_project_tables_elems floats; fp32 entity table; `attention` = whether t0 will be given: it fixes c = 1/K or 1) and
 * must be called again whenever E, W1, W2, b1, b2, A0 or a0 changed -- mvin_score_l2_fwd and mvin_amd.MVIN call it in
 * every pass.  mvin_gather_attn_l2_prj_fwd: as mvin_gather_attn_l2_enc_fwd with `ws` in place of the table and the weights;
 * D in {32, 64, 128}, tables below 1 GiB each.  adjacency_encoded = 0: enc_entity / enc_relation are the PLAIN adjacency
 * (D = 32, K in {8, 16} only: the wave-per-parent kernel, BASELINE config C2's, reads either form).  mvin_project_rows is the plain two-matrix form (out[0] = src W1 (+ b1),
 * out[1] = src W2 (+ b2)). */
MVIN_API int mvin_gather_attn_l2_prj_supported(int D, int K, int adjacency_encoded, int n_entity, int nR);   /* 1: _prj_fwd takes these tables */
MVIN_API int mvin_project_rows(const float* src, int64_t rows, int D, const float* W1, const float* W2, const float* b1,
                               const float* b2, float* out, void* stream);
MVIN_API size_t mvin_project_tables_elems(int n_entity, int D);
MVIN_API int mvin_project_tables(const float* entity_emb, const float* W1, const float* W2, const float* b1, const float* b2,
                                 const float* A0, const float* a0, int attention, int K, int n_entity, int D, float* ws,
                                 void* stream);
MVIN_API int mvin_gather_attn_l2_prj_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, int adjacency_encoded,
                                const void* parent_ids, int parent_ids_i64, const float* t0, const float* t1, const float* q,
                                int B, int parents_per_pair, int K, int D, int n_entity, int nR, float* nagg0, float* nagg1,
                                void* stream);
/* order [B] int32 = a permutation of 0 .. B-1 in which equal keys are neighbours (a partition by the key's low 14 bits, any order
 * inside a bucket: NOT a sort, and the order inside a bucket may differ from run to run).  keys: int64 [B] (low words read) or
 * int32 [B]; workspace: mvin_order_by_key_ws_elems(B) int32, rewritten by every call.  Three small launches (LDS histograms per
 * chunk, one scan, scatter), ~30 us per 524 288 keys whatever their skew. */
MVIN_API size_t mvin_order_by_key_ws_elems(int64_t B);
MVIN_API int mvin_order_by_key(const int64_t* keys_i64, const int32_t* keys_i32, int64_t B, int32_t* workspace, int32_t* order, void* stream);
/* The same launch with its parents taken in the order `order` (int32 [B], a permutation of the launch's parents; NULL = as given):
 * slot i works on parent order[i] -- its id, its query row, its rows of nagg0 / nagg1 -- so the results do not depend on it.
 * Pairs that share an item gather the same rows; next to each other (mvin_order_by_key over the item ids) their loads are cache
 * hits instead of trips past the L2.  Taken by the wave-per-parent kernel only: encoded adjacency, D = 64, K <= 32,
 * parents_per_pair = 1 (-3 otherwise). */
MVIN_API int mvin_gather_attn_l2_prj_ordered_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, int adjacency_encoded,
                                const void* parent_ids, int parent_ids_i64, const int32_t* order, const float* t0, const float* t1,
                                const float* q, int B, int parents_per_pair, int K, int D, int n_entity, int nR, float* nagg0,
                                float* nagg1, void* stream);
/* PER-ENTITY AGGREGATES of the projected tables (D = 64, K in {16, 32, 64}, encoded adjacency): the two deepest levels of
 * MVIN.aggregate_delta_whole (model.py:267-283 + :295-305, aggregators.py:98-146) once more re-associated.  With one logit per relation
 * (aggregators.py:118-146: User_orient_rela scores a relation, not a user) the softmax weights w(e)_k of entity e's slots under
 * aggregator (0,.) belong to the ENTITY, and so does everything of the formulas above that does not carry the query:
 *     G[e]  = TA1[e] + sum_k w(e)_k TA2[y_ek]          S0[e] = sum_k w(e)_k T1[y_ek]            (once per ENTITY and call)
 *     nagg0 = S0[x] + c0 u1                            nagg1 = sum_c (p1_c / K) relu(G[x_c] + v)  (per parent x with children x_c)
 * -- the same sums in another association (the order of the additions differs: results agree to rounding, not bit for bit).  A
 * parent gathers its distinct children's G rows and one S0 row instead of its grandchildren's rows of three tables: ~12 rows
 * instead of ~120 at BASELINE C3.  mvin_entity_aggregates writes S0 | G ([2][n_entity][D] fp32, mvin_entity_aggregates_elems
 * floats) from `ws` (mvin_project_tables' output, the CURRENT call's), the encoded adjacency and t0 (NULL: plain mean); it costs
 * ~17 gathered rows per entity and must follow every mvin_project_tables.  mvin_gather_attn_l2_agg_fwd: as
 * mvin_gather_attn_l2_prj_ordered_fwd (same arguments and outputs; t0 is only tested for presence) with `agg` beside `ws`.
 * MVIN_L2_AGG=0 in the environment makes _supported answer 0 (A/B against the kernels over the tables themselves). */
MVIN_API int mvin_gather_attn_l2_agg_supported(int D, int K, int n_entity, int nR);
MVIN_API size_t mvin_entity_aggregates_elems(int n_entity, int D);
MVIN_API int mvin_entity_aggregates(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, const float* t0, int K, int D,
                                    int n_entity, int nR, float* agg, void* stream);
MVIN_API int mvin_gather_attn_l2_agg_fwd(const float* ws, const float* agg, const int32_t* enc_entity, const int32_t* enc_relation,
                                const void* parent_ids, int parent_ids_i64, const int32_t* order, const float* t0, const float* t1,
                                const float* q, int B, int parents_per_pair, int K, int D, int n_entity, int nR, float* nagg0,
                                float* nagg1, void* stream);
/* FOLDED-TAIL form of everything above key addressing for n_mix_hop = 1, h_hop = 2, User_orient on -- MVIN.aggregate_delta_whole,
 * model.py:259-324, with SumAggregator_urh_matrix._call, aggregators.py:98-146, and the score of model.py:158-159 (mvin_l2_tail_fwd's formulas
 * over the aggregates above; D = 64, K in {16, 32, 64} -- and D = 32, K in {16, 32}, where the aggregates exist as this form's H0 | G only).
 * nagg0 only ever enters through (ev0 + nagg0) A0 + a0, and ev0 through that and
 * the combiner, so with c = the sum of a row's slot weights over K (1/K with attention, 1 without):
 *     (ev0 + nagg0) A0 + a0 = H0[x] + q Wq + bq        H0[e] = E[e] W0 A0 + sum_k w(e)_k TA1[y_ek],  Wq = (W0 + c W1) A0,
 *                                                       bq = (b0 + c b1) A0 + a0                      (S0[e] A0 = sum_k w(e)_k TA1[y_ek])
 *     ev0 Wm0               = M0[x] + q Wqm + b0 Wm0    M0 = E W0 Wm0,  Wqm = W0 Wm0   (Wm0 | Wm1 | Wm2 = the row blocks of Wmix)
 * mvin_fold_tables builds, per call and from the current parameters, TA1 | TA2 | T0A = E W0 A0 | M0 (one four-matrix table build),
 * H0 | G (mvin_entity_aggregates' kernel) and the parameter block into `ws` (mvin_fold_tables_elems floats).
 * mvin_score_l2_folded_fwd then scores B pairs in ONE launch (a batch of 16 pairs per wave from item id and query row to score):
 *     out0 = relu(H0[x] + q Wq + bq) ;  Z2 = out0 + sum_c (p1_c / K) relu(G[x_c] + q Wv + bv)            (kept in LDS)
 *     out2 = relu(Z2 A1 + a1) ;  item_emb = M0[x] + q Wqm + out0 Wm1 + out2 Wm2 + bmix + b0 Wm0 ;  scores = <user_o, item_emb>
 * -- six D x D products per pair instead of eight, the same sums in another association (agreement to rounding).  item_emb / sig may
 * be NULL; so may out0 / z2 ([B, D] scratch rows): only the two-launch A/B variant (MVIN_L2_FOLD_TWO=1 in the environment: pair kernel +
 * four-product tile kernel) writes them, the default is ONE launch that keeps them in LDS.  _supported: the shapes above. */
MVIN_API int mvin_score_l2_folded_supported(int D, int K, int n_entity, int nR);
MVIN_API size_t mvin_fold_tables_elems(int n_entity, int D);
MVIN_API int mvin_fold_tables(const float* entity_emb, const int32_t* enc_entity, const int32_t* enc_relation, const float* t0, const float* W0,
                              const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* A0,
                              const float* a0, const float* Wmix, const float* bmix, const float* A1, int K, int D, int n_entity, int nR,
                              float* ws, void* stream);
/* The same with EVERY PAIR GATHERING ITS OWN ROWS (D = 64, K in {16, 32}): no per-entity sums -- mvin_fold_tables_ex with aggregates = 0
 * builds the four per-row tables only (SURVEY 7.3-c's route 2: the matrices moved to the tables, the same rows gathered), and
 * mvin_score_l2_folded_gather_fwd walks a pair's distinct children and grandchildren like mvin_gather_attn_l2_prj_ordered_fwd
 * (nagg0 A0 = sum_c (p0_c / K) TA1[x_c]: one T1 row per child less) and finishes the pair in the same launch.  `order`: as there.
 * MVIN_L2_FOLD_GATHER=0 in the environment makes _supported answer 0 (A/B: wave-per-parent kernel + mvin_l2_tail_fwd). */
MVIN_API int mvin_fold_tables_ex(const float* entity_emb, const int32_t* enc_entity, const int32_t* enc_relation, const float* t0, const float* W0,
                                 const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* A0,
                                 const float* a0, const float* Wmix, const float* bmix, const float* A1, int aggregates, int K, int D,
                                 int n_entity, int nR, float* ws, void* stream);
MVIN_API int mvin_score_l2_folded_gather_supported(int D, int K, int n_entity, int nR);
MVIN_API int mvin_score_l2_folded_gather_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, const int64_t* items_i64,
                                             const int32_t* items_i32, const int32_t* order, const float* t0, const float* t1, const float* q,
                                             const float* user_o, const float* A1, const float* a1, const float* Wmix, int64_t B, int K, int D,
                                             int n_entity, int nR, float* item_emb, float* scores, float* sig, void* stream);
MVIN_API int mvin_score_l2_folded_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, const int64_t* items_i64,
                                      const int32_t* items_i32, const float* t0, const float* t1, const float* q, const float* user_o,
                                      const float* A1, const float* a1, const float* Wmix, int64_t B, int K, int D, int n_entity, int nR,
                                      float* out0, float* z2, float* item_emb, float* scores, float* sig, void* stream);
/* Which kernel mvin_gather_attn_l2_fwd takes for a call of this shape: 0 = none (returns -3), 1 = the symmetric
 * fused kernel (every wave gathers and multiplies; the only one that writes probs_parent / probs_child),
 * 2 = the role-split pipeline (gather waves + MFMA waves; D in {32,64,128}, K in {16 (D=32), 32, 64, 128}, no
 * probs, adjacency and outputs below 2 GiB), 3 = the wave-per-parent kernel for D = 16, K in {4, 8, 16} (the
 * reference's shipped settings: no workgroup phases at all; no probs, table below 4 GiB), 4 = the wave-per-parent
 * kernel for D = 32, K in {8, 16} (BASELINE config C2; same conditions; it takes these shapes ahead of the pipeline).
 * n_parents = B * parents_per_pair.  For tests and benchmarks. */
MVIN_API int mvin_gather_attn_l2_variant(int D, int K, int64_t n_parents, int n_entity, int want_probs);     /* fp32 table */
MVIN_API int mvin_gather_attn_l2_variant_ex(int D, int K, int64_t n_parents, int n_entity, int want_probs, int table_bf16);
/* Measurement aid: the row gathers of mvin_gather_attn_l2_fwd and nothing else, written the plain way (one wave per
 * parent, 8 loads in flight per lane).  child_ids [n_parents, K] and grandchild_ids [n_parents, K*K] are levels 1 and 2
 * of mvin_expand_ids for the parents; every listed row is read once (the same 16-byte lane loads, ids fetched one round
 * ahead) and all its elements are added into sums[p].  bench.py times it on the timed region's own entity table and
 * pairs as a reference point for the fused kernel's row rate.  Row bytes (D * 4, or D * 2 with table_bf16) in
 * {64, 128, 256, 512}. */
MVIN_API int mvin_probe_gather_l2(const void* table, const int32_t* child_ids, const int32_t* grandchild_ids, int64_t n_parents, int K,
                         int D, int n_entity, int table_bf16, float* sums, void* stream);

/* SumAggregator_urh_matrix._call on materialised levels (every aggregator application
 * other than the deepest hop; aggregators.py:98-152, model.py:295-305):
 *     p = softmax_k(rel_score[rel_ids[t*K+k]])  (or 1), agg = (1/K) sum_k p_k * neigh[t*K+k, :]
 *     out[t, :] = relu((self_vec[t, :] + agg) . Wagg + bagg)
 * rel_ids == NULL with rel_score != NULL: rel_score holds one logit per child ([T*K]), the form
 * Aggregator.__call__ needs when it is handed relation VECTORS (aggregators.py:29-31). */
MVIN_API int mvin_agg_fwd(const float* self_vec, const float* neigh, const int32_t* rel_ids,
                 const float* rel_score, const float* Wagg, const float* bagg,
                 int B, int N, int K, int D, float* out, float* probs, void* stream);

/* One attention read over a user's ripple set (MVIN._key_addressing, model.py:161-240):
 *   mode 0 (hop loop, :210-230): s_m = E[score_ids[b,m]] . V[b, rel_ids[b,m], :] where
 *          V[b,r,:] = E[item_b] . R_KGE[r]  (== (R_KGE[r] . h_m) . item_b, :214-220);
 *   mode 1 (soft_attention_h_set, :162-197): s_m = E[score_ids[b,m]] . w[0:D]
 *          (the user term and the bias are constant over m and cancel in the softmax);
 *   o[b, :] = sum_m softmax_m(s)_m * E[value_ids[b,m], :]  written at out + b*ldo. */
MVIN_API int mvin_ripple_attn_fwd(const float* entity_emb, const int32_t* score_ids, const int32_t* rel_ids,
                         const int32_t* value_ids, const float* V, const float* w, int mode,
                         int B, int Nm, int D, int nR, float* out, int64_t ldo, void* stream);
/* The same for either table type (table_bf16 = 1: bf16 rows; arithmetic stays fp32): the general fallback
 * for every (D, Nm) the one-pass kernels of mvin_key_addressing_fwd do not take. */
MVIN_API int mvin_ripple_attn_fwd_ex(const void* entity_emb, const int32_t* score_ids, const int32_t* rel_ids,
                            const int32_t* value_ids, const float* V, const float* w, int mode,
                            int B, int Nm, int D, int nR, float* out, int64_t ldo, int table_bf16, void* stream);

/* All attention reads of MVIN._key_addressing for a batch in one pass (model.py:161-240):
 * out[b, :] = [ o_hset (if w != NULL) | o_hop0 | ... | o_hop{P-1} ], each D wide, row stride ldo.
 *   o_hset = sum_m softmax_m(E[mem_h[0][b,m]] . w[0:D])_m * E[mem_h[0][b,m]]                 (:162-197)
 *   o_hop  = sum_m softmax_m(E[mem_h[hop][b,m]] . V[b, mem_r[hop][b,m], :])_m * E[mem_t[hop][b,m]]  (:210-230)
 * mem_h/mem_r/mem_t are HOST arrays of max(1,P) device pointers ([B, Nm] int32 each).  Every table
 * row is read once.  Two kernels sit behind the entry point: a streaming LDS-DMA pipeline (P >= 1, Nm <= 64,
 * D in {16,32,64,128}, nR*D*4 <= 3072: rows go global -> LDS without VGPR staging, the h-set read is one
 * online-softmax pass) and a register-resident one (head rows stay in VGPRs between the logit and the
 * weighted-sum pass; ceil(Nm / (64/ceil_pow2(D/4))) <= 16).  Returns -3 when neither takes the shape;
 * callers then use mvin_ripple_attn_fwd per read.  mvin_key_addressing_supported(Nm, D) answers for the
 * register-resident kernel alone (it does not know nR): sufficient, not necessary. */
MVIN_API int mvin_key_addressing_fwd(const void* entity_emb, const float* V, const float* w,
                            const int32_t* const* mem_h, const int32_t* const* mem_r,
                            const int32_t* const* mem_t, int P, int B, int Nm, int D, int nR, int n_entity,
                            float* out, int64_t ldo, int table_bf16, void* stream);
MVIN_API int mvin_key_addressing_supported(int Nm, int D);
/* The same reads with the feed assembly of train.py:117-120 inside the kernel: pair b uses the ripple sets of user
 * users[b] straight out of user_triplet_set `uts` [n_user, max(1,P), 3, Nm] int32 on the device (h, r, t lists per hop;
 * data_loader_user_set.py:392-441) -- no per-pair [B, Nm] arrays exist.  One of users_i64 / users_i32 is given.  For
 * batches whose users repeat, mvin_key_addressing_grouped_fwd reads a user's rows once instead.
 * Device-resident ids are not validated per launch (the reference's CPU tf.gather raises InvalidArgument; here that
 * is the host wrapper's job): a user id outside [0, n_user) is CLAMPED into the table, never read out of bounds. */
MVIN_API int mvin_key_addressing_users_fwd(const void* entity_emb, const float* V, const float* w, const int32_t* uts,
                                  const int64_t* users_i64, const int32_t* users_i32, int P, int B, int Nm, int D, int nR,
                                  int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream);

/* The same reads for pairs GROUPED BY USER (the feeds of train.py:117-120 / util.py:208-230 give every pair its
 * user's ripple sets, so all pairs of one user gather the same rows).  `uts` = user_triplet_set on the device,
 * [n_user, max(1,P), 3, Nm] int32 (h | r | t per hop, data_loader_user_set.py:392-441).  The batch's pairs are
 * described in user order: segment s = pairs pair_index[seg_ptr[s] .. seg_ptr[s+1]) of user seg_user[s]
 * (pair_index: position in the caller's batch, i.e. row of `items` and of `out`).  One workgroup stages a
 * user's 2*P*Nm rows in LDS once, forms V[pair, r, :] = E[item] . R_KGE[r] with MFMA per 16 pairs (no [B,nR,D]
 * tensor) and runs the attention reads of mvin_key_addressing_fwd from LDS.  out as in mvin_key_addressing_fwd.
 * items are int64 (items_i64) or int32 (items_i32).  nseg_dev (optional, device) holds the actual segment count
 * when the caller built the segments on the device without a host sync; nseg is then an upper bound (array
 * sizes).  -3: shape outside the LDS budget. */
MVIN_API int mvin_key_addressing_grouped_fwd(const void* entity_emb, const float* relation_kge, const float* w,
                                    const int32_t* uts, const int32_t* seg_user, const int32_t* seg_ptr,
                                    const int32_t* nseg_dev, const int32_t* pair_index, const int64_t* items_i64,
                                    const int32_t* items_i32, int nseg, int B, int P, int Nm, int D, int nR, int n_entity, int n_user,
                                    float* out, int64_t ldo, int table_bf16, void* stream);
MVIN_API int mvin_key_addressing_grouped_supported(int D, int P, int Nm, int nR);
/* STATIC per-user records for the grouped form (ABI v8).  A user's ripple sets are built once per data set
 * (data_loader_user_set.py: user_triplet_set feeds the same (h, r, t) ids for a user in every batch, model.py:66-76), and so
 * is everything the dense grouped kernel derives from the ids alone: the relation buckets of the memories (which rows share
 * an R_KGE[r], model.py:214-216), its tile table, the clamped head / tail ids.  mvin_build_user_records writes them once,
 * one record of mvin_user_records_len(P, Nm, nR) int32 words per user (0: no record form for this shape):
 *   [0] tiles  [4 ..] members per relation | first bucket row per relation | relation of each tile |
 *   bucket slots -> row (hop * NmP + m, inside a bucket in row order; -1: empty) | head id per row | tail id per row
 *   (NmP = Nm rounded up to 16; ids clamped to [0, n_entity); padding rows and unused words -1; whole 256-byte lines),
 * and mvin_key_addressing_grouped_rec_fwd (same arguments and results as mvin_key_addressing_grouped_fwd, bit for bit) lands a
 * user's record in LDS a segment ahead instead of bucketing the segment's ids.  user_records == NULL, or a shape / table
 * type outside mvin_user_records_supported: exactly mvin_key_addressing_grouped_fwd. */
MVIN_API int mvin_user_records_len(int P, int Nm, int nR);
MVIN_API int mvin_user_records_supported(int D, int P, int Nm, int nR, int table_bf16);
MVIN_API int mvin_build_user_records(const int32_t* uts, int n_user, int P, int Nm, int nR, int n_entity, int32_t* records, void* stream);
MVIN_API int mvin_key_addressing_grouped_rec_fwd(const void* entity_emb, const float* relation_kge, const float* w,
                                        const int32_t* uts, const int32_t* user_records, const int32_t* seg_user,
                                        const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                        const int64_t* items_i64, const int32_t* items_i32, int nseg, int B, int P, int Nm, int D,
                                        int nR, int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream);
/* The same pass with the users' U rows GATHERED.  U_m = R_KGE[r_m] . E[h_m] (model.py:214-220 re-associated) depends on the
 * (relation, entity) pair alone: mvin_project_relations writes it for every pair -- ws = [nR, nE, D] fp32, then E[e] . w (the
 * h-set read's logits) and scratch; mvin_project_relations_elems floats; from the CURRENT parameters, to be called again whenever
 * E, R_KGE or w changed (mvin_score_l2_fwd and mvin_amd.MVIN call it in every pass) -- and the kernel over the user records
 * loads a user's rows from there instead of multiplying them (half of the records kernel's matrix work).  D = 64, fp32 table,
 * shapes of mvin_user_records_supported, at most 64 memories per hop when w is given, nR * n_entity < 2^31
 * (mvin_key_addressing_grouped_er_supported). */
MVIN_API size_t mvin_project_relations_elems(int n_entity, int nR, int D);
MVIN_API int mvin_project_relations(const float* entity_emb, const float* relation_kge, const float* w, int n_entity, int nR, int D,
                                    float* ws, void* stream);
MVIN_API int mvin_key_addressing_grouped_er_supported(int D, int P, int Nm, int nR, int n_entity, int has_set);
MVIN_API int mvin_key_addressing_grouped_er_fwd(const void* entity_emb, const float* relation_kge, const float* w,
                                       const int32_t* uts, const int32_t* user_records, const float* er_ws, const int32_t* seg_user,
                                       const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                       const int64_t* items_i64, const int32_t* items_i32, int nseg, int B, int P, int Nm, int D,
                                       int nR, int n_entity, int n_user, float* out, int64_t ldo, void* stream);
/* MVIN._key_addressing (model.py:161-240) AND the user MLP behind it (model.py:232-236) for pairs grouped by user, as ONE
 * barrier-free kernel: every wave walks tiles of up to 32 pairs of one user on its own -- logits, softmax and reads of each hop
 * as TRANSPOSED products on the matrix cores, the accumulator of the first the B operand of the second, every gathered row
 * loaded straight into its operand layout once per tile, nothing staged in LDS (mvin_keyaddr_flash.hip).  Two exact
 * re-associations move the per-row matrix products to per-call TABLES: U_m = R_KGE[r_m] . E[h_m] depends on (relation, entity)
 * alone, and the user MLP is linear in the tail rows,
 *     user_o = bias + sum_m p_hset[m] TW_0[h_m] + sum_hop sum_m p_hop[m] TW_{1+hop}[t_m],   TW_j = E . user_mlp_W[D j : D j + D, :].
 * mvin_key_addressing_flash_prepare writes them from the CURRENT parameters into `ws` (mvin_key_addressing_flash_tables_elems
 * floats: [nR, nE, D] R_KGE[r] . E[e] | E[e] . w | scratch | [P + has_set][nE, D] TW); it must be called again whenever E, R_KGE,
 * w or user_mlp_W changed (mvin_score_l2_fwd and mvin_amd.MVIN call it in every pass).  w != NULL: the h-set read
 * (model.py:162-197) is part of the pass (has_set).  user_records: mvin_build_user_records.  sched_ws:
 * mvin_key_addressing_flash_ws_elems(B, n_user) int32 words of scheduling scratch, rewritten by every call.  D = 64, fp32
 * tables, 1 <= P <= 8, Nm <= 64, nR * n_entity < 2^31 (mvin_key_addressing_flash_supported; -3 otherwise). */
MVIN_API int mvin_key_addressing_flash_supported(int D, int P, int Nm, int nR, int n_entity);
MVIN_API size_t mvin_key_addressing_flash_tables_elems(int n_entity, int nR, int D, int P, int has_set);
MVIN_API int mvin_key_addressing_flash_prepare(const float* entity_emb, const float* relation_kge, const float* w,
                                      const float* user_mlp_W, int n_entity, int nR, int D, int P, float* ws, void* stream);
MVIN_API size_t mvin_key_addressing_flash_ws_elems(int64_t B, int n_user);
MVIN_API int mvin_key_addressing_flash_fwd(const float* entity_emb, const float* ws, const int32_t* user_records,
                                  const int32_t* seg_user, const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                  const int64_t* items_i64, const int32_t* items_i32, int64_t B, int P, int Nm, int D, int nR,
                                  int n_entity, int n_user, int has_set, const float* user_mlp_b, float* user_o,
                                  int32_t* sched_ws, void* stream);
/* The batch in user order for mvin_key_addressing_grouped_fwd, built on the device (counting sort by user id, int
 * atomics; no host sync): seg_user [>= min(B, n_user)] = the users that occur, increasing; seg_ptr [>= min(B, n_user) + 1]
 * = first position of each user's pairs (+ the total at [nseg]); nseg [1]; pair_index [B] = original index of the pair
 * at each position (order inside a segment unspecified).  workspace: 2 * n_user + B int32 (ABI v6: + B for the per-pair ranks).  Ids outside [0, n_user) are
 * CLAMPED into the table (every pair keeps a segment, so no row of `out` stays unwritten); the reference's CPU
 * tf.gather would raise InvalidArgument -- the Python wrapper does that for host feeds and, on request, for device
 * feeds (MVIN.validate_device_ids). */
MVIN_API int mvin_group_pairs_by_user(const int64_t* users_i64, const int32_t* users_i32, int64_t B, int n_user, int32_t* workspace,
                             int32_t* seg_user, int32_t* seg_ptr, int32_t* nseg, int32_t* pair_index, void* stream);

/* out[r, :] = softmax(x[r, :]) over n columns (tf.nn.softmax, model.py:189 / :223).  Building block of the
 * SHARED-USER form of MVIN._key_addressing (one user scored against many items, as util.py:145-181 does for
 * top-K evaluation): with one ripple set for the whole batch the reads become dense products,
 *   logits = E[items] . A^T,  A[m] = R_KGE[r_m] . E[h_m]   ->  row softmax  ->  o = P . E[t]   (mvin_linear_fwd),
 * instead of 2*Nm row gathers per pair. */
MVIN_API int mvin_row_softmax_fwd(const float* x, int64_t rows, int n, float* out, void* stream);

/* Aggregator._mix_neighbor_vectors / _mix_neighbor_vectors_urv (aggregators.py:37-77: KGCN's user-relation mixer; defined by
 * the reference, never called by MVIN) on materialised tensors:
 *   s[b,n,k] = mean_d(user_embeddings[b,d] * neighbor_relations[b,n,k,d]) ; p = softmax_k(s) ;
 *   out[b,n,:] = mean_k(p[b,n,k] * neighbor_vectors[b,n,k,:])
 * neighbor_vectors / neighbor_relations [B, N, K, D], user_embeddings [B, D], out [B, N, D], probs [B, N, K] or NULL.  K <= 64.
 * logits_or_null [B, N, K]: the scores s are taken from there instead (SumAggregator_urh_matrix._mix_neighbor_vectors_urh,
 * aggregators.py:118-146, with s = relation . urh_weights[D:2D]: the user and self terms cancel in the softmax); with neither
 * logits nor relations / user every weight is 1 (_mix_neighbor_vectors_no_ur, :148-152: the plain mean over K). */
MVIN_API int mvin_mix_neighbor_vectors_fwd(const float* neighbor_vectors, const float* neighbor_relations, const float* user_embeddings,
                                           const float* logits_or_null, int B, int N, int K, int D, float* out, float* probs_or_null,
                                           void* stream);

/* Everything of MVIN.aggregate_delta_whole (model.py:259-324) above mvin_gather_attn_l2_fwd for the shape
 * n_mix_hop = 1, h_hop = 2 (tree depth 2), in one launch:
 *   ev0 = (E[item] + q) W0 + b0  (W0 == NULL: ev0 = E[item], User_orient off)        model.py:270-283
 *   out0 = relu((ev0 + nagg0) A0 + a0) ; out2 = relu((out0 + nagg1) A1 + a1)          aggregators.py:108-116
 *   item_emb = [ev0 | out0 | out2] Wmix + bmix                                        model.py:310-315
 *   scores = sum_d user_o * item_emb ; sig = sigmoid(scores)                          model.py:158-159
 * nagg0 / nagg1 [B, D]: the outputs of mvin_gather_attn_l2_fwd with parents_per_pair = 1.  D in {16, 32, 64}
 * (-3 otherwise: use mvin_linear_fwd per stage).  item_emb and sig may be NULL. */
MVIN_API int mvin_l2_tail_fwd(const void* entity_emb, const int64_t* items_i64, const int32_t* items_i32, const float* q,
                     const float* user_o, const float* nagg0, const float* nagg1, const float* W0, const float* b0,
                     const float* A0, const float* a0, const float* A1, const float* a1, const float* Wmix,
                     const float* bmix, int64_t B, int D, int n_entity, float* item_emb, float* scores, float* sig,
                     int table_bf16, void* stream);
MVIN_API int mvin_l2_tail_supported(int D);

/* The whole get_scores pass (model.py:125-159, default wiring: preference sets AND high-order part, query =
 * user_o, wide_deep, n_mix_hop = 1, h_hop = 2) enqueued by ONE native call: at the reference's own batch sizes
 * (512 / 1024, src/bash/mvin_*.sh) the pass is a handful of short kernels and the host-side cost of issuing them
 * one foreign call at a time dominates.  Sequence (each step is the entry point of the same name):
 *   mvin_linear_fwd (V = E[item] . R_KGE[r]) -> mvin_key_addressing_fwd (or _users_fwd; or, with group_ws, mvin_group_pairs_by_user ->
 *   mvin_key_addressing_grouped_fwd in place of both) -> mvin_linear_fwd (user MLP, :232-236)
 *   -> mvin_expand_ids (level 0) -> mvin_gather_attn_l2_fwd -> mvin_l2_tail_fwd.
 * All pointers are device pointers except mem_h / mem_r / mem_t (host arrays of max(1,P) device pointers).
 * Workspace and outputs are caller-owned.  Returns the first failing step's code. */
typedef struct {
    const void* entity_emb;        /* [nE, D] fp32 or bf16 */
    const int32_t* adj_entity;     /* [nE, K] */
    const int32_t* adj_relation;
    const float* relation_kge;     /* [nR, D, D] */
    const float* h_set_w;          /* [D] or NULL (PS_O_ft off) */
    const float* user_mlp_W;       /* [(P + (h_set_w != NULL)) * D, D] */
    const float* user_mlp_b;
    const float* t0;               /* [nR] relation logits of aggregator (0,0) / (1,0) or NULL (User_orient_rela off) */
    const float* t1;
    const float* W0;               /* projections of levels 0, 1, 2 or all NULL (User_orient off) */
    const float* b0;
    const float* W1;
    const float* b1;
    const float* W2;
    const float* b2;
    const float* A0;
    const float* a0;
    const float* A1;
    const float* a1;
    const float* Wmix;             /* [3D, D] */
    const float* bmix;
    const int64_t* items;          /* [B] */
    const int32_t* const* mem_h;
    const int32_t* const* mem_r;
    const int32_t* const* mem_t;
    const int32_t* uts;            /* user_triplet_set [n_user, P, 3, Nm] + users [B]: instead of mem_h / mem_r / mem_t */
    const int64_t* users;          /*   (mvin_key_addressing_users_fwd); NULL = the per-pair arrays above */
    float* V;                      /* workspace [B, nR, D] */
    float* o_cat;                  /* workspace [B, (P + (h_set_w != NULL)) * D] */
    int32_t* parents;              /* unused since the depth-2 kernel reads `items` in place; kept for layout */
    float* nagg0;                  /* workspace [B, D] */
    float* nagg1;
    float* user_o;                 /* out [B, D] */
    float* item_emb;               /* out [B, D] */
    float* scores;                 /* out [B] */
    float* sig;                    /* out [B] */
    int64_t B;
    int D, K, P, Nm, n_entity, n_relation, table_bf16;
    int n_user;                    /* rows of uts (users feed); user ids are clamped to [0, n_user) */
    const int32_t* enc_entity;     /* duplicate-slot encoding of the adjacency (mvin_encode_adjacency) or NULL: when given */
    const int32_t* enc_relation;   /*   the two deepest levels take mvin_gather_attn_l2_enc_fwd */
    int32_t* group_ws;             /* users feed only, or NULL: workspace of 2 * n_user + 4 * B + 3 int32 -> key addressing in its
                                      GROUPED form (mvin_group_pairs_by_user + mvin_key_addressing_grouped_fwd instead of the V
                                      projection + mvin_key_addressing_users_fwd; V may then be NULL) */
    const int32_t* user_records;   /* grouped form only, or NULL: mvin_build_user_records(uts) -> mvin_key_addressing_grouped_rec_fwd */
    int depth;                     /* mvin_score_small_fwd only: tree depth h_hop (n_mix_hop = 1): 0 or 2 = two hops; 1 = ONE hop
                                      (BASELINE configs[0]: W2 / b2 / A1 / a1 / t1 unused and may be NULL, Wmix is [2D, D]).
                                      mvin_score_l2_fwd ignores it (depth 2) */
    float* prj_tables;             /* mvin_score_l2_fwd only, or NULL: workspace of mvin_project_tables_elems(nE, D) floats -- with an
                                      fp32 table, the encoded adjacency and the projection on (W1), the two deepest levels run in their
                                      PROJECTED-TABLES form: mvin_project_tables -> mvin_gather_attn_l2_prj_fwd.  Rewritten by every
                                      call (nothing cached) */
    float* ka_er;                  /* mvin_score_l2_fwd only, or NULL: workspace of mvin_project_relations_elems(nE, nR, D) floats -- with
                                      user_records and an fp32 table the grouped key addressing runs in its GATHERED form:
                                      mvin_project_relations -> mvin_key_addressing_grouped_er_fwd.  Rewritten by every call */
    float* ka_flash;               /* mvin_score_l2_fwd only, or NULL: workspace of mvin_key_addressing_flash_tables_elems(nE, nR, D, P,
                                      h_set_w != NULL) floats -- with user_records, group_ws and an fp32 table of a shape
                                      mvin_key_addressing_flash_supported takes, key addressing AND the user MLP run as
                                      mvin_key_addressing_flash_prepare -> mvin_key_addressing_flash_fwd (o_cat is then not written; the
                                      scheduling scratch is the part of group_ws the grouping leaves behind).  Takes precedence over
                                      ka_er.  Rewritten by every call */
    int32_t* item_order_ws;        /* mvin_score_l2_fwd only, or NULL: mvin_order_by_key_ws_elems(B) + B int32 -- when the two deepest levels
                                      run as the wave-per-parent kernel over projected tables (D = 64, K <= 32, encoded adjacency), its
                                      parents are taken in ITEM order (mvin_order_by_key -> mvin_gather_attn_l2_prj_ordered_fwd): pairs of
                                      the same item back to back, their identical rows cache hits.  Results do not depend on it */
    float* agg_tables;             /* mvin_score_l2_fwd only, or NULL: workspace of mvin_entity_aggregates_elems(nE, D) floats -- with
                                      prj_tables, the encoded adjacency and a shape mvin_gather_attn_l2_agg_supported takes, the two deepest
                                      levels run as mvin_project_tables -> mvin_entity_aggregates -> mvin_gather_attn_l2_agg_fwd (in item
                                      order when item_order_ws is given).  Rewritten by every call */
    float* fold_ws;                /* mvin_score_l2_fwd only, or NULL: workspace of mvin_fold_tables_elems(nE, D) floats -- with the encoded
                                      adjacency, User_orient on, an fp32 table and a shape mvin_score_l2_folded_supported takes, everything
                                      above key addressing runs as mvin_fold_tables -> mvin_score_l2_folded_fwd (nagg0 / nagg1 are its
                                      scratch rows); takes precedence over prj_tables / agg_tables.  Rewritten by every call */
    int fold_gather;               /* with fold_ws: 1 = the form in which every pair gathers its own rows (mvin_fold_tables_ex without
                                      aggregates -> mvin_score_l2_folded_gather_fwd, in item order when item_order_ws is given) */
} mvin_score_l2_args;
MVIN_API int mvin_score_l2_fwd(const mvin_score_l2_args* args, void* stream);

/* The same pass as ONE KERNEL LAUNCH, for the batch sizes the reference itself calls the path with (512 / 1024 pairs per
 * sess.run: train.py:62-64, util.py:44-56, src/bash/mvin_*.sh; SURVEY 8(d) sweeps 512 .. 16 384): a workgroup takes
 * `group` consecutive pairs from their ids to their scores -- V projection, attention reads over the ripple sets
 * (model.py:161-240), user MLP, the two-level neighbor gather + attention (:259-305, aggregators.py:98-146) and the
 * mix-hop tail + score (:286-317, :158-159) with nothing but LDS in between.  Same argument block as mvin_score_l2_fwd;
 * the workspaces (V, o_cat, parents, nagg0, nagg1, group_ws, user_records) are not used and may be NULL.  Per-pair feed
 * (mem_h / mem_r / mem_t) or users feed (uts + users); plain adjacency or, with enc_entity / enc_relation, its
 * duplicate-slot encoding (distinct rows only).  group <= 0: chosen from the batch size (1 .. 16 pairs per workgroup).
 * args->depth = 1: the one-hop tree (model.py:286-317 with h_hop = 1: aggregator (0,0) at hop 0, combiner over [ev0 | out0]).
 * D in {16, 32, 64}, K <= 64, P in 1..4, Nm <= 64, fp32 table and adjacency below 4 GiB: -3 otherwise
 * (mvin_score_small_supported). */
MVIN_API int mvin_score_small_fwd(const mvin_score_l2_args* args, int group, void* stream);
MVIN_API int mvin_score_small_supported(int D, int K, int P, int Nm, int nR);

/* Row movers of the multi-GPU layer (mvin_amd/dist.py; no reference counterpart -- the reference is single
 * device): out[i, :] = table[ids[i], :] (gather) and table[ids[i], :] = rows[i, :] (scatter; ids distinct),
 * rows of `row_bytes` bytes (a multiple of 4: fp32 or bf16 entity rows move untouched). */
MVIN_API int mvin_gather_rows(const void* table, const int32_t* ids, int64_t n, int row_bytes, void* out, void* stream);
MVIN_API int mvin_scatter_rows(void* table, const int32_t* ids, int64_t n, int row_bytes, const void* rows, void* stream);
/* Entity ids into the sharded table's id space (mvin_amd/dist.py: cyclic ownership, owner(x) = x mod world, rank r's rows
 * contiguous): out[i] = (ids[i] mod world) * n_local + ids[i] div world.  ids / out: int64 when ids_are_i64, else int32
 * (same type in and out; out may alias ids).  One launch instead of four elementwise torch kernels per step. */
MVIN_API int mvin_shard_space_ids(const void* ids, int ids_are_i64, int64_t n, int world, int n_local, void* out, void* stream);

/* Entity-table ("hoisted") mode building block -- an inference-side re-association of
 * model.py:295-305 / aggregators.py:118-146 at the two deepest levels (SURVEY.md 7.3-c route 2b):
 *   out[i, :] = (1/K) sum_k w_k * f(table[adj_entity[x_i, k], :] + rowbias[i / nodes_per_group, :])
 *   x_i = node_ids ? node_ids[i] : i;  w = softmax_k(rel_score[adj_relation[x_i, k]]) or 1 (rel_score NULL);
 *   f = relu when relu != 0, identity otherwise.  table: [n_entity, D] fp32 or bf16; out [nodes, D] fp32.
 * Used twice by mvin_amd/model.py: over ALL entities to build S[e] (the user-independent neighbor mix
 * of aggregator (0,.)), and per level-(L-2) node over the hoisted table R1 with the pair's constant as
 * rowbias.  Returns -3 for K > 256. */
MVIN_API int mvin_gather_mix_fwd(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                        const int32_t* node_ids, const float* rel_score, const float* rowbias, int64_t nodes,
                        int nodes_per_group, int K, int D, int n_entity, int nR, int relu, float* out,
                        int table_bf16, void* stream);

/* ---- training (model.py:378-417): forward variants that keep what the backward needs, and the
 * backward / optimizer kernels.  Gradients are ACCUMULATED into caller-zeroed buffers. ------------ */

/* mvin_gather_attn_fwd / mvin_agg_fwd with two extra optional outputs:
 * s_out [T,D] = (1/K) sum_k p_k child_k (before the projection), z_out [T,D] = self + neighbors_agg. */
MVIN_API int mvin_gather_attn_fwd_ex(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                            const int32_t* node_ids, const float* rel_score, const float* self_vec,
                            const float* Wc, const float* c_child, const float* Wagg, const float* bagg,
                            int B, int N, int K, int D, int n_entity, float* out, float* probs,
                            float* s_out, float* z_out, int table_bf16, void* stream);
MVIN_API int mvin_agg_fwd_ex(const float* self_vec, const float* neigh, const int32_t* rel_ids, const float* rel_score,
                    const float* Wagg, const float* bagg, int B, int N, int K, int D, float* out, float* probs,
                    float* s_out, float* z_out, void* stream);

/* element-wise helpers: mode 0 y = alpha x + beta y | 1 sigmoid cross entropy (model.py:379): y = (sigmoid(x) -
 * z) alpha, *accum += beta * ce(x, z) | 2 relu backward y = z > 0 ? x : 0 | 3 *accum += alpha sum x^2 |
 * 4 Adam step (x param, y grad, z m, w v, alpha = lr_t) | 5 y[r,:] = beta y[r,:] + alpha z[r] x[r,:] (n = rows*D) |
 * 6 y[g,:] = alpha sum_{q<N} x[g*N+q,:] (n = groups*D) | 7 *accum += alpha sum_r z[r] sum x[r,:]^2 (n = rows*D) |
 * 8 *accum += alpha sum_r sum x[ids[r],:]^2 with ids = (const int32_t*) z (n = rows*D): gathered rows, not materialised. */
MVIN_API int mvin_eltwise(int mode, int64_t n, float* x, float* y, float* z, float* w, float* accum, float alpha,
                 float beta, float beta1, float beta2, float eps, int D, int N, void* stream);
/* out[b] += |{i : ids[i] == b}| for b < nbins <= 4096 (the occurrence counts of the relations in a ripple-set list: the
 * weights of the sum(r_emb^2) regulariser's gradient, model.py:383-386); out is accumulated, caller zeroes it. */
MVIN_API int mvin_count_ids(const int32_t* ids, int64_t n, int nbins, float* out, void* stream);

/* All parameters in one launch: the L2 terms of model.py:387-412 and (apply_adam != 0) the
 * tf.train.AdamOptimizer update of model.py:414.  Gradients and Adam moments are flat buffers of
 * `total` floats; segs_device[s] = {x: first element of the parameter (slice), off: its offset in the
 * flat buffers (ascending, segs[0].off == 0, contiguous), n: elements, l2: coefficient c of the
 * term (c/2) sum(x^2)}:  g += c x ; *loss_accum += (c/2) sum x^2 ; then m, v, x <- Adam(g, lr_t). */
typedef struct {
    float* x;
    int64_t off;
    int64_t n;
    float l2;
    int pad_;
} mvin_param_seg;
MVIN_API int mvin_l2_adam_multi(const mvin_param_seg* segs_device, int nseg, int64_t total, float* g_flat, float* m_flat,
                       float* v_flat, float* loss_accum, int apply_adam, float lr_t, float beta1, float beta2,
                       float eps, void* stream);
/* Same, with the bias-corrected step size lr_t read from device memory when the kernel runs: a training step
 * captured into a hipGraph (mvin_amd/training.py:GraphedTrainer) replays with a new lr_t without re-capture. */
MVIN_API int mvin_l2_adam_multi_dev(const mvin_param_seg* segs_device, int nseg, int64_t total, float* g_flat, float* m_flat,
                           float* v_flat, float* loss_accum, int apply_adam, const float* lr_t_device, float beta1,
                           float beta2, float eps, void* stream);

/* dtable[ids[r], :] += alpha * x[r, :]  -- backward of tf.nn.embedding_lookup (ids int32 or int64). */
MVIN_API int mvin_scatter_add_rows(float* dtable, const void* ids, int ids64, const float* x, int64_t rows, int D,
                          float alpha, void* stream);

/* dW[z] += X^T . (dY[z] masked by mask > 0), db[z] += column sums; X staged as in mvin_linear_fwd
 * (args->src/ids/nsrc/Dsrc/Dout/rows/nz/sum_sources/ids64 are read, the rest ignored). */
MVIN_API int mvin_linear_wgrad(const mvin_linear_args* args, const float* dY, int64_t ldy, int64_t dy_zstride,
                      const float* mask, int64_t ldm, int64_t mask_zstride, float* dW, int64_t dw_zstride,
                      float* db, int64_t db_zstride, void* stream);
/* n (<= 64) such problems in as few launches as their shapes allow: problems whose (Din, Dout) take the same matrix-core
 * tile kernel share a launch (8 per launch).  At the reference's batch sizes a weight gradient is microseconds of work
 * behind ~10 us of launch and ramp, and a training step has eleven of them.  The results are those of n
 * mvin_linear_wgrad calls (accumulation into dW / db by float atomics in both forms). */
typedef struct {
    mvin_linear_args lin;          /* as for mvin_linear_wgrad */
    const float* dY;
    int64_t ldy, dy_zstride;
    const float* mask;             /* or NULL */
    int64_t ldm, mask_zstride;
    float* dW;
    int64_t dw_zstride;
    float* db;                     /* or NULL */
    int64_t db_zstride;
} mvin_wgrad_problem;
MVIN_API int mvin_linear_wgrad_multi(const mvin_wgrad_problem* problems, int n, void* stream);

/* backward of the neighbor mix agg[t] = (1/K) sum_k p[t,k] c[t,k], p = softmax_k(t[rel]) (aggregators.py:118-152)
 * given dvec = dL/d agg: children from the table through the adjacency (table/adj/node_ids given: dc_k is added
 * atomically to dtable) or dense (child/rel_ids given: dchild written).  dT [nR] accumulates the logit gradients.
 * By-entity form (gather, node_ids == NULL, rel_score [nR] given instead of probs): task t is entity t and dvec
 * [T = n_entity, D] holds dL/d agg summed over every tree node carrying that entity -- the backward is linear in
 * dvec and depends on a node only through its entity, so the duplicates of a batch cost one pass; all-zero rows
 * are skipped. */
MVIN_API int mvin_agg_bwd(const float* table, const int32_t* adj_entity, const int32_t* adj_relation, const int32_t* node_ids,
                 const float* child, const int32_t* rel_ids, const float* probs, const float* rel_score,
                 const float* dvec, int64_t T, int K, int D, int nR, float* dtable, float* dchild, float* dT,
                 void* stream);

/* backward of mvin_rel_score: drel[r,:] += dT[r] urh_w[D:2D]; durh[D:2D] += sum_r dT[r] rel[r,:]. */
MVIN_API int mvin_rel_score_bwd(const float* relation_emb, const float* urh_weights, const float* dT, int nR, int D,
                       float* drel, float* durh, void* stream);

/* backward of mvin_key_addressing_fwd (+ the 2*l2*(h,t) regulariser rows of model.py:383-385): dout is the
 * gradient of out; dE [nE,D], dV [B,nR,D], dw [D] are accumulated. */
MVIN_API int mvin_key_addressing_bwd(const float* entity_emb, const float* V, const float* w,
                            const int32_t* const* mem_h, const int32_t* const* mem_r, const int32_t* const* mem_t,
                            int P, int B, int Nm, int D, int nR, const float* dout, int64_t ldo, float l2,
                            float* dE, float* dV, float* dw, void* stream);
/* Same, and the VALUE of that regulariser, which the kernel has in registers anyway:
 * *reg_accum += l2 * sum over hops, pairs and memories of (|E[h]|^2 + |E[t]|^2)   (model.py:383-385 summed over
 * model.py:387's hops); reg_accum may be NULL.  dw is [dw_replicas, D] (a power of two; 1 = the plain form): every
 * pair adds its share to one replica and the caller sums them -- B atomic updates of the same D floats serialise.
 * relation_kge [nR, D, D] + items [B] (int64 when items64; both or neither): the kernel also adds the item's share of
 * V = E[item] . R_KGE[r], dE[item_b, :] += sum_r dV[b, r, :] . R_KGE[r]^T, from the dV block it holds in LDS; allowed only
 * where mvin_key_addressing_bwd_adds_item_grad(P, Nm, D, nR) != 0 (else -3: do that product with mvin_linear_fwd). */
MVIN_API int mvin_key_addressing_bwd_reg(const float* entity_emb, const float* V, const float* w,
                                const int32_t* const* mem_h, const int32_t* const* mem_r,
                                const int32_t* const* mem_t, int P, int B, int Nm, int D, int nR, const float* dout,
                                int64_t ldo, float l2, float* dE, float* dV, float* dw, int dw_replicas,
                                float* reg_accum, const float* relation_kge, const void* items, int items64,
                                void* stream);
MVIN_API int mvin_key_addressing_bwd_adds_item_grad(int P, int Nm, int D, int nR);

/* ---- inputs of the path, built on the GPU (data_loader_user_set.py) ------------------------
 * Both take the undirected KG as CSR: indptr [nE+1] int64, dst/rel [nnz] int32, every triple
 * listed under its head and under its tail in file order (construct_kg, :324-343).  Draws are a
 * pure function of `seed` (the reference uses unseeded global generators: its RULES are
 * reproduced, its draws cannot be).
 *
 * mvin_sample_adjacency: contruct_random_adj (:375-388).  adj_entity/adj_relation [nE, K] int32:
 * K distinct edges when deg >= K, K draws with replacement when 0 < deg < K, zero row when deg == 0. */
MVIN_API int mvin_sample_adjacency(const int64_t* indptr, const int32_t* dst, const int32_t* rel, int n_entity, int K,
                          uint64_t seed, int32_t* adj_entity, int32_t* adj_relation, void* stream);

/* mvin_encode_adjacency: the duplicate-slot encoding of a sampled adjacency (K <= 128, relation ids < 65536), built
 * once per adjacency.  Row x of enc_entity / enc_relation [nE, K] holds the DISTINCT (neighbour, relation) slots of row x
 * first -- ordered by the distinct-slot count of the neighbour's own row, descending, ties in first-occurrence order --
 * and padding (a copy of slot 0) behind them:
 *     enc_entity   = neighbour | cnt[neighbour] << 24                 (n_entity <= 2^24: the length of the neighbour's
 *                                                                       own list, known one fetch early; unsigned word)
 *     enc_relation = relation | multiplicity << 16 | cnt[x] << 24     (multiplicity 0: padding; unsigned word)
 * cnt [nE] = distinct slots per row (also written).  adj_relation may be NULL (relations read as 0). */
MVIN_API int mvin_encode_adjacency(const int32_t* adj_entity, const int32_t* adj_relation, int n_entity, int K, int32_t* cnt,
                          int32_t* enc_entity, int32_t* enc_relation, void* stream);

/* mvin_build_ripple_sets: get_user_triplet_set / _get_user_triplet_set (:392-441).
 * hist_ptr [nU+1] int64 / hist_items int32: each user's positive train items in interaction order.
 * out [nU, P, 3, Nm] int32 = (heads, relations, tails) per hop; n_neighbor (16 in the reference, <= 32)
 * edges per seed entity enter the candidate list. */
MVIN_API int mvin_build_ripple_sets(const int64_t* indptr, const int32_t* dst, const int32_t* rel,
                           const int64_t* hist_ptr, const int32_t* hist_items, int n_user, int P, int Nm,
                           int n_neighbor, uint64_t seed, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVIN_HIP_H */
