// extern "C" boundary of libmvin_hip.so (see include/mvin_hip.h).  Argument validation,
// per-thread error string, kernel launches.  No allocation, no synchronisation, no state.
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "mvin_kernels.h"
#include "mvin_fused_agg.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int hip_result(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

bool bad_dim(int D) { return D < 4 || D > MVIN_MAX_DIM || (D & 3) != 0; }

size_t geom_sum(int B, int K, int from, int to) {  // B * sum_{e=from..to} K^e
    size_t s = 0, p = 1;
    for (int e = 0; e <= to; ++e) {
        if (e >= from) s += p;
        p *= (size_t)K;
    }
    return s * (size_t)B;
}

}  // namespace

extern "C" {

int mvin_abi_version(void) { return MVIN_ABI_VERSION; }

int mvin_debug_read_trace(long long* host_dst, size_t n) {
    if (!host_dst) return -1;
    static const bool ka = getenv("MVIN_KA_TRACE") != nullptr;     // which kernel's stamps
    static const bool pk = getenv("MVIN_PACK_TRACE") != nullptr;
    if (getenv("MVIN_SMALL_TRACE")) return (int)mvin::small_read_trace(host_dst, n);
    if (getenv("MVIN_KAF_TRACE")) return (int)mvin::kaf_read_trace(host_dst, n);
    if (pk) return (int)mvin::pack_read_prof(host_dst, n);
    if (ka && getenv("MVIN_KA_TRACE")[0] == '2') return (int)mvin::kas_read_trace(host_dst, n);     // the kernel over static records
    return (int)(ka ? mvin::ka_read_trace(host_dst, n) : mvin::split_read_trace(host_dst, n));
}

const char* mvin_last_error(void) { return g_last_error.c_str(); }

size_t mvin_ent_elems(int B, int K, int levels) { return geom_sum(B, K, 0, levels); }

size_t mvin_rel_elems(int B, int K, int levels) { return levels > 0 ? geom_sum(B, K, 1, levels) : 0; }

int mvin_expand_ids(const int32_t* adj_entity, const int32_t* adj_relation, const int64_t* items_i64,
                    const int32_t* items_i32, int B, int K, int levels, int n_entity,
                    int32_t* ent_out, int32_t* rel_out, void* stream) {
    if (!ent_out || (!items_i64 && !items_i32) || (items_i64 && items_i32))
        return fail(-1, "mvin_expand_ids: need ent_out and exactly one of items_i64/items_i32");
    if (levels > 0 && (!adj_entity || !adj_relation || !rel_out))
        return fail(-1, "mvin_expand_ids: null adjacency/rel_out with levels=%d", levels);
    if (B <= 0 || K <= 0 || levels < 0 || n_entity <= 0)
        return fail(-2, "mvin_expand_ids: bad sizes B=%d K=%d levels=%d n_entity=%d", B, K, levels, n_entity);
    return hip_result(mvin::launch_expand(adj_entity, adj_relation, items_i64, items_i32, B, K, levels,
                                          n_entity, ent_out, rel_out, (hipStream_t)stream),
                      "mvin_expand_ids");
}

int mvin_rel_score(const float* relation_emb, const float* urh_weights, int nR, int D, float* t_out,
                   void* stream) {
    if (!relation_emb || !urh_weights || !t_out) return fail(-1, "mvin_rel_score: null pointer");
    if (nR <= 0 || D <= 0) return fail(-2, "mvin_rel_score: bad sizes nR=%d D=%d", nR, D);
    return hip_result(mvin::launch_rel_score(relation_emb, urh_weights, nR, D, t_out, (hipStream_t)stream),
                      "mvin_rel_score");
}

int mvin_linear_fwd(const mvin_linear_args* a, void* stream) {
    if (!a) return fail(-1, "mvin_linear_fwd: null args");
    if (a->nsrc < 1 || a->nsrc > MVIN_MAX_SRC) return fail(-2, "mvin_linear_fwd: nsrc=%d", a->nsrc);
    if (a->Dsrc < 4 || (a->Dsrc & 3) || a->Dout < 1 || a->Dout > MVIN_MAX_DIM)
        return fail(-2, "mvin_linear_fwd: Dsrc=%d (need %%4==0) Dout=%d (need 1..%d)", a->Dsrc, a->Dout,
                    MVIN_MAX_DIM);
    if ((size_t)a->nsrc * a->Dsrc > 4096) return fail(-2, "mvin_linear_fwd: nsrc*Dsrc > 4096");
    if (a->rows < 0) return fail(-2, "mvin_linear_fwd: rows < 0");
    if (!a->out) return fail(-1, "mvin_linear_fwd: null out");
    for (int s = 0; s < a->nsrc; ++s)
        if (!a->src[s]) return fail(-1, "mvin_linear_fwd: null src[%d]", s);
    if (!a->W && ((a->nsrc != 1 && !a->sum_sources) || a->Dsrc != a->Dout))
        return fail(-2, "mvin_linear_fwd: identity (W=NULL) needs nsrc==1 and Dsrc==Dout");
    if (a->ldo < a->Dout) return fail(-2, "mvin_linear_fwd: ldo < Dout");
    if (a->rowbias && a->rows_per_group < 1) return fail(-2, "mvin_linear_fwd: rows_per_group < 1");
    if (a->score_u && a->Dout > (a->sum_sources ? 1 : a->nsrc) * a->Dsrc + 4)
        return fail(-2, "mvin_linear_fwd: fused score needs Dout <= Din + 4");
    if (a->rows == 0) return 0;
    return hip_result(mvin::launch_linear(*a, (hipStream_t)stream), "mvin_linear_fwd");
}

static int agg_common(mvin::GatherAttnArgs& g, int B, int N, int K, int D, const char* who) {
    if (B <= 0 || N <= 0 || K <= 0) return fail(-2, "%s: bad sizes B=%d N=%d K=%d", who, B, N, K);
    if (bad_dim(D)) return fail(-2, "%s: D=%d (need %%4==0, 4..%d)", who, D, MVIN_MAX_DIM);
    if (K > 4096) return fail(-2, "%s: K=%d > 4096", who, K);
    if (!g.self_vec || !g.Wagg || !g.out) return fail(-1, "%s: null self_vec/Wagg/out", who);
    g.T = (int64_t)B * N;
    g.N = N;
    g.K = K;
    g.D = D;
    g.lpr_log2 = mvin::lpr_log2_for(D);
    return 0;
}

int mvin_gather_attn_fwd(const float* table, const int32_t* adj_entity, const int32_t* adj_relation,
                         const int32_t* node_ids, const float* rel_score, const float* self_vec,
                         const float* Wc, const float* c_child, const float* Wagg, const float* bagg, int B,
                         int N, int K, int D, int n_entity, float* out, float* probs, void* stream) {
    return mvin_gather_attn_fwd_ex(table, adj_entity, adj_relation, node_ids, rel_score, self_vec, Wc, c_child,
                                   Wagg, bagg, B, N, K, D, n_entity, out, probs, nullptr, nullptr, 0, stream);
}

int mvin_gather_attn_fwd_ex(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                            const int32_t* node_ids, const float* rel_score, const float* self_vec,
                            const float* Wc, const float* c_child, const float* Wagg, const float* bagg, int B,
                            int N, int K, int D, int n_entity, float* out, float* probs, float* s_out,
                            float* z_out, int table_bf16, void* stream) {
    mvin::GatherAttnArgs g{};
    g.table_bf16 = table_bf16 ? 1 : 0;
    g.s_out = s_out;
    g.z_out = z_out;
    g.gather = 1;
    g.table = table;
    g.adj_e = adj_entity;
    g.adj_r = adj_relation;
    g.node_ids = node_ids;
    g.rel_score = rel_score;
    g.self_vec = self_vec;
    g.Wc = Wc;
    g.c_child = c_child;
    g.Wagg = Wagg;
    g.bagg = bagg;
    g.out = out;
    g.probs = probs;
    if (!table || !adj_entity || !node_ids || (rel_score && !adj_relation))
        return fail(-1, "mvin_gather_attn_fwd: null table/adjacency/node_ids");
    if (n_entity <= 0) return fail(-2, "mvin_gather_attn_fwd: n_entity=%d", n_entity);
    if (probs && !rel_score) return fail(-2, "mvin_gather_attn_fwd: probs requested without rel_score");
    if (int rc = agg_common(g, B, N, K, D, "mvin_gather_attn_fwd")) return rc;
    return hip_result(mvin::launch_gather_attn(g, (hipStream_t)stream), "mvin_gather_attn_fwd_ex");
}

int mvin_gather_attn_l2_supported(int D, int K) { return mvin::fused_l2_supported(D, K) ? 1 : 0; }

int mvin_probe_gather_l2(const void* table, const int32_t* child_ids, const int32_t* grandchild_ids, int64_t n_parents, int K,
                         int D, int n_entity, int table_bf16, float* sums, void* stream) {
    const char* who = "mvin_probe_gather_l2";
    if (!table || !child_ids || !grandchild_ids || !sums) return fail(-1, "%s: null pointer", who);
    if (n_parents <= 0 || K <= 0 || n_entity <= 0) return fail(-2, "%s: bad sizes", who);
    const int rb = D * (table_bf16 ? 2 : 4);
    if (!(rb == 64 || rb == 128 || rb == 256 || (rb == 512 && !table_bf16)))
        return fail(-3, "%s: row bytes %d (64, 128, 256 or 512)", who, rb);
    return hip_result(mvin::launch_gather_probe_l2(table, child_ids, grandchild_ids, n_parents, K, D, table_bf16, sums,
                                                   (hipStream_t)stream), who);
}

int mvin_gather_attn_l2_variant(int D, int K, int64_t n_parents, int n_entity, int want_probs) {
    return mvin_gather_attn_l2_variant_ex(D, K, n_parents, n_entity, want_probs, 0);
}

int mvin_gather_attn_l2_variant_ex(int D, int K, int64_t n_parents, int n_entity, int want_probs, int table_bf16) {
    if (!mvin::fused_l2_supported(D, K)) return 0;
    mvin::FusedL2Args f{};
    f.K = K;
    f.P = n_parents;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    float probe = 0.f;
    if (want_probs) f.probs_parent = f.probs_child = &probe;     // only tested for presence
    f.table_bytes = (uint64_t)n_entity * (uint64_t)D * (table_bf16 ? 2 : 4);
    if (mvin::fused_d32_applies(f, D)) return 4;
    if (mvin::fused_l2_split_in_use() && mvin::fused_split_applies(f, D)) return 2;
    return mvin::fused_d16_applies(f, D) ? 3 : 1;
}

// pid_stride 2: parent_ids points at an int64 [P] array whose low words are read (the item ids of the reference's
// placeholder, model.py:50) -- mvin_score_l2_fwd's depth-2 pass then needs no id-conversion launch
static int gather_attn_l2_impl(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                               const int32_t* parent_ids, int pid_stride, const float* t0, const float* t1,
                               const float* W1, const float* W2, const float* b1, const float* b2, const float* q,
                               const float* A0, const float* a0, int B, int parents_per_pair, int K, int D,
                               int n_entity, int nR, float* nagg0, float* nagg1, float* probs_parent,
                               float* probs_child, int table_bf16, void* stream, bool encoded = false, bool prj = false,
                               const int32_t* order = nullptr, const float* agg = nullptr) {
    const char* who = prj ? "mvin_gather_attn_l2_prj_fwd" : encoded ? "mvin_gather_attn_l2_enc_fwd" : "mvin_gather_attn_l2_fwd";
    if (prj && (table_bf16 || !W1 || !W2 || !q))
        return fail(-1, "%s: projected tables are fp32 and go with the queries", who);
    if (encoded && !mvin::fused_packed_supported(D, K))
        return fail(-3, "%s: unsupported shape D=%d K=%d (D in {32,64,128}, K in {16,32,64,128})", who, D, K);
    if (!mvin::fused_l2_supported(D, K))
        return fail(-3, "%s: unsupported shape D=%d K=%d (D in {16,32,64,128}, K power of two in [4,256])",
                    who, D, K);
    if (!table || !adj_entity || !parent_ids || !A0 || !nagg0 || !nagg1) return fail(-1, "%s: null pointer", who);
    if ((t0 || t1) && !adj_relation) return fail(-1, "%s: attention needs adj_relation", who);
    if ((W1 == nullptr) != (W2 == nullptr)) return fail(-1, "%s: W1 and W2 must be given together", who);
    if (W1 && !q) return fail(-1, "%s: projection needs the query vectors q", who);
    if ((probs_parent || probs_child) && !t0) return fail(-2, "%s: probs requested without t0", who);
    if (B <= 0 || parents_per_pair <= 0 || n_entity <= 0 || nR <= 0 || nR > 4096)
        return fail(-2, "%s: bad sizes B=%d parents_per_pair=%d n_entity=%d nR=%d", who, B, parents_per_pair,
                    n_entity, nR);
    mvin::FusedL2Args f{};
    f.table = table;
    f.adj_e = adj_entity;
    f.adj_r = adj_relation;
    f.parent_ids = parent_ids;
    f.t0 = t0;
    f.t1 = t1;
    f.W1 = W1;
    f.W2 = W2;
    f.b1 = b1;
    f.b2 = b2;
    f.q = q;
    f.A0 = A0;
    f.a0 = a0;
    f.nagg0 = nagg0;
    f.nagg1 = nagg1;
    f.probs_parent = probs_parent;
    f.probs_child = probs_child;
    f.P = (int64_t)B * parents_per_pair;
    f.table_bytes = (uint64_t)n_entity * (uint64_t)D * (table_bf16 ? 2 : 4);
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    f.parents_per_pair = parents_per_pair;
    f.K = K;
    f.nR = nR;
    f.pid_stride = pid_stride;
    f.max_id = (unsigned)(n_entity - 1);
    f.prj = prj ? 1 : 0;
    int l = 0;
    if (agg) {                   // per-entity aggregates in place of the tables (mvin_gather_attn_l2_agg_fwd)
        f.agg = const_cast<float*>(agg);
        f.order = order;
        if (!(prj && encoded && mvin::fused_agg_applies(f, D) && (!order || parents_per_pair == 1)))
            return fail(-3, "mvin_gather_attn_l2_agg_fwd: D = 64, K in {16, 32, 64}, n_entity <= 2^24, tables < 1 GiB, adjacency and outputs < 2 GiB "
                            "(a parent order: one parent per pair)");
        return hip_result(mvin::launch_gather_attn_l2_agg(f, (hipStream_t)stream), "mvin_gather_attn_l2_agg_fwd");
    }
    if (order) {
        f.order = order;
        if (!(prj && encoded && parents_per_pair == 1 && mvin::fused_wpp_applies(f, D)))
            return fail(-3, "%s: a parent order is taken by the wave-per-parent kernel only (projected tables, encoded adjacency, D = 64, K <= 32, "
                            "one parent per pair)", who);
    }
    while ((4 << l) < K) ++l;
    f.lpn_log2 = l;
    {
        static const char* dbg = getenv("MVIN_SPLIT_DBG");
        f.dbg = dbg ? atoi(dbg) : 0;
    }
    if (prj && !encoded) {       // plain adjacency: only the wave-per-parent kernel (D = 32, K <= 16) has a projected-tables form
        if (!mvin::fused_d32_applies(f, D)) return fail(-3, "%s: the projected-tables form over a plain adjacency exists for D = 32, K in {8, 16}", who);
        return hip_result(mvin::launch_gather_attn_l2_d32(f, 0, (hipStream_t)stream, false), who);
    }
    if (encoded) {
        if (!adj_relation) return fail(-1, "%s: null enc_relation", who);
        if (!mvin::fused_packed_applies(f, D))
            return fail(-3, "%s: tables too large (n_entity <= 2^24, table < 4 GiB, adjacency and outputs < 2 GiB)", who);
        // D = 32, K <= 16 (BASELINE C2): the wave-per-parent kernel reads the encoding too (MVIN_L2_D32ENC=0: the packed-tile kernel, A/B)
        static const bool d32enc_off = getenv("MVIN_L2_D32ENC") && atoi(getenv("MVIN_L2_D32ENC")) == 0;
        if (!d32enc_off && mvin::fused_d32_applies(f, D))
            return hip_result(mvin::launch_gather_attn_l2_d32(f, table_bf16, (hipStream_t)stream, true), who);
        // dim 64 over projected tables: the wave-per-parent kernel (MVIN_L2_WPP=0: the packed-tile kernel, A/B)
        if (prj && mvin::fused_wpp_applies(f, D)) return hip_result(mvin::launch_gather_attn_l2_wpp(f, (hipStream_t)stream), who);
        return hip_result(mvin::launch_gather_attn_l2_packed(f, D, table_bf16, (hipStream_t)stream), who);
    }
    return hip_result(mvin::launch_gather_attn_l2(f, D, table_bf16, (hipStream_t)stream), who);
}

int mvin_gather_attn_l2_enc_supported(int D, int K) { return mvin::fused_packed_supported(D, K) ? 1 : 0; }

int mvin_gather_attn_l2_enc_fwd(const void* table, const int32_t* enc_entity, const int32_t* enc_relation,
                                const void* parent_ids, int parent_ids_i64, const float* t0, const float* t1,
                                const float* W1, const float* W2, const float* b1, const float* b2, const float* q,
                                const float* A0, const float* a0, int B, int parents_per_pair, int K, int D, int n_entity,
                                int nR, float* nagg0, float* nagg1, int table_bf16, void* stream) {
    return gather_attn_l2_impl(table, enc_entity, enc_relation, reinterpret_cast<const int32_t*>(parent_ids),
                               parent_ids_i64 ? 2 : 1, t0, t1, W1, W2, b1, b2, q, A0, a0, B, parents_per_pair, K, D, n_entity,
                               nR, nagg0, nagg1, nullptr, nullptr, table_bf16, stream, true);
}

// out[z] = src . W_z (+ b_z), z = 0, 1: the two projections of the levels the fused kernel gathers, of table rows or of queries
static int prj_linear(const float* src, int64_t rows, int D, const float* W1, const float* W2, const float* b1, const float* b2,
                      float* out, void* stream) {
    mvin_linear_args l{};
    l.src[0] = src;
    l.nsrc = 1;
    l.Dsrc = D;
    l.Dout = D;
    l.rows = rows;
    l.rows_per_group = 1;
    l.W = W1;
    l.w_zstride = W2 - W1;                    // (elements; the two matrices need not be adjacent)
    l.bias = b1;
    l.bias_zstride = b1 ? b2 - b1 : 0;
    l.out = out;
    l.ldo = D;
    l.nz = 2;
    l.out_zstride = rows * D;
    return mvin_linear_fwd(&l, stream);
}

int mvin_project_rows(const float* src, int64_t rows, int D, const float* W1, const float* W2, const float* b1, const float* b2,
                      float* out, void* stream) {
    const char* who = "mvin_project_rows";
    if (!src || !W1 || !W2 || !out) return fail(-1, "%s: null pointer", who);
    if ((b1 == nullptr) != (b2 == nullptr)) return fail(-1, "%s: b1 and b2 go together", who);
    if (rows <= 0 || D <= 0) return fail(-2, "%s: rows=%lld D=%d", who, (long long)rows, D);
    return prj_linear(src, rows, D, W1, W2, b1, b2, out, stream);
}

size_t mvin_project_tables_elems(int n_entity, int D) {
    if (n_entity <= 0 || D <= 0) return 0;
    return (size_t)3 * n_entity * D + (size_t)4 * D * D + (size_t)2 * D;
}

int mvin_project_tables(const float* entity_emb, const float* W1, const float* W2, const float* b1, const float* b2, const float* A0,
                        const float* a0, int attention, int K, int n_entity, int D, float* ws, void* stream) {
    const char* who = "mvin_project_tables";
    if (!entity_emb || !W1 || !W2 || !A0 || !ws) return fail(-1, "%s: null pointer", who);
    if ((b1 == nullptr) != (b2 == nullptr)) return fail(-1, "%s: b1 and b2 go together", who);
    if (n_entity <= 0 || K <= 0 || (D != 32 && D != 64 && D != 128)) return fail(-2, "%s: n_entity=%d K=%d D=%d", who, n_entity, K, D);
    float* blk = ws + (size_t)3 * n_entity * D;
    // c = (sum of the K attention weights) / K: softmax weights sum to 1, plain-mean weights to K  (aggregators.py:139-152)
    const float c = attention ? 1.f / (float)K : 1.f;
    if (int rc = hip_result(mvin::launch_prj_prepare(W1, W2, b1, b2, A0, a0, c, D, blk, (hipStream_t)stream), who)) return rc;
    mvin_linear_args l{};
    l.src[0] = entity_emb;
    l.nsrc = 1;
    l.Dsrc = D;
    l.Dout = D;
    l.rows = n_entity;
    l.rows_per_group = 1;
    l.W = blk;                                // W1 | W1.A0 | W2.A0
    l.w_zstride = (int64_t)D * D;
    l.out = ws;
    l.ldo = D;
    l.nz = 3;
    l.out_zstride = (int64_t)n_entity * D;
    return mvin_linear_fwd(&l, stream);
}

// does mvin_gather_attn_l2_prj_fwd run for these tables?  (the plain-adjacency form exists in the wave-per-parent kernel only)
static bool prj_applies(int D, int K, bool encoded, int n_entity, int nR, int64_t n_parents) {
    if (n_entity <= 0 || nR <= 0 || nR > 4096 || !mvin::fused_l2_supported(D, K)) return false;
    mvin::FusedL2Args f{};
    f.K = K;
    f.nR = nR;
    f.P = n_parents > 0 ? n_parents : 1;
    f.prj = 1;
    f.parents_per_pair = 1;
    f.max_id = (unsigned)(n_entity - 1);
    static const int32_t present = 0;
    f.adj_r = &present;                       // (only tested for presence)
    f.table_bytes = (uint64_t)n_entity * (uint64_t)D * 4;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    if (f.table_bytes >= (1ull << 30)) return false;
    if (encoded) return mvin::fused_packed_supported(D, K) && mvin::fused_packed_applies(f, D);
    return mvin::fused_d32_applies(f, D);
}

int mvin_gather_attn_l2_prj_supported(int D, int K, int adjacency_encoded, int n_entity, int nR) {
    return prj_applies(D, K, adjacency_encoded != 0, n_entity, nR, 1) ? 1 : 0;
}

int mvin_gather_attn_l2_prj_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, int adjacency_encoded,
                                const void* parent_ids, int parent_ids_i64, const float* t0, const float* t1, const float* q,
                                int B, int parents_per_pair, int K, int D, int n_entity, int nR, float* nagg0, float* nagg1,
                                void* stream) {
    return mvin_gather_attn_l2_prj_ordered_fwd(ws, enc_entity, enc_relation, adjacency_encoded, parent_ids, parent_ids_i64, nullptr, t0, t1, q,
                                               B, parents_per_pair, K, D, n_entity, nR, nagg0, nagg1, stream);
}

size_t mvin_order_by_key_ws_elems(int64_t B) { return B > 0 ? mvin::order_ws_elems(B) : 0; }

int mvin_order_by_key(const int64_t* keys_i64, const int32_t* keys_i32, int64_t B, int32_t* workspace, int32_t* order, void* stream) {
    const char* who = "mvin_order_by_key";
    if ((keys_i64 == nullptr) == (keys_i32 == nullptr)) return fail(-1, "%s: exactly one of keys_i64 / keys_i32", who);
    if (!workspace || !order) return fail(-1, "%s: null pointer", who);
    if (B <= 0 || B >= (int64_t(1) << 31)) return fail(-2, "%s: B=%lld", who, (long long)B);
    return hip_result(mvin::launch_order_by_key(keys_i64, keys_i32, B, workspace, order, (hipStream_t)stream), who);
}

int mvin_gather_attn_l2_prj_ordered_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, int adjacency_encoded,
                                        const void* parent_ids, int parent_ids_i64, const int32_t* order, const float* t0, const float* t1,
                                        const float* q, int B, int parents_per_pair, int K, int D, int n_entity, int nR, float* nagg0,
                                        float* nagg1, void* stream) {
    if (!ws || n_entity <= 0 || D <= 0) return fail(-1, "mvin_gather_attn_l2_prj_fwd: null workspace / bad sizes");
    const float* blk = ws + (size_t)3 * n_entity * D;
    const float* Wv = blk + (size_t)3 * D * D;
    const float* b1c = Wv + (size_t)D * D;
    // (W1, b1) and (the combined matrix, its bias) project the parents' queries; A0 / a0 are inside the tables and the bias
    return gather_attn_l2_impl(ws, enc_entity, enc_relation, reinterpret_cast<const int32_t*>(parent_ids),
                               parent_ids_i64 ? 2 : 1, t0, t1, blk, Wv, b1c, b1c + D, q, blk, nullptr, B, parents_per_pair, K,
                               D, n_entity, nR, nagg0, nagg1, nullptr, nullptr, 0, stream, adjacency_encoded != 0, true, order);
}

// does the per-entity aggregates form run for these tables?
static bool agg_applies(int D, int K, int n_entity, int nR, int64_t n_parents) {
    if (n_entity <= 0 || nR <= 0 || nR > 4096 || !mvin::fused_agg_supported(D, K)) return false;
    static const char* e = getenv("MVIN_L2_AGG");
    if (e && e[0] == '0') return false;                   // A/B: the kernels over the projected tables themselves
    mvin::FusedL2Args f{};
    f.K = K;
    f.nR = nR;
    f.P = n_parents > 0 ? n_parents : 1;
    f.prj = 1;
    f.parents_per_pair = 1;
    f.max_id = (unsigned)(n_entity - 1);
    static const int32_t present = 0;
    f.adj_r = &present;                       // (only tested for presence)
    f.table_bytes = (uint64_t)n_entity * (uint64_t)D * 4;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    return mvin::fused_agg_applies(f, D);
}

int mvin_gather_attn_l2_agg_supported(int D, int K, int n_entity, int nR) { return agg_applies(D, K, n_entity, nR, 1) ? 1 : 0; }

size_t mvin_entity_aggregates_elems(int n_entity, int D) { return n_entity > 0 && D > 0 ? (size_t)2 * n_entity * D : 0; }

int mvin_entity_aggregates(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, const float* t0, int K, int D,
                           int n_entity, int nR, float* agg, void* stream) {
    const char* who = "mvin_entity_aggregates";
    if (!ws || !enc_entity || !enc_relation || !agg) return fail(-1, "%s: null pointer", who);
    if (!agg_applies(D, K, n_entity, nR, 1))
        return fail(-3, "%s: D = 64, K in {16, 32, 64}, n_entity <= 2^24, tables < 1 GiB, adjacency < 2 GiB, nR <= 4096 (D=%d K=%d n_entity=%d nR=%d)",
                    who, D, K, n_entity, nR);
    mvin::EntityAggArgs f{};
    const size_t tab = (size_t)n_entity * D;
    f.tabS = ws;                              // T1
    f.selfG = ws + tab;                       // TA1
    f.tabG = ws + 2 * tab;                    // TA2
    f.adj_e = enc_entity;
    f.adj_r = enc_relation;
    f.t0 = t0;
    f.outS = agg;
    f.outG = agg + tab;
    f.n_entity = n_entity;
    f.K = K;
    f.nR = nR;
    f.table_bytes = (uint64_t)n_entity * (uint64_t)D * 4;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    return hip_result(mvin::launch_entity_aggregates(f, (hipStream_t)stream), who);
}

int mvin_gather_attn_l2_agg_fwd(const float* ws, const float* agg, const int32_t* enc_entity, const int32_t* enc_relation,
                                const void* parent_ids, int parent_ids_i64, const int32_t* order, const float* t0, const float* t1,
                                const float* q, int B, int parents_per_pair, int K, int D, int n_entity, int nR, float* nagg0,
                                float* nagg1, void* stream) {
    if (!ws || !agg || n_entity <= 0 || D <= 0) return fail(-1, "mvin_gather_attn_l2_agg_fwd: null workspace / bad sizes");
    const float* blk = ws + (size_t)3 * n_entity * D;
    const float* Wv = blk + (size_t)3 * D * D;
    const float* b1c = Wv + (size_t)D * D;
    return gather_attn_l2_impl(ws, enc_entity, enc_relation, reinterpret_cast<const int32_t*>(parent_ids), parent_ids_i64 ? 2 : 1, t0, t1,
                               blk, Wv, b1c, b1c + D, q, blk, nullptr, B, parents_per_pair, K, D, n_entity, nR, nagg0, nagg1, nullptr,
                               nullptr, 0, stream, true, true, order, agg);
}

static bool fold_gather_applies(int D, int K, int n_entity, int nR, int64_t B);

// does the folded-tail form run for these tables?  (dim 64 like the aggregates form, and dim 32)
static bool fold_applies(int D, int K, int n_entity, int nR, int64_t B) {
    if (D == 64) return agg_applies(D, K, n_entity, nR, B);
    static const char* e = getenv("MVIN_L2_AGG");
    if (e && e[0] == '0') return false;
    if (n_entity <= 0 || nR <= 0 || nR > 4096 || !mvin::fused_fold_supported(D, K)) return false;
    const uint64_t tb = (uint64_t)n_entity * (uint64_t)D * 4, ab = (uint64_t)n_entity * (uint64_t)K * 4;
    return tb < (1ull << 30) && ab < (1ull << 31) && n_entity <= (1 << 24) && (B <= 0 || (uint64_t)B * D * 4 < (1ull << 31)) &&
           mvin::fused_fold_lds_bytes(D, nR, K) <= 48 * 1024;
}

// ---- folded-tail form: tables TA1 | TA2 | T0A | M0, aggregates H0 | G, parameter block ----
static size_t fold_blk_elems(int D) { return (size_t)12 * D * D + (size_t)3 * D; }      // Wstack[4] | Wv | Wq | bv | bq | bm | Wperm[6]

size_t mvin_fold_tables_elems(int n_entity, int D) {
    return n_entity > 0 && D > 0 ? (size_t)6 * n_entity * D + fold_blk_elems(D) : 0;
}

int mvin_score_l2_folded_supported(int D, int K, int n_entity, int nR) { return fold_applies(D, K, n_entity, nR, 1) ? 1 : 0; }

int mvin_fold_tables(const float* entity_emb, const int32_t* enc_entity, const int32_t* enc_relation, const float* t0, const float* W0,
                     const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* A0, const float* a0,
                     const float* Wmix, const float* bmix, const float* A1, int K, int D, int n_entity, int nR, float* ws, void* stream) {
    return mvin_fold_tables_ex(entity_emb, enc_entity, enc_relation, t0, W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, 1, K, D, n_entity, nR, ws,
                               stream);
}

int mvin_fold_tables_ex(const float* entity_emb, const int32_t* enc_entity, const int32_t* enc_relation, const float* t0, const float* W0,
                        const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* A0, const float* a0,
                        const float* Wmix, const float* bmix, const float* A1, int aggregates, int K, int D, int n_entity, int nR, float* ws,
                        void* stream) {
    const char* who = "mvin_fold_tables";
    if (!entity_emb || !enc_entity || !enc_relation || !W0 || !W1 || !W2 || !A0 || !Wmix || !A1 || !ws) return fail(-1, "%s: null pointer", who);
    if (!(aggregates ? fold_applies(D, K, n_entity, nR, 1) : fold_gather_applies(D, K, n_entity, nR, 1)))
        return fail(-3, "%s: D = 64 with K in {16, 32, 64} or D = 32 with K in {16, 32}; n_entity <= 2^24, tables < 1 GiB, adjacency < 2 GiB, nR <= 4096 "
                        "(D=%d K=%d n_entity=%d nR=%d)", who, D, K, n_entity, nR);
    const size_t tab = (size_t)n_entity * D;
    float* blk = ws + 6 * tab;
    const float c = t0 ? 1.f / (float)K : 1.f;            // sum of a row's slot weights over K  (aggregators.py:139-152)
    if (int rc = hip_result(mvin::launch_fold_prepare(W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, c, D, blk, (hipStream_t)stream), who)) return rc;
    mvin_linear_args l{};
    l.src[0] = entity_emb;
    l.nsrc = 1;
    l.Dsrc = D;
    l.Dout = D;
    l.rows = n_entity;
    l.rows_per_group = 1;
    l.W = blk;                                // W1.A0 | W2.A0 | W0.A0 | W0.Wm0
    l.w_zstride = (int64_t)D * D;
    l.out = ws;                               // TA1 | TA2 | T0A | M0
    l.ldo = D;
    l.nz = 4;
    l.out_zstride = (int64_t)n_entity * D;
    if (int rc = mvin_linear_fwd(&l, stream)) return rc;
    if (!aggregates) return 0;                // (the gather form: per-row tables only, every pair walks its own children)
    mvin::EntityAggArgs f{};
    f.tabS = ws;                              // H0[e] = T0A[e] + sum_k w_k TA1[y_k]
    f.selfS = ws + 2 * tab;
    f.tabG = ws + tab;                        // G[e]  = TA1[e] + sum_k w_k TA2[y_k]
    f.selfG = ws;
    f.adj_e = enc_entity;
    f.adj_r = enc_relation;
    f.t0 = t0;
    f.outS = ws + 4 * tab;
    f.outG = ws + 5 * tab;
    f.n_entity = n_entity;
    f.K = K;
    f.D = D;
    f.nR = nR;
    f.table_bytes = (uint64_t)tab * 4;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    return hip_result(mvin::launch_entity_aggregates(f, (hipStream_t)stream), who);
}

int mvin_score_l2_folded_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, const int64_t* items_i64,
                             const int32_t* items_i32, const float* t0, const float* t1, const float* q, const float* user_o, const float* A1,
                             const float* a1, const float* Wmix, int64_t B, int K, int D, int n_entity, int nR, float* out0, float* z2,
                             float* item_emb, float* scores, float* sig, void* stream) {
    const char* who = "mvin_score_l2_folded_fwd";
    if (!ws || !enc_entity || !enc_relation || !q || !user_o || !A1 || !Wmix || !scores) return fail(-1, "%s: null pointer", who);
    if ((items_i64 == nullptr) == (items_i32 == nullptr)) return fail(-1, "%s: exactly one of items_i64 / items_i32", who);
    if (B <= 0 || B >= (int64_t(1) << 31)) return fail(-2, "%s: B=%lld", who, (long long)B);
    if (!fold_applies(D, K, n_entity, nR, B)) return fail(-3, "%s: unsupported shape / sizes D=%d K=%d n_entity=%d nR=%d B=%lld", who, D, K, n_entity, nR, (long long)B);
    const size_t tab = (size_t)n_entity * D;
    const float* blk = ws + 6 * tab;
    const float* Wv = blk + (size_t)4 * D * D;
    const float* Wq = Wv + (size_t)D * D;
    const float* bv = Wq + (size_t)D * D;
    const float* bq = bv + D;
    const float* bm = bq + D;
    static const bool two = getenv("MVIN_L2_FOLD_TWO") && atoi(getenv("MVIN_L2_FOLD_TWO")) != 0;
    if (!two || D != 64)                      // ONE launch: pair kernel and tail on the same batches of 16 pairs (out0 / z2 stay unused)
    {       // (the regrouped copies Wperm of the six blocks -- of the CURRENT A1 / Wmix too: mvin_fold_tables wrote them)
        const float* Wp = bm + D;
        const size_t DD = (size_t)D * D;
        return hip_result(mvin::launch_score_l2_folded(ws + 4 * tab, ws + 3 * tab, enc_entity, enc_relation,
                                                       items_i64 ? reinterpret_cast<const int32_t*>(items_i64) : items_i32, items_i64 ? 2 : 1, t1, q,
                                                       user_o, Wp, bq, Wp + DD, bv, Wp + 2 * DD, Wp + 3 * DD, a1, Wp + 4 * DD, Wp + 5 * DD, bm,
                                                       item_emb, scores, sig, B, K, D, nR, n_entity, (hipStream_t)stream),
                          who);
    }
    // MVIN_L2_FOLD_TWO=1 (A/B): the pair kernel writes out0 and Z2, the tile kernel of mvin_tail.hip takes them from there
    if (!out0 || !z2) return fail(-1, "%s: the two-launch variant (MVIN_L2_FOLD_TWO=1) needs the out0 / z2 scratch rows", who);
    mvin::FusedL2Args f{};
    f.table = ws;
    f.agg = const_cast<float*>(ws + 4 * tab);     // H0 | G
    f.adj_e = enc_entity;
    f.adj_r = enc_relation;
    f.parent_ids = items_i64 ? reinterpret_cast<const int32_t*>(items_i64) : items_i32;
    f.pid_stride = items_i64 ? 2 : 1;
    f.t0 = t0;
    f.t1 = t1;
    f.W1 = Wq;
    f.b1 = bq;
    f.W2 = Wv;
    f.b2 = bv;
    f.q = q;
    f.nagg0 = out0;
    f.nagg1 = z2;
    f.P = B;
    f.table_bytes = (uint64_t)tab * 4;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    f.parents_per_pair = 1;
    f.K = K;
    f.nR = nR;
    f.max_id = (unsigned)(n_entity - 1);
    f.prj = 1;
    f.fold = 1;
    if (int rc = hip_result(mvin::launch_gather_attn_l2_agg(f, (hipStream_t)stream), who)) return rc;
    mvin::TailFoldArgs t{};
    t.M0 = ws + 3 * tab;
    t.items64 = items_i64;
    t.items32 = items_i32;
    t.q = q;
    t.user_o = user_o;
    t.out0 = out0;
    t.z2 = z2;
    t.Wqm = blk + (size_t)3 * D * D;
    t.A1 = A1;
    t.a1 = a1;
    t.Wmix = Wmix;
    t.bm = bm;
    t.item_emb = item_emb;
    t.scores = scores;
    t.sig = sig;
    t.B = B;
    t.n_entity = n_entity;
    return hip_result(mvin::launch_l2_tail_fold(t, (hipStream_t)stream), who);
}

// the folded tail with every pair gathering its own rows (no per-entity sums): dim 64, K in {16, 32}
static bool fold_gather_applies(int D, int K, int n_entity, int nR, int64_t B) {
    if (D != 64 || (K != 16 && K != 32) || n_entity <= 0 || nR <= 0 || nR > 4096) return false;
    static const char* e = getenv("MVIN_L2_FOLD_GATHER");
    if (e && e[0] == '0') return false;                   // A/B: wave-per-parent kernel + mvin_l2_tail_fwd
    const uint64_t tb = (uint64_t)n_entity * (uint64_t)D * 4, ab = (uint64_t)n_entity * (uint64_t)K * 4;
    return tb < (1ull << 30) && ab < (1ull << 31) && n_entity <= (1 << 24) && (B <= 0 || (uint64_t)B * D * 4 < (1ull << 31)) &&
           mvin::fused_wppfold_lds_bytes(nR, K) <= 48 * 1024;
}

int mvin_score_l2_folded_gather_supported(int D, int K, int n_entity, int nR) { return fold_gather_applies(D, K, n_entity, nR, 1) ? 1 : 0; }

int mvin_score_l2_folded_gather_fwd(const float* ws, const int32_t* enc_entity, const int32_t* enc_relation, const int64_t* items_i64,
                                    const int32_t* items_i32, const int32_t* order, const float* t0, const float* t1, const float* q,
                                    const float* user_o, const float* A1, const float* a1, const float* Wmix, int64_t B, int K, int D,
                                    int n_entity, int nR, float* item_emb, float* scores, float* sig, void* stream) {
    const char* who = "mvin_score_l2_folded_gather_fwd";
    if (!ws || !enc_entity || !enc_relation || !q || !user_o || !A1 || !Wmix || !scores) return fail(-1, "%s: null pointer", who);
    if ((items_i64 == nullptr) == (items_i32 == nullptr)) return fail(-1, "%s: exactly one of items_i64 / items_i32", who);
    if (B <= 0 || B >= (int64_t(1) << 31)) return fail(-2, "%s: B=%lld", who, (long long)B);
    if (!fold_gather_applies(D, K, n_entity, nR, B))
        return fail(-3, "%s: D = 64, K in {16, 32}; n_entity <= 2^24, tables < 1 GiB, adjacency < 2 GiB (D=%d K=%d n_entity=%d nR=%d B=%lld)", who, D, K,
                    n_entity, nR, (long long)B);
    const size_t tab = (size_t)n_entity * D, DD = (size_t)D * D;
    const float* blk = ws + 6 * tab;
    const float* bv = blk + 6 * DD;
    const float* bq = bv + D;
    const float* bm = bq + D;
    const float* Wp = bm + D;                 // the regrouped copies of the six blocks: Wq | Wv | Wqm | A1 | Wm1 | Wm2
    mvin::FoldArgs f{};
    f.tables = ws;                            // TA1 | TA2 | T0A
    f.M0 = ws + 3 * tab;
    f.adj_e = enc_entity;
    f.adj_r = enc_relation;
    f.items = items_i64 ? reinterpret_cast<const int32_t*>(items_i64) : items_i32;
    f.pid_stride = items_i64 ? 2 : 1;
    f.order = order;
    f.t0 = t0;
    f.t1 = t1;
    f.q = q;
    f.user_o = user_o;
    f.Wq = Wp, f.bq = bq, f.Wv = Wp + DD, f.bv = bv, f.Wqm = Wp + 2 * DD, f.A1 = Wp + 3 * DD, f.a1 = a1, f.Wm1 = Wp + 4 * DD, f.Wm2 = Wp + 5 * DD, f.bm = bm;
    f.item_emb = item_emb, f.scores = scores, f.sig = sig;
    f.B = B, f.K = K, f.nR = nR;
    f.max_id = (unsigned)(n_entity - 1);
    f.table_bytes = (uint64_t)tab * 4;
    f.adj_bytes = (uint64_t)n_entity * (uint64_t)K * 4;
    return hip_result(mvin::launch_score_l2_folded_gather(f, (hipStream_t)stream), who);
}

int mvin_encode_adjacency(const int32_t* adj_entity, const int32_t* adj_relation, int n_entity, int K, int32_t* cnt,
                          int32_t* enc_entity, int32_t* enc_relation, void* stream) {
    const char* who = "mvin_encode_adjacency";
    if (!adj_entity || !cnt || !enc_entity || !enc_relation) return fail(-1, "%s: null pointer", who);
    if (n_entity <= 0 || n_entity > (1 << 24) || K <= 0 || K > 128)
        return fail(-2, "%s: n_entity=%d K=%d (n_entity <= 2^24, K <= 128)", who, n_entity, K);
    return hip_result(mvin::launch_encode_adjacency(adj_entity, adj_relation, n_entity, K, cnt, enc_entity, enc_relation,
                                                    (hipStream_t)stream), who);
}

int mvin_gather_attn_l2_fwd(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                            const int32_t* parent_ids, const float* t0, const float* t1, const float* W1,
                            const float* W2, const float* b1, const float* b2, const float* q,
                            const float* A0, const float* a0, int B, int parents_per_pair, int K, int D, int n_entity, int nR,
                            float* nagg0, float* nagg1, float* probs_parent, float* probs_child,
                            int table_bf16, void* stream) {
    return gather_attn_l2_impl(table, adj_entity, adj_relation, parent_ids, 1, t0, t1, W1, W2, b1, b2, q, A0, a0, B,
                               parents_per_pair, K, D, n_entity, nR, nagg0, nagg1, probs_parent, probs_child,
                               table_bf16, stream);
}

int mvin_gather_attn_l2_fwd_i64(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                                const int64_t* parent_ids, const float* t0, const float* t1, const float* W1,
                                const float* W2, const float* b1, const float* b2, const float* q,
                                const float* A0, const float* a0, int B, int parents_per_pair, int K, int D, int n_entity, int nR,
                                float* nagg0, float* nagg1, float* probs_parent, float* probs_child,
                                int table_bf16, void* stream) {
    // little-endian low words, stride 2; ids are clamped to the table inside the kernels like every device-resident id
    return gather_attn_l2_impl(table, adj_entity, adj_relation, reinterpret_cast<const int32_t*>(parent_ids), 2, t0, t1, W1, W2,
                               b1, b2, q, A0, a0, B, parents_per_pair, K, D, n_entity, nR, nagg0, nagg1, probs_parent,
                               probs_child, table_bf16, stream);
}

int mvin_agg_fwd(const float* self_vec, const float* neigh, const int32_t* rel_ids, const float* rel_score,
                 const float* Wagg, const float* bagg, int B, int N, int K, int D, float* out, float* probs,
                 void* stream) {
    return mvin_agg_fwd_ex(self_vec, neigh, rel_ids, rel_score, Wagg, bagg, B, N, K, D, out, probs, nullptr,
                           nullptr, stream);
}

int mvin_agg_fwd_ex(const float* self_vec, const float* neigh, const int32_t* rel_ids, const float* rel_score,
                    const float* Wagg, const float* bagg, int B, int N, int K, int D, float* out, float* probs,
                    float* s_out, float* z_out, void* stream) {
    mvin::GatherAttnArgs g{};
    g.s_out = s_out;
    g.z_out = z_out;
    g.gather = 0;
    g.neigh = neigh;
    g.rel_ids = rel_ids;
    g.rel_score = rel_score;
    g.self_vec = self_vec;
    g.Wagg = Wagg;
    g.bagg = bagg;
    g.out = out;
    g.probs = probs;
    if (!neigh) return fail(-1, "mvin_agg_fwd: null neigh");
    if (probs && !rel_score) return fail(-2, "mvin_agg_fwd: probs requested without rel_score");
    if (int rc = agg_common(g, B, N, K, D, "mvin_agg_fwd")) return rc;
    return hip_result(mvin::launch_gather_attn(g, (hipStream_t)stream), "mvin_agg_fwd");
}

int mvin_ripple_attn_fwd_ex(const void* entity_emb, const int32_t* score_ids, const int32_t* rel_ids,
                            const int32_t* value_ids, const float* V, const float* w, int mode, int B, int Nm,
                            int D, int nR, float* out, int64_t ldo, int table_bf16, void* stream) {
    if (!entity_emb || !score_ids || !value_ids || !out) return fail(-1, "mvin_ripple_attn_fwd: null pointer");
    if (mode != 0 && mode != 1) return fail(-2, "mvin_ripple_attn_fwd: mode=%d", mode);
    if (mode == 0 && (!V || !rel_ids || nR <= 0)) return fail(-1, "mvin_ripple_attn_fwd: mode 0 needs V, rel_ids, nR");
    if (mode == 1 && !w) return fail(-1, "mvin_ripple_attn_fwd: mode 1 needs w");
    if (B <= 0 || Nm <= 0 || Nm > 8192) return fail(-2, "mvin_ripple_attn_fwd: bad sizes B=%d Nm=%d", B, Nm);
    if (bad_dim(D)) return fail(-2, "mvin_ripple_attn_fwd: D=%d (need %%4==0, 4..%d)", D, MVIN_MAX_DIM);
    if (ldo < D || (ldo & 3)) return fail(-2, "mvin_ripple_attn_fwd: ldo=%lld (need >= D, %%4==0)", (long long)ldo);
    mvin::RippleArgs r{};
    r.E = entity_emb;
    r.table_bf16 = table_bf16 ? 1 : 0;
    r.score_ids = score_ids;
    r.rel_ids = rel_ids;
    r.value_ids = value_ids;
    r.V = V;
    r.w = w;
    r.out = out;
    r.ldo = ldo;
    r.B = B;
    r.mode = mode;
    r.Nm = Nm;
    r.D = D;
    r.nR = nR;
    r.lpr_log2 = mvin::lpr_log2_for(D);
    return hip_result(mvin::launch_ripple(r, (hipStream_t)stream), "mvin_ripple_attn_fwd");
}

int mvin_ripple_attn_fwd(const float* entity_emb, const int32_t* score_ids, const int32_t* rel_ids,
                         const int32_t* value_ids, const float* V, const float* w, int mode, int B, int Nm,
                         int D, int nR, float* out, int64_t ldo, void* stream) {
    return mvin_ripple_attn_fwd_ex(entity_emb, score_ids, rel_ids, value_ids, V, w, mode, B, Nm, D, nR, out, ldo, 0, stream);
}

int mvin_key_addressing_supported(int Nm, int D) {
    return (!bad_dim(D) && Nm > 0 && mvin::key_addr_nj(Nm, D) <= 16) ? 1 : 0;
}

// shared body of the two key-addressing entry points: ripple sets as per-pair arrays, or user_triplet_set + user ids
static int key_addressing_impl(const char* who, const void* entity_emb, const float* V, const float* w,
                               const int32_t* const* mem_h, const int32_t* const* mem_r, const int32_t* const* mem_t,
                               const int32_t* uts, const int64_t* users64, const int32_t* users32, int P, int B, int Nm,
                               int D, int nR, int n_entity, int n_user, float* out, int64_t ldo, int table_bf16,
                               void* stream) {
    if (n_entity <= 0) return fail(-2, "%s: n_entity=%d", who, n_entity);
    if (uts && n_user <= 0) return fail(-2, "%s: n_user=%d", who, n_user);
    if (!entity_emb || !out || (!uts && !mem_h)) return fail(-1, "%s: null pointer", who);
    if (uts && !users64 && !users32) return fail(-1, "%s: user_triplet_set given without user ids", who);
    if (P < 0 || P > 8) return fail(-2, "%s: P=%d (0..8)", who, P);
    if (P == 0 && !w) return fail(-2, "%s: nothing to do (P == 0 and w == NULL)", who);
    if (P > 0 && (!V || nR <= 0 || (!uts && (!mem_r || !mem_t)))) return fail(-1, "%s: hops need V, mem_r, mem_t, nR", who);
    if (B <= 0 || Nm <= 0) return fail(-2, "%s: bad sizes B=%d Nm=%d", who, B, Nm);
    if (bad_dim(D)) return fail(-2, "%s: D=%d (need %%4==0, 4..%d)", who, D, MVIN_MAX_DIM);
    const int n_o = P + (w ? 1 : 0);
    if (ldo < (int64_t)n_o * D || (ldo & 3)) return fail(-2, "%s: ldo=%lld", who, (long long)ldo);
    mvin::KeyAddrArgs k{};
    k.E = entity_emb;
    k.V = V;
    k.w = w;
    k.uts = uts;
    k.users64 = users64;
    k.users32 = users32;
    k.n_user = n_user;
    k.n_entity = n_entity;
    const int nh = P > 0 ? P : 1;
    for (int i = 0; i < nh && !uts; ++i) {
        if (!mem_h[i]) return fail(-1, "%s: null mem_h[%d]", who, i);
        k.mem_h[i] = mem_h[i];
        if (i < P) {
            if (!mem_r[i] || !mem_t[i]) return fail(-1, "%s: null mem_r/mem_t[%d]", who, i);
            k.mem_r[i] = mem_r[i];
            k.mem_t[i] = mem_t[i];
        }
    }
    k.out = out;
    k.ldo = ldo;
    k.B = B;
    k.P = P;
    k.Nm = Nm;
    k.D = D;
    k.nR = nR;
    k.lpr_log2 = mvin::lpr_log2_for(D);
    k.table_bytes = (uint64_t)n_entity * D * (table_bf16 ? 2 : 4);
    if (!mvin::key_addr_stream_supported(k, table_bf16) && mvin::key_addr_nj(Nm, D) > 16)
        return fail(-3, "%s: unsupported shape Nm=%d D=%d (streaming kernel: Nm <= 64 and nR*D*4 <= 3072; "
                    "register-resident kernel: ceil(Nm / (64 / lanes per row)) <= 16)", who, Nm, D);
    return hip_result(mvin::launch_key_addr(k, table_bf16, (hipStream_t)stream), who);
}

int mvin_key_addressing_fwd(const void* entity_emb, const float* V, const float* w,
                            const int32_t* const* mem_h, const int32_t* const* mem_r,
                            const int32_t* const* mem_t, int P, int B, int Nm, int D, int nR, int n_entity,
                            float* out, int64_t ldo, int table_bf16, void* stream) {
    return key_addressing_impl("mvin_key_addressing_fwd", entity_emb, V, w, mem_h, mem_r, mem_t, nullptr, nullptr, nullptr, P,
                               B, Nm, D, nR, n_entity, 0, out, ldo, table_bf16, stream);
}

int mvin_key_addressing_users_fwd(const void* entity_emb, const float* V, const float* w, const int32_t* uts,
                                  const int64_t* users_i64, const int32_t* users_i32, int P, int B, int Nm, int D, int nR,
                                  int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream) {
    return key_addressing_impl("mvin_key_addressing_users_fwd", entity_emb, V, w, nullptr, nullptr, nullptr, uts, users_i64,
                               users_i32, P, B, Nm, D, nR, n_entity, n_user, out, ldo, table_bf16, stream);
}

int mvin_l2_tail_supported(int D) { return mvin::l2_tail_supported(D) ? 1 : 0; }

int mvin_l2_tail_fwd(const void* entity_emb, const int64_t* items_i64, const int32_t* items_i32, const float* q,
                     const float* user_o, const float* nagg0, const float* nagg1, const float* W0, const float* b0,
                     const float* A0, const float* a0, const float* A1, const float* a1, const float* Wmix,
                     const float* bmix, int64_t B, int D, int n_entity, float* item_emb, float* scores, float* sig,
                     int table_bf16, void* stream) {
    const char* who = "mvin_l2_tail_fwd";
    if (!mvin::l2_tail_supported(D)) return fail(-3, "%s: unsupported D=%d (16, 32, 64)", who, D);
    if (!entity_emb || !user_o || !nagg0 || !nagg1 || !A0 || !A1 || !Wmix || !scores) return fail(-1, "%s: null pointer", who);
    if ((items_i64 == nullptr) == (items_i32 == nullptr)) return fail(-1, "%s: exactly one of items_i64 / items_i32", who);
    if (W0 && !q) return fail(-1, "%s: the projection needs q", who);
    if (B <= 0 || n_entity <= 0) return fail(-2, "%s: bad sizes B=%lld n_entity=%d", who, (long long)B, n_entity);
    mvin::TailArgs t{};
    t.E = entity_emb;
    t.items64 = items_i64;
    t.items32 = items_i32;
    t.q = q;
    t.user_o = user_o;
    t.nagg0 = nagg0;
    t.nagg1 = nagg1;
    t.W0 = W0;
    t.b0 = b0;
    t.A0 = A0;
    t.a0 = a0;
    t.A1 = A1;
    t.a1 = a1;
    t.Wmix = Wmix;
    t.bmix = bmix;
    t.item_emb = item_emb;
    t.scores = scores;
    t.sig = sig;
    t.B = B;
    t.table_bf16 = table_bf16;
    t.n_entity = n_entity;
    return hip_result(mvin::launch_l2_tail(t, D, (hipStream_t)stream), who);
}

int mvin_score_l2_fwd(const mvin_score_l2_args* a, void* stream) {
    const char* who = "mvin_score_l2_fwd";
    if (!a) return fail(-1, "%s: null args", who);
    if (a->P < 1 || a->P > 8) return fail(-2, "%s: P=%d (1..8)", who, a->P);
    const bool grouped = a->uts && a->group_ws;
    if ((!a->V && !grouped) || !a->o_cat || !a->parents || !a->nagg0 || !a->nagg1 || !a->user_o || !a->scores || !a->items)
        return fail(-1, "%s: null workspace / output / items", who);
    if (a->uts && !a->users) return fail(-1, "%s: user_triplet_set given without user ids", who);
    if (a->group_ws && !a->uts) return fail(-1, "%s: group_ws is for the users feed (uts + users)", who);
    const int D = a->D, nR = a->n_relation;
    const int n_o = a->P + (a->h_set_w ? 1 : 0);
    // V[b, r, :] = E[item_b] . R_KGE[r]   (model.py:214-220 re-associated: (R h).v == h.(v R))
    mvin_linear_args l{};
    l.src[0] = reinterpret_cast<const float*>(a->entity_emb);
    l.ids[0] = reinterpret_cast<const int32_t*>(a->items);
    l.ids64 = 1;
    l.src_bf16 = a->table_bf16 ? 1 : 0;
    l.src_rows = a->n_entity;
    l.nsrc = 1;
    l.Dsrc = D;
    l.Dout = D;
    l.rows = a->B;
    l.W = a->relation_kge;
    l.rows_per_group = 1;
    l.out = a->V;
    l.ldo = (int64_t)nR * D;
    l.nz = nR;
    l.w_zstride = (int64_t)D * D;
    l.out_zstride = D;
    int rc = 0;
    bool flash = false;
    if (grouped) {
        // the batch in user order (device-side counting sort, no host sync), then a user's rows staged once per segment
        int32_t* ws = a->group_ws;
        int32_t* seg_user = ws + 2 * (size_t)a->n_user + (size_t)a->B;
        int32_t* seg_ptr = seg_user + a->B;
        int32_t* nseg = seg_ptr + a->B + 2;
        int32_t* pair_index = nseg + 1;
        rc = mvin_group_pairs_by_user(a->users, nullptr, a->B, a->n_user, ws, seg_user, seg_ptr, nseg, pair_index, stream);
        if (rc) return rc;
        if (a->ka_flash && a->user_records && !a->table_bf16 && mvin::key_addr_flash_supported(D, a->P, a->Nm, nR, a->n_entity)) {
            // flash form: per-call tables (R_KGE[r] . E[e], E . Wmlp blocks) from the CURRENT parameters, then ONE barrier-free kernel
            // for the attention reads and the user MLP.  Its scheduling scratch: the counters / offsets / ranks of the counting sort
            // (2 n_user + B words at the head of group_ws), dead once the batch is grouped
            rc = mvin_key_addressing_flash_prepare(reinterpret_cast<const float*>(a->entity_emb), a->relation_kge, a->h_set_w, a->user_mlp_W,
                                                   a->n_entity, nR, D, a->P, a->ka_flash, stream);
            if (rc) return rc;
            if (mvin_key_addressing_flash_ws_elems(a->B, a->n_user) > (size_t)2 * a->n_user + (size_t)a->B)
                return fail(-2, "%s: group_ws too small for the flash form's scheduling scratch", who);
            rc = mvin_key_addressing_flash_fwd(reinterpret_cast<const float*>(a->entity_emb), a->ka_flash, a->user_records, seg_user, seg_ptr,
                                               nseg, pair_index, a->items, nullptr, a->B, a->P, a->Nm, D, nR, a->n_entity, a->n_user,
                                               a->h_set_w != nullptr, a->user_mlp_b, a->user_o, ws, stream);
            flash = true;
        } else if (a->ka_er && a->user_records && !a->table_bf16 &&
            mvin_key_addressing_grouped_er_supported(D, a->P, a->Nm, nR, a->n_entity, a->h_set_w != nullptr)) {
            // gathered form of the U rows: R_KGE[r] . E[e] for every (relation, entity) from the CURRENT parameters, per call
            rc = mvin_project_relations(reinterpret_cast<const float*>(a->entity_emb), a->relation_kge, a->h_set_w, a->n_entity, nR, D,
                                        a->ka_er, stream);
            if (rc) return rc;
            rc = mvin_key_addressing_grouped_er_fwd(a->entity_emb, a->relation_kge, a->h_set_w, a->uts, a->user_records, a->ka_er, seg_user,
                                                    seg_ptr, nseg, pair_index, a->items, nullptr, (int)a->B, (int)a->B, a->P, a->Nm, D, nR,
                                                    a->n_entity, a->n_user, a->o_cat, (int64_t)n_o * D, stream);
        } else
        rc = mvin_key_addressing_grouped_rec_fwd(a->entity_emb, a->relation_kge, a->h_set_w, a->uts, a->user_records, seg_user, seg_ptr,
                                                 nseg, pair_index, a->items, nullptr, (int)a->B, (int)a->B, a->P, a->Nm, D, nR,
                                                 a->n_entity, a->n_user, a->o_cat, (int64_t)n_o * D, a->table_bf16, stream);
    } else {
        rc = mvin_linear_fwd(&l, stream);
        if (rc) return rc;
    }
    if (grouped) {
    } else if (a->uts)
        rc = mvin_key_addressing_users_fwd(a->entity_emb, a->V, a->h_set_w, a->uts, a->users, nullptr, a->P, (int)a->B, a->Nm,
                                           D, nR, a->n_entity, a->n_user, a->o_cat, (int64_t)n_o * D, a->table_bf16, stream);
    else
        rc = mvin_key_addressing_fwd(a->entity_emb, a->V, a->h_set_w, a->mem_h, a->mem_r, a->mem_t, a->P, (int)a->B, a->Nm, D,
                                     nR, a->n_entity, a->o_cat, (int64_t)n_o * D, a->table_bf16, stream);
    if (rc) return rc;
    if (!flash) {
    mvin_linear_args u{};                     // user_o = o_cat . user_mlp + bias (model.py:232-236)
    u.src[0] = a->o_cat;
    u.nsrc = 1;
    u.Dsrc = n_o * D;
    u.Dout = D;
    u.rows = a->B;
    u.W = a->user_mlp_W;
    u.bias = a->user_mlp_b;
    u.rows_per_group = 1;
    u.out = a->user_o;
    u.ldo = D;
    u.nz = 1;
    rc = mvin_linear_fwd(&u, stream);
    if (rc) return rc;
    }
    // the parents of a depth-2 tree are the items themselves: the kernel reads the int64 ids in place (no expand launch)
    if (a->fold_ws && a->fold_gather && a->enc_entity && a->enc_relation && a->W0 && a->W1 && a->W2 && !a->table_bf16 &&
        fold_gather_applies(D, a->K, a->n_entity, nR, a->B)) {
        // folded tail, every pair gathering its own rows: the four per-row tables (no aggregates), parents in item order, one launch
        rc = mvin_fold_tables_ex(reinterpret_cast<const float*>(a->entity_emb), a->enc_entity, a->enc_relation, a->t0, a->W0, a->b0, a->W1, a->b1,
                                 a->W2, a->b2, a->A0, a->a0, a->Wmix, a->bmix, a->A1, 0, a->K, D, a->n_entity, nR, a->fold_ws, stream);
        if (rc) return rc;
        const int32_t* order = nullptr;
        if (a->item_order_ws) {
            int32_t* ord = a->item_order_ws + mvin::order_ws_elems(a->B);
            rc = mvin_order_by_key(a->items, nullptr, a->B, a->item_order_ws, ord, stream);
            if (rc) return rc;
            order = ord;
        }
        return mvin_score_l2_folded_gather_fwd(a->fold_ws, a->enc_entity, a->enc_relation, a->items, nullptr, order, a->t0, a->t1, a->user_o,
                                               a->user_o, a->A1, a->a1, a->Wmix, a->B, a->K, D, a->n_entity, nR, a->item_emb, a->scores, a->sig,
                                               stream);
    }
    if (a->fold_ws && !a->fold_gather && a->enc_entity && a->enc_relation && a->W0 && a->W1 && a->W2 && !a->table_bf16 && fold_applies(D, a->K, a->n_entity, nR, a->B)) {
        // folded-tail form: four per-entity tables + the aggregates H0 | G from the CURRENT parameters, then two launches per batch
        rc = mvin_fold_tables(reinterpret_cast<const float*>(a->entity_emb), a->enc_entity, a->enc_relation, a->t0, a->W0, a->b0, a->W1, a->b1,
                              a->W2, a->b2, a->A0, a->a0, a->Wmix, a->bmix, a->A1, a->K, D, a->n_entity, nR, a->fold_ws, stream);
        if (rc) return rc;
        return mvin_score_l2_folded_fwd(a->fold_ws, a->enc_entity, a->enc_relation, a->items, nullptr, a->t0, a->t1, a->user_o, a->user_o, a->A1,
                                        a->a1, a->Wmix, a->B, a->K, D, a->n_entity, nR, a->nagg0, a->nagg1, a->item_emb, a->scores, a->sig,
                                        stream);
    }
    const bool enc = a->enc_entity && a->enc_relation && mvin::fused_packed_supported(D, a->K);
    const bool d32 = mvin::fused_d32_supported(D, a->K);       // the wave-per-parent kernel: projected tables over either adjacency
    // (asked for on a shape / table size the projected-tables kernels do not take: the unprojected form, not an error)
    if ((enc || d32) && a->prj_tables && a->W1 && !a->table_bf16 && prj_applies(D, a->K, enc, a->n_entity, nR, a->B)) {
        // projected-tables form: E.W1 | E.W1.A0 | E.W2.A0 once per entity from the CURRENT parameters (nothing is kept between
        // calls), then the packed kernel without any product per distinct child
        rc = mvin_project_tables(reinterpret_cast<const float*>(a->entity_emb), a->W1, a->W2, a->b1, a->b2, a->A0, a->a0,
                                 a->t0 != nullptr, a->K, a->n_entity, D, a->prj_tables, stream);
        if (rc) return rc;
        const int32_t* order = nullptr;
        if (enc && a->agg_tables && agg_applies(D, a->K, a->n_entity, nR, a->B)) {
            // per-entity aggregates form: S0 | G once per entity from the tables just built, then ~cnt rows per pair instead of ~cnt^2
            rc = mvin_entity_aggregates(a->prj_tables, a->enc_entity, a->enc_relation, a->t0, a->K, D, a->n_entity, nR, a->agg_tables, stream);
            if (rc) return rc;
            if (a->item_order_ws) {
                int32_t* ord = a->item_order_ws + mvin::order_ws_elems(a->B);
                rc = mvin_order_by_key(a->items, nullptr, a->B, a->item_order_ws, ord, stream);
                if (rc) return rc;
                order = ord;
            }
            rc = mvin_gather_attn_l2_agg_fwd(a->prj_tables, a->agg_tables, a->enc_entity, a->enc_relation, a->items, 1, order, a->t0, a->t1,
                                             a->user_o, (int)a->B, 1, a->K, D, a->n_entity, nR, a->nagg0, a->nagg1, stream);
        } else {
        if (enc && a->item_order_ws && mvin::fused_wpp_supported(D, a->K) && nR <= 4096) {
            // item order for the wave-per-parent kernel (it takes the launch when fused_wpp_applies; otherwise the order is not passed on)
            mvin::FusedL2Args f{};
            f.prj = 1, f.K = a->K, f.nR = nR, f.P = a->B, f.parents_per_pair = 1, f.max_id = (unsigned)(a->n_entity - 1);
            f.table_bytes = (uint64_t)a->n_entity * D * 4, f.adj_bytes = (uint64_t)a->n_entity * a->K * 4;
            f.adj_r = a->enc_relation, f.W1 = a->W1, f.W2 = a->W2, f.q = a->user_o;
            if (mvin::fused_wpp_applies(f, D)) {
                int32_t* ord = a->item_order_ws + mvin::order_ws_elems(a->B);
                rc = mvin_order_by_key(a->items, nullptr, a->B, a->item_order_ws, ord, stream);
                if (rc) return rc;
                order = ord;
            }
        }
        rc = mvin_gather_attn_l2_prj_ordered_fwd(a->prj_tables, enc ? a->enc_entity : a->adj_entity, enc ? a->enc_relation : a->adj_relation,
                                                 enc ? 1 : 0, a->items, 1, order, a->t0, a->t1, a->user_o, (int)a->B, 1, a->K, D, a->n_entity, nR,
                                                 a->nagg0, a->nagg1, stream);
        }
    } else
    rc = gather_attn_l2_impl(a->entity_emb, enc ? a->enc_entity : a->adj_entity, enc ? a->enc_relation : a->adj_relation,
                             reinterpret_cast<const int32_t*>(a->items), 2,
                             a->t0, a->t1, a->W1, a->W2, a->b1, a->b2, a->W1 ? a->user_o : nullptr, a->A0, a->a0,
                             (int)a->B, 1, a->K, D, a->n_entity, nR, a->nagg0, a->nagg1, nullptr, nullptr,
                             a->table_bf16, stream, enc);
    if (rc) return rc;
    return mvin_l2_tail_fwd(a->entity_emb, a->items, nullptr, a->W0 ? a->user_o : nullptr, a->user_o, a->nagg0, a->nagg1,
                            a->W0, a->b0, a->A0, a->a0, a->A1, a->a1, a->Wmix, a->bmix, a->B, D, a->n_entity, a->item_emb,
                            a->scores, a->sig, a->table_bf16, stream);
}

int mvin_score_small_supported(int D, int K, int P, int Nm, int nR) { return mvin::score_small_supported(D, K, P, Nm, nR) ? 1 : 0; }

int mvin_score_small_fwd(const mvin_score_l2_args* a, int group, void* stream) {
    const char* who = "mvin_score_small_fwd";
    if (!a) return fail(-1, "%s: null args", who);
    if (!mvin::score_small_supported(a->D, a->K, a->P, a->Nm, a->n_relation))
        return fail(-3, "%s: unsupported shape D=%d K=%d P=%d Nm=%d nR=%d", who, a->D, a->K, a->P, a->Nm, a->n_relation);
    if (a->depth < 0 || a->depth > 2) return fail(-2, "%s: depth=%d (1 or 2)", who, a->depth);
    const bool d1 = a->depth == 1;
    if (!a->entity_emb || !a->adj_entity || !a->relation_kge || !a->user_mlp_W || !a->A0 || (!d1 && !a->A1) || !a->Wmix || !a->items ||
        !a->user_o || !a->scores)
        return fail(-1, "%s: null table / weight / items / output", who);
    if ((a->W0 == nullptr) != (a->W1 == nullptr) || (!d1 && (a->W1 == nullptr) != (a->W2 == nullptr)))
        return fail(-1, "%s: W0 / W1 (/ W2) must be given together (User_orient) or not at all", who);
    if (a->uts ? !a->users : (!a->mem_h || !a->mem_r || !a->mem_t))
        return fail(-1, "%s: needs uts + users, or mem_h / mem_r / mem_t", who);
    if (a->B <= 0 || a->n_entity <= 0 || (a->uts && a->n_user <= 0)) return fail(-2, "%s: bad sizes B=%lld", who, (long long)a->B);
    const bool enc = a->enc_entity && a->enc_relation;
    if (enc && a->n_entity > (1 << 24)) return fail(-2, "%s: the encoded adjacency holds 24-bit ids", who);
    if (a->table_bf16) return fail(-3, "%s: fp32 entity table only", who);
    const uint64_t table_bytes = (uint64_t)a->n_entity * a->D * 4, adj_bytes = (uint64_t)a->n_entity * a->K * 4;
    if (table_bytes >= (1ull << 32) - 4096 || adj_bytes >= (1ull << 32) - 4096)
        return fail(-3, "%s: entity table / adjacency of 4 GiB or more (32-bit buffer offsets)", who);
    mvin::ScoreSmallArgs s{};
    s.E = a->entity_emb;
    s.adj_e = enc ? a->enc_entity : a->adj_entity;
    s.adj_r = enc ? a->enc_relation : a->adj_relation;
    s.R = a->relation_kge;
    s.w_h = a->h_set_w;
    s.Wu = a->user_mlp_W;
    s.bu = a->user_mlp_b;
    s.t0 = a->t0;
    s.t1 = a->t1;
    s.W0 = a->W0, s.b0 = a->b0, s.W1 = a->W1, s.b1 = a->b1, s.W2 = a->W2, s.b2 = a->b2;
    s.A0 = a->A0, s.a0 = a->a0, s.A1 = a->A1, s.a1 = a->a1, s.Wmix = a->Wmix, s.bmix = a->bmix;
    s.items = a->items;
    if (a->uts) {
        s.uts = a->uts;
        s.users = a->users;
    } else {
        for (int h = 0; h < a->P; ++h) {
            if (!a->mem_h[h] || !a->mem_r[h] || !a->mem_t[h]) return fail(-1, "%s: null ripple-set array of hop %d", who, h);
            s.mem_h[h] = a->mem_h[h], s.mem_r[h] = a->mem_r[h], s.mem_t[h] = a->mem_t[h];
        }
    }
    s.user_o = a->user_o, s.item_emb = a->item_emb, s.scores = a->scores, s.sig = a->sig;
    s.B = a->B;
    s.G = group;
    s.K = a->K, s.P = a->P, s.Nm = a->Nm, s.nR = a->n_relation, s.n_entity = a->n_entity, s.n_user = a->n_user > 0 ? a->n_user : 1;
    s.enc = enc ? 1 : 0;
    s.depth1 = d1 ? 1 : 0;
    s.table_bf16 = 0;
    s.table_bytes = table_bytes;
    s.adj_bytes = adj_bytes;
    return hip_result(mvin::launch_score_small(s, a->D, (hipStream_t)stream), who);
}

int mvin_gather_rows(const void* table, const int32_t* ids, int64_t n, int row_bytes, void* out, void* stream) {
    const char* who = "mvin_gather_rows";
    if (n < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail(-2, "%s: n=%lld row_bytes=%d", who, (long long)n, row_bytes);
    if (n > 0 && (!table || !ids || !out)) return fail(-1, "%s: null pointer", who);
    return hip_result(mvin::launch_move_rows(const_cast<void*>(table), ids, n, row_bytes, out, false, (hipStream_t)stream), who);
}

int mvin_scatter_rows(void* table, const int32_t* ids, int64_t n, int row_bytes, const void* rows, void* stream) {
    const char* who = "mvin_scatter_rows";
    if (n < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail(-2, "%s: n=%lld row_bytes=%d", who, (long long)n, row_bytes);
    if (n > 0 && (!table || !ids || !rows)) return fail(-1, "%s: null pointer", who);
    return hip_result(mvin::launch_move_rows(table, ids, n, row_bytes, const_cast<void*>(rows), true, (hipStream_t)stream), who);
}

int mvin_shard_space_ids(const void* ids, int ids_are_i64, int64_t n, int world, int n_local, void* out, void* stream) {
    const char* who = "mvin_shard_space_ids";
    if (n < 0 || world <= 0 || n_local <= 0) return fail(-2, "%s: n=%lld world=%d n_local=%d", who, (long long)n, world, n_local);
    if (n > 0 && (!ids || !out)) return fail(-1, "%s: null pointer", who);
    return hip_result(mvin::launch_shard_space_ids(ids, ids_are_i64 != 0, n, world, n_local, out, (hipStream_t)stream), who);
}

int mvin_group_pairs_by_user(const int64_t* users_i64, const int32_t* users_i32, int64_t B, int n_user, int32_t* workspace,
                             int32_t* seg_user, int32_t* seg_ptr, int32_t* nseg, int32_t* pair_index, void* stream) {
    const char* who = "mvin_group_pairs_by_user";
    if ((!users_i64 && !users_i32) || !workspace || !seg_user || !seg_ptr || !nseg || !pair_index)
        return fail(-1, "%s: null pointer", who);
    if (B <= 0 || B > 0x7fffffff || n_user <= 0) return fail(-2, "%s: bad sizes B=%lld n_user=%d", who, (long long)B, n_user);
    return hip_result(mvin::launch_group_pairs(users_i64, users_i32, B, n_user, workspace, workspace + n_user, workspace + 2 * (size_t)n_user,
                                               seg_user, seg_ptr,
                                               nseg, pair_index, (hipStream_t)stream), who);
}

int mvin_key_addressing_grouped_supported(int D, int P, int Nm, int nR) {
    return mvin::key_addr_grouped_supported(D, P, Nm, nR) ? 1 : 0;
}

int mvin_user_records_len(int P, int Nm, int nR) { return mvin::ka_rec_layout(P, Nm, nR).len; }

int mvin_user_records_supported(int D, int P, int Nm, int nR, int table_bf16) {
    return !table_bf16 && mvin::key_addr_static_supported(D, P, Nm, nR) ? 1 : 0;
}

int mvin_build_user_records(const int32_t* uts, int n_user, int P, int Nm, int nR, int n_entity, int32_t* records, void* stream) {
    const char* who = "mvin_build_user_records";
    if (!uts || !records) return fail(-1, "%s: null pointer", who);
    if (n_user <= 0 || n_entity <= 0) return fail(-2, "%s: bad sizes n_user=%d n_entity=%d", who, n_user, n_entity);
    if (mvin::ka_rec_layout(P, Nm, nR).len == 0) return fail(-3, "%s: no record form for P=%d Nm=%d nR=%d (P 1..8, Nm 1..256, nR 1..4096)", who, P, Nm, nR);
    return hip_result(mvin::launch_user_records(uts, n_user, P, Nm, nR, n_entity, records, (hipStream_t)stream), who);
}

int mvin_key_addressing_grouped_fwd(const void* entity_emb, const float* relation_kge, const float* w,
                                    const int32_t* uts, const int32_t* seg_user, const int32_t* seg_ptr,
                                    const int32_t* nseg_dev, const int32_t* pair_index, const int64_t* items_i64,
                                    const int32_t* items_i32, int nseg, int B, int P, int Nm, int D, int nR,
                                    int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream) {
    return mvin_key_addressing_grouped_rec_fwd(entity_emb, relation_kge, w, uts, nullptr, seg_user, seg_ptr, nseg_dev, pair_index, items_i64,
                                               items_i32, nseg, B, P, Nm, D, nR, n_entity, n_user, out, ldo, table_bf16, stream);
}

size_t mvin_project_relations_elems(int n_entity, int nR, int D) {
    if (n_entity <= 0 || nR <= 0 || D <= 0) return 0;
    return (size_t)nR * n_entity * D + (size_t)((n_entity + 3) & ~3) + (size_t)nR * D * D;
}

int mvin_project_relations(const float* entity_emb, const float* relation_kge, const float* w, int n_entity, int nR, int D, float* ws,
                           void* stream) {
    const char* who = "mvin_project_relations";
    if (!entity_emb || !relation_kge || !ws) return fail(-1, "%s: null pointer", who);
    if (n_entity <= 0 || nR <= 0 || (D != 16 && D != 32 && D != 64 && D != 128)) return fail(-2, "%s: n_entity=%d nR=%d D=%d", who, n_entity, nR, D);
    float* hs = ws + (size_t)nR * n_entity * D;
    float* RT = hs + (size_t)((n_entity + 3) & ~3);
    if (int rc = hip_result(mvin::launch_transpose_blocks(relation_kge, nR, D, RT, (hipStream_t)stream), who)) return rc;
    mvin_linear_args l{};                     // ER[r][e][n] = sum_k E[e][k] R_KGE[r][n][k]  (= R_KGE[r] . E[e], model.py:214-220)
    l.src[0] = entity_emb;
    l.nsrc = 1;
    l.Dsrc = D;
    l.Dout = D;
    l.rows = n_entity;
    l.rows_per_group = 1;
    l.W = RT;
    l.w_zstride = (int64_t)D * D;
    l.out = ws;
    l.ldo = D;
    l.nz = nR;
    l.out_zstride = (int64_t)n_entity * D;
    if (int rc = mvin_linear_fwd(&l, stream)) return rc;
    if (w) return hip_result(mvin::launch_entity_dot(entity_emb, w, n_entity, D, hs, (hipStream_t)stream), who);
    return 0;
}

static int key_addressing_grouped_impl(const void* entity_emb, const float* relation_kge, const float* w,
                                        const int32_t* uts, const int32_t* user_records, const int32_t* seg_user,
                                        const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                        const int64_t* items_i64, const int32_t* items_i32, int nseg, int B, int P, int Nm, int D,
                                        int nR, int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream,
                                        const float* er_ws);

int mvin_key_addressing_flash_supported(int D, int P, int Nm, int nR, int n_entity) {
    return mvin::key_addr_flash_supported(D, P, Nm, nR, n_entity) ? 1 : 0;
}

static int flash_nseg_bound(int64_t B, int n_user) { return (int)(B < (int64_t)n_user ? B : (int64_t)n_user); }

size_t mvin_key_addressing_flash_ws_elems(int64_t B, int n_user) {
    if (B <= 0 || n_user <= 0) return 0;
    return mvin::key_addr_flash_ws_elems(B, flash_nseg_bound(B, n_user));
}

size_t mvin_key_addressing_flash_tables_elems(int n_entity, int nR, int D, int P, int has_set) {
    if (n_entity <= 0 || nR <= 0 || D <= 0 || P < 0) return 0;
    return mvin_project_relations_elems(n_entity, nR, D) + (size_t)(P + (has_set ? 1 : 0)) * n_entity * D;
}

int mvin_key_addressing_flash_prepare(const float* entity_emb, const float* relation_kge, const float* w, const float* user_mlp_W,
                                      int n_entity, int nR, int D, int P, float* ws, void* stream) {
    const char* who = "mvin_key_addressing_flash_prepare";
    if (!entity_emb || !relation_kge || !user_mlp_W || !ws) return fail(-1, "%s: null pointer", who);
    if (n_entity <= 0 || nR <= 0 || D != 64 || P < 1 || P > 8) return fail(-2, "%s: n_entity=%d nR=%d D=%d P=%d", who, n_entity, nR, D, P);
    if (int rc = mvin_project_relations(entity_emb, relation_kge, w, n_entity, nR, D, ws, stream)) return rc;
    mvin_linear_args l{};                     // TW[j][e][n] = sum_k E[e][k] Wmlp[D j + k][n]  (model.py:232-236 taken per entity)
    l.src[0] = entity_emb;
    l.nsrc = 1;
    l.Dsrc = D;
    l.Dout = D;
    l.rows = n_entity;
    l.rows_per_group = 1;
    l.W = user_mlp_W;
    l.w_zstride = (int64_t)D * D;
    l.out = ws + mvin_project_relations_elems(n_entity, nR, D);
    l.ldo = D;
    l.nz = P + (w ? 1 : 0);
    l.out_zstride = (int64_t)n_entity * D;
    return mvin_linear_fwd(&l, stream);
}

int mvin_key_addressing_flash_fwd(const float* entity_emb, const float* ws, const int32_t* user_records, const int32_t* seg_user,
                                  const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index, const int64_t* items_i64,
                                  const int32_t* items_i32, int64_t B, int P, int Nm, int D, int nR, int n_entity, int n_user, int has_set,
                                  const float* user_mlp_b, float* user_o, int32_t* sched_ws, void* stream) {
    const char* who = "mvin_key_addressing_flash_fwd";
    if (!entity_emb || !ws || !user_records || !seg_user || !seg_ptr || !pair_index || !user_o || !sched_ws)
        return fail(-1, "%s: null pointer", who);
    if ((items_i64 == nullptr) == (items_i32 == nullptr)) return fail(-1, "%s: exactly one of items_i64 / items_i32", who);
    if (B <= 0 || B >= (int64_t(1) << 31) || n_user <= 0 || n_entity <= 0 || nR <= 0)
        return fail(-2, "%s: bad sizes B=%lld n_user=%d n_entity=%d nR=%d", who, (long long)B, n_user, n_entity, nR);
    if (!mvin::key_addr_flash_supported(D, P, Nm, nR, n_entity))
        return fail(-3, "%s: unsupported shape D=%d P=%d Nm=%d nR=%d n_entity=%d (D = 64, 1 <= P <= 8, Nm <= 64, nR * n_entity < 2^31)", who,
                    D, P, Nm, nR, n_entity);
    mvin::KaFlashArgs k{};
    k.E = entity_emb;
    k.ER = ws;                                                          // (the layout of mvin_project_relations' workspace, then TW)
    k.hs = has_set ? ws + (size_t)nR * n_entity * D : nullptr;
    k.TW = ws + mvin_project_relations_elems(n_entity, nR, D);
    k.records = user_records;
    k.seg_user = seg_user;
    k.seg_ptr = seg_ptr;
    k.nseg_dev = nseg_dev;
    k.pair_index = pair_index;
    k.items64 = items_i64;
    k.items32 = items_i32;
    k.bmlp = user_mlp_b;
    k.user_o = user_o;
    k.P = P;
    k.Nm = Nm;
    k.nR = nR;
    k.n_entity = n_entity;
    k.B = B;
    return hip_result(mvin::launch_key_addr_flash(k, flash_nseg_bound(B, n_user), has_set != 0, sched_ws, (hipStream_t)stream), who);
}

int mvin_key_addressing_grouped_rec_fwd(const void* entity_emb, const float* relation_kge, const float* w,
                                        const int32_t* uts, const int32_t* user_records, const int32_t* seg_user,
                                        const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                        const int64_t* items_i64, const int32_t* items_i32, int nseg, int B, int P, int Nm, int D,
                                        int nR, int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream) {
    return key_addressing_grouped_impl(entity_emb, relation_kge, w, uts, user_records, seg_user, seg_ptr, nseg_dev, pair_index, items_i64,
                                       items_i32, nseg, B, P, Nm, D, nR, n_entity, n_user, out, ldo, table_bf16, stream, nullptr);
}

int mvin_key_addressing_grouped_er_supported(int D, int P, int Nm, int nR, int n_entity, int has_set) {
    return (D == 64 && mvin::key_addr_static_er_ok(P, Nm, nR, n_entity, has_set != 0)) ? 1 : 0;
}

int mvin_key_addressing_grouped_er_fwd(const void* entity_emb, const float* relation_kge, const float* w,
                                       const int32_t* uts, const int32_t* user_records, const float* er_ws, const int32_t* seg_user,
                                       const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                       const int64_t* items_i64, const int32_t* items_i32, int nseg, int B, int P, int Nm, int D,
                                       int nR, int n_entity, int n_user, float* out, int64_t ldo, void* stream) {
    const char* who = "mvin_key_addressing_grouped_er_fwd";
    if (!user_records || !er_ws) return fail(-1, "%s: the gathered form needs the user records and the workspace of mvin_project_relations", who);
    if (!mvin_key_addressing_grouped_er_supported(D, P, Nm, nR, n_entity, w != nullptr))
        return fail(-3, "%s: unsupported shape D=%d P=%d Nm=%d nR=%d n_entity=%d", who, D, P, Nm, nR, n_entity);
    return key_addressing_grouped_impl(entity_emb, relation_kge, w, uts, user_records, seg_user, seg_ptr, nseg_dev, pair_index, items_i64,
                                       items_i32, nseg, B, P, Nm, D, nR, n_entity, n_user, out, ldo, 0, stream, er_ws);
}

static int key_addressing_grouped_impl(const void* entity_emb, const float* relation_kge, const float* w,
                                        const int32_t* uts, const int32_t* user_records, const int32_t* seg_user,
                                        const int32_t* seg_ptr, const int32_t* nseg_dev, const int32_t* pair_index,
                                        const int64_t* items_i64, const int32_t* items_i32, int nseg, int B, int P, int Nm, int D,
                                        int nR, int n_entity, int n_user, float* out, int64_t ldo, int table_bf16, void* stream,
                                        const float* er_ws) {
    const char* who = "mvin_key_addressing_grouped_fwd";
    if (!entity_emb || !uts || !seg_user || !seg_ptr || !pair_index || !out) return fail(-1, "%s: null pointer", who);
    if ((items_i64 == nullptr) == (items_i32 == nullptr)) return fail(-1, "%s: exactly one of items_i64 / items_i32", who);
    if (P < 0 || P > 8) return fail(-2, "%s: P=%d (0..8)", who, P);
    if (P == 0 && !w) return fail(-2, "%s: nothing to do (P == 0 and w == NULL)", who);
    if (P > 0 && !relation_kge) return fail(-1, "%s: hops need relation_kge", who);
    if (nseg <= 0 || B <= 0 || Nm <= 0 || n_entity <= 0 || n_user <= 0 || nR <= 0)
        return fail(-2, "%s: bad sizes nseg=%d B=%d Nm=%d n_entity=%d n_user=%d nR=%d", who, nseg, B, Nm, n_entity, n_user, nR);
    const int n_o = P + (w ? 1 : 0);
    if (ldo < (int64_t)n_o * D || (ldo & 3)) return fail(-2, "%s: ldo=%lld", who, (long long)ldo);
    if (!mvin::key_addr_grouped_supported(D, P, Nm, nR))
        return fail(-3, "%s: unsupported shape D=%d P=%d Nm=%d nR=%d (D in {16,32,64,128}; rows + V tile must fit 160 KB "
                    "of LDS)", who, D, P, Nm, nR);
    mvin::KeyAddrGroupedArgs k{};
    k.E = entity_emb;
    k.R = relation_kge;
    k.w = w;
    k.uts = uts;
    k.seg_user = seg_user;
    k.seg_ptr = seg_ptr;
    k.nseg_dev = nseg_dev;
    k.pair_index = pair_index;
    k.items64 = items_i64;
    k.items32 = items_i32;
    k.out = out;
    k.ldo = ldo;
    k.nseg = nseg;
    k.P = P;
    k.Nm = Nm;
    k.D = D;
    k.nR = nR;
    k.n_entity = n_entity;
    k.records = user_records;
    if (er_ws) {
        k.ER = er_ws;
        k.hs = er_ws + (size_t)nR * n_entity * D;
    }
    return hip_result(mvin::launch_key_addr_grouped(k, table_bf16, (hipStream_t)stream), who);
}

int mvin_gather_mix_fwd(const void* table, const int32_t* adj_entity, const int32_t* adj_relation,
                        const int32_t* node_ids, const float* rel_score, const float* rowbias, int64_t nodes,
                        int nodes_per_group, int K, int D, int n_entity, int nR, int relu, float* out,
                        int table_bf16, void* stream) {
    const char* who = "mvin_gather_mix_fwd";
    if (!table || !adj_entity || !out) return fail(-1, "%s: null pointer", who);
    if (rel_score && (!adj_relation || nR <= 0)) return fail(-1, "%s: rel_score needs adj_relation and nR", who);
    if (nodes <= 0 || K <= 0 || n_entity <= 0) return fail(-2, "%s: bad sizes nodes=%lld K=%d n_entity=%d", who,
                                                          (long long)nodes, K, n_entity);
    if (!node_ids && nodes > n_entity) return fail(-2, "%s: nodes=%lld > n_entity=%d without node_ids", who,
                                                   (long long)nodes, n_entity);
    if (rowbias && nodes_per_group <= 0) return fail(-2, "%s: nodes_per_group=%d", who, nodes_per_group);
    if (bad_dim(D)) return fail(-2, "%s: D=%d (need %%4==0, 4..%d)", who, D, MVIN_MAX_DIM);
    if (K > 256) return fail(-3, "%s: K=%d > 256 is outside this kernel", who, K);
    mvin::GatherMixArgs g{};
    g.table = table;
    g.adj_e = adj_entity;
    g.adj_r = adj_relation;
    g.node_ids = node_ids;
    g.rel_score = rel_score;
    g.rowbias = rowbias;
    g.out = out;
    g.nodes = nodes;
    g.npg = nodes_per_group > 0 ? nodes_per_group : 1;
    g.K = K;
    g.D = D;
    g.lpr_log2 = mvin::lpr_log2_for(D);
    g.relu = relu ? 1 : 0;
    g.table_bf16 = table_bf16 ? 1 : 0;
    return hip_result(mvin::launch_gather_mix(g, (hipStream_t)stream), who);
}

int mvin_row_softmax_fwd(const float* x, int64_t rows, int n, float* out, void* stream) {
    if (!x || !out) return fail(-1, "mvin_row_softmax_fwd: null pointer");
    if (rows <= 0 || n <= 0 || n > 4096) return fail(-2, "mvin_row_softmax_fwd: bad sizes rows=%lld n=%d", (long long)rows, n);
    return hip_result(mvin::launch_row_softmax(x, rows, n, out, (hipStream_t)stream), "mvin_row_softmax_fwd");
}

int mvin_mix_neighbor_vectors_fwd(const float* neighbor_vectors, const float* neighbor_relations, const float* user_embeddings,
                                  const float* logits_or_null, int B, int N, int K, int D, float* out, float* probs_or_null,
                                  void* stream) {
    const char* who = "mvin_mix_neighbor_vectors_fwd";
    if (!neighbor_vectors || !out) return fail(-1, "%s: null pointer", who);
    if ((neighbor_relations == nullptr) != (user_embeddings == nullptr))
        return fail(-1, "%s: neighbor_relations and user_embeddings go together", who);
    if (B <= 0 || N <= 0 || K <= 0 || K > 64 || D <= 0) return fail(-2, "%s: bad sizes B=%d N=%d K=%d D=%d (K <= 64)", who, B, N, K, D);
    return hip_result(mvin::launch_mix_urv(neighbor_vectors, neighbor_relations, user_embeddings, logits_or_null, (int64_t)B * N, N, K,
                                           D, out, probs_or_null, (hipStream_t)stream), who);
}

int mvin_sample_adjacency(const int64_t* indptr, const int32_t* dst, const int32_t* rel, int n_entity, int K,
                          uint64_t seed, int32_t* adj_entity, int32_t* adj_relation, void* stream) {
    if (!indptr || !dst || !rel || !adj_entity || !adj_relation) return fail(-1, "mvin_sample_adjacency: null pointer");
    if (n_entity <= 0 || K <= 0) return fail(-2, "mvin_sample_adjacency: bad sizes n_entity=%d K=%d", n_entity, K);
    return hip_result(mvin::launch_sample_adjacency(indptr, dst, rel, n_entity, K, seed, adj_entity, adj_relation,
                                                    (hipStream_t)stream), "mvin_sample_adjacency");
}

int mvin_build_ripple_sets(const int64_t* indptr, const int32_t* dst, const int32_t* rel, const int64_t* hist_ptr,
                           const int32_t* hist_items, int n_user, int P, int Nm, int n_neighbor, uint64_t seed,
                           int32_t* out, void* stream) {
    if (!indptr || !dst || !rel || !hist_ptr || !hist_items || !out) return fail(-1, "mvin_build_ripple_sets: null pointer");
    if (n_user <= 0 || P <= 0 || Nm <= 0 || Nm > 4096 || n_neighbor <= 0 || n_neighbor > 32)
        return fail(-2, "mvin_build_ripple_sets: bad sizes n_user=%d P=%d Nm=%d n_neighbor=%d", n_user, P, Nm, n_neighbor);
    mvin::RippleBuildArgs r{};
    r.indptr = indptr;
    r.dst = dst;
    r.rel = rel;
    r.hist_ptr = hist_ptr;
    r.hist_items = hist_items;
    r.out = out;
    r.n_user = n_user;
    r.seed = seed;
    r.P = P;
    r.Nm = Nm;
    r.n_neighbor = n_neighbor;
    return hip_result(mvin::launch_ripple_build(r, (hipStream_t)stream), "mvin_build_ripple_sets");
}

// ---------------------------------------------------------------------------- training
int mvin_count_ids(const int32_t* ids, int64_t n, int nbins, float* out, void* stream) {
    if (n < 0 || nbins < 1 || nbins > 4096) return fail(-2, "mvin_count_ids: n=%lld nbins=%d (1..4096)", (long long)n, nbins);
    if (n == 0) return 0;
    if (!ids || !out) return fail(-1, "mvin_count_ids: null pointer");
    return hip_result(mvin::launch_count_ids(ids, n, nbins, out, (hipStream_t)stream), "mvin_count_ids");
}

int mvin_eltwise(int mode, int64_t n, float* x, float* y, float* z, float* w, float* accum, float alpha, float beta,
                 float beta1, float beta2, float eps, int D, int N, void* stream) {
    if (mode < 0 || mode > 8) return fail(-2, "mvin_eltwise: mode=%d", mode);
    if (mode == 6 && (!y || D <= 0 || N <= 0)) return fail(-2, "mvin_eltwise: mode 6 needs y, D, N");
    if (n < 0) return fail(-2, "mvin_eltwise: n < 0");
    if (n == 0) return 0;
    if (!x) return fail(-1, "mvin_eltwise: null x");
    if ((mode == 0 || mode == 1 || mode == 2 || mode == 4 || mode == 5) && !y) return fail(-1, "mvin_eltwise: null y");
    if ((mode == 1 || mode == 2 || mode == 4 || mode == 5 || mode == 7 || mode == 8) && !z) return fail(-1, "mvin_eltwise: null z");
    if (mode == 4 && !w) return fail(-1, "mvin_eltwise: null w");
    if ((mode == 3 || mode == 7 || mode == 8) && !accum) return fail(-1, "mvin_eltwise: null accum");
    if ((mode == 5 || mode == 7 || mode == 8) && D <= 0) return fail(-2, "mvin_eltwise: D <= 0");
    mvin::EltArgs e{};
    e.mode = mode;
    e.n = n;
    e.x = x;
    e.y = y;
    e.z = z;
    e.w = w;
    e.accum = accum;
    e.alpha = alpha;
    e.beta = beta;
    e.beta1 = beta1;
    e.beta2 = beta2;
    e.eps = eps;
    e.D = D > 0 ? D : 1;
    e.N = N > 0 ? N : 1;
    return hip_result(mvin::launch_eltwise(e, (hipStream_t)stream), "mvin_eltwise");
}

static int l2_adam_multi_impl(const char* who, const mvin_param_seg* segs_device, int nseg, int64_t total,
                              float* g_flat, float* m_flat, float* v_flat, float* loss_accum, int apply_adam,
                              float lr_t, const float* lr_t_device, float beta1, float beta2, float eps, void* stream) {
    if (!segs_device || !g_flat) return fail(-1, "%s: null pointer", who);
    if (nseg <= 0 || nseg > 256) return fail(-2, "%s: nseg=%d (1..256)", who, nseg);
    if (total <= 0) return fail(-2, "%s: total=%lld", who, (long long)total);
    if (apply_adam && (!m_flat || !v_flat)) return fail(-1, "%s: Adam step needs the moment buffers", who);
    return hip_result(mvin::launch_l2_adam_multi(segs_device, nseg, total, g_flat, m_flat, v_flat, loss_accum,
                                                 apply_adam, lr_t, lr_t_device, beta1, beta2, eps,
                                                 (hipStream_t)stream), who);
}

int mvin_l2_adam_multi(const mvin_param_seg* segs_device, int nseg, int64_t total, float* g_flat, float* m_flat,
                       float* v_flat, float* loss_accum, int apply_adam, float lr_t, float beta1, float beta2,
                       float eps, void* stream) {
    return l2_adam_multi_impl("mvin_l2_adam_multi", segs_device, nseg, total, g_flat, m_flat, v_flat, loss_accum,
                              apply_adam, lr_t, nullptr, beta1, beta2, eps, stream);
}

int mvin_l2_adam_multi_dev(const mvin_param_seg* segs_device, int nseg, int64_t total, float* g_flat, float* m_flat,
                           float* v_flat, float* loss_accum, int apply_adam, const float* lr_t_device, float beta1,
                           float beta2, float eps, void* stream) {
    if (!lr_t_device) return fail(-1, "mvin_l2_adam_multi_dev: null lr_t_device");
    return l2_adam_multi_impl("mvin_l2_adam_multi_dev", segs_device, nseg, total, g_flat, m_flat, v_flat,
                              loss_accum, apply_adam, 0.f, lr_t_device, beta1, beta2, eps, stream);
}

int mvin_scatter_add_rows(float* dtable, const void* ids, int ids64, const float* x, int64_t rows, int D, float alpha,
                          void* stream) {
    if (!dtable || !ids || !x) return fail(-1, "mvin_scatter_add_rows: null pointer");
    if (rows < 0 || D < 4 || (D & 3)) return fail(-2, "mvin_scatter_add_rows: bad sizes rows=%lld D=%d", (long long)rows, D);
    if (rows == 0) return 0;
    return hip_result(mvin::launch_scatter_add_rows(dtable, (const int32_t*)ids, ids64, x, rows, D, alpha,
                                                    (hipStream_t)stream), "mvin_scatter_add_rows");
}

int mvin_linear_wgrad(const mvin_linear_args* a, const float* dY, int64_t ldy, int64_t dy_zstride, const float* mask,
                      int64_t ldm, int64_t mask_zstride, float* dW, int64_t dw_zstride, float* db, int64_t db_zstride,
                      void* stream) {
    if (!a || !dY || !dW) return fail(-1, "mvin_linear_wgrad: null pointer");
    if (a->nsrc < 1 || a->nsrc > MVIN_MAX_SRC) return fail(-2, "mvin_linear_wgrad: nsrc=%d", a->nsrc);
    if (a->Dsrc < 4 || (a->Dsrc & 3) || a->Dout < 1 || a->Dout > MVIN_MAX_DIM)
        return fail(-2, "mvin_linear_wgrad: Dsrc=%d Dout=%d", a->Dsrc, a->Dout);
    if ((size_t)a->nsrc * a->Dsrc > 4096) return fail(-2, "mvin_linear_wgrad: nsrc*Dsrc > 4096");
    for (int s = 0; s < a->nsrc; ++s)
        if (!a->src[s]) return fail(-1, "mvin_linear_wgrad: null src[%d]", s);
    if (ldy < a->Dout || (mask && ldm < a->Dout)) return fail(-2, "mvin_linear_wgrad: ldy/ldm < Dout");
    if (a->rows <= 0) return a->rows == 0 ? 0 : fail(-2, "mvin_linear_wgrad: rows < 0");
    mvin::WgradArgs w{};
    w.lin = *a;
    w.dY = dY;
    w.ldy = ldy;
    w.dy_zstride = dy_zstride;
    w.mask = mask;
    w.ldm = ldm;
    w.mask_zstride = mask_zstride;
    w.dW = dW;
    w.dw_zstride = dw_zstride;
    w.db = db;
    w.db_zstride = db_zstride;
    return hip_result(mvin::launch_linear_wgrad(w, (hipStream_t)stream), "mvin_linear_wgrad");
}

static int wgrad_check(const char* who, const mvin_linear_args* a, const float* dY, int64_t ldy, const float* mask, int64_t ldm,
                       const float* dW) {
    if (!a || !dY || !dW) return fail(-1, "%s: null pointer", who);
    if (a->nsrc < 1 || a->nsrc > MVIN_MAX_SRC) return fail(-2, "%s: nsrc=%d", who, a->nsrc);
    if (a->Dsrc < 4 || (a->Dsrc & 3) || a->Dout < 1 || a->Dout > MVIN_MAX_DIM)
        return fail(-2, "%s: Dsrc=%d Dout=%d", who, a->Dsrc, a->Dout);
    if ((size_t)a->nsrc * a->Dsrc > 4096) return fail(-2, "%s: nsrc*Dsrc > 4096", who);
    for (int s = 0; s < a->nsrc; ++s)
        if (!a->src[s]) return fail(-1, "%s: null src[%d]", who, s);
    if (ldy < a->Dout || (mask && ldm < a->Dout)) return fail(-2, "%s: ldy/ldm < Dout", who);
    if (a->rows < 0) return fail(-2, "%s: rows < 0", who);
    return 0;
}

int mvin_linear_wgrad_multi(const mvin_wgrad_problem* problems, int n, void* stream) {
    const char* who = "mvin_linear_wgrad_multi";
    if (n < 0 || n > 64) return fail(-2, "%s: n=%d (0..64)", who, n);
    if (n > 0 && !problems) return fail(-1, "%s: null pointer", who);
    mvin::WgradArgs w[64];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const mvin_wgrad_problem& p = problems[i];
        const int rc = wgrad_check(who, &p.lin, p.dY, p.ldy, p.mask, p.ldm, p.dW);
        if (rc) return rc;
        if (p.lin.rows == 0) continue;
        mvin::WgradArgs& d = w[m++];
        d = mvin::WgradArgs{};
        d.lin = p.lin;
        d.dY = p.dY;
        d.ldy = p.ldy;
        d.dy_zstride = p.dy_zstride;
        d.mask = p.mask;
        d.ldm = p.ldm;
        d.mask_zstride = p.mask_zstride;
        d.dW = p.dW;
        d.dw_zstride = p.dw_zstride;
        d.db = p.db;
        d.db_zstride = p.db_zstride;
    }
    if (m == 0) return 0;
    return hip_result(mvin::launch_linear_wgrad_multi(w, m, (hipStream_t)stream), who);
}

int mvin_agg_bwd(const float* table, const int32_t* adj_entity, const int32_t* adj_relation, const int32_t* node_ids,
                 const float* child, const int32_t* rel_ids, const float* probs, const float* rel_score,
                 const float* dvec, int64_t T, int K, int D, int nR, float* dtable, float* dchild, float* dT,
                 void* stream) {
    const bool gather = table != nullptr;
    if (!dvec) return fail(-1, "mvin_agg_bwd: null dvec");
    if (gather && (!adj_entity || !dtable)) return fail(-1, "mvin_agg_bwd: gather form needs adjacency and dtable");
    if (!gather && (!child || !dchild)) return fail(-1, "mvin_agg_bwd: dense form needs child and dchild");
    if (rel_score && (!gather || probs)) return fail(-2, "mvin_agg_bwd: rel_score is for the gather form without probs");
    if ((probs || rel_score) && (!dT || nR <= 0 || (gather ? !adj_relation : !rel_ids)))
        return fail(-1, "mvin_agg_bwd: attention needs dT, nR and relation ids");
    if (T <= 0 || K <= 0 || K > 4096) return fail(-2, "mvin_agg_bwd: bad sizes T=%lld K=%d", (long long)T, K);
    if (bad_dim(D)) return fail(-2, "mvin_agg_bwd: D=%d", D);
    mvin::AggBwdArgs g{};
    g.gather = gather ? 1 : 0;
    g.table = table;
    g.adj_e = adj_entity;
    g.adj_r = adj_relation;
    g.node_ids = node_ids;
    g.child = child;
    g.rel_ids = rel_ids;
    g.probs = probs;
    g.rel_score = rel_score;
    g.skip_zero = (gather && !node_ids) ? 1 : 0;
    g.dvec = dvec;
    g.dtable = dtable;
    g.dchild = dchild;
    g.dT = (probs || rel_score) ? dT : nullptr;
    g.T = T;
    g.K = K;
    g.D = D;
    g.nR = nR > 0 ? nR : 1;
    g.lpr_log2 = mvin::lpr_log2_for(D);
    return hip_result(mvin::launch_agg_bwd(g, (hipStream_t)stream), "mvin_agg_bwd");
}

int mvin_rel_score_bwd(const float* relation_emb, const float* urh_weights, const float* dT, int nR, int D, float* drel,
                       float* durh, void* stream) {
    if (!relation_emb || !urh_weights || !dT || !drel || !durh) return fail(-1, "mvin_rel_score_bwd: null pointer");
    if (nR <= 0 || D <= 0) return fail(-2, "mvin_rel_score_bwd: bad sizes");
    return hip_result(mvin::launch_rel_score_bwd(relation_emb, urh_weights, dT, nR, D, drel, durh, (hipStream_t)stream),
                      "mvin_rel_score_bwd");
}

int mvin_key_addressing_bwd_adds_item_grad(int P, int Nm, int D, int nR) {
    // mirrors launch_key_addr_bwd: the dV block of a (pair, hop) must fit the kernel's LDS budget next to its tiles
    if (P <= 0 || D > 64 || (D & 3)) return 0;
    const size_t lds = ((size_t)((2 * Nm + 3) & ~3) + 4 * (3 * 256 + 3 * 64)) * sizeof(float);
    return lds + (size_t)nR * D * sizeof(float) <= 48 * 1024 ? 1 : 0;
}

int mvin_key_addressing_bwd(const float* entity_emb, const float* V, const float* w, const int32_t* const* mem_h,
                            const int32_t* const* mem_r, const int32_t* const* mem_t, int P, int B, int Nm, int D,
                            int nR, const float* dout, int64_t ldo, float l2, float* dE, float* dV, float* dw,
                            void* stream) {
    return mvin_key_addressing_bwd_reg(entity_emb, V, w, mem_h, mem_r, mem_t, P, B, Nm, D, nR, dout, ldo, l2, dE, dV,
                                       dw, 1, nullptr, nullptr, nullptr, 0, stream);
}

int mvin_key_addressing_bwd_reg(const float* entity_emb, const float* V, const float* w,
                                const int32_t* const* mem_h, const int32_t* const* mem_r,
                                const int32_t* const* mem_t, int P, int B, int Nm, int D, int nR, const float* dout,
                                int64_t ldo, float l2, float* dE, float* dV, float* dw, int dw_replicas,
                                float* reg_accum, const float* relation_kge, const void* items, int items64,
                                void* stream) {
    const char* who = "mvin_key_addressing_bwd";
    if (dw_replicas < 1 || dw_replicas > 1024 || (dw_replicas & (dw_replicas - 1)))
        return fail(-2, "%s: dw_replicas=%d (a power of two in 1..1024)", who, dw_replicas);
    if (!entity_emb || !mem_h || !dout || !dE) return fail(-1, "%s: null pointer", who);
    if (P < 0 || P > 8 || (P == 0 && !w)) return fail(-2, "%s: P=%d", who, P);
    if (P > 0 && (!V || !mem_r || !mem_t || !dV || nR <= 0)) return fail(-1, "%s: hops need V, mem_r, mem_t, dV, nR", who);
    if (w && !dw) return fail(-1, "%s: w given without dw", who);
    if (B <= 0 || Nm <= 0 || Nm > 8192) return fail(-2, "%s: bad sizes B=%d Nm=%d", who, B, Nm);
    if (bad_dim(D)) return fail(-2, "%s: D=%d", who, D);
    mvin::KeyAddrBwdArgs k{};
    k.f.E = entity_emb;
    k.f.V = V;
    k.f.w = w;
    const int nh = P > 0 ? P : 1;
    for (int i = 0; i < nh; ++i) {
        if (!mem_h[i]) return fail(-1, "%s: null mem_h[%d]", who, i);
        k.f.mem_h[i] = mem_h[i];
        if (i < P) {
            if (!mem_r[i] || !mem_t[i]) return fail(-1, "%s: null mem_r/mem_t[%d]", who, i);
            k.f.mem_r[i] = mem_r[i];
            k.f.mem_t[i] = mem_t[i];
        }
    }
    k.f.ldo = ldo;
    k.f.B = B;
    k.f.P = P;
    k.f.Nm = Nm;
    k.f.D = D;
    k.f.nR = nR;
    k.f.lpr_log2 = mvin::lpr_log2_for(D);
    k.dout = dout;
    k.dE = dE;
    k.dV = dV;
    k.dw = dw;
    k.l2 = l2;
    k.reg_accum = reg_accum;
    k.dw_rep = dw_replicas;
    if ((relation_kge == nullptr) != (items == nullptr)) return fail(-1, "%s: relation_kge and items go together", who);
    if (relation_kge && !mvin_key_addressing_bwd_adds_item_grad(P, Nm, D, nR))
        return fail(-3, "%s: the item gradient is not added in-kernel at this shape (query "
                        "mvin_key_addressing_bwd_adds_item_grad)", who);
    k.Rk = relation_kge;
    k.items = items;
    k.items64 = items64;
    return hip_result(mvin::launch_key_addr_bwd(k, (hipStream_t)stream), who);
}

}  // extern "C"
