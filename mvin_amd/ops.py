"""Torch-tensor front end of the C ABI (include/mvin_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; all arithmetic of the
path runs in libmvin_hip.so.  Every op requires CUDA(ROCm) tensors and raises otherwise --
there is no CPU / eager fallback.
"""
import ctypes as C
import os

import torch

from . import _lib

F32 = torch.float32
I32 = torch.int32
BF16 = torch.bfloat16


def _chk_table(t, name):
    """Entity-style table: fp32 or bf16 (bf16 rows are widened to fp32 inside the kernels)."""
    if t is None:
        return 0
    _chk(t, BF16 if t.dtype == BF16 else F32, name)
    return 1 if t.dtype == BF16 else 0


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MvinHipError(f"{name}: expected a CUDA/ROCm tensor (mvin_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _p(t, offset_elems=0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())


def ent_level_offsets(B, K, levels):
    """Element offsets of entity levels 0..levels and relation levels 0..levels-1 in the
    flat buffers written by mvin_expand_ids (see include/mvin_hip.h)."""
    ent, rel, off_e, off_r, n = [], [], 0, 0, B
    for e in range(levels + 1):
        ent.append((off_e, n))
        off_e += n
        if e < levels:
            n *= K
            rel.append((off_r, n))
            off_r += n
    return ent, rel


def expand_ids(adj_entity, adj_relation, items, K, levels, n_entity):
    """MVIN.get_neighbors (model.py:243-256).  Returns (entities, relations): lists of int32
    tensors [B, K^e] (e = 0..levels) and [B, K^(e+1)] (e = 0..levels-1), views of two flat
    buffers."""
    lib = _lib.load()
    _chk(adj_entity, I32, "adj_entity")
    _chk(adj_relation, I32, "adj_relation")
    if items.dtype == torch.int64:
        i64, i32 = _chk(items, torch.int64, "items"), None
    else:
        i64, i32 = None, _chk(items, I32, "items")
    B = items.shape[0]
    ent_flat = torch.empty(lib.mvin_ent_elems(B, K, levels), dtype=I32, device=items.device)
    rel_flat = torch.empty(max(1, lib.mvin_rel_elems(B, K, levels)), dtype=I32, device=items.device)
    _lib.check(lib.mvin_expand_ids(_p(adj_entity), _p(adj_relation), _p(i64), _p(i32), B, K, levels,
                                   n_entity, _p(ent_flat), _p(rel_flat), _stream()), "mvin_expand_ids")
    eo, ro = ent_level_offsets(B, K, levels)
    ents = [ent_flat[o:o + n].view(B, -1) for o, n in eo]
    rels = [rel_flat[o:o + n].view(B, -1) for o, n in ro]
    return ents, rels


def rel_score(relation_emb, urh_weights):
    """t[r] = relation_emb[r] . urh_weights[D:2D]  (aggregators.py:130-133, k-dependent term)."""
    lib = _lib.load()
    _chk(relation_emb, F32, "relation_emb")
    _chk(urh_weights, F32, "urh_weights")
    nR, D = relation_emb.shape
    t = torch.empty(nR, dtype=F32, device=relation_emb.device)
    _lib.check(lib.mvin_rel_score(_p(relation_emb), _p(urh_weights), nR, D, _p(t), _stream()),
               "mvin_rel_score")
    return t


def linear(srcs, W, Dout, *, ids=None, bias=None, rowbias=None, rows_per_group=1, relu=False,
           rows=None, out=None, out_offset=0, ldo=None, nz=1, w_zstride=0, bias_zstride=0,
           out_zstride=0, score_u=None, sum_sources=False):
    """mvin_linear_fwd: out[z][r] = act(concat_s X_s[r] . W[z] + bias[z] + rowbias[r // rpg]).
    ``srcs``: list of [*, Dsrc] fp32 tensors; ``ids``: matching list of int32 row-id tensors
    or None.  Returns out, or (out, score, sigmoid) when ``score_u`` is given."""
    lib = _lib.load()
    a = _lib.LinearArgs()
    nsrc = len(srcs)
    ids = ids or [None] * nsrc
    ids64 = None
    Dsrc = srcs[0].shape[-1]
    src_bf16 = 0
    for s in range(nsrc):
        src_bf16 |= _chk_table(srcs[s], f"src[{s}]") << s
        if srcs[s].shape[-1] != Dsrc:
            raise ValueError("all sources must share the row width")
        a.src[s] = srcs[s].data_ptr()
        if ids[s] is not None:
            if ids64 is None:
                ids64 = ids[s].dtype == torch.int64
            _chk(ids[s], torch.int64 if ids64 else I32, f"ids[{s}]")
            a.ids[s] = ids[s].data_ptr()
    if rows is None:
        first = next((i for i in ids if i is not None), None)
        rows = first.numel() if first is not None else srcs[0].numel() // Dsrc
    dev = srcs[0].device
    if out is None:
        ldo = ldo or Dout
        out = torch.empty((nz, rows, Dout) if nz > 1 else (rows, Dout), dtype=F32, device=dev)
        if nz > 1 and out_zstride == 0:
            out_zstride = rows * Dout
    else:
        _chk(out, F32, "out")
        ldo = ldo or Dout
    a.nsrc, a.Dsrc, a.Dout, a.rows = nsrc, Dsrc, Dout, rows
    a.W = _chk(W, F32, "W").data_ptr() if W is not None else None
    a.bias = _chk(bias, F32, "bias").data_ptr() if bias is not None else None
    a.rowbias = _chk(rowbias, F32, "rowbias").data_ptr() if rowbias is not None else None
    a.rows_per_group = rows_per_group
    a.relu = 1 if relu else 0
    a.ids64 = 1 if ids64 else 0
    a.src_bf16 = src_bf16
    gathered = [srcs[s].numel() // Dsrc for s in range(nsrc) if ids[s] is not None]
    if len(set(gathered)) > 1:
        # mvin_linear_args carries ONE src_rows: ids valid for the larger table would be clamped to the smaller one's last row
        raise ValueError(f"gathered sources of one mvin_linear_fwd call must have the same row count, got {gathered}")
    a.src_rows = gathered[0] if gathered else 0           # ids are clamped into the gathered table
    a.sum_sources = 1 if sum_sources else 0
    a.out = out.data_ptr() + out_offset * 4
    a.ldo = ldo
    a.nz, a.w_zstride, a.bias_zstride, a.out_zstride = nz, w_zstride, bias_zstride, out_zstride
    score = sig = None
    if score_u is not None:
        _chk(score_u, F32, "score_u")
        score = torch.empty(rows, dtype=F32, device=dev)
        sig = torch.empty(rows, dtype=F32, device=dev)
        a.score_u, a.score_out, a.sigmoid_out = score_u.data_ptr(), score.data_ptr(), sig.data_ptr()
    _lib.check(lib.mvin_linear_fwd(C.byref(a), _stream()), "mvin_linear_fwd")
    return (out, score, sig) if score_u is not None else out


def score_small_supported(D, K, P, Nm, nR):
    """mvin_score_small_supported: the whole depth-2 pass as one launch exists for this shape."""
    return bool(_lib.load().mvin_score_small_supported(int(D), int(K), int(P), int(Nm), int(nR)))


def gather_attn(table, adj_entity, adj_relation, node_ids, rel_score_t, self_vec, Wc, c_child,
                Wagg, bagg, B, N, K, D, want_probs=False):
    """mvin_gather_attn_fwd(_ex): deepest hop, children gathered from ``table`` (fp32 or bf16)
    through the adjacency of ``node_ids`` [B*N]; returns (out [B,N,D], probs [B,N,K] or None)."""
    lib = _lib.load()
    bf = _chk_table(table, "table")
    for t, dt, nm in ((adj_entity, I32, "adj_entity"),
                      (adj_relation, I32, "adj_relation"), (node_ids, I32, "node_ids"),
                      (rel_score_t, F32, "rel_score"), (self_vec, F32, "self_vec"), (Wc, F32, "Wc"),
                      (c_child, F32, "c_child"), (Wagg, F32, "Wagg"), (bagg, F32, "bagg")):
        _chk(t, dt, nm)
    out = torch.empty((B, N, D), dtype=F32, device=table.device)
    probs = torch.empty((B, N, K), dtype=F32, device=table.device) if want_probs else None
    _lib.check(lib.mvin_gather_attn_fwd_ex(_p(table), _p(adj_entity), _p(adj_relation), _p(node_ids),
                                           _p(rel_score_t), _p(self_vec), _p(Wc), _p(c_child), _p(Wagg),
                                           _p(bagg), B, N, K, D, table.shape[0], _p(out), _p(probs), None, None,
                                           bf, _stream()), "mvin_gather_attn_fwd_ex")
    return out, probs


def agg(self_vec, neigh, rel_ids, rel_score_t, Wagg, bagg, B, N, K, D, want_probs=False):
    """mvin_agg_fwd on materialised levels; ``rel_ids`` None with ``rel_score_t`` given means
    ``rel_score_t`` already holds one logit per child ([B*N*K])."""
    lib = _lib.load()
    for t, dt, nm in ((self_vec, F32, "self_vec"), (neigh, F32, "neigh"), (rel_ids, I32, "rel_ids"),
                      (rel_score_t, F32, "rel_score"), (Wagg, F32, "Wagg"), (bagg, F32, "bagg")):
        _chk(t, dt, nm)
    out = torch.empty((B, N, D), dtype=F32, device=self_vec.device)
    probs = torch.empty((B, N, K), dtype=F32, device=self_vec.device) if want_probs else None
    _lib.check(lib.mvin_agg_fwd(_p(self_vec), _p(neigh), _p(rel_ids), _p(rel_score_t), _p(Wagg),
                                _p(bagg), B, N, K, D, _p(out), _p(probs), _stream()), "mvin_agg_fwd")
    return out, probs


def ripple_attn(entity_emb, score_ids, rel_ids, value_ids, V, w, mode, out, out_offset, ldo, nR):
    """mvin_ripple_attn_fwd_ex: one ripple-set attention read per pair (fp32 or bf16 table), written into
    ``out`` (a [B, ldo] buffer) at column offset ``out_offset``."""
    lib = _lib.load()
    bf = _chk_table(entity_emb, "entity_emb")
    for t, dt, nm in ((score_ids, I32, "score_ids"),
                      (rel_ids, I32, "rel_ids"), (value_ids, I32, "value_ids"), (V, F32, "V"),
                      (w, F32, "w"), (out, F32, "out")):
        _chk(t, dt, nm)
    B, Nm = score_ids.shape
    D = entity_emb.shape[1]
    _lib.check(lib.mvin_ripple_attn_fwd_ex(_p(entity_emb), _p(score_ids), _p(rel_ids), _p(value_ids),
                                           _p(V), _p(w), mode, B, Nm, D, nR, _p(out, out_offset), ldo, bf,
                                           _stream()), "mvin_ripple_attn_fwd")
    return out


def gather_attn_l2_supported(D, K):
    return bool(_lib.load().mvin_gather_attn_l2_supported(D, K))


def probe_gather_l2(table, child_ids, grandchild_ids, K, sums=None):
    """mvin_probe_gather_l2: read the K child rows and K*K grandchild rows of every parent (id lists = levels 1 and 2
    of expand_ids) and add them up; returns sums [n_parents] (measurement aid, include/mvin_hip.h)."""
    bf = _chk_table(table, "table")
    _chk(child_ids, I32, "child_ids"), _chk(grandchild_ids, I32, "grandchild_ids")
    n = child_ids.numel() // K
    if grandchild_ids.numel() != n * K * K:
        raise ValueError("grandchild_ids must hold K*K ids per parent")
    if sums is None:
        sums = torch.empty(n, dtype=F32, device=table.device)
    else:
        _chk(sums, F32, "sums")
    _lib.check(_lib.load().mvin_probe_gather_l2(_p(table), _p(child_ids), _p(grandchild_ids), n, K, table.shape[1],
                                                table.shape[0], bf, _p(sums), _stream()), "mvin_probe_gather_l2")
    return sums


def gather_attn_l2_prj_supported(D, K, encoded, n_entity, n_relation):
    """Does mvin_gather_attn_l2_prj_fwd take these tables (encoded: the packed-tile kernel; plain: D = 32, K in {8, 16})?"""
    return bool(_lib.load().mvin_gather_attn_l2_prj_supported(D, K, int(bool(encoded)), n_entity, n_relation))


def gather_attn_l2_variant(D, K, n_parents, n_entity, want_probs=False, table_bf16=False):
    """0 = unsupported, 1 = symmetric fused kernel, 2 = role-split pipeline, 3 / 4 = wave-per-parent kernels (include/mvin_hip.h)."""
    return int(_lib.load().mvin_gather_attn_l2_variant_ex(D, K, n_parents, n_entity, int(bool(want_probs)), int(bool(table_bf16))))


def gather_attn_l2(table, adj_entity, adj_relation, parent_ids, t0, t1, W1, W2, b1, b2, q, A0, a0,
                   B, parents_per_pair, K, D, nR, want_probs=False):
    """mvin_gather_attn_l2_fwd: the two deepest levels in one pass.  ``parent_ids``: int32, or int64 read in place
    (mvin_gather_attn_l2_fwd_i64: the batch's item ids at tree depth 2).  Returns
    (nagg0 [P,D], nagg1 [P,D], probs_parent [P,K] | None, probs_child [P*K,K] | None)."""
    lib = _lib.load()
    bf = _chk_table(table, "table")
    for t, dt, nm in ((adj_entity, I32, "adj_entity"), (adj_relation, I32, "adj_relation"),
                      (parent_ids, torch.int64 if parent_ids.dtype == torch.int64 else I32, "parent_ids"), (t0, F32, "t0"), (t1, F32, "t1"), (W1, F32, "W1"),
                      (W2, F32, "W2"), (b1, F32, "b1"), (b2, F32, "b2"), (q, F32, "q"), (A0, F32, "A0"),
                      (a0, F32, "a0")):
        _chk(t, dt, nm)
    P = B * parents_per_pair
    dev = table.device
    nagg0 = torch.empty((P, D), dtype=F32, device=dev)
    nagg1 = torch.empty((P, D), dtype=F32, device=dev)
    pp = torch.empty((P, K), dtype=F32, device=dev) if want_probs else None
    pc = torch.empty((P * K, K), dtype=F32, device=dev) if want_probs else None
    fn = lib.mvin_gather_attn_l2_fwd_i64 if parent_ids.dtype == torch.int64 else lib.mvin_gather_attn_l2_fwd
    _lib.check(fn(_p(table), _p(adj_entity), _p(adj_relation), _p(parent_ids), _p(t0),
                  _p(t1), _p(W1), _p(W2), _p(b1), _p(b2), _p(q), _p(A0), _p(a0), B,
                  parents_per_pair, K, D, table.shape[0], nR, _p(nagg0), _p(nagg1),
                  _p(pp), _p(pc), bf, _stream()), "mvin_gather_attn_l2_fwd")
    return nagg0, nagg1, pp, pc


def encode_adjacency_supported(D, K):
    """The packed-tile fused kernel (mvin_gather_attn_l2_enc_fwd) exists for this shape."""
    return bool(_lib.load().mvin_gather_attn_l2_enc_supported(D, K))


def encode_adjacency(adj_entity, adj_relation):
    """mvin_encode_adjacency: the duplicate-slot encoding of a sampled adjacency (include/mvin_hip.h) ->
    (enc_entity [nE,K] int32, enc_relation [nE,K] int32, cnt [nE] int32 = distinct slots per row)."""
    _chk(adj_entity, I32, "adj_entity"), _chk(adj_relation, I32, "adj_relation")
    nE, K = adj_entity.shape
    enc_e, enc_r = torch.empty_like(adj_entity), torch.empty_like(adj_entity)
    cnt = torch.empty(nE, dtype=I32, device=adj_entity.device)
    _lib.check(_lib.load().mvin_encode_adjacency(_p(adj_entity), _p(adj_relation), nE, K, _p(cnt), _p(enc_e), _p(enc_r),
                                                 _stream()), "mvin_encode_adjacency")
    return enc_e, enc_r, cnt


def gather_attn_l2_enc(table, enc_entity, enc_relation, parent_ids, t0, t1, W1, W2, b1, b2, q, A0, a0,
                       B, parents_per_pair, K, D, nR):
    """mvin_gather_attn_l2_enc_fwd: gather_attn_l2 over the duplicate-slot encoding (packed tiles; no attention
    outputs).  Returns (nagg0 [P,D], nagg1 [P,D])."""
    lib = _lib.load()
    bf = _chk_table(table, "table")
    for t, dt, nm in ((enc_entity, I32, "enc_entity"), (enc_relation, I32, "enc_relation"),
                      (parent_ids, torch.int64 if parent_ids.dtype == torch.int64 else I32, "parent_ids"), (t0, F32, "t0"),
                      (t1, F32, "t1"), (W1, F32, "W1"), (W2, F32, "W2"), (b1, F32, "b1"), (b2, F32, "b2"), (q, F32, "q"),
                      (A0, F32, "A0"), (a0, F32, "a0")):
        _chk(t, dt, nm)
    P = B * parents_per_pair
    nagg0 = torch.empty((P, D), dtype=F32, device=table.device)
    nagg1 = torch.empty((P, D), dtype=F32, device=table.device)
    _lib.check(lib.mvin_gather_attn_l2_enc_fwd(_p(table), _p(enc_entity), _p(enc_relation), _p(parent_ids),
                                               int(parent_ids.dtype == torch.int64), _p(t0), _p(t1), _p(W1), _p(W2), _p(b1),
                                               _p(b2), _p(q), _p(A0), _p(a0), B, parents_per_pair, K, D, table.shape[0], nR,
                                               _p(nagg0), _p(nagg1), bf, _stream()), "mvin_gather_attn_l2_enc_fwd")
    return nagg0, nagg1


def project_rows(src, W1, W2, b1=None, b2=None):
    """mvin_project_rows: [2, rows, D] = (src . W1 (+ b1), src . W2 (+ b2)) -- the two projections of the levels the fused
    two-level kernel gathers, of the entity table's rows or of the pairs' query vectors."""
    lib = _lib.load()
    for t, nm in ((src, "src"), (W1, "W1"), (W2, "W2"), (b1, "b1"), (b2, "b2")):
        _chk(t, F32, nm)
    rows, D = src.shape
    out = torch.empty((2, rows, D), dtype=F32, device=src.device)
    _lib.check(lib.mvin_project_rows(_p(src), rows, D, _p(W1), _p(W2), _p(b1), _p(b2), _p(out), _stream()), "mvin_project_rows")
    return out


def project_tables(entity_emb, W1, W2, b1, b2, A0, a0, K, attention, out=None):
    """mvin_project_tables: the workspace of the projected-tables form -- E.W1 | E.W1.A0 | E.W2.A0 and the per-call parameter
    block -- from the CURRENT parameters.  ``attention``: whether the relation logits t0 will be given to the gather."""
    lib = _lib.load()
    for t, nm in ((entity_emb, "entity_emb"), (W1, "W1"), (W2, "W2"), (b1, "b1"), (b2, "b2"), (A0, "A0"), (a0, "a0")):
        _chk(t, F32, nm)
    nE, D = entity_emb.shape
    n = lib.mvin_project_tables_elems(nE, D)
    if out is None:
        out = torch.empty((n,), dtype=F32, device=entity_emb.device)
    elif out.numel() != n or out.dtype != F32 or not out.is_contiguous():
        raise ValueError("project_tables: workspace of mvin_project_tables_elems floats expected")
    _lib.check(lib.mvin_project_tables(_p(entity_emb), _p(W1), _p(W2), _p(b1), _p(b2), _p(A0), _p(a0), 1 if attention else 0, K, nE, D,
                                       _p(out), _stream()), "mvin_project_tables")
    return out


def gather_attn_l2_wpp_supported(D, K):
    """The shapes the wave-per-parent kernel over projected tables takes (it is the one that honours ``order``)."""
    return D == 64 and K in (16, 32) and os.environ.get("MVIN_L2_WPP", "1") != "0"


def order_by_key(keys, ws=None, out=None):
    """mvin_order_by_key: a permutation of 0 .. B-1 (int32) in which equal keys (int64 / int32 ids) are neighbours -- a partition by
    the key's low bits, not a sort; the order inside a bucket is unspecified."""
    lib = _lib.load()
    B = keys.shape[0]
    _chk(keys, torch.int64 if keys.dtype == torch.int64 else I32, "keys")
    n = lib.mvin_order_by_key_ws_elems(B)
    if ws is None or ws.numel() < n:
        ws = torch.empty((n,), dtype=I32, device=keys.device)
    if out is None:
        out = torch.empty((B,), dtype=I32, device=keys.device)
    k64, k32 = (_p(keys), None) if keys.dtype == torch.int64 else (None, _p(keys))
    _lib.check(lib.mvin_order_by_key(k64, k32, B, _p(ws), _p(out), _stream()), "mvin_order_by_key")
    return out


def gather_attn_l2_prj(ws, enc_entity, enc_relation, parent_ids, t0, t1, q, B, parents_per_pair, K, D, nR, n_entity, encoded=True, order=None):
    """mvin_gather_attn_l2_prj_fwd: gather_attn_l2_enc over the workspace of ``project_tables`` (built with attention =
    (t0 is not None)).  ``encoded=False``: the two adjacency arrays are the plain adjacency (D = 32, K in {8, 16}).
    Returns (nagg0 [P,D], nagg1 [P,D])."""
    lib = _lib.load()
    for t, dt, nm in ((ws, F32, "ws"), (enc_entity, I32, "enc_entity"), (enc_relation, I32, "enc_relation"),
                      (parent_ids, torch.int64 if parent_ids.dtype == torch.int64 else I32, "parent_ids"), (t0, F32, "t0"),
                      (t1, F32, "t1"), (q, F32, "q")):
        _chk(t, dt, nm)
    if ws.numel() != lib.mvin_project_tables_elems(n_entity, D) or q is None or tuple(q.shape) != (B, D):
        raise ValueError("gather_attn_l2_prj: the workspace of project_tables(n_entity, D) and q [B, D] expected")
    P = B * parents_per_pair
    nagg0 = torch.empty((P, D), dtype=F32, device=ws.device)
    nagg1 = torch.empty((P, D), dtype=F32, device=ws.device)
    _chk(order, I32, "order")
    if order is not None and order.numel() != P:
        raise ValueError("gather_attn_l2_prj: order must be a permutation of the launch's parents")
    _lib.check(lib.mvin_gather_attn_l2_prj_ordered_fwd(_p(ws), _p(enc_entity), _p(enc_relation), 1 if encoded else 0, _p(parent_ids),
                                                       int(parent_ids.dtype == torch.int64), _p(order), _p(t0), _p(t1), _p(q), B,
                                                       parents_per_pair, K, D, n_entity, nR, _p(nagg0), _p(nagg1), _stream()),
               "mvin_gather_attn_l2_prj_fwd")
    return nagg0, nagg1


def gather_attn_l2_agg_supported(D, K, n_entity, nR):
    """mvin_gather_attn_l2_agg_supported: does the per-entity aggregates form take these tables?  (D = 64, K in {16, 32, 64}.)"""
    return bool(_lib.load().mvin_gather_attn_l2_agg_supported(D, K, n_entity, nR))


def entity_aggregates(ws, enc_entity, enc_relation, t0, K, D, nR, n_entity, out=None):
    """mvin_entity_aggregates: S0 | G ([2, n_entity, D] fp32) from the workspace of ``project_tables`` (the CURRENT call's), the
    encoded adjacency and the relation logits t0 of aggregator (0,.) (None: plain mean)."""
    lib = _lib.load()
    for t, dt, nm in ((ws, F32, "ws"), (enc_entity, I32, "enc_entity"), (enc_relation, I32, "enc_relation"), (t0, F32, "t0")):
        _chk(t, dt, nm)
    if ws.numel() != lib.mvin_project_tables_elems(n_entity, D):
        raise ValueError("entity_aggregates: the workspace of project_tables(n_entity, D) expected")
    n = lib.mvin_entity_aggregates_elems(n_entity, D)
    if out is None:
        out = torch.empty((n,), dtype=F32, device=ws.device)
    elif out.numel() != n or out.dtype != F32 or not out.is_contiguous():
        raise ValueError("entity_aggregates: workspace of mvin_entity_aggregates_elems floats expected")
    _lib.check(lib.mvin_entity_aggregates(_p(ws), _p(enc_entity), _p(enc_relation), _p(t0), K, D, n_entity, nR, _p(out), _stream()),
               "mvin_entity_aggregates")
    return out


def gather_attn_l2_agg(ws, agg, enc_entity, enc_relation, parent_ids, t0, t1, q, B, parents_per_pair, K, D, nR, n_entity, order=None):
    """mvin_gather_attn_l2_agg_fwd: gather_attn_l2_prj with the per-entity aggregates of ``entity_aggregates`` beside the
    workspace.  Returns (nagg0 [P,D], nagg1 [P,D])."""
    lib = _lib.load()
    for t, dt, nm in ((ws, F32, "ws"), (agg, F32, "agg"), (enc_entity, I32, "enc_entity"), (enc_relation, I32, "enc_relation"),
                      (parent_ids, torch.int64 if parent_ids.dtype == torch.int64 else I32, "parent_ids"), (t0, F32, "t0"),
                      (t1, F32, "t1"), (q, F32, "q"), (order, I32, "order")):
        _chk(t, dt, nm)
    if (ws.numel() != lib.mvin_project_tables_elems(n_entity, D) or agg.numel() != lib.mvin_entity_aggregates_elems(n_entity, D)
            or q is None or tuple(q.shape) != (B, D)):
        raise ValueError("gather_attn_l2_agg: the workspaces of project_tables / entity_aggregates (n_entity, D) and q [B, D] expected")
    P = B * parents_per_pair
    if order is not None and order.numel() != P:
        raise ValueError("gather_attn_l2_agg: order must be a permutation of the launch's parents")
    nagg0 = torch.empty((P, D), dtype=F32, device=ws.device)
    nagg1 = torch.empty((P, D), dtype=F32, device=ws.device)
    _lib.check(lib.mvin_gather_attn_l2_agg_fwd(_p(ws), _p(agg), _p(enc_entity), _p(enc_relation), _p(parent_ids),
                                               int(parent_ids.dtype == torch.int64), _p(order), _p(t0), _p(t1), _p(q), B,
                                               parents_per_pair, K, D, n_entity, nR, _p(nagg0), _p(nagg1), _stream()),
               "mvin_gather_attn_l2_agg_fwd")
    return nagg0, nagg1


def score_l2_folded_supported(D, K, n_entity, nR):
    """mvin_score_l2_folded_supported: does the folded-tail form take these tables?  (D = 64, K in {16, 32, 64}.)"""
    return bool(_lib.load().mvin_score_l2_folded_supported(D, K, n_entity, nR))


def fold_tables(entity_emb, enc_entity, enc_relation, t0, W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, K, nR, out=None, aggregates=True):
    """mvin_fold_tables: the workspace of the folded-tail form -- TA1 | TA2 | T0A | M0, the aggregates H0 | G and the parameter
    block -- from the CURRENT parameters, the encoded adjacency and the relation logits t0 of aggregator (0,.) (None: plain mean)."""
    lib = _lib.load()
    for t, nm in ((entity_emb, "entity_emb"), (t0, "t0"), (W0, "W0"), (b0, "b0"), (W1, "W1"), (b1, "b1"), (W2, "W2"), (b2, "b2"),
                  (A0, "A0"), (a0, "a0"), (Wmix, "Wmix"), (bmix, "bmix"), (A1, "A1")):
        _chk(t, F32, nm)
    _chk(enc_entity, I32, "enc_entity")
    _chk(enc_relation, I32, "enc_relation")
    nE, D = entity_emb.shape
    n = lib.mvin_fold_tables_elems(nE, D)
    if out is None:
        out = torch.empty((n,), dtype=F32, device=entity_emb.device)
    elif out.numel() != n or out.dtype != F32 or not out.is_contiguous():
        raise ValueError("fold_tables: workspace of mvin_fold_tables_elems floats expected")
    # (aggregates=False: the four per-row tables only -- the workspace of score_l2_folded_gather, where every pair gathers its own rows)
    _lib.check(lib.mvin_fold_tables_ex(_p(entity_emb), _p(enc_entity), _p(enc_relation), _p(t0), _p(W0), _p(b0), _p(W1), _p(b1), _p(W2), _p(b2),
                                       _p(A0), _p(a0), _p(Wmix), _p(bmix), _p(A1), 1 if aggregates else 0, K, D, nE, nR, _p(out), _stream()),
               "mvin_fold_tables")
    return out


def score_l2_folded_gather_supported(D, K, n_entity, nR):
    """mvin_score_l2_folded_gather_supported: the folded tail with every pair gathering its own rows (D = 64, K in {16, 32})."""
    return bool(_lib.load().mvin_score_l2_folded_gather_supported(D, K, n_entity, nR))


def score_l2_folded_gather(ws, enc_entity, enc_relation, items, t0, t1, q, user_o, A1, a1, Wmix, K, D, nR, n_entity, order=None,
                           want_item_emb=True):
    """mvin_score_l2_folded_gather_fwd over the workspace of ``fold_tables(..., aggregates=False)``: (item_emb [B,D] or None, scores [B],
    sigmoid(scores) [B]).  ``order``: a permutation of the pairs (order_by_key over the items) or None."""
    lib = _lib.load()
    for t, dt, nm in ((ws, F32, "ws"), (enc_entity, I32, "enc_entity"), (enc_relation, I32, "enc_relation"),
                      (items, torch.int64 if items.dtype == torch.int64 else I32, "items"), (t0, F32, "t0"), (t1, F32, "t1"), (q, F32, "q"),
                      (user_o, F32, "user_o"), (A1, F32, "A1"), (a1, F32, "a1"), (Wmix, F32, "Wmix"), (order, I32, "order")):
        _chk(t, dt, nm)
    B = items.shape[0]
    if ws.numel() != lib.mvin_fold_tables_elems(n_entity, D) or tuple(q.shape) != (B, D) or tuple(user_o.shape) != (B, D):
        raise ValueError("score_l2_folded_gather: the workspace of fold_tables(n_entity, D) and q, user_o [B, D] expected")
    if order is not None and order.numel() != B:
        raise ValueError("score_l2_folded_gather: order must be a permutation of the pairs")
    item_emb = torch.empty((B, D), dtype=F32, device=ws.device) if want_item_emb else None
    scores = torch.empty((B,), dtype=F32, device=ws.device)
    sig = torch.empty((B,), dtype=F32, device=ws.device)
    i64 = items.dtype == torch.int64
    _lib.check(lib.mvin_score_l2_folded_gather_fwd(_p(ws), _p(enc_entity), _p(enc_relation), _p(items) if i64 else None,
                                                   None if i64 else _p(items), _p(order), _p(t0), _p(t1), _p(q), _p(user_o), _p(A1), _p(a1),
                                                   _p(Wmix), B, K, D, n_entity, nR, _p(item_emb), _p(scores), _p(sig), _stream()),
               "mvin_score_l2_folded_gather_fwd")
    return item_emb, scores, sig


def score_l2_folded(ws, enc_entity, enc_relation, items, t0, t1, q, user_o, A1, a1, Wmix, K, D, nR, n_entity, want_item_emb=True):
    """mvin_score_l2_folded_fwd over the workspace of ``fold_tables``: (item_emb [B,D] or None, scores [B], sigmoid(scores) [B])."""
    lib = _lib.load()
    for t, dt, nm in ((ws, F32, "ws"), (enc_entity, I32, "enc_entity"), (enc_relation, I32, "enc_relation"),
                      (items, torch.int64 if items.dtype == torch.int64 else I32, "items"), (t0, F32, "t0"), (t1, F32, "t1"), (q, F32, "q"),
                      (user_o, F32, "user_o"), (A1, F32, "A1"), (a1, F32, "a1"), (Wmix, F32, "Wmix")):
        _chk(t, dt, nm)
    B = items.shape[0]
    if ws.numel() != lib.mvin_fold_tables_elems(n_entity, D) or tuple(q.shape) != (B, D) or tuple(user_o.shape) != (B, D):
        raise ValueError("score_l2_folded: the workspace of fold_tables(n_entity, D) and q, user_o [B, D] expected")
    two = os.environ.get("MVIN_L2_FOLD_TWO", "0") not in ("", "0")      # (the two-launch A/B variant needs scratch rows for out0 / Z2)
    out0 = torch.empty((B, D), dtype=F32, device=ws.device) if two else None
    z2 = torch.empty((B, D), dtype=F32, device=ws.device) if two else None
    item_emb = torch.empty((B, D), dtype=F32, device=ws.device) if want_item_emb else None
    scores = torch.empty((B,), dtype=F32, device=ws.device)
    sig = torch.empty((B,), dtype=F32, device=ws.device)
    i64 = items.dtype == torch.int64
    _lib.check(lib.mvin_score_l2_folded_fwd(_p(ws), _p(enc_entity), _p(enc_relation), _p(items) if i64 else None, None if i64 else _p(items),
                                            _p(t0), _p(t1), _p(q), _p(user_o), _p(A1), _p(a1), _p(Wmix), B, K, D, n_entity, nR, _p(out0),
                                            _p(z2), _p(item_emb), _p(scores), _p(sig), _stream()), "mvin_score_l2_folded_fwd")
    return item_emb, scores, sig


def gather_mix(table, adj_entity, adj_relation, node_ids, rel_score_t, rowbias, nodes, nodes_per_group, K, nR,
               relu=False):
    """mvin_gather_mix_fwd: out[i] = (1/K) sum_k w_k f(table[adj_entity[x_i,k]] + rowbias[i // npg]) ->
    [nodes, D] fp32 (x_i = node_ids[i], or i when node_ids is None)."""
    lib = _lib.load()
    bf = _chk_table(table, "table")
    for t, dt, nm in ((adj_entity, I32, "adj_entity"), (adj_relation, I32, "adj_relation"),
                      (node_ids, I32, "node_ids"), (rel_score_t, F32, "rel_score"), (rowbias, F32, "rowbias")):
        _chk(t, dt, nm)
    D = table.shape[1]
    out = torch.empty((nodes, D), dtype=F32, device=table.device)
    _lib.check(lib.mvin_gather_mix_fwd(_p(table), _p(adj_entity), _p(adj_relation), _p(node_ids), _p(rel_score_t),
                                       _p(rowbias), nodes, nodes_per_group, K, D, table.shape[0], nR,
                                       1 if relu else 0, _p(out), bf, _stream()), "mvin_gather_mix_fwd")
    return out


def row_softmax(x):
    """mvin_row_softmax_fwd: softmax over the last axis of a [rows, n] fp32 tensor."""
    lib = _lib.load()
    _chk(x, F32, "x")
    out = torch.empty_like(x)
    _lib.check(lib.mvin_row_softmax_fwd(_p(x), x.shape[0], x.shape[1], _p(out), _stream()), "mvin_row_softmax_fwd")
    return out


def mix_neighbor_vectors(neighbor_vectors, neighbor_relations=None, user_embeddings=None, want_probs=False, logits=None):
    """mvin_mix_neighbor_vectors_fwd (aggregators.py:37-77 / :118-152): neighbor_vectors [B,N,K,D] -> mean_k(p * neighbor) [B,N,D]
    (and p [B,N,K]); p = softmax_k(mean_d(user * relation)), or softmax_k(logits [B,N,K]), or 1 when neither is given."""
    lib = _lib.load()
    _chk(neighbor_vectors, F32, "neighbor_vectors")
    B, N, K, D = neighbor_vectors.shape
    if logits is not None:
        _chk(logits, F32, "logits")
        if logits.numel() != B * N * K:
            raise ValueError("logits must be [B,N,K]")
    elif neighbor_relations is not None:
        _chk(neighbor_relations, F32, "neighbor_relations")
        _chk(user_embeddings, F32, "user_embeddings")
        if tuple(neighbor_relations.shape) != (B, N, K, D) or tuple(user_embeddings.shape) != (B, D):
            raise ValueError("neighbor_relations must be [B,N,K,D] and user_embeddings [B,D]")
    out = torch.empty((B, N, D), dtype=F32, device=neighbor_vectors.device)
    probs = torch.empty((B, N, K), dtype=F32, device=neighbor_vectors.device) if want_probs else None
    use_rel = logits is None and neighbor_relations is not None
    _lib.check(lib.mvin_mix_neighbor_vectors_fwd(_p(neighbor_vectors), _p(neighbor_relations) if use_rel else None,
                                                 _p(user_embeddings) if use_rel else None, _p(logits), B, N, K, D,
                                                 _p(out), _p(probs), _stream()), "mvin_mix_neighbor_vectors_fwd")
    return (out, probs) if want_probs else out


def key_addressing_supported(Nm, D):
    return bool(_lib.load().mvin_key_addressing_supported(Nm, D))


def key_addressing(entity_emb, V, w, mem_h, mem_r, mem_t, P, out, ldo, nR):
    """mvin_key_addressing_fwd: every preference-hop attention read of a batch in one launch;
    fills ``out`` [B, ldo] with [o_hset | o_hop0 | ...]."""
    lib = _lib.load()
    bf = _chk_table(entity_emb, "entity_emb")
    _chk(V, F32, "V"), _chk(w, F32, "w"), _chk(out, F32, "out")
    nh = max(1, P)
    arr_t = C.c_void_p * nh
    for lst, nm in ((mem_h[:nh], "mem_h"), (mem_r[:P], "mem_r"), (mem_t[:P], "mem_t")):
        for t in lst:
            _chk(t, I32, nm)
    ph = arr_t(*[t.data_ptr() for t in mem_h[:nh]])
    pr = arr_t(*([t.data_ptr() for t in mem_r[:P]] + [None] * (nh - P)))
    pt = arr_t(*([t.data_ptr() for t in mem_t[:P]] + [None] * (nh - P)))
    B, Nm = mem_h[0].shape
    D = entity_emb.shape[1]
    _lib.check(lib.mvin_key_addressing_fwd(_p(entity_emb), _p(V), _p(w), ph, pr, pt, P, B, Nm, D, nR,
                                           entity_emb.shape[0], _p(out), ldo, bf, _stream()),
               "mvin_key_addressing_fwd")
    return out


def key_addressing_users(entity_emb, V, w, uts, users, P, out, ldo, nR):
    """mvin_key_addressing_users_fwd: key_addressing() with pair b reading the ripple sets of users[b] out of the
    device-resident user_triplet_set ``uts`` [n_user, max(1,P), 3, Nm] int32 (no per-pair [B, Nm] arrays)."""
    lib = _lib.load()
    bf = _chk_table(entity_emb, "entity_emb")
    _chk(V, F32, "V"), _chk(w, F32, "w"), _chk(out, F32, "out"), _chk(uts, I32, "uts")
    if users.dtype not in (torch.int64, I32):
        raise TypeError("users: int64 or int32")
    _chk(users, users.dtype, "users")
    if uts.dim() != 4 or uts.shape[1] != max(1, P) or uts.shape[2] != 3:
        raise ValueError("uts must be [n_user, max(1,P), 3, Nm]")
    B, Nm, D = users.shape[0], uts.shape[3], entity_emb.shape[1]
    u64, u32 = (_p(users), None) if users.dtype == torch.int64 else (None, _p(users))
    _lib.check(lib.mvin_key_addressing_users_fwd(_p(entity_emb), _p(V), _p(w), _p(uts), u64, u32, P, B, Nm, D, nR,
                                                 entity_emb.shape[0], uts.shape[0], _p(out), ldo, bf, _stream()),
               "mvin_key_addressing_users_fwd")
    return out


def l2_tail_supported(D):
    return bool(_lib.load().mvin_l2_tail_supported(D))


def l2_tail(entity_emb, items, q, user_o, nagg0, nagg1, W0, b0, A0, a0, A1, a1, Wmix, bmix):
    """mvin_l2_tail_fwd: projection of level 0, both hop-0 aggregators, the mix-hop combiner and the score in one
    launch (depth-2 trees).  Returns (item_emb [B,D], scores [B], sigmoid [B])."""
    bf = _chk_table(entity_emb, "entity_emb")
    for t, nm in ((q, "q"), (user_o, "user_o"), (nagg0, "nagg0"), (nagg1, "nagg1"), (W0, "W0"), (b0, "b0"), (A0, "A0"),
                  (a0, "a0"), (A1, "A1"), (a1, "a1"), (Wmix, "Wmix"), (bmix, "bmix")):
        _chk(t, F32, nm)
    B, D = user_o.shape
    dev = user_o.device
    i64 = items if items.dtype == torch.int64 else None
    i32 = items if items.dtype == I32 else None
    item_emb = torch.empty((B, D), dtype=F32, device=dev)
    scores = torch.empty((B,), dtype=F32, device=dev)
    sig = torch.empty((B,), dtype=F32, device=dev)
    _lib.check(_lib.load().mvin_l2_tail_fwd(_p(entity_emb), _p(i64), _p(i32), _p(q), _p(user_o), _p(nagg0), _p(nagg1),
                                            _p(W0), _p(b0), _p(A0), _p(a0), _p(A1), _p(a1), _p(Wmix), _p(bmix), B, D,
                                            entity_emb.shape[0], _p(item_emb), _p(scores), _p(sig), bf, _stream()),
               "mvin_l2_tail_fwd")
    return item_emb, scores, sig


def gather_rows(table, ids):
    """mvin_gather_rows: out[i] = table[ids[i]] for rows of any 4-byte-multiple width (fp32 / bf16 entity rows)."""
    _chk(ids, I32, "ids")
    if not table.is_cuda or not table.is_contiguous():
        raise _lib.MvinHipError("table: expected a contiguous CUDA/ROCm tensor")
    out = torch.empty((ids.shape[0], table.shape[1]), dtype=table.dtype, device=table.device)
    _lib.check(_lib.load().mvin_gather_rows(_p(table), _p(ids), ids.shape[0], table.shape[1] * table.element_size(),
                                            _p(out), _stream()), "mvin_gather_rows")
    return out


def scatter_rows(table, ids, rows):
    """mvin_scatter_rows: table[ids[i]] = rows[i] (distinct ids)."""
    _chk(ids, I32, "ids")
    if rows.dtype != table.dtype or not rows.is_contiguous() or not table.is_contiguous():
        raise TypeError("rows/table: same dtype, contiguous")
    _lib.check(_lib.load().mvin_scatter_rows(_p(table), _p(ids), ids.shape[0], table.shape[1] * table.element_size(),
                                             _p(rows), _stream()), "mvin_scatter_rows")
    return table


def shard_space_ids(ids, world, n_local):
    """mvin_shard_space_ids: (ids mod world) * n_local + ids div world, int64 or int32, one launch."""
    if ids.dtype not in (torch.int64, I32) or not ids.is_cuda:
        raise TypeError("ids: int64 or int32 device tensor")
    ids = ids.contiguous()
    out = torch.empty_like(ids)
    _lib.check(_lib.load().mvin_shard_space_ids(_p(ids), int(ids.dtype == torch.int64), ids.numel(), world, n_local,
                                                _p(out), _stream()), "mvin_shard_space_ids")
    return out


def key_addressing_grouped_supported(D, P, Nm, nR):
    return bool(_lib.load().mvin_key_addressing_grouped_supported(D, P, Nm, nR))


def group_pairs_by_user(users, n_user=None):
    """Segments of a batch in user order, built on the device with static shapes (no host sync, graph-capturable):
    returns (seg_user [B] int32, seg_ptr [B+2] int32, nseg [1] int32, pair_index [B] int32); only the first
    nseg entries of seg_user / nseg+1 of seg_ptr are meaningful.  With ``n_user`` (ids in [0, n_user)):
    mvin_group_pairs_by_user, a counting sort in three small kernels; without: torch.sort + scans."""
    B = users.shape[0]
    if n_user is not None and users.is_cuda and users.dtype in (torch.int64, I32) and users.is_contiguous():
        dev = users.device
        ws = torch.empty(2 * n_user + B, dtype=I32, device=dev)      # counters | offsets | per-pair ranks
        seg_user = torch.empty(B, dtype=I32, device=dev)
        seg_ptr = torch.empty(B + 2, dtype=I32, device=dev)
        nseg = torch.empty(1, dtype=I32, device=dev)
        pair_index = torch.empty(B, dtype=I32, device=dev)
        u64, u32 = (_p(users), None) if users.dtype == torch.int64 else (None, _p(users))
        _lib.check(_lib.load().mvin_group_pairs_by_user(u64, u32, B, n_user, _p(ws), _p(seg_user), _p(seg_ptr), _p(nseg),
                                                        _p(pair_index), _stream()), "mvin_group_pairs_by_user")
        return seg_user, seg_ptr, nseg, pair_index
    su, perm = torch.sort(users)
    start = torch.ones(B, dtype=torch.bool, device=users.device)
    start[1:] = su[1:] != su[:-1]
    seg_id = torch.cumsum(start, 0) - 1
    nseg = (seg_id[-1:] + 1).to(I32)
    pos = torch.arange(B, dtype=I32, device=users.device)
    seg_ptr = torch.full((B + 2,), B, dtype=I32, device=users.device)
    seg_ptr.scatter_(0, torch.where(start, seg_id, torch.full_like(seg_id, B + 1)), pos)
    seg_user = su[seg_ptr[:B].clamp(max=B - 1).long()].to(I32)
    return seg_user, seg_ptr, nseg, perm.to(I32)


def user_records_len(P, Nm, nR):
    """int32 words of one user's static record (mvin_user_records_len); 0: no record form for this shape."""
    return int(_lib.load().mvin_user_records_len(P, Nm, nR))


def user_records_supported(D, P, Nm, nR, table_bf16=False):
    """True when mvin_key_addressing_grouped_rec_fwd has a kernel over the static records for this shape."""
    return bool(_lib.load().mvin_user_records_supported(D, P, Nm, nR, 1 if table_bf16 else 0))


def build_user_records(uts, P, nR, n_entity):
    """mvin_build_user_records: the static per-user records of ``uts`` [n_user, P, 3, Nm] int32 (relation buckets, tile
    table, clamped head / tail ids) -> [n_user, user_records_len] int32.  Built once per data set, like the adjacency
    encoding: the user's ripple sets are fixed (data_loader_user_set.py), every batch re-reads them."""
    _chk(uts, I32, "uts")
    n_user, Ph, three, Nm = uts.shape
    if Ph != P or three != 3:
        raise ValueError(f"uts shape {tuple(uts.shape)} does not match P={P}")
    n = user_records_len(P, Nm, nR)
    if n == 0:
        raise ValueError(f"no record form for P={P} Nm={Nm} nR={nR}")
    rec = torch.empty((n_user, n), dtype=I32, device=uts.device)
    _lib.check(_lib.load().mvin_build_user_records(_p(uts), n_user, P, Nm, nR, n_entity, _p(rec), _stream()),
               "mvin_build_user_records")
    return rec


def key_addressing_grouped_er_supported(D, P, Nm, nR, n_entity, has_set):
    return bool(_lib.load().mvin_key_addressing_grouped_er_supported(D, P, Nm, nR, n_entity, 1 if has_set else 0))


def project_relations(entity_emb, relation_kge, w=None, out=None):
    """mvin_project_relations: the workspace of the gathered key addressing -- R_KGE[r] . E[e] for every (relation, entity)
    and E[e] . w -- from the CURRENT parameters."""
    lib = _lib.load()
    _chk(entity_emb, F32, "entity_emb"), _chk(relation_kge, F32, "relation_kge"), _chk(w, F32, "w")
    nE, D = entity_emb.shape
    nR = relation_kge.shape[0]
    n = lib.mvin_project_relations_elems(nE, nR, D)
    if out is None:
        out = torch.empty((n,), dtype=F32, device=entity_emb.device)
    elif out.numel() != n or out.dtype != F32 or not out.is_contiguous():
        raise ValueError("project_relations: workspace of mvin_project_relations_elems floats expected")
    _lib.check(lib.mvin_project_relations(_p(entity_emb), _p(relation_kge), _p(w), nE, nR, D, _p(out), _stream()), "mvin_project_relations")
    return out


def key_addressing_grouped(entity_emb, relation_kge, w, uts, groups, items, P, out, ldo, nR, records=None, er=None):
    """mvin_key_addressing_grouped_fwd: the attention reads of a batch whose pairs are grouped by user
    (``groups`` = group_pairs_by_user(users)); fills ``out`` [B, ldo] with [o_hset | o_hop0 | ...].
    ``records`` = build_user_records(uts, ...): the same results from the kernel over static per-user records;
    ``er`` = project_relations(...): that kernel with the users' U rows gathered (mvin_key_addressing_grouped_er_fwd)."""
    lib = _lib.load()
    bf = _chk_table(entity_emb, "entity_emb")
    _chk(relation_kge, F32, "relation_kge"), _chk(w, F32, "w"), _chk(out, F32, "out"), _chk(uts, I32, "uts")
    seg_user, seg_ptr, nseg, perm = groups
    B = items.shape[0]
    n_user, Ph, three, Nm = uts.shape
    D = entity_emb.shape[1]
    i64 = items if items.dtype == torch.int64 else None
    i32 = items if items.dtype == I32 else None
    if i64 is None and i32 is None:
        raise TypeError("items must be int64 or int32")
    if records is not None:
        _chk(records, I32, "records")
        if tuple(records.shape) != (n_user, user_records_len(P, Nm, nR)):
            raise ValueError(f"records shape {tuple(records.shape)} is not that of build_user_records(uts, {P}, {nR}, ...)")
    if er is not None:
        _chk(er, F32, "er")
        if records is None or er.numel() != lib.mvin_project_relations_elems(entity_emb.shape[0], nR, D):
            raise ValueError("key_addressing_grouped: er = project_relations(entity_emb, relation_kge, w) goes with the user records")
        _lib.check(lib.mvin_key_addressing_grouped_er_fwd(_p(entity_emb), _p(relation_kge), _p(w), _p(uts), _p(records), _p(er),
                                                          _p(seg_user), _p(seg_ptr), _p(nseg), _p(perm), _p(i64), _p(i32), B, B, P, Nm, D,
                                                          nR, entity_emb.shape[0], n_user, _p(out), ldo, _stream()),
                   "mvin_key_addressing_grouped_er_fwd")
        return out
    _lib.check(lib.mvin_key_addressing_grouped_rec_fwd(_p(entity_emb), _p(relation_kge), _p(w), _p(uts), _p(records), _p(seg_user),
                                                       _p(seg_ptr), _p(nseg), _p(perm), _p(i64), _p(i32), B, B, P, Nm, D, nR,
                                                       entity_emb.shape[0], n_user, _p(out), ldo, bf, _stream()),
               "mvin_key_addressing_grouped_fwd")
    return out


def key_addressing_flash_supported(D, P, Nm, nR, n_entity):
    return bool(_lib.load().mvin_key_addressing_flash_supported(D, P, Nm, nR, n_entity))


def key_addressing_flash_prepare(entity_emb, relation_kge, w, user_mlp_W, P, out=None):
    """mvin_key_addressing_flash_prepare: the per-call tables of the flash form -- R_KGE[r] . E[e] for every (relation, entity),
    E[e] . w (``w`` None: no h-set read) and E . user_mlp_W[D j : D j + D] per block of o_list -- from the CURRENT parameters."""
    lib = _lib.load()
    for t, nm in ((entity_emb, "entity_emb"), (relation_kge, "relation_kge"), (w, "w"), (user_mlp_W, "user_mlp_W")):
        _chk(t, F32, nm)
    nE, D = entity_emb.shape
    nR = relation_kge.shape[0]
    has_set = w is not None
    if tuple(user_mlp_W.shape) != ((P + (1 if has_set else 0)) * D, D):
        raise ValueError("key_addressing_flash_prepare: user_mlp_W [(P + has_set) D, D] expected")
    n = lib.mvin_key_addressing_flash_tables_elems(nE, nR, D, P, 1 if has_set else 0)
    if out is None:
        out = torch.empty((n,), dtype=F32, device=entity_emb.device)
    elif out.numel() != n or out.dtype != F32 or not out.is_contiguous():
        raise ValueError("key_addressing_flash_prepare: workspace of mvin_key_addressing_flash_tables_elems floats expected")
    _lib.check(lib.mvin_key_addressing_flash_prepare(_p(entity_emb), _p(relation_kge), _p(w), _p(user_mlp_W), nE, nR, D, P, _p(out), _stream()),
               "mvin_key_addressing_flash_prepare")
    return out


def key_addressing_flash(entity_emb, tables, records, groups, items, P, Nm, nR, has_set, user_mlp_b, n_user, sched_ws=None, out=None):
    """mvin_key_addressing_flash_fwd: MVIN._key_addressing + the user MLP for a batch grouped by user (``groups`` =
    group_pairs_by_user(users)) in one barrier-free kernel over the static per-user ``records`` and ``tables`` =
    key_addressing_flash_prepare(entity_emb, relation_kge, w if has_set else None, user_mlp_W, P).  Returns user_o [B, D]."""
    lib = _lib.load()
    for t, dt, nm in ((entity_emb, F32, "entity_emb"), (tables, F32, "tables"), (records, I32, "records"), (user_mlp_b, F32, "user_mlp_b")):
        _chk(t, dt, nm)
    seg_user, seg_ptr, nseg, perm = groups
    B = items.shape[0]
    nE, D = entity_emb.shape
    i64 = items if items.dtype == torch.int64 else None
    i32 = items if items.dtype == I32 else None
    if i64 is None and i32 is None:
        raise TypeError("items must be int64 or int32")
    if tables.numel() != lib.mvin_key_addressing_flash_tables_elems(nE, nR, D, P, 1 if has_set else 0):
        raise ValueError("key_addressing_flash: tables = key_addressing_flash_prepare(entity_emb, relation_kge, w, user_mlp_W, P) expected")
    if tuple(records.shape) != (n_user, user_records_len(P, Nm, nR)):
        raise ValueError(f"records shape {tuple(records.shape)} is not that of build_user_records(uts, {P}, {nR}, ...)")
    n_ws = lib.mvin_key_addressing_flash_ws_elems(B, n_user)
    if sched_ws is None:
        sched_ws = torch.empty((n_ws,), dtype=I32, device=entity_emb.device)
    elif sched_ws.numel() < n_ws or sched_ws.dtype != I32:
        raise ValueError("key_addressing_flash: sched_ws of mvin_key_addressing_flash_ws_elems int32 words expected")
    user_o = out if out is not None else torch.empty((B, D), dtype=F32, device=entity_emb.device)
    _chk(user_o, F32, "out")
    _lib.check(lib.mvin_key_addressing_flash_fwd(_p(entity_emb), _p(tables), _p(records), _p(seg_user), _p(seg_ptr), _p(nseg), _p(perm),
                                                 _p(i64), _p(i32), B, P, Nm, D, nR, nE, n_user, 1 if has_set else 0,
                                                 _p(user_mlp_b), _p(user_o), _p(sched_ws), _stream()),
               "mvin_key_addressing_flash_fwd")
    return user_o


# ------------------------------------------------------------------------------- training ops
def _fill_linear_args(a, srcs, ids, Dout, rows, nz, sum_sources):
    nsrc = len(srcs)
    ids = ids or [None] * nsrc
    ids64 = None
    Dsrc = srcs[0].shape[-1]
    for s in range(nsrc):
        _chk(srcs[s], F32, f"src[{s}]")   # training keeps every table in fp32
        a.src[s] = srcs[s].data_ptr()
        if ids[s] is not None:
            if ids64 is None:
                ids64 = ids[s].dtype == torch.int64
            _chk(ids[s], torch.int64 if ids64 else I32, f"ids[{s}]")
            a.ids[s] = ids[s].data_ptr()
    if rows is None:
        first = next((i for i in ids if i is not None), None)
        rows = first.numel() if first is not None else srcs[0].numel() // Dsrc
    a.nsrc, a.Dsrc, a.Dout, a.rows, a.nz = nsrc, Dsrc, Dout, rows, nz
    a.ids64 = 1 if ids64 else 0
    a.sum_sources = 1 if sum_sources else 0
    return rows


def gather_attn_ex(table, adj_entity, adj_relation, node_ids, rel_score_t, self_vec, Wc, c_child, Wagg, bagg,
                   B, N, K, D):
    """mvin_gather_attn_fwd_ex -> (out, probs | None, s_out, z_out)."""
    lib = _lib.load()
    dev = table.device
    out = torch.empty((B, N, D), dtype=F32, device=dev)
    probs = torch.empty((B, N, K), dtype=F32, device=dev) if rel_score_t is not None else None
    s_out = torch.empty((B * N, D), dtype=F32, device=dev)
    z_out = torch.empty((B * N, D), dtype=F32, device=dev)
    _lib.check(lib.mvin_gather_attn_fwd_ex(_p(table), _p(adj_entity), _p(adj_relation), _p(node_ids), _p(rel_score_t),
                                           _p(self_vec), _p(Wc), _p(c_child), _p(Wagg), _p(bagg), B, N, K, D,
                                           table.shape[0], _p(out), _p(probs), _p(s_out), _p(z_out), 0, _stream()),
               "mvin_gather_attn_fwd_ex")
    return out, probs, s_out, z_out


def agg_ex(self_vec, neigh, rel_ids, rel_score_t, Wagg, bagg, B, N, K, D):
    """mvin_agg_fwd_ex -> (out, probs | None, z_out)."""
    lib = _lib.load()
    dev = self_vec.device
    out = torch.empty((B, N, D), dtype=F32, device=dev)
    probs = torch.empty((B, N, K), dtype=F32, device=dev) if rel_score_t is not None else None
    z_out = torch.empty((B * N, D), dtype=F32, device=dev)
    _lib.check(lib.mvin_agg_fwd_ex(_p(self_vec), _p(neigh), _p(rel_ids), _p(rel_score_t), _p(Wagg), _p(bagg), B, N, K,
                                   D, _p(out), _p(probs), None, _p(z_out), _stream()), "mvin_agg_fwd_ex")
    return out, probs, z_out


def eltwise(mode, n, x, y=None, z=None, w=None, accum=None, alpha=1.0, beta=0.0, beta1=0.0, beta2=0.0, eps=0.0,
            D=1, N=1):
    lib = _lib.load()
    _lib.check(lib.mvin_eltwise(mode, n, _p(x), _p(y), _p(z), _p(w), _p(accum), alpha, beta, beta1, beta2, eps, D, N,
                                _stream()), "mvin_eltwise")


def count_ids(ids, nbins, out=None):
    """mvin_count_ids: float occurrence counts [nbins] of an int32 id list (no host sync); added to ``out``."""
    _chk(ids, I32, "ids")
    if out is None:
        out = torch.zeros(nbins, dtype=F32, device=ids.device)
    _lib.check(_lib.load().mvin_count_ids(_p(ids), ids.numel(), nbins, _p(out), _stream()), "mvin_count_ids")
    return out


def l2_adam_multi(segs, nseg, total, g, m, v, loss_accum, apply_adam, lr_t, beta1, beta2, eps, lr_dev=None):
    """mvin_l2_adam_multi over the flat gradient / Adam-moment buffers (see include/mvin_hip.h);
    ``lr_dev`` (1-element fp32 device tensor): the step size is read on the device (mvin_l2_adam_multi_dev)."""
    lib = _lib.load()
    for t, nm in ((g, "g"), (m, "m"), (v, "v"), (loss_accum, "loss_accum")):
        _chk(t, F32, nm)
    if lr_dev is not None:
        _chk(lr_dev, F32, "lr_dev")
        _lib.check(lib.mvin_l2_adam_multi_dev(_p(segs), nseg, total, _p(g), _p(m), _p(v), _p(loss_accum),
                                              1 if apply_adam else 0, _p(lr_dev), beta1, beta2, eps, _stream()),
                   "mvin_l2_adam_multi_dev")
        return
    _lib.check(lib.mvin_l2_adam_multi(_p(segs), nseg, total, _p(g), _p(m), _p(v), _p(loss_accum),
                                      1 if apply_adam else 0, lr_t, beta1, beta2, eps, _stream()),
               "mvin_l2_adam_multi")


def axpby(alpha, x, beta, y):
    """y = alpha*x + beta*y (in place on y)."""
    eltwise(0, x.numel(), x, y, alpha=alpha, beta=beta)
    return y


def scatter_add_rows(dtable, ids, x, alpha=1.0):
    lib = _lib.load()
    _chk(dtable, F32, "dtable"), _chk(x, F32, "x")
    D = dtable.shape[-1]
    _lib.check(lib.mvin_scatter_add_rows(_p(dtable), _p(ids), 1 if ids.dtype == torch.int64 else 0, _p(x),
                                         ids.numel(), D, alpha, _stream()), "mvin_scatter_add_rows")


def linear_wgrad(srcs, dY, dW, *, ids=None, db=None, mask=None, sum_sources=False, rows=None, nz=1, ldy=None,
                 dy_zstride=0, ldm=None, mask_zstride=0, dw_zstride=0, db_zstride=0):
    """dW[z] += X^T . dY[z] (X staged like ops.linear), db[z] += column sums."""
    lib = _lib.load()
    a = _lib.LinearArgs()
    Dout = dW.shape[-1]
    _fill_linear_args(a, srcs, ids, Dout, rows, nz, sum_sources)
    ldy = ldy or Dout
    ldm = ldm or Dout
    _lib.check(lib.mvin_linear_wgrad(C.byref(a), _p(dY), ldy, dy_zstride, _p(mask), ldm, mask_zstride, _p(dW),
                                     dw_zstride, _p(db), db_zstride, _stream()), "mvin_linear_wgrad")


def wgrad_problem(srcs, dY, dW, *, ids=None, db=None, mask=None, sum_sources=False, rows=None, nz=1, ldy=None,
                  dy_zstride=0, ldm=None, mask_zstride=0, dw_zstride=0, db_zstride=0):
    """One entry for linear_wgrad_multi (same arguments as linear_wgrad).  Returns (struct, tensors kept alive)."""
    pr = _lib.WgradProblem()
    Dout = dW.shape[-1]
    _fill_linear_args(pr.lin, srcs, ids, Dout, rows, nz, sum_sources)
    pr.dY, pr.ldy, pr.dy_zstride = dY.data_ptr(), ldy or Dout, dy_zstride
    pr.mask, pr.ldm, pr.mask_zstride = (mask.data_ptr() if mask is not None else None), ldm or Dout, mask_zstride
    pr.dW, pr.dw_zstride = dW.data_ptr(), dw_zstride
    pr.db, pr.db_zstride = (db.data_ptr() if db is not None else None), db_zstride
    for t, nm in ((dY, "dY"), (dW, "dW"), (db, "db"), (mask, "mask")):
        _chk(t, F32, nm)
    return pr, (list(srcs), list(ids or ()), dY, dW, db, mask)


def linear_wgrad_multi(problems):
    """mvin_linear_wgrad_multi: the weight gradients of ``problems`` (wgrad_problem entries) in as few launches as
    their shapes allow.  The tensors of every entry must stay alive (and unmodified) until this call."""
    if not problems:
        return
    lib = _lib.load()
    for lo in range(0, len(problems), 64):       # the entry point takes at most 64 problems per call (deep trees queue more)
        chunk = problems[lo:lo + 64]
        arr = (_lib.WgradProblem * len(chunk))(*[p for p, _ in chunk])
        _lib.check(lib.mvin_linear_wgrad_multi(arr, len(chunk), _stream()), "mvin_linear_wgrad_multi")


def agg_bwd(dvec, probs, T, K, D, nR, *, table=None, adj_entity=None, adj_relation=None, node_ids=None, child=None,
            rel_ids=None, dtable=None, dT=None, rel_score=None):
    """mvin_agg_bwd; returns dchild (dense form) or None (gather form: dtable updated in place).
    Gather form with node_ids=None and rel_score given = the by-entity form (dvec is [n_entity, D])."""
    lib = _lib.load()
    dchild = torch.empty((T * K, D), dtype=F32, device=dvec.device) if table is None else None
    _lib.check(lib.mvin_agg_bwd(_p(table), _p(adj_entity), _p(adj_relation), _p(node_ids), _p(child), _p(rel_ids),
                                _p(probs), _p(rel_score), _p(dvec), T, K, D, nR, _p(dtable), _p(dchild), _p(dT),
                                _stream()), "mvin_agg_bwd")
    return dchild


def rel_score_bwd(relation_emb, urh_weights, dT, drel, durh):
    lib = _lib.load()
    nR, D = relation_emb.shape
    _lib.check(lib.mvin_rel_score_bwd(_p(relation_emb), _p(urh_weights), _p(dT), nR, D, _p(drel), _p(durh), _stream()),
               "mvin_rel_score_bwd")


def key_addressing_bwd_adds_item_grad(P, Nm, D, nR):
    return bool(_lib.load().mvin_key_addressing_bwd_adds_item_grad(P, Nm, D, nR))


def key_addressing_bwd(entity_emb, V, w, mem_h, mem_r, mem_t, P, dout, ldo, nR, l2, dE, dV, dw, reg_accum=None,
                       relation_kge=None, items=None):
    """mvin_key_addressing_bwd_reg; ``reg_accum`` (1-element fp32): += l2 * (sum h^2 + sum t^2) of the hop rows;
    ``relation_kge`` + ``items``: the kernel adds dE[item] += sum_r dV[:, r] . R[r]^T itself (see the header)."""
    lib = _lib.load()
    nh = max(1, P)
    arr_t = C.c_void_p * nh
    ph = arr_t(*[t.data_ptr() for t in mem_h[:nh]])
    pr = arr_t(*([t.data_ptr() for t in mem_r[:P]] + [None] * (nh - P)))
    pt = arr_t(*([t.data_ptr() for t in mem_t[:P]] + [None] * (nh - P)))
    B, Nm = mem_h[0].shape
    D = entity_emb.shape[1]
    _lib.check(lib.mvin_key_addressing_bwd_reg(_p(entity_emb), _p(V), _p(w), ph, pr, pt, P, B, Nm, D, nR, _p(dout),
                                               ldo, l2, _p(dE), _p(dV), _p(dw),
                                               1 if dw is None or dw.dim() == 1 else dw.shape[0], _p(reg_accum),
                                               _p(relation_kge), _p(items),
                                               1 if items is not None and items.dtype == torch.int64 else 0, _stream()),
               "mvin_key_addressing_bwd")
