"""From-the-equations numpy fp64 restatement of the MVIN scoring path (TEST INFRASTRUCTURE).

Pinned to the reference's wiring by tests/golden/ref (see oracle/__init__.py); TF arithmetic unpinned.  Written independently of
oracle/mirror_fp32.py: one (user,item) pair at a time, no batch axis, no
tile/concat -- the attention score is formed from the three slices of
``urh_weights`` and the tree is walked level by level with explicit child
indexing ``j*K + k`` (row-major reshape of model.py:251-252, :296-301).

Equations (reference lines in brackets, paths under src/model/MVIN/):
  ids      x_0 = item ; x_{e+1}[jK+k] = adj_entity[x_e[j], k] ; r_e[jK+k] = adj_relation[x_e[j], k]   [model.py:243-256]
  proj     v_e[j] = (E[x_e[j]] + q) W_e + b_e      (User_orient)  else  E[x_e[j]]                     [model.py:267-283]
  score    s[j,k] = q.w_u + Rel[r_e[jK+k]].w_r + self[j].w_s ; p = softmax_k(s)                        [aggregators.py:118-139]
  mix      a[j] = (1/K) sum_k p[j,k] child[jK+k]    (or (1/K) sum_k child[jK+k] w/o User_orient_rela)  [aggregators.py:141-152]
  out      relu((self[j] + a[j]) A + a_bias)                                                           [aggregators.py:108-116]
  combine  v'_e[j] = [stage_0 ; ... ; stage_H]_e[j] Wmix_n + bmix_n                                    [model.py:309-315]
  user     o_m-attention over ripple sets, user_o = [o_*] U_mlp + u_bias                               [model.py:161-240]
  score    sigma(user_o . item_emb)                                                                    [model.py:158-159]
"""
from types import SimpleNamespace

import numpy as np

F64 = np.float64


def _softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def _p64(params):
    return {k: np.asarray(v, dtype=F64) for k, v in params.items()}


def pair_user_vector(args, p, user, item, mem_h, mem_r, mem_t):
    """model.py:161-240 for one pair.  mem_*[hop] are 1-D int arrays of length Nm."""
    E, U, RK = p["entity_emb_matrix"], p["user_emb_matrix"], p["relation_emb_KGE_matrix"]
    D = args.dim
    v = E[item]
    outs = []
    if args.PS_O_ft:
        w = p["h_emb_item_mlp_matrix"][:, 0]
        w_h, w_u = w[:D], w[D:]
        h0 = E[np.asarray(mem_h[0], dtype=np.int64)]
        s = h0 @ w_h + U[user] @ w_u + p["h_emb_item_mlp_bias"][0]
        outs.append(_softmax(s) @ h0)
    for hop in range(args.p_hop):
        hs = E[np.asarray(mem_h[hop], dtype=np.int64)]
        ts = E[np.asarray(mem_t[hop], dtype=np.int64)]
        rs = np.asarray(mem_r[hop], dtype=np.int64)
        s = np.array([(RK[rs[m]] @ hs[m]) @ v for m in range(len(rs))])
        outs.append(_softmax(s) @ ts)
    o = np.concatenate(outs)
    return o @ p["user_mlp_matrix"] + p["user_mlp_bias"]


def pair_ids(args, adj_entity, adj_relation, item):
    """model.py:243-256 for one pair."""
    L = args.n_mix_hop * args.h_hop
    ents = [np.array([item], dtype=np.int64)]
    rels = []
    for _ in range(L):
        ents.append(adj_entity[ents[-1]].reshape(-1))
        rels.append(adj_relation[ents[-2]].reshape(-1))
    return ents, rels


def _aggregate_once(args, p, tag, self_v, child_v, rel_ids, q, use_att):
    """One aggregator application at one hop for one pair.
    self_v [N,D], child_v [N*K,D], rel_ids [N*K]."""
    D, K = args.dim, args.neighbor_sample_size
    N = self_v.shape[0]
    child = child_v.reshape(N, K, D)
    if use_att:
        w = p[tag + "urh_weights"][:, 0]
        w_u, w_r, w_s = w[:D], w[D:2 * D], w[2 * D:]
        rel = p["relation_emb_matrix"][rel_ids].reshape(N, K, D)
        s = (q @ w_u) + rel @ w_r + (self_v @ w_s)[:, None]
        prob = _softmax(s)
        a = np.einsum("nk,nkd->nd", prob, child) / K
    else:
        prob = None
        a = child.sum(axis=1) / K
    out = np.maximum((self_v + a) @ p[tag + "weights"] + p[tag + "bias"], 0.0)
    return out, prob


def pair_item_vector(args, p, ents, rels, q):
    """model.py:259-324 (wide_deep) or :327-376 (legacy) for one pair."""
    H, M = args.h_hop, args.n_mix_hop
    L = H * M
    E = p["entity_emb_matrix"]
    levels = []
    for e in range(L + 1):
        x = E[ents[e]]
        if args.User_orient:
            x = (x + q) @ p[f"transfer_matrix_{e}"] + p[f"transfer_bias_{e}"]
        levels.append(x)
    importance = []
    if not args.wide_deep:
        cur = levels
        for i in range(H):
            cur = [_aggregate_once(args, p, f"agg_{i}_0_", cur[h], cur[h + 1], rels[h], q, True)[0]
                   for h in range(H - i)]
        return cur[0][0], importance
    for n in range(M):
        stages = [levels]
        cur = levels
        for i in range(H):
            nxt, probs = [], []
            for h in range(L - (H * n + i)):
                o, pr = _aggregate_once(args, p, f"agg_{i}_{n}_", cur[h], cur[h + 1], rels[h], q,
                                        args.User_orient_rela)
                nxt.append(o)
                probs.append(pr)
            if i == 0:
                importance = probs
            cur = nxt
            stages.append(cur)
        keep = (M - n - 1) * H + 1
        Wm, bm = p[f"enti_transfer_matrix_{n}"], p[f"enti_transfer_bias_{n}"]
        levels = [np.concatenate([st[e] for st in stages], axis=-1) @ Wm + bm for e in range(keep)]
    return levels[0][0], importance


def forward(args, params, adj_entity, adj_relation, user_indices, item_indices,
            memories_h, memories_r, memories_t):
    """model.py:137-159 wiring, one pair at a time, fp64."""
    p = _p64(params)
    adj_entity = np.asarray(adj_entity, dtype=np.int64)
    adj_relation = np.asarray(adj_relation, dtype=np.int64)
    user_indices = np.asarray(user_indices, dtype=np.int64)
    item_indices = np.asarray(item_indices, dtype=np.int64)
    B = len(user_indices)
    scores, user_os, item_embs, imps = [], [], [], []
    for b in range(B):
        u, it = int(user_indices[b]), int(item_indices[b])
        mh = [np.asarray(m)[b] for m in memories_h]
        mr = [np.asarray(m)[b] for m in memories_r]
        mt = [np.asarray(m)[b] for m in memories_t]
        need_ps = args.PS_only or (not args.HO_only) or args.User_orient_kg_eh
        ps = pair_user_vector(args, p, u, it, mh, mr, mt) if need_ps else None
        if args.PS_only:
            user_o, item_emb, imp = ps, p["entity_emb_matrix"][it], []
        else:
            user_o = p["user_emb_matrix"][u] if args.HO_only else ps
            q = ps if args.User_orient_kg_eh else p["user_emb_matrix"][u]
            ents, rels = pair_ids(args, adj_entity, adj_relation, it)
            item_emb, imp = pair_item_vector(args, p, ents, rels, q.copy())
        scores.append(float(user_o @ item_emb))
        user_os.append(user_o)
        item_embs.append(item_emb)
        imps.append(imp)
    scores = np.array(scores, dtype=F64)
    n_imp = len(imps[0]) if imps else 0
    importance = []
    for h in range(n_imp):
        if imps[0][h] is None:
            importance.append(None)
        else:
            importance.append(np.stack([imps[b][h] for b in range(B)]))
    return SimpleNamespace(scores=scores, scores_normalized=1.0 / (1.0 + np.exp(-scores)),
                           user_o=np.stack(user_os), item_embeddings=np.stack(item_embs),
                           importance_list=importance)


def mix_neighbor_vectors(neighbor_vectors, neighbor_relations, user_embeddings):
    """Aggregator._mix_neighbor_vectors / _mix_neighbor_vectors_urv (aggregators.py:37-77; the `avg = False` branch, the only
    one reachable): scores = mean_d(user[:, None, None, :] * relations) (:43 / :65), softmax over the K neighbours (:44 / :66),
    mean over K of p * neighbour vectors (:50 / :72).  float64.  -> (aggregated [B,N,D], p [B,N,K])"""
    nv = np.asarray(neighbor_vectors, dtype=np.float64)
    nr = np.asarray(neighbor_relations, dtype=np.float64)
    u = np.asarray(user_embeddings, dtype=np.float64).reshape(nv.shape[0], 1, 1, nv.shape[-1])
    s = (u * nr).mean(-1)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p[..., None] * nv).mean(2), p
