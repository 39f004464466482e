"""Op-by-op torch-CPU fp32 mirror of the reference TF1.x graph (TEST INFRASTRUCTURE).

Restatement, not the reference; pinned to the reference's wiring by tests/golden/ref (oracle/__init__.py).

Every function cites the reference lines it follows (paths relative to
/root/reference/src/model/MVIN/).  The mirror keeps TF's op order and TF's
materialised intermediates on purpose -- the tiled [B,N,K,3D] concat of
aggregators.py:121-133, the [B,Nm,D,D] relation-matrix lookup of
model.py:132, one [B,K^e,D] tensor per level -- so that timing it gives a
"TF-graph-equivalent" CPU baseline.

TF op -> torch op used here:
  tf.gather / tf.nn.embedding_lookup -> tensor[index]
  tf.tile -> Tensor.repeat         tf.concat -> torch.cat
  tf.matmul -> torch.matmul        tf.reshape -> Tensor.reshape (row-major)
  tf.nn.softmax -> softmax(dim=-1) tf.reduce_mean/sum -> mean/sum
  tf.nn.dropout(keep_prob=1) -> identity
"""
from types import SimpleNamespace

import torch

F32 = torch.float32


def _t(x, dtype=F32):
    return torch.as_tensor(x).to(dtype)


def as_torch_params(params):
    """numpy/torch dict -> dict of fp32 torch CPU tensors."""
    return {k: _t(v) for k, v in params.items()}


# --------------------------------------------------------------------------- #
# aggregators.py:79-152  SumAggregator_urh_matrix
# --------------------------------------------------------------------------- #
def mix_neighbor_vectors_urh(self_vectors, user_embeddings, neighbor_vectors,
                             neighbor_relations, urh_weights, batch_size, dim):
    """aggregators.py:118-146."""
    B, N, K = neighbor_relations.shape[0], neighbor_relations.shape[1], neighbor_relations.shape[2]
    # :121-122 reshape + tile user embedding to [B,N,K,D]
    u = user_embeddings.reshape(batch_size, 1, 1, dim).repeat(1, N, K, 1)
    # :126-127 expand + tile self vectors to [B,N,K,D]
    s = self_vectors.unsqueeze(2).repeat(1, 1, K, 1)
    # :130-133 concat [user, relation, self] and the skinny matmul (no bias: urh_bias unused)
    urh = torch.cat([u, neighbor_relations, s], dim=-1)
    urh = torch.matmul(urh.reshape(-1, 3 * dim), urh_weights)
    # :136
    probs = urh.reshape(neighbor_vectors.shape[0], neighbor_vectors.shape[1], neighbor_vectors.shape[2])
    # :139
    probs_normalized = torch.softmax(probs, dim=-1)
    # :141-144  reduce_mean over the K axis of p*v  (softmax weights AND a further 1/K)
    neighbors_aggregated = (probs_normalized.unsqueeze(-1) * neighbor_vectors).mean(dim=2)
    return neighbors_aggregated, probs_normalized


def mix_neighbor_vectors_no_ur(neighbor_vectors):
    """aggregators.py:148-152."""
    return neighbor_vectors.mean(dim=2)


def aggregator_call(agg, self_vectors, neighbor_vectors, neighbor_relations, user_embeddings,
                    batch_size, dim):
    """aggregators.py:98-116.  ``agg`` = dict(weights, bias, urh_weights, User_orient_rela)."""
    if agg["User_orient_rela"]:
        neighbors_agg, probs = mix_neighbor_vectors_urh(
            self_vectors, user_embeddings, neighbor_vectors, neighbor_relations,
            agg["urh_weights"], batch_size, dim)
    else:
        neighbors_agg, probs = mix_neighbor_vectors_no_ur(neighbor_vectors), None
    # :108-110 (dropout keep_prob=1 is the identity)
    output = (self_vectors + neighbors_agg).reshape(-1, dim)
    output = torch.matmul(output, agg["weights"]) + agg["bias"]
    # :113-116 (act is always relu, :96)
    output = output.reshape(batch_size, -1, dim)
    return torch.relu(output), probs


# --------------------------------------------------------------------------- #
# model.py
# --------------------------------------------------------------------------- #
def get_neighbors(args, adj_entity, adj_relation, seeds):
    """model.py:243-256."""
    B = args.batch_size
    seeds = seeds.unsqueeze(1)
    entities, relations = [seeds], []
    n = args.neighbor_sample_size
    for i in range(args.n_mix_hop * args.h_hop):
        entities.append(adj_entity[entities[i]].reshape(B, n))
        relations.append(adj_relation[entities[i]].reshape(B, n))
        n *= args.neighbor_sample_size
    return entities, relations


def key_addressing(args, p, user_indices, item_indices, memories_h, memories_r, memories_t):
    """model.py:125-134 (lookups) + :161-240."""
    D = args.dim
    n_lists = max(1, args.p_hop)
    h_emb_list = [p["entity_emb_matrix"][memories_h[i].long()] for i in range(n_lists)]          # :130
    r_emb_list = [p["relation_emb_KGE_matrix"][memories_r[i].long()] for i in range(n_lists)]    # :132 [B,Nm,D,D]
    t_emb_list = [p["entity_emb_matrix"][memories_t[i].long()] for i in range(n_lists)]          # :134

    item_embeddings = p["entity_emb_matrix"][item_indices]                                         # :199
    o_list = []
    if args.PS_O_ft:                                                                               # :204-206, :162-197
        user_embedding_key = p["user_emb_matrix"][user_indices]
        item = user_embedding_key.unsqueeze(1).repeat(1, h_emb_list[0].shape[1], 1)
        h_emb_item = torch.cat([h_emb_list[0], item], dim=2).reshape(-1, D * 2)
        probs = torch.matmul(h_emb_item, p["h_emb_item_mlp_matrix"]).squeeze(-1) + p["h_emb_item_mlp_bias"]
        probs = probs.reshape(-1, h_emb_list[0].shape[1])
        probs_normalized = torch.softmax(probs, dim=-1)
        o_list.append((h_emb_list[0] * probs_normalized.unsqueeze(2)).sum(dim=1))
    for hop in range(args.p_hop):                                                                  # :210-230
        Rh = torch.matmul(r_emb_list[hop], h_emb_list[hop].unsqueeze(3)).squeeze(3)
        v = item_embeddings.unsqueeze(2)
        probs = torch.matmul(Rh, v).squeeze(2)
        probs_normalized = torch.softmax(probs, dim=-1)
        o_list.append((t_emb_list[hop] * probs_normalized.unsqueeze(2)).sum(dim=1))
    o_cat = torch.cat(o_list, dim=-1)                                                              # :232
    n_o = args.p_hop + 1 if args.PS_O_ft else args.p_hop
    user_o = torch.matmul(o_cat.reshape(-1, D * n_o), p["user_mlp_matrix"]) + p["user_mlp_bias"]   # :233-236
    return user_o, [user_o]


def _agg_params(p, i, n, User_orient_rela):
    tag = f"agg_{i}_{n}_"
    return {"weights": p[tag + "weights"], "bias": p[tag + "bias"],
            "urh_weights": p[tag + "urh_weights"], "User_orient_rela": User_orient_rela}


def _user_orient_projection(args, p, entity_vectors, transfer_o):
    """model.py:270-283 (shared by aggregate_delta_whole and aggregate :336-355)."""
    B, D, K = args.batch_size, args.dim, args.neighbor_sample_size
    for index in range(len(transfer_o)):
        transfer_o[index] = transfer_o[index].unsqueeze(1)
    for index in range(len(transfer_o)):
        for e_i in range(len(entity_vectors)):
            n_entities = entity_vectors[e_i] + transfer_o[index]
            n_entities = torch.matmul(n_entities.reshape(-1, D), p[f"transfer_matrix_{e_i}"]) + p[f"transfer_bias_{e_i}"]
            entity_vectors[e_i] = n_entities.reshape(B, entity_vectors[e_i].shape[1], D)
            transfer_o[index] = transfer_o[index].repeat(1, K, 1)
    return entity_vectors


def aggregate_delta_whole(args, p, entities, relations, transfer_o, trace=None):
    """model.py:259-324."""
    B, D, K, H, M = args.batch_size, args.dim, args.neighbor_sample_size, args.h_hop, args.n_mix_hop
    user_query = transfer_o[0]
    entity_vectors = [p["entity_emb_matrix"][i] for i in entities]        # :267
    relation_vectors = [p["relation_emb_matrix"][i] for i in relations]   # :268
    if args.User_orient:
        entity_vectors = _user_orient_projection(args, p, entity_vectors, transfer_o)
    if trace is not None:
        trace["ev_proj"] = [e.clone() for e in entity_vectors]
    importance_list = []
    for n in range(M):                                                    # :286
        mix_hop_tmp = [entity_vectors]
        for i in range(H):                                                # :289
            agg = _agg_params(p, i, n, args.User_orient_rela)
            nxt = []
            if i == 0:
                importance_list = []
            for hop in range(H * M - (H * n + i)):                        # :295
                shape = [B, entity_vectors[hop].shape[1], K, D]
                vector, probs = aggregator_call(
                    agg, entity_vectors[hop], entity_vectors[hop + 1].reshape(shape),
                    relation_vectors[hop].reshape(shape), user_query, B, D)
                if i == 0:
                    importance_list.append(probs)
                nxt.append(vector)
            entity_vectors = nxt
            mix_hop_tmp.append(entity_vectors)
        if trace is not None:
            trace.setdefault("stages", []).append([[t.clone() for t in st] for st in mix_hop_tmp])
        entity_vectors = []
        for mip_hop in zip(*mix_hop_tmp):                                 # :310-315
            mip = torch.cat(mip_hop, dim=-1)
            mip = torch.matmul(mip.reshape(-1, D * (H + 1)), p[f"enti_transfer_matrix_{n}"]) + p[f"enti_transfer_bias_{n}"]
            entity_vectors.append(mip.reshape(B, -1, D))
            if len(entity_vectors) == (M - (n + 1)) * H + 1:
                break
    res = entity_vectors[0].reshape(B, D)                                 # :317
    return res, importance_list


def aggregate(args, p, entities, relations, transfer_o):
    """model.py:327-376 (wide_deep=False).  The reference revision is broken here
    (tuple returned by the aggregator is used as a tensor, :366-374); this follows the
    evident intent: take element [0].  Aggregators are built without User_orient_rela
    (:359) so the attention path is always on; their name is ``i`` (no mix index)."""
    B, D, K, H = args.batch_size, args.dim, args.neighbor_sample_size, args.h_hop
    user_query = transfer_o[0]
    entity_vectors = [p["entity_emb_matrix"][i] for i in entities]
    relation_vectors = [p["relation_emb_matrix"][i] for i in relations]
    if args.User_orient:
        entity_vectors = _user_orient_projection(args, p, entity_vectors, transfer_o)
    for i in range(H):
        agg = _agg_params(p, i, 0, True)
        nxt = []
        for hop in range(H - i):
            shape = [B, entity_vectors[hop].shape[1], K, D]
            vector, _ = aggregator_call(agg, entity_vectors[hop], entity_vectors[hop + 1].reshape(shape),
                                        relation_vectors[hop].reshape(shape), user_query, B, D)
            nxt.append(vector)
        entity_vectors = nxt
    return entity_vectors[0].reshape(B, D), []


def forward(args, params, adj_entity, adj_relation, user_indices, item_indices,
            memories_h, memories_r, memories_t, trace=False):
    """model.py:137-159 wiring.  Returns a SimpleNamespace with scores etc."""
    p = as_torch_params(params)
    adj_entity = torch.as_tensor(adj_entity).long()
    adj_relation = torch.as_tensor(adj_relation).long()
    user_indices = torch.as_tensor(user_indices).long()
    item_indices = torch.as_tensor(item_indices).long()
    memories_h = [torch.as_tensor(m) for m in memories_h]
    memories_r = [torch.as_tensor(m) for m in memories_r]
    memories_t = [torch.as_tensor(m) for m in memories_t]
    tr = {} if trace else None

    entities, relations = get_neighbors(args, adj_entity, adj_relation, item_indices)            # :137
    agg_fun = aggregate_delta_whole if args.wide_deep else aggregate                             # :46-47
    importance = []

    def run_agg(transfer_o):
        if args.wide_deep:
            return aggregate_delta_whole(args, p, entities, relations, transfer_o, tr)
        return aggregate(args, p, entities, relations, transfer_o)

    if args.PS_only:                                                                             # :142-144
        user_o, _ = key_addressing(args, p, user_indices, item_indices, memories_h, memories_r, memories_t)
        item_embeddings = p["entity_emb_matrix"][item_indices]
    elif args.HO_only:                                                                           # :146-150
        user_o = p["user_emb_matrix"][user_indices]
        if args.User_orient_kg_eh:
            _, transfer_o = key_addressing(args, p, user_indices, item_indices, memories_h, memories_r, memories_t)
        else:
            transfer_o = [user_o]
        item_embeddings, importance = run_agg(transfer_o)
    else:                                                                                        # :152-156
        user_o, transfer_o = key_addressing(args, p, user_indices, item_indices, memories_h, memories_r, memories_t)
        if not args.User_orient_kg_eh:
            transfer_o = [p["user_emb_matrix"][user_indices]]
        item_embeddings, importance = run_agg(transfer_o)
    scores = (user_o * item_embeddings).sum(dim=1)                                               # :158
    scores_normalized = torch.sigmoid(scores)                                                    # :159
    del agg_fun
    return SimpleNamespace(scores=scores, scores_normalized=scores_normalized, user_o=user_o,
                           item_embeddings=item_embeddings, importance_list=importance,
                           entities=entities, relations=relations, trace=tr)
