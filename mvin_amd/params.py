"""Parameter set of the MVIN graph (model.py:69-122, aggregators.py:83-93) as a flat dict
of fp32 numpy arrays, initialised the way the reference initialises its tf variables:
tf.contrib.layers.xavier_initializer (uniform, limit sqrt(6/(fan_in+fan_out)), fans as
TF's _compute_fans defines them) and zeros for the aggregator biases.

Names (n = mix block, e = tree level, i = aggregator iteration):
  user_emb_matrix [nU,D]  entity_emb_matrix [nE,D]  relation_emb_matrix [nR,D]
  relation_emb_KGE_matrix [nR,D,D]
  enti_transfer_matrix_{n} [(H+1)D, D]  enti_transfer_bias_{n} [D]        (model.py:91-98)
  user_mlp_matrix [(P+1 or P)D, D]      user_mlp_bias [D]                 (model.py:100-106)
  transfer_matrix_{e} [D,D]             transfer_bias_{e} [D]  e=0..M*H   (model.py:107-116)
  h_emb_item_mlp_matrix [2D,1]          h_emb_item_mlp_bias [1]           (model.py:118-122)
  agg_{i}_{n}_weights [D,D]  agg_{i}_{n}_bias [D]  agg_{i}_{n}_urh_weights [3D,1]
  agg_{i}_{n}_urh_bias [1] (created, never used: aggregators.py:92-93 vs :133)
"""
import numpy as np


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def xavier_uniform(rng, shape):
    fan_in, fan_out = _fans(shape)
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def aggregator_keys(args):
    """(i, n) of every aggregator the graph builds: model.py:286-291 (wide_deep) or
    :357-360 (legacy ``aggregate``: one per h_hop iteration, mix index fixed to 0).  With
    ``PS_only`` the aggregation function is never called (model.py:142-144), so the reference graph
    holds no aggregator variables at all (pinned by tests/test_ref_pins.py)."""
    if getattr(args, "PS_only", False):
        return []
    if args.wide_deep:
        return [(i, n) for n in range(args.n_mix_hop) for i in range(args.h_hop)]
    return [(i, 0) for i in range(args.h_hop)]


def init_params(args, n_user, n_entity, n_relation, seed=0, random_agg_bias=False):
    rng = np.random.default_rng(seed)
    D, H, M, P = args.dim, args.h_hop, args.n_mix_hop, args.p_hop
    p = {
        "user_emb_matrix": xavier_uniform(rng, (n_user, D)),
        "entity_emb_matrix": xavier_uniform(rng, (n_entity, D)),
        "relation_emb_matrix": xavier_uniform(rng, (n_relation, D)),
        "relation_emb_KGE_matrix": xavier_uniform(rng, (n_relation, D, D)),
    }
    for n in range(M):
        p[f"enti_transfer_matrix_{n}"] = xavier_uniform(rng, (D * (H + 1), D))
        p[f"enti_transfer_bias_{n}"] = xavier_uniform(rng, (D,))
    n_o = P + 1 if args.PS_O_ft else P
    if n_o < 1:
        raise ValueError("p_hop == 0 needs PS_O_ft (model.py:232 concatenates an empty list otherwise)")
    p["user_mlp_matrix"] = xavier_uniform(rng, (D * n_o, D))
    p["user_mlp_bias"] = xavier_uniform(rng, (D,))
    for e in range(M * H + 1):
        p[f"transfer_matrix_{e}"] = xavier_uniform(rng, (D, D))
        p[f"transfer_bias_{e}"] = xavier_uniform(rng, (D,))
    p["h_emb_item_mlp_matrix"] = xavier_uniform(rng, (2 * D, 1))
    p["h_emb_item_mlp_bias"] = xavier_uniform(rng, (1,))
    for (i, n) in aggregator_keys(args):
        tag = f"agg_{i}_{n}_"
        p[tag + "weights"] = xavier_uniform(rng, (D, D))
        p[tag + "bias"] = (xavier_uniform(rng, (D,)) if random_agg_bias else np.zeros(D, np.float32))
        p[tag + "urh_weights"] = xavier_uniform(rng, (3 * D, 1))
        p[tag + "urh_bias"] = np.zeros(1, np.float32)
    return p
