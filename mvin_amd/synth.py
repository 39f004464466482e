"""Synthetic inputs for the MVIN scoring path (numpy only; seeded; build-owned).

The reference ships neither its KG files nor trained tables in this mount
(.MISSING_LARGE_BLOBS), so benchmarks and tests run on synthetic data shaped like the
reference's datasets (data/*/info.txt, data_statiscs/data_statistics.ipynb):

  * ``DATASETS``     -- table sizes / degree statistics / shipped ripple-set settings;
  * ``synth_kg``     -- undirected KG with a heavy-tailed degree distribution and a skewed
                        relation histogram, as (head, relation, tail) triples (the layout of
                        kg_final.npy);
  * ``sample_adjacency`` -- the reference's fixed-fan-out sampler rule
                        (data_loader_user_set.py:375-388): K neighbors without replacement
                        when deg >= K, with replacement otherwise, all-zero row for entities
                        absent from the KG;
  * ``ripple_sets``  -- per-user [P, 3, Nm] int32 (h, r, t) memories
                        (data_loader_user_set.py:392-441 layout; uniform ids);
  * ``pairs``        -- (user, item) batches, items Zipf- or uniformly distributed.
"""
from types import SimpleNamespace

import numpy as np

# info.txt / notebook statistics (mean degree and the share of entities with >= 20 neighbours,
# data_statiscs/data_statistics.ipynb); p_hop / n_memory from src/bash/mvin_*.sh.  ``tail_exponent`` / ``head_sigma`` are the
# two knobs of synth_kg, set per dataset so that BOTH statistics come out (tests/test_host.py pins them within 3 points:
# the duplicate-slot gain of the fused kernels depends on the whole degree distribution, not on the mean alone).
DATASETS = {
    "MovieLens-1M": dict(n_entity=182011, n_user=6036, n_relation=12, n_item=2445, mean_degree=13.65, share_ge20=0.170,
                         tail_exponent=0.5, head_sigma=0.8, p_hop=2, n_memory=64, batch_size=1024),
    "last-fm_50core": dict(n_entity=106389, n_user=23553, n_relation=9, n_item=48091, mean_degree=8.73, share_ge20=0.025,
                           tail_exponent=0.75, head_sigma=0.0, p_hop=2, n_memory=64, batch_size=512),
    "amazon-book_20core": dict(n_entity=113487, n_user=70585, n_relation=39, n_item=24915, mean_degree=45.08,
                               share_ge20=0.959, tail_exponent=0.75, head_sigma=0.25, p_hop=1, n_memory=16, batch_size=512),
}


def synth_kg(n_entity, n_relation, mean_degree, seed=1, tail_exponent=0.75, head_sigma=0.0):
    """[n_triples, 3] int64 (h, r, t).  Undirected mean degree = 2*n_triples/n_entity.
    Heads are uniform (``head_sigma`` = 0) or drawn in proportion to log-normal per-entity weights (a broad middle of the
    degree distribution: MovieLens-1M has 17 % of its entities at >= 20 neighbours with a mean of 13.65), tails follow a
    power law over a random permutation of the entities (heavy tail: a few hubs), relations follow a Zipf(1) histogram."""
    rng = np.random.default_rng(seed)
    n_triples = int(round(n_entity * mean_degree / 2.0))
    if head_sigma > 0:
        hw = rng.lognormal(0.0, head_sigma, n_entity)
        heads = np.minimum(np.searchsorted(np.cumsum(hw / hw.sum()), rng.random(n_triples)), n_entity - 1)
    else:
        heads = rng.integers(0, n_entity, n_triples)
    w = 1.0 / np.power(np.arange(1, n_entity + 1, dtype=np.float64), tail_exponent)
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(n_entity)
    tails = perm[np.minimum(np.searchsorted(cdf, rng.random(n_triples)), n_entity - 1)]
    rw = 1.0 / np.arange(1, n_relation + 1, dtype=np.float64)
    rels = rng.choice(n_relation, size=n_triples, p=rw / rw.sum())
    return np.stack([heads, rels, tails], axis=1).astype(np.int64)


def kg_to_csr(kg, n_entity):
    """Undirected adjacency in CSR form (construct_kg, data_loader_user_set.py:324-343:
    every triple is inserted under its head and under its tail)."""
    src = np.concatenate([kg[:, 0], kg[:, 2]])
    dst = np.concatenate([kg[:, 2], kg[:, 0]])
    rel = np.concatenate([kg[:, 1], kg[:, 1]])
    order = np.argsort(src, kind="stable")
    src, dst, rel = src[order], dst[order], rel[order]
    indptr = np.zeros(n_entity + 1, dtype=np.int64)
    np.add.at(indptr, src + 1, 1)
    np.cumsum(indptr, out=indptr)
    return indptr, dst.astype(np.int64), rel.astype(np.int64)


def sample_adjacency(indptr, dst, rel, K, seed=1):
    """contruct_random_adj (data_loader_user_set.py:375-388), vectorised.
    Returns adj_entity, adj_relation [n_entity, K] int64."""
    rng = np.random.default_rng(seed)
    n_entity = len(indptr) - 1
    deg = np.diff(indptr)
    adj_e = np.zeros((n_entity, K), dtype=np.int64)
    adj_r = np.zeros((n_entity, K), dtype=np.int64)
    # deg < K: sample with replacement
    small = np.nonzero((deg > 0) & (deg < K))[0]
    if small.size:
        pick = (rng.random((small.size, K)) * deg[small, None]).astype(np.int64)
        pos = indptr[small, None] + pick
        adj_e[small], adj_r[small] = dst[pos], rel[pos]
    # deg >= K: K without replacement = first K of a random permutation of the row's edges
    big = np.nonzero(deg >= K)[0]
    if big.size:
        keys = rng.random(len(dst))
        row_of = np.repeat(np.arange(n_entity), deg)
        order = np.lexsort((keys, row_of))          # within each row: random order
        rank = np.arange(len(dst)) - indptr[row_of]  # position inside its row after sorting
        take = order[(rank < K) & (deg[row_of] >= K)]
        adj_e[big] = dst[take].reshape(-1, K)
        adj_r[big] = rel[take].reshape(-1, K)
    return adj_e, adj_r


def uniform_adjacency(n_entity, n_relation, K, seed=1):
    """Worst-case-locality variant: every neighbor uniform over the entities."""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, n_entity, (n_entity, K), dtype=np.int64),
            rng.integers(0, n_relation, (n_entity, K), dtype=np.int64))


def ripple_sets(n_user, n_entity, n_relation, p_hop, n_memory, seed=3):
    """user_triplet_set as one array [n_user, max(1,P), 3, Nm] int32 (h, r, t rows)."""
    rng = np.random.default_rng(seed)
    P = max(1, p_hop)
    out = np.empty((n_user, P, 3, n_memory), dtype=np.int32)
    out[:, :, 0] = rng.integers(0, n_entity, (n_user, P, n_memory))
    out[:, :, 1] = rng.integers(0, n_relation, (n_user, P, n_memory))
    out[:, :, 2] = rng.integers(0, n_entity, (n_user, P, n_memory))
    return out


def pairs(n_user, n_item, B, seed=2, zipf=True):
    """(users [B] int64, items [B] int64): users uniform; items Zipf(1.0) over a random
    permutation of the item range (popularity skew) or uniform."""
    rng = np.random.default_rng(seed)
    users = rng.integers(0, n_user, B, dtype=np.int64)
    if zipf:
        w = 1.0 / np.arange(1, n_item + 1, dtype=np.float64)
        cdf = np.cumsum(w / w.sum())
        perm = rng.permutation(n_item)
        items = perm[np.minimum(np.searchsorted(cdf, rng.random(B)), n_item - 1)].astype(np.int64)
    else:
        items = rng.integers(0, n_item, B, dtype=np.int64)
    return users, items


def memories_for(user_triplet_set, users):
    """Feed assembly of train.py:117-120: memories_x[i] = stack(user_triplet_set[u][i][x])."""
    sel = user_triplet_set[np.asarray(users)]
    P = sel.shape[1]
    return ([np.ascontiguousarray(sel[:, i, 0]) for i in range(P)],
            [np.ascontiguousarray(sel[:, i, 1]) for i in range(P)],
            [np.ascontiguousarray(sel[:, i, 2]) for i in range(P)])


def dataset_case(name, K, B, seed=0, zipf=True, uniform_adj=False, p_hop=None, n_memory=None):
    """Everything a scoring run needs for a dataset-shaped synthetic workload."""
    d = DATASETS[name]
    p_hop = d["p_hop"] if p_hop is None else p_hop
    n_memory = d["n_memory"] if n_memory is None else n_memory
    if uniform_adj:
        adj_e, adj_r = uniform_adjacency(d["n_entity"], d["n_relation"], K, seed=seed + 1)
    else:
        kg = synth_kg(d["n_entity"], d["n_relation"], d["mean_degree"], seed=seed + 1,
                      tail_exponent=d.get("tail_exponent", 0.75), head_sigma=d.get("head_sigma", 0.0))
        adj_e, adj_r = sample_adjacency(*kg_to_csr(kg, d["n_entity"]), K, seed=seed + 1)
    uts = ripple_sets(d["n_user"], d["n_entity"], d["n_relation"], p_hop, n_memory, seed=seed + 3)
    users, items = pairs(d["n_user"], d["n_item"], B, seed=seed + 2, zipf=zipf)
    mh, mr, mt = memories_for(uts, users)
    return SimpleNamespace(name=name, n_entity=d["n_entity"], n_user=d["n_user"], n_relation=d["n_relation"],
                           n_item=d["n_item"], adj_entity=adj_e, adj_relation=adj_r, user_triplet_set=uts,
                           users=users, items=items, memories_h=mh, memories_r=mr, memories_t=mt,
                           p_hop=p_hop, n_memory=n_memory)


def small_case(args, n_user=8, n_entity=64, n_relation=5, seed=0, zero_rows=0, repeats=False):
    """Tiny random case for parity tests: uniform adjacency (optionally with all-zero rows,
    the 'entity absent from the KG' case of data_loader_user_set.py:377-380).
    ``repeats``: rows as contruct_random_adj builds them for low-degree entities (:383-384) -- an entity has
    deg edges (1 .. 2K: most have few, a fifth are hubs) and its row is K draws WITH replacement when deg < K, K distinct edges
    otherwise; an edge may also repeat a neighbour under another relation."""
    rng = np.random.default_rng(seed)
    B, K, Nm = args.batch_size, args.neighbor_sample_size, args.n_memory
    adj_e = rng.integers(0, n_entity, (n_entity, K), dtype=np.int64)
    adj_r = rng.integers(0, n_relation, (n_entity, K), dtype=np.int64)
    if repeats:
        deg = np.minimum(1 + rng.geometric(0.25, n_entity), 2 * K)
        hub = rng.random(n_entity) < 0.2
        deg[hub] = rng.integers(max(1, K // 2), 2 * K + 1, int(hub.sum()))
        for x in range(n_entity):
            ne = rng.integers(0, n_entity, deg[x])
            nr = rng.integers(0, n_relation, deg[x])
            pick = rng.choice(deg[x], K, replace=deg[x] < K)
            adj_e[x], adj_r[x] = ne[pick], nr[pick]
    if zero_rows:
        z = rng.choice(n_entity, zero_rows, replace=False)
        adj_e[z] = 0
        adj_r[z] = 0
    users = rng.integers(0, n_user, B, dtype=np.int64)
    items = rng.integers(0, n_entity, B, dtype=np.int64)
    P = max(1, args.p_hop)
    mh = [rng.integers(0, n_entity, (B, Nm)).astype(np.int32) for _ in range(P)]
    mr = [rng.integers(0, n_relation, (B, Nm)).astype(np.int32) for _ in range(P)]
    mt = [rng.integers(0, n_entity, (B, Nm)).astype(np.int32) for _ in range(P)]
    return SimpleNamespace(n_user=n_user, n_entity=n_entity, n_relation=n_relation, adj_entity=adj_e,
                           adj_relation=adj_r, users=users, items=items, memories_h=mh, memories_r=mr,
                           memories_t=mt)
