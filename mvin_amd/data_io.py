"""On-disk inputs of the path: the reference's `load_data` stage
(src/model/MVIN/data_loader_user_set.py:18-110, :241-308) without its Python dict loops.

Formats (data/<dataset>/):
  ratings_final.{npy,txt}   [n, 3] int64  (user, item, label)                       :33-43
  kg_final.{npy,txt}        [n, 3] int64  (head, relation, tail)                    :276-286
  {train,eval,test}_pd.csv  columns  <row index>,item,like,user                     :241-254

`load_data` returns the reference's tuple up to `user_triplet_set` with the adjacency and the
ripple sets built on the GPU (mvin_amd.data_prep -> mvin_sample_adjacency / mvin_build_ripple_sets);
everything before that is host numpy.  Random draws (dataset split, samplers) take a seed here; the
reference uses the unseeded global generators, so its draws are not reproducible, its rules are.
"""
import os

import numpy as np


def _load_table(path_no_ext, cache_npy=False):
    """`.npy` if present, else `.txt` (whitespace-separated int64); the reference also writes the
    `.npy` next to the `.txt` (:40-41, :284-285) -- only done here when asked to."""
    if os.path.exists(path_no_ext + ".npy"):
        return np.load(path_no_ext + ".npy").astype(np.int64, copy=False)
    arr = np.loadtxt(path_no_ext + ".txt", dtype=np.int64, ndmin=2)
    if cache_npy:
        np.save(path_no_ext + ".npy", arr)
    return arr


def read_pd_csv(path):
    """One `*_pd.csv` -> [n, 3] int64 (user, item, like)  (:242-244: drop the first column, then
    select ['user', 'item', 'like'] by NAME, whatever the column order in the file)."""
    with open(path) as f:
        header = f.readline().rstrip("\r\n").split(",")
    cols = {name: i for i, name in enumerate(header) if i > 0}
    missing = [c for c in ("user", "item", "like") if c not in cols]
    if missing:
        raise ValueError(f"{path}: missing column(s) {missing}; header = {header}")
    arr = np.loadtxt(path, delimiter=",", skiprows=1, dtype=np.int64, ndmin=2,
                     usecols=[cols["user"], cols["item"], cols["like"]])
    return arr


def load_pre_data(data_dir):
    """:241-254."""
    return tuple(read_pd_csv(os.path.join(data_dir, f"{s}_pd.csv")) for s in ("train", "eval", "test"))


def dataset_split(rating_np, ratio=1.0, seed=0, eval_ratio=0.2, test_ratio=0.2):
    """:256-273: 20 % eval, 20 % test (of ALL ratings, without replacement), rest train, optional
    sub-sampling of the train part."""
    rng = np.random.default_rng(seed)
    n = rating_np.shape[0]
    perm = rng.permutation(n)
    n_eval, n_test = int(n * eval_ratio), int(n * test_ratio)
    eval_idx, test_idx, train_idx = perm[:n_eval], perm[n_eval:n_eval + n_test], np.sort(perm[n_eval + n_test:])
    if ratio < 1:
        train_idx = rng.choice(train_idx, size=int(len(train_idx) * ratio), replace=False)
    return rating_np[train_idx], rating_np[eval_idx], rating_np[test_idx]


def most_popular_items(rating_np, top_k=500):
    """:49-57: the `top_k` most frequent items; ties keep first-appearance order (a stable sort of
    the insertion-ordered dict)."""
    items, first, counts = np.unique(rating_np[:, 1], return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")                 # dict insertion order
    items, counts = items[order], counts[order]
    top = np.argsort(-counts, kind="stable")[:top_k]
    return set(int(i) for i in items[top])


def user_history(train_data):
    """:78-88: user -> list of positively rated train items, in file order."""
    pos = train_data[train_data[:, 2] == 1]
    order = np.argsort(pos[:, 0], kind="stable")
    users, starts = np.unique(pos[order, 0], return_index=True)
    chunks = np.split(pos[order, 1], starts[1:])
    return {int(u): [int(i) for i in c] for u, c in zip(users, chunks)}


def load_rating(data_dir, new_load_data=False, ratio=1.0, seed=0, top_k=500, cache_npy=False):
    """:33-110 -> (n_user, n_item, train, eval, test, user_history_dict, item_set_most_pop)."""
    rating_np = _load_table(os.path.join(data_dir, "ratings_final"), cache_npy)
    n_user = int(rating_np[:, 0].max()) + 1
    n_item = int(rating_np[:, 1].max()) + 1
    item_set_most_pop = most_popular_items(rating_np, top_k)
    if new_load_data:
        train, ev, test = dataset_split(rating_np, ratio, seed)
    else:
        train, ev, test = load_pre_data(data_dir)
    hist = user_history(train)
    known = np.fromiter(hist.keys(), dtype=np.int64, count=len(hist))
    train, ev, test = (d[np.isin(d[:, 0], known)] for d in (train, ev, test))   # :90-96
    return n_user, n_item, train, ev, test, hist, item_set_most_pop


def load_kg_triples(data_dir, cache_npy=False):
    """:276-289 -> (kg_np [n,3] (h, r, t), n_entity, n_relation); counts are numbers of DISTINCT
    ids, as the reference computes them."""
    kg_np = _load_table(os.path.join(data_dir, "kg_final"), cache_npy)
    n_entity = int(np.union1d(kg_np[:, 0], kg_np[:, 2]).size)
    n_relation = int(np.unique(kg_np[:, 1]).size)
    return kg_np, n_entity, n_relation


def load_data(data_dir, neighbor_sample_size, p_hop, n_memory, device="cuda", new_load_data=False, ratio=1.0,
              seed=0, n_neighbor=16):
    """:18-31.  Returns the reference's 16-tuple, position for position (train.py:17-21 reads it by
    index): (n_user, n_item, n_entity, n_relation, train_data, eval_data, test_data, adj_entity,
    adj_relation, user_triplet_set, user_path, user_path_top_k, item_set_most_pop, user_history_dict,
    entity_index_2_name, rela_index_2_name).  ``user_path`` / ``user_path_top_k`` are None in the
    reference too (:301-306); the two name tables are {} unless the amazon case-study files are given
    (:21-24, not read here).  Adjacency ([n_entity, K] int32) and ripple sets ([n_user, p_hop, 3,
    n_memory] int32, an all-zero block for users without history) are device tensors, ready for
    MVIN(...) / harness.DeviceFeeder."""
    from . import data_prep
    n_user, n_item, train, ev, test, hist, pop = load_rating(data_dir, new_load_data, ratio, seed)
    kg_np, n_entity, n_relation = load_kg_triples(data_dir)
    n_rows = max(n_entity, int(max(kg_np[:, 0].max(), kg_np[:, 2].max())) + 1, n_item)
    csr = data_prep.build_csr(kg_np, n_rows, device=device)
    adj_e, adj_r = data_prep.construct_adj(csr, n_rows, neighbor_sample_size, seed=seed + 1)
    hcsr = data_prep.history_csr(train, n_user, device=device)
    uts = data_prep.get_user_triplet_set(csr, hcsr, n_user, p_hop, n_memory, seed=seed + 2, n_neighbor=n_neighbor)
    return (n_user, n_item, n_rows, n_relation, train, ev, test, adj_e, adj_r, uts, None, None, pop, hist, {}, {})
