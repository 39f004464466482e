"""Inputs of the scoring path built on the GPU (scope row f-1): counterparts of
construct_kg / contruct_random_adj / get_user_triplet_set
(src/model/MVIN/data_loader_user_set.py:324-343, :375-388, :392-441).

The reference builds these with pure-Python dict loops (minutes on amazon-book, repeated for
every stage-wise restart, main.py:16).  Here the KG becomes a CSR on the device (torch sort =
plumbing), and the two samplers are HIP kernels (mvin_sample_adjacency, mvin_build_ripple_sets)
whose draws are a pure function of a seed, so adjacency can be re-sampled every epoch.
"""
import numpy as np
import torch

from . import _lib
from .ops import _p, _stream


def build_csr(kg, n_entity, device="cuda"):
    """construct_kg (:324-343): treat the KG as undirected; every triple (h, r, t) is listed
    under h as (t, r) and under t as (h, r), in file order (head entry before tail entry).
    Returns (indptr int64 [nE+1], dst int32 [2n], rel int32 [2n]) on ``device``."""
    kg = torch.as_tensor(np.asarray(kg), dtype=torch.int64).to(device)
    n = kg.shape[0]
    src = torch.stack([kg[:, 0], kg[:, 2]], dim=1).reshape(-1)      # interleaved: h_0, t_0, h_1, t_1 ...
    dst = torch.stack([kg[:, 2], kg[:, 0]], dim=1).reshape(-1)
    rel = torch.stack([kg[:, 1], kg[:, 1]], dim=1).reshape(-1)
    order = torch.sort(src, stable=True).indices                    # keeps insertion order per entity
    deg = torch.bincount(src, minlength=n_entity)
    indptr = torch.zeros(n_entity + 1, dtype=torch.int64, device=kg.device)
    indptr[1:] = torch.cumsum(deg, 0)
    del n
    return indptr.contiguous(), dst[order].to(torch.int32).contiguous(), rel[order].to(torch.int32).contiguous()


def construct_adj(csr, n_entity, K, seed=1):
    """contruct_random_adj (:375-388) -> (adj_entity, adj_relation) int32 [nE, K] on the device."""
    indptr, dst, rel = csr
    lib = _lib.load()
    if dst.numel() == 0:      # a KG without triples: every entity keeps the all-zero row of :377-380 (found by the prep fuzz test)
        z = torch.zeros((n_entity, K), dtype=torch.int32, device=indptr.device)
        return z, z.clone()
    adj_e = torch.empty((n_entity, K), dtype=torch.int32, device=indptr.device)
    adj_r = torch.empty((n_entity, K), dtype=torch.int32, device=indptr.device)
    _lib.check(lib.mvin_sample_adjacency(_p(indptr), _p(dst), _p(rel), n_entity, K, seed, _p(adj_e), _p(adj_r),
                                         _stream()), "mvin_sample_adjacency")
    return adj_e, adj_r


def history_csr(train_data, n_user, device="cuda"):
    """user_history_dict of load_rating (data_loader_user_set.py:74-85) as CSR: each user's
    positive train items in interaction order."""
    d = np.asarray(train_data)
    pos = d[d[:, 2] == 1]
    order = np.argsort(pos[:, 0], kind="stable")
    users, items = pos[order, 0], pos[order, 1]
    ptr = np.zeros(n_user + 1, dtype=np.int64)
    np.add.at(ptr, users + 1, 1)
    np.cumsum(ptr, out=ptr)
    return (torch.from_numpy(ptr).to(device), torch.from_numpy(items.astype(np.int32)).to(device))


def get_user_triplet_set(csr, hist, n_user, p_hop, n_memory, seed=1, n_neighbor=16):
    """get_user_triplet_set (:392-441) -> int32 [n_user, max(1,P), 3, n_memory] on the device
    (the layout mvin_amd.harness.DeviceFeeder consumes).  Users without positive items keep
    zero rows (the reference simply has no entry for them)."""
    indptr, dst, rel = csr
    hist_ptr, hist_items = hist
    P = max(1, p_hop)
    lib = _lib.load()
    out = torch.zeros((n_user, P, 3, n_memory), dtype=torch.int32, device=indptr.device)
    if hist_items.numel() == 0 or dst.numel() == 0:     # nobody has a positive item / the KG has no triples: no entries
        return out
    _lib.check(lib.mvin_build_ripple_sets(_p(indptr), _p(dst), _p(rel), _p(hist_ptr), _p(hist_items), n_user, P,
                                          n_memory, n_neighbor, seed, _p(out), _stream()),
               "mvin_build_ripple_sets")
    return out
