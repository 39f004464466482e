#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped boundary: model.get_scores(sess, feed_dict) with HOST
numpy feeds (ids + per-pair ripple sets cross PCIe every call; 1.5 KB per pair at C3), next to the
device-resident rate bench.py reports.  Run on the GPU box."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=131072); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
ds = "last-fm_50core"; d = synth.DATASETS[ds]
args = make_args(dataset=ds, dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"],
                 n_memory=d["n_memory"], batch_size=a.batch)
case = synth.dataset_case(ds, K=32, B=a.batch, seed=0)
params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
m = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
         device="cuda:0")
feed = {m.user_indices: case.users, m.item_indices: case.items}
for i in range(d["p_hop"]):
    feed[m.memories_h[i]], feed[m.memories_r[i]], feed[m.memories_t[i]] = (case.memories_h[i], case.memories_r[i],
                                                                           case.memories_t[i])
for pinned in (False, True):
    if pinned:   # page-locked host arrays: the copies run at full PCIe rate
        feed = {k: torch.from_numpy(np.ascontiguousarray(v)).pin_memory() for k, v in feed.items()}
    m.get_scores(None, feed); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        items, scores = m.get_scores(None, feed)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    host_bytes = sum(np.asarray(v).nbytes if not torch.is_tensor(v) else v.numel() * v.element_size() for v in feed.values())
    print(json.dumps({"workload": f"C3 get_scores(host feeds), B={a.batch}", "pinned_host_memory": pinned,
                      "ms_per_call": round(dt * 1e3, 3), "pairs_per_s": round(a.batch / dt),
                      "host_bytes_per_pair": round(host_bytes / a.batch), "h2d_GBs": round(host_bytes / dt / 1e9, 1)}),
          flush=True)
