#!/usr/bin/env python3
"""Top-K evaluation workload (util.py:145-181): one user scored against every candidate item.
Times the per-pair form (the user's ripple sets replicated per pair, as the reference feeds them)
against the shared-user form, with and without entity tables.  Run on the GPU box."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import harness, synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params

ap = argparse.ArgumentParser()
ap.add_argument("--dataset", default="last-fm_50core"); ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--fanout", type=int, default=32); ap.add_argument("--hop", type=int, default=2)
ap.add_argument("--users", type=int, default=20); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
d = synth.DATASETS[a.dataset]
args = make_args(dataset=a.dataset, dim=a.dim, neighbor_sample_size=a.fanout, h_hop=a.hop, n_mix_hop=1,
                 p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=512)
case = synth.dataset_case(a.dataset, K=a.fanout, B=8, seed=0)
params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
rng = np.random.default_rng(1)
P, Nm = max(1, d["p_hop"]), d["n_memory"]
uts = np.zeros((case.n_user, P, 3, Nm), dtype=np.int32)
uts[:, :, 0] = rng.integers(0, case.n_entity, (case.n_user, P, Nm))
uts[:, :, 1] = rng.integers(0, case.n_relation, (case.n_user, P, Nm))
uts[:, :, 2] = rng.integers(0, case.n_entity, (case.n_user, P, Nm))
items = np.arange(d["n_item"], dtype=np.int64)
users = rng.integers(0, case.n_user, a.users)
for hoist in (False, True):
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device="cuda:0", hoist=hoist)
    feeder = harness.DeviceFeeder(model, uts)
    for form in ("per-pair", "shared-user"):
        fn = (lambda u: feeder.scores(np.full(items.size, u), items)) if form == "per-pair" else \
             (lambda u: feeder.scores_user(u, items))
        ref = fn(users[0]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            for u in users:
                out = fn(u)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (a.iters * len(users))
        print(json.dumps({"workload": f"{a.dataset} D={a.dim} K={a.fanout} H={a.hop}: 1 user x {items.size} items",
                          "entity_tables": hoist, "form": form, "ms_per_user": round(dt * 1e3, 3),
                          "pairs_per_s": round(items.size / dt)}), flush=True)
