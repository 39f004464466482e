#!/usr/bin/env python3
"""GPU box: whole get_scores path at the reference's own batch sizes, both feeds, eager (no profiling hooks)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params

ds = "last-fm_50core"
d = synth.DATASETS[ds]
dev = torch.device("cuda:0")
for B in (512, 1024, 4096):
    args = make_args(dataset=ds, dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=B)
    case = synth.dataset_case(ds, K=32, B=B, seed=0)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
    users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    uts = torch.from_numpy(case.user_triplet_set).to(dev)
    m = synth.memories_for(case.user_triplet_set, case.users)
    mh, mr, mt = [[torch.from_numpy(x).to(dev) for x in lst] for lst in m]
    fns = {"users feed": lambda: model.forward_users(users, items, uts), "per-pair feed": lambda: model.forward_device(users, items, mh, mr, mt)}
    outs = {}
    for name, fn in fns.items():
        for _ in range(10):
            o = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            o = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        outs[name] = o.scores
        print(f"B={B:5d} {name:14s} {1e6 * dt:8.1f} us/step  {B / dt / 1e6:7.2f} M pairs/s")
    print("      max |score diff| between the feeds:", float((outs["users feed"] - outs["per-pair feed"]).abs().max()))
