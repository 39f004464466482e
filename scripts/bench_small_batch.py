#!/usr/bin/env python3
"""GPU box: whole get_scores path at the reference's own batch sizes (C3 tables), both feeds -- the single-launch kernel
(mvin_score_small_fwd) per group size against the multi-launch native schedule; wall clock over back-to-back steps and
the kernel's own duration from HIP events.

    python scripts/bench_small_batch.py [--sizes 512,4096,16384] [--groups 0,1,2,4,8,16] [--dataset last-fm_50core]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import synth  # noqa: E402
from mvin_amd.config import make_args  # noqa: E402
from mvin_amd.model import MVIN  # noqa: E402
from mvin_amd.params import init_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="512,4096,16384")
ap.add_argument("--groups", default="0,1,2,4,8,16")
ap.add_argument("--dataset", default="last-fm_50core")
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--fanout", type=int, default=32)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--only", choices=["both", "enc", "plain"], default="both", help="adjacency form(s) of the single launch")
ap.add_argument("--feeds", default="pairs,users")
ap.add_argument("--no-multi", action="store_true", help="skip the multi-launch reference (no score comparison)")
a = ap.parse_args()

ds = a.dataset
d = synth.DATASETS[ds]
dev = torch.device("cuda:0")


def timed(fn, steps):
    for _ in range(10):
        o = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        o = fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    evs = []
    for _ in range(20):                      # one launch at a time: the kernel's own duration
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = fn()
        e1.record()
        torch.cuda.synchronize()
        evs.append(e0.elapsed_time(e1))
    evs.sort()
    return wall, evs[len(evs) // 2] * 1e-3, o


for B in [int(x) for x in a.sizes.split(",")]:
    args = make_args(dataset=ds, dim=a.dim, neighbor_sample_size=a.fanout, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"],
                     n_memory=d["n_memory"], batch_size=B)
    case = synth.dataset_case(ds, K=a.fanout, B=B, seed=0)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
    users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    uts = torch.from_numpy(case.user_triplet_set).to(dev)
    m = synth.memories_for(case.user_triplet_set, case.users)
    mh, mr, mt = [[torch.from_numpy(x).to(dev) for x in lst] for lst in m]
    feeds = {"pairs": lambda: model.forward_device(users, items, mh, mr, mt), "users": lambda: model.forward_users(users, items, uts)}
    feeds = {k: v for k, v in feeds.items() if k in a.feeds.split(",")}
    model.small_max_batch = 0
    ref = {}
    for name, fn in ({} if a.no_multi else feeds).items():
        wall, ev, o = timed(fn, a.steps)
        ref[name] = o.scores.clone()
        print(f"B={B:6d} {name:6s} multi-launch        wall {1e6 * wall:8.1f} us  events {1e6 * ev:8.1f} us  {B / wall / 1e6:7.2f} M pairs/s", flush=True)
    model.small_max_batch = 1 << 30
    for G in [int(x) for x in a.groups.split(",")]:
        model.small_group = G
        for dedup in {"both": (None, False), "enc": (None,), "plain": (False,)}[a.only]:
            model.dedup = dedup
            for name, fn in feeds.items():
                wall, ev, o = timed(fn, a.steps)
                err = float((o.scores - ref[name]).abs().max()) if name in ref else float("nan")
                print(f"B={B:6d} {name:6s} single launch G={G:2d} {'enc  ' if dedup is None else 'plain'} wall {1e6 * wall:8.1f} us  events {1e6 * ev:8.1f} us  "
                      f"{B / wall / 1e6:7.2f} M pairs/s   max|d| {err:.2e}", flush=True)
