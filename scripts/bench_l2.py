#!/usr/bin/env python3
"""Micro-benchmark of mvin_gather_attn_l2_fwd alone: sweep the table size (cache-resident ->
HBM-bound), uniform adjacency.  Run on the GPU box."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=64); ap.add_argument("--K", type=int, default=32)
ap.add_argument("--B", type=int, default=16384)
ap.add_argument("--nE", type=str, default="1000,106389,1000000,16000000")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--noproj", action="store_true"); ap.add_argument("--noatt", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
D, K, B = a.D, a.K, a.B
for nE in [int(x) for x in a.nE.split(",")]:
    table = torch.rand((nE, D), device=dev, generator=g) - 0.5
    adj_e = torch.randint(0, nE, (nE, K), device=dev, generator=g, dtype=torch.int32)
    adj_r = torch.randint(0, 9, (nE, K), device=dev, generator=g, dtype=torch.int32)
    parents = torch.randint(0, nE, (B,), device=dev, generator=g, dtype=torch.int32)
    t0 = None if a.noatt else torch.rand(9, device=dev, generator=g)
    W = torch.rand((D, D), device=dev, generator=g) - 0.5
    c = torch.rand((B, D), device=dev, generator=g)
    bias = torch.zeros(D, device=dev)
    args = (table, adj_e, adj_r, parents, t0, t0, None if a.noproj else W, None if a.noproj else W,
            None if a.noproj else bias, None if a.noproj else bias, None if a.noproj else c, W, bias, B, 1, K, D, 9)
    for _ in range(2):
        ops.gather_attn_l2(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.gather_attn_l2(*args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    T = 1 + K + K * K
    alg = B * (T * D * 4 + (T - K * K) * K * 8 + D * 4 + 4)
    print(json.dumps({"nE": nE, "table_MB": round(nE * D * 4 / 1e6, 1), "ms": round(ms, 4),
                      "alg_GBs": round(alg / ms / 1e6, 1), "pairs_per_s": round(B / ms * 1e3)}), flush=True)
    del table, adj_e, adj_r
