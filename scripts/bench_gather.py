#!/usr/bin/env python3
"""Micro-benchmark of mvin_gather_attn_fwd alone: sweep the table size (cache-resident ->
HBM-bound) and report algorithmic GB/s of the row gather.  Run on the GPU box."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=64); ap.add_argument("--K", type=int, default=32)
ap.add_argument("--B", type=int, default=16384); ap.add_argument("--N", type=int, default=32)
ap.add_argument("--nE", type=str, default="1000,10000,106389,1000000,16000000")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--noproj", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
D, K, B, N = a.D, a.K, a.B, a.N
for nE in [int(x) for x in a.nE.split(",")]:
    table = torch.rand((nE, D), device=dev, generator=g) - 0.5
    adj_e = torch.randint(0, nE, (nE, K), device=dev, generator=g, dtype=torch.int32)
    adj_r = torch.randint(0, 9, (nE, K), device=dev, generator=g, dtype=torch.int32)
    node = torch.randint(0, nE, (B * N,), device=dev, generator=g, dtype=torch.int32)
    t = torch.rand(9, device=dev, generator=g)
    selfv = torch.rand((B * N, D), device=dev, generator=g)
    W = torch.rand((D, D), device=dev, generator=g) - 0.5
    c = torch.rand((B, D), device=dev, generator=g)
    bias = torch.zeros(D, device=dev)
    for _ in range(2):
        ops.gather_attn(table, adj_e, adj_r, node, t, selfv, None if a.noproj else W, None if a.noproj else c, W, bias, B, N, K, D)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.gather_attn(table, adj_e, adj_r, node, t, selfv, None if a.noproj else W, None if a.noproj else c, W, bias, B, N, K, D)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    rows_bytes = B * N * K * D * 4
    all_bytes = rows_bytes + B * N * (K * 8 + 4 + D * 4 * 2)
    print(json.dumps({"nE": nE, "table_MB": nE * D * 4 / 1e6, "ms": ms, "row_GBs": rows_bytes / ms / 1e6,
                      "all_GBs": all_bytes / ms / 1e6, "tasks_per_us": B * N / ms / 1e3}), flush=True)
    del table, adj_e, adj_r, node, selfv
