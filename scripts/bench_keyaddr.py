#!/usr/bin/env python3
"""Micro-benchmark of mvin_key_addressing_fwd alone (uniform ripple sets), table-size sweep."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=64); ap.add_argument("--Nm", type=int, default=64)
ap.add_argument("--P", type=int, default=2); ap.add_argument("--B", type=int, default=131072)
ap.add_argument("--nR", type=int, default=9)
ap.add_argument("--nE", type=str, default="1000,106389,16000000"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(0)
D, Nm, P, B, nR = a.D, a.Nm, a.P, a.B, a.nR
for nE in [int(x) for x in a.nE.split(",")]:
    E = torch.rand((nE, D), device=dev, generator=g) - 0.5
    V = torch.rand((B, nR, D), device=dev, generator=g) - 0.5
    w = torch.rand(2 * D, device=dev, generator=g) - 0.5
    mh = [torch.randint(0, nE, (B, Nm), device=dev, generator=g, dtype=torch.int32) for _ in range(P)]
    mt = [torch.randint(0, nE, (B, Nm), device=dev, generator=g, dtype=torch.int32) for _ in range(P)]
    mr = [torch.randint(0, nR, (B, Nm), device=dev, generator=g, dtype=torch.int32) for _ in range(P)]
    out = torch.empty((B, (P + 1) * D), device=dev)
    for _ in range(2): ops.key_addressing(E, V, w, mh, mr, mt, P, out, (P + 1) * D, nR)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): ops.key_addressing(E, V, w, mh, mr, mt, P, out, (P + 1) * D, nR)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    alg = B * (2 * P * Nm * D * 4 + 3 * P * Nm * 4)
    print(json.dumps({"nE": nE, "ms": round(ms, 4), "alg_GBs": round(alg / ms / 1e6, 1), "pairs_per_s": round(B / ms * 1e3)}), flush=True)
    del E, V, mh, mt, mr
