#!/usr/bin/env python3
"""Phase timeline of the role-split fused kernel (development aid): run with MVIN_SPLIT_DBG=4 (optionally
|1 = no MFMAs, |2 = no row loads) on the GPU box; prints per-step cycle deltas of workgroup 0's dense wave 0
and gather wave 0 (s_memtime ticks = shader cycles)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MVIN_SPLIT_DBG", "4")
from mvin_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
D, K, B, nE = int(os.environ.get("D", 64)), int(os.environ.get("K", 32)), 65536, int(os.environ.get("NE", 106389))
table = torch.rand((nE, D), device=dev, generator=g) - 0.5
if "--kg" in sys.argv:      # the bench.py default workload: synthetic KG adjacency + Zipf items
    from mvin_amd import synth
    case = synth.dataset_case("last-fm_50core", K=K, B=B, seed=0)
    adj_e = torch.from_numpy(case.adj_entity.astype(np.int32)).to(dev)
    adj_r = torch.from_numpy(case.adj_relation.astype(np.int32)).to(dev)
    parents = torch.from_numpy(case.items.astype(np.int32)).to(dev)
else:
    adj_e = torch.randint(0, nE, (nE, K), device=dev, generator=g, dtype=torch.int32)
    adj_r = torch.randint(0, 9, (nE, K), device=dev, generator=g, dtype=torch.int32)
    parents = torch.randint(0, nE, (B,), device=dev, generator=g, dtype=torch.int32)
t0 = torch.rand(9, device=dev, generator=g); W = torch.rand((D, D), device=dev, generator=g) - 0.5
c = torch.rand((B, D), device=dev, generator=g); bias = torch.zeros(D, device=dev)
args = (table, adj_e, adj_r, parents, t0, t0, W, W, bias, bias, c, W, bias, B, 1, K, D, 9)
for _ in range(3):
    ops.gather_attn_l2(*args)
torch.cuda.synchronize()
buf = np.zeros(2 * 64 * 8, dtype=np.int64)
rc = _lib.load().mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size)
assert rc == 0, rc
t = buf.reshape(2, 64, 8)
names = {0: ["top", "B done", "sync done", "C done", "chunk fin", "parent fin", "barrier"],
         1: ["top", "round0", "round1", "-", "-", "done", "barrier"]}
for role, label in ((0, "dense wave 0"), (1, "gather wave 0")):
    x = t[role, 4:60, :7].astype(np.float64)
    d = np.diff(x, axis=1)
    print(label, "mean cycles per step:", round(float(np.mean(x[1:, 0] - x[:-1, 0]))))
    for i in range(6):
        col = d[:, i]
        col = col[(col > 0) & (col < 1e6)]
        if len(col):
            print("   %-10s -> %-10s %8.0f" % (names[role][i], names[role][i + 1], col.mean()))
