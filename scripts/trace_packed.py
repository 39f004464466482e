#!/usr/bin/env python3
"""Phase cycle totals of the packed-tile fused kernel (development aid): MVIN_SPLIT_DBG=8 build, C3 bench workload.
Prints mean cycles per tile and phase for the first dense wave and the first front wave of every workgroup."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MVIN_SPLIT_DBG", "8")     # |1: no MFMAs, |2: no row gathers
os.environ["MVIN_PACK_TRACE"] = "1"
from mvin_amd import _lib, ops, synth
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
D, K, B = 64, int(os.environ.get("K", 32)), int(os.environ.get("B", 524288))
case = synth.dataset_case(os.environ.get("DATASET", "last-fm_50core"), K=K, B=B, seed=0, uniform_adj="--uniform" in sys.argv)
nE = case.n_entity
table = torch.rand((nE, D), device=dev, generator=g) - 0.5
adj_e = torch.from_numpy(case.adj_entity.astype(np.int32)).to(dev)
adj_r = torch.from_numpy(case.adj_relation.astype(np.int32)).to(dev)
parents = torch.from_numpy(case.items.astype(np.int32)).to(dev)
if "--sorted" in sys.argv:
    parents = torch.sort(parents).values.contiguous()
enc_e, enc_r, cnt = ops.encode_adjacency(adj_e, adj_r)
t0 = torch.rand(9, device=dev, generator=g); W = torch.rand((D, D), device=dev, generator=g) - 0.5
c = torch.rand((B, D), device=dev, generator=g); bias = torch.zeros(D, device=dev)
args = (table, enc_e, enc_r, parents, t0, t0, W, W, bias, bias, c, W, bias, B, 1, K, D, 9)
run = ops.gather_attn_l2_enc
if "--prj" in sys.argv:                      # the projected-tables form of the same pass
    args = (ops.project_tables(table, W, W, bias, bias, W, bias, K, True), enc_e, enc_r, parents, t0, t0, c, B, 1, K, D, 9, nE)
    run = ops.gather_attn_l2_prj
buf = np.zeros(16, dtype=np.int64)
lib = _lib.load()
for _ in range(2):
    run(*args)
torch.cuda.synchronize()
lib.mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size)      # reads and clears
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(*args); e1.record(); torch.cuda.synchronize()
lib.mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size)
t = buf.reshape(2, 8).astype(np.float64)
print("launch %.3f ms, distinct children per parent %.2f" % (e0.elapsed_time(e1), float(cnt[parents.long()].float().mean())))
for role, names in ((0, ["barrier wait", "phase B", "B->C sync", "phase C + stores"]),
                    (1, ["pack + issue ids", "parent softmax", "child softmax (id wait)", "gather", "barrier wait"])):
    n = t[role, 7]
    print(["dense", "front"][role], "tiles", int(n), "cycles per tile:",
          ", ".join("%s %.0f" % (nm, t[role, i] / max(n, 1)) for i, nm in enumerate(names)),
          "| total %.0f" % (t[role, :5].sum() / max(n, 1)))
