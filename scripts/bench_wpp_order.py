#!/usr/bin/env python3
"""The wave-per-parent kernel over projected tables on the C3 bench batch (Zipf items): parents as given vs in item order (torch.argsort
here: the experiment measures the KERNEL and, under rocprofv3 --pmc, its bytes past the L2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops, synth
D, K = 64, 32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
case = synth.dataset_case("last-fm_50core", K=K, B=B, seed=0, zipf=True, uniform_adj=False)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
nE, nR = case.n_entity, case.n_relation
E = torch.rand((nE, D), device=dev, generator=g) - 0.5
W = (torch.rand((3, D, D), device=dev, generator=g) - 0.5) / 8
b = torch.rand((3, D), device=dev, generator=g) - 0.5
t0, t1 = torch.rand(nR, device=dev, generator=g), torch.rand(nR, device=dev, generator=g)
q = torch.rand((B, D), device=dev, generator=g) - 0.5
ae = torch.from_numpy(case.adj_entity.astype("int32")).to(dev)
ar = torch.from_numpy(case.adj_relation.astype("int32")).to(dev)
enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
items = torch.from_numpy(case.items).to(dev)
ws = ops.project_tables(E, W[0], W[1], b[0], b[1], W[2], b[2], K, True)
order = torch.argsort(items, stable=True).to(torch.int32)
order_dev = ops.order_by_key(items)
torch.cuda.synchronize()
assert torch.equal(torch.sort(order_dev.long()).values, torch.arange(B, device=dev)), "not a permutation"
kk = items[order_dev.long()] & 16383
assert bool((kk[1:] >= kk[:-1]).all()), "buckets not contiguous"


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


ref = None
if mode in ("both", "plain"):
    us, ref = timed(lambda: ops.gather_attn_l2_prj(ws, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE))
    print(f"parents as given:      {us:8.1f} us")
if mode == "both":
    wso = torch.empty(int(ops._lib.load().mvin_order_by_key_ws_elems(B)), dtype=torch.int32, device=dev)
    us, _ = timed(lambda: ops.order_by_key(items, ws=wso, out=order_dev), n=20)
    print(f"mvin_order_by_key:     {us:8.1f} us")
    us, out = timed(lambda: ops.gather_attn_l2_prj(ws, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE, order=order_dev))
    print(f"parents in bucket order (device): {us:8.1f} us   equal: {torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])}")
if mode in ("both", "order"):
    us, out = timed(lambda: ops.gather_attn_l2_prj(ws, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE, order=order))
    print(f"parents in item order: {us:8.1f} us" + (f"   equal: {torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])}" if ref is not None else ""))
