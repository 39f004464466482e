#!/usr/bin/env python3
"""The per-entity aggregates form of the two deepest levels on the C3 bench batch (Zipf items): table build + aggregates + launch against
the wave-per-parent kernel over the projected tables.  Development aid, GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops, synth
D, K = 64, 32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
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
agg = ops.entity_aggregates(ws, enc_e, enc_r, t0, K, D, nR, nE)
order = ops.order_by_key(items)
torch.cuda.synchronize()


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


us, _ = timed(lambda: ops.project_tables(E, W[0], W[1], b[0], b[1], W[2], b[2], K, True, out=ws))
print(f"project_tables:                   {us:8.1f} us")
us, _ = timed(lambda: ops.entity_aggregates(ws, enc_e, enc_r, t0, K, D, nR, nE, out=agg))
print(f"entity_aggregates:                {us:8.1f} us")
us, ref = timed(lambda: ops.gather_attn_l2_prj(ws, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE, order=order))
print(f"wave-per-parent, item order:      {us:8.1f} us")
us, a = timed(lambda: ops.gather_attn_l2_agg(ws, agg, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE))
print(f"aggregates form, as given:        {us:8.1f} us   max |diff| {float((a[0] - ref[0]).abs().max()):.2e} {float((a[1] - ref[1]).abs().max()):.2e}")
us, a = timed(lambda: ops.gather_attn_l2_agg(ws, agg, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE, order=order))
print(f"aggregates form, item order:      {us:8.1f} us   max |diff| {float((a[0] - ref[0]).abs().max()):.2e} {float((a[1] - ref[1]).abs().max()):.2e}")

# ---- folded-tail form against aggregates + tail kernel (everything above key addressing) ----
W0 = (torch.rand((D, D), device=dev, generator=g) - 0.5) / 8
A1 = (torch.rand((D, D), device=dev, generator=g) - 0.5) / 8
Wmix = (torch.rand((3 * D, D), device=dev, generator=g) - 0.5) / 8
uo = torch.rand((B, D), device=dev, generator=g) - 0.5
fw = ops.fold_tables(E, enc_e, enc_r, t0, W0, b[0], W[0], b[0], W[1], b[1], W[2], b[2], Wmix, b[0], A1, K, nR)
us, _ = timed(lambda: ops.fold_tables(E, enc_e, enc_r, t0, W0, b[0], W[0], b[0], W[1], b[1], W[2], b[2], Wmix, b[0], A1, K, nR, out=fw))
print(f"fold_tables (4 tables + H0 | G):  {us:8.1f} us")
us, fo = timed(lambda: ops.score_l2_folded(fw, enc_e, enc_r, items, t0, t1, q, uo, A1, b[1], Wmix, K, D, nR, nE))
print(f"score_l2_folded (2 launches):     {us:8.1f} us")
def unfolded():
    n0, n1 = ops.gather_attn_l2_agg(ws, agg, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, nE)
    return ops.l2_tail(E, items, q, uo, n0, n1, W0, b[0], W[2], b[2], A1, b[1], Wmix, b[0])
us, un = timed(unfolded)
print(f"aggregates launch + l2_tail:      {us:8.1f} us   max |score diff| {float((fo[1] - un[1]).abs().max()):.2e} of {float(un[1].abs().max()):.2e}")
