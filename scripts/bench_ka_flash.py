#!/usr/bin/env python3
"""us per call of the grouped key-addressing forms on a dataset-shaped batch (users feed), each with what it needs around it:
  rec    mvin_key_addressing_grouped_rec_fwd (static records) + the user MLP (mvin_linear_fwd)
  er     mvin_project_relations + mvin_key_addressing_grouped_er_fwd + the user MLP
  flash  mvin_key_addressing_flash_prepare + mvin_key_addressing_flash_fwd (user MLP inside)
usage: scripts/bench_ka_flash.py [dataset] [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops, synth
D = 64
ds = sys.argv[1] if len(sys.argv) > 1 else "last-fm_50core"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
d = synth.DATASETS[ds]
dev = torch.device("cuda:0"); P, Nm, nR, nE, nU = d["p_hop"], d["n_memory"], d["n_relation"], d["n_entity"], d["n_user"]
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.rand((nE, D), device=dev, generator=g) - 0.5
R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
w = torch.rand(D, device=dev, generator=g)
W = torch.rand(((P + 1) * D, D), device=dev, generator=g) - 0.5
b = torch.rand(D, device=dev, generator=g)
uts = torch.from_numpy(synth.ripple_sets(nU, nE, nR, P, Nm, seed=3)).to(dev)
users = torch.randint(0, nU, (B,), device=dev, generator=g)
items = torch.randint(0, d["n_item"], (B,), device=dev, generator=g)
out = torch.empty((B, (P + 1) * D), device=dev)
groups = ops.group_pairs_by_user(users, n_user=nU)
rec = ops.build_user_records(uts, P, nR, nE)
er = ops.project_relations(E, R, w)
sched = torch.empty(int(ops._lib.load().mvin_key_addressing_flash_ws_elems(B, nU)), dtype=torch.int32, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


forms = {}
if ops.user_records_supported(D, P, Nm, nR):
    forms["rec + mlp"] = lambda: (ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, (P + 1) * D, nR, records=rec), ops.linear([out], W, D, bias=b))
    forms["rec alone"] = lambda: ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, (P + 1) * D, nR, records=rec)
    if ops.key_addressing_grouped_er_supported(D, P, Nm, nR, nE, True):
        forms["project + er + mlp"] = lambda: (ops.project_relations(E, R, w, out=er), ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, (P + 1) * D, nR, records=rec, er=er), ops.linear([out], W, D, bias=b))
forms["project_relations alone"] = lambda: ops.project_relations(E, R, w, out=er)
tabs = ops.key_addressing_flash_prepare(E, R, w, W, P)
uo = torch.empty((B, D), device=dev)
forms["flash prepare alone"] = lambda: ops.key_addressing_flash_prepare(E, R, w, W, P, out=tabs)
forms["flash alone"] = lambda: ops.key_addressing_flash(E, tabs, rec, groups, items, P, Nm, nR, True, b, nU, sched_ws=sched, out=uo)
forms["prepare + flash"] = lambda: (ops.key_addressing_flash_prepare(E, R, w, W, P, out=tabs), ops.key_addressing_flash(E, tabs, rec, groups, items, P, Nm, nR, True, b, nU, sched_ws=sched, out=uo))
only = os.environ.get("ONLY")
for name, fn in forms.items():
    if only and only not in name:
        continue
    print(f"{ds} B={B} P={P} Nm={Nm} nR={nR}: {name:28s} {timed(fn):9.1f} us", flush=True)
