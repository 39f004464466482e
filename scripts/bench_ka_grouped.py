#!/usr/bin/env python3
"""us per mvin_key_addressing_grouped_fwd call on a dataset-shaped batch (users feed): scripts/bench_ka_grouped.py [dim] [dataset]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import ops, synth
D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ds = sys.argv[2] if len(sys.argv) > 2 else "last-fm_50core"
d = synth.DATASETS[ds]
dev = torch.device("cuda:0"); B, P, Nm, nR = 524288, d["p_hop"], d["n_memory"], d["n_relation"]
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.rand((d["n_entity"], D), device=dev, generator=g) - 0.5
R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
w = torch.rand(D, device=dev, generator=g)
uts = torch.from_numpy(synth.ripple_sets(d["n_user"], d["n_entity"], nR, P, Nm, seed=3)).to(dev)
users = torch.randint(0, d["n_user"], (B,), device=dev, generator=g)
items = torch.randint(0, d["n_item"], (B,), device=dev, generator=g)
out = torch.empty((B, (P + 1) * D), device=dev)
groups = ops.group_pairs_by_user(users, n_user=d["n_user"])
for _ in range(3):
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, (P + 1) * D, nR)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, (P + 1) * D, nR)
e1.record(); torch.cuda.synchronize()
print(f"{ds} D={D} P={P} Nm={Nm} nR={nR}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call  (MVIN_KA_WAVE_DBG={os.environ.get('MVIN_KA_WAVE_DBG', '0')}, MVIN_KA_WAVE={os.environ.get('MVIN_KA_WAVE', '1')}, MVIN_KA_WAVE32={os.environ.get('MVIN_KA_WAVE32', '1')})", flush=True)
