#!/usr/bin/env python3
"""Stage timeline of one wave of the flash key-addressing kernel (development aid): MVIN_KAF_TRACE=1, GPU box."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ["MVIN_KAF_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import _lib, ops, synth
dev = torch.device("cuda:0"); B, D, P, Nm, nR = 524288, 64, 2, 64, 9
d = synth.DATASETS["last-fm_50core"]
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.rand((d["n_entity"], D), device=dev, generator=g) - 0.5
R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
w = torch.rand(D, device=dev, generator=g)
W = torch.rand((3 * D, D), device=dev, generator=g) - 0.5
b = torch.rand(D, device=dev, generator=g)
uts = torch.from_numpy(synth.ripple_sets(d["n_user"], d["n_entity"], nR, P, Nm, seed=3)).to(dev)
users = torch.randint(0, d["n_user"], (B,), device=dev, generator=g)
items = torch.randint(0, d["n_item"], (B,), device=dev, generator=g)
groups = ops.group_pairs_by_user(users, n_user=d["n_user"])
rec = ops.build_user_records(uts, P, nR, d["n_entity"])
tabs = ops.key_addressing_flash_prepare(E, R, w, W, P)
for _ in range(3):
    ops.key_addressing_flash(E, tabs, rec, groups, items, P, Nm, nR, True, b, d["n_user"])
torch.cuda.synchronize()
buf = np.zeros(64 * 16, dtype=np.int64)
assert _lib.load().mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
f = buf.reshape(64, 16).astype(np.float64)
ok = (f[:, 0] > 0) & (f[:, 15] > f[:, 0])
f = f[ok][2:50]
names = {0: "slot top", 1: "descriptors", 2: "h-set done, U requested", 3: "hop0 top", 4: "hop0 logits", 5: "hop0 softmax", 6: "hop0 reads",
         7: "hop1 top", 8: "hop1 logits", 9: "hop1 softmax", 10: "hop1 reads", 15: "slot end"}
print("slots traced: %d ; cycles per slot %.0f (s_memtime ticks; 100 MHz counter x ~21-24 = shader cycles if constant-rate)" % (len(f), np.mean(f[:, 15] - f[:, 0])))
names.update({11: "h-set rows requested", 12: "hs values requested", 13: "h-set softmax", 14: "h-set reads + U requested"})
order = [0, 1, 11, 12, 13, 14, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15]
for a_, b_ in zip(order[:-1], order[1:]):
    dlt = f[:, b_] - f[:, a_]
    print("   %-26s -> %-26s %9.1f  (min %.0f max %.0f)" % (names[a_], names[b_], dlt.mean(), dlt.min(), dlt.max()))
