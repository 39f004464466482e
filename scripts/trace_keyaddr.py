#!/usr/bin/env python3
"""Phase timeline of the dense grouped key-addressing kernel (development aid): MVIN_KA_TRACE=1, GPU box."""
import ctypes as C, os, sys
import numpy as np, torch
STATIC = "--static" in sys.argv          # the kernel over static per-user records (mvin_keyaddr_static.hip)
os.environ["MVIN_KA_TRACE"] = "2" if STATIC else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import _lib, ops, synth
dev = torch.device("cuda:0"); B, D, P, Nm, nR = 524288, 64, 2, 64, 9
d = synth.DATASETS["last-fm_50core"]
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.rand((d["n_entity"], D), device=dev, generator=g) - 0.5
R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
w = torch.rand(D, device=dev, generator=g)
uts = torch.from_numpy(synth.ripple_sets(d["n_user"], d["n_entity"], nR, P, Nm, seed=3)).to(dev)
users = torch.randint(0, d["n_user"], (B,), device=dev, generator=g)
items = torch.randint(0, d["n_entity"], (B,), device=dev, generator=g)
out = torch.empty((B, 3 * D), device=dev)
groups = ops.group_pairs_by_user(users)
rec = ops.build_user_records(uts, P, nR, d["n_entity"]) if STATIC else None
er = ops.project_relations(E, R, w) if (STATIC and "--er" in sys.argv) else None      # the gathered form of the U rows
_kag = ops.key_addressing_grouped
ops.key_addressing_grouped = lambda *a_, **k_: _kag(*a_, er=er, **k_)
for _ in range(3):
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, 3 * D, nR, records=rec)
torch.cuda.synchronize()
buf = np.zeros(64 * 16, dtype=np.int64)
assert _lib.load().mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
full = buf.reshape(64, 16).astype(np.float64)
if STATIC:
    # the kernel over static records: three barriers per tile; per wave, own work per phase (phase start -> arrival at its barrier)
    print("cycles per segment %d ; U phase (top barrier -> tiles' first barrier) %d ; tile 0: logits %d softmax %d reads %d" % (
        np.mean(np.diff(full[4:60, 0])), (full[4:60, 5] - full[4:60, 13]).mean(), (full[4:60, 6] - full[4:60, 5]).mean(),
        (full[4:60, 7] - full[4:60, 6]).mean(), (full[4:60, 8] - full[4:60, 7]).mean()))
    print("per wave: [top barrier -> U work starts] [-> own U tiles / h-set read done] [-> tail rows written: phase work done] "
          "[logits: own work] [softmax: own work] [reads: own work] [tile 0 end -> next top]")
    for wv in range(12):
        os.environ["MVIN_KA_TRACE_WAVE"] = str(wv)
        for _ in range(2):
            ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, 3 * D, nR, records=rec)
        torch.cuda.synchronize()
        assert _lib.load().mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
        f = buf.reshape(64, 16).astype(np.float64)[4:60]
        u_done = (f[:, 9] - f[:, 3]).mean() if wv < 11 else float("nan")     # (wave 11: the h-set read, no stamp of its own)
        print("  wave %2d: %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f" % (wv, (f[:, 3] - f[:, 13]).mean(), u_done, (f[:, 4] - f[:, 3]).mean(),
                                                                  (f[:, 10] - f[:, 5]).mean(), (f[:, 11] - f[:, 6]).mean(), (f[:, 8] - f[:, 7]).mean(),
                                                                  (f[1:, 0] - f[:-1, 8]).mean()))
    sys.exit(0)
print("U tiles of wave 0 done (cycles since 'rows->LDS'):", round(float(np.mean(full[8:56, 9] - full[8:56, 3]))))
t = full[:, :9]
names = ["top", "ids+rank", "tile table", "rows->LDS", "U + hset", "tile0: Ei", "tile0: logits", "tile0: softmax", "tile0: reads"]
print("cycles per segment:", round(float(np.mean(np.diff(t[4:60, 0])))))
dd = np.diff(t[4:60], axis=1)
for i in range(8):
    print("   %-14s -> %-14s %8.0f" % (names[i], names[i + 1], dd[:, i].mean()))
ex = full[4:60]
print("   top -> top barrier released %6.0f ; -> DMA / clears issued %6.0f" % ((ex[:, 13] - ex[:, 0]).mean(), (ex[:, 14] - ex[:, 13]).mean()))
print("   U done -> tile0 barrier + Ei written %6.0f ; -> head-row DMA issued %6.0f ; -> barrier %6.0f" % (
    (ex[:, 10] - ex[:, 4]).mean(), (ex[:, 11] - ex[:, 10]).mean(), (ex[:, 5] - ex[:, 11]).mean()))

# which wave is late where: the same stamps taken by each wave of workgroup 0 in turn (MVIN_KA_TRACE_WAVE)
print("per wave: [top -> top barrier released] [U phase start -> own U work done] [U done -> tile0 first barrier + Ei] [tile0 reads done -> next top]")
for wv in range(12):
    os.environ["MVIN_KA_TRACE_WAVE"] = str(wv)
    for _ in range(2):
        ops.key_addressing_grouped(E, R, w, uts, groups, items, P, out, 3 * D, nR, records=rec)
    torch.cuda.synchronize()
    assert _lib.load().mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    f = buf.reshape(64, 16).astype(np.float64)[4:60]
    print("  wave %2d: %6.0f %6.0f %6.0f %6.0f" % (wv, (f[:, 13] - f[:, 0]).mean(), (f[:, 4] - f[:, 3]).mean(), (f[:, 10] - f[:, 4]).mean(),
                                                (f[1:, 0] - f[:-1, 8]).mean()))
