#!/usr/bin/env python3
"""Stage timeline of the single-launch pass (development aid; GPU box): MVIN_SMALL_DBG=99 makes workgroup 0 stamp s_memtime at
its stage boundaries; prints the per-wave cycle deltas.  usage: scripts/trace_small.py [B] [G] [plain]"""
import ctypes as C, os, sys
os.environ["MVIN_SMALL_DBG"] = "99"
os.environ["MVIN_SMALL_TRACE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import _lib, synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ds = "last-fm_50core"; d = synth.DATASETS[ds]; dev = torch.device("cuda:0")
args = make_args(dataset=ds, dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=B)
case = synth.dataset_case(ds, K=32, B=B, seed=0)
params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
model.small_group = G
model.small_max_batch = 1 << 30
model.dedup = False if "plain" in sys.argv else None
users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
mh, mr, mt = [[torch.from_numpy(x).to(dev) for x in lst] for lst in synth.memories_for(case.user_triplet_set, case.users)]
names = {0: "start", 1: "ids", 2: "item rows / adj / KA ids", 3: "child list", 4: "V", 5: "chunk-0 lists + issue", 6: "reads + combine", 7: "user MLP",
         20: "gather finish", 21: "tile", 22: "W1/W2 product", 23: "A0 product", 8: "row sums (tree end)", 10: "tail X", 11: "ev0", 12: "out0", 13: "out2", 14: "item/score", 19: "end"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 20, 21, 22, 23, 8, 10, 11, 12, 13, 14, 19]
acc = []
for it in range(8):
    model.forward_device(users, items, mh, mr, mt)
    torch.cuda.synchronize()
    buf = np.zeros(4 * 32, dtype=np.int64)
    assert _lib.load().mvin_debug_read_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    if it >= 3:
        acc.append(buf.reshape(4, 32).astype(np.float64))
t = np.mean(acc, 0)
print("stage (cycles since the previous stamp; waves 0..3)          cumulative (wave 0)")
prev = t[:, 0]
for k in order[1:]:
    dlt = t[:, k] - prev
    print("%-28s %7.0f %7.0f %7.0f %7.0f      %8.0f" % (names[k], *dlt, t[0, k] - t[0, 0]))
    prev = t[:, k]
