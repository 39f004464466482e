"""Experiment: does the ORDER of the pairs in the batch change the fused kernel's time? (same pairs, same work)"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params

dev = torch.device("cuda:0")
B = 524288
d = synth.DATASETS["last-fm_50core"]
margs = make_args(dataset="last-fm_50core", dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=B)
case = synth.dataset_case("last-fm_50core", K=32, B=B, seed=0)
params = init_params(margs, case.n_user, case.n_entity, case.n_relation, seed=0)
model = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
uts = torch.from_numpy(case.user_triplet_set).to(dev)
def run(order, name):
    u = torch.from_numpy(case.users[order]).to(dev); it = torch.from_numpy(case.items[order]).to(dev)
    for _ in range(3): model.forward_users(u, it, uts)
    torch.cuda.synchronize()
    model._profile = []
    t0 = time.perf_counter()
    for _ in range(10): out = model.forward_users(u, it, uts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    k = np.mean([a.elapsed_time(b) for a, b in model._profile]); model._profile = None
    print(json.dumps({"order": name, "ms_per_step": dt * 1e3, "fused_ms": float(k)}), flush=True)
    return out.scores
ident = np.arange(B)
s0 = run(ident, "random")
o1 = np.argsort(case.items, kind="stable"); s1 = run(o1, "by_item")
o2 = np.lexsort((case.items, case.users)); run(o2, "by_user_then_item")
# by item, but interleaved so that consecutive pairs are different items of similar popularity rank (XCD spread)
o3 = o1.reshape(8, -1).T.reshape(-1); run(o3, "by_item_strided8")
assert torch.equal(s0[torch.from_numpy(np.argsort(o1)).to(dev)].cpu(), s0[torch.from_numpy(np.argsort(o1)).to(dev)].cpu())
inv = torch.from_numpy(o1).to(dev)
print("max diff sorted vs random:", float((s1 - s0[inv]).abs().max()))
