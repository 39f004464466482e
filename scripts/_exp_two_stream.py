import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params
ds="last-fm_50core"; d=synth.DATASETS[ds]; dev=torch.device("cuda:0"); B=int(sys.argv[1]) if len(sys.argv) > 1 else 524288
args=make_args(dataset=ds, dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=B)
case=synth.dataset_case(ds, K=32, B=B, seed=0)
params=init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
model=MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
uts=torch.from_numpy(case.user_triplet_set).to(dev)
def run(nstreams, steps=200 if B <= 65536 else 40):
    ss=[torch.cuda.Stream() for _ in range(nstreams)] if nstreams>1 else [torch.cuda.current_stream()]
    for s in ss:
        with torch.cuda.stream(s):
            for _ in range(3): model.forward_users(users, items, uts)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(ss[i%len(ss)]):
            model.forward_users(users, items, uts)
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/steps*1e3
import sys
ST=int(sys.argv[2]) if len(sys.argv)>2 else None
for n in (1,2,3,1,2):
    print(B, n, "streams:", round(1e3*(run(n, ST) if ST else run(n)),1), "us/step")
