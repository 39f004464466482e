import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params
ds="last-fm_50core"; dev=torch.device("cuda:0"); B=int(sys.argv[1])
args=make_args(dataset=ds, dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=B)
case=synth.dataset_case(ds, K=32, B=B, seed=0)
params=init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
model=MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
uts=torch.from_numpy(case.user_triplet_set).to(dev)
mh, mr, mt = [[torch.from_numpy(x).to(dev) for x in lst] for lst in synth.memories_for(case.user_triplet_set, case.users)]
for r in bench.batch_sweep(model, users, items, mh, mr, mt, [512], uts=uts):
    print(B, {k: round(v["us_per_step"], 1) for k, v in r.items() if k != "batch"})
