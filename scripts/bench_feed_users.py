#!/usr/bin/env python3
"""Whole scoring pass through MVIN.forward_users (bench.py default workload) -- for rocprofv3 kernel stats."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params
B = int(os.environ.get("B", 524288))
d = synth.DATASETS["last-fm_50core"]
args = make_args(dataset="last-fm_50core", dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=B)
case = synth.dataset_case("last-fm_50core", K=32, B=B, seed=0)
params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
dev = torch.device("cuda:0")
model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
uts = torch.from_numpy(case.user_triplet_set).to(dev)
for _ in range(2):
    model.forward_users(users, items, uts)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    model.forward_users(users, items, uts)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
