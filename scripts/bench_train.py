#!/usr/bin/env python3
"""Time the GPU training step (forward + loss + backward + Adam) on a dataset-shaped synthetic
workload; not the headline metric (scoring is), reported for completeness."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params
from mvin_amd.training import GraphedTrainer, Trainer
ap = argparse.ArgumentParser()
ap.add_argument("--dataset", default="last-fm_50core"); ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--hop", type=int, default=2); ap.add_argument("--fanout", type=int, default=32)
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--uniform-adj", action="store_true", help="uniform random adjacency (no KG hubs) instead of the heavy-tailed one")
ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (training.GraphedTrainer)")
a = ap.parse_args()
d = synth.DATASETS[a.dataset]
args = make_args(dataset=a.dataset, dim=a.dim, neighbor_sample_size=a.fanout, h_hop=a.hop, n_mix_hop=1, p_hop=d["p_hop"],
                 n_memory=d["n_memory"], batch_size=a.batch, l2_weight=1e-7, l2_agg_weight=1e-7, lr=1e-3)
case = synth.dataset_case(a.dataset, K=a.fanout, B=a.batch, uniform_adj=a.uniform_adj)
params = init_params(args, case.n_user, case.n_entity, case.n_relation)
model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device="cuda:0")
dev = model.device
feed = (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
        torch.from_numpy((np.arange(a.batch) % 2).astype(np.float32)).to(dev),
        [torch.from_numpy(m).to(dev) for m in case.memories_h], [torch.from_numpy(m).to(dev) for m in case.memories_r],
        [torch.from_numpy(m).to(dev) for m in case.memories_t])
tr = Trainer(model)
if a.graph:
    gt = GraphedTrainer(tr, a.batch)
    step = lambda: gt.step(*feed).clone()          # batch copied into the static buffers every step, loss stays on the device
else:
    step = lambda: tr.step(*feed)
losses = [step() for _ in range(2)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): losses.append(step())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
losses = [float(x) for x in losses]
print(json.dumps({"workload": f"{a.dataset} D={a.dim} H={a.hop} K={a.fanout} B={a.batch}", "mode": "hipgraph" if a.graph else "eager", "ms_per_train_step": dt * 1e3,
                  "pairs_per_s": a.batch / dt, "loss_first": losses[0], "loss_last": losses[-1]}))
