"""Generate the golden fixtures under tests/golden/ (run from the repo root):

    python tests/golden/make_golden.py

SELF-GENERATED vectors (the reference-produced ones are tests/golden/ref/, make_ref_fixtures.py): the reference has no tests / golden vectors for this path and its
TensorFlow-1.x implementation cannot be imported in this image, so these vectors come from
the repo's own two restatements (oracle/mirror_fp32.py, oracle/equations_fp64.py), which must
agree with each other before a fixture is written.  A fixture is data only: args, inputs
(ids, adjacency, ripple sets, every weight) and expected outputs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mvin_amd import synth  # noqa: E402
from mvin_amd.config import make_args  # noqa: E402
from mvin_amd.params import init_params  # noqa: E402
from oracle import equations_fp64, mirror_fp32  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

SHAPES = {
    "d8k3h2m1p2": dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=4),
    "d8k3h2m2p1": dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=2, p_hop=1, n_memory=4, batch_size=4),
    "d16k8h1m1p1": dict(dim=16, neighbor_sample_size=8, h_hop=1, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=5),
    "d8k2h3m1p2": dict(dim=8, neighbor_sample_size=2, h_hop=3, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=3),
    "d12k5h1m2p2": dict(dim=12, neighbor_sample_size=5, h_hop=1, n_mix_hop=2, p_hop=2, n_memory=6, batch_size=4),
}
ABLS = {
    "d8k3h2m1p2": ["all", "no_kg_eh_uo", "no_uo", "no_uor", "no_wd", "no_ps_o_ft", "ps_only", "ho_only",
                   "ho_only_uo_kg_eh", "no_wd_ho_only", "no_uo_ho_only", "no_uor_ho_only",
                   "no_uo_and_no_kg_eh_uo", "no_uor_and_no_kg_eh_uo"],
    "d8k3h2m2p1": ["all", "no_uor", "no_uo", "ho_only"],
    "d16k8h1m1p1": ["all", "no_kg_eh_uo"],
    "d8k2h3m1p2": ["all", "no_wd"],
    "d12k5h1m2p2": ["all", "no_ps_o_ft"],
}
ARG_KEYS = ("dim", "neighbor_sample_size", "h_hop", "n_mix_hop", "p_hop", "n_memory", "batch_size", "ablation")


def make(shape_name, ablation, seed):
    kw = dict(SHAPES[shape_name], ablation=ablation)
    args = make_args(**kw)
    case = synth.small_case(args, n_user=8, n_entity=64, n_relation=5, seed=seed, zero_rows=3)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=seed + 100,
                         random_agg_bias=True)
    m = mirror_fp32.forward(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                            case.memories_h, case.memories_r, case.memories_t)
    e = equations_fp64.forward(args, params, case.adj_entity, case.adj_relation, case.users, case.items,
                               case.memories_h, case.memories_r, case.memories_t)
    err = np.abs(m.scores.numpy() - e.scores).max()
    assert err < 1e-6, (shape_name, ablation, err)
    d = {"args_json": np.array(json.dumps({k: kw[k] for k in ARG_KEYS})),
         "n_user": case.n_user, "n_entity": case.n_entity, "n_relation": case.n_relation,
         "adj_entity": case.adj_entity, "adj_relation": case.adj_relation,
         "users": case.users, "items": case.items,
         "scores_fp32": m.scores.numpy(), "scores_fp64": e.scores,
         "scores_normalized": m.scores_normalized.numpy(),
         "user_o": m.user_o.numpy(), "item_embeddings": m.item_embeddings.numpy()}
    for i in range(len(case.memories_h)):
        d[f"memories_h_{i}"], d[f"memories_r_{i}"], d[f"memories_t_{i}"] = \
            case.memories_h[i], case.memories_r[i], case.memories_t[i]
    for k, v in params.items():
        d["p__" + k] = v
    for i, x in enumerate(m.entities):
        d[f"entities_{i}"] = x.numpy()
    for i, x in enumerate(m.relations):
        d[f"relations_{i}"] = x.numpy()
    for i, x in enumerate(m.importance_list):
        if x is not None:
            d[f"importance_{i}"] = x.numpy()
    path = os.path.join(OUT, f"{shape_name}__{ablation}.npz")
    np.savez_compressed(path, **d)
    return path, err


if __name__ == "__main__":
    seed = 1000
    for shape_name, abls in ABLS.items():
        for abl in abls:
            seed += 1
            path, err = make(shape_name, abl, seed)
            print(f"{os.path.basename(path):44s} mirror-vs-fp64 {err:.2e}  {os.path.getsize(path)} B")
