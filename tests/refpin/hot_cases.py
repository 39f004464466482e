"""Seeded inputs of the reference-run fixtures AT THE SHAPES THE HOT KERNELS SERVE (D in {32, 64}, K in {16, 32, 64}):
shared by tests/golden/make_ref_fixtures.py (which runs the reference's model.py / aggregators.py over them and stores
``ref_scores_{32,64}``) and by the tests that replay them on the oracles and on the HIP path.

The fixture files hold NO inputs (a [3000, 64] entity table per file would be megabytes): inputs are regenerated from the
seeds here and verified against the ``inputs_crc32`` the generator stored, exactly as tests/golden/make_c5_fixture.py does.
Own code; nothing here comes from the reference.
"""
import zlib
from types import SimpleNamespace

import numpy as np

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params

# name -> (model shape, table sizes, adjacency kind).  "repeats": rows as contruct_random_adj builds them for low-degree
# entities (slots repeat: the packed-tile kernel's case), with all-zero rows; "distinct": every row K distinct slots
# (the role-split kernel's case).  B * K^(L-2) = 2 048 parents where the packed kernel's auto rule starts.
HOT_CASES = {
    # C3's shape: the packed / split fused kernels <64,32>, key_addr_static_kernel<2,64,9> (users feed), l2_tail_kernel<64>
    "d64k32p2m64_repeats": dict(shape=dict(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64,
                                           batch_size=64), n_user=8, n_entity=3000, n_relation=9, adj="repeats", seed=3101),
    "d64k32p2m64_distinct": dict(shape=dict(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64,
                                            batch_size=64), n_user=8, n_entity=3000, n_relation=9, adj="distinct", seed=3102),
    # C2's shape: the wave-per-parent kernels <32,16>, key_addr_wave_kernel<32>
    "d32k16p2m64_repeats": dict(shape=dict(dim=32, neighbor_sample_size=16, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64,
                                           batch_size=64), n_user=8, n_entity=2000, n_relation=12, adj="repeats", seed=3103),
    # C4's shape: fused kernels <64,64>, one preference hop of 16 memories, 39 relations
    "d64k64p1m16_repeats": dict(shape=dict(dim=64, neighbor_sample_size=64, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=16,
                                           batch_size=32), n_user=6, n_entity=4000, n_relation=39, adj="repeats", seed=3104),
}
HOT_ABLATIONS = ("all", "no_uor", "no_uo")


def build(name, ablation="all"):
    """-> (args, case, params, user_triplet_set).  The per-pair ripple sets ARE user_triplet_set[users] (train.py:117-120),
    so the same fixture drives the per-pair feed and the users feed."""
    c = HOT_CASES[name]
    args = make_args(**dict(c["shape"], ablation=ablation))
    case = synth.small_case(args, n_user=c["n_user"], n_entity=c["n_entity"], n_relation=c["n_relation"], seed=c["seed"],
                            zero_rows=7 if c["adj"] == "repeats" else 0, repeats=c["adj"] == "repeats")
    if c["adj"] == "distinct":
        rng = np.random.default_rng(c["seed"] + 1)
        K = args.neighbor_sample_size
        case.adj_entity = np.stack([rng.choice(c["n_entity"], K, replace=False) for _ in range(c["n_entity"])]).astype(np.int64)
    uts = synth.ripple_sets(c["n_user"], c["n_entity"], c["n_relation"], args.p_hop, args.n_memory, seed=c["seed"] + 2)
    case.memories_h, case.memories_r, case.memories_t = synth.memories_for(uts, case.users)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=c["seed"] + 100, random_agg_bias=True)
    return args, case, params, uts


def inputs_crc32(case, params):
    h = 0
    for arr in (case.adj_entity, case.adj_relation, case.users, case.items, *case.memories_h, *case.memories_r,
                *case.memories_t, *(params[k] for k in sorted(params))):
        h = zlib.crc32(np.ascontiguousarray(arr).tobytes(), h)
    return h


def expected(path):
    z = np.load(path)
    return SimpleNamespace(scores_32=z["ref_scores_32"], scores_64=z["ref_scores_64"], crc=int(z["inputs_crc32"]),
                           sig_64=z["ref_scores_normalized_64"])
