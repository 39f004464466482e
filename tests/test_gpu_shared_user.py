"""-m gpu: shared-user form of the scoring path (one user's ripple sets for the whole batch,
MVIN._key_addressing_shared / DeviceFeeder.scores_user) against the per-pair form fed with the same
sets replicated, and against the oracle."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import ABLATIONS, make_args
from mvin_amd.params import init_params

from parity import assert_close, run_oracles

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=9),
    dict(dim=16, neighbor_sample_size=8, h_hop=1, n_mix_hop=1, p_hop=1, n_memory=16, batch_size=37),
    dict(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=70),
    dict(dim=64, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=16, batch_size=33),
    dict(dim=32, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=3, n_memory=70, batch_size=5),
    dict(dim=12, neighbor_sample_size=5, h_hop=1, n_mix_hop=2, p_hop=2, n_memory=7, batch_size=6),
]


def one_user_case(args, seed, n_entity=300):
    """A batch in which every pair belongs to user 3 and carries that user's ripple sets."""
    case = synth.small_case(args, n_user=16, n_entity=n_entity, n_relation=7, seed=seed)
    B = case.users.shape[0]
    case.users[:] = 3
    for lst in (case.memories_h, case.memories_r, case.memories_t):
        for i in range(len(lst)):
            lst[i] = np.ascontiguousarray(np.repeat(lst[i][:1], B, axis=0))
    return case


def run(args, case, params, shared, hoist=False, table_dtype="f32"):
    from mvin_amd.model import MVIN
    m = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
             device="cuda:0", hoist=hoist, table_dtype=table_dtype)
    dev = m.device
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    if shared:
        out = m.forward_device(d(case.users[:1]), d(case.items), [d(x[0]) for x in case.memories_h],
                               [d(x[0]) for x in case.memories_r], [d(x[0]) for x in case.memories_t])
    else:
        out = m.forward_device(d(case.users), d(case.items), [d(x) for x in case.memories_h],
                               [d(x) for x in case.memories_r], [d(x) for x in case.memories_t])
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "D{dim}P{p_hop}Nm{n_memory}".format(**s))
def test_shared_user_matches_oracle_and_per_pair(shape, hip_lib):
    args = make_args(**shape)
    case = one_user_case(args, seed=13)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=14, random_agg_bias=True)
    mir, _ = run_oracles(args, case, params)
    got = run(args, case, params, shared=True)
    ref = run(args, case, params, shared=False)
    assert_close(got.scores.cpu().numpy(), mir.scores.numpy(), "shared-user scores vs fp32 mirror")
    assert_close(got.user_o.cpu().numpy(), mir.user_o.numpy(), "shared-user user_o vs fp32 mirror")
    assert_close(got.scores.cpu().numpy(), ref.scores.cpu().numpy(), "shared-user vs per-pair HIP path")


@pytest.mark.parametrize("ablation", sorted(ABLATIONS))
def test_shared_user_every_ablation(ablation, hip_lib):
    args = make_args(ablation=ablation, **SHAPES[0])
    case = one_user_case(args, seed=5, n_entity=64)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=7, random_agg_bias=True)
    mir, _ = run_oracles(args, case, params)
    got = run(args, case, params, shared=True)
    assert_close(got.scores.cpu().numpy(), mir.scores.numpy(), f"{ablation}: shared-user scores vs fp32 mirror")


def test_shared_user_with_hoist_and_bf16(hip_lib):
    args = make_args(**SHAPES[2])
    case = one_user_case(args, seed=23, n_entity=2000)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=24, random_agg_bias=True)
    ref = run(args, case, params, shared=False)
    got = run(args, case, params, shared=True, hoist=True)
    assert_close(got.scores.cpu().numpy(), ref.scores.cpu().numpy(), "shared-user + entity tables vs faithful")
    a = run(args, case, params, shared=False, table_dtype="bf16")
    b = run(args, case, params, shared=True, table_dtype="bf16")
    assert_close(b.scores.cpu().numpy(), a.scores.cpu().numpy(), "bf16 table: shared-user vs per-pair", rtol=2e-5,
                 atol=2e-6)


def test_feeder_scores_user(hip_lib):
    from mvin_amd import harness
    from mvin_amd.model import MVIN
    args = make_args(**SHAPES[3])
    case = synth.small_case(args, n_user=16, n_entity=300, n_relation=7, seed=3)
    rng = np.random.default_rng(0)
    uts = rng.integers(0, 7, (case.n_user, 1, 3, 16)).astype(np.int32)
    uts[:, :, 0] = rng.integers(0, 300, (case.n_user, 1, 16))
    uts[:, :, 2] = rng.integers(0, 300, (case.n_user, 1, 16))
    m = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, device="cuda:0",
             seed=2)
    f = harness.DeviceFeeder(m, uts)
    items = np.arange(100, 180)
    a = f.scores_user(5, items).cpu().numpy()
    b = f.scores(np.full(items.size, 5), items).cpu().numpy()
    assert_close(a, b, "scores_user vs scores")
