"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on seeded small cases."""
import itertools

import pytest

from mvin_amd import synth
from mvin_amd.config import ABLATIONS, make_args

from parity import check_case

pytestmark = pytest.mark.gpu

SHAPES = [
    # dim, K, H, M, P, Nm, B
    dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=4),
    dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=2, p_hop=1, n_memory=4, batch_size=4),
    dict(dim=16, neighbor_sample_size=8, h_hop=1, n_mix_hop=1, p_hop=1, n_memory=16, batch_size=37),
    dict(dim=16, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=33),
    dict(dim=32, neighbor_sample_size=16, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=19),
    dict(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=9),
    dict(dim=8, neighbor_sample_size=2, h_hop=3, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=5),
    dict(dim=12, neighbor_sample_size=5, h_hop=1, n_mix_hop=2, p_hop=2, n_memory=7, batch_size=6),
    dict(dim=128, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=16, batch_size=3),
    # shapes the fused two-level kernel (mvin_gather_attn_l2_fwd) takes
    dict(dim=64, neighbor_sample_size=64, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=16, batch_size=5),
    dict(dim=32, neighbor_sample_size=8, h_hop=3, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=4),
    dict(dim=16, neighbor_sample_size=4, h_hop=2, n_mix_hop=2, p_hop=2, n_memory=8, batch_size=7),
    dict(dim=128, neighbor_sample_size=16, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=3),
    dict(dim=64, neighbor_sample_size=128, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=2),
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "perlevel"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "D{dim}K{neighbor_sample_size}H{h_hop}M{n_mix_hop}".format(**s))
def test_default_ablation(shape, fused, hip_lib):
    args = make_args(**shape)
    case = synth.small_case(args, n_user=16, n_entity=200, n_relation=7, seed=11, zero_rows=5)
    check_case(args, case, seed=3, fused=fused)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "perlevel"])
@pytest.mark.parametrize("ablation", sorted(ABLATIONS))
def test_every_ablation(ablation, fused, hip_lib):
    # SHAPES[0:2] only reach the per-level kernels (dim=8, K=3); SHAPES[3] and [11] the fused one
    for shape in (SHAPES[0], SHAPES[1], SHAPES[3], SHAPES[11]):
        args = make_args(ablation=ablation, **shape)
        case = synth.small_case(args, seed=5)
        check_case(args, case, seed=7, fused=fused)


def test_per_level_kernel_large_grid_instance(hip_lib):
    """mvin_gather_attn_fwd has two instances: few tiles (child-row loads issued ahead of the softmax) and more than
    1 024 tiles (not).  B = 3 000 pairs x K = 12 children = 36 000 level-1 nodes takes the second one; slices of 500
    pairs take the first; both against the fp32 mirror on a sample."""
    import torch
    from mvin_amd.model import MVIN
    from mvin_amd.params import init_params
    from oracle import mirror_fp32
    from parity import assert_close
    B = 3000
    args = make_args(dim=12, neighbor_sample_size=12, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=4, batch_size=B)
    case = synth.small_case(args, n_user=40, n_entity=900, n_relation=6, seed=21, zero_rows=4)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=22, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device="cuda:0", fused=False)
    dev = model.device

    def run(sl):
        return model.forward_device(torch.from_numpy(case.users[sl]).to(dev), torch.from_numpy(case.items[sl]).to(dev),
                                    [torch.from_numpy(m[sl]).to(dev) for m in case.memories_h],
                                    [torch.from_numpy(m[sl]).to(dev) for m in case.memories_r],
                                    [torch.from_numpy(m[sl]).to(dev) for m in case.memories_t]).scores.cpu().numpy()
    big = run(slice(None))
    for s0 in (0, 2500):
        sl = slice(s0, s0 + 500)
        assert_close(run(sl), big[sl], f"slice {s0}: few-tile instance vs many-tile instance")
    n = 64
    sargs = make_args(**dict(vars(args), batch_size=n))
    ref = mirror_fp32.forward(sargs, params, case.adj_entity, case.adj_relation, case.users[:n], case.items[:n],
                              [m[:n] for m in case.memories_h], [m[:n] for m in case.memories_r],
                              [m[:n] for m in case.memories_t])
    assert_close(big[:n], ref.scores.numpy(), "many-tile instance vs fp32 mirror")
