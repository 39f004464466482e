"""-m gpu: dataset-sized runs (BASELINE.json configs C1-C4 on one GPU) checked against the
oracle on a sample and through size-independent properties."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from oracle import mirror_fp32

from parity import assert_close

pytestmark = pytest.mark.gpu

CONFIGS = {
    "C1": ("MovieLens-1M", dict(dim=16, h_hop=1, neighbor_sample_size=8), 1024),
    "C2": ("MovieLens-1M", dict(dim=32, h_hop=2, neighbor_sample_size=16), 1024),
    "C3": ("last-fm_50core", dict(dim=64, h_hop=2, neighbor_sample_size=32), 512),
    "C4": ("amazon-book_20core", dict(dim=64, h_hop=2, neighbor_sample_size=64), 128),
}


def setup(name, B=None, **extra):
    from mvin_amd.model import MVIN
    ds, kw, B0 = CONFIGS[name]
    B = B or B0
    d = synth.DATASETS[ds]
    args = make_args(dataset=ds, n_mix_hop=1, p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=B,
                     **kw, **extra)
    case = synth.dataset_case(ds, K=kw["neighbor_sample_size"], B=B, seed=5)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=6, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                 params=params, device="cuda:0")
    return args, case, params, model


def run(model, case, sl=slice(None)):
    dev = model.device
    out = model.forward_device(torch.from_numpy(case.users[sl]).to(dev), torch.from_numpy(case.items[sl]).to(dev),
                               [torch.from_numpy(m[sl]).to(dev) for m in case.memories_h],
                               [torch.from_numpy(m[sl]).to(dev) for m in case.memories_r],
                               [torch.from_numpy(m[sl]).to(dev) for m in case.memories_t], want_probs=True)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_size_against_oracle_sample(name, hip_lib):
    args, case, params, model = setup(name)
    out = run(model, case)
    n = {"C1": 256, "C2": 64, "C3": 32, "C4": 8}[name]   # what the CPU oracle finishes in seconds
    sargs = make_args(**dict(vars(args), batch_size=n))
    sl = slice(0, n)
    ref = mirror_fp32.forward(sargs, params, case.adj_entity, case.adj_relation, case.users[sl], case.items[sl],
                              [m[sl] for m in case.memories_h], [m[sl] for m in case.memories_r],
                              [m[sl] for m in case.memories_t])
    assert_close(out.scores[sl].cpu().numpy(), ref.scores.numpy(), f"{name} scores")
    assert_close(out.importance_list[0][sl].cpu().numpy(), ref.importance_list[0].numpy(), f"{name} probs hop0")
    # batch independence: a pair's score does not depend on what else is in the batch
    out2 = run(model, case, slice(3, 3 + n))
    assert torch.equal(out2.scores, out.scores[3:3 + n])
    # attention weights are a distribution over the K neighbors
    for p in out.importance_list:
        assert torch.allclose(p.sum(-1), torch.ones_like(p.sum(-1)), atol=1e-5)


def test_c3_permuting_children_leaves_scores_unchanged(hip_lib):
    from mvin_amd.model import MVIN
    args, case, params, model = setup("C3", B=256)
    ref = run(model, case).scores.cpu().numpy()
    rng = np.random.default_rng(2)
    K = args.neighbor_sample_size
    perm = np.argsort(rng.random((case.n_entity, K)), axis=1)
    adj_e = np.take_along_axis(case.adj_entity, perm, axis=1)
    adj_r = np.take_along_axis(case.adj_relation, perm, axis=1)
    model2 = MVIN(args, case.n_user, case.n_entity, case.n_relation, adj_e, adj_r, params=params, device="cuda:0")
    got = run(model2, case).scores.cpu().numpy()
    assert_close(got, ref, "scores after permuting every node's children")


def test_c3_identical_relations_give_uniform_attention(hip_lib):
    from mvin_amd.model import MVIN
    args, case, params, model = setup("C3", B=64)
    adj_r = np.full_like(case.adj_relation, 3)
    model2 = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, adj_r, params=params,
                  device="cuda:0")
    out = run(model2, case)
    for p in out.importance_list:
        assert torch.allclose(p, torch.full_like(p, 1.0 / args.neighbor_sample_size), atol=1e-7)
    # and the no-attention ablation then differs from it only by the extra 1/K of the softmax weights
    # (aggregators.py:144 vs :150) -- so the two runs must NOT be equal
    args3 = make_args(**dict(vars(args), ablation="no_uor", User_orient_rela=0))
    model3 = MVIN(args3, case.n_user, case.n_entity, case.n_relation, case.adj_entity, adj_r, params=params,
                  device="cuda:0")
    assert not torch.allclose(run(model3, case).scores, out.scores)


def test_c3_urh_user_and_self_slices_cancel(hip_lib):
    args, case, params, model = setup("C3", B=128)
    ref = run(model, case).scores.clone()
    D = args.dim
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for agg in model.aggregators:
        agg.urh_weights[:D] += torch.randn(D, 1, device="cuda:0", generator=g)
        agg.urh_weights[2 * D:] += torch.randn(D, 1, device="cuda:0", generator=g)
        agg.invalidate()
    assert torch.equal(run(model, case).scores, ref)


def test_empty_like_edges(hip_lib):
    # batch of one pair; item whose adjacency row is all zeros; every pair identical
    args, case, params, model = setup("C1", B=64)
    z = np.nonzero((case.adj_entity == 0).all(axis=1))[0]
    case.items[:] = z[0] if len(z) else 0
    case.users[:] = case.users[0]
    for m in case.memories_h + case.memories_r + case.memories_t:
        m[:] = m[0]
    out = run(model, case)
    assert torch.equal(out.scores, out.scores[0].expand_as(out.scores))
    one = run(model, case, slice(0, 1))
    assert torch.equal(one.scores, out.scores[:1])


@pytest.mark.parametrize("name", ["C2", "C3", "C4"])
def test_full_size_entity_table_and_shared_user_modes(name, hip_lib):
    """Dataset-sized tables: the entity-table mode and the shared-user form give the faithful path's
    scores (same tolerance as against the oracle), batch independence holds in both."""
    from mvin_amd.model import MVIN
    args, case, params, model = setup(name, B=2048)
    hm = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
              device="cuda:0", hoist=True)
    dev = model.device
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    feed = (d(case.users), d(case.items), [d(m) for m in case.memories_h], [d(m) for m in case.memories_r],
            [d(m) for m in case.memories_t])
    ref = model.forward_device(*feed).scores
    got = hm.forward_device(*feed).scores
    assert_close(got.cpu().numpy(), ref.cpu().numpy(), f"{name} entity-table mode vs faithful")
    part = hm.forward_device(feed[0][5:105], feed[1][5:105], *[[m[5:105].contiguous() for m in lst] for lst in feed[2:]])
    assert torch.equal(part.scores, got[5:105])
    # one user's ripple sets for every pair: shared-user form == per-pair form fed the replicated sets
    u = int(case.users[0])
    one = lambda lst: [m[0].contiguous() for m in lst]
    rep = lambda lst: [m[:1].expand(m.shape[0], -1).contiguous() for m in lst]
    users = torch.full_like(feed[0], u)
    a = model.forward_device(users, feed[1], rep(feed[2]), rep(feed[3]), rep(feed[4])).scores
    for mdl in (model, hm):
        b = mdl.forward_device(users[:1], feed[1], one(feed[2]), one(feed[3]), one(feed[4])).scores
        assert_close(b.cpu().numpy(), a.cpu().numpy(), f"{name} shared-user form vs per-pair")
