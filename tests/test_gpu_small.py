"""-m gpu: the whole get_scores pass as ONE launch (mvin_score_small_fwd, mvin_score_small.hip) -- what MVIN.forward_device /
forward_users take for batches of at most 16 384 pairs (the reference's own batch sizes: 512 / 1024 per sess.run,
train.py:62-64, util.py:44-56) -- against the fp32 mirror of the reference graph and the fp64 equations, for every group
size, both feeds, encoded and plain adjacencies, the ablation presets of the default wiring, bf16 tables, ragged batches."""
import copy

import numpy as np
import pytest
import torch

from mvin_amd import ops, synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params

from parity import assert_close, run_oracles

pytestmark = pytest.mark.gpu

# (D, K, P, Nm, nR)
SHAPES = [(64, 32, 2, 64, 9), (64, 16, 2, 64, 9), (64, 64, 1, 16, 39), (64, 8, 3, 8, 5), (32, 16, 2, 64, 12), (32, 8, 1, 16, 7),
          (16, 8, 2, 64, 9), (16, 4, 2, 4, 5), (64, 48, 2, 32, 9), (32, 32, 2, 32, 6), (16, 64, 1, 64, 3)]


def _case(D, K, P, Nm, nR, B, kind="repeats", seed=0, n_user=11, n_entity=1500, ablation="all", h_hop=2):
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=h_hop, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B, ablation=ablation)
    case = synth.small_case(args, n_user=n_user, n_entity=n_entity, n_relation=nR, seed=seed,
                            zero_rows=5 if kind == "repeats" else 0, repeats=kind == "repeats")
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=seed + 1)
    case.memories_h, case.memories_r, case.memories_t = synth.memories_for(uts, case.users)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=seed + 2, random_agg_bias=True)
    return args, case, params, uts


def _model(args, case, params, **kw):
    m = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
             device="cuda:0", **kw)
    assert m.small_max_batch == 1024          # the product default; these tests drive the launch at every size it accepts
    m.small_max_batch = 16384
    return m


def _feeds(case, uts, dev):
    users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    mem = [[torch.from_numpy(m).to(dev) for m in lst] for lst in (case.memories_h, case.memories_r, case.memories_t)]
    return users, items, mem, torch.from_numpy(uts).to(dev)


def _took_small(model, B):
    return model._small_state is not None and model._small_state["args"].B == B


def _check(out, m, e, what, rtol=1e-5, atol=1e-6):
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), f"{what}: scores vs fp32 mirror", rtol=rtol, atol=atol)
    assert_close(out.scores_normalized.cpu().numpy(), m.scores_normalized.numpy(), f"{what}: sigmoid", rtol=rtol, atol=atol)
    assert_close(out.user_o.cpu().numpy(), m.user_o.numpy(), f"{what}: user_o", rtol=rtol, atol=atol)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), f"{what}: item_embeddings", rtol=rtol, atol=atol)
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 1e-6, f"{what}: HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}"


@pytest.mark.parametrize("kind", ["repeats", "uniform"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "D%dK%dP%dNm%dnR%d" % s)
def test_small_kernel_vs_oracles(shape, kind, hip_lib):
    D, K, P, Nm, nR = shape
    assert ops.score_small_supported(D, K, P, Nm, nR)
    B = 37 if K <= 32 else 9
    args, case, params, uts = _case(D, K, P, Nm, nR, B, kind=kind, seed=D + K)
    m, e = run_oracles(args, case, params)
    model = _model(args, case, params)
    users, items, mem, uts_d = _feeds(case, uts, model.device)
    for G in (0, 1, 2, 4, 8, 16):
        for dedup in (None, False):
            model.small_group, model.dedup = G, dedup
            out = model.forward_device(users, items, *mem)
            assert _took_small(model, B)
            _check(out, m, e, f"per-pair feed G={G} dedup={dedup}")
            out_u = model.forward_users(users, items, uts_d)
            _check(out_u, m, e, f"users feed G={G} dedup={dedup}")
            assert torch.equal(out.scores, out_u.scores)          # the same reads, the same arithmetic


@pytest.mark.parametrize("ablation", ["no_uor", "no_uo", "no_ps_o_ft", "no_uor_and_no_kg_eh_uo", "no_sw"])
@pytest.mark.parametrize("shape", [(64, 32, 2, 64, 9), (32, 16, 2, 64, 12), (16, 8, 1, 16, 5)], ids=lambda s: "D%dK%d" % s[:2])
def test_small_kernel_ablations(shape, ablation, hip_lib):
    D, K, P, Nm, nR = shape
    args, case, params, uts = _case(D, K, P, Nm, nR, 21, seed=7, ablation=ablation)
    m, e = run_oracles(args, case, params)
    model = _model(args, case, params)
    users, items, mem, uts_d = _feeds(case, uts, model.device)
    out = model.forward_device(users, items, *mem)
    if not args.User_orient_kg_eh:        # the query is the user table row, not user_o: outside the single-launch wiring
        assert not _took_small(model, 21)
    else:
        assert _took_small(model, 21)
    _check(out, m, e, ablation)


@pytest.mark.parametrize("ablation", ["all", "no_uor", "no_uo"])
@pytest.mark.parametrize("shape", [(16, 8, 2, 64, 12), (64, 32, 2, 64, 9), (32, 16, 1, 16, 5)], ids=lambda s: "D%dK%d" % s[:2])
def test_small_kernel_one_hop_trees(shape, ablation, hip_lib):
    """h_hop = 1 (BASELINE configs[0], the reference's plumbing case: dim 16, one hop, fan-out 8): aggregator (0,0) at hop 0
    only, combiner over [ev0 | out0] (model.py:286-317) -- the same launch with args->depth = 1."""
    D, K, P, Nm, nR = shape
    args, case, params, uts = _case(D, K, P, Nm, nR, 29, seed=9, ablation=ablation, h_hop=1)
    m, e = run_oracles(args, case, params)
    model = _model(args, case, params)
    users, items, mem, uts_d = _feeds(case, uts, model.device)
    for G in (0, 1, 4, 16):
        for dedup in (None, False):
            model.small_group, model.dedup = G, dedup
            out = model.forward_device(users, items, *mem)
            assert _took_small(model, 29)
            _check(out, m, e, f"one hop, G={G} dedup={dedup}")
            assert torch.equal(model.forward_users(users, items, uts_d).scores, out.scores)


def test_bf16_tables_keep_the_multi_launch_schedule(hip_lib):
    """The single launch reads fp32 rows through 32-bit buffer offsets: a bf16 table (BASELINE C5's storage) is not its case."""
    args, case, params, uts = _case(64, 32, 2, 64, 9, 19, seed=3)
    rounded = dict(params, entity_emb_matrix=torch.from_numpy(params["entity_emb_matrix"]).to(torch.bfloat16).float().numpy())
    m, e = run_oracles(args, case, rounded)
    model = _model(args, case, params, table_dtype="bf16")
    users, items, mem, uts_d = _feeds(case, uts, model.device)
    out = model.forward_device(users, items, *mem)
    assert not _took_small(model, 19)
    _check(out, m, e, "bf16 table")


@pytest.mark.parametrize("B", [1, 2, 15, 16, 17, 255, 1000])
def test_small_kernel_ragged_batches_equal_the_multi_launch_schedule(B, hip_lib):
    """HIP vs HIP (not counted as parity): the single launch against the five-launch native schedule on the same batch, plus
    the mirror on a slice."""
    args, case, params, uts = _case(64, 32, 2, 64, 9, B, seed=B, n_user=50, n_entity=4000)
    model = _model(args, case, params)
    users, items, mem, uts_d = _feeds(case, uts, model.device)
    out = model.forward_device(users, items, *mem)
    assert _took_small(model, B)
    model.small_max_batch = 0
    ref = model.forward_device(users, items, *mem)
    assert torch.allclose(out.scores, ref.scores, rtol=1e-5, atol=1e-6)
    assert torch.allclose(out.item_embeddings, ref.item_embeddings, rtol=1e-5, atol=2e-6)
    n = min(B, 24)
    sub = copy.copy(case)
    sub.users, sub.items = case.users[:n], case.items[:n]
    sub.memories_h, sub.memories_r, sub.memories_t = ([x[:n] for x in lst] for lst in (case.memories_h, case.memories_r, case.memories_t))
    m, e = run_oracles(make_args(**dict(vars(args), batch_size=n)), sub, params)
    assert_close(out.scores[:n].cpu().numpy(), m.scores.numpy(), "scores vs fp32 mirror")


def test_small_kernel_is_deterministic_and_graph_capturable(hip_lib):
    from mvin_amd.graph import GraphedScorer
    B = 512
    args, case, params, uts = _case(64, 32, 2, 64, 9, B, seed=12, n_user=100, n_entity=5000)
    model = _model(args, case, params)
    users, items, mem, uts_d = _feeds(case, uts, model.device)
    a = model.forward_device(users, items, *mem).scores.clone()
    for _ in range(3):
        assert torch.equal(model.forward_device(users, items, *mem).scores, a)
    sc = GraphedScorer(model, B)
    sc.load(users, items, *mem)
    assert torch.equal(sc.replay().scores, a)
    assert torch.equal(sc.replay().scores, a)


def test_small_kernel_clamps_out_of_range_ids(hip_lib):
    """Device feeds are not validated per batch: ids beyond a table clamp to its last row (as in every other kernel)."""
    args, case, params, uts = _case(64, 32, 2, 64, 9, 40, seed=5)
    model = _model(args, case, params)
    model._check_uts = lambda t: None
    rng = np.random.default_rng(0)
    bad_uts = uts.copy()
    hit = rng.random(bad_uts.shape) < 0.05
    hit[:, :, 1, :] = False
    bad_uts[hit] += case.n_entity * 5
    bad_items = case.items.copy()
    bad_items[::7] += case.n_entity * 3
    dev = model.device
    users = torch.from_numpy(case.users).to(dev)
    got = model.forward_users(users, torch.from_numpy(bad_items).to(dev), torch.from_numpy(bad_uts).to(dev))
    ref = model.forward_users(users, torch.from_numpy(np.minimum(bad_items, case.n_entity - 1)).to(dev),
                              torch.from_numpy(np.minimum(bad_uts, case.n_entity - 1)).to(dev))
    assert torch.isfinite(got.scores).all() and torch.equal(got.scores, ref.scores)


@pytest.mark.parametrize("B", [512, 4096, 16384])
def test_small_kernel_at_the_reference_batch_sizes(B, hip_lib):
    """C3's tables (last-fm-shaped, D 64, K 32, P 2, Nm 64) at the batch sizes bench.py sweeps: the single launch vs the
    mirror on 96 pairs sampled across the batch, and vs the multi-launch schedule on all of them (HIP vs HIP)."""
    d = synth.DATASETS["last-fm_50core"]
    args = make_args(dataset="last-fm_50core", dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=B)
    case = synth.dataset_case("last-fm_50core", K=32, B=B, seed=0)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=0)
    model = _model(args, case, params)
    users, items, mem, uts_d = _feeds(case, case.user_triplet_set, model.device)
    out = model.forward_device(users, items, *mem)
    assert _took_small(model, B)
    out_u = model.forward_users(users, items, uts_d)
    assert torch.equal(out.scores, out_u.scores)
    model.small_max_batch = 0
    ref = model.forward_device(users, items, *mem)
    assert torch.allclose(out.scores, ref.scores, rtol=1e-5, atol=1e-6)
    idx = np.random.default_rng(1).choice(B, 96, replace=False)
    sub = copy.copy(case)
    sub.users, sub.items = case.users[idx], case.items[idx]
    sub.memories_h, sub.memories_r, sub.memories_t = ([x[idx] for x in lst] for lst in (case.memories_h, case.memories_r, case.memories_t))
    m, _ = run_oracles(make_args(**dict(vars(args), batch_size=96)), sub, params)
    assert_close(out.scores[torch.from_numpy(idx).to(model.device)].cpu().numpy(), m.scores.numpy(), "sampled scores vs fp32 mirror")
