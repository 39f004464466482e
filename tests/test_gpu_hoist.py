"""-m gpu: entity-table ("hoisted") mode of the two deepest levels (MVIN(hoist=...),
mvin_gather_mix_fwd) -- same scores as the oracle and as the faithful per-pair gather, cache
invalidation, and the gather-mix primitive on its own."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import ABLATIONS, make_args
from mvin_amd.params import init_params

from parity import assert_close, check_case, run_hip

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=4),
    dict(dim=16, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=33),
    dict(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=9),
    dict(dim=8, neighbor_sample_size=2, h_hop=3, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=5),
    dict(dim=32, neighbor_sample_size=8, h_hop=3, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=4),
    dict(dim=16, neighbor_sample_size=4, h_hop=2, n_mix_hop=2, p_hop=2, n_memory=8, batch_size=7),
    dict(dim=12, neighbor_sample_size=5, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=7, batch_size=6),
    dict(dim=128, neighbor_sample_size=16, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=3),
    dict(dim=64, neighbor_sample_size=128, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=2),
    dict(dim=16, neighbor_sample_size=70, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=3),
]


@pytest.mark.parametrize("mode", [True, "step"], ids=["cached", "step"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "D{dim}K{neighbor_sample_size}H{h_hop}M{n_mix_hop}".format(**s))
def test_hoisted_scores_match_oracle(shape, mode, hip_lib):
    args = make_args(**shape)
    case = synth.small_case(args, n_user=16, n_entity=200, n_relation=7, seed=11, zero_rows=5)
    check_case(args, case, seed=3, hoist=mode)


@pytest.mark.parametrize("ablation", sorted(ABLATIONS))
def test_hoisted_every_ablation(ablation, hip_lib):
    """Ablations the mode does not apply to (PS_only, wide_deep off) must silently take the faithful path."""
    for shape in (SHAPES[0], SHAPES[1], SHAPES[5]):
        args = make_args(ablation=ablation, **shape)
        case = synth.small_case(args, seed=5)
        check_case(args, case, seed=7, hoist=True)


def test_hoisted_vs_faithful_c3_shape(hip_lib):
    """At the metric config's shape the two HIP paths agree far inside the oracle tolerance."""
    args = make_args(dim=64, neighbor_sample_size=32, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=257)
    case = synth.small_case(args, n_user=64, n_entity=3000, n_relation=9, seed=21)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=22, random_agg_bias=True)
    _, a = run_hip(args, case, params, want_probs=False)
    m, b = run_hip(args, case, params, want_probs=False, hoist=True)
    assert m._hoisted is not None
    assert_close(b.scores.cpu().numpy(), a.scores.cpu().numpy(), "hoisted vs faithful scores")
    assert_close(b.item_embeddings.cpu().numpy(), a.item_embeddings.cpu().numpy(), "hoisted vs faithful item_emb")


def test_hoisted_bf16_table(hip_lib):
    args = make_args(dim=32, neighbor_sample_size=8, h_hop=3, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=6)
    case = synth.small_case(args, n_user=16, n_entity=300, n_relation=5, seed=31)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=32, random_agg_bias=True)
    _, a = run_hip(args, case, params, want_probs=False, table_dtype="bf16")
    _, b = run_hip(args, case, params, want_probs=False, table_dtype="bf16", hoist=True)
    assert_close(b.scores.cpu().numpy(), a.scores.cpu().numpy(), "bf16 hoisted vs faithful", rtol=2e-5, atol=2e-6)


def _feed(case, dev):
    return (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
            [torch.from_numpy(m).to(dev) for m in case.memories_h],
            [torch.from_numpy(m).to(dev) for m in case.memories_r],
            [torch.from_numpy(m).to(dev) for m in case.memories_t])


def test_cache_invalidation(hip_lib):
    from mvin_amd.model import MVIN
    args = make_args(dim=16, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=12,
                     lr=1e-2)
    case = synth.small_case(args, n_user=16, n_entity=100, n_relation=5, seed=41)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=42, random_agg_bias=True)
    mk = lambda hoist: MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                            params=params, device="cuda:0", hoist=hoist)
    hm, fm = mk(True), mk(False)
    feed = _feed(case, hm.device)
    s0 = hm.forward_device(*feed).scores.clone()
    tables = hm._hoisted
    hm.forward_device(*feed)
    assert hm._hoisted is tables                      # nothing changed: the tables are re-used
    # a torch-side in-place change of a parameter is seen through the tensor version counter
    for m in (hm, fm):
        m.entity_emb_matrix.mul_(1.5)
    s1 = hm.forward_device(*feed).scores
    assert hm._hoisted is not tables
    assert not torch.allclose(s1, s0)
    assert_close(s1.cpu().numpy(), fm.forward_device(*feed).scores.cpu().numpy(), "after in-place table change")
    # a new adjacency drops the tables
    rng = np.random.default_rng(5)
    adj_e = rng.integers(0, case.n_entity, case.adj_entity.shape)
    for m in (hm, fm):
        m.set_adjacency(adj_e, case.adj_relation)
    assert hm._hoisted is None
    assert_close(hm.forward_device(*feed).scores.cpu().numpy(), fm.forward_device(*feed).scores.cpu().numpy(),
                 "after set_adjacency")
    # an optimizer step (raw-pointer updates) drops them too, and both models stay in step
    fd = lambda m: {m.user_indices: case.users, m.item_indices: case.items,
                    m.labels: (np.arange(case.users.size) % 2).astype(np.float32),
                    **{m.memories_h[i]: case.memories_h[i] for i in range(1)},
                    **{m.memories_r[i]: case.memories_r[i] for i in range(1)},
                    **{m.memories_t[i]: case.memories_t[i] for i in range(1)}}
    hm.forward_device(*feed)
    assert hm._hoisted is not None
    hm.train(None, fd(hm))
    fm.train(None, fd(fm))
    assert hm._hoisted is None
    assert_close(hm.forward_device(*feed).scores.cpu().numpy(), fm.forward_device(*feed).scores.cpu().numpy(),
                 "after a training step")
    # attention outputs are only produced by the faithful path
    out = hm.forward_device(*feed, want_probs=True)
    assert out.importance_list and out.importance_list[0] is not None


@pytest.mark.parametrize("D,K,bf", [(64, 32, False), (16, 5, False), (128, 128, True), (8, 200, False), (36, 7, False)])
def test_gather_mix_primitive(D, K, bf, hip_lib):
    """mvin_gather_mix_fwd against a direct fp64 evaluation of its definition."""
    from mvin_amd import ops
    rng = np.random.default_rng(D * 1000 + K)
    nE, nR, nodes, npg = 500, 6, 77, 7
    dev = torch.device("cuda:0")
    T = rng.standard_normal((nE, D)).astype(np.float32)
    Tt = torch.from_numpy(T).to(dev)
    if bf:
        Tt = Tt.to(torch.bfloat16)
        T = Tt.float().cpu().numpy()
    adj_e = rng.integers(0, nE, (nE, K)).astype(np.int32)
    adj_r = rng.integers(0, nR, (nE, K)).astype(np.int32)
    ids = rng.integers(0, nE, nodes).astype(np.int32)
    t = rng.standard_normal(nR).astype(np.float32)
    bias = rng.standard_normal((nodes // npg, D)).astype(np.float32)
    d = lambda x: torch.from_numpy(x).to(dev)
    for use_t, use_ids, use_bias, relu in [(1, 1, 1, 1), (0, 1, 1, 0), (1, 0, 0, 0), (0, 0, 0, 1)]:
        n = nodes if use_ids else nE
        got = ops.gather_mix(Tt, d(adj_e), d(adj_r), d(ids) if use_ids else None, d(t) if use_t else None,
                             d(bias) if use_bias else None, n, npg, K, nR, relu=bool(relu)).cpu().numpy()
        x = ids.astype(np.int64) if use_ids else np.arange(nE)
        rows = T[adj_e[x]].astype(np.float64)                                     # [n, K, D]
        if use_bias:
            rows = rows + bias[np.arange(n) // npg][:, None, :]
        if relu:
            rows = np.maximum(rows, 0)
        if use_t:
            lg = t[adj_r[x]].astype(np.float64)
            w = np.exp(lg - lg.max(1, keepdims=True))
            w /= w.sum(1, keepdims=True)
        else:
            w = np.ones((n, K))
        ref = (w[:, :, None] * rows).sum(1) / K
        assert_close(got, ref, f"gather_mix t={use_t} ids={use_ids} bias={use_bias} relu={relu}", rtol=1e-5, atol=1e-6)


def test_gather_mix_argument_errors(hip_lib):
    from mvin_amd import _lib, ops
    dev = torch.device("cuda:0")
    T = torch.zeros((10, 8), device=dev)
    adj = torch.zeros((10, 300), dtype=torch.int32, device=dev)
    with pytest.raises(_lib.MvinHipError):
        ops.gather_mix(T, adj, adj, None, None, None, 10, 1, 300, 1)          # K > 256
    adj = torch.zeros((10, 4), dtype=torch.int32, device=dev)
    with pytest.raises(_lib.MvinHipError):
        ops.gather_mix(T, adj, adj, None, None, None, 11, 1, 4, 1)            # more nodes than entities, no ids
