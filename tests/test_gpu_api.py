"""-m gpu: the reference's call surface (MVIN run wrappers, Aggregator.__call__) on the HIP path."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from oracle import mirror_fp32

from parity import assert_close, run_oracles

pytestmark = pytest.mark.gpu


def build(**kw):
    from mvin_amd.model import MVIN
    d = dict(dim=16, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=8, batch_size=24)
    d.update(kw)
    args = make_args(**d)
    case = synth.small_case(args, n_user=20, n_entity=300, n_relation=6, seed=31, zero_rows=4)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=32, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                 params=params, device="cuda:0")
    return args, case, params, model


def feed_of(model, case, labels=None):
    B = len(case.users)
    feed = {model.user_indices: case.users, model.item_indices: case.items,
            model.labels: labels if labels is not None else np.ones(B, np.float32)}
    for i in range(len(case.memories_h)):
        # train.py:117-120 feeds python lists of per-user numpy rows
        feed[model.memories_h[i]] = [row for row in case.memories_h[i]]
        feed[model.memories_r[i]] = [row for row in case.memories_r[i]]
        feed[model.memories_t[i]] = [row for row in case.memories_t[i]]
    return feed


def test_get_scores_and_eval(hip_lib):
    args, case, params, model = build()
    m, _ = run_oracles(args, case, params)
    labels = (np.arange(len(case.users)) % 2).astype(np.float32)
    feed = feed_of(model, case, labels)
    items, scores = model.get_scores(None, feed)
    assert items.dtype == np.int64 and scores.dtype == np.float32
    np.testing.assert_array_equal(items, case.items)
    assert_close(scores, m.scores_normalized.numpy(), "get_scores")
    auc, acc, f1 = model.eval(None, feed)
    from sklearn.metrics import f1_score, roc_auc_score
    ref = m.scores_normalized.numpy()
    assert abs(auc - roc_auc_score(labels, ref)) < 1e-6
    pred = (ref >= 0.5).astype(np.float32)
    assert abs(acc - np.mean(pred == labels)) < 1e-6 and abs(f1 - f1_score(labels, pred)) < 1e-6


def test_eval_case_study(hip_lib):
    args, case, params, model = build()
    m, _ = run_oracles(args, case, params)
    u, lab, it, ents, rels, imp0, imp1 = model.eval_case_study(None, feed_of(model, case))
    np.testing.assert_array_equal(u, case.users)
    assert len(ents) == 3 and len(rels) == 2
    for a, b in zip(ents, m.entities):
        np.testing.assert_array_equal(a, b.numpy())
    for a, b in zip(rels, m.relations):
        np.testing.assert_array_equal(a, b.numpy())
    K = args.neighbor_sample_size
    assert imp0.shape == (len(u), 1, K) and imp1.shape == (len(u), K, K)
    assert_close(imp0, m.importance_list[0].numpy(), "importance_list_0")
    assert_close(imp1, m.importance_list[1].numpy(), "importance_list_1")


def test_any_batch_length_and_train_wrapper(hip_lib):
    args, case, params, model = build()
    sub = slice(0, 7)  # != args.batch_size: the reference's static reshapes would reject this
    feed = {model.user_indices: case.users[sub], model.item_indices: case.items[sub], model.labels: np.ones(7)}
    for i in range(len(case.memories_h)):
        feed[model.memories_h[i]], feed[model.memories_r[i]], feed[model.memories_t[i]] = \
            case.memories_h[i][sub], case.memories_r[i][sub], case.memories_t[i][sub]
    _, s = model.get_scores(None, feed)
    m, _ = run_oracles(args, case, params)
    assert_close(s, m.scores_normalized.numpy()[sub], "ragged batch")
    _, loss = model.train(None, feed)           # model.py:416-417 -> (_, loss)
    assert np.isfinite(loss) and loss > 0


def test_out_of_range_ids_raise_like_tf_gather(hip_lib):
    args, case, params, model = build()
    feed = feed_of(model, case)
    bad = case.items.copy()
    bad[3] = case.n_entity
    feed[model.item_indices] = bad
    with pytest.raises(IndexError):
        model.get_scores(None, feed)


def test_stage_wise_tables_roundtrip(hip_lib, tmp_path):
    from types import SimpleNamespace
    args, case, params, model = build()
    model.path = SimpleNamespace(emb=str(tmp_path) + "/")
    model.save_pretrain_emb_fuc(None, None)
    before = model.entity_emb_matrix.clone()
    model.entity_emb_matrix.zero_()
    model.restore_pretrain_emb()
    assert torch.equal(before, model.entity_emb_matrix)


@pytest.mark.parametrize("rela", [True, False])
def test_aggregator_call_surface(rela, hip_lib):
    from mvin_amd.aggregators import SumAggregator_urh_matrix
    B, N, K, D, nR = 5, 7, 6, 16, 4
    g = torch.Generator().manual_seed(3)
    selfv, neigh = torch.randn(B, N, D, generator=g), torch.randn(B, N, K, D, generator=g)
    rel_emb = torch.randn(nR, D, generator=g)
    rel_ids = torch.randint(0, nR, (B, N, K), generator=g)
    user = torch.randn(B, D, generator=g)
    agg = SumAggregator_urh_matrix("m", B, D, name="0_0", User_orient_rela=rela, device="cuda:0", seed=5)
    agg.bias.copy_(torch.randn(D, generator=g))
    assert agg.name == "sumaggregator_urh_matrix_m_0_0"
    ref_out, ref_p = mirror_fp32.aggregator_call(
        {"weights": agg.weights.cpu(), "bias": agg.bias.cpu(), "urh_weights": agg.urh_weights.cpu(),
         "User_orient_rela": rela}, selfv, neigh, rel_emb[rel_ids], user, B, D)
    dev = "cuda:0"
    # (a) the reference form: relation VECTORS
    out, p = agg(selfv.to(dev), neigh.to(dev), rel_emb[rel_ids].to(dev), user.to(dev), None)
    assert_close(out.cpu().numpy(), ref_out.numpy(), "aggregator out (vectors)")
    # (b) relation IDS + bound table
    agg.bind_relation_table(rel_emb.to(dev).contiguous())
    out2, p2 = agg(selfv.to(dev), neigh.to(dev), rel_ids.to(dev), user.to(dev), None)
    assert_close(out2.cpu().numpy(), ref_out.numpy(), "aggregator out (ids)")
    if rela:
        assert_close(p.cpu().numpy(), ref_p.numpy(), "probs (vectors)")
        assert_close(p2.cpu().numpy(), ref_p.numpy(), "probs (ids)")
    else:
        assert p is None and p2 is None and ref_p is None


def test_linear_refuses_gathered_sources_of_different_heights(hip_lib):
    """mvin_linear_args has ONE src_rows; two gathered tables of different heights would have the larger one's
    ids clamped to the smaller one's last row -- ops.linear raises instead."""
    import torch
    from mvin_amd import ops
    dev = torch.device("cuda:0")
    big, small = torch.rand(100, 16, device=dev), torch.rand(10, 16, device=dev)
    ids = torch.tensor([3, 50], dtype=torch.int32, device=dev)
    with pytest.raises(ValueError, match="same row count"):
        ops.linear([big, small], torch.eye(16, device=dev), 16, ids=[ids, ids], sum_sources=True)
    got = ops.linear([big, big], torch.eye(16, device=dev), 16, ids=[ids, ids], sum_sources=True)
    assert torch.allclose(got, 2 * big[ids.long()], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("shape", [(3, 1, 8, 16), (2, 4, 4, 64), (5, 3, 64, 12), (1, 7, 1, 8)])
def test_kgcn_style_mixer_of_the_aggregator_base(shape, hip_lib):
    """Aggregator._mix_neighbor_vectors / _urv (aggregators.py:37-77): defined by the reference, never called by MVIN; built on
    mvin_mix_neighbor_vectors_fwd and checked against the float64 restatement."""
    from mvin_amd import ops
    from mvin_amd.aggregators import SumAggregator_urh_matrix
    from oracle import equations_fp64
    B, N, K, D = shape
    rng = np.random.default_rng(B + K)
    nv, nr = rng.standard_normal((B, N, K, D)).astype(np.float32), rng.standard_normal((B, N, K, D)).astype(np.float32)
    u = rng.standard_normal((B, D)).astype(np.float32)
    want, want_p = equations_fp64.mix_neighbor_vectors(nv, nr, u)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    got, probs = ops.mix_neighbor_vectors(t(nv), t(nr), t(u), want_probs=True)
    assert_close(got.cpu().numpy(), want, "aggregated neighbours")
    assert_close(probs.cpu().numpy(), want_p, "user-relation scores (softmax)", atol=1e-7)
    agg = SumAggregator_urh_matrix("m", B, D, name="0_0", weights=np.eye(D, dtype=np.float32), bias=np.zeros(D, np.float32),
                                   urh_weights=np.zeros((3 * D, 1), np.float32), urh_bias=np.zeros(1, np.float32), device=dev)
    for fn in (agg._mix_neighbor_vectors, agg._mix_neighbor_vectors_urv):
        assert_close(fn(t(nv), t(nr), t(u)).cpu().numpy(), want, "class surface")
    # SumAggregator_urh_matrix's own mixers on their own (aggregators.py:118-152): the fused _call does not come through them
    w = rng.standard_normal((3 * D, 1)).astype(np.float32)
    agg2 = SumAggregator_urh_matrix("m", B, D, name="0_0", weights=np.eye(D, dtype=np.float32), bias=np.zeros(D, np.float32),
                                    urh_weights=w, urh_bias=np.zeros(1, np.float32), device=dev)
    sv = rng.standard_normal((B, N, D)).astype(np.float32)
    cat = np.concatenate([np.broadcast_to(u[:, None, None, :], nr.shape), nr, np.broadcast_to(sv[:, :, None, :], nr.shape)], -1)
    s_ = (cat.astype(np.float64) @ w.astype(np.float64))[..., 0]                    # [user ; relation ; self] . urh_weights (:130-133)
    p_ = np.exp(s_ - s_.max(-1, keepdims=True))
    p_ /= p_.sum(-1, keepdims=True)
    got_a, got_p = agg2._mix_neighbor_vectors_urh(t(sv), t(u), t(nv), t(nr))
    assert_close(got_p.cpu().numpy(), p_, "urh probs", atol=1e-6)
    assert_close(got_a.cpu().numpy(), (p_[..., None] * nv).mean(2), "urh aggregate", rtol=1e-5, atol=2e-6)
    assert_close(agg2._mix_neighbor_vectors_no_ur(t(sv), t(u), t(nv), t(nr)).cpu().numpy(), nv.astype(np.float64).mean(2), "plain mean")
