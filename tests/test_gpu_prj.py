"""-m gpu: the PROJECTED-TABLES form of the fused two-level pass (mvin_project_rows + mvin_gather_attn_l2_prj_fwd).

The user-oriented projection (reference model.py:270-283) is linear and the attention weights are scalars, so
(sum_k w_k E[y_k] + c q) W2 + c b2 = sum_k w_k (E W2)[y_k] + c (q W2 + b2): the packed-tile kernel gathers rows of E.W1 /
E.W2 (built once per call, per ENTITY) and adds q.W1 + b1 / q.W2 + b2 (once per PARENT, inside the kernel) instead of
multiplying every distinct child by W1 and W2; round 5's final form folds the aggregator's matrix A0 in as well (E.W1 | E.W1.A0 | E.W2.A0:
mvin_project_tables), so no product per distinct child is left.  Same ids, same rows per pair; results equal to fp32 round-off -- checked here against the
faithful encoded kernel, the fp32 mirror of the reference graph and the float64 equations."""
import os

import numpy as np
import pytest
import torch

from mvin_amd import ops, synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params

from parity import assert_close, run_oracles

pytestmark = pytest.mark.gpu

DK = [(32, 16), (32, 32), (32, 64), (64, 16), (64, 32), (64, 64), (64, 128), (128, 32), (128, 64), (128, 128)]


def _shape(D, K, H=2, B=None):
    if B is None:
        B = max(2, min(37, 4096 // (K * K)))
    return dict(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=2, n_memory=8, batch_size=B)


def _model(args, case, params, prj):
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                 params=params, device="cuda:0")
    model.dedup = True
    model.prj = prj
    return model


def _pairs(model, case):
    dev = model.device
    out = model.forward_device(
        torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
        [torch.from_numpy(m).to(dev) for m in case.memories_h],
        [torch.from_numpy(m).to(dev) for m in case.memories_r],
        [torch.from_numpy(m).to(dev) for m in case.memories_t])
    torch.cuda.synchronize()
    return out


def test_project_rows_is_the_two_products(hip_lib):
    rng = np.random.default_rng(0)
    for rows, D in ((1, 64), (37, 32), (1000, 64), (513, 128)):
        f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).cuda()
        src, W1, W2, b1, b2 = f(rows, D), f(D, D), f(D, D), f(D), f(D)
        out = ops.project_rows(src, W1, W2, b1, b2)
        want = torch.stack([src.double() @ W1.double() + b1.double(), src.double() @ W2.double() + b2.double()])
        assert_close(out.cpu().numpy(), want.float().cpu().numpy(), "project_rows", rtol=1e-5, atol=1e-6)
        out = ops.project_rows(src, W2, W1)                  # no biases; the matrices in the other order in memory
        want = torch.stack([src.double() @ W2.double(), src.double() @ W1.double()])
        assert_close(out.cpu().numpy(), want.float().cpu().numpy(), "project_rows without biases", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("att", ["both", "none", "t0", "t1"])
@pytest.mark.parametrize("ppp", [1, "K"])
@pytest.mark.parametrize("kind", ["repeats", "uniform"])
@pytest.mark.parametrize("dk", DK, ids=lambda dk: "D%dK%d" % dk)
def test_projected_kernel_matches_faithful_kernel(dk, kind, ppp, att, hip_lib):
    """The two entry points on the same inputs: per-parent neighbor aggregates to fp32 round-off."""
    D, K = dk
    if att != "both" and (kind == "uniform" or ppp != 1):
        pytest.skip("attention variants: one adjacency kind and tree depth")
    ppp = K if ppp == "K" else 1
    B = max(2, min(19, 2048 // (K * ppp)))
    args = make_args(**_shape(D, K, B=B))
    case = synth.small_case(args, n_user=8, n_entity=1200, n_relation=7, seed=11 + D + K, zero_rows=4, repeats=kind == "repeats")
    rng = np.random.default_rng(D * K)
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)
    E = f(case.n_entity, D)
    ae = torch.from_numpy(case.adj_entity.astype(np.int32)).to(dev)
    ar = torch.from_numpy(case.adj_relation.astype(np.int32)).to(dev)
    enc_e, enc_r, _ = ops.encode_adjacency(ae, ar)
    parents = torch.from_numpy(rng.integers(0, case.n_entity, size=B * ppp).astype(np.int32)).to(dev)
    parents[0] = int(np.flatnonzero((case.adj_entity == 0).all(1))[0])       # a zero-row parent
    t0 = f(7) if att in ("both", "t0") else None
    t1 = f(7) if att in ("both", "t1") else None
    W1, W2, b1, b2, q, A0, a0 = f(D, D), f(D, D), f(D), f(D), f(B, D), f(D, D), f(D)
    want0, want1 = ops.gather_attn_l2_enc(E, enc_e, enc_r, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, B, ppp, K, D, 7)
    ws = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, t0 is not None)
    got0, got1 = ops.gather_attn_l2_prj(ws, enc_e, enc_r, parents, t0, t1, q, B, ppp, K, D, 7, case.n_entity)
    torch.cuda.synchronize()
    assert_close(got0.cpu().numpy(), want0.cpu().numpy(), "nagg0", rtol=3e-5, atol=6e-6)       # two fp32 programs, sums of up to 128 x 128 terms
    assert_close(got1.cpu().numpy(), want1.cpu().numpy(), "nagg1", rtol=3e-5, atol=6e-6)
    # int64 parent ids read in place, and the launch is deterministic
    again0, again1 = ops.gather_attn_l2_prj(ws, enc_e, enc_r, parents.long(), t0, t1, q, B, ppp, K, D, 7, case.n_entity)
    assert torch.equal(again0, got0) and torch.equal(again1, got1)


@pytest.mark.parametrize("feed", ["pairs", "users", "python-schedule"])
@pytest.mark.parametrize("kind", ["repeats", "uniform"])
@pytest.mark.parametrize("dk", [(32, 32), (64, 16), (64, 32), (64, 64), (128, 32)], ids=lambda dk: "D%dK%d" % dk)
def test_model_with_projected_tables_vs_oracles(dk, kind, feed, hip_lib):
    D, K = dk
    args = make_args(**_shape(D, K))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=7, seed=241 + D + K, zero_rows=4, repeats=kind == "repeats")
    case.items[0] = np.flatnonzero((case.adj_entity == 0).all(1))[0]
    # the per-pair arrays as train.py:117-120 assembles them from the users' ripple sets: both feeds hold the same ids
    uts_np = synth.ripple_sets(16, 900, 7, 2, 8, seed=77)
    mem = synth.memories_for(uts_np, case.users)
    case.memories_h, case.memories_r, case.memories_t = [[np.ascontiguousarray(x) for x in lst] for lst in mem]
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=43, random_agg_bias=True)
    model = _model(args, case, params, True)
    if feed == "python-schedule":
        model.native_l2_max_batch = 0                       # the per-step Python schedule (what a big pairs-feed batch takes)
    if feed == "users":
        dev = model.device
        uts = torch.from_numpy(uts_np).to(dev)
        out = model.forward_users(torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev), uts)
        torch.cuda.synchronize()
    else:
        out = _pairs(model, case)
    assert model._prj_for_l2(len(case.items)), "the projected-tables form was not taken"
    m, e = run_oracles(args, case, params)
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), "projected-tables scores vs fp32 mirror", rtol=1e-5, atol=1e-6)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), "item_embeddings", rtol=1e-5, atol=1e-6)
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 1e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}"
    # and against the faithful encoded kernel on the same model
    model.prj = False
    ref = _pairs(model, case) if feed != "users" else model.forward_users(
        torch.from_numpy(case.users).to(model.device), torch.from_numpy(case.items).to(model.device), uts)
    assert_close(got, ref.scores.cpu().numpy(), "projected vs faithful encoded kernel", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("ablation", ["no_uor", "no_uor_and_no_kg_eh_uo"])
def test_projected_tables_without_relation_attention(ablation, hip_lib):
    args = make_args(ablation=ablation, **_shape(64, 32, B=11))
    case = synth.small_case(args, n_user=8, n_entity=600, n_relation=6, seed=161, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=62, random_agg_bias=True)
    out = _pairs(_model(args, case, params, True), case)
    m, _ = run_oracles(args, case, params)
    assert_close(out.scores.cpu().numpy(), m.scores.numpy(), "scores vs fp32 mirror", rtol=1e-5, atol=1e-6)


def test_projected_tables_depth3(hip_lib):
    args = make_args(**_shape(64, 16, H=3, B=3))
    case = synth.small_case(args, n_user=8, n_entity=700, n_relation=5, seed=151, zero_rows=3, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=52, random_agg_bias=True)
    model = _model(args, case, params, True)
    out = _pairs(model, case)
    m, _ = run_oracles(args, case, params)
    assert_close(out.scores.cpu().numpy(), m.scores.numpy(), "scores vs fp32 mirror", rtol=1e-5, atol=1e-6)


def test_auto_rule_and_refusals(hip_lib):
    """Automatic: only when the batch's children outnumber the entities (B K >= 16 n_entity; 10 / 5 n_entity at K <= 32 / K = 64 where the
    per-entity aggregates exist); never for a bf16 table or
    without the projection; the entry point refuses what it cannot do."""
    args = make_args(**_shape(64, 32, B=8))
    case = synth.small_case(args, n_user=8, n_entity=500, n_relation=6, seed=5, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=6)
    model = _model(args, case, params, None)
    # (D = 64, K = 32: the aggregates stand behind the tables -> 10 n_entity; the kernels over the tables themselves -> 16 n_entity)
    assert not model._prj_for_l2(8) and not model._prj_for_l2(156)
    assert model._prj_for_l2(157) and model._prj_for_l2(4, n_parents=157)
    model.agg = False
    assert not model._prj_for_l2(249) and model._prj_for_l2(250) and model._prj_for_l2(4, n_parents=250)
    model.agg = None
    model.prj = False
    assert not model._prj_for_l2(1 << 20)
    bf = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
              device="cuda:0", table_dtype="bf16")
    bf.prj = True
    assert not bf._prj_for_l2(1 << 20)
    a2 = make_args(ablation="no_uo", **_shape(64, 32, B=8))
    nouo = MVIN(a2, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                params=init_params(a2, case.n_user, case.n_entity, case.n_relation, seed=6), device="cuda:0")
    nouo.prj = True
    assert not nouo._prj_for_l2(1 << 20)
    enc = model.encoded_adjacency()
    E, W = model.entity_emb_matrix, model._agg[(0, 0)].weights
    ws = ops.project_tables(E, W, W, None, None, W, None, 32, True)
    ids = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    with pytest.raises(ValueError):                          # queries of another batch size
        ops.gather_attn_l2_prj(ws, enc[0], enc[1], ids, None, None, torch.zeros(5, 64, device="cuda:0"), 4, 1, 32, 64, 6, 500)
    with pytest.raises(ValueError):                          # a workspace built for another table
        ops.gather_attn_l2_prj(ws[:-64].contiguous(), enc[0], enc[1], ids, None, None, torch.zeros(4, 64, device="cuda:0"), 4, 1, 32, 64, 6, 500)


def test_project_tables_holds_the_three_products(hip_lib):
    """The workspace: E.W1 | E.W1.A0 | E.W2.A0, then W1 | W1.A0 | W2.A0, (W1 + c W2).A0, b1, (b1 + c b2).A0 + a0."""
    rng = np.random.default_rng(5)
    nE, D, K = 777, 64, 32
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).cuda()
    E, W1, W2, b1, b2, A0, a0 = f(nE, D), f(D, D), f(D, D), f(D), f(D), f(D, D), f(D)
    for att in (True, False):
        c = 1.0 / K if att else 1.0
        ws = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, att).double().cpu()
        d = lambda t: t.double().cpu()
        tabs = ws[:3 * nE * D].view(3, nE, D)
        want = torch.stack([d(E) @ d(W1), d(E) @ d(W1) @ d(A0), d(E) @ d(W2) @ d(A0)])
        assert_close(tabs.numpy(), want.numpy(), "tables", rtol=2e-5, atol=2e-6)
        blk = ws[3 * nE * D:]
        assert_close(blk[:3 * D * D].view(3, D, D).numpy(), torch.stack([d(W1), d(W1) @ d(A0), d(W2) @ d(A0)]).numpy(), "Wstack",
                     rtol=1e-5, atol=1e-6)
        assert_close(blk[3 * D * D:4 * D * D].view(D, D).numpy(), ((d(W1) + c * d(W2)) @ d(A0)).numpy(), "Wv", rtol=1e-5, atol=1e-6)
        assert_close(blk[4 * D * D:4 * D * D + D].numpy(), d(b1).numpy(), "b1", rtol=0, atol=0)
        assert_close(blk[4 * D * D + D:].numpy(), ((d(b1) + c * d(b2)) @ d(A0) + d(a0)).numpy(), "bv", rtol=1e-5, atol=1e-6)


def test_parameters_changed_between_calls_are_seen(hip_lib):
    """Nothing of the projected tables is kept between calls: an in-place change of the entity table or of a projection
    matrix shows in the next call's scores exactly as on the faithful path."""
    args = make_args(**_shape(64, 32, B=9))
    case = synth.small_case(args, n_user=8, n_entity=400, n_relation=6, seed=9, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=10, random_agg_bias=True)
    model = _model(args, case, params, True)
    before = _pairs(model, case).scores.clone()
    with torch.no_grad():
        model.entity_emb_matrix.mul_(1.25)
        model.transfer_matrix_list[2].add_(0.05)
    after = _pairs(model, case).scores.clone()
    model.prj = False
    want = _pairs(model, case).scores
    assert not torch.allclose(before, after)
    assert_close(after.cpu().numpy(), want.cpu().numpy(), "scores after an in-place parameter change", rtol=1e-5, atol=1e-6)


# ---- D = 32, K in {8, 16}: the wave-per-parent kernel (BASELINE C2) reads projected tables over EITHER adjacency form ----
@pytest.mark.parametrize("encoded", [False, True], ids=["plain", "encoded"])
@pytest.mark.parametrize("att", ["both", "none"])
@pytest.mark.parametrize("ppp", [1, "K"])
@pytest.mark.parametrize("K", [8, 16])
def test_projected_wave_per_parent_kernel_matches_faithful(K, ppp, att, encoded, hip_lib):
    D = 32
    if encoded and K == 8:
        pytest.skip("the encoded entry points start at K = 16")
    ppp = K if ppp == "K" else 1
    B = 23 if ppp == 1 else 5
    args = make_args(**_shape(D, K, B=B))
    case = synth.small_case(args, n_user=8, n_entity=900, n_relation=7, seed=31 + K, zero_rows=4, repeats=True)
    rng = np.random.default_rng(K * 7 + ppp)
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)
    E = f(case.n_entity, D)
    ae = torch.from_numpy(case.adj_entity.astype(np.int32)).to(dev)
    ar = torch.from_numpy(case.adj_relation.astype(np.int32)).to(dev)
    parents = torch.from_numpy(rng.integers(0, case.n_entity, size=B * ppp).astype(np.int32)).to(dev)
    parents[0] = int(np.flatnonzero((case.adj_entity == 0).all(1))[0])
    t0 = f(7) if att == "both" else None
    t1 = f(7) if att == "both" else None
    W1, W2, b1, b2, q, A0, a0 = f(D, D), f(D, D), f(D), f(D), f(B, D), f(D, D), f(D)
    want0, want1, _, _ = ops.gather_attn_l2(E, ae, ar, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, B, ppp, K, D, 7)
    ws = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, t0 is not None)
    if encoded:
        ee, er, _ = ops.encode_adjacency(ae, ar)
    else:
        ee, er = ae, ar
    got0, got1 = ops.gather_attn_l2_prj(ws, ee, er, parents, t0, t1, q, B, ppp, K, D, 7, case.n_entity, encoded=encoded)
    torch.cuda.synchronize()
    assert_close(got0.cpu().numpy(), want0.cpu().numpy(), "nagg0", rtol=3e-5, atol=6e-6)
    assert_close(got1.cpu().numpy(), want1.cpu().numpy(), "nagg1", rtol=3e-5, atol=6e-6)


@pytest.mark.parametrize("feed", ["pairs", "python-schedule"])
@pytest.mark.parametrize("K", [8, 16])
def test_c2_shaped_model_with_projected_tables_vs_oracles(K, feed, hip_lib):
    """dim 32, fan-out 8 / 16 with the PLAIN adjacency (what MVIN takes at BASELINE C2's shape) and the projected tables."""
    args = make_args(**_shape(32, K, B=29))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=7, seed=77 + K, zero_rows=4, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=43, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device="cuda:0")
    model.prj = True
    assert model._enc_for_l2(n_parents=29) is None and model._prj_plain_ok()
    if feed == "python-schedule":
        model.native_l2_max_batch = 0
    out = _pairs(model, case)
    m, e = run_oracles(args, case, params)
    assert_close(out.scores.cpu().numpy(), m.scores.numpy(), "scores vs fp32 mirror", rtol=1e-5, atol=1e-6)
    model.prj = False
    ref = _pairs(model, case)
    assert_close(out.scores.cpu().numpy(), ref.scores.cpu().numpy(), "projected vs faithful", rtol=1e-5, atol=1e-6)


def test_plain_adjacency_form_exists_only_for_the_wave_per_parent_kernel(hip_lib):
    from mvin_amd._lib import MvinHipError
    dev = "cuda:0"
    E = torch.zeros(300, 64, device=dev)
    W = torch.zeros(64, 64, device=dev)
    ws = ops.project_tables(E, W, W, None, None, W, None, 32, True)
    adj = torch.zeros((300, 32), dtype=torch.int32, device=dev)
    with pytest.raises(MvinHipError):
        ops.gather_attn_l2_prj(ws, adj, adj, torch.zeros(4, dtype=torch.int32, device=dev), None, None, torch.zeros(4, 64, device=dev),
                               4, 1, 32, 64, 6, 300, encoded=False)


@pytest.mark.parametrize("K", [8, 16])
def test_attention_outputs_keep_the_unprojected_form(K, hip_lib):
    """the projected-tables kernels write no attention outputs, so a want_probs pass (eval_case_study,
    model.py:428-441) must not take them even when the form is forced: importance_list == the reference graph's
    probs_normalized, not [None, None]."""
    args = make_args(**_shape(32, K, B=13))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=7, seed=177 + K, zero_rows=4, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=44, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device="cuda:0")
    model.prj = True
    model.native_l2_max_batch = 0
    assert model._prj_plain_ok() and model._prj_for_l2(13)
    dev = model.device
    out = model.forward_device(torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
                               [torch.from_numpy(m).to(dev) for m in case.memories_h], [torch.from_numpy(m).to(dev) for m in case.memories_r],
                               [torch.from_numpy(m).to(dev) for m in case.memories_t], want_probs=True)
    m, _ = run_oracles(args, case, params)
    assert len(out.importance_list) == 2 and all(p is not None for p in out.importance_list)
    for got, want in zip(out.importance_list, m.importance_list):
        assert_close(got.cpu().numpy(), want.numpy().reshape(got.shape), "importance_list", rtol=1e-5, atol=1e-7)
    assert_close(out.scores.cpu().numpy(), m.scores.numpy(), "scores vs fp32 mirror", rtol=1e-5, atol=1e-6)


def test_forced_projection_falls_back_where_no_kernel_takes_it(hip_lib):
    """D = 32, K = 16 over the plain adjacency with a relation table too large for the wave-per-parent kernel's LDS
    (fused_d32_applies false): a forced / automatic projected-tables form must fall back to the unprojected kernels, not raise."""
    nR = 4000
    args = make_args(**_shape(32, 16, B=21))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=nR, seed=99, zero_rows=4, repeats=False)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=45, random_agg_bias=True)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device="cuda:0")
    assert not ops.gather_attn_l2_prj_supported(32, 16, False, case.n_entity, nR)
    assert ops.gather_attn_l2_prj_supported(32, 16, False, case.n_entity, 7) and ops.gather_attn_l2_prj_supported(64, 32, True, case.n_entity, nR)
    assert not ops.gather_attn_l2_prj_supported(64, 32, False, case.n_entity, 7)
    model.prj = True
    model.dedup = False
    assert not model._prj_plain_ok()
    m, _ = run_oracles(args, case, params)
    for native in (65536, 0):
        model.native_l2_max_batch = native
        out = _pairs(model, case)
        assert_close(out.scores.cpu().numpy(), m.scores.numpy(), f"scores vs fp32 mirror (native_l2_max_batch={native})", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K", [16, 32])
def test_wave_per_parent_kernel_every_distinct_count(K, hip_lib):
    """The wave-per-parent kernel of dim 64 (mvin_fused_wpp.hip) walks a parent's distinct children four at a time, one per 16-lane group:
    every distinct-children count 1 .. K as parent AND as child (partly filled last passes; a last pass whose only valid group reads the
    parent's slot word from a lane of an invalid group -- a dropped 17th child was the first bug of that kernel), with and without
    attention, in item order and as given, against the float64 evaluation of the unprojected formulas on the PLAIN adjacency."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    D, nR, n_entity = 64, 7, 600
    rng = np.random.default_rng(K)
    adj_e = np.zeros((n_entity, K), dtype=np.int64)
    adj_r = np.zeros((n_entity, K), dtype=np.int64)
    for x in range(n_entity):
        nd = x % K + 1                                        # distinct (neighbour, relation) slots of this row
        ne = rng.choice(n_entity, nd, replace=False)
        nr = rng.integers(0, nR, nd)
        pick = np.concatenate([np.arange(nd), rng.integers(0, nd, K - nd)])      # every distinct slot at least once, the rest repeats
        rng.shuffle(pick)
        adj_e[x], adj_r[x] = ne[pick], nr[pick]
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)      # noqa: E731
    E = f(n_entity, D)
    ae, ar = torch.from_numpy(adj_e.astype(np.int32)).to(dev), torch.from_numpy(adj_r.astype(np.int32)).to(dev)
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    assert sorted(set(cnt.cpu().tolist())) == list(range(1, K + 1))
    B = 4 * K + 37
    parents = torch.from_numpy((np.arange(B) * 7 % n_entity).astype(np.int64)).to(dev)          # every count 1 .. K several times
    W1, W2, b1, b2, q, A0, a0 = f(D, D), f(D, D), f(D), f(D), f(B, D), f(D, D), f(D)
    # (att = the relation logits' scale: 1 -> softmaxes over the exp tables; 130 -> a spread above the kernel's threshold of 60, the
    #  per-row-maximum form: rows whose logits all lie far below the global maximum must keep their weights)
    for att in (1.0, 130.0, False):
        t0 = f(nR) * att if att else None
        t1 = f(nR) * att if att else None
        if att and att > 1:
            assert float(t0.max() - t0.min()) > 60 and float(t1.max() - t1.min()) > 60
        ws = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, bool(att))
        for order in (None, ops.order_by_key(parents), torch.flip(torch.arange(B, dtype=torch.int32, device=dev), dims=[0])):
            got0, got1 = ops.gather_attn_l2_prj(ws, enc_e, enc_r, parents, t0, t1, q, B, 1, K, D, nR, n_entity, order=order)
            torch.cuda.synchronize()
            if att:
                r0, r1 = bench.l2_reference_f64(E, ae, ar, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, K)
                assert_close(got0.cpu().numpy(), r0.cpu().numpy(), "nagg0 vs float64", rtol=1e-5, atol=2e-6)
                assert_close(got1.cpu().numpy(), r1.cpu().numpy(), "nagg1 vs float64", rtol=1e-5, atol=2e-6)
            want0, want1 = ops.gather_attn_l2_enc(E, enc_e, enc_r, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, B, 1, K, D, nR)
            assert_close(got0.cpu().numpy(), want0.cpu().numpy(), "nagg0 vs the packed-tile kernel", rtol=3e-5, atol=6e-6)
            assert_close(got1.cpu().numpy(), want1.cpu().numpy(), "nagg1 vs the packed-tile kernel", rtol=3e-5, atol=6e-6)


@pytest.mark.parametrize("K", [16, 32, 64])
def test_entity_aggregates_form_every_distinct_count(K, hip_lib):
    """The per-entity aggregates form (mvin_entity_aggregates -> mvin_gather_attn_l2_agg_fwd, mvin_fused_agg.hip): S0 | G against their
    float64 definitions over the projected tables, then the launch -- every distinct-children count 1 .. K as parent and as child,
    ragged last batches and quads, both softmax forms, with and without attention, parents as given / in key order / reversed -- against
    the float64 evaluation of the UNPROJECTED formulas on the plain adjacency and against the wave-per-parent kernel over the tables."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    D, nR, n_entity = 64, 7, 603
    rng = np.random.default_rng(K + 100)
    adj_e = np.zeros((n_entity, K), dtype=np.int64)
    adj_r = np.zeros((n_entity, K), dtype=np.int64)
    for x in range(n_entity):
        nd = x % K + 1
        ne = rng.choice(n_entity, nd, replace=False)
        nr = rng.integers(0, nR, nd)
        pick = np.concatenate([np.arange(nd), rng.integers(0, nd, K - nd)])
        rng.shuffle(pick)
        adj_e[x], adj_r[x] = ne[pick], nr[pick]
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)      # noqa: E731
    E = f(n_entity, D)
    ae, ar = torch.from_numpy(adj_e.astype(np.int32)).to(dev), torch.from_numpy(adj_r.astype(np.int32)).to(dev)
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    assert ops.gather_attn_l2_agg_supported(D, K, n_entity, nR)
    W1, W2, b1, b2, A0, a0 = f(D, D), f(D, D), f(D), f(D), f(D, D), f(D)
    for B in (4 * K + 37, 1, 18):
        parents = torch.from_numpy((np.arange(B) * 7 % n_entity).astype(np.int64)).to(dev)
        q = f(B, D)
        for att in (1.0, 130.0, False):
            t0 = f(nR) * att if att else None
            t1 = f(nR) * att if att else None
            ws = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, bool(att))
            agg = ops.entity_aggregates(ws, enc_e, enc_r, t0, K, D, nR, n_entity)
            if B > 18:
                # S0 | G from their definitions: w(e) = softmax over the K slots of t0[relation] (plain mean without), over K
                T = ws[: 3 * n_entity * D].view(3, n_entity, D).double()
                lg = t0.double()[ar.long()] if att else torch.zeros((n_entity, K), dtype=torch.float64, device=dev)
                w = torch.softmax(lg, dim=1) / K if att else torch.full_like(lg, 1.0 / K)
                S0 = (w[:, :, None] * T[0][ae.long()]).sum(1)
                G = T[1] + (w[:, :, None] * T[2][ae.long()]).sum(1)
                got = agg.view(2, n_entity, D)
                assert_close(got[0].cpu().numpy(), S0.cpu().numpy(), "S0", rtol=1e-5, atol=2e-6)
                assert_close(got[1].cpu().numpy(), G.cpu().numpy(), "G", rtol=1e-5, atol=2e-6)
            orders = (None,) if B == 1 else (None, ops.order_by_key(parents), torch.flip(torch.arange(B, dtype=torch.int32, device=dev), dims=[0]))
            for order in orders:
                got0, got1 = ops.gather_attn_l2_agg(ws, agg, enc_e, enc_r, parents, t0, t1, q, B, 1, K, D, nR, n_entity, order=order)
                torch.cuda.synchronize()
                if att:
                    r0, r1 = bench.l2_reference_f64(E, ae, ar, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, K)
                    assert_close(got0.cpu().numpy(), r0.cpu().numpy(), "nagg0 vs float64", rtol=1e-5, atol=2e-6)
                    assert_close(got1.cpu().numpy(), r1.cpu().numpy(), "nagg1 vs float64", rtol=1e-5, atol=2e-6)
                want0, want1 = ops.gather_attn_l2_prj(ws, enc_e, enc_r, parents, t0, t1, q, B, 1, K, D, nR, n_entity)
                assert_close(got0.cpu().numpy(), want0.cpu().numpy(), "nagg0 vs the kernel over the tables", rtol=3e-5, atol=6e-6)
                assert_close(got1.cpu().numpy(), want1.cpu().numpy(), "nagg1 vs the kernel over the tables", rtol=3e-5, atol=6e-6)
    # several parents per pair (deeper trees: the level-(L-2) nodes of a pair share its query row)
    B, ppp = 9, K
    parents = torch.from_numpy(rng.integers(0, n_entity, B * ppp).astype(np.int32)).to(dev)
    q, t0, t1 = f(B, D), f(nR), f(nR)
    ws = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, True)
    agg = ops.entity_aggregates(ws, enc_e, enc_r, t0, K, D, nR, n_entity)
    got0, got1 = ops.gather_attn_l2_agg(ws, agg, enc_e, enc_r, parents, t0, t1, q, B, ppp, K, D, nR, n_entity)
    want0, want1 = ops.gather_attn_l2_prj(ws, enc_e, enc_r, parents, t0, t1, q, B, ppp, K, D, nR, n_entity)
    assert_close(got0.cpu().numpy(), want0.cpu().numpy(), "nagg0, K parents per pair", rtol=3e-5, atol=6e-6)
    assert_close(got1.cpu().numpy(), want1.cpu().numpy(), "nagg1, K parents per pair", rtol=3e-5, atol=6e-6)


@pytest.mark.parametrize("K", [16, 32, 64])
def test_folded_tail_form_matches_aggregates_plus_tail(K, hip_lib):
    """mvin_fold_tables -> mvin_score_l2_folded_fwd (H0 | G aggregates, M0 table, four products in the tail) against mvin_project_tables ->
    mvin_entity_aggregates -> mvin_gather_attn_l2_agg_fwd -> mvin_l2_tail_fwd on the same parameters: every distinct-children count, ragged
    batches (tiles of 32 and quads of 4 partly filled), int64 and int32 items, with and without attention / biases."""
    D, nR, n_entity = 64, 7, 603
    rng = np.random.default_rng(K + 200)
    adj_e = np.zeros((n_entity, K), dtype=np.int64)
    adj_r = np.zeros((n_entity, K), dtype=np.int64)
    for x in range(n_entity):
        nd = x % K + 1
        ne = rng.choice(n_entity, nd, replace=False)
        nr = rng.integers(0, nR, nd)
        pick = np.concatenate([np.arange(nd), rng.integers(0, nd, K - nd)])
        rng.shuffle(pick)
        adj_e[x], adj_r[x] = ne[pick], nr[pick]
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)      # noqa: E731
    E = f(n_entity, D)
    ae, ar = torch.from_numpy(adj_e.astype(np.int32)).to(dev), torch.from_numpy(adj_r.astype(np.int32)).to(dev)
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    assert ops.score_l2_folded_supported(D, K, n_entity, nR)
    W0, W1, W2, A0, A1, Wmix = f(D, D), f(D, D), f(D, D), f(D, D), f(D, D), f(3 * D, D)
    for B, att, bias, i64 in ((4 * K + 37, 1.0, True, True), (1, 1.0, True, False), (33, 130.0, True, True), (95, False, False, False)):
        b0, b1, b2, a0, a1, bmix = (f(D) if bias else None for _ in range(6))
        items = torch.from_numpy((np.arange(B) * 7 % n_entity).astype(np.int64 if i64 else np.int32)).to(dev)
        q, user_o = f(B, D), f(B, D)
        t0 = f(nR) * att if att else None
        t1 = f(nR) * att if att else None
        ws = ops.fold_tables(E, enc_e, enc_r, t0, W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, K, nR)
        item, scores, sig = ops.score_l2_folded(ws, enc_e, enc_r, items, t0, t1, q, user_o, A1, a1, Wmix, K, D, nR, n_entity)
        torch.cuda.synchronize()
        pt = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, bool(att))
        agg = ops.entity_aggregates(pt, enc_e, enc_r, t0, K, D, nR, n_entity)
        n0, n1 = ops.gather_attn_l2_agg(pt, agg, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, n_entity)
        want_item, want_scores, want_sig = ops.l2_tail(E, items, q, user_o, n0, n1, W0, b0, A0, a0, A1, a1, Wmix, bmix)
        torch.cuda.synchronize()
        assert_close(item.cpu().numpy(), want_item.cpu().numpy(), f"item_emb B={B}", rtol=3e-5, atol=3e-5)
        assert_close(scores.cpu().numpy(), want_scores.cpu().numpy(), f"scores B={B}", rtol=3e-5, atol=3e-5)
        assert_close(sig.cpu().numpy(), want_sig.cpu().numpy(), f"sigmoid B={B}", rtol=3e-5, atol=1e-5)
        # the definitions of H0 and M0 (float64)
        if B > 100:
            tab = n_entity * D
            T = ws[: 6 * tab].view(6, n_entity, D).double()
            c = (1.0 / K) if att else 1.0
            lg = t0.double()[ar.long()] if att else torch.zeros((n_entity, K), dtype=torch.float64, device=dev)
            w = torch.softmax(lg, dim=1) / K if att else torch.full_like(lg, 1.0 / K)
            Ed = E.double()
            TA1 = Ed @ W1.double() @ A0.double()
            H0 = Ed @ W0.double() @ A0.double() + (w[:, :, None] * TA1[ae.long()]).sum(1)
            M0 = Ed @ W0.double() @ Wmix[:D].double()
            assert_close(T[4].cpu().numpy(), H0.cpu().numpy(), "H0", rtol=2e-5, atol=5e-6)
            assert_close(T[3].cpu().numpy(), M0.cpu().numpy(), "M0", rtol=2e-5, atol=5e-6)
            assert c > 0


@pytest.mark.parametrize("K", [16, 32])
def test_folded_tail_form_dim32(K, hip_lib):
    """The folded-tail form at dim 32 (mvin_fused_agg32.hip: eight rows per wave, two-tile products) against mvin_project_tables ->
    mvin_gather_attn_l2_prj_fwd (the wave-per-parent kernel of dim 32 where it exists, the packed-tile kernel otherwise) -> mvin_l2_tail_fwd on the
    same parameters, and H0 | G | M0 against their float64 definitions: every distinct-children count, ragged batches, both id widths,
    with and without attention / biases."""
    D, nR, n_entity = 32, 7, 611
    rng = np.random.default_rng(K + 300)
    adj_e = np.zeros((n_entity, K), dtype=np.int64)
    adj_r = np.zeros((n_entity, K), dtype=np.int64)
    for x in range(n_entity):
        nd = x % K + 1
        ne = rng.choice(n_entity, nd, replace=False)
        nr = rng.integers(0, nR, nd)
        pick = np.concatenate([np.arange(nd), rng.integers(0, nd, K - nd)])
        rng.shuffle(pick)
        adj_e[x], adj_r[x] = ne[pick], nr[pick]
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)      # noqa: E731
    E = f(n_entity, D)
    ae, ar = torch.from_numpy(adj_e.astype(np.int32)).to(dev), torch.from_numpy(adj_r.astype(np.int32)).to(dev)
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    assert ops.score_l2_folded_supported(D, K, n_entity, nR) and not ops.gather_attn_l2_agg_supported(D, K, n_entity, nR)
    W0, W1, W2, A0, A1, Wmix = f(D, D), f(D, D), f(D, D), f(D, D), f(D, D), f(3 * D, D)
    for B, att, bias, i64 in ((4 * K + 37, 1.0, True, True), (1, 1.0, True, False), (9, 130.0, True, True), (95, False, False, False)):
        b0, b1, b2, a0, a1, bmix = (f(D) if bias else None for _ in range(6))
        items = torch.from_numpy((np.arange(B) * 7 % n_entity).astype(np.int64 if i64 else np.int32)).to(dev)
        q, user_o = f(B, D), f(B, D)
        t0 = f(nR) * att if att else None
        t1 = f(nR) * att if att else None
        ws = ops.fold_tables(E, enc_e, enc_r, t0, W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, K, nR)
        item, scores, sig = ops.score_l2_folded(ws, enc_e, enc_r, items, t0, t1, q, user_o, A1, a1, Wmix, K, D, nR, n_entity)
        torch.cuda.synchronize()
        pt = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, bool(att))
        n0, n1 = ops.gather_attn_l2_prj(pt, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, n_entity)
        want_item, want_scores, want_sig = ops.l2_tail(E, items, q, user_o, n0, n1, W0, b0, A0, a0, A1, a1, Wmix, bmix)
        torch.cuda.synchronize()
        assert_close(item.cpu().numpy(), want_item.cpu().numpy(), f"item_emb B={B}", rtol=3e-5, atol=3e-5)
        assert_close(scores.cpu().numpy(), want_scores.cpu().numpy(), f"scores B={B}", rtol=3e-5, atol=3e-5)
        assert_close(sig.cpu().numpy(), want_sig.cpu().numpy(), f"sigmoid B={B}", rtol=3e-5, atol=1e-5)
        if B > 100:
            T = ws[: 6 * n_entity * D].view(6, n_entity, D).double()
            lg = t0.double()[ar.long()] if att else torch.zeros((n_entity, K), dtype=torch.float64, device=dev)
            w = torch.softmax(lg, dim=1) / K if att else torch.full_like(lg, 1.0 / K)
            Ed = E.double()
            TA1 = Ed @ W1.double() @ A0.double()
            TA2 = Ed @ W2.double() @ A0.double()
            H0 = Ed @ W0.double() @ A0.double() + (w[:, :, None] * TA1[ae.long()]).sum(1)
            G = TA1 + (w[:, :, None] * TA2[ae.long()]).sum(1)
            M0 = Ed @ W0.double() @ Wmix[:D].double()
            assert_close(T[4].cpu().numpy(), H0.cpu().numpy(), "H0", rtol=2e-5, atol=5e-6)
            assert_close(T[5].cpu().numpy(), G.cpu().numpy(), "G", rtol=2e-5, atol=5e-6)
            assert_close(T[3].cpu().numpy(), M0.cpu().numpy(), "M0", rtol=2e-5, atol=5e-6)


FOLD_PRESETS = ["all", "no_sw", "no_kg_eh_uo", "no_uor", "no_uor_and_no_kg_eh_uo", "no_ps_o_ft", "ho_only", "ho_only_uo_kg_eh", "no_uor_ho_only"]


@pytest.mark.parametrize("i", range(int(os.environ.get("MVIN_FUZZ_OFFSET", "0")), int(os.environ.get("MVIN_FUZZ_OFFSET", "0")) + 36))
def test_random_configuration_in_folded_form(i, hip_lib):
    """Seeded random sweep over what the folded-tail form can meet -- dim 32 / 64, its fan-outs, ripple hops and memories, 2 .. 130 relations,
    ragged batches from one pair on, every ablation preset that keeps User_orient and the wide-and-deep combiner (a query that is NOT user_o,
    no relation attention, user embeddings instead of preference sets, ...), the three feeds, int32 ids -- through the model (forced: the
    automatic rule needs bench-size batches), against the fp32 mirror and the fp64 equations."""
    rng = np.random.default_rng(77000 + i)
    D = int(rng.choice([32, 64]))
    K = int(rng.choice([16, 32] if D == 32 else [16, 32, 64]))
    P, Nm = int(rng.choice([1, 2, 2, 3])), int(rng.choice([4, 8, 16, 32, 64]))
    nR = int(rng.choice([2, 5, 9, 39, 130]))
    B = int(rng.choice([1, 7, 33, 64, 129, 300]))
    n_user = int(rng.choice([1, 3, 17, 200]))
    abl = str(rng.choice(FOLD_PRESETS))
    feed = str(rng.choice(["pairs", "users", "users_grouped"]))
    ids32 = bool(rng.random() < 0.3)
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=2, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B, ablation=abl)
    case = synth.small_case(args, n_user=n_user, n_entity=150 + 41 * (i % 7), n_relation=nR, seed=77100 + i, zero_rows=3, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=77200 + i, random_agg_bias=True)
    uts = synth.ripple_sets(case.n_user, case.n_entity, case.n_relation, max(1, P), Nm, seed=77300 + i)
    if feed != "pairs":
        case.memories_h, case.memories_r, case.memories_t = synth.memories_for(uts, case.users)
    m, e = run_oracles(args, case, params)
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device="cuda:0")
    model.prj, model.dedup = True, True
    dev = model.device
    u_d, i_d = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    if ids32:
        u_d, i_d = u_d.to(torch.int32), i_d.to(torch.int32)
    mem = [[torch.from_numpy(x).to(dev) for x in lst] for lst in (case.memories_h, case.memories_r, case.memories_t)]
    if feed == "pairs":
        out = model.forward_device(u_d, i_d, *mem)
    else:
        model.group_min_pairs_per_user = 0 if feed == "users_grouped" else 10 ** 9
        out = model.forward_users(u_d, i_d, torch.from_numpy(uts).to(dev))
    torch.cuda.synchronize()
    what = f"case {i}: D={D} K={K} P={P} Nm={Nm} nR={nR} B={B} users={n_user} {abl} {feed} ids32={ids32}"
    enc = model._enc_for_l2(n_parents=B)
    assert enc is not None and model._fold_for(enc), what
    assert any(t is not None for t in model._fold_ws.values()), f"the folded-tail form did not run, {what}"
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), f"scores vs fp32 mirror, {what}", rtol=1e-5, atol=1e-6)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), f"item_embeddings, {what}", rtol=1e-5, atol=1e-6)
    err_hip, err_mir = np.abs(got - e.scores).max(), np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 2e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}, {what}"


@pytest.mark.parametrize("K", [16, 32])
def test_folded_tail_gather_form(K, hip_lib):
    """mvin_fold_tables_ex(aggregates = 0) -> mvin_score_l2_folded_gather_fwd (every pair gathers its own rows; the tail in the same launch)
    against mvin_project_tables -> mvin_gather_attn_l2_prj_fwd -> mvin_l2_tail_fwd: every distinct-children count as parent and as child,
    ragged batches, parents as given / in key order / reversed, both softmax forms, with and without attention / biases."""
    D, nR, n_entity = 64, 7, 603
    rng = np.random.default_rng(K + 400)
    adj_e = np.zeros((n_entity, K), dtype=np.int64)
    adj_r = np.zeros((n_entity, K), dtype=np.int64)
    for x in range(n_entity):
        nd = x % K + 1
        ne = rng.choice(n_entity, nd, replace=False)
        nr = rng.integers(0, nR, nd)
        pick = np.concatenate([np.arange(nd), rng.integers(0, nd, K - nd)])
        rng.shuffle(pick)
        adj_e[x], adj_r[x] = ne[pick], nr[pick]
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)      # noqa: E731
    E = f(n_entity, D)
    ae, ar = torch.from_numpy(adj_e.astype(np.int32)).to(dev), torch.from_numpy(adj_r.astype(np.int32)).to(dev)
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    assert ops.score_l2_folded_gather_supported(D, K, n_entity, nR) and not ops.score_l2_folded_gather_supported(D, 64, n_entity, nR)
    W0, W1, W2, A0, A1, Wmix = f(D, D), f(D, D), f(D, D), f(D, D), f(D, D), f(3 * D, D)
    for B, att, bias, i64 in ((4 * K + 37, 1.0, True, True), (1, 1.0, True, False), (33, 130.0, True, True), (95, False, False, False)):
        b0, b1, b2, a0, a1, bmix = (f(D) if bias else None for _ in range(6))
        items = torch.from_numpy((np.arange(B) * 7 % n_entity).astype(np.int64 if i64 else np.int32)).to(dev)
        q, user_o = f(B, D), f(B, D)
        t0 = f(nR) * att if att else None
        t1 = f(nR) * att if att else None
        ws = ops.fold_tables(E, enc_e, enc_r, t0, W0, b0, W1, b1, W2, b2, A0, a0, Wmix, bmix, A1, K, nR, aggregates=False)
        pt = ops.project_tables(E, W1, W2, b1, b2, A0, a0, K, bool(att))
        n0, n1 = ops.gather_attn_l2_prj(pt, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, n_entity)
        want_item, want_scores, want_sig = ops.l2_tail(E, items, q, user_o, n0, n1, W0, b0, A0, a0, A1, a1, Wmix, bmix)
        orders = (None,) if B == 1 else (None, ops.order_by_key(items), torch.flip(torch.arange(B, dtype=torch.int32, device=dev), dims=[0]))
        for order in orders:
            item, scores, sig = ops.score_l2_folded_gather(ws, enc_e, enc_r, items, t0, t1, q, user_o, A1, a1, Wmix, K, D, nR, n_entity, order=order)
            torch.cuda.synchronize()
            assert_close(item.cpu().numpy(), want_item.cpu().numpy(), f"item_emb B={B}", rtol=3e-5, atol=3e-5)
            assert_close(scores.cpu().numpy(), want_scores.cpu().numpy(), f"scores B={B}", rtol=3e-5, atol=3e-5)
            assert_close(sig.cpu().numpy(), want_sig.cpu().numpy(), f"sigmoid B={B}", rtol=3e-5, atol=1e-5)


def test_folded_form_with_thousands_of_relations(hip_lib):
    """The relation logits live in LDS next to the kernels' per-wave blocks: up to the 48 KB a launch gets without a function attribute
    (2 600 relations at K = 32) the folded form runs, beyond it the _supported queries say no and the callers keep the other kernels."""
    D, K, n_entity = 64, 32, 300
    assert ops.score_l2_folded_supported(D, K, n_entity, 2600) and not ops.score_l2_folded_supported(D, K, n_entity, 2800)
    assert not ops.gather_attn_l2_agg_supported(D, 64, n_entity, 1700) and ops.gather_attn_l2_agg_supported(D, 64, n_entity, 1600)
    nR = 2600
    rng = np.random.default_rng(5)
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)      # noqa: E731
    ae = torch.from_numpy(rng.integers(0, n_entity, (n_entity, K)).astype(np.int32)).to(dev)
    ar = torch.from_numpy(rng.integers(0, nR, (n_entity, K)).astype(np.int32)).to(dev)
    ae[:, K // 2:] = ae[:, : K // 2]                     # repeated slots
    ar[:, K // 2:] = ar[:, : K // 2]
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    E, W0, W1, W2, A0, A1, Wmix = f(n_entity, D), f(D, D), f(D, D), f(D, D), f(D, D), f(D, D), f(3 * D, D)
    B = 77
    items = torch.from_numpy(rng.integers(0, n_entity, B).astype(np.int64)).to(dev)
    q, t0, t1 = f(B, D), f(nR), f(nR)
    ws = ops.fold_tables(E, enc_e, enc_r, t0, W0, None, W1, None, W2, None, A0, None, Wmix, None, A1, K, nR)
    item, scores, sig = ops.score_l2_folded(ws, enc_e, enc_r, items, t0, t1, q, q, A1, None, Wmix, K, D, nR, n_entity)
    pt = ops.project_tables(E, W1, W2, None, None, A0, None, K, True)
    n0, n1 = ops.gather_attn_l2_prj(pt, enc_e, enc_r, items, t0, t1, q, B, 1, K, D, nR, n_entity)
    want_item, want_scores, _ = ops.l2_tail(E, items, q, q, n0, n1, W0, None, A0, None, A1, None, Wmix, None)
    torch.cuda.synchronize()
    assert_close(scores.cpu().numpy(), want_scores.cpu().numpy(), "scores, 2 600 relations", rtol=3e-5, atol=3e-5)
    assert_close(item.cpu().numpy(), want_item.cpu().numpy(), "item_emb, 2 600 relations", rtol=3e-5, atol=3e-5)


def test_order_by_key_is_a_bucket_partition(hip_lib):
    """mvin_order_by_key: a permutation in which the keys' buckets (low 14 bits) are contiguous and ascending; any key skew, both widths."""
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for B, hi, dt in ((1, 10, torch.int64), (1000, 5, torch.int32), (70000, 48091, torch.int64), (300000, 1 << 20, torch.int64)):
        keys = torch.randint(0, hi, (B,), device=dev, generator=g).to(dt)
        if B > 1000:
            keys[: B // 3] = 7                               # one key holds a third of the batch (Zipf's head)
        order = ops.order_by_key(keys)
        torch.cuda.synchronize()
        assert torch.equal(torch.sort(order.long()).values, torch.arange(B, device=dev))
        kb = keys[order.long()].long() & 16383
        assert bool((kb[1:] >= kb[:-1]).all())
