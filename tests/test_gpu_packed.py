"""-m gpu: the duplicate-slot encoding (mvin_encode_adjacency) and the packed-tile fused kernel over it
(mvin_gather_attn_l2_enc_fwd, mvin_fused_packed.hip) against the oracles and against the plain-adjacency kernels.

The reference's sampler repeats (neighbour, relation) slots whenever an entity has fewer than K edges
(data_loader_user_set.py:383-384); the encoded path adds the softmax weights of equal slots and gathers every distinct
row once.  Scores must agree with the fp32 mirror of the reference graph (which walks all K slots) within the path's
tolerance, on adjacencies with repeats, all-zero rows (one distinct slot, multiplicity K) and K distinct slots."""
import copy

import numpy as np
import pytest
import torch

from mvin_amd import ops, synth
from mvin_amd.config import make_args
from mvin_amd.model import MVIN
from mvin_amd.params import init_params
from oracle import prep_ref

from parity import assert_close, run_oracles

pytestmark = pytest.mark.gpu

DK = [(32, 16), (32, 32), (32, 64), (32, 128), (64, 16), (64, 32), (64, 64), (64, 128), (128, 16), (128, 32), (128, 64),
      (128, 128)]


def _shape(D, K, H=2, B=None):
    if B is None:
        B = max(2, min(37, 4096 // (K * K)))
    return dict(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=2, n_memory=8, batch_size=B)


def _run(args, case, params, dedup, table_dtype="f32", want_probs=False):
    model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                 params=params, device="cuda:0", table_dtype=table_dtype)
    model.dedup = dedup
    dev = model.device
    out = model.forward_device(
        torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
        [torch.from_numpy(m).to(dev) for m in case.memories_h],
        [torch.from_numpy(m).to(dev) for m in case.memories_r],
        [torch.from_numpy(m).to(dev) for m in case.memories_t], want_probs=want_probs)
    torch.cuda.synchronize()
    if dedup:
        assert model._enc_for_l2() is not None, "the encoded path was not taken"
    return model, out


def _check(args, case, params, table_dtype="f32", oracle_params=None, rtol=1e-5, atol=1e-6):
    _, out = _run(args, case, params, True, table_dtype)
    m, e = run_oracles(args, case, oracle_params or params)
    got = out.scores.cpu().numpy()
    assert_close(got, m.scores.numpy(), "packed-tile scores vs fp32 mirror", rtol=rtol, atol=atol)
    assert_close(out.item_embeddings.cpu().numpy(), m.item_embeddings.numpy(), "item_embeddings", rtol=rtol, atol=atol)
    err_hip = np.abs(got - e.scores).max()
    err_mir = np.abs(m.scores.numpy() - e.scores).max()
    assert err_hip <= 4 * err_mir + 1e-6, f"HIP-vs-fp64 {err_hip:.3e} > 4x mirror-vs-fp64 {err_mir:.3e}"
    return out


@pytest.mark.parametrize("K", [4, 16, 32, 64, 128])
@pytest.mark.parametrize("kind", ["repeats", "uniform"])
def test_encode_adjacency_bit_exact(K, kind, hip_lib):
    args = make_args(**_shape(32, K, B=2))
    case = synth.small_case(args, n_entity=300, n_relation=11, seed=K, zero_rows=5, repeats=kind == "repeats")
    ae = torch.from_numpy(case.adj_entity.astype(np.int32)).cuda()
    ar = torch.from_numpy(case.adj_relation.astype(np.int32)).cuda()
    enc_e, enc_r, cnt = ops.encode_adjacency(ae, ar)
    torch.cuda.synchronize()
    ref_e, ref_r, ref_c = prep_ref.encode_adjacency(case.adj_entity, case.adj_relation)
    assert np.array_equal(cnt.cpu().numpy(), ref_c)
    assert np.array_equal(enc_e.cpu().numpy(), ref_e)
    assert np.array_equal(enc_r.cpu().numpy(), ref_r)
    # and the encoding holds the row's multiset of slots
    dec = prep_ref.decode_adjacency(enc_e.cpu().numpy(), enc_r.cpu().numpy())
    plain = np.stack([case.adj_entity, case.adj_relation], -1)
    plain = np.array([sorted(map(tuple, row)) for row in plain])
    assert np.array_equal(dec, plain)
    if kind == "repeats" and K >= 16:
        assert ref_c.mean() < 0.8 * K


def test_encoder_clamps_out_of_range_ids(hip_lib):
    """a neighbour id outside [0, n_entity) or a relation id outside its 16-bit field used to be OR-ed into the
    packed words raw and overwrite the count / multiplicity bytes; the encoder now clamps them first, as the plain-adjacency
    kernels clamp where they index (same scores on both paths)."""
    args = make_args(**_shape(64, 32, B=8))
    case = synth.small_case(args, n_entity=300, n_relation=11, seed=3, zero_rows=5, repeats=True)
    ae, ar = case.adj_entity.astype(np.int32).copy(), case.adj_relation.astype(np.int32).copy()
    rng = np.random.default_rng(0)
    bad = rng.random(ae.shape) < 0.05
    ae[bad] = rng.choice(np.array([-7, 300, 1 << 24, (1 << 24) + 5, 2 ** 31 - 1], dtype=np.int64), size=int(bad.sum())).astype(np.int32)
    badr = rng.random(ar.shape) < 0.05
    ar[badr] = rng.choice(np.array([-1, 70000, 1 << 20], dtype=np.int64), size=int(badr.sum())).astype(np.int32)
    enc_e, enc_r, cnt = ops.encode_adjacency(torch.from_numpy(ae).cuda(), torch.from_numpy(ar).cuda())
    ref_e, ref_r, ref_c = prep_ref.encode_adjacency(ae, ar)
    assert np.array_equal(cnt.cpu().numpy(), ref_c) and np.array_equal(enc_e.cpu().numpy(), ref_e)
    assert np.array_equal(enc_r.cpu().numpy(), ref_r)
    c = cnt.cpu().numpy()
    assert c.min() >= 1 and c.max() <= 32
    assert np.array_equal((enc_r.cpu().numpy().view(np.uint32) >> 24), np.broadcast_to(c[:, None], ae.shape))
    # the model over an adjacency with ids BEYOND the table: the encoded path scores it as the clamped adjacency
    ae2 = case.adj_entity.astype(np.int64).copy()
    ae2[bad] = rng.choice(np.array([300, 1 << 24, (1 << 24) + 5, 2 ** 31 - 1], dtype=np.int64), size=int(bad.sum()))
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=4, random_agg_bias=True)
    case2, case3 = copy.copy(case), copy.copy(case)
    case2.adj_entity = ae2
    case3.adj_entity = np.minimum(ae2, case.n_entity - 1)
    _, enc_out = _run(args, case2, params, True)              # the encoding of the raw adjacency: ids clamped by the encoder
    _, enc_clamped = _run(args, case3, params, True)
    _, clamped_out = _run(args, case3, params, False)
    assert torch.equal(enc_out.scores, enc_clamped.scores)
    assert torch.allclose(enc_out.scores, clamped_out.scores, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("table", ["f32", "bf16"])
@pytest.mark.parametrize("kind", ["repeats", "uniform"])
@pytest.mark.parametrize("dk", DK, ids=lambda dk: "D%dK%d" % dk)
def test_packed_kernel_vs_oracles(dk, kind, table, hip_lib):
    D, K = dk
    args = make_args(**_shape(D, K))
    case = synth.small_case(args, n_user=16, n_entity=900, n_relation=7, seed=141 + D + K, zero_rows=4,
                            repeats=kind == "repeats")
    case.items[0] = np.flatnonzero((case.adj_entity == 0).all(1))[0]       # a zero-row parent
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=43, random_agg_bias=True)
    if table == "bf16":
        rounded = dict(params, entity_emb_matrix=torch.from_numpy(params["entity_emb_matrix"]).to(torch.bfloat16).float().numpy())
        _check(args, case, params, "bf16", oracle_params=rounded)
    else:
        _check(args, case, params)


@pytest.mark.parametrize("dk", [(32, 16), (64, 32), (64, 64)], ids=lambda dk: "D%dK%d" % dk)
def test_packed_kernel_depth3(dk, hip_lib):
    """h_hop = 3: K parents per pair, the query row shared by the K parents of a pair."""
    D, K = dk
    args = make_args(**_shape(D, K, H=3, B=3))
    case = synth.small_case(args, n_user=8, n_entity=700, n_relation=5, seed=151, zero_rows=3, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=52, random_agg_bias=True)
    _check(args, case, params)


@pytest.mark.parametrize("ablation", ["no_uor", "no_uo", "no_uor_and_no_kg_eh_uo", "no_uo_and_no_kg_eh_uo"])
@pytest.mark.parametrize("dk", [(32, 16), (64, 32), (128, 32)], ids=lambda dk: "D%dK%d" % dk)
def test_packed_kernel_without_attention_or_projection(dk, ablation, hip_lib):
    D, K = dk
    args = make_args(ablation=ablation, **_shape(D, K, B=11))
    case = synth.small_case(args, n_user=8, n_entity=600, n_relation=6, seed=161, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=62, random_agg_bias=True)
    _check(args, case, params)


@pytest.mark.parametrize("dk", [(32, 16), (64, 32), (64, 64)], ids=lambda dk: "D%dK%d" % dk)
@pytest.mark.parametrize("B", [1, 2, 255, 4097])
def test_packed_kernel_matches_plain_kernels_at_ragged_sizes(dk, B, hip_lib):
    """Parent counts around the kernel's work split (one parent, a partial last workgroup, tiles that straddle parents):
    encoded vs plain adjacency (independent programs) to fp32 round-off; and a pair's score does not depend on where in
    the batch it sits."""
    D, K = dk
    args = make_args(**_shape(D, K, B=B))
    case = synth.small_case(args, n_user=32, n_entity=2000, n_relation=9, seed=171 + B, repeats=True)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=72, random_agg_bias=True)
    _, a = _run(args, case, params, True)
    _, b = _run(args, case, params, False)
    assert_close(a.scores.cpu().numpy(), b.scores.cpu().numpy(), "encoded vs plain adjacency")
    _, c = _run(args, case, params, False, want_probs=True)
    assert_close(a.scores.cpu().numpy(), c.scores.cpu().numpy(), "encoded vs symmetric fused kernel")
    if B > 4:
        sl = slice(B - 3, B)
        c2 = copy.copy(case)
        for f in ("users", "items"):
            setattr(c2, f, getattr(case, f)[sl])
        for f in ("memories_h", "memories_r", "memories_t"):
            setattr(c2, f, [m[sl] for m in getattr(case, f)])
        a2 = _run(make_args(**_shape(D, K, B=3)), c2, params, True)[1]
        assert_close(a2.scores.cpu().numpy(), a.scores[sl].cpu().numpy(), "batch position", rtol=1e-6, atol=1e-7)


def test_auto_mode_takes_the_encoded_path_only_when_rows_repeat(hip_lib):
    args = make_args(**_shape(64, 32, B=8))
    params = None
    for rep, expect in ((True, True), (False, False)):
        case = synth.small_case(args, n_user=8, n_entity=500, n_relation=6, seed=5, repeats=rep)
        params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=6)
        model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                     params=params, device="cuda:0")
        assert (model._enc_for_l2() is not None) == expect
        assert model._enc_for_l2(want_probs=True) is None


def test_packed_kernel_is_deterministic(hip_lib):
    """No atomics anywhere in the packed-tile kernel (per-parent sums are MFMAs over fixed row orders, a straddling parent is
    carried in a fixed order): the same launch twice gives bit-identical outputs.  The same pairs at another batch position
    land at other rows of other tiles, so their per-parent sums associate differently: equal to fp32 round-off, not bit for
    bit (the plain-adjacency kernels, one parent per tile, are position-independent bit for bit)."""
    D, K, B = 64, 32, 3000
    args = make_args(**_shape(D, K, B=B))
    case = synth.small_case(args, n_user=32, n_entity=2500, n_relation=9, seed=191, repeats=True)
    rng = np.random.default_rng(7)
    dev = "cuda:0"
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).to(dev)
    E = f(case.n_entity, D)
    ae = torch.from_numpy(case.adj_entity.astype(np.int32)).to(dev)
    ar = torch.from_numpy(case.adj_relation.astype(np.int32)).to(dev)
    items = torch.from_numpy(case.items).to(dev)
    t0, t1, W1, W2, b1, b2, q, A0, a0 = f(9), f(9), f(D, D), f(D, D), f(D), f(D), f(B, D), f(D, D), f(D)
    enc_e, enc_r, _ = ops.encode_adjacency(ae, ar)
    run = lambda it, qq: ops.gather_attn_l2_enc(E, enc_e, enc_r, it, t0, t1, W1, W2, b1, b2, qq, A0, a0, it.shape[0], 1, K, D, 9)
    a0_, a1_ = run(items, q)
    b0_, b1_ = run(items, q)
    torch.cuda.synchronize()
    assert torch.equal(a0_, b0_) and torch.equal(a1_, b1_)
    sl = slice(1234, 1234 + 700)                            # other tiles, other workgroups, other straddles
    c0_, c1_ = run(items[sl].contiguous(), q[sl].contiguous())
    torch.cuda.synchronize()
    assert_close(c0_.cpu().numpy(), a0_[sl].cpu().numpy(), "nagg0 at another batch position", rtol=1e-6, atol=1e-7)
    assert_close(c1_.cpu().numpy(), a1_[sl].cpu().numpy(), "nagg1 at another batch position", rtol=1e-6, atol=1e-7)
