"""-m gpu: key addressing with the pairs grouped by user (mvin_key_addressing_grouped_fwd, MVIN.forward_users)
against the per-pair kernel fed with the same users' ripple sets and against the CPU oracle."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from parity import assert_close

pytestmark = pytest.mark.gpu

CASES = [
    # (dim, K, H, P, Nm, nR, n_user, B, ablation, table_dtype)
    (64, 4, 2, 2, 64, 9, 40, 700, "all", "f32"),          # C3 key-addressing shape, ~17 pairs per user
    (64, 4, 2, 1, 16, 39, 11, 333, "all", "f32"),          # amazon-book shape: nR > P*Nm (relation compaction)
    (16, 4, 1, 2, 8, 5, 300, 257, "all", "f32"),           # most users appear once; B not a multiple of 16
    (32, 4, 2, 2, 12, 6, 5, 100, "no_ps_o_ft", "f32"),     # Nm not a multiple of the rows per wave, no h-set
    (128, 4, 2, 1, 16, 7, 9, 90, "all", "f32"),
    (64, 4, 2, 2, 32, 9, 13, 200, "ps_only", "f32"),
    (64, 4, 2, 2, 32, 9, 13, 200, "no_kg_eh_uo", "f32"),
    (64, 4, 2, 2, 64, 9, 21, 300, "all", "bf16"),
    (128, 4, 2, 1, 16, 39, 9, 150, "all", "bf16"),
    # more user segments than workgroups (256): the LDS-DMA form of the dense kernel hands ids / head rows from a segment to the next
    (64, 4, 2, 2, 40, 9, 1500, 6000, "all", "f32"),        # 5-6 segments per workgroup, padding rows (Nm = 40 -> 48)
    (64, 4, 2, 3, 48, 9, 700, 3000, "no_ps_o_ft", "f32"),  # 144 rows = the most the DMA form takes; no h-set
    (64, 4, 2, 2, 64, 9, 300, 1300, "all", "f32"),         # some workgroups with two segments, most with one
    (64, 4, 2, 2, 80, 9, 600, 2500, "all", "f32"),         # 160 rows: beyond it -> register staging, same loop
    (64, 4, 2, 2, 64, 9, 900, 3000, "all", "bf16"),        # bf16 table: register staging
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "d%dP%dNm%dnR%d_%s_%s" % (c[0], c[3], c[4], c[5], c[8], c[9]))
def test_grouped_matches_per_pair_and_oracle(case, hip_lib):
    from mvin_amd.model import MVIN
    from oracle import mirror_fp32
    D, K, H, P, Nm, nR, n_user, B, abl, tdt = case
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B, ablation=abl)
    n_entity = 500
    rng = np.random.default_rng(D + Nm + B)
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, K, seed=3)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=4)
    users = rng.integers(0, n_user, B, dtype=np.int64)
    users[:3] = users[0]                                    # a run of equal users at the front of the batch
    items = rng.integers(0, n_entity, B, dtype=np.int64)
    params = init_params(args, n_user, n_entity, nR, seed=5, random_agg_bias=True)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, params=params, device="cuda:0", table_dtype=tdt)
    model.group_min_pairs_per_user = 0                     # always take the grouped kernel here
    dev = model.device
    u_d, i_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
    uts_d = torch.from_numpy(uts).to(dev)
    got = model.forward_users(u_d, i_d, uts_d)
    mh, mr, mt = synth.memories_for(uts, users)
    ref_dev = model.forward_device(u_d, i_d, [torch.from_numpy(m).to(dev) for m in mh],
                                   [torch.from_numpy(m).to(dev) for m in mr], [torch.from_numpy(m).to(dev) for m in mt])
    torch.cuda.synchronize()
    tol = dict(rtol=1e-5, atol=1e-6) if tdt == "f32" else dict(rtol=1e-5, atol=2e-6)
    assert_close(got.user_o.cpu().numpy(), ref_dev.user_o.cpu().numpy(), "user_o grouped vs per-pair kernel", **tol)
    assert_close(got.scores.cpu().numpy(), ref_dev.scores.cpu().numpy(), "scores grouped vs per-pair kernel", **tol)
    if tdt == "f32":
        ref = mirror_fp32.forward(args, params, adj_e, adj_r, users, items, mh, mr, mt)
        assert_close(got.user_o.cpu().numpy(), ref.user_o.numpy(), "user_o vs fp32 mirror")
        assert_close(got.scores.cpu().numpy(), ref.scores.numpy(), "scores vs fp32 mirror")


def test_grouping_is_a_permutation_with_segments(hip_lib):
    from mvin_amd import ops
    users = torch.tensor([5, 1, 5, 5, 2, 1, 9, 9, 0], device="cuda:0")
    seg_user, seg_ptr, nseg, perm = ops.group_pairs_by_user(users)
    n = int(nseg.item())
    assert n == 5
    assert seg_user[:n].tolist() == [0, 1, 2, 5, 9]
    assert seg_ptr[:n + 1].tolist() == [0, 1, 3, 4, 7, 9]
    assert sorted(perm.tolist()) == list(range(9))
    for s in range(n):
        assert all(int(users[perm[p]]) == seg_user[s] for p in range(int(seg_ptr[s]), int(seg_ptr[s + 1])))


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("B,n_user", [(9, 10), (1, 5), (5000, 37), (70000, 3000), (1025, 1025), (4096, 100000),
                                      (200000, 3000), (131072, 37000), (300, 37500), (300, 38500), (140000, 23553),
                                      (32768, 1000), (32769, 1000)])
def test_native_grouping_matches_the_sort_based_one(B, n_user, dtype, hip_lib):
    """mvin_group_pairs_by_user (counting sort: ONE workgroup with the counters in LDS up to 32 768 pairs and 150 KB of
    counters, three kernels beyond, their scan staged in LDS when the counters fit -- the sizes straddle every limit) against the torch.sort-based grouping: same users,
    same segment boundaries, and every segment holds the same SET of pairs (order inside a segment is free)."""
    from mvin_amd import ops
    g = torch.Generator(device="cuda:0")
    g.manual_seed(B + n_user)
    users = torch.randint(0, n_user, (B,), device="cuda:0", generator=g).to(dtype)
    su_a, sp_a, ns_a, pi_a = ops.group_pairs_by_user(users, n_user=n_user)
    su_b, sp_b, ns_b, pi_b = ops.group_pairs_by_user(users.long())
    n = int(ns_b.item())
    assert int(ns_a.item()) == n
    assert torch.equal(su_a[:n], su_b[:n]) and torch.equal(sp_a[:n + 1], sp_b[:n + 1])
    assert torch.equal(torch.sort(pi_a.long())[0], torch.arange(B, device="cuda:0"))
    # a pair sits in the segment of its user
    seg_of_pos = torch.bucketize(torch.arange(B, device="cuda:0"), sp_a[1:n + 1].long(), right=True)
    assert torch.equal(su_a[:n].long()[seg_of_pos], users.long()[pi_a.long()])


def test_device_feeder_uses_the_grouped_path(hip_lib):
    from mvin_amd import harness
    from mvin_amd.model import MVIN
    args = make_args(dim=64, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=32, batch_size=64)
    n_user, n_entity, nR = 20, 300, 6
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, 4, seed=1)
    uts = synth.ripple_sets(n_user, n_entity, nR, 2, 32, seed=2)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, device="cuda:0", seed=3)
    model.group_min_pairs_per_user = 0
    feeder = harness.DeviceFeeder(model, uts)
    rng = np.random.default_rng(0)
    users, items = rng.integers(0, n_user, 100), rng.integers(0, n_entity, 100)
    a = feeder.scores(users, items).cpu().numpy()
    mh, mr, mt = feeder.memories(torch.from_numpy(users).to(model.device))
    b = model.forward_device(torch.from_numpy(users).to(model.device), torch.from_numpy(items).to(model.device),
                             mh, mr, mt).scores_normalized.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def _oob_setup(P=2, Nm=32):
    from mvin_amd.model import MVIN
    args = make_args(dim=64, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=64)
    n_user, n_entity, nR = 20, 300, 6
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, 4, seed=1)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=2)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, device="cuda:0", seed=3)
    rng = np.random.default_rng(1)
    users, items = rng.integers(0, n_user, 96), rng.integers(0, n_entity, 96)
    return model, uts, users, items


@pytest.mark.parametrize("grouped", [True, False], ids=["grouped", "users_feed"])
def test_out_of_range_device_user_ids_are_clamped_not_dropped(grouped, hip_lib):
    """device-resident ids are not validated per batch; a user id outside [0, n_user) must neither leave
    an output row unwritten (grouped form) nor index user_triplet_set out of bounds (per-pair form): it is clamped."""
    model, uts, users, items = _oob_setup()
    model.group_min_pairs_per_user = 0 if grouped else 10 ** 9
    dev = model.device
    uts_d = torch.from_numpy(uts).to(dev)
    bad = users.copy()
    bad[5], bad[17], bad[40] = 10 ** 6, -3, 20                     # way above, negative, n_user exactly
    clamped = np.clip(bad, 0, 19)
    got = model.forward_users(torch.from_numpy(bad).to(dev), torch.from_numpy(items).to(dev), uts_d)
    want = model.forward_users(torch.from_numpy(clamped).to(dev), torch.from_numpy(items).to(dev), uts_d)
    torch.cuda.synchronize()
    assert torch.isfinite(got.scores).all()
    assert torch.equal(got.user_o, want.user_o) and torch.equal(got.scores, want.scores)


def test_device_id_validation_raises_like_the_host_feed(hip_lib):
    """MVIN.validate_device_ids (MVIN_CHECK_IDS=1): IndexError on out-of-range device ids, as _to_device_ids raises for
    host feeds and the reference's CPU tf.gather raises InvalidArgument; a user_triplet_set is always checked once."""
    model, uts, users, items = _oob_setup()
    dev = model.device
    u_d, i_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
    bad_uts = uts.copy()
    bad_uts[3, 1, 1, 7] = 6                                          # relation id == n_relation
    with pytest.raises(IndexError):
        model.forward_users(u_d, i_d, torch.from_numpy(bad_uts).to(dev))
    bad_uts = uts.copy()
    bad_uts[0, 0, 0, 0] = 300                                        # head id == n_entity
    with pytest.raises(IndexError):
        model.forward_users(u_d, i_d, torch.from_numpy(bad_uts).to(dev))
    with pytest.raises(ValueError):
        model.forward_users(u_d, i_d, torch.from_numpy(np.ascontiguousarray(uts[:, :1])).to(dev))   # wrong hop count
    uts_d = torch.from_numpy(uts).to(dev)
    model.forward_users(u_d, i_d, uts_d)                             # valid: passes, and is cached
    assert model._uts_ok is not None
    model.validate_device_ids = True
    bu = u_d.clone()
    bu[2] = 20
    with pytest.raises(IndexError, match="user_indices"):
        model.forward_users(bu, i_d, uts_d)
    bi = i_d.clone()
    bi[9] = -1
    with pytest.raises(IndexError, match="item_indices"):
        model.forward_users(u_d, bi, uts_d)
    mh, mr, mt = synth.memories_for(uts, users)
    mr_d = [torch.from_numpy(m).to(dev) for m in mr]
    mr_d[1][4, 4] = 6
    with pytest.raises(IndexError, match="memories_r"):
        model.forward_device(u_d, i_d, [torch.from_numpy(m).to(dev) for m in mh], mr_d,
                             [torch.from_numpy(m).to(dev) for m in mt])
    model.forward_users(u_d, i_d, uts_d)                             # valid ids still pass


@pytest.mark.parametrize("shape", [(64, 2, 64, 9), (64, 3, 48, 9), (32, 2, 12, 6), (16, 2, 8, 5), (128, 1, 16, 7)],
                         ids=lambda s: "d%dP%dNm%d" % s[:3])
@pytest.mark.parametrize("feed", ["users", "pairs"])
def test_out_of_range_device_ids_are_clamped(shape, feed, hip_lib):
    """Device-resident feeds are not validated per batch (MVIN.validate_device_ids = False): every kernel that indexes
    the entity table with a head / tail / item id clamps it into the table -- a batch with ids beyond n_entity must
    score exactly like the batch with those ids clamped (and not fault)."""
    from mvin_amd.model import MVIN
    D, P, Nm, nR = shape
    n_user, n_entity, B, K = 40, 300, 500, 4
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=2, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B)
    rng = np.random.default_rng(D + P)
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, K, seed=3)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=4)
    bad = uts.copy()
    hit = rng.random(bad.shape) < 0.05
    hit[:, :, 1, :] = False                                 # relations stay valid (they are clamped too, tested elsewhere)
    bad[hit] += n_entity * 7                                # far beyond the table
    users = rng.integers(0, n_user, B, dtype=np.int64)
    items = rng.integers(0, n_entity, B, dtype=np.int64)
    bad_items = items.copy()
    bad_items[::17] += n_entity * 3
    params = init_params(args, n_user, n_entity, nR, seed=5, random_agg_bias=True)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, params=params, device="cuda:0")
    model.validate_device_ids = False
    model._check_uts = lambda t: None                      # (the once-per-tensor range check would raise the reference's IndexError)
    model.group_min_pairs_per_user = 0
    dev = model.device

    def run(u3, it):
        u_d, i_d = torch.from_numpy(users).to(dev), torch.from_numpy(it).to(dev)
        if feed == "users":
            return model.forward_users(u_d, i_d, torch.from_numpy(u3).to(dev))
        mh, mr, mt = synth.memories_for(u3, users)
        return model.forward_device(u_d, i_d, [torch.from_numpy(m).to(dev) for m in mh],
                                    [torch.from_numpy(m).to(dev) for m in mr], [torch.from_numpy(m).to(dev) for m in mt])

    got = run(bad, bad_items)
    torch.cuda.synchronize()
    ref = run(np.minimum(bad, n_entity - 1), np.minimum(bad_items, n_entity - 1))
    torch.cuda.synchronize()
    assert torch.isfinite(got.scores).all()
    assert torch.equal(got.scores, ref.scores)


@pytest.mark.parametrize("B", [300, 5000, 70000])
def test_grouped_native_call_equals_python_schedule(B, hip_lib):
    """The grouped users feed enqueued by ONE native call (mvin_score_l2_fwd with group_ws: device-side sort + grouped key
    addressing + user MLP + fused kernel + tail) against the same launches issued one by one from Python: same kernels,
    same order -> identical scores; batch sizes below and above the per-pair native path's cap."""
    from mvin_amd.model import MVIN
    D, K, P, Nm, nR, n_user, n_entity = 64, 32, 2, 64, 9, 60, 3000
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=2, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B)
    rng = np.random.default_rng(B)
    case = synth.small_case(make_args(**dict(vars(args), batch_size=4)), n_user=n_user, n_entity=n_entity, n_relation=nR,
                            seed=3, repeats=True)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=4)
    users = rng.integers(0, n_user, B, dtype=np.int64)
    items = rng.integers(0, n_entity, B, dtype=np.int64)
    params = init_params(args, n_user, n_entity, nR, seed=5, random_agg_bias=True)
    model = MVIN(args, n_user, n_entity, nR, case.adj_entity, case.adj_relation, params=params, device="cuda:0")
    model.group_min_pairs_per_user = 0
    dev = model.device
    u_d, i_d, uts_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), torch.from_numpy(uts).to(dev)
    calls = []
    orig = model._score_l2_native
    model._score_l2_native = lambda *a_, **k_: (calls.append(k_.get("grouped")), orig(*a_, **k_))[1]
    native = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    assert calls == [True], "the grouped batch did not take the native call"
    model._profile = []                                     # event hooks requested: the Python schedule
    python = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    model._profile = None
    assert calls == [True]
    assert torch.equal(native.scores, python.scores) and torch.equal(native.user_o, python.user_o)
    assert torch.equal(native.item_embeddings, python.item_embeddings)
