"""-m gpu: static per-user records (mvin_build_user_records) and the grouped key-addressing kernel over them
(mvin_key_addressing_grouped_rec_fwd, mvin_keyaddr_static.hip): the records bit for bit against the oracle's restatement, the
kernel against the kernel that buckets every segment's ids itself (same sums, another order), and against the CPU oracle of
MVIN._key_addressing (model.py:161-240)."""
import numpy as np
import pytest
import torch

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from parity import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,Nm,nR,n_user", [(2, 64, 9, 50), (1, 16, 39, 30), (3, 48, 9, 20), (2, 40, 9, 33), (2, 20, 100, 17),
                                            (8, 256, 70, 3), (1, 1, 1, 5), (2, 64, 1, 4)],
                         ids=lambda v: str(v))
def test_records_equal_the_oracle(P, Nm, nR, n_user, hip_lib):
    from mvin_amd import ops
    from oracle import prep_ref
    n_entity = 777
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=P + Nm + nR)
    # device ids are never validated per launch: out-of-range heads / tails / relations are clamped as unsigned words
    uts[0, 0, 0, 0] = n_entity + 5
    uts[0, 0, 2, 0] = -1
    uts[1 % n_user, P - 1, 1, Nm - 1] = nR + 3
    want = prep_ref.user_records(uts, nR, n_entity)
    L = prep_ref.user_records_layout(P, Nm, nR)
    assert ops.user_records_len(P, Nm, nR) == L["len"] == want.shape[1]
    got = ops.build_user_records(torch.from_numpy(uts).cuda(), P, nR, n_entity).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # every real row sits in exactly one bucket slot, in the bucket of its relation
    for u in range(n_user):
        slots = got[u, L["o_bidx"]:L["o_head"]]
        rows = sorted(int(v) for v in slots if v >= 0)
        assert rows == [h * L["NmP"] + m for h in range(P) for m in range(Nm)]
        assert got[u, 0] * 16 <= slots.size


def test_no_record_form_outside_the_shape_limits(hip_lib):
    from mvin_amd import ops
    assert ops.user_records_len(0, 16, 5) == 0 and ops.user_records_len(9, 16, 5) == 0 and ops.user_records_len(2, 257, 5) == 0
    assert not ops.user_records_supported(32, 2, 64, 9) and not ops.user_records_supported(64, 1, 16, 39)     # D != 64; < 64 rows
    assert ops.user_records_supported(64, 2, 64, 9) and not ops.user_records_supported(64, 2, 64, 9, table_bf16=True)
    assert not ops.user_records_supported(64, 3, 48, 9)                                                         # > 128 rows
    with pytest.raises(ValueError):
        ops.build_user_records(torch.zeros((3, 2, 3, 300), dtype=torch.int32, device="cuda:0"), 2, 5, 10)


SHAPES = [
    # (P, Nm, nR, n_user, B, with h-set, item dtype)
    (2, 64, 9, 40, 700, True, torch.int64),            # BASELINE C3's key-addressing shape; ~17 pairs per user: two tiles
    (2, 64, 9, 2000, 9000, True, torch.int64),         # eight segments per workgroup: every hand-over of the pipeline
    (2, 64, 9, 300, 1300, True, torch.int32),          # some workgroups with two segments, most with one
    (2, 64, 9, 3000, 3500, True, torch.int64),         # most users appear once: one tile per segment
    (1, 64, 9, 500, 4000, True, torch.int64),          # one hop
    (2, 40, 9, 1500, 6000, True, torch.int64),         # padding rows (Nm = 40 -> 48 per hop)
    (3, 32, 9, 700, 3000, False, torch.int64),         # three hops (the side waves take softmax / reads tiles too); no h-set
    (4, 32, 7, 600, 5000, True, torch.int64),
    (2, 64, 25, 500, 4000, True, torch.int64),         # more relations than resident fragments: R_KGE tiles fetched per task
    (2, 32, 70, 400, 3000, False, torch.int64),        # more relations than a hop has memories
    (2, 56, 9, 600, 2500, True, torch.int64),          # padding rows, 128 rows = the most the staging registers take
    (2, 64, 9, 5, 3000, True, torch.int64),            # 600 pairs per user: 38 tiles per segment, fewer segments than CUs
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "P%dNm%dnR%d_u%d_B%d%s" % (s[0], s[1], s[2], s[3], s[4], "" if s[5] else "_noset"))
def test_kernel_over_records_equals_the_bucketing_kernel(shape, hip_lib):
    from mvin_amd import ops
    P, Nm, nR, n_user, B, has_set, idt = shape
    D, n_entity = 64, 5000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(B + Nm)
    E = torch.rand((n_entity, D), device=dev, generator=g) - 0.5
    R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
    w = (torch.rand(D, device=dev, generator=g) - 0.5) if has_set else None
    uts = torch.from_numpy(synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=B)).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g).to(idt)
    assert ops.user_records_supported(D, P, Nm, nR)
    rec = ops.build_user_records(uts, P, nR, n_entity)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    n_o = P + (1 if has_set else 0)
    a = torch.full((B, n_o * D), float("nan"), device=dev)
    b = torch.full((B, n_o * D), float("nan"), device=dev)
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, a, n_o * D, nR)
    first = None
    for _ in range(2):                                       # twice: nothing of a launch may leak into the next
        b.fill_(float("nan"))
        ops.key_addressing_grouped(E, R, w, uts, groups, items, P, b, n_o * D, nR, records=rec)
        torch.cuda.synchronize()
        assert torch.isfinite(b).all()
        # same sums in another order (U_m one contraction step per MFMA instead of four, the h-set read four rows per step)
        assert_close(b.cpu().numpy(), a.cpu().numpy(), "records kernel vs bucketing kernel", rtol=1e-5, atol=1e-6)
        if first is None:
            first = b.clone()
        assert torch.equal(first, b)                         # and the same bits from launch to launch


CASES = [
    # (dim, K, H, P, Nm, nR, n_user, B, ablation)
    (64, 4, 2, 2, 64, 9, 40, 700, "all"),
    (64, 4, 2, 2, 40, 9, 1500, 6000, "all"),
    (64, 4, 2, 3, 32, 9, 700, 3000, "no_ps_o_ft"),
    (64, 4, 2, 2, 64, 30, 300, 1300, "all"),
]


@pytest.mark.parametrize("native", [True, False], ids=["one_native_call", "python_schedule"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "P%dNm%dnR%d_%s" % (c[3], c[4], c[5], c[8]))
def test_forward_users_over_records_against_the_oracle(case, native, hip_lib):
    """MVIN.forward_users builds the records once per user_triplet_set tensor and both of its schedules (mvin_score_l2_fwd with
    group_ws + user_records; op by op) take the kernel over them: against the fp32 mirror of the reference's graph."""
    from mvin_amd.model import MVIN
    from oracle import mirror_fp32
    D, K, H, P, Nm, nR, n_user, B, abl = case
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B, ablation=abl)
    n_entity = 500
    rng = np.random.default_rng(D + Nm + B)
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, K, seed=3)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=4)
    users = rng.integers(0, n_user, B, dtype=np.int64)
    items = rng.integers(0, n_entity, B, dtype=np.int64)
    params = init_params(args, n_user, n_entity, nR, seed=5, random_agg_bias=True)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, params=params, device="cuda:0")
    model.group_min_pairs_per_user = 0
    if not native:
        model._profile = []                                  # event hooks requested: the Python schedule
    dev = model.device
    u_d, i_d, uts_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), torch.from_numpy(uts).to(dev)
    got = model.forward_users(u_d, i_d, uts_d)
    assert model._uts_records is not None and model._uts_records[0]() is uts_d
    rec = model._uts_records[3]
    again = model.forward_users(u_d, i_d, uts_d)
    assert model._uts_records[3] is rec                      # built once per tensor
    model.static_user_records = False
    plain = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    assert torch.equal(got.scores, again.scores)
    assert_close(got.user_o.cpu().numpy(), plain.user_o.cpu().numpy(), "user_o records vs bucketing kernel")
    mh, mr, mt = synth.memories_for(uts, users)
    ref = mirror_fp32.forward(args, params, adj_e, adj_r, users, items, mh, mr, mt)
    assert_close(got.user_o.cpu().numpy(), ref.user_o.numpy(), "user_o vs fp32 mirror")
    assert_close(got.scores.cpu().numpy(), ref.scores.numpy(), "scores vs fp32 mirror")


def test_records_follow_the_tensor_not_its_address(hip_lib):
    """The cache is keyed by the tensor object and its version counter: an in-place edit of the ripple sets rebuilds them."""
    from mvin_amd.model import MVIN
    args = make_args(dim=64, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=32, batch_size=64)
    n_user, n_entity, nR = 20, 300, 6
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, 4, seed=1)
    uts = synth.ripple_sets(n_user, n_entity, nR, 2, 32, seed=2)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, device="cuda:0", seed=3)
    model.group_min_pairs_per_user = 0
    rng = np.random.default_rng(1)
    users = torch.from_numpy(rng.integers(0, n_user, 96)).cuda()
    items = torch.from_numpy(rng.integers(0, n_entity, 96)).cuda()
    uts_d = torch.from_numpy(uts).cuda()
    a = model.forward_users(users, items, uts_d).scores.clone()
    rec = model._uts_records[3]
    uts_d[:, :, 2] = torch.flip(uts_d[:, :, 2], dims=[-1]).clone()       # same sets, tails in another order: new reads
    uts_d[:, 0, 2, 0] = 7
    b = model.forward_users(users, items, uts_d).scores
    assert model._uts_records[3] is not rec
    model.static_user_records = False
    c = model.forward_users(users, items, uts_d).scores
    torch.cuda.synchronize()
    assert not torch.equal(a, b)
    np.testing.assert_allclose(b.cpu().numpy(), c.cpu().numpy(), rtol=1e-5, atol=1e-6)


N_FUZZ = int(__import__("os").environ.get("MVIN_FUZZ_CASES", "24"))      # a longer campaign: MVIN_FUZZ_CASES=400 pytest tests/test_gpu_user_records.py -k fuzz


@pytest.mark.parametrize("i", range(N_FUZZ))
def test_fuzz_kernel_over_records(i, hip_lib):
    """Random shapes inside the record kernel's range (D = 64, fp32, 64..128 ripple rows per user) against the bucketing kernel
    and, pair by pair on a sample, against a float64 evaluation of MVIN._key_addressing (model.py:161-240)."""
    from mvin_amd import ops
    rng = np.random.default_rng(77000 + i)
    D = 64
    while True:
        P = int(rng.choice([1, 2, 2, 3, 4, 8]))
        Nm = int(rng.choice([8, 12, 16, 20, 31, 32, 40, 48, 64, 64, 100, 128]))
        if 64 <= P * ((Nm + 15) // 16 * 16) <= 128:
            break
    nR = int(rng.choice([1, 2, 5, 9, 9, 11, 12, 24, 39]))
    n_user = int(rng.choice([1, 7, 100, 300, 700, 3000]))
    B = int(rng.choice([1, 17, 300, 2000, 9000, 30000]))
    n_entity = int(rng.choice([50, 5000, 200000]))
    has_set = bool(rng.random() < 0.8)
    idt = torch.int32 if rng.random() < 0.3 else torch.int64
    if not ops.user_records_supported(D, P, Nm, nR):
        pytest.skip("record too large for the LDS at this shape")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(i)
    E = torch.rand((n_entity, D), device=dev, generator=g) - 0.5
    R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
    w = (torch.rand(D, device=dev, generator=g) - 0.5) if has_set else None
    uts_h = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=i)
    if rng.random() < 0.3:                                   # a skewed relation histogram: one relation takes most memories
        mask = rng.random(uts_h[:, :, 1].shape) < 0.7
        uts_h[:, :, 1][mask] = int(rng.integers(0, nR))
    uts = torch.from_numpy(uts_h).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g).to(idt)
    rec = ops.build_user_records(uts, P, nR, n_entity)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    n_o = P + (1 if has_set else 0)
    a = torch.full((B, n_o * D), float("nan"), device=dev)
    b = torch.full((B, n_o * D), float("nan"), device=dev)
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, a, n_o * D, nR)
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, b, n_o * D, nR, records=rec)
    torch.cuda.synchronize()
    assert torch.isfinite(b).all()
    assert_close(b.cpu().numpy(), a.cpu().numpy(), "records kernel vs bucketing kernel", rtol=1e-5, atol=1e-6)
    # float64, straight from the equations, on up to 64 pairs spread over the batch
    pick = torch.linspace(0, B - 1, min(B, 64), device=dev).long()
    E64, R64 = E.double(), R.double()
    it, us = items[pick].long(), users[pick].long()
    want = []
    if has_set:
        h0 = E64[uts[us, 0, 0].long()]                       # [n, Nm, D]
        p = torch.softmax(h0 @ w.double(), dim=1)
        want.append((p[:, :, None] * h0).sum(1))
    for hop in range(P):
        h, r, t = (uts[us, hop, j].long() for j in range(3))
        Rh = torch.einsum("nmij,nmj->nmi", R64[r], E64[h])   # R_KGE[r] . h   (model.py:214-216)
        p = torch.softmax(torch.einsum("nmi,ni->nm", Rh, E64[it]), dim=1)
        want.append((p[:, :, None] * E64[t]).sum(1))
    want = torch.cat(want, dim=1)
    assert_close(b[pick].double().cpu().numpy(), want.cpu().numpy(), "records kernel vs float64 equations", rtol=1e-5, atol=2e-6)


# ---- the gathered form: U rows from mvin_project_relations' table (mvin_key_addressing_grouped_er_fwd) ----
def test_project_relations_holds_the_products(hip_lib):
    from mvin_amd import ops
    rng = np.random.default_rng(3)
    nE, nR, D = 333, 7, 64
    f = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32) * 0.3).cuda()
    E, R, w = f(nE, D), f(nR, D, D), f(D)
    ws = ops.project_relations(E, R, w).double().cpu()
    want = torch.einsum("rnk,ek->ren", R.double().cpu(), E.double().cpu())         # R_KGE[r] . E[e]
    assert_close(ws[:nR * nE * D].view(nR, nE, D).numpy(), want.numpy(), "R_KGE[r] . E[e]", rtol=2e-5, atol=2e-6)
    assert_close(ws[nR * nE * D:nR * nE * D + nE].numpy(), (E.double().cpu() @ w.double().cpu()).numpy(), "E[e] . w", rtol=1e-5, atol=1e-6)


ER_SHAPES = [s for s in SHAPES if s[1] <= 64 or not s[5]]


@pytest.mark.parametrize("shape", ER_SHAPES, ids=lambda s: "P%dNm%dnR%d_u%d_B%d%s" % (s[0], s[1], s[2], s[3], s[4], "" if s[5] else "_noset"))
def test_gathered_form_equals_the_kernel_over_records(shape, hip_lib):
    from mvin_amd import ops
    P, Nm, nR, n_user, B, has_set, idt = shape
    D, n_entity = 64, 5000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(B + Nm)
    E = torch.rand((n_entity, D), device=dev, generator=g) - 0.5
    R = torch.rand((nR, D, D), device=dev, generator=g) - 0.5
    w = (torch.rand(D, device=dev, generator=g) - 0.5) if has_set else None
    uts_np = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=B)
    uts_np[0, 0, 0, 0] = n_entity + 5                        # out-of-range ids: clamped by the records
    uts_np[0, P - 1, 1, Nm - 1] = nR + 3
    uts = torch.from_numpy(uts_np).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g).to(idt)
    assert ops.key_addressing_grouped_er_supported(D, P, Nm, nR, n_entity, has_set)
    rec = ops.build_user_records(uts, P, nR, n_entity)
    er = ops.project_relations(E, R, w)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    n_o = P + (1 if has_set else 0)
    a = torch.full((B, n_o * D), float("nan"), device=dev)
    b = torch.full((B, n_o * D), float("nan"), device=dev)
    ops.key_addressing_grouped(E, R, w, uts, groups, items, P, a, n_o * D, nR, records=rec)
    first = None
    for _ in range(2):
        b.fill_(float("nan"))
        ops.key_addressing_grouped(E, R, w, uts, groups, items, P, b, n_o * D, nR, records=rec, er=er)
        torch.cuda.synchronize()
        assert torch.isfinite(b).all()
        assert_close(b.cpu().numpy(), a.cpu().numpy(), "gathered form vs kernel over records", rtol=1e-5, atol=1e-6)
        if first is None:
            first = b.clone()
        assert torch.equal(first, b)


@pytest.mark.parametrize("native", [True, False], ids=["one_native_call", "python_schedule"])
@pytest.mark.parametrize("case", [c for c in CASES if c[4] <= 64], ids=lambda c: "P%dNm%dnR%d_%s" % (c[3], c[4], c[5], c[8]))
def test_forward_users_gathered_form_against_the_oracle(case, native, hip_lib):
    from mvin_amd.model import MVIN
    from oracle import mirror_fp32
    D, K, H, P, Nm, nR, n_user, B, abl = case
    args = make_args(dim=D, neighbor_sample_size=K, h_hop=H, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=B, ablation=abl)
    n_entity = 500
    rng = np.random.default_rng(D + Nm + B)
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, K, seed=3)
    uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=4)
    users = rng.integers(0, n_user, B, dtype=np.int64)
    items = rng.integers(0, n_entity, B, dtype=np.int64)
    params = init_params(args, n_user, n_entity, nR, seed=5, random_agg_bias=True)
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, params=params, device="cuda:0")
    model.group_min_pairs_per_user = 0
    model.ka_er = True
    if not native:
        model._profile = []
    dev = model.device
    u_d, i_d, uts_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), torch.from_numpy(uts).to(dev)
    got = model.forward_users(u_d, i_d, uts_d)
    assert model._ka_er_for(uts_d, model.user_records(uts_d)), "the gathered form was not taken"
    model.ka_er = False
    plain = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    assert_close(got.user_o.cpu().numpy(), plain.user_o.cpu().numpy(), "user_o gathered form vs kernel over records")
    mh, mr, mt = synth.memories_for(uts, users)
    ref = mirror_fp32.forward(args, params, adj_e, adj_r, users, items, mh, mr, mt)
    assert_close(got.user_o.cpu().numpy(), ref.user_o.numpy(), "user_o vs fp32 mirror")
    assert_close(got.scores.cpu().numpy(), ref.scores.numpy(), "scores vs fp32 mirror")
    # nothing of the table is kept between calls: an in-place change of R_KGE shows in the next call
    model.ka_er = True
    with torch.no_grad():
        model.relation_emb_KGE_matrix.mul_(1.5)
    changed = model.forward_users(u_d, i_d, uts_d)
    model.ka_er = False
    want = model.forward_users(u_d, i_d, uts_d)
    assert not torch.allclose(changed.user_o, got.user_o)
    assert_close(changed.user_o.cpu().numpy(), want.user_o.cpu().numpy(), "user_o after an in-place change of R_KGE")


def test_gathered_form_is_on_request_only(hip_lib):
    from mvin_amd.model import MVIN
    args = make_args(dim=64, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=64, batch_size=64)
    n_user, n_entity, nR = 40, 300, 6
    adj_e, adj_r = synth.uniform_adjacency(n_entity, nR, 4, seed=1)
    uts_d = torch.from_numpy(synth.ripple_sets(n_user, n_entity, nR, 2, 64, seed=2)).cuda()
    model = MVIN(args, n_user, n_entity, nR, adj_e, adj_r, device="cuda:0", seed=3)
    rec = model.user_records(uts_d)
    assert not model._ka_er_for(uts_d, rec)                  # measured neutral on the default (two-stream) line: opt-in
    model.ka_er = True
    assert model._ka_er_for(uts_d, rec) and not model._ka_er_for(uts_d, None)
