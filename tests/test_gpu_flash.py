"""-m gpu: the barrier-free ("flash") form of the grouped key addressing + user MLP (mvin_key_addressing_flash_fwd,
mvin_keyaddr_flash.hip) -- MVIN._key_addressing (model.py:161-240) with the user MLP of :232-236 folded in, every wave walking
tiles of 16 pairs of one user on its own over the static per-user records and the per-call table R_KGE[r] . E[e]
(mvin_project_relations).  Checked against a float64 evaluation of the reference's formulas pair by pair, against the kernel
that buckets every segment's ids itself (mvin_key_addressing_grouped_fwd) + mvin_linear_fwd, and -- through MVIN.forward_users
-- against the fp32 mirror of the reference graph."""
import numpy as np
import pytest
import torch

from mvin_amd import ops, synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params
from parity import assert_close

pytestmark = pytest.mark.gpu

SHAPES = [
    # (P, Nm, nR, n_user, B, with h-set, item dtype)
    (2, 64, 9, 40, 700, True, torch.int64),            # BASELINE C3's key-addressing shape; ~17 pairs per user: two tiles per slot
    (2, 64, 9, 2000, 9000, True, torch.int64),
    (2, 64, 9, 300, 1300, True, torch.int32),
    (2, 64, 9, 3000, 3500, True, torch.int64),         # most users appear once: one partly filled tile per slot
    (1, 64, 9, 500, 4000, True, torch.int64),          # one hop
    (1, 16, 39, 800, 5000, True, torch.int64),         # amazon-book's shape (BASELINE C4): one memory tile
    (2, 40, 9, 1500, 6000, True, torch.int64),         # padding memories (Nm = 40 -> 48 per hop)
    (3, 32, 9, 700, 3000, False, torch.int64),         # three hops, no h-set
    (4, 32, 7, 600, 5000, True, torch.int64),
    (8, 8, 5, 100, 900, True, torch.int64),            # eight hops of half a tile
    (2, 64, 25, 500, 4000, True, torch.int64),
    (2, 33, 70, 400, 3000, False, torch.int64),        # three memory tiles, more relations than a hop has memories
    (2, 56, 9, 600, 2500, True, torch.int64),
    (2, 64, 9, 5, 3000, True, torch.int64),            # 600 pairs per user: ten slots per segment
    (2, 64, 9, 7, 1, True, torch.int64),               # one pair
    (2, 1, 1, 3, 50, True, torch.int64),               # one memory, one relation
]
IDS = lambda s: "P%dNm%dnR%d_u%d_B%d%s" % (s[0], s[1], s[2], s[3], s[4], "" if s[5] else "_noset")      # noqa: E731


def reference_f64(E, R, w, W, b, uts, users, items, P, Nm, has_set, idx):
    """model.py:161-240 for the pairs ``idx``, float64, written from the reference's formulas: o_hset = softmax_m([h; user] . w_h) h
    (the user term and bias are constant over m: softmax shift invariance), per hop Rh_m = R_KGE[r_m] h_m, p = softmax_m(Rh_m . item),
    o = sum_m p_m t_m, user_o = concat(o_list) . user_mlp_matrix + bias."""
    Ed, Rd = E.double(), R.double()
    u = users[idx].long()
    v = Ed[items[idx].long()]                                   # [n, D]
    outs = []
    if has_set:
        h = Ed[uts[u, 0, 0].long()]                             # [n, Nm, D]
        p = torch.softmax(h @ w.double(), dim=-1)
        outs.append((p[..., None] * h).sum(1))
    for hop in range(P):
        h = Ed[uts[u, hop, 0].long()]
        t = Ed[uts[u, hop, 2].long()]
        Rm = Rd[uts[u, hop, 1].long()]                          # [n, Nm, D, D]
        Rh = torch.einsum("bmnk,bmk->bmn", Rm, h)
        p = torch.softmax(torch.einsum("bmn,bn->bm", Rh, v), dim=-1)
        outs.append((p[..., None] * t).sum(1))
    o_cat = torch.cat(outs, dim=1)
    return o_cat, o_cat @ W.double() + b.double()


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_flash_kernel_against_float64_and_the_bucketing_kernel(shape, hip_lib):
    P, Nm, nR, n_user, B, has_set, idt = shape
    D, n_entity = 64, 5000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(B + Nm)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5     # noqa: E731
    E, R = rnd(n_entity, D), rnd(nR, D, D) * 0.5
    w = rnd(D) if has_set else None
    n_o = P + (1 if has_set else 0)
    W, b = rnd(n_o * D, D) * 0.3, rnd(D)
    uts = torch.from_numpy(synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=B)).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g).to(idt)
    assert ops.key_addressing_flash_supported(D, P, Nm, nR, n_entity)
    rec = ops.build_user_records(uts, P, nR, n_entity)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    tabs = ops.key_addressing_flash_prepare(E, R, w, W, P)
    first = None
    for _ in range(2):                                        # twice: the scheduling scratch of a launch must not leak into the next
        user_o = torch.full((B, D), float("nan"), device=dev)
        ops.key_addressing_flash(E, tabs, rec, groups, items, P, Nm, nR, has_set, b, n_user, out=user_o)
        torch.cuda.synchronize()
        assert torch.isfinite(user_o).all()                   # every pair's row written, whatever its slot
        if first is None:
            first = user_o.clone()
        assert torch.equal(first, user_o)                     # same bits whoever draws which slot
    # float64, pair by pair, on a sample across the batch
    idx = torch.from_numpy(np.unique(np.random.default_rng(B).integers(0, B, 300))).to(dev)
    ref_cat, ref_uo = reference_f64(E, R, w, W, b, uts, users, items, P, Nm, has_set, idx)
    assert_close(user_o[idx].cpu().numpy(), ref_uo.cpu().numpy(), "user_o vs float64", rtol=1e-5, atol=2e-6)
    # the bucketing kernel + the MFMA linear kernel on the same inputs (other summation orders)
    a = torch.full((B, n_o * D), float("nan"), device=dev)
    if ops.key_addressing_grouped_supported(D, P, Nm, nR):
        ops.key_addressing_grouped(E, R, w, uts, groups, items, P, a, n_o * D, nR)
        assert_close(user_o.cpu().numpy(), ops.linear([a], W, D, bias=b).cpu().numpy(), "flash user_o vs linear kernel", rtol=1e-5, atol=2e-6)


def test_unsupported_shapes_are_refused(hip_lib):
    assert not ops.key_addressing_flash_supported(32, 2, 64, 9, 1000)          # D != 64
    assert not ops.key_addressing_flash_supported(64, 0, 64, 9, 1000)          # no hop
    assert not ops.key_addressing_flash_supported(64, 2, 65, 9, 1000)          # more than four memory tiles
    assert not ops.key_addressing_flash_supported(64, 2, 64, 4096, 1 << 20)    # row numbers beyond 31 bits
    assert ops.key_addressing_flash_supported(64, 2, 64, 9, 106389) and ops.key_addressing_flash_supported(64, 1, 16, 39, 113487)


def test_out_of_range_item_ids_are_clamped(hip_lib):
    """Device-resident ids are never validated per launch: an item id outside the table reads the last row (like every other kernel)."""
    dev = torch.device("cuda:0")
    P, Nm, nR, n_user, B, D, n_entity = 2, 64, 9, 10, 200, 64, 300
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5     # noqa: E731
    E, R, w, W, b = rnd(n_entity, D), rnd(nR, D, D), rnd(D), rnd(3 * D, D), rnd(D)
    uts = torch.from_numpy(synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=3)).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g)
    bad = items.clone()
    bad[::7] = n_entity + 12345
    items[::7] = n_entity - 1
    rec = ops.build_user_records(uts, P, nR, n_entity)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    tabs = ops.key_addressing_flash_prepare(E, R, w, W, P)
    x = ops.key_addressing_flash(E, tabs, rec, groups, items, P, Nm, nR, True, b, n_user)
    y = ops.key_addressing_flash(E, tabs, rec, groups, bad, P, Nm, nR, True, b, n_user)
    assert torch.equal(x, y)


CASES = [
    # (dim, K, H, P, Nm, nR, n_user, B, ablation)
    (64, 4, 2, 2, 64, 9, 40, 700, "all"),
    (64, 4, 2, 2, 40, 9, 1500, 6000, "all"),
    (64, 4, 2, 3, 32, 9, 700, 3000, "no_ps_o_ft"),
    (64, 4, 2, 1, 16, 39, 300, 1300, "all"),              # amazon-book's key-addressing shape: no records kernel, flash form only
    (64, 32, 2, 2, 64, 9, 60, 1500, "all"),               # the metric config's shapes (packed-tile kernel below)
]


@pytest.mark.parametrize("native", [True, False], ids=["one_native_call", "python_schedule"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "K%dP%dNm%dnR%d_%s" % (c[1], c[3], c[4], c[5], c[8]))
def test_forward_users_in_flash_form_against_the_oracle(case, native, hip_lib):
    """MVIN.forward_users with the flash form forced (both schedules: mvin_score_l2_fwd with ka_flash; op by op) against the fp32
    mirror of the reference's graph and against the form it replaces; the automatic rule by table size."""
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
    model.small_max_batch = 0
    if not native:
        model.native_l2_max_batch = 0
        model._profile = []                                  # event hooks requested: the Python schedule
    dev = model.device
    u_d, i_d, uts_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), torch.from_numpy(uts).to(dev)
    # automatic rule: users x P x Nm >= nR x n_entity
    model.ka_flash = None
    model.user_records(uts_d)
    rec = model._uts_records[3]
    assert model._ka_flash_for(uts_d, rec, B) == (min(B, n_user) * P * Nm >= nR * n_entity)
    assert not model._ka_flash_for(uts_d, rec, 1)
    model.ka_flash = True
    got = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    assert any(t is not None for t in model._ka_flash_ws.values()), "mvin_key_addressing_flash_prepare was not called"
    model.ka_flash = False
    plain = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    assert_close(got.user_o.cpu().numpy(), plain.user_o.cpu().numpy(), "user_o flash form vs the form it replaces", rtol=1e-5, atol=2e-6)
    mh, mr, mt = synth.memories_for(uts, users)
    ref = mirror_fp32.forward(args, params, adj_e, adj_r, users, items, mh, mr, mt)
    assert_close(got.user_o.cpu().numpy(), ref.user_o.numpy(), "user_o vs fp32 mirror")
    assert_close(got.scores.cpu().numpy(), ref.scores.numpy(), "scores vs fp32 mirror")
    # parameters changed in place between calls are seen (nothing of the tables is kept)
    model.ka_flash = True
    with torch.no_grad():
        model.entity_emb_matrix.mul_(1.1)
        model.user_mlp_matrix.add_(0.01)
        model.relation_emb_KGE_matrix.mul_(0.9)
    model.invalidate()
    after = model.forward_users(u_d, i_d, uts_d)
    model.ka_flash = False
    want = model.forward_users(u_d, i_d, uts_d)
    torch.cuda.synchronize()
    assert not torch.allclose(after.user_o, got.user_o)
    assert_close(after.user_o.cpu().numpy(), want.user_o.cpu().numpy(), "user_o after an in-place parameter change", rtol=1e-5, atol=2e-6)


N_FUZZ = int(__import__("os").environ.get("MVIN_FUZZ_CASES", "24"))      # a longer campaign: MVIN_FUZZ_CASES=400 pytest tests/test_gpu_flash.py -k fuzz


@pytest.mark.parametrize("i", range(N_FUZZ))
def test_fuzz_flash_kernel(i, hip_lib):
    """Random shapes inside the flash form's range (D = 64, fp32, 1 <= P <= 8, Nm <= 64) -- batch sizes from one pair to 30 000, one to
    3 000 users (one pair per user up to thousands), tables from 50 to 200 000 entities, with and without the h-set read, both id
    widths -- pair by pair on a sample against a float64 evaluation of MVIN._key_addressing + the user MLP (model.py:161-240)."""
    rng = np.random.default_rng(91000 + i)
    D = 64
    P = int(rng.choice([1, 1, 2, 2, 3, 4, 8]))
    Nm = int(rng.choice([1, 5, 8, 15, 16, 17, 31, 32, 40, 48, 63, 64]))
    nR = int(rng.choice([1, 2, 5, 9, 9, 12, 39]))
    n_user = int(rng.choice([1, 7, 100, 300, 700, 3000]))
    B = int(rng.choice([1, 17, 64, 65, 300, 2000, 9000, 30000]))
    n_entity = int(rng.choice([50, 5000, 200000]))
    has_set = bool(rng.random() < 0.8)
    idt = torch.int32 if rng.random() < 0.3 else torch.int64
    if not ops.key_addressing_flash_supported(D, P, Nm, nR, n_entity):
        pytest.skip("outside the flash form")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + i)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5     # noqa: E731
    E, R = rnd(n_entity, D), rnd(nR, D, D) * 0.5
    w = rnd(D) if has_set else None
    n_o = P + (1 if has_set else 0)
    W, b = rnd(n_o * D, D) * 0.3, rnd(D)
    uts = torch.from_numpy(synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=i)).to(dev)
    users = torch.randint(0, n_user, (B,), device=dev, generator=g)
    items = torch.randint(0, n_entity, (B,), device=dev, generator=g).to(idt)
    rec = ops.build_user_records(uts, P, nR, n_entity)
    groups = ops.group_pairs_by_user(users, n_user=n_user)
    tabs = ops.key_addressing_flash_prepare(E, R, w, W, P)
    user_o = torch.full((B, D), float("nan"), device=dev)
    ops.key_addressing_flash(E, tabs, rec, groups, items, P, Nm, nR, has_set, b, n_user, out=user_o)
    torch.cuda.synchronize()
    assert torch.isfinite(user_o).all(), f"case {i}: P={P} Nm={Nm} nR={nR} n_user={n_user} B={B} n_entity={n_entity} set={has_set}"
    idx = torch.from_numpy(np.unique(rng.integers(0, B, 200))).to(dev)
    _, ref_uo = reference_f64(E, R, w, W, b if b is not None else torch.zeros(D, device=dev), uts, users, items, P, Nm, has_set, idx)
    assert_close(user_o[idx].cpu().numpy(), ref_uo.cpu().numpy(),
                 f"case {i}: P={P} Nm={Nm} nR={nR} n_user={n_user} B={B} n_entity={n_entity} set={has_set}", rtol=1e-5, atol=2e-6)
