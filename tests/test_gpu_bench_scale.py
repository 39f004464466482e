"""-m gpu: parity AT THE SIZES bench.py TIMES, IN THE FORM bench.py TIMES (the automatic rules: no kernel pinned by this
module) -- the default metric workload (C3: 524 288 last-fm-shaped pairs, users feed, KG adjacency with repeats ->
PROJECTED-TABLES form of the packed-tile kernel over the encoded adjacency (B K >= 16 n_entity), grouped key addressing + user
MLP in their FLASH form over the static per-user records (users x P x Nm >= nR x n_entity)) and the C2 (projected tables, wave-per-parent kernel) / C4 (K = 64: unprojected) bench sizes:
  * a sample of pairs drawn across the WHOLE batch against the fp32 mirror of the reference graph (oracle/mirror_fp32.py),
  * the users feed (grouped key addressing) against the per-pair feed (the reference's feed_dict contents) over the
    whole batch,
  * the encoded adjacency against the plain one over the whole batch,
all within 1e-5 |ref| + 1e-6 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from mvin_amd.config import make_args
from oracle import mirror_fp32

from parity import assert_close
from test_gpu_properties import setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B,n_ref", [("C3", 524288, 2048), ("C2", 524288, 2048), ("C4", 32768, 256)])
def test_bench_scale_parity(name, B, n_ref, hip_lib):
    args, case, params, model = setup(name, B=B)
    dev = model.device
    users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    uts = torch.from_numpy(case.user_triplet_set).to(dev)
    assert model.prj is None and model.dedup is None, "this module runs the automatic rules (tests/conftest.py must not pin a form)"
    took_prj = (model._enc_for_l2(n_parents=B) is not None or model._prj_plain_ok()) and model._prj_for_l2(B)
    # (C4 -- K = 64 -- takes the tables as the input of the per-entity aggregates; its 39 relations keep key addressing on the records kernel)
    assert took_prj, f"{name}: projected-tables form taken = {took_prj}"
    got = model.forward_users(users, items, uts)
    torch.cuda.synchronize()
    # every bench config of dim <= 64 takes the folded tail over per-entity aggregates (mvin_fold_tables -> mvin_score_l2_folded_fwd): the
    # workspace of that call exists.  (The aggregates + tail-kernel form exists at dim 64 only: C3, C4.)
    took_fold = bool(took_prj and model._fold_for(model._enc_for_l2(n_parents=B)))
    took_agg = bool(took_prj and model._agg_for(model._enc_for_l2(n_parents=B)))
    assert took_fold, f"{name}: folded-tail form taken = {took_fold}"
    assert took_agg == (name in ("C3", "C4")), f"{name}: per-entity aggregates form available = {took_agg}"
    assert any(t is not None for t in model._fold_ws.values()), "mvin_fold_tables was not called"
    if name == "C3":
        assert model._uts_records is not None, "static per-user records were expected at C3"
        assert model._ka_flash_for(uts, model._uts_records[3], B) and any(t is not None for t in model._ka_flash_ws.values()), \
            "the flash form of key addressing (mvin_key_addressing_flash_fwd) was expected at C3"
    s_users = got.scores.cpu().numpy()
    assert np.isfinite(s_users).all()
    # (a) sample across the batch vs the mirror
    rng = np.random.default_rng(11)
    idx = np.sort(rng.choice(B, n_ref, replace=False))
    sargs = make_args(**dict(vars(args), batch_size=n_ref))
    ref = mirror_fp32.forward(sargs, params, case.adj_entity, case.adj_relation, case.users[idx], case.items[idx],
                              [m[idx] for m in case.memories_h], [m[idx] for m in case.memories_r],
                              [m[idx] for m in case.memories_t])
    assert_close(s_users[idx], ref.scores.numpy(), f"{name}: {n_ref} pairs sampled across {B} vs fp32 mirror")
    assert_close(got.user_o[torch.from_numpy(idx).to(dev)].cpu().numpy(), ref.user_o.numpy(), f"{name}: user_o sample")
    # (b) users feed vs per-pair feed, whole batch
    per_pair = model.forward_device(users, items, [torch.from_numpy(m).to(dev) for m in case.memories_h],
                                    [torch.from_numpy(m).to(dev) for m in case.memories_r],
                                    [torch.from_numpy(m).to(dev) for m in case.memories_t])
    torch.cuda.synchronize()
    assert_close(s_users, per_pair.scores.cpu().numpy(), f"{name}: users feed vs per-pair feed over {B} pairs")
    del per_pair
    # (c) encoded vs plain adjacency, whole batch (only where the encoded path is the one taken)
    took_enc = model._enc_for_l2(n_parents=B) is not None
    if name in ("C3", "C4"):
        assert took_enc, "the bench-size launch is expected on the packed-tile kernel"
    model.dedup = not took_enc
    other = model.forward_users(users, items, uts)
    torch.cuda.synchronize()
    if model._enc_for_l2(n_parents=B) is not None or took_enc:
        assert_close(s_users, other.scores.cpu().numpy(), f"{name}: encoded vs plain adjacency over {B} pairs")
    model.dedup = None
    if took_prj:
        # (d) the projected-tables form against the unprojected form of the same two levels, whole batch
        model.prj = False
        other = model.forward_users(users, items, uts)
        torch.cuda.synchronize()
        assert_close(s_users, other.scores.cpu().numpy(), f"{name}: projected tables vs per-row projection over {B} pairs")
        model.prj = None
    if took_fold:
        # (d') the folded tail over per-entity aggregates against the kernels over the projected tables themselves (+ tail kernel), whole batch
        model.agg = False
        other = model.forward_users(users, items, uts)
        torch.cuda.synchronize()
        assert_close(s_users, other.scores.cpu().numpy(), f"{name}: per-entity aggregates vs the kernel over the projected tables, {B} pairs")
        model.agg = None
    if took_agg:
        # (d'') the folded-tail form the rule took against aggregates + mvin_l2_tail_fwd, whole batch, scores and item embeddings
        model.fold = False
        other = model.forward_users(users, items, uts)
        torch.cuda.synchronize()
        assert any(t is not None for t in model._agg_tables.values()), "mvin_entity_aggregates was not called"
        assert_close(s_users, other.scores.cpu().numpy(), f"{name}: folded tail vs aggregates + tail kernel, {B} pairs")
        assert_close(got.item_embeddings[:4096].cpu().numpy(), other.item_embeddings[:4096].cpu().numpy(), f"{name}: item embeddings, folded tail")
        model.fold = None
    if name == "C3":
        # (e) the other forms of the grouped key addressing, whole batch: the kernel over the static records + the MLP launch, and its
        #     gathered form (R_KGE[r] . E[h] per (relation, entity), rebuilt per call) -- against the flash form the automatic rule took
        for ka_er in (False, True):
            model.ka_flash, model.ka_er = False, ka_er
            other = model.forward_users(users, items, uts)
            torch.cuda.synchronize()
            assert_close(s_users, other.scores.cpu().numpy(), f"{name}: flash form vs records kernel (gathered U rows: {ka_er}) over {B} pairs")
            assert_close(other.user_o[torch.from_numpy(idx).to(dev)].cpu().numpy(), ref.user_o.numpy(), f"{name}: user_o sample, records kernel, ka_er={ka_er}")
        model.ka_flash, model.ka_er = None, False
