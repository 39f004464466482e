"""CPU: the oracles and the host logic against fixtures produced by the REFERENCE'S OWN CODE
(tests/golden/ref/, written by tests/golden/make_ref_fixtures.py in the build container):

  * model__*.npz   -- model.py + aggregators.py executed unmodified over the numpy TF stand-in
                      (tests/refpin/tf1_standin.py): pins the WIRING of oracle/mirror_fp32.py,
                      oracle/equations_fp64.py and oracle/train_ref.py's loss; the arithmetic of the
                      fixture is numpy's, in fp32 and fp64
  * data_loader / load_rating / metrics / early_stop / ablations -- TF-free reference modules run as is
  * harness.npz    -- util.py / train.py loops run by the reference over that model

Nothing here reads /root/reference.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from mvin_amd import config, data_io, harness
from mvin_amd.config import make_args
from oracle import equations_fp64, mirror_fp32, prep_ref, train_ref

REF = os.path.join(os.path.dirname(__file__), "golden", "ref")
MODEL_FIX = sorted(glob.glob(os.path.join(REF, "model__*.npz")))


def load_model_fixture(path):
    z = np.load(path)
    kw = json.loads(str(z["args_json"]))
    args = make_args(**kw)
    params = {k[3:]: z[k] for k in z.files if k.startswith("p__")}
    P = max(1, args.p_hop)
    mem = [[z[f"memories_{x}_{i}"] for i in range(P)] for x in "hrt"]
    return z, args, params, mem


HOT_FIX = sorted(glob.glob(os.path.join(REF, "hot__*.npz")))


def load_hot_fixture(path):
    """A hot-shape fixture (tests/refpin/hot_cases.py): inputs regenerated from its seeds and checked against the checksum
    the generator stored -> (expected, args, case, params, user_triplet_set)."""
    from refpin import hot_cases
    name, ablation = os.path.basename(path)[len("hot__"):-4].split("__")
    exp = hot_cases.expected(path)
    args, case, params, uts = hot_cases.build(name, ablation)
    assert hot_cases.inputs_crc32(case, params) == exp.crc, "regenerated inputs differ from the ones the reference ran on"
    return exp, args, case, params, uts


def test_hot_shape_fixtures_present():
    from refpin import hot_cases
    assert len(HOT_FIX) == len(hot_cases.HOT_CASES) * len(hot_cases.HOT_ABLATIONS)


@pytest.mark.parametrize("path", HOT_FIX, ids=lambda p: os.path.basename(p)[5:-4])
def test_oracles_match_reference_graph_at_hot_shapes(path):
    """the reference's own model.py:259-324 / :161-240 at D 32 / 64, K 16 / 32 / 64 -- the shapes the packed,
    split, records and tail kernels serve -- against both oracles."""
    exp, args, case, params, _ = load_hot_fixture(path)
    feed = (case.users, case.items, case.memories_h, case.memories_r, case.memories_t)
    m = mirror_fp32.forward(args, params, case.adj_entity, case.adj_relation, *feed)
    np.testing.assert_allclose(m.scores.numpy(), exp.scores_32, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m.scores.numpy(), exp.scores_64, rtol=1e-5, atol=1e-6)
    e = equations_fp64.forward(args, params, case.adj_entity, case.adj_relation, *feed)
    np.testing.assert_allclose(e.scores, exp.scores_64, rtol=1e-9, atol=1e-11)


def test_reference_fixtures_present():
    assert len(MODEL_FIX) >= 30
    names = {os.path.basename(p)[len("model__"):-4].split("__")[1] for p in MODEL_FIX}
    raises = json.load(open(os.path.join(REF, "reference_raises.json")))
    # every --ablation preset of parameter_ablation.py either ran or is recorded as failing IN the reference
    assert names | {k.split("__")[1] for k in raises} == set(config.ABLATIONS)
    # wide_deep=False is the only thing the reference cannot run: model.py:366-374
    assert {k.split("__")[1] for k in raises} == {"no_wd", "no_wd_ho_only"}
    assert all("tuple" in v for v in raises.values())


@pytest.mark.parametrize("path", MODEL_FIX, ids=lambda p: os.path.basename(p)[7:-4])
def test_oracles_match_reference_graph(path):
    z, args, params, (mh, mr, mt) = load_model_fixture(path)
    m = mirror_fp32.forward(args, params, z["adj_entity"], z["adj_relation"], z["users"], z["items"], mh, mr, mt)
    e = equations_fp64.forward(args, params, z["adj_entity"], z["adj_relation"], z["users"], z["items"], mh, mr, mt)
    ref64, ref32 = z["ref_scores_64"], z["ref_scores_32"]
    # the from-the-equations fp64 restatement vs the reference graph evaluated in fp64: round-off only
    np.testing.assert_allclose(e.scores, ref64, rtol=1e-9, atol=1e-11)
    # the fp32 op-by-op mirror vs the reference graph in fp32 (same op order; BLAS summation order differs)
    np.testing.assert_allclose(m.scores.numpy(), ref32, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m.scores.numpy(), ref64, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m.scores_normalized.numpy(), z["ref_scores_normalized_64"], rtol=1e-5, atol=1e-6)
    if args.PS_only:
        return
    # id expansion (model.py:243-256): bit-exact
    for i, x in enumerate(m.entities):
        np.testing.assert_array_equal(x.numpy(), z[f"ref_entities_{i}"])
    for i, x in enumerate(m.relations):
        np.testing.assert_array_equal(x.numpy(), z[f"ref_relations_{i}"])
    # attention outputs (model.py:294,304,319-323)
    for i in range(2):
        key = f"ref_importance_{i}"
        if key in z.files:
            np.testing.assert_allclose(m.importance_list[i].numpy(), z[key], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(e.importance_list[i], z[key], rtol=1e-9, atol=1e-12)
        elif i < len(m.importance_list):
            assert m.importance_list[i] is None          # User_orient_rela = False
    if args.n_mix_hop * args.h_hop == 1:
        assert "ref_importance_1" not in z.files         # model.py:323: the int 0


@pytest.mark.parametrize("path", MODEL_FIX, ids=lambda p: os.path.basename(p)[7:-4])
def test_parameter_set_and_loss_match_reference_graph(path):
    z, args, params, (mh, mr, mt) = load_model_fixture(path)
    # a12: the reference graph created exactly the variables mvin_amd.params creates (shapes were
    # asserted by the stand-in's get_variable) ...
    used = set(z["ref_params_used"].tolist())
    assert used == set(params), (sorted(used ^ set(params)))
    assert len(z["ref_variables"]) == len(params)
    if not args.PS_only:
        from mvin_amd.params import aggregator_keys
        assert int(z["ref_n_aggregators"]) == len(aggregator_keys(args))
        # a7: "<classname lower>_<save_model_name>_<i>_<n>" (aggregators.py:20-23)
        want = [f"sumaggregator_urh_matrix_{args.save_model_name}_{i}_{n}" for (i, n) in aggregator_keys(args)]
        assert z["ref_aggregator_names"].tolist() == want
    # f-2: loss pieces of model.py:378-412
    p = {k: torch.tensor(v) for k, v in params.items()}
    loss, pieces, _ = train_ref.loss_from_params(args, p, z["adj_entity"], z["adj_relation"], z["users"], z["items"],
                                                 z["labels"], mh, mr, mt)
    ref = z["ref_loss_64"]                               # loss, base, l2, l2_agg
    np.testing.assert_allclose(float(loss), ref[0], rtol=2e-6)
    np.testing.assert_allclose(float(pieces["base"]), ref[1], rtol=2e-6)
    np.testing.assert_allclose(float(pieces["l2"]), ref[2], rtol=2e-6)
    np.testing.assert_allclose(float(pieces["l2agg"]), ref[3], rtol=2e-6)


def test_ablation_table_equals_reference():
    t = json.load(open(os.path.join(REF, "ablations.json")))
    assert tuple(t["switches"]) == config._SWITCHES
    for name, sw in t["table"].items():
        if name == "__unknown__":
            continue
        a = make_args(ablation=name)
        assert [bool(getattr(a, s)) for s in config._SWITCHES] == sw, name
    assert set(t["table"]) - {"__unknown__"} == set(config.ABLATIONS)
    # an unknown name leaves the switches as given, then coerces them (parameter_ablation.py:167-175)
    from types import SimpleNamespace
    a = config.parameter_env(SimpleNamespace(ablation="__unknown__", abla_exp=0, SW=1, User_orient=1, User_orient_rela=1,
                                             User_orient_kg_eh=1, PS_O_ft=1, wide_deep=1, PS_only=0, HO_only=0))
    assert [bool(getattr(a, s)) for s in config._SWITCHES] == t["table"]["__unknown__"]


def test_ranking_metrics_equal_reference():
    t = json.load(open(os.path.join(REF, "metrics.json")))
    for c in t["cases"]:
        r_hit = [1 if i in c["answers"] else 0 for i in c["ranked"][:100]]
        for k, out in c["out"].items():
            k = int(k)
            assert harness.precision_at_k(c["ranked"], c["answers"], k) == out["precision"]
            assert harness.recall_at_k(c["ranked"], c["answers"], k) == out["recall"]
            assert abs(harness.ndcg_at_k(r_hit, k) - out["ndcg"]) < 1e-12
            assert abs(harness.dcg_at_k(r_hit, k) - out["dcg"]) < 1e-12
    for g in t["graded"]:
        for k, out in g["out"].items():
            assert abs(harness.dcg_at_k(g["r"], int(k)) - out["dcg"]) < 1e-12
            assert abs(harness.ndcg_at_k(g["r"], int(k)) - out["ndcg"]) < 1e-12


def test_early_stop_decisions_equal_reference():
    class M:
        path = type("P", (), {"emb": "x"})()

        def __init__(self):
            self.saves = 0

        def save_pretrain_emb_fuc(self, sess, saver):
            self.saves += 1

    for s in json.load(open(os.path.join(REF, "early_stop.json"))):
        stop, model = harness.EarlyStop(s["tolerance"], s["early_stop"], s["save_final_model"]), M()
        for ep, (score, want, saves) in enumerate(zip(s["scores"], s["returns"], s["saves"])):
            got = stop.update(ep, score, model)
            assert got == (want == "EarlyStopping"), (s, ep)
            assert model.saves == saves
        assert len(s["returns"]) == len(s["scores"]) or s["returns"][-1] == "EarlyStopping"


# ------------------------------------------------------------------------------ data loader
def _edges_of(indptr, dst, rel, e):
    return list(zip(dst[indptr[e]:indptr[e + 1]].tolist(), rel[indptr[e]:indptr[e + 1]].tolist()))


def check_adjacency_rule(indptr, dst, rel, adj_e, adj_r, K):
    """contruct_random_adj (:375-388): zero row when absent; every slot an edge of the entity; deg >= K
    => K DISTINCT edge positions (as a multiset of (tail, relation) no edge used more often than it
    exists); deg < K => drawn with replacement (any multiset of its edges)."""
    from collections import Counter
    for e in range(len(indptr) - 1):
        edges = _edges_of(indptr, dst, rel, e)
        got = list(zip(adj_e[e].tolist(), adj_r[e].tolist()))
        if not edges:
            assert got == [(0, 0)] * K
            continue
        have, used = Counter(edges), Counter(got)
        assert set(used) <= set(have), e
        if len(edges) >= K:
            assert all(used[x] <= have[x] for x in used), e


def test_csr_and_adjacency_rules_against_reference():
    z = np.load(os.path.join(REF, "data_loader.npz"))
    n_entity, K = int(z["n_entity"]), int(z["K"])
    indptr, dst, rel = prep_ref.build_csr(z["kg_np"], n_entity)
    # construct_kg (:324-343): same per-entity neighbor lists, in the reference's insertion order
    np.testing.assert_array_equal(indptr, z["csr_indptr"])
    np.testing.assert_array_equal(dst, z["csr_dst"])
    np.testing.assert_array_equal(rel, z["csr_rel"])
    # the reference's own sample obeys the rule checker, and so does the oracle's (other draws, same rule)
    check_adjacency_rule(indptr, dst, rel, z["adj_entity"], z["adj_relation"], K)
    assert (np.diff(indptr) == 0).sum() >= 3 and (np.diff(indptr) >= K).any() and ((np.diff(indptr) > 0) & (np.diff(indptr) < K)).any()
    for seed in (1, 2):
        a_e, a_r = prep_ref.sample_adjacency(indptr, dst, rel, n_entity, K, seed)
        check_adjacency_rule(indptr, dst, rel, a_e, a_r, K)
    with pytest.raises(AssertionError):
        bad = z["adj_entity"].copy()
        bad[int(np.argmax(np.diff(indptr)))] += 1000
        check_adjacency_rule(indptr, dst, rel, bad, z["adj_relation"], K)


def check_ripple_rule(indptr, dst, rel, hist, uts, P, Nm, n_neighbor):
    """_get_user_triplet_set (:407-441): hop 0 heads come from the history, hop h heads from hop h-1's
    tails; every (h, r, t) is an edge of h; exactly n_memory entries; without replacement iff the hop has
    at least n_memory candidates; an empty hop copies the previous one."""
    from collections import Counter
    seeds = list(hist)
    for h in range(P):
        hh, rr, tt = (uts[h][i].tolist() for i in range(3))
        assert len(hh) == Nm
        cands = Counter()
        for s in seeds:
            ed = _edges_of(indptr, dst, rel, s)
            assert set(zip(tt, rr)) or True
            cands[s] += min(len(ed), n_neighbor)
        total = sum(cands.values())
        if total == 0:
            assert h > 0 and (uts[h] == uts[h - 1]).all()
        else:
            for a, r, t in zip(hh, rr, tt):
                assert a in cands and (t, r) in _edges_of(indptr, dst, rel, a)
            if total >= Nm:     # np.random.choice(replace=False): no head can appear more often than it has candidates
                assert all(c <= cands[a] for a, c in Counter(hh).items())
        seeds = tt


def test_ripple_set_rules_against_reference():
    z = np.load(os.path.join(REF, "data_loader.npz"))
    n_entity = int(z["n_entity"])
    indptr, dst, rel = prep_ref.build_csr(z["kg_np"], n_entity)
    P, Nm, nn = int(z["uts_p_hop"]), int(z["uts_n_memory"]), int(z["uts_n_neighbor"])
    ptr, items = z["uts_hist_ptr"], z["uts_hist_items"]
    for i, u in enumerate(z["uts_users"]):
        check_ripple_rule(indptr, dst, rel, items[ptr[i]:ptr[i + 1]].tolist(), z["uts"][i], P, Nm, nn)
    # the oracle's sampler on the same KG/history obeys the same checker
    mine = prep_ref.ripple_sets(indptr, dst, rel, ptr, items, len(z["uts_users"]), P, Nm, nn, seed=7)
    assert mine.shape == z["uts"].shape and mine.dtype == z["uts"].dtype
    for i in range(len(z["uts_users"])):
        check_ripple_rule(indptr, dst, rel, items[ptr[i]:ptr[i + 1]].tolist(), mine[i], P, Nm, nn)
    # layout: [hop][h|r|t][n_memory] (np.array(ret) of :440), two-entity graph: hop 1 walks back
    pg = z["uts_pair_graph"]
    assert pg.shape == (2, 3, 3) and (pg[0, 0] == 50).all() and (pg[0, 2] == 51).all() and (pg[1, 0] == 51).all()


def test_load_rating_equals_reference(tmp_path):
    z = np.load(os.path.join(REF, "load_rating.npz"))
    d = tmp_path / "data"
    d.mkdir()
    np.save(d / "ratings_final.npy", z["ratings"])
    for name in ("train", "eval", "test"):
        with open(d / f"{name}_pd.csv", "w") as f:
            f.write(",item,like,user\n")
            for i, (u, it, like) in enumerate(z[f"{name}_csv"]):
                f.write(f"{i},{it},{like},{u}\n")
    n_user, n_item, tr, ev, te, hist, pop = data_io.load_rating(str(d))
    assert (n_user, n_item) == (int(z["n_user"]), int(z["n_item"]))
    np.testing.assert_array_equal(tr, z["train"])
    np.testing.assert_array_equal(ev, z["eval"])
    np.testing.assert_array_equal(te, z["test"])
    assert sorted(pop) == z["pop"].tolist()
    assert sorted(hist) == z["hist_users"].tolist()
    for i, u in enumerate(z["hist_users"]):
        assert hist[int(u)] == z["hist_items"][z["hist_ptr"][i]:z["hist_ptr"][i + 1]].tolist()
    # tie rule of the popularity ranking (:55: stable sort of an insertion-ordered dict)
    ranked = z["pop_ranked_like_ref"].tolist()
    for k in (1, 5, 17, 40):
        assert data_io.most_popular_items(z["ratings"], k) == set(ranked[:k])


# ------------------------------------------------------------------------------ harness (f-3 / f-4)
def test_harness_matches_reference_loops():
    from mirror_model import MirrorModel, load_harness_fixture
    z, args, params, uts = load_harness_fixture()
    model = MirrorModel(args, params, z["adj_entity"], z["adj_relation"])
    # feed assembly (train.py:112-122)
    feed = harness.get_feed_dict(args, model, z["test_data"], uts, 3, 11)
    np.testing.assert_array_equal(feed[model.user_indices], z["feed_users"])
    np.testing.assert_array_equal(feed[model.item_indices], z["feed_items"])
    np.testing.assert_array_equal(feed[model.labels], z["feed_labels"])
    for i in range(2):
        np.testing.assert_array_equal(np.asarray(feed[model.memories_h[i]]), z[f"feed_h_{i}"])
        np.testing.assert_array_equal(np.asarray(feed[model.memories_r[i]]), z[f"feed_r_{i}"])
        np.testing.assert_array_equal(np.asarray(feed[model.memories_t[i]]), z[f"feed_t_{i}"])
    # CTR evaluation (util.py:44-56)
    for name in ("train", "eval", "test"):
        aucs, accs, f1s, auc, acc, f1 = harness.ctr_eval(args, model, z[f"{name}_data"], uts, args.batch_size)
        np.testing.assert_allclose(np.array([aucs, accs, f1s]), z[f"ctr_{name}_lists"], atol=1e-9)
        np.testing.assert_allclose([auc, acc, f1], z[f"ctr_{name}_means"], atol=1e-9)
    # top-K (util.py:14-41, :137-205)
    users, tr, ev, te, item_set, k_list = harness.topk_settings(z["train_data"], z["eval_data"], z["test_data"], int(z["n_item"]))
    assert list(users) == z["topk_user_list"].tolist() and k_list == z["topk_k_list"].tolist()
    assert len(item_set) == int(z["topk_item_set_size"])
    cand = set(z["topk_candidates"].tolist())
    for mode in ("eval", "test"):
        p, r, nd, _, _ = harness.topk_eval(args, uts, model, users, tr, ev, te, cand, k_list, args.batch_size, mode=mode)
        np.testing.assert_allclose(np.array([p, r, nd]), z[f"topk_{mode}"], atol=1e-9)


def test_case_study_dump_matches_reference_text(tmp_path):
    from mirror_model import MirrorModel, compare_case_study_text, load_harness_fixture
    z, args, params, uts = load_harness_fixture()
    model = MirrorModel(args, params, z["adj_entity"], z["adj_relation"])
    hist = {u: sorted(s) for u, s in harness.get_user_record(z["train_data"]).items()}
    path = tmp_path / "case.log"
    harness.ctr_eval_case_study(args, model, z["test_data"][:16], uts, hist, {"3": "Entity Three"}, {"0": "rel zero"},
                                z["case_user_list"].tolist(), set(z["topk_candidates"].tolist()), args.batch_size, str(path))
    n_att = compare_case_study_text(path.read_text(), str(z["case_study_text"]), atol=2e-6)
    assert n_att > 0
