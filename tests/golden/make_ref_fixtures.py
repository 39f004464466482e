"""Pin the oracles to the REFERENCE'S OWN CODE (build container only; run from the repo root):

    python tests/golden/make_ref_fixtures.py            # writes tests/golden/ref/*.npz|*.json
    python tests/golden/make_ref_fixtures.py --only-hot # only the hot-shape score fixtures (hot__*.npz)

This script imports modules from /root/reference/src/model/MVIN, runs them on seeded synthetic inputs
and stores inputs + the reference's outputs as small data fixtures.  Nothing of the reference's source
is copied or travels; /root/reference does not exist on the GPU box, the fixtures do.

Two kinds of pin:

 A. TF-free reference modules, imported and executed as they are (numpy / pandas / stdlib only):
      data_loader_user_set.py  construct_kg :324-343, contruct_random_adj :375-388,
                               _get_user_triplet_set :407-441, load_rating :33-110 (+ load_pre_data)
      metrics.py :3-148        every ranking metric
      train_util.py :20-61     Early_stop_info.update_score decisions
      parameter_ablation.py    parameter_env: the 8 switches of every --ablation name
    The samplers draw from the unseeded global generators; they are seeded here, their outputs are
    stored and checked for the RULES (membership, with/without replacement, copy-previous-hop), which
    is what oracle/prep_ref.py and the HIP samplers must reproduce -- the draws themselves cannot be.

 B. The hot path: model.py + aggregators.py (and util.py / train.py, their callers) executed
    UNMODIFIED over tests/refpin/tf1_standin.py, a numpy stand-in for the TensorFlow-1.x symbols they
    use.  This pins the reference's WIRING -- its own Python control flow builds the graph: hop / mix
    / level loops, concat orders, reshapes, which variable is used where, feed assembly, evaluation
    loops -- while the arithmetic of each op is numpy's, not TensorFlow's (run in fp32 and fp64).
    It removes the shared-misreading risk between oracle/mirror_fp32.py and oracle/equations_fp64.py;
    it does not certify TF's kernels.  `wide_deep = False` presets raise inside the reference
    (model.py:366-374 treats the aggregator's tuple as a tensor); the exception is recorded.

Environment shims (not stand-ins for reference code): `np.asfarray` was removed in numpy 2 and is
restored as `asarray(dtype=float)` for metrics.py:10.
"""
import importlib
import io
import json
import os
import random
import re
import shutil
import sys
import tempfile
from contextlib import redirect_stdout
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_DIR = "/root/reference/src/model/MVIN"
OUT = os.path.join(HERE, "ref")

from mvin_amd import synth  # noqa: E402
from mvin_amd.config import ABLATIONS, make_args  # noqa: E402
from mvin_amd.params import init_params  # noqa: E402
from refpin import tf1_standin as tf  # noqa: E402

if not hasattr(np, "asfarray"):
    np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)  # numpy-2 removal; metrics.py:10


def ref_import(name):
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    sys.modules["tensorflow"] = tf
    with redirect_stdout(io.StringIO()):
        return importlib.import_module(name)


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# =============================================================================== part A
def small_kg(n_entity, n_relation, n_triples, seed, absent=()):
    rng = np.random.default_rng(seed)
    ok = np.array([e for e in range(n_entity) if e not in set(absent)])
    kg = np.stack([rng.choice(ok, n_triples), rng.integers(0, n_relation, n_triples), rng.choice(ok, n_triples)], 1)
    return kg.astype(np.int64)


def fixture_data_loader():
    dl = ref_import("data_loader_user_set")
    d = {}
    # --- construct_kg + contruct_random_adj
    n_entity, K = 40, 4
    kg_np = small_kg(n_entity, 5, 90, seed=11, absent=(3, 17, 39))
    kg, enti, rela = quiet(dl.construct_kg, None, kg_np)
    d["kg_np"], d["n_entity"], d["K"] = kg_np, n_entity, K
    indptr = np.zeros(n_entity + 1, np.int64)
    dst, rel = [], []
    for e in range(n_entity):
        for (t, r) in kg.get(e, []):
            dst.append(t)
            rel.append(r)
        indptr[e + 1] = len(dst)
    d["csr_indptr"], d["csr_dst"], d["csr_rel"] = indptr, np.array(dst, np.int64), np.array(rel, np.int64)
    d["kg_max_entity"], d["kg_max_relation"] = int(enti), int(rela)
    np.random.seed(5)
    adj_e, adj_r = dl.contruct_random_adj(SimpleNamespace(neighbor_sample_size=K), kg, n_entity)
    d["adj_entity"], d["adj_relation"] = adj_e, adj_r
    # --- _get_user_triplet_set (the worker of get_user_triplet_set, :407-441), P=3 so that a hop
    #     can come up empty only if the KG is disconnected there; n_neighbor=2 exercises random.sample
    hist = {0: [1, 2, 5], 1: [8], 2: [4, 4, 9, 12, 20, 21, 22], 3: [30, 31]}
    dl.g_kg = kg
    rows = []
    for u, h in hist.items():
        random.seed(100 + u)
        np.random.seed(200 + u)
        user, ret, _ = dl._get_user_triplet_set(u, h, p_hop=3, n_memory=6, n_neighbor=2)
        rows.append(np.array(ret, dtype=np.int32))
    d["uts_users"] = np.array(list(hist.keys()))
    d["uts_hist_ptr"] = np.cumsum([0] + [len(h) for h in hist.values()])
    d["uts_hist_items"] = np.concatenate([np.array(h) for h in hist.values()])
    d["uts"] = np.stack(rows)                       # [n_user, P, 3, Nm]
    d["uts_p_hop"], d["uts_n_memory"], d["uts_n_neighbor"] = 3, 6, 2
    # an isolated-tail case: entity 50's only neighbor is 51 and vice versa, hop 0 from history [50]
    kg2 = {50: [(51, 0)], 51: [(50, 0)]}
    dl.g_kg = kg2
    random.seed(1)
    np.random.seed(1)
    _, ret2, _ = dl._get_user_triplet_set(0, [50], p_hop=2, n_memory=3, n_neighbor=16)
    d["uts_pair_graph"] = np.array(ret2, dtype=np.int32)
    del dl.g_kg
    np.savez_compressed(os.path.join(OUT, "data_loader.npz"), **d)

    # --- load_rating on a small on-disk dataset (temp dir; the file CONTENTS go into the fixture)
    tmp = tempfile.mkdtemp()
    try:
        data_dir, misc = os.path.join(tmp, "data") + "/", os.path.join(tmp, "misc") + "/"
        os.makedirs(data_dir)
        os.makedirs(misc)
        rng = np.random.default_rng(21)
        n = 400
        ratings = np.stack([rng.integers(0, 25, n), rng.integers(0, 60, n), rng.integers(0, 2, n)], 1).astype(np.int64)
        np.save(data_dir + "ratings_final.npy", ratings)
        perm = rng.permutation(n)
        splits = {"train": ratings[perm[:240]], "eval": ratings[perm[240:320]], "test": ratings[perm[320:]]}
        # users 23 and 24 get no positive train row -> dropped from every split (:90-96)
        tr = splits["train"]
        tr[np.isin(tr[:, 0], (23, 24)), 2] = 0
        for name, arr in splits.items():
            with open(data_dir + f"{name}_pd.csv", "w") as f:      # column order of data/*/eval_pd.csv:1
                f.write(",item,like,user\n")
                for i, (u, it, like) in enumerate(arr):
                    f.write(f"{i},{it},{like},{u}\n")
        args = SimpleNamespace(path=SimpleNamespace(data=data_dir, misc=misc), dataset="MovieLens-1M",
                               new_load_data=False)
        n_user, n_item, trd, evd, ted, hist_d, pop = quiet(dl.load_rating, args)
        r = {"ratings": ratings, "train_csv": splits["train"], "eval_csv": splits["eval"], "test_csv": splits["test"],
             "n_user": int(n_user), "n_item": int(n_item), "train": trd, "eval": evd, "test": ted,
             "pop": np.array(sorted(pop)), "hist_users": np.array(sorted(hist_d)),
             "hist_ptr": np.cumsum([0] + [len(hist_d[u]) for u in sorted(hist_d)]),
             "hist_items": np.concatenate([np.array(hist_d[u]) for u in sorted(hist_d)])}
        # popularity with ties and a cut below the number of items: the inline code of :47-60 on top_k=500
        # keeps everything here (60 items), so the tie rule is pinned through a second call with fewer
        # distinct items than... (not reachable: top_k is hard-coded) -> store the full ranked order instead
        item_count = {}
        for i in range(ratings.shape[0]):
            item_count[ratings[i, 1]] = item_count.get(ratings[i, 1], 0) + 1
        r["pop_ranked_like_ref"] = np.array([k for k, _ in sorted(item_count.items(), key=lambda x: x[1], reverse=True)])
        np.savez_compressed(os.path.join(OUT, "load_rating.npz"), **r)
    finally:
        shutil.rmtree(tmp)


def fixture_metrics():
    m = ref_import("metrics")
    rng = np.random.default_rng(31)
    cases = []
    for c in range(40):
        n = int(rng.integers(1, 60))
        ranked = rng.permutation(80)[:n].tolist()
        answers = rng.permutation(80)[: int(rng.integers(1, 12))].tolist()
        r_hit = [1 if i in answers else 0 for i in ranked[:100]]
        rec = {"ranked": ranked, "answers": answers, "out": {}}
        for k in (1, 2, 5, 10, 25, 50, 100):
            rec["out"][str(k)] = {
                "precision": m.precision_at_k(ranked, answers, k), "recall": m.recall_at_k(ranked, answers, k),
                "ndcg": float(m.ndcg_at_k(r_hit, k)), "dcg": float(m.dcg_at_k(r_hit, k)),
                "hit_ratio": m.hit_ratio_at_k(ranked, answers, k), "mrr": m.mrr_at_k(ranked, answers, k),
                "map": m.map_at_k(ranked, answers, min(k, len(ranked)))}
        cases.append(rec)
    # graded relevance for dcg/ndcg
    graded = [[3, 2, 3, 0, 0, 1, 2, 2, 3, 0], [0, 0, 0], [1], [0, 1, 0, 1, 1, 1]]
    g = [{"r": r, "out": {str(k): {"dcg": float(m.dcg_at_k(r, k)), "ndcg": float(m.ndcg_at_k(r, k)),
                                   "dcg_m0": float(m.dcg_at_k(r, k, 0))} for k in (1, 3, 5, 10)}} for r in graded]
    with open(os.path.join(OUT, "metrics.json"), "w") as f:
        json.dump({"cases": cases, "graded": g}, f)


def fixture_early_stop():
    tu = ref_import("train_util")

    class M:
        def __init__(self):
            self.saves = 0

        def save_pretrain_emb_fuc(self, sess, saver):
            self.saves += 1

    seqs = []
    rng = np.random.default_rng(41)
    for tol, es, save in ((2, 3, True), (0, 1, True), (3, 2, False), (2, 3, True), (1, 4, True)):
        scores = np.round(rng.random(14), 3).tolist()
        if len(seqs) == 3:
            scores = sorted(scores)                                  # always improving: never stops
        args = SimpleNamespace(early_decrease_lr=2, early_stop=es, tolerance=tol, save_final_model=save)
        info, model = tu.Early_stop_info(args, False), M()
        ret, saves = [], []
        for ep, s in enumerate(scores):
            ret.append(quiet(info.update_score, ep, s, None, model, None))
            saves.append(model.saves)
            if ret[-1] == "EarlyStopping":
                break
        seqs.append({"tolerance": tol, "early_stop": es, "save_final_model": save, "scores": scores,
                     "returns": ret, "saves": saves})
    with open(os.path.join(OUT, "early_stop.json"), "w") as f:
        json.dump(seqs, f)


def fixture_ablations():
    pa = ref_import("parameter_ablation")
    src = open(os.path.join(REF_DIR, "parameter_ablation.py")).read()
    names = sorted(set(re.findall(r"args\.ablation\s*==\s*['\"]([A-Za-z0-9_]+)['\"]", src)))
    sw = ("SW", "User_orient", "User_orient_rela", "User_orient_kg_eh", "PS_O_ft", "wide_deep", "PS_only", "HO_only")
    table = {}
    for name in names + ["__unknown__"]:
        a = SimpleNamespace(ablation=name, abla_exp=0, **{s: 1 for s in sw})
        a.PS_only = a.HO_only = 0
        quiet(pa.parameter_env, a)
        table[name] = [bool(getattr(a, s)) for s in sw]
    with open(os.path.join(OUT, "ablations.json"), "w") as f:
        json.dump({"switches": sw, "table": table}, f, indent=0)
    return names


# =============================================================================== part B
_VAR_RULES = [
    (r"^(user|entity|relation|relation_emb_KGE)(_emb)?_matrix_STWS/(.+)_STWS$", lambda m: m.group(3)),
    (r"^enti_mlp_matrix(\d+)/transfer_matrix\1$", lambda m: f"enti_transfer_matrix_{m.group(1)}"),
    (r"^enti_mlp_matrix(\d+)/transfer_bias\1$", lambda m: f"enti_transfer_bias_{m.group(1)}"),
    (r"^user_mlp_matrix/(user_mlp_matrix|user_mlp_bias)$", lambda m: m.group(1)),
    (r"^transfer_agg_matrix(\d+)/transfer_agg_matrix\1$", lambda m: f"transfer_matrix_{m.group(1)}"),
    (r"^transfer_agg_matrix(\d+)/transfer_agg_bias\1$", lambda m: f"transfer_bias_{m.group(1)}"),
    (r"^h_emb_item_mlp_matrix/(h_emb_item_mlp_matrix|h_emb_item_mlp_bias)$", lambda m: m.group(1)),
    # aggregators.py:20-23,83-93: "<class>_<save_model_name>_<name>" + _wights/_bias/_urh_wights/_urh_bias
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_(\d+)_wights/weights$", lambda m: f"agg_{m.group(1)}_{m.group(2)}_weights"),
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_(\d+)_bias/bias$", lambda m: f"agg_{m.group(1)}_{m.group(2)}_bias"),
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_(\d+)_urh_wights/weights$", lambda m: f"agg_{m.group(1)}_{m.group(2)}_urh_weights"),
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_(\d+)_urh_bias/bias$", lambda m: f"agg_{m.group(1)}_{m.group(2)}_urh_bias"),
    # legacy aggregate (model.py:359): name = i (no mix index)
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_wights/weights$", lambda m: f"agg_{m.group(1)}_0_weights"),
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_bias/bias$", lambda m: f"agg_{m.group(1)}_0_bias"),
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_urh_wights/weights$", lambda m: f"agg_{m.group(1)}_0_urh_weights"),
    (r"^sumaggregator_urh_matrix_[^/]*?_(\d+)_urh_bias/bias$", lambda m: f"agg_{m.group(1)}_0_urh_bias"),
]


def provider_for(params, used):
    def provide(full, shape):
        for pat, key in _VAR_RULES:
            m = re.match(pat, full)
            if m:
                k = key(m)
                if k == "relation_emb_KGE_matrix" or k in params:
                    used.add(k)
                    return params[k]
        raise KeyError(f"reference variable {full} {shape} has no counterpart in mvin_amd.params")
    return provide


def build_ref_model(args, case, params, float_dtype):
    model_mod = ref_import("model")
    used = set()
    tf.reset(args.batch_size, provider_for(params, used), float_dtype)
    model = quiet(model_mod.MVIN, args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation)
    return model, used


def feed_of(model, case, labels):
    feed = {model.user_indices: case.users, model.item_indices: case.items, model.labels: labels}
    for i in range(len(case.memories_h)):
        feed[model.memories_h[i]] = case.memories_h[i]
        feed[model.memories_r[i]] = case.memories_r[i]
        feed[model.memories_t[i]] = case.memories_t[i]
    return feed


MODEL_SHAPES = {
    "d8k3h2m1p2": dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=4),
    "d8k3h2m2p1": dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=2, p_hop=1, n_memory=4, batch_size=4),
    "d16k8h1m1p1": dict(dim=16, neighbor_sample_size=8, h_hop=1, n_mix_hop=1, p_hop=1, n_memory=8, batch_size=5),
    "d8k2h3m1p2": dict(dim=8, neighbor_sample_size=2, h_hop=3, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=3),
    "d12k5h1m2p2": dict(dim=12, neighbor_sample_size=5, h_hop=1, n_mix_hop=2, p_hop=2, n_memory=6, batch_size=4),
    "d8k4h3m2p1": dict(dim=8, neighbor_sample_size=2, h_hop=3, n_mix_hop=2, p_hop=1, n_memory=4, batch_size=2),
}
ARG_KEYS = ("dim", "neighbor_sample_size", "h_hop", "n_mix_hop", "p_hop", "n_memory", "batch_size", "ablation",
            "l2_weight", "l2_agg_weight")


def fixture_model(shape_name, ablation, seed):
    kw = dict(MODEL_SHAPES[shape_name], ablation=ablation, l2_weight=1e-2, l2_agg_weight=1e-3)
    args = make_args(**kw)
    case = synth.small_case(args, n_user=8, n_entity=64, n_relation=5, seed=seed, zero_rows=3)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=seed + 100, random_agg_bias=True)
    labels = (np.random.default_rng(seed).random(args.batch_size) < 0.5).astype(np.float32)
    d = {"args_json": np.array(json.dumps({k: kw[k] for k in ARG_KEYS})),
         "n_user": case.n_user, "n_entity": case.n_entity, "n_relation": case.n_relation,
         "adj_entity": case.adj_entity, "adj_relation": case.adj_relation,
         "users": case.users, "items": case.items, "labels": labels}
    for i in range(len(case.memories_h)):
        d[f"memories_h_{i}"], d[f"memories_r_{i}"], d[f"memories_t_{i}"] = \
            case.memories_h[i], case.memories_r[i], case.memories_t[i]
    for k, v in params.items():
        d["p__" + k] = v
    for tag, fdt in (("32", np.float32), ("64", np.float64)):
        try:
            model, used = build_ref_model(args, case, params, fdt)
        except Exception as e:  # noqa: BLE001  (wide_deep=False: the reference itself fails)
            return None, f"{type(e).__name__}: {e}"
        sess = tf.Session()
        feed = feed_of(model, case, labels)
        items, sn = model.get_scores(sess, feed)
        assert np.array_equal(items, case.items)
        scores, loss, base, l2, l2a = sess.run([model.scores, model.loss, model.base_loss, model.l2_loss,
                                                model.l2_agg_loss], feed)
        d["ref_scores_" + tag], d["ref_scores_normalized_" + tag] = scores, sn
        d["ref_loss_" + tag] = np.array([loss, base, l2, l2a])
        if tag == "64":
            if not args.PS_only:
                u, lab, it, ents, rels, imp0, imp1 = model.eval_case_study(sess, feed)
                for i, x in enumerate(ents):
                    d[f"ref_entities_{i}"] = x
                for i, x in enumerate(rels):
                    d[f"ref_relations_{i}"] = x
                if imp0 is not None:
                    d["ref_importance_0"] = imp0
                if imp1 is not None and not isinstance(imp1, int):
                    d["ref_importance_1"] = imp1
                d["ref_n_aggregators"] = len(model.aggregators)
                d["ref_aggregator_names"] = np.array([a.name for a in model.aggregators])
            d["ref_variables"] = np.array(sorted(v.name for v in tf.global_variables()))
            d["ref_params_used"] = np.array(sorted(used))
    path = os.path.join(OUT, f"model__{shape_name}__{ablation}.npz")
    np.savez_compressed(path, **d)
    return path, None


def fixture_hot(name, ablation):
    """The reference's model.py / aggregators.py over the stand-in AT A HOT-KERNEL SHAPE (tests/refpin/hot_cases.py:
    D 32 / 64, K 16 / 32 / 64, 2 048 level-1 nodes per batch, adjacencies with and without repeated slots).  Stores
    the reference's scores (fp32 and fp64 arithmetic) and a checksum of the inputs, which the tests regenerate."""
    from refpin import hot_cases
    args, case, params, _ = hot_cases.build(name, ablation)
    labels = np.zeros(args.batch_size, np.float32)
    d = {"args_json": np.array(json.dumps(dict(hot_cases.HOT_CASES[name]["shape"], ablation=ablation))),
         "inputs_crc32": np.array(hot_cases.inputs_crc32(case, params), dtype=np.int64)}
    for tag, fdt in (("32", np.float32), ("64", np.float64)):
        model, _ = build_ref_model(args, case, params, fdt)
        sess = tf.Session()
        feed = feed_of(model, case, labels)
        items, sn = model.get_scores(sess, feed)
        assert np.array_equal(items, case.items)
        d["ref_scores_" + tag] = sess.run(model.scores, feed)
        d["ref_scores_normalized_" + tag] = sn
    path = os.path.join(OUT, f"hot__{name}__{ablation}.npz")
    np.savez_compressed(path, **d)
    return path


def fixture_all_hot():
    from refpin import hot_cases
    for name in hot_cases.HOT_CASES:
        for abl in hot_cases.HOT_ABLATIONS:
            path = fixture_hot(name, abl)
            print(f"{os.path.basename(path):52s} {os.path.getsize(path)} B")


def fixture_harness():
    """util.py's evaluation loops + train.py's feed assembly, run by the reference over its own
    (stand-in-backed) model on a small synthetic dataset; outputs only."""
    util = ref_import("util")
    train_mod = ref_import("train")
    tmp = tempfile.mkdtemp()
    try:
        kw = dict(dim=8, neighbor_sample_size=3, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=8)
        args = make_args(**kw)
        n_user, n_item, n_entity, n_relation = 14, 30, 64, 5
        rng = np.random.default_rng(51)
        adj_e, adj_r = synth.uniform_adjacency(n_entity, n_relation, 3, seed=52)
        uts = synth.ripple_sets(n_user, n_entity, n_relation, 2, 4, seed=53)
        params = init_params(args, n_user, n_entity, n_relation, seed=54, random_agg_bias=True)
        n = 330
        data = np.stack([rng.integers(0, n_user, n), rng.integers(0, n_item, n), rng.integers(0, 2, n)], 1).astype(np.int64)
        train_data, eval_data, test_data = data[:200], data[200:265], data[265:]
        case = SimpleNamespace(n_user=n_user, n_entity=n_entity, n_relation=n_relation, adj_entity=adj_e, adj_relation=adj_r)
        args.path = SimpleNamespace(misc=tmp + "/", case_st=tmp + "/")
        args.save_record_user_list, args.log_name, args.epoch, args.SW_stage = False, "fx", 0, 4
        d = {"args_json": np.array(json.dumps(kw)), "n_user": n_user, "n_item": n_item, "n_entity": n_entity,
             "n_relation": n_relation, "adj_entity": adj_e, "adj_relation": adj_r, "uts": uts,
             "train_data": train_data, "eval_data": eval_data, "test_data": test_data}
        for k, v in params.items():
            d["p__" + k] = v
        model, _ = build_ref_model(args, case, params, np.float64)
        sess = tf.Session()
        # feed assembly (train.py:112-122) -> plain arrays
        feed = train_mod.get_feed_dict(args, None, model, test_data, uts, 3, 11)
        d["feed_users"], d["feed_items"], d["feed_labels"] = (np.asarray(feed[model.user_indices]),
                                                              np.asarray(feed[model.item_indices]), np.asarray(feed[model.labels]))
        for i in range(2):
            d[f"feed_h_{i}"], d[f"feed_r_{i}"], d[f"feed_t_{i}"] = (np.asarray(feed[model.memories_h[i]]),
                                                                    np.asarray(feed[model.memories_r[i]]), np.asarray(feed[model.memories_t[i]]))
        # CTR evaluation (util.py:44-56)
        for name, dd in (("train", train_data), ("eval", eval_data), ("test", test_data)):
            aucs, accs, f1s, auc, acc, f1 = quiet(util.ctr_eval, args, None, sess, model, dd, uts, args.batch_size)
            d[f"ctr_{name}_lists"] = np.array([aucs, accs, f1s])
            d[f"ctr_{name}_means"] = np.array([auc, acc, f1])
        # top-K settings + evaluation (util.py:14-41, :137-205); candidate set = most popular items, as
        # train.py:69-77 passes item_set_most_pop
        np.random.seed(0)
        user_list, tr, ev, te, item_set, k_list = quiet(util.topk_settings, args, True, train_data, eval_data, test_data,
                                                        n_item, False, "fx")
        d["topk_user_list"], d["topk_k_list"] = np.array(user_list), np.array(k_list)
        d["topk_item_set_size"] = len(item_set)
        cnt = np.bincount(data[:, 1], minlength=n_item)
        pop = set(np.argsort(-cnt, kind="stable")[:20].tolist())
        d["topk_candidates"] = np.array(sorted(pop))
        for mode in ("eval", "test"):
            p, r, nd, _, _ = quiet(util.topk_eval, sess, args, uts, model, user_list, tr, ev, te, pop, k_list,
                                   args.batch_size, mode=mode)
            d[f"topk_{mode}"] = np.array([p, r, nd])
        # case-study dump (util.py:59-127) -> its text
        hist = util.get_user_record(train_data, True)
        quiet(util.ctr_eval_case_study, args, None, sess, model, test_data[:16], uts, {u: sorted(s) for u, s in hist.items()},
              {"3": "Entity Three"}, {"0": "rel zero"}, user_list, pop, args.batch_size)
        with open(f"{tmp}/fx_ep_0_st_4.log") as f:
            d["case_study_text"] = np.array(f.read())
        d["case_user_list"] = np.array(user_list)
        np.savez_compressed(os.path.join(OUT, "harness.npz"), **d)
    finally:
        shutil.rmtree(tmp)


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-hot" in sys.argv:
        fixture_all_hot()
        return
    fixture_data_loader()
    fixture_metrics()
    fixture_early_stop()
    names = fixture_ablations()
    assert set(names) == set(ABLATIONS), (sorted(names), sorted(ABLATIONS))
    unrunnable = {}
    seed = 2000
    plan = [("d8k3h2m1p2", names), ("d8k3h2m2p1", ["all", "no_uor", "no_uo", "ho_only", "no_kg_eh_uo"]),
            ("d16k8h1m1p1", ["all", "no_kg_eh_uo", "no_uor", "ps_only"]), ("d8k2h3m1p2", ["all", "no_uo", "no_wd"]),
            ("d12k5h1m2p2", ["all", "no_ps_o_ft", "ho_only_uo_kg_eh"]), ("d8k4h3m2p1", ["all", "no_uor_and_no_kg_eh_uo"])]
    for shape, abls in plan:
        for abl in abls:
            seed += 1
            path, err = fixture_model(shape, abl, seed)
            if err:
                unrunnable[f"{shape}__{abl}"] = err
                print(f"{shape}__{abl}: reference raises -> {err[:90]}")
            else:
                print(f"{os.path.basename(path):52s} {os.path.getsize(path)} B")
    with open(os.path.join(OUT, "reference_raises.json"), "w") as f:
        json.dump(unrunnable, f, indent=1)
    fixture_harness()
    fixture_all_hot()
    print("done ->", OUT)


if __name__ == "__main__":
    main()
