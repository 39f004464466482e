"""CPU: host-side logic -- flags/ablations, parameter set, synthetic inputs, id layout."""
import numpy as np
import pytest

from mvin_amd import synth
from mvin_amd.config import ABLATIONS, make_args, tree_depth
from mvin_amd.ops import ent_level_offsets
from mvin_amd.params import aggregator_keys, init_params, xavier_uniform


def test_parser_defaults_match_reference():
    a = make_args()
    # parser.py:8-57 defaults
    assert (a.dim, a.neighbor_sample_size, a.h_hop, a.n_mix_hop, a.p_hop, a.n_memory, a.batch_size) == \
        (8, 8, 3, 2, 1, 16, 512)
    assert a.User_orient is True and a.wide_deep is True and a.PS_only is False
    assert tree_depth(a) == 6


def test_ablation_presets_match_reference_table():
    # parameter_ablation.py:4-165 (SW, UO, UOR, UO_kg_eh, PS_O_ft, wide_deep, PS_only, HO_only)
    assert ABLATIONS["all"] == (1, 1, 1, 1, 1, 1, 0, 0)
    assert ABLATIONS["ho_only"] == (1, 1, 1, 0, 1, 1, 0, 1)
    assert ABLATIONS["no_wd_ho_only"] == (1, 1, 1, 0, 1, 0, 0, 1)
    assert ABLATIONS["no_uor_and_no_kg_eh_uo"] == (1, 1, 0, 0, 1, 1, 0, 0)
    assert len(ABLATIONS) == 18
    a = make_args(ablation="ps_only")
    assert a.PS_only is True and a.HO_only is False
    a = make_args(ablation="all", User_orient_rela=0)  # explicit switch wins
    assert a.User_orient_rela is False


def test_parameter_shapes_and_init():
    a = make_args(dim=16, h_hop=2, n_mix_hop=2, p_hop=2)
    p = init_params(a, 7, 50, 5, seed=0)
    assert p["relation_emb_KGE_matrix"].shape == (5, 16, 16)
    assert p["enti_transfer_matrix_1"].shape == (48, 16)
    assert p["user_mlp_matrix"].shape == (48, 16)          # PS_O_ft: (P+1)*D rows (model.py:101)
    assert "transfer_matrix_4" in p and "transfer_matrix_5" not in p
    assert aggregator_keys(a) == [(0, 0), (1, 0), (0, 1), (1, 1)]
    assert p["agg_1_1_urh_weights"].shape == (48, 1) and not p["agg_0_0_bias"].any()
    a2 = make_args(dim=16, h_hop=2, n_mix_hop=2, p_hop=2, ablation="no_ps_o_ft")
    assert init_params(a2, 7, 50, 5)["user_mlp_matrix"].shape == (32, 16)
    assert aggregator_keys(make_args(ablation="no_wd", h_hop=3)) == [(0, 0), (1, 0), (2, 0)]
    # xavier limit sqrt(6/(fan_in+fan_out)); 3-D fans as TF computes them
    w = xavier_uniform(np.random.default_rng(0), (5, 16, 16))
    assert np.abs(w).max() <= np.sqrt(6.0 / (16 * 5 + 16 * 5)) + 1e-7
    with pytest.raises(ValueError):
        init_params(make_args(p_hop=0, ablation="no_ps_o_ft"), 4, 10, 3)


def test_level_offsets():
    ent, rel = ent_level_offsets(B=3, K=4, levels=2)
    assert ent == [(0, 3), (3, 12), (15, 48)]
    assert rel == [(0, 12), (12, 48)]


def test_sampler_rule_matches_reference():
    # data_loader_user_set.py:375-388
    kg = synth.synth_kg(500, 5, 6.0, seed=3)
    indptr, dst, rel = synth.kg_to_csr(kg, 500)
    K = 4
    ae, ar = synth.sample_adjacency(indptr, dst, rel, K, seed=4)
    deg = np.diff(indptr)
    assert ae.shape == (500, K) and ae.dtype == np.int64
    for x in range(500):
        edges = list(zip(dst[indptr[x]:indptr[x + 1]], rel[indptr[x]:indptr[x + 1]]))
        got = list(zip(ae[x], ar[x]))
        if deg[x] == 0:
            assert got == [(0, 0)] * K          # absent entity keeps the zero row
        elif deg[x] >= K:
            # without replacement: a sub-multiset of the edge list
            pool = list(edges)
            for g in got:
                assert g in pool
                pool.remove(g)
        else:
            assert set(got) <= set(edges)       # with replacement
    # undirected: every triple appears under head and tail (construct_kg :324-343)
    assert deg.sum() == 2 * len(kg)


def test_dataset_case_shapes():
    c = synth.dataset_case("amazon-book_20core", K=8, B=64, seed=1)
    assert c.adj_entity.shape == (113487, 8) and c.n_relation == 39
    assert len(c.memories_h) == 1 and c.memories_h[0].shape == (64, 16) and c.memories_h[0].dtype == np.int32
    assert c.items.max() < 24915 and c.users.max() < 70585
    mh, mr, mt = synth.memories_for(c.user_triplet_set, c.users[:5])
    np.testing.assert_array_equal(mh[0], c.user_triplet_set[c.users[:5], 0, 0])
    np.testing.assert_array_equal(mt[0], c.user_triplet_set[c.users[:5], 0, 2])


def test_bench_algorithmic_bytes_match_survey():
    import bench
    assert bench.algorithmic_bytes_per_pair(16, 8, 1) == 708
    assert bench.algorithmic_bytes_per_pair(32, 16, 2) == 37252
    assert bench.algorithmic_bytes_per_pair(64, 32, 2) == 279300
    assert bench.algorithmic_bytes_per_pair(64, 64, 2) == 1098756
    assert bench.algorithmic_bytes_per_pair(128, 128, 3, s=2) == 558007812


def test_algorithmic_bytes_per_pair_matches_survey_table():
    """bench.py's roofline numerator is SURVEY.md 8(d)'s figure: T*D*s + (T-K^L)*K*8 + D*s + 4."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.algorithmic_bytes_per_pair
    assert f(16, 8, 1) == 708                       # C1
    assert f(32, 16, 2) == 37252                    # C2
    assert f(64, 32, 2) == 279300                   # C3 (metric config)
    assert f(64, 64, 2) == 1098756                  # C4
    assert f(128, 128, 3, s=2) == 558007812         # C5 (bf16 table)
    assert bench.HBM_PEAK_GBS == 8000.0


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must START two ranks (the flag was parsed
    and never read).  Plumbing only: --launch-check rendezvouses over gloo, counts the ranks with an all-reduce and exits
    before any GPU work."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MVIN_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    lc = json.loads(lines[0])["launch_check"]
    assert lc == {"gpus_asked": 2, "ranks_launched": 2, "world_size": 2, "backend": "gloo", "launcher": "self"}
    # a launcher that started a different number of ranks than --gpus says is an error, not a mislabelled line
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.parametrize("name", ["MovieLens-1M", "last-fm_50core", "amazon-book_20core"])
def test_synthetic_kg_reproduces_both_notebook_statistics(name):
    """The synthetic KGs are calibrated on the notebook's TWO degree statistics: the mean degree and the
    share of entities with >= 20 neighbours (ML-1M 17.0 %, last-fm 2.5 %, amazon-book 95.9 %) -- within 3 points, for the
    seeds bench.py uses.  The duplicate-slot gain of the fused kernels depends on the whole distribution."""
    from mvin_amd import synth
    d = synth.DATASETS[name]
    for seed in (1, 2):
        kg = synth.synth_kg(d["n_entity"], d["n_relation"], d["mean_degree"], seed=seed,
                            tail_exponent=d["tail_exponent"], head_sigma=d["head_sigma"])
        deg = np.bincount(np.concatenate([kg[:, 0], kg[:, 2]]), minlength=d["n_entity"])
        assert abs(deg.mean() - d["mean_degree"]) < 0.01
        assert abs((deg >= 20).mean() - d["share_ge20"]) < 0.03, (name, seed, (deg >= 20).mean())
        assert kg[:, 1].min() >= 0 and kg[:, 1].max() < d["n_relation"]


def test_projected_tables_identity_in_float64():
    """everything the two deepest levels apply to a gathered row before the ReLU is linear, so the
    matrices can be applied to the TABLE (model.py:270-283, aggregators.py:108-116 re-associated).  Checked here in float64,
    independent of any kernel: per-child self1 and out1, and the per-parent sums, in both forms."""
    rng = np.random.default_rng(11)
    nE, D, K, B = 200, 16, 8, 5
    E, W1, W2, A0 = (rng.normal(size=s) for s in ((nE, D), (D, D), (D, D), (D, D)))
    b1, b2, a0 = (rng.normal(size=D) for _ in range(3))
    q = rng.normal(size=(B, D))
    x1 = rng.integers(0, nE, size=(B, K))                       # children of the pair's item
    y = rng.integers(0, nE, size=(B, K, K))                     # grandchildren
    w = rng.random(size=(B, K, K))
    w /= w.sum(-1, keepdims=True) * K                           # attention weights p_k / K: c = sum_k w_k = 1 / K
    c = 1.0 / K
    # the reference's order: project every gathered row, then the weighted sum, then the aggregator
    self1 = (E[x1] + q[:, None, :]) @ W1 + b1
    neigh = ((E[y] + q[:, None, None, :]) @ W2 + b2) * w[..., None]
    out1 = np.maximum((self1 + neigh.sum(2)) @ A0 + a0, 0.0)
    # the projected-tables form: T1 = E W1, TA1 = E W1 A0, TA2 = E W2 A0 per entity; u1, v per pair
    T1, TA1, TA2 = E @ W1, E @ W1 @ A0, E @ W2 @ A0
    u1 = q @ W1 + b1
    v = q @ ((W1 + c * W2) @ A0) + (b1 + c * b2) @ A0 + a0
    self1_p = T1[x1] + u1[:, None, :]
    out1_p = np.maximum(TA1[x1] + (TA2[y] * w[..., None]).sum(2) + v[:, None, :], 0.0)
    np.testing.assert_allclose(self1_p, self1, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out1_p, out1, rtol=1e-11, atol=1e-11)
