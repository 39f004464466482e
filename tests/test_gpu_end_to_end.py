"""-m gpu: KG -> GPU-built inputs -> training epochs -> evaluation; the model must learn the
synthetic signal (loss falls, eval AUC well above chance)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_train_and_eval_on_synthetic_signal(hip_lib, monkeypatch):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import run_synthetic
    monkeypatch.setattr(sys, "argv", ["run_synthetic.py", "--epochs", "5"])
    out = run_synthetic.main()
    assert out["loss"][-1] < out["loss"][0] - 0.05          # it trains
    assert max(out["auc"]) > 0.54                           # and generalises to held-out pairs (chance = 0.5)


def test_from_reference_file_formats(hip_lib, tmp_path):
    """data/<dataset>/ in the reference's on-disk formats -> load_data (GPU samplers) -> MVIN ->
    CTR / top-K evaluation through the harness."""
    import numpy as np
    import torch
    from mvin_amd import data_io, harness
    from mvin_amd.config import make_args
    from mvin_amd.model import MVIN

    rng = np.random.default_rng(0)
    n_user, n_item, n_ent, n_rel = 30, 20, 80, 4
    ratings = np.stack([rng.integers(0, n_user, 900), rng.integers(0, n_item, 900), rng.integers(0, 2, 900)], 1)
    heads = np.concatenate([np.arange(n_ent), rng.integers(0, n_ent, 400)])       # every entity appears
    kg = np.stack([heads, rng.integers(0, n_rel, heads.size), rng.integers(0, n_ent, heads.size)], 1)
    np.savetxt(tmp_path / "ratings_final.txt", ratings, fmt="%d")
    np.savetxt(tmp_path / "kg_final.txt", kg, fmt="%d")
    parts = np.split(ratings[rng.permutation(900)], [540, 720])
    for name, part in zip(("train", "eval", "test"), parts):
        with open(tmp_path / f"{name}_pd.csv", "w") as f:
            f.write(",item,like,user\n")
            for i, (u, it, l) in enumerate(part):
                f.write(f"{i},{it},{l},{u}\n")
    K, P, Nm = 4, 2, 8
    (nu, ni, ne, nr, train, ev, test, adj_e, adj_r, uts, _, _, pop, hist, _, _) = data_io.load_data(
        str(tmp_path), K, P, Nm, device="cuda:0")
    assert (nu, ni, ne, nr) == (ratings[:, 0].max() + 1, ratings[:, 1].max() + 1, n_ent, n_rel)
    assert tuple(adj_e.shape) == (n_ent, K) and tuple(uts.shape) == (nu, P, 3, Nm)
    # every sampled (neighbor, relation) is an edge of the undirected KG
    edges = {(int(h), int(t), int(r)) for h, r, t in kg} | {(int(t), int(h), int(r)) for h, r, t in kg}
    ae, ar = adj_e.cpu().numpy(), adj_r.cpu().numpy()
    assert all((e, int(ae[e, k]), int(ar[e, k])) in edges for e in range(n_ent) for k in range(K))
    # hop-0 heads of a user's ripple set are items of its history
    u0 = next(iter(hist))
    assert set(uts[u0, 0, 0].cpu().tolist()) <= set(hist[u0])
    args = make_args(dim=16, neighbor_sample_size=K, h_hop=2, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=64)
    model = MVIN(args, nu, ne, nr, adj_e, adj_r, device="cuda:0", seed=1)
    feeder = harness.DeviceFeeder(model, uts)
    res = harness.ctr_eval_device(feeder, ev, 64)
    assert len(res[0]) == ev.shape[0] // 64 and 0.0 <= res[3] <= 1.0
    tr_rec, ev_rec, te_rec = (harness.get_user_record(d, flag) for d, flag in ((train, True), (ev, False), (test, False)))
    users = [u for u in te_rec if u in tr_rec][:5]
    p, r, n, _, _ = harness.topk_eval_device(feeder, users, tr_rec, ev_rec, te_rec, set(range(ni)), [1, 5], 64)
    assert len(p) == 2 and all(0.0 <= x <= 1.0 for x in p + r + n)
    torch.cuda.synchronize()


def test_train_loop_counterpart(hip_lib, tmp_path):
    """harness.train = train.py:16-109: epochs with CTR evaluation and early stopping, then the same
    with top-K evaluation, on the synthetic-signal data of scripts/run_synthetic.py."""
    import types
    import numpy as np
    from mvin_amd import data_prep, harness, synth
    from mvin_amd.config import make_args
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import run_synthetic
    rng = np.random.default_rng(0)
    n_user, n_item, n_ent, n_rel, K, P, Nm = 300, 200, 1500, 6, 8, 2, 16
    kg = np.stack([rng.integers(0, n_ent, 8000), rng.integers(0, n_rel, 8000), rng.integers(0, n_ent, 8000)], 1)
    inter = run_synthetic.synthetic_interactions(kg, n_user, n_item, n_ent, 30, rng)
    parts = np.split(inter[rng.permutation(inter.shape[0])], [int(0.6 * len(inter)), int(0.8 * len(inter))])
    csr = data_prep.build_csr(kg, n_ent, device="cuda:0")
    adj_e, adj_r = data_prep.construct_adj(csr, n_ent, K, seed=1)
    uts = data_prep.get_user_triplet_set(csr, data_prep.history_csr(parts[0], n_user, device="cuda:0"), n_user, P, Nm,
                                         seed=2)
    data = (n_user, n_item, n_ent, n_rel, parts[0], parts[1], parts[2], adj_e, adj_r, uts)
    args = make_args(dim=16, neighbor_sample_size=K, h_hop=2, n_mix_hop=1, p_hop=P, n_memory=Nm, batch_size=256,
                     lr=2e-2, l2_weight=1e-6, l2_agg_weight=1e-6)
    args.n_epochs, args.tolerance, args.early_stop, args.save_final_model = 4, 1, 2, True
    args.path = types.SimpleNamespace(emb=str(tmp_path / "emb"))
    model, hist = harness.train(args, data, device="cuda:0", rng=np.random.default_rng(1))
    assert 1 <= len(hist) <= 4 and hist[-1]["loss"] < hist[0]["loss"]
    assert all(0.0 <= h["eval"]["auc"] <= 1.0 for h in hist)
    assert os.path.exists(model._emb_path())                       # best-epoch checkpoint of the STWS tables
    args.n_epochs = 1
    _, hist2 = harness.train(args, data, show_topk=True, model=model, rng=np.random.default_rng(2))
    assert len(hist2[0]["eval"]["recall"]) == 7 and all(0.0 <= x <= 1.0 for x in hist2[0]["test"]["ndcg"])
