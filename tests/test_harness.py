"""CPU: harness counterpart logic (mvin_amd/harness.py) against brute-force restatements,
with a stand-in model that exposes the reference's run-wrapper surface."""
import numpy as np

from mvin_amd import harness
from mvin_amd.config import make_args


class FakeModel:
    """Scores = a fixed function of (user, item); same placeholder/run-wrapper surface as MVIN."""

    def __init__(self, p_hop):
        self.user_indices, self.item_indices, self.labels = "u", "i", "l"
        self.memories_h = [f"h{i}" for i in range(max(1, p_hop))]
        self.memories_r = [f"r{i}" for i in range(max(1, p_hop))]
        self.memories_t = [f"t{i}" for i in range(max(1, p_hop))]
        self.calls = []

    @staticmethod
    def score(u, i):
        return 1.0 / (1.0 + np.exp(-np.sin(0.37 * np.asarray(u) + 1.3 * np.asarray(i))))

    def get_scores(self, sess, feed):
        self.calls.append(len(feed[self.item_indices]))
        assert len(feed[self.memories_h[0]]) == len(feed[self.item_indices])
        return np.asarray(feed[self.item_indices]), self.score(feed[self.user_indices], feed[self.item_indices])

    def eval(self, sess, feed):
        from sklearn.metrics import f1_score, roc_auc_score
        s = self.score(feed[self.user_indices], feed[self.item_indices])
        lab = np.asarray(feed[self.labels], dtype=np.float32)
        pred = (s >= 0.5).astype(np.float32)
        return roc_auc_score(lab, s), float(np.mean(pred == lab)), f1_score(lab, pred)


def make_data(n_user=12, n_item=40, n=500, seed=0):
    rng = np.random.default_rng(seed)
    d = np.stack([rng.integers(0, n_user, n), rng.integers(0, n_item, n), rng.integers(0, 2, n)], axis=1)
    uts = rng.integers(0, 50, (n_user, 2, 3, 4)).astype(np.int32)
    return d, uts


def test_feed_dict_assembly_matches_reference_layout():
    args = make_args(p_hop=2, n_memory=4)
    model = FakeModel(2)
    data, uts = make_data()
    feed = harness.get_feed_dict(args, model, data, uts, 3, 9)
    np.testing.assert_array_equal(feed["u"], data[3:9, 0])
    assert len(feed["h1"]) == 6
    for row, u in zip(feed["t1"], data[3:9, 0]):
        np.testing.assert_array_equal(row, uts[u][1][2])       # user_triplet_set[user][hop][2] = tails
    f2 = harness.get_feed_dict_top_k(args, model, [5, 5], [1, 2], [1, 1], uts)
    np.testing.assert_array_equal(f2["r0"][1], uts[5][0][1])


def test_ctr_eval_drops_ragged_tail_and_averages_batches():
    args = make_args(p_hop=2, n_memory=4, batch_size=64)
    data, uts = make_data(n=500)
    model = FakeModel(2)
    aucs, accs, f1s, auc, acc, f1 = harness.ctr_eval(args, model, data, uts, 64)
    assert len(aucs) == 500 // 64                                  # util.py:49
    assert abs(auc - np.mean(aucs)) < 1e-12 and abs(f1 - np.mean(f1s)) < 1e-12


def test_user_record_and_topk_settings():
    data, _ = make_data()
    rec = harness.get_user_record(data)
    for u, items in rec.items():
        assert items == set(data[(data[:, 0] == u) & (data[:, 2] == 1), 1].tolist())
    users, tr, ev, te, item_set, k_list = harness.topk_settings(data[:300], data[300:400], data[400:], 40, user_num=5)
    assert len(users) <= 5 and item_set == set(range(40)) and k_list[-1] == 100
    counts = [len(tr[u]) for u in users]
    assert counts == sorted(counts, reverse=True)


def test_topk_eval_padding_and_stale_k_ndcg():
    args = make_args(p_hop=2, n_memory=4, batch_size=16)
    data, uts = make_data(n_item=40, n=600, seed=3)
    users, tr, ev, te, item_set, k_list = harness.topk_settings(data[:400], data[400:500], data[500:], 40, user_num=6)
    k_list = [1, 2, 5, 10]
    model = FakeModel(2)
    prec, rec, ndcg, _, _ = harness.topk_eval(args, uts, model, users, tr, ev, te, item_set, k_list, 16, mode="test")
    assert all(c == 16 for c in model.calls)                       # every call is a full (padded) batch
    # brute force
    P, R, N = {k: [] for k in k_list}, {k: [] for k in k_list}, {k: [] for k in k_list}
    for u in users:
        if u not in te:
            continue
        cand = list(item_set - tr[u])
        ranked = [i for i, _ in sorted(zip(cand, FakeModel.score(u, cand)), key=lambda x: x[1], reverse=True)]
        for k in k_list:
            P[k].append(len(set(ranked[:k]) & te[u]) / k)
            R[k].append(len(set(ranked[:k]) & te[u]) / len(te[u]))
        hits = [1 if i in te[u] else 0 for i in ranked[:k_list[-1]]]   # stale k = last of k_list
        for k in k_list:
            dcg = sum(h / np.log2(j + 2) for j, h in enumerate(hits[:k]))
            best = sum(h / np.log2(j + 2) for j, h in enumerate(sorted(hits, reverse=True)[:k]))
            N[k].append(dcg / best if best else 0.0)
    np.testing.assert_allclose(prec, [np.mean(P[k]) for k in k_list])
    np.testing.assert_allclose(rec, [np.mean(R[k]) for k in k_list])
    np.testing.assert_allclose(ndcg, [np.mean(N[k]) for k in k_list])


def test_ranking_metric_definitions():
    ranked, truth = [4, 9, 1, 7, 3], {9, 3, 8}
    assert harness.precision_at_k(ranked, truth, 2) == 0.5
    assert abs(harness.recall_at_k(ranked, truth, 5) - 2 / 3) < 1e-12
    assert abs(harness.dcg_at_k([0, 1, 0, 0, 1], 5) - (1 / np.log2(3) + 1 / np.log2(6))) < 1e-12
    assert harness.ndcg_at_k([1, 1, 0], 3) == 1.0 and harness.ndcg_at_k([0, 0], 2) == 0.0


def test_early_stop_rule():
    """train_util.py:33-61: saves on every improvement, counts non-improving epochs only after
    ``tolerance`` epochs, stops at ``early_stop`` of them in a row."""
    from mvin_amd.harness import EarlyStop

    class M(object):
        path = None
        saved = 0

        def save_pretrain_emb_fuc(self, *a):
            self.saved += 1

    m = M()
    st = EarlyStop(tolerance=2, early_stop=2, save_final_model=True)
    scores = [0.5, 0.4, 0.6, 0.55, 0.58, 0.7]
    stopped = []
    for e, sc in enumerate(scores):
        stopped.append(st.update(e, sc, m))
        if stopped[-1]:
            break
    # epochs 0,1 are inside the tolerance window; best is set at epoch 2 (0.6); 3 and 4 do not improve -> stop at 4
    assert stopped == [False, False, False, False, True]
    assert st.best == 0.6 and m.saved == 0            # no path.emb: nothing written
    import types
    m.path = types.SimpleNamespace(emb="/tmp/x")
    st2 = EarlyStop(tolerance=0, early_stop=3)
    for e, s in enumerate([0.1, 0.2, 0.15, 0.3]):
        st2.update(e, s, m)
    assert m.saved == 3                                # improvements at epochs 0, 1, 3
