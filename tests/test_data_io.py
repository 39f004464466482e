"""CPU: on-disk formats of the reference (mvin_amd/data_io.py) against loop restatements of the
reference's rules (oracle/io_ref.py) and a slice of the reference's own eval_pd.csv."""
import os

import numpy as np
import pytest

from mvin_amd import data_io
from oracle import io_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _write_csv(path, data, rng, cols=("item", "like", "user")):
    """The reference's layout: an unnamed running index, then the three columns in any order."""
    col_of = {"user": 0, "item": 1, "like": 2}
    with open(path, "w") as f:
        f.write("," + ",".join(cols) + "\n")
        for i, row in enumerate(data):
            f.write(str(i) + "," + ",".join(str(int(row[col_of[c]])) for c in cols) + "\n")


@pytest.fixture
def dataset(tmp_path):
    rng = np.random.default_rng(3)
    n_user, n_item, n = 40, 25, 600
    ratings = np.stack([rng.integers(0, n_user, n), rng.integers(0, n_item, n), rng.integers(0, 2, n)], 1)
    ratings[ratings[:, 0] == 7, 2] = 0          # user 7 never has a positive: must be filtered out everywhere
    kg = np.stack([rng.integers(0, 60, 300), rng.integers(0, 5, 300), rng.integers(0, 60, 300)], 1)
    np.savetxt(tmp_path / "ratings_final.txt", ratings, fmt="%d", delimiter="\t")
    np.save(tmp_path / "kg_final.npy", kg)
    parts = np.split(ratings[rng.permutation(n)], [360, 480])
    _write_csv(tmp_path / "train_pd.csv", parts[0], rng)
    _write_csv(tmp_path / "eval_pd.csv", parts[1], rng, cols=("user", "item", "like"))
    _write_csv(tmp_path / "test_pd.csv", parts[2], rng, cols=("like", "user", "item"))
    return str(tmp_path), ratings, kg, parts


def test_reference_csv_slice():
    got = data_io.read_pd_csv(os.path.join(GOLD, "io_ml1m_eval_pd_head.csv"))
    want = np.load(os.path.join(GOLD, "io_ml1m_eval_pd_head.npy"))
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert np.array_equal(got, io_ref.read_pd_csv_ref(os.path.join(GOLD, "io_ml1m_eval_pd_head.csv")))


def test_load_pre_data_column_order(dataset):
    d, _, _, parts = dataset
    for got, want in zip(data_io.load_pre_data(d), parts):
        assert np.array_equal(got, want)


def test_load_rating_matches_loop_rules(dataset):
    d, ratings, _, parts = dataset
    n_user, n_item, train, ev, test, hist, pop = data_io.load_rating(d, top_k=5)
    assert n_user == ratings[:, 0].max() + 1 and n_item == ratings[:, 1].max() + 1
    assert pop == io_ref.most_popular_items_ref(ratings, 5)
    want_hist = io_ref.user_history_ref(parts[0])
    assert hist == want_hist and 7 not in hist
    for got, raw in zip((train, ev, test), parts):
        assert np.array_equal(got, io_ref.filter_known_users_ref(raw, want_hist))
    assert not (train[:, 0] == 7).any()
    assert not os.path.exists(os.path.join(d, "ratings_final.npy"))      # nothing written unless asked
    data_io.load_rating(d, cache_npy=True)
    assert np.array_equal(np.load(os.path.join(d, "ratings_final.npy")), ratings)


def test_popular_items_tie_order():
    # counts: 4 -> 2, 9 -> 2, 1 -> 2, 5 -> 1; ties keep first-appearance order (4, 9, 1)
    r = np.array([[0, 4, 1], [0, 9, 1], [1, 1, 0], [1, 5, 1], [2, 9, 0], [2, 4, 1], [3, 1, 1]])
    for k in (1, 2, 3, 4):
        assert data_io.most_popular_items(r, k) == io_ref.most_popular_items_ref(r, k)
    assert data_io.most_popular_items(r, 2) == {4, 9}


def test_dataset_split_rule(dataset):
    _, ratings, _, _ = dataset
    train, ev, test = data_io.dataset_split(ratings, seed=1)
    n = ratings.shape[0]
    assert ev.shape[0] == int(n * 0.2) and test.shape[0] == int(n * 0.2)
    assert train.shape[0] == n - ev.shape[0] - test.shape[0]
    allrows = np.concatenate([train, ev, test])
    assert np.array_equal(np.sort(allrows.view([("", allrows.dtype)] * 3), axis=0),
                          np.sort(np.ascontiguousarray(ratings).view([("", ratings.dtype)] * 3), axis=0))
    t2, _, _ = data_io.dataset_split(ratings, ratio=0.5, seed=1)
    assert t2.shape[0] == int(train.shape[0] * 0.5)
    a, _, _ = data_io.dataset_split(ratings, seed=1)
    assert np.array_equal(a, train)                                        # same seed, same split


def test_kg_counts(dataset):
    d, _, kg, _ = dataset
    got, n_entity, n_relation = data_io.load_kg_triples(d)
    assert np.array_equal(got, kg)
    assert (n_entity, n_relation) == io_ref.kg_counts_ref(kg)


def test_missing_column_is_an_error(tmp_path):
    p = tmp_path / "bad.csv"
    p.write_text(",item,rating,user\n0,1,1,2\n")
    with pytest.raises(ValueError, match="missing column"):
        data_io.read_pd_csv(str(p))
