"""CPU: the sampling rules of the data-preparation oracle (oracle/prep_ref.py) against the
reference's rules (data_loader_user_set.py:375-388, :407-441)."""
import numpy as np

from mvin_amd import synth
from oracle import prep_ref


def small_kg(n_entity=80, seed=0):
    kg = synth.synth_kg(n_entity, 4, 7.0, seed=seed)
    return kg, prep_ref.build_csr(kg, n_entity)


def test_csr_keeps_reference_insertion_order():
    kg = np.array([[0, 1, 2], [2, 0, 0], [3, 2, 3], [0, 3, 1]])
    indptr, dst, rel = prep_ref.build_csr(kg, 5)
    # entity 0: (2,1) from triple 0 as head, (2,0) from triple 1 as tail, (1,3) from triple 3
    assert list(zip(dst[indptr[0]:indptr[1]], rel[indptr[0]:indptr[1]])) == [(2, 1), (2, 0), (1, 3)]
    # self loop 3-3 is listed twice (:336, :339)
    assert list(zip(dst[indptr[3]:indptr[4]], rel[indptr[3]:indptr[4]])) == [(3, 2), (3, 2)]
    assert indptr[5] == 8 and indptr[4] == indptr[5]     # entity 4 absent


def test_adjacency_rule():
    kg, (indptr, dst, rel) = small_kg()
    K = 5
    ae, ar = prep_ref.sample_adjacency(indptr, dst, rel, 80, K, seed=11)
    deg = np.diff(indptr)
    assert (deg >= K).any() and ((deg > 0) & (deg < K)).any()
    for x in range(80):
        edges = list(zip(dst[indptr[x]:indptr[x + 1]].tolist(), rel[indptr[x]:indptr[x + 1]].tolist()))
        got = list(zip(ae[x].tolist(), ar[x].tolist()))
        if deg[x] == 0:
            assert got == [(0, 0)] * K
        elif deg[x] >= K:
            pool = list(edges)
            for g in got:                # distinct edge positions: a sub-multiset of the edge list
                assert g in pool
                pool.remove(g)
        else:
            assert set(got) <= set(edges)
    # a different seed gives a different sample, the same seed the same one
    ae2, _ = prep_ref.sample_adjacency(indptr, dst, rel, 80, K, seed=12)
    ae3, _ = prep_ref.sample_adjacency(indptr, dst, rel, 80, K, seed=11)
    assert (ae != ae2).any() and (ae == ae3).all()


def test_floyd_is_uniform_over_subsets():
    counts = {}
    for trial in range(6000):
        s = tuple(sorted(prep_ref.floyd(5, 2, lambda i, b: prep_ref.rnd_below(b, 99, 7, trial, i, 0))))
        counts[s] = counts.get(s, 0) + 1
    assert len(counts) == 10 and min(counts.values()) > 480 and max(counts.values()) < 720


def test_ripple_set_rule():
    kg, (indptr, dst, rel) = small_kg(seed=3)
    rng = np.random.default_rng(1)
    n_user, Nm, P = 9, 8, 2
    hist = [rng.choice(30, size=rng.integers(0, 5), replace=False) for _ in range(n_user)]
    hist_ptr = np.concatenate([[0], np.cumsum([len(h) for h in hist])]).astype(np.int64)
    hist_items = np.concatenate(hist + [np.zeros(0, int)]).astype(np.int32)
    out = prep_ref.ripple_sets(indptr, dst, rel, hist_ptr, hist_items, n_user, P, Nm, 16, seed=5)
    deg = np.diff(indptr)
    for u in range(n_user):
        seeds0 = set(hist[u].tolist())
        if sum(min(deg[e], 16) for e in seeds0) == 0:
            assert not out[u].any()
            continue
        for h in range(P):
            heads, rels, tails = out[u, h]
            seeds = seeds0 if h == 0 else set(out[u, h - 1, 2].tolist())
            for e, r, t in zip(heads, rels, tails):
                assert e in seeds                                   # heads are seeds (:422)
                edges = set(zip(dst[indptr[e]:indptr[e + 1]].tolist(), rel[indptr[e]:indptr[e + 1]].tolist()))
                assert (t, r) in edges                              # (tail, relation) is an edge of the head
        # without replacement when enough candidates (:433)
        C0 = sum(min(deg[e], 16) for e in hist[u])
        if C0 >= Nm and all(deg[e] <= 16 for e in hist[u]):
            trip = list(zip(*out[u, 0].tolist()))
            edge_mult = {}
            for e in hist[u]:
                for t, r in zip(dst[indptr[e]:indptr[e + 1]].tolist(), rel[indptr[e]:indptr[e + 1]].tolist()):
                    edge_mult[(e, r, t)] = edge_mult.get((e, r, t), 0) + 1
            for k in set(trip):
                assert trip.count(k) <= edge_mult[k]


def test_encode_adjacency_round_trip_and_order():
    """Duplicate-slot encoding (oracle of mvin_encode_adjacency): decoding gives back every row's multiset of
    (neighbour, relation) slots; distinct slots come first, ordered by the neighbour's own distinct count; an all-zero
    row (entity absent from the KG, data_loader_user_set.py:377-380) is ONE slot of multiplicity K."""
    import numpy as np
    from oracle import prep_ref
    rng = np.random.default_rng(3)
    for K in (4, 16, 32, 128):
        nE = 80
        e = rng.integers(0, nE, (nE, K))
        r = rng.integers(0, 7, (nE, K))
        low = rng.random(nE) < 0.6                      # low-degree entities: draws with replacement from few edges
        for x in np.flatnonzero(low):
            d = rng.integers(1, max(2, K // 3))
            pick = rng.integers(0, d, K)
            e[x], r[x] = e[x][pick], r[x][pick]
        e[5], r[5] = 0, 0
        ee, er, cnt = prep_ref.encode_adjacency(e, r)
        dec = prep_ref.decode_adjacency(ee, er)
        ref = np.array([sorted(map(tuple, row)) for row in np.stack([e, r], -1)])
        assert np.array_equal(dec, ref)
        eu = ee.view(np.uint32).astype(np.int64)
        ru = er.view(np.uint32).astype(np.int64)
        assert cnt[5] == 1 and ((ru[5, 0] >> 16) & 0xFF) == K and (ru[5, 1:] >> 16 & 0xFF).max() == 0
        for x in range(nE):
            c = cnt[x]
            assert ((ru[x] >> 24) == c).all()
            assert ((ru[x, :c] >> 16) & 0xFF).min() >= 1 and ((ru[x, :c] >> 16) & 0xFF).sum() == K
            child_cnt = eu[x, :c] >> 24
            assert np.array_equal(child_cnt, cnt[eu[x, :c] & 0xFFFFFF])
            assert (np.diff(child_cnt) <= 0).all()       # longest lists first
            assert (eu[x, c:] == eu[x, 0]).all()


def test_user_records_layout_and_buckets(hip_lib):
    """oracle.prep_ref.user_records (the restatement of mvin_build_user_records): every memory row sits in exactly one slot of
    its relation's bucket, in row order; buckets are whole 16-row tiles; ids are clamped; the record length is the library's
    (mvin_user_records_len is host arithmetic: no GPU needed)."""
    for P, Nm, nR in [(2, 64, 9), (1, 16, 39), (3, 40, 7), (2, 20, 100), (1, 1, 1)]:
        n_user, n_entity = 9, 123
        uts = synth.ripple_sets(n_user, n_entity, nR, P, Nm, seed=P * Nm + nR)
        uts[0, 0, 0, 0] = n_entity + 7                                    # clamped like every device id
        uts[0, P - 1, 1, Nm - 1] = nR + 2
        rec = prep_ref.user_records(uts, nR, n_entity)
        L = prep_ref.user_records_layout(P, Nm, nR)
        assert rec.shape == (n_user, L["len"]) and L["len"] % 64 == 0
        assert hip_lib.mvin_user_records_len(P, Nm, nR) == L["len"]
        for u in range(n_user):
            r_ = rec[u]
            cnt, off = r_[L["o_cnt"]:L["o_cnt"] + nR], r_[L["o_off"]:L["o_off"] + nR]
            assert cnt.sum() == P * Nm and r_[0] == sum((c + 15) // 16 for c in cnt) <= L["maxtiles"]
            seen = []
            for r in range(nR):
                assert off[r] % 16 == 0
                slots = r_[L["o_bidx"] + off[r]:L["o_bidx"] + off[r] + ((cnt[r] + 15) // 16) * 16]
                rows = [int(v) for v in slots if v >= 0]
                assert rows == sorted(rows) and len(rows) == cnt[r] and all(v == -1 for v in slots[cnt[r]:])
                for i in rows:
                    hop, m = divmod(i, L["NmP"])
                    assert min(int(uts[u, hop, 1, m]), nR - 1) == r
                    assert r_[L["o_head"] + i] == min(int(uts[u, hop, 0, m]), n_entity - 1)
                    assert r_[L["o_tail"] + i] == min(int(uts[u, hop, 2, m]), n_entity - 1)
                assert all(r_[L["o_trel"] + off[r] // 16 + t] == r for t in range((cnt[r] + 15) // 16))
                seen += rows
            assert sorted(seen) == [h * L["NmP"] + m for h in range(P) for m in range(Nm)]
    assert hip_lib.mvin_user_records_len(0, 16, 5) == 0 and hip_lib.mvin_user_records_len(2, 300, 5) == 0
