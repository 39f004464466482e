"""CPU restatement (Python integers / numpy) of the GPU samplers of mvin_amd/csrc/mvin_prep.hip
(TEST INFRASTRUCTURE).  Integer work: the GPU output must match bit for bit.

The sampling RULES follow the reference -- contruct_random_adj
(data_loader_user_set.py:375-388) and _get_user_triplet_set (:407-441); the draws themselves
cannot (the reference uses unseeded global generators), so the draw function below is this
repo's own: a splitmix64 finaliser over (seed, stream, a, b, c).
"""
import numpy as np

M64 = (1 << 64) - 1


def rnd32(seed, stream, a, b, c):
    z = (seed ^ (stream * 0xD1B54A32D192ED03) ^ (a * 0x9E3779B97F4A7C15) ^ (b * 0xC2B2AE3D27D4EB4F)
         ^ (c * 0x165667B19E3779F9)) & M64
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    return z >> 32


def rnd_below(n, seed, stream, a, b, c):
    return (rnd32(seed, stream, a, b, c) * n) >> 32


def floyd(n, k, draw):
    """k distinct values of [0, n) (Floyd); ``draw(i, bound)`` -> uniform in [0, bound)."""
    chosen = []
    for i in range(k):
        j = n - k + i
        t = draw(i, j + 1)
        chosen.append(j if t in chosen else t)
    return chosen


def build_csr(kg, n_entity):
    """construct_kg (:324-343) as CSR with the reference's per-entity insertion order."""
    nbrs = [[] for _ in range(n_entity)]
    for h, r, t in np.asarray(kg).tolist():
        nbrs[h].append((t, r))
        nbrs[t].append((h, r))
    indptr = np.zeros(n_entity + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(x) for x in nbrs])
    flat = [e for x in nbrs for e in x]
    dst = np.array([e[0] for e in flat], dtype=np.int32)
    rel = np.array([e[1] for e in flat], dtype=np.int32)
    return indptr, dst, rel


def sample_adjacency(indptr, dst, rel, n_entity, K, seed):
    adj_e = np.zeros((n_entity, K), dtype=np.int32)
    adj_r = np.zeros((n_entity, K), dtype=np.int32)
    for x in range(n_entity):
        lo, deg = int(indptr[x]), int(indptr[x + 1] - indptr[x])
        if deg == 0:
            continue
        if deg >= K:                                             # :383 without replacement
            pos = floyd(deg, K, lambda i, bound: rnd_below(bound, seed, 1, x, i, 0))
        else:                                                    # :384 with replacement
            pos = [rnd_below(deg, seed, 1, x, i, 0) for i in range(K)]
        adj_e[x] = dst[lo + np.array(pos)]
        adj_r[x] = rel[lo + np.array(pos)]
    return adj_e, adj_r


def ripple_sets(indptr, dst, rel, hist_ptr, hist_items, n_user, P, Nm, n_neighbor, seed):
    out = np.zeros((n_user, P, 3, Nm), dtype=np.int32)
    deg_of = np.diff(indptr)
    for u in range(n_user):
        tails = np.zeros(Nm, dtype=np.int64)
        for h in range(P):
            seeds = hist_items[hist_ptr[u]:hist_ptr[u + 1]] if h == 0 else tails
            cnt = [min(int(deg_of[e]), n_neighbor) for e in seeds]
            C = sum(cnt)
            if C == 0:
                if h == 0:                                       # no usable history: no entry (zero rows)
                    break
                out[u, h] = out[u, h - 1]                        # :429-430 copy the previous hop
                continue
            if C >= Nm:                                          # :433-434 replace = len < n_memory
                V = floyd(C, Nm, lambda i, bound: rnd_below(bound, seed, 2, u, h, i))
            else:
                V = [rnd_below(C, seed, 2, u, h, i) for i in range(Nm)]
            prefix = np.concatenate([[0], np.cumsum(cnt)])
            for m, v in enumerate(V):
                s = int(np.searchsorted(prefix, v, side="right") - 1)
                while cnt[s] == 0:                               # skip empty seeds (ties in prefix)
                    s += 1
                w = v - int(prefix[s])
                e = int(seeds[s])
                deg = int(deg_of[e])
                pick = w
                if deg > n_neighbor:                             # :421 random.sample(g_kg[entity], 16)
                    sub = floyd(deg, n_neighbor, lambda i, bound: rnd_below(bound, seed, 3, (u << 8) | h, s, i))
                    pick = sub[w]
                pos = int(indptr[e]) + pick
                out[u, h, 0, m], out[u, h, 1, m], out[u, h, 2, m] = e, rel[pos], dst[pos]
            tails = out[u, h, 2].astype(np.int64)
    return out


def encode_adjacency(adj_e, adj_r):
    """Restatement of mvin_encode_adjacency (mvin_prep.hip): the duplicate-slot encoding of a sampled adjacency.

    The reference's sampler draws K neighbours WITH replacement whenever an entity has fewer than K edges
    (data_loader_user_set.py:383-384), so a row of adj_entity / adj_relation repeats (neighbour, relation) slots.
    Row x is rewritten as: its DISTINCT (neighbour, relation) slots first -- ordered by how many distinct slots the
    neighbour's own row has (descending; ties in first-occurrence order) --, then padding that repeats slot 0.
      enc_e[x, i] = neighbour id | cnt[neighbour] << 24          (n_entity <= 2^24)
      enc_r[x, i] = relation | multiplicity << 16 | cnt[x] << 24      (multiplicity 0 = padding slot)
    Returns (enc_e, enc_r, cnt) with cnt [n_entity] = distinct slots per row.  Pure integer work: bit-exact."""
    adj_e = np.asarray(adj_e, dtype=np.int64)
    adj_r = np.asarray(adj_r, dtype=np.int64)
    nE, K = adj_e.shape
    assert K <= 128 and nE <= (1 << 24)
    # ids are clamped into their fields first (neighbour into the table, relation into 16 bits), like every kernel that
    # indexes a table with a device-resident id; two slots that clamp to the same (neighbour, relation) are one slot
    # (unsigned min, the library's rule: a negative id is a huge unsigned one and becomes the field's last value)
    adj_e = np.where((adj_e < 0) | (adj_e > nE - 1), nE - 1, adj_e)
    adj_r = np.where((adj_r < 0) | (adj_r > 0xFFFF), 0xFFFF, adj_r)
    firsts, mults = [], []
    cnt = np.zeros(nE, dtype=np.int32)
    for x in range(nE):
        seen = {}
        for s in range(K):
            key = (int(adj_e[x, s]), int(adj_r[x, s]))
            if key in seen:
                seen[key][1] += 1
            else:
                seen[key] = [s, 1]
        firsts.append(seen)
        cnt[x] = len(seen)
    enc_e = np.zeros((nE, K), dtype=np.int64)
    enc_r = np.zeros((nE, K), dtype=np.int64)
    for x in range(nE):
        items = sorted(firsts[x].items(), key=lambda kv: (-int(cnt[kv[0][0]]), kv[1][0]))
        c = len(items)
        for i, ((y, r), (_, m)) in enumerate(items):
            enc_e[x, i] = y | (int(cnt[y]) << 24)
            enc_r[x, i] = r | (m << 16) | (c << 24)
        enc_e[x, c:] = enc_e[x, 0]
        enc_r[x, c:] = (enc_r[x, 0] & 0xFFFF) | (c << 24)
    return enc_e.astype(np.uint32).view(np.int32), enc_r.astype(np.uint32).view(np.int32), cnt      # cnt = 128 sets the sign bit: the word is unsigned


def user_records_layout(P, Nm, nR):
    """Word offsets of one user's static record (mvin_user_records_len / ka_rec_layout, mvin_keyaddr_static.hip)."""
    pad4 = lambda v: (v + 3) & ~3
    NmP = (Nm + 15) & ~15
    rows = P * NmP
    maxtiles = rows // 16 + min(nR, P * Nm)
    o_cnt = 4
    o_off = o_cnt + pad4(nR)
    o_trel = o_off + pad4(nR)
    o_bidx = o_trel + pad4(maxtiles)
    o_head = o_bidx + maxtiles * 16
    o_tail = o_head + rows
    o_hr = o_tail + rows
    return dict(NmP=NmP, rows=rows, maxtiles=maxtiles, o_cnt=o_cnt, o_off=o_off, o_trel=o_trel, o_bidx=o_bidx, o_head=o_head,
                o_tail=o_tail, o_hr=o_hr, len=(o_hr + rows + 63) & ~63)


def user_records(uts, nR, n_entity):
    """Restatement of mvin_build_user_records: what the dense grouped key-addressing kernel derives from a user's ripple
    sets alone (uts [n_user, P, 3, Nm]: the (h, r, t) memories user_triplet_set feeds for that user in every batch,
    model.py:66-76 / data_loader_user_set.py), one record per user:
      [0] tiles, [1..3] 0 | members per relation | first bucket row per relation | relation of each tile |
      bucket slots -> row hop * NmP + m (rows of a relation -- they share R_KGE[r], model.py:214-216 -- in row order, every
      bucket padded to whole 16-row tiles) | head id per row | tail id per row | relation * n_entity + head per row (the row of
      mvin_project_relations' [nR, nE, D] table; 0 where that does not fit 31 bits); ids clamped into the tables, all other words -1.
    Pure integer work: bit-exact."""
    uts = np.asarray(uts, dtype=np.int64)
    n_user, P, three, Nm = uts.shape
    assert three == 3
    L = user_records_layout(P, Nm, nR)
    rec = np.full((n_user, L["len"]), -1, dtype=np.int32)
    for u in range(n_user):
        r_ = rec[u]
        r_[1:4] = 0
        members = [[] for _ in range(nR)]
        for hop in range(P):
            for m in range(Nm):
                i = hop * L["NmP"] + m
                rel = min(int(uts[u, hop, 1, m]) & 0xFFFFFFFF, nR - 1)
                members[rel].append(i)                                                    # clamped as unsigned words, like every device id
                r_[L["o_head"] + i] = min(int(uts[u, hop, 0, m]) & 0xFFFFFFFF, n_entity - 1)
                hr = rel * n_entity + int(r_[L["o_head"] + i])
                r_[L["o_hr"] + i] = hr if hr < (1 << 31) else 0
                r_[L["o_tail"] + i] = min(int(uts[u, hop, 2, m]) & 0xFFFFFFFF, n_entity - 1)
        tile = 0
        for r in range(nR):
            cnt = len(members[r])
            r_[L["o_cnt"] + r] = cnt
            r_[L["o_off"] + r] = tile * 16
            for j, i in enumerate(members[r]):
                r_[L["o_bidx"] + tile * 16 + j] = i
            nt = (cnt + 15) // 16
            r_[L["o_trel"] + tile:L["o_trel"] + tile + nt] = r
            tile += nt
        r_[0] = tile
    return rec


def decode_adjacency(enc_e, enc_r):
    """Inverse up to slot order: the multiset of (neighbour, relation) slots of every row, as a sorted [nE, K, 2] array."""
    enc_e = np.asarray(enc_e).astype(np.int32).view(np.uint32).astype(np.int64) & 0xFFFFFF
    enc_r = np.asarray(enc_r).astype(np.int32).view(np.uint32).astype(np.int64)
    nE, K = enc_e.shape
    out = np.zeros((nE, K, 2), dtype=np.int64)
    for x in range(nE):
        rows = []
        for i in range(K):
            m = (enc_r[x, i] >> 16) & 0xFF
            rows += [(enc_e[x, i], enc_r[x, i] & 0xFFFF)] * int(m)
        assert len(rows) == K, (x, len(rows))
        out[x] = sorted(rows)
    return out
