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
