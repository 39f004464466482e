"""CPU, world_size=2 over gloo: the row-sharded entity table (cyclic ownership x mod W, shard-space ids)
and its row fetch (mvin_amd/dist.py).  The row movers are torch indexing here (the HIP kernels in
production), everything else is the production code path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.dist import (ShardedEntityTable, exchange_wire_bytes, from_shard_space, mark_needed, mark_needed_static,
                           n_local_rows, permute_adjacency, permute_ripple_sets, shard_rows, to_shard_space)
from mvin_amd.params import init_params


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _torch_gather(table, idx):
    return table[idx.long()]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import mirror_fp32
        args = make_args(dim=8, neighbor_sample_size=4, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=4, batch_size=6)
        case = synth.small_case(args, n_user=10, n_entity=97, n_relation=5, seed=40)   # same on every rank
        params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=41)
        full = torch.from_numpy(params["entity_emb_matrix"])
        sl = slice(rank * 3, rank * 3 + 3)                                             # my pairs
        nE, nl = case.n_entity, n_local_rows(case.n_entity, world)
        pi = lambda x: to_shard_space(x, nE, world)
        items_p = pi(torch.from_numpy(case.items[sl]))
        mh_p = [pi(torch.from_numpy(m[sl]).long()) for m in case.memories_h]
        mt_p = [pi(torch.from_numpy(m[sl]).long()) for m in case.memories_t]
        pe, pr = permute_adjacency(case.adj_entity, case.adj_relation, nE, world)
        adj_p = torch.from_numpy(pe)
        full_p = torch.zeros((world * nl, full.shape[1]))                    # the table in shard space
        full_p[pi(torch.arange(nE))] = full
        need = mark_needed(world * nl, adj_p, items_p, 2, extra_ids=mh_p + mt_p)
        table = ShardedEntityTable(shard_rows(full, rank, world), nE, rank, world, _torch_gather)
        assert table.local.shape[0] == nl and torch.equal(table.local[:len(range(rank, nE, world))], full[rank::world])
        # ---- sparse regime: only the touched rows move ----
        work = table.fetch(need)
        idx = need.nonzero(as_tuple=True)[0]
        assert torch.equal(work[idx], full_p[idx]), "fetched rows differ from the owners' rows"
        assert table.last_stats["requested"] == idx.numel() and table.last_stats["remote"] > 0
        rest = torch.ones(work.shape[0], dtype=torch.bool)
        rest[idx] = False
        assert not work[rest].any()                                        # untouched rows never written
        # the marked set covers everything the scoring path reads: scores in shard space from the working
        # table == scores from the full table in the original id space
        sargs = make_args(**dict(vars(args), batch_size=3))
        feed = (case.users[sl], case.items[sl], [m[sl] for m in case.memories_h],
                [m[sl] for m in case.memories_r], [m[sl] for m in case.memories_t])
        ref = mirror_fp32.forward(sargs, params, case.adj_entity, case.adj_relation, *feed)
        feed_p = (case.users[sl], items_p.numpy(), [m.numpy() for m in mh_p], feed[3], [m.numpy() for m in mt_p])
        got = mirror_fp32.forward(sargs, dict(params, entity_emb_matrix=work.numpy()), pe, pr, *feed_p)
        assert torch.equal(ref.scores, got.scores)
        # a second fetch with a different need set reuses the working table
        need2 = torch.zeros_like(need)
        need2[[1, 2, 50 + rank]] = True
        work = table.fetch(need2)
        assert torch.equal(work[need2], full_p[need2])
        # ---- sparse regime with fixed-capacity buffers (no host sync: static splits, two collectives) ----
        assert torch.equal(mark_needed_static(world * nl, adj_p, items_p, 2, extra_ids=mh_p + mt_p), need)
        cap = 3 * (1 + 4 + 16 + 2 * 2 * 4)                                 # the batch share's row references: a static bound
        w3 = table.fetch_static(need, cap, table.new_work_table())
        assert torch.equal(w3[idx], full_p[idx]) and not bool(table.overflow)
        touched = w3.abs().sum(1) > 0                                      # beyond the needed rows only the padding row of each owner
        extra = set(touched.nonzero(as_tuple=True)[0].tolist()) - set(idx.tolist())
        assert extra <= {w * nl for w in range(world)}
        assert all(torch.equal(w3[e], full_p[e]) for e in extra)
        assert table.last_stats["mode"] == "sparse" and table.last_stats["requested"] == idx.numel()
        assert table.last_stats["wire_bytes_per_rank"] == (world - 1) * min(cap, nl) * (4 + 4 * full.shape[1])
        table.fetch_static(need, 2, table.new_work_table())                # too small a capacity is FLAGGED (device flag)
        assert bool(table.overflow)
        # ---- dense regime: every rank receives every shard ----
        w2 = table.fetch_all(table.new_work_table())
        assert torch.equal(w2, full_p)
        assert table.last_stats["mode"] == "dense"
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        ret[rank] = f"FAIL {type(e).__name__}: {e}"
        raise
    finally:
        dist.destroy_process_group()


def test_sharded_table_all_to_all_fetch_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_cyclic_partition_and_shard_space():
    full = torch.arange(11 * 4, dtype=torch.float32).view(11, 4) + 1
    for world in (1, 2, 3, 8):
        nl = n_local_rows(11, world)
        shards = [shard_rows(full, r, world) for r in range(world)]
        assert all(s.shape == (nl, 4) for s in shards)
        x = torch.arange(11)
        p = to_shard_space(x, 11, world)
        assert torch.equal(from_shard_space(p, 11, world), x) and len(set(p.tolist())) == 11
        assert torch.equal(p // nl, x % world)                     # owner(x) = x mod W is a contiguous block
        assert torch.equal(torch.cat(shards)[p], full)             # all-gather of the shards = the table in shard space
    # adjacency and ripple sets relabelled consistently
    rng = np.random.default_rng(0)
    adj_e, adj_r = rng.integers(0, 11, (11, 3)), rng.integers(0, 4, (11, 3))
    pe, pr = permute_adjacency(adj_e, adj_r, 11, 4)
    for x in range(11):
        px = int(to_shard_space(np.int64(x), 11, 4))
        assert from_shard_space(pe[px].astype(np.int64), 11, 4).tolist() == adj_e[x].tolist()
        assert pr[px].tolist() == adj_r[x].tolist()
    uts = rng.integers(0, 11, (5, 2, 3, 4)).astype(np.int32)
    pu = permute_ripple_sets(uts, 11, 4)
    assert np.array_equal(pu[:, :, 1], uts[:, :, 1])
    assert np.array_equal(from_shard_space(pu[:, :, 0].astype(np.int64), 11, 4), uts[:, :, 0])
    # single rank: shard space is the identity
    t = ShardedEntityTable(full.clone(), 11, 0, 1, _torch_gather)
    need = torch.zeros(11, dtype=torch.bool)
    need[[0, 3, 10]] = True
    w = t.fetch(need)
    assert torch.equal(w[need], full[need]) and not w[~need].any()
    assert torch.equal(t.fetch_all(), full)


def test_wire_bytes_argument():
    """owner-side partial sums against replicating the shards, bytes a rank receives per step."""
    for nE, row, B, K, L in ((106389, 256, 524288, 32, 2), (113487, 256, 32768, 64, 2), (113487, 256, 64, 128, 3)):
        b = exchange_wire_bytes(nE, row, 8, B // 8, K, L)
        assert b["replicate"] == 7 * n_local_rows(nE, 8) * row
        assert b["partial_sums"] > 9 * b["replicate"]         # every BASELINE config: the tree dwarfs the table (C5: 235 vs 25 MB)
    small = exchange_wire_bytes(113487, 256, 8, 4, 128, 2)   # 4 pairs per rank at depth 2: 512 nodes < 14 186 rows per shard
    assert small["partial_sums"] < small["replicate"]


def test_mark_needed_matches_bruteforce_tree():
    rng = np.random.default_rng(0)
    nE, K = 60, 3
    adj = rng.integers(0, nE, (nE, K))
    items = np.array([5, 17, 5])
    want = set(items.tolist())
    level = items
    for _ in range(3):
        level = adj[level].reshape(-1)
        want |= set(level.tolist())
    extra = np.array([[58, 59]])
    want |= {58, 59}
    got = mark_needed(nE, torch.from_numpy(adj.astype(np.int32)), torch.from_numpy(items), 3,
                      [torch.from_numpy(extra)])
    assert set(got.nonzero(as_tuple=True)[0].tolist()) == want


def test_static_sparse_band_is_dense():
    """with fixed-capacity id buffers every owner is asked for min(refs, n_local) rows, so from refs >= n_local
    on the 'sparse' step would move the whole table plus ids, a sort and a scatter: that band takes the dense regime.  The
    count-exchange form keeps the table-size threshold."""
    from types import SimpleNamespace
    from mvin_amd.dist import ShardedMVIN, n_local_rows
    nE, W = 113487, 8
    model = SimpleNamespace(n_neighbor=64, n_memory=16, p_hop=1, n_mix_hop=1, h_hop=2,
                            args=SimpleNamespace(PS_only=False, wide_deep=True))
    stub = SimpleNamespace(regime="auto", model=model, world=W, n_entity=nE, static_sparse=True,
                           table=SimpleNamespace(n_local=n_local_rows(nE, W)))
    stub._depth = lambda: ShardedMVIN._depth(stub)
    per_pair = 1 + 64 + 64 * 64 + 2 * 16
    just_below = (n_local_rows(nE, W) - 1) // per_pair           # pairs per rank whose references stay below a shard
    assert not ShardedMVIN.is_dense(stub, just_below * W)
    assert ShardedMVIN.is_dense(stub, (just_below + 1) * W)       # refs >= n_local: dense, although refs << n_entity
    stub.static_sparse = False
    assert not ShardedMVIN.is_dense(stub, (just_below + 1) * W)
    assert ShardedMVIN.is_dense(stub, (nE // per_pair + 1) * W)
