"""Worker of tests/test_gpu_dist.py::test_ranks_time_sharing_one_gpu: one of WORLD ranks that time-share
cuda:0 and talk over gloo (RCCL refuses two ranks on one device).  Everything but the transport
is the production path: HIP kernels, ShardedMVIN in both regimes, the two-stream pipeline."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mvin_amd import synth  # noqa: E402
from mvin_amd.config import make_args  # noqa: E402
from mvin_amd.dist import ShardedMVIN, shard_rows  # noqa: E402
from mvin_amd import synth as _synth  # noqa: E402,F401
from mvin_amd.model import MVIN  # noqa: E402
from mvin_amd.params import init_params  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    Bl = 48
    args = make_args(dim=32, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=16, batch_size=Bl)
    case = synth.small_case(make_args(**dict(vars(args), batch_size=Bl * world)),   # the global batch
                            n_user=50, n_entity=5003, n_relation=7, seed=61)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=62, random_agg_bias=True)
    mk = lambda p: MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                        params=p, device=dev)
    ref_model = mk(params)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    feed = (torch.from_numpy(case.users[sl]).to(dev), torch.from_numpy(case.items[sl]).to(dev),
            [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_h],
            [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_r],
            [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_t])
    ref = ref_model.forward_device(*feed).scores
    full = torch.from_numpy(params["entity_emb_matrix"])
    zeroed = dict(params, entity_emb_matrix=np.zeros_like(params["entity_emb_matrix"]))
    build = lambda regime, **kw: ShardedMVIN.build(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity,
                                                   case.adj_relation, params, shard_rows(full, rank, world), rank, world,
                                                   device=dev, regime=regime, **kw)
    uts = torch.from_numpy(_synth.ripple_sets(case.n_user, case.n_entity, case.n_relation, 2, 16, seed=63)).to(dev)
    mhu, mru, mtu = _synth.memories_for(uts.cpu().numpy(), case.users[sl])
    ref_u = ref_model.forward_device(feed[0], feed[1], [torch.from_numpy(m).to(dev) for m in mhu],
                                     [torch.from_numpy(m).to(dev) for m in mru],
                                     [torch.from_numpy(m).to(dev) for m in mtu]).scores
    for regime in ("dense", "sparse"):
        sh = build(regime)
        got = sh.forward_device(*feed).scores
        assert torch.equal(got, ref), f"rank {rank} {regime}: sharded scores differ from replicated"
        st = sh.table.last_stats
        assert st["mode"] == regime and st["remote"] > 0, st
        sh.enable_pipeline()
        sh.prefetch(0, feed[0], feed[1], feed[2], feed[4])
        for i in range(3):
            out = sh.forward_prefetched(i % 2, *feed)
            sh.prefetch((i + 1) % 2, feed[0], feed[1], feed[2], feed[4])
            assert torch.equal(out.scores, ref), f"rank {rank} {regime}: pipelined step {i} differs"
        # the user_triplet_set feed (relabelled once): same scores as the per-pair arrays of the same users
        sh.set_user_triplet_set(uts)
        sh.model.group_min_pairs_per_user = 0
        got_u = sh.forward_users(feed[0], feed[1]).scores
        assert torch.allclose(got_u, ref_u, rtol=1e-5, atol=1e-6), f"rank {rank} {regime}: forward_users differs"
        torch.cuda.synchronize()
    # entity-table mode on top of the sharded table: the per-entity tables must be rebuilt whenever an
    # exchange refills the working table (tracked through the tensor version counter)
    sh = build("dense", hoist=True)
    got = sh.forward_device(*feed).scores
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6), f"rank {rank}: hoisted sharded scores differ"
    sh.table.local.mul_(1.5)
    sh.table.refresh()
    scaled = dict(params, entity_emb_matrix=params["entity_emb_matrix"] * np.float32(1.5))
    ref2 = mk(scaled).forward_device(*feed).scores
    got2 = sh.forward_device(*feed).scores
    assert not torch.allclose(ref2, ref)
    assert torch.allclose(got2, ref2, rtol=1e-5, atol=1e-6), f"rank {rank}: stale entity tables after an exchange"
    # sparse regime + entity-table mode over TWO different batches: the row movers write the working table through raw
    # pointers (no torch version bump), so the derived tables must be dropped by the exchange itself
    sl2 = slice(((rank + 1) % world) * Bl, ((rank + 1) % world + 1) * Bl)
    feed2 = (torch.from_numpy(case.users[sl2]).to(dev), torch.from_numpy(case.items[sl2]).to(dev),
             [torch.from_numpy(np.ascontiguousarray(m[sl2])).to(dev) for m in case.memories_h],
             [torch.from_numpy(np.ascontiguousarray(m[sl2])).to(dev) for m in case.memories_r],
             [torch.from_numpy(np.ascontiguousarray(m[sl2])).to(dev) for m in case.memories_t])
    ref_b = ref_model.forward_device(*feed2).scores
    assert not torch.equal(ref_b, ref)
    sh = build("sparse", hoist=True)
    for i, (fd, want) in enumerate(((feed, ref), (feed2, ref_b), (feed, ref))):
        got = sh.forward_device(*fd).scores
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), f"rank {rank}: sparse + entity tables, batch {i}: stale tables"
    sh.enable_pipeline()
    seq = [(feed, ref), (feed2, ref_b), (feed2, ref_b), (feed, ref)]
    sh.prefetch(0, seq[0][0][0], seq[0][0][1], seq[0][0][2], seq[0][0][4])
    for i, (fd, want) in enumerate(seq):
        out = sh.forward_prefetched(i % 2, *fd)
        if i + 1 < len(seq):
            nx = seq[i + 1][0]
            sh.prefetch((i + 1) % 2, nx[0], nx[1], nx[2], nx[4])
        assert torch.allclose(out.scores, want, rtol=1e-5, atol=1e-6), f"rank {rank}: pipelined sparse + entity tables, step {i}"
    # dense regime + pipeline: the two working tables hold the same content, so the entity tables are built once
    sh = build("dense", hoist=True)
    sh.enable_pipeline()
    sh.prefetch(0, feed[0], feed[1], feed[2], feed[4])
    built = []
    for i in range(3):
        out = sh.forward_prefetched(i % 2, *feed)
        sh.prefetch((i + 1) % 2, feed[0], feed[1], feed[2], feed[4])
        built.append(sh.model._hoisted)
        assert torch.allclose(out.scores, ref, rtol=1e-5, atol=1e-6)
    assert built[0] is built[1] is built[2], "dense regime: entity tables rebuilt although the table content is unchanged"
    # data-parallel training: each rank steps on its slice, one all-reduce of the flat gradient buffer;
    # after 3 steps every rank holds the parameters of a single-process trainer fed the whole batch
    from mvin_amd.training import Trainer
    targs = make_args(**dict(vars(args), lr=5e-3, l2_weight=1e-4, l2_agg_weight=1e-5))
    lab = (np.arange(Bl * world) % 3 == 0).astype(np.float32)
    mkt = lambda: MVIN(targs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                       params=params, device=dev)
    whole, part = mkt(), mkt()
    t_whole, t_part = Trainer(whole), Trainer(part, world=world)
    full_feed = (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
                 torch.from_numpy(lab).to(dev),
                 [torch.from_numpy(m_).to(dev) for m_ in case.memories_h],
                 [torch.from_numpy(m_).to(dev) for m_ in case.memories_r],
                 [torch.from_numpy(m_).to(dev) for m_ in case.memories_t])
    my_feed = (feed[0], feed[1], torch.from_numpy(lab[sl]).to(dev), feed[2], feed[3], feed[4])
    for step in range(3):
        lw = t_whole.step(*full_feed)
        lp = t_part.step(*my_feed)
        assert abs(lw - lp) <= 1e-5 * abs(lw) + 1e-6, f"rank {rank} step {step}: loss {lp} vs {lw}"
    for k, v in t_whole.params.items():
        assert torch.allclose(t_part.params[k], v, rtol=1e-4, atol=1e-6), f"rank {rank}: parameter {k} diverged"
    mine = torch.cat([v.reshape(-1) for v in t_part.params.values()]).double().sum().cpu().reshape(1)
    sums2 = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums2, mine)
    assert len({float(x) for x in sums2}) == 1, "parameters differ between ranks"
    # ranks really scored different pairs: gather a checksum of every rank's slice
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, ref.double().sum().cpu().reshape(1))
    assert len({float(x) for x in sums}) == world
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
