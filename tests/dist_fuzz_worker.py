"""Worker of tests/test_gpu_dist.py::test_sharded_configuration_fuzz: WORLD ranks time-sharing cuda:0 over gloo score
seeded random configurations through ShardedMVIN (cyclic row ownership, relabelled id space, both exchange regimes, the
two-stream pipeline, per-pair and user_triplet_set feeds, entity-table mode, bf16 shards) and compare their slice of
the batch with a replicated MVIN holding the full table.  Usage: torchrun ... dist_fuzz_worker.py <n_cases> <offset>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mvin_amd import synth  # noqa: E402
from mvin_amd.config import ABLATIONS, make_args  # noqa: E402
from mvin_amd.dist import ShardedMVIN, shard_rows  # noqa: E402
from mvin_amd.model import MVIN  # noqa: E402
from mvin_amd.params import init_params  # noqa: E402


def draw(i):
    rng = np.random.default_rng(11000 + i)
    D = int(rng.choice([8, 16, 32, 64, 128]))
    K = int(rng.choice([2, 3, 4, 8, 16]))
    H = int(rng.choice([1, 2, 2, 3]))
    M = int(rng.choice([1, 1, 2]))
    while K ** (H * M) > 1024:
        if M > 1:
            M = 1
        elif H > 1:
            H -= 1
        else:
            K = 4
    return dict(D=D, K=K, H=H, M=M, P=int(rng.choice([1, 2, 3])), Nm=int(rng.choice([3, 8, 16, 32])),
                nR=int(rng.choice([2, 7, 12])), Bl=int(rng.choice([1, 5, 24, 48])), n_user=int(rng.choice([2, 9, 50])),
                n_entity=int(rng.choice([17, 101, 1000, 5003])), abl=str(rng.choice(sorted(ABLATIONS))),
                bf16=bool(rng.random() < 0.25 and D % 8 == 0), regime=str(rng.choice(["dense", "sparse", "auto"])),
                feed=str(rng.choice(["pairs", "users", "users_grouped"])), hoist=[False, False, True, "step"][int(rng.integers(0, 4))],
                pipeline=bool(rng.random() < 0.5))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n_cases, offset = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    for i in range(offset, offset + n_cases):
        c = draw(i)
        what = f"rank {rank} case {i} {c}"
        Bl = c["Bl"]
        args = make_args(dim=c["D"], neighbor_sample_size=c["K"], h_hop=c["H"], n_mix_hop=c["M"], p_hop=c["P"], n_memory=c["Nm"],
                         batch_size=Bl, ablation=c["abl"])
        case = synth.small_case(make_args(**dict(vars(args), batch_size=Bl * world)), n_user=c["n_user"], n_entity=c["n_entity"],
                                n_relation=c["nR"], seed=11100 + i, zero_rows=2)
        params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=11200 + i, random_agg_bias=True)
        tdt = "bf16" if c["bf16"] else "f32"
        ref_model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                         device=dev, table_dtype=tdt)
        uts = synth.ripple_sets(case.n_user, case.n_entity, case.n_relation, max(1, c["P"]), c["Nm"], seed=11300 + i)
        sl = slice(rank * Bl, (rank + 1) * Bl)
        users, items = case.users[sl], case.items[sl]
        if c["feed"] == "pairs":
            mem = [[np.ascontiguousarray(m[sl]) for m in lst] for lst in (case.memories_h, case.memories_r, case.memories_t)]
        else:
            mem = synth.memories_for(uts, users)
        u_d, i_d = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
        mem_d = [[torch.from_numpy(x).to(dev) for x in lst] for lst in mem]
        ref = ref_model.forward_device(u_d, i_d, *mem_d).scores
        full = torch.from_numpy(params["entity_emb_matrix"])
        if c["bf16"]:
            full = full.to(torch.bfloat16)
        sh = ShardedMVIN.build(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params,
                               shard_rows(full, rank, world), rank, world, device=dev, regime=c["regime"], table_dtype=tdt,
                               hoist=c["hoist"])
        if c["feed"] != "pairs":
            sh.set_user_triplet_set(torch.from_numpy(uts).to(dev))
            sh.model.group_min_pairs_per_user = 0 if c["feed"] == "users_grouped" else 10 ** 9
        feed = (u_d, i_d, None, None, None) if c["feed"] != "pairs" else (u_d, i_d, *mem_d)
        gb = Bl * world
        outs = []
        if c["pipeline"]:
            sh.enable_pipeline()
            sh.prefetch(0, feed[0], feed[1], feed[2], feed[4], global_batch=gb)
            for s in range(3):
                outs.append(sh.forward_prefetched(s % 2, *feed).scores)
                sh.prefetch((s + 1) % 2, feed[0], feed[1], feed[2], feed[4], global_batch=gb)
        else:
            for s in range(2):
                outs.append(sh.forward_device(*feed, global_batch=gb).scores)
        torch.cuda.synchronize()
        for s, got in enumerate(outs):
            err = (got - ref).abs()
            ok = bool((err <= 1e-5 * ref.abs() + 2e-6).all())
            assert ok, f"{what}: step {s}: max abs err {float(err.max()):.3e}"
        # every rank must have taken the same regime (they issue different collectives)
        mode = sh.table.last_stats.get("mode")
        modes = [None] * world
        dist.all_gather_object(modes, mode)
        assert len(set(modes)) == 1, f"{what}: regimes differ across ranks: {modes}"
        del sh, ref_model
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok ({n_cases} cases)", flush=True)


if __name__ == "__main__":
    main()
