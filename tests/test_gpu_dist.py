"""-m gpu: the row-sharded entity table on a real GPU (single rank): HIP owner-side gather,
RCCL all-to-all collectives at world size 1, and identical scores to the replicated model."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from mvin_amd import synth
from mvin_amd.config import make_args
from mvin_amd.params import init_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("regime", ["sparse", "dense"])
@pytest.mark.parametrize("collective", [False, True])
def test_sharded_scores_equal_replicated(collective, regime, hip_lib, nccl_world1):
    from mvin_amd.dist import ShardedMVIN
    from mvin_amd.model import MVIN
    args = make_args(dim=32, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=16, batch_size=64)
    case = synth.small_case(args, n_user=50, n_entity=5000, n_relation=7, seed=51)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=52, random_agg_bias=True)
    dev = torch.device("cuda:0")
    ref_model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                     params=params, device=dev)
    sh = ShardedMVIN.build(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                           params, torch.from_numpy(params["entity_emb_matrix"]), 0, 1, device=dev,
                           always_collective=collective, regime=regime)
    feed = (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
            [torch.from_numpy(m).to(dev) for m in case.memories_h],
            [torch.from_numpy(m).to(dev) for m in case.memories_r],
            [torch.from_numpy(m).to(dev) for m in case.memories_t])
    got = sh.forward_device(*feed)
    ref = ref_model.forward_device(*feed)
    assert torch.equal(got.scores, ref.scores)
    st = sh.table.last_stats
    assert st["mode"] == regime
    if regime == "sparse":
        assert 0 < st["requested"] < case.n_entity      # only the touched rows were fetched
        untouched = ~sh.needed(feed[1], list(feed[2]) + list(feed[4]))
        assert not sh.table.work[:case.n_entity][untouched].any()
    # pipelined form (exchange on a side stream into a second working table): same scores
    sh.enable_pipeline()
    sh.prefetch(1, feed[0], feed[1], feed[2], feed[4])
    got2 = sh.forward_prefetched(1, *feed)
    torch.cuda.synchronize()
    assert torch.equal(got2.scores, ref.scores)


def test_sparse_regime_entity_tables_follow_the_exchange(hip_lib, nccl_world1):
    """the sparse exchange fills the working table through raw-pointer HIP kernels (no torch version
    bump); entity-table mode (hoist=True) must still rebuild its derived tables for every batch's row set."""
    from mvin_amd.dist import ShardedMVIN
    from mvin_amd.model import MVIN
    args = make_args(dim=32, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=2, n_memory=16, batch_size=32)
    case = synth.small_case(make_args(**dict(vars(args), batch_size=64)), n_user=50, n_entity=5000, n_relation=7, seed=53)
    params = init_params(args, case.n_user, case.n_entity, case.n_relation, seed=54, random_agg_bias=True)
    dev = torch.device("cuda:0")
    ref_model = MVIN(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                     params=params, device=dev)
    sh = ShardedMVIN.build(args, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                           params, torch.from_numpy(params["entity_emb_matrix"]), 0, 1, device=dev,
                           always_collective=True, regime="sparse", hoist=True)

    def feed(sl):
        return (torch.from_numpy(case.users[sl]).to(dev), torch.from_numpy(case.items[sl]).to(dev),
                [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_h],
                [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_r],
                [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_t])
    fa, fb = feed(slice(0, 32)), feed(slice(32, 64))
    ra, rb = ref_model.forward_device(*fa).scores, ref_model.forward_device(*fb).scores
    for i, (fd, want) in enumerate(((fa, ra), (fb, rb), (fa, ra))):
        got = sh.forward_device(*fd).scores
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), f"batch {i}: stale entity tables"
        assert sh.table.last_stats["mode"] == "sparse"


def _run_ranks(argv, world, extra_env=None, timeout=600):
    import subprocess
    import sys
    env = dict(os.environ, MVIN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", "29561"] + argv
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_time_sharing_one_gpu(world, hip_lib):
    """world_size 2 and 3 on the HIP path: the processes time-share cuda:0 (gloo moves the device rows;
    RCCL needs one device per rank).  Both regimes + the pipeline, scores bit-equal to replicated; with three
    ranks the 5 003-row table does not divide (padding rows in the last shard positions) and W is odd."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run_ranks([os.path.join(root, "tests", "dist_gpu_worker.py")], world)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert all(f"rank {k} ok" in r.stdout for k in range(world))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_configuration_fuzz(world, hip_lib):
    """Seeded random configurations (dims, fan-outs, depths, ablation presets, table sizes that do not divide, bf16
    shards, both exchange regimes and the static rule, pipeline, the three feed forms, entity-table mode) scored through
    ShardedMVIN by ``world`` ranks time-sharing the GPU against a replicated model: tests/dist_fuzz_worker.py.
    MVIN_DIST_FUZZ_CASES / MVIN_DIST_FUZZ_OFFSET size a longer campaign."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = os.environ.get("MVIN_DIST_FUZZ_CASES", "10")
    off = os.environ.get("MVIN_DIST_FUZZ_OFFSET", "0")
    r = _run_ranks([os.path.join(root, "tests", "dist_fuzz_worker.py"), n, off], world, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert all(f"rank {k} ok" in r.stdout for k in range(world))


def test_bench_two_ranks_one_gpu(hip_lib):
    """bench.py's N>1 branch end to end (torchrun launch line of the driver, gloo transport)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run_ranks([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                    "--batch", "4096"], 2)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["config"]["pairs_per_gpu_per_step"] == 2048
    assert "row-sharded" in rec["config"]["parallelism"] and rec["roofline"]["achieved"] > 0


def test_bench_gpus_flag_self_launch_one_gpu(hip_lib):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py starts the two ranks itself (the form the driver uses
    at N = 1 with another N would otherwise print a one-rank line) and the line says who launched and how many answered."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MVIN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "4096"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = rec["distributed"]
    assert rec["n_gpus"] == 2 and d["ranks_launched"] == 2 and d["launcher"] == "self" and d["backend"] == "gloo"
    assert d["production_transport"] is False and d["exchange_bytes_received_per_rank_per_step"] > 0
    # over RCCL two ranks need two GPUs: on a smaller box the command fails instead of printing a one-GPU line
    if torch.cuda.device_count() < 2:
        env.pop("MVIN_DIST_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                            "--batch", "4096"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]


def test_row_movers_fp32_and_bf16(hip_lib):
    from mvin_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for dt in (torch.float32, torch.bfloat16):
        table = torch.rand((1000, 64), device=dev, generator=g).to(dt)
        ids = torch.randperm(1000, device=dev, generator=g)[:300].to(torch.int32)
        rows = ops.gather_rows(table, ids)
        assert torch.equal(rows, table[ids.long()])
        dst = torch.zeros_like(table)
        ops.scatter_rows(dst, ids, rows)
        assert torch.equal(dst[ids.long()], rows) and int((dst != 0).any(dim=1).sum()) <= 300


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL wants one device per rank")
def test_bench_two_ranks_over_rccl(hip_lib):
    """bench.py --gpus 2 under the driver's torchrun line with the PRODUCTION transport (RCCL): self-skips on the
    1-GPU boxes, so that the scaling run is not the first time RCCL sees two ranks wherever two GPUs exist."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run_ranks([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8192"],
                   2, extra_env={"MVIN_DIST_BACKEND": "nccl", "MVIN_DIST_CHECK": "1"})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "row-sharded" in rec["config"]["parallelism"]


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("world,n_entity,n", [(8, 113487, 70001), (2, 5000, 1), (3, 17, 300), (1, 100, 64)])
def test_shard_space_ids_kernel_is_the_definition(world, n_entity, n, dtype, hip_lib):
    """mvin_shard_space_ids == dist.to_shard_space (pi(x) = (x mod W) * n_local + x div W), and from_shard_space undoes it."""
    from mvin_amd import ops
    from mvin_amd.dist import from_shard_space, n_local_rows, to_shard_space
    rng = np.random.default_rng(world + n)
    x = rng.integers(0, n_entity, n)
    want = (x % world) * n_local_rows(n_entity, world) + x // world
    got = ops.shard_space_ids(torch.from_numpy(x).to("cuda:0").to(dtype), world, n_local_rows(n_entity, world))
    assert got.dtype == dtype
    np.testing.assert_array_equal(got.cpu().numpy().astype(np.int64), want)
    np.testing.assert_array_equal(to_shard_space(x, n_entity, world), want)
    np.testing.assert_array_equal(from_shard_space(want, n_entity, world), x)
