#!/usr/bin/env python3
"""Benchmark of the MVIN scoring hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the scoring path (MVIN.get_scores semantics: key addressing ->
id expansion -> projection -> K-hop attention aggregation -> mix-hop combiner -> score)
over one batch of synthetic (user, item, ripple-set) inputs already resident in HBM.
Default workload = BASELINE.json's metric config C3: last-fm_50core-shaped tables,
dim=64, hop=2 (n_mix_hop=1), fan-out=32, fp32.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field):
  value     = pairs scored per second, whole job (all ranks), max-over-ranks time
  roofline  = the dominant kernel (gather_attn_l2_kernel, mvin_gather_attn_l2_fwd) in its HBM-BOUND
              regime: the same kernel at the same D / K on a 4 GB entity table (16 M rows, uniform
              adjacency, far larger than the 256 MB Infinity Cache), algorithmic bytes per launch
              (SURVEY.md 8(d) bytes(pair) x pairs per launch) / mean launch duration measured live
              with HIP events on the launch stream, against the 8 TB/s HBM peak (frac <= 1)
    .timed_region = the same kernel as launched inside the timed steps on the metric workload, whose
              27 MB table is L2 / Infinity-Cache resident: labelled bound "l2", algorithmic GB/s
              against the 34.5 TB/s L2 aggregate, plus the measured HBM fraction from PMC traffic
  batch_sweep  = whole-path pairs/s at the reference's own batch sizes (512, 4096,
              16384), eager and as one hipGraph replay
  cpu_baseline = oracle/mirror_fp32.py (TF-graph-equivalent torch-CPU restatement) timed on
              this host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth (8 XCDs x 4 MiB), same guide, "L2 (per XCD)"
HBM_LEG_ROWS = 16_000_000   # x 64 x 4 B = 4.1 GB: 16x the Infinity Cache, < 4 GiB (32-bit buffer offsets)
HBM_LEG_PAIRS = 32768
# projected-tables form: three tables of 4 M x 64 x 4 B = 1.02 GB each (below the kernel's 1 GiB-per-table limit), 3.07 GB = 12x the
# Infinity Cache
HBM_LEG_PRJ_ROWS = 4_000_000


def prj_bytes_per_pair(D, K, s=4):
    """What mvin_gather_attn_l2_prj_fwd must read per pair of a depth-2 tree: SURVEY.md 8(d)'s figure with TWO self rows per child
    (T1[x], TA1[x]) instead of one -- (1 + 2K + K^2) rows, the adjacency rows of the item and its K children, the query row, the score."""
    return (1 + 2 * K + K * K) * D * s + (1 + K) * K * 2 * 4 + D * s + 4


def algorithmic_bytes_per_pair(D, K, L, s=4):
    """T*D*s + (T-K^L)*K*2*4 + D*s + 4, T = sum_{e<=L} K^e."""
    T = sum(K ** e for e in range(L + 1))
    return T * D * s + (T - K ** L) * K * 2 * 4 + D * s + 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dataset", default="last-fm_50core")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--hop", type=int, default=2)
    ap.add_argument("--mix", type=int, default=1)
    ap.add_argument("--fanout", type=int, default=32)
    ap.add_argument("--batch", type=int, default=524288,
                    help="TOTAL pairs per step over all ranks (strong scaling: each rank scores batch/N)")
    ap.add_argument("--shard", choices=["auto", "rowshard", "replicate"], default="auto",
                    help="entity table placement: row-sharded over the ranks with all-to-all row fetch "
                         "(auto: when N>1) or replicated")
    ap.add_argument("--table-dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the entity table (arithmetic is fp32 either way; bf16 = BASELINE config C5)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the scoring pass as one hipGraph (for launch-bound batch sizes; single GPU)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="row-sharded mode: serialise the row exchange and the scoring (default: the exchange "
                         "for step i+1 runs on a side stream while step i is scored)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="with --shard rowshard on one GPU: still run the RCCL all-to-alls (world size 1)")
    ap.add_argument("--hoist", choices=["off", "cached", "step"], default="off",
                    help="entity-table mode of the two deepest levels (MVIN.hoist_entity_tables; SURVEY 7.3-c route "
                         "2b): 'cached' builds the per-entity tables once (weights frozen, e.g. top-K eval), 'step' "
                         "rebuilds them inside every timed step.  A separate mode with its own bytes per pair -- "
                         "never the headline number")
    ap.add_argument("--adj", choices=["kg", "uniform"], default="kg")
    ap.add_argument("--items", choices=["zipf", "uniform"], default="zipf")
    ap.add_argument("--cpu-batch", type=int, default=512)
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the 4 GB-table launch of the dominant kernel")
    ap.add_argument("--no-probe", action="store_true", help="skip the gather-only probe of the timed region's access stream")
    ap.add_argument("--hbm-leg-only", action="store_true",
                    help="run ONLY the 4 GB-table leg and print its record (the command profiles/r2/*hbm_leg* were "
                         "collected with: one kernel name, one regime per rocprofv3 run)")
    ap.add_argument("--hbm-leg-form", choices=["enc", "plain", "prj"], default="enc",
                    help="with --hbm-leg-only: enc = mvin_gather_attn_l2_enc_fwd (16 M-row table), plain = mvin_gather_attn_l2_fwd, "
                         "prj = mvin_gather_attn_l2_prj_fwd (projected tables of a 4 M-row table: the instance the default line times)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep")
    ap.add_argument("--feed", choices=["pairs", "users"], default="users",
                    help="pairs: per-pair ripple-set arrays [B, n_memory] resident in HBM (the reference's feed_dict "
                         "contents); users: user_triplet_set resident, pairs grouped by user inside key addressing")
    ap.add_argument("--sweep", default="512,4096,16384")
    ap.add_argument("--n-entity", type=int, default=0,
                    help="override the entity count (synthetic large-table variant, e.g. 16000000 = 4 GB at dim 64; "
                         "implies --adj uniform)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="builder-side projection on ONE GPU: score rank 0's share of the global batch as one of W ranks "
                         "would (user-sorted split, row-shard exchange of the whole table per step); the line is marked "
                         "`emulated_world` and its `value` is per-rank pairs/s x W -- never a measurement of W GPUs")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent steps are enqueued round-robin on this many HIP streams (single GPU, tables replicated, no "
                         "hipGraph replay): the drain of one step's kernels overlaps the ramp of the next one's; 1 = one stream; "
                         "0 (default) = what mvin_amd.harness.ctr_eval_device takes for this batch size (harness.eval_streams: two for "
                         "the single-launch pass, three for the multi-launch schedule)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ablation", default="all",
                    help="the reference's --ablation preset (parameter_ablation.py); anything but 'all' is a different workload, "
                         "named in config.ablation")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: launch / join the --gpus N ranks, all-reduce one number through the process group, "
                         "print {\"launch_check\": ...} on rank 0 and exit before any GPU work (argument plumbing test)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher --
    it re-executes the same command line under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1 at a
    free port) exactly as the driver's own line does, streams the ranks' output through and exits with their code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MVIN_BENCH_LAUNCHER="self")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(args, margs, case, params):
    """Time the op-by-op torch-CPU mirror of the TF graph on a bounded sample.  The thread count is
    the fastest of a short probe over {all cores, 64, 32, 16, 8} (all cores is far from the fastest
    on a 256-core host: the graph is many small ops); ``cores`` reports the threads actually used."""
    import torch
    from oracle import mirror_fp32
    from mvin_amd.config import make_args
    ncpu = os.cpu_count() or 1
    pt = mirror_fp32.as_torch_params(params)

    def feed_of(n):
        sl = slice(0, n)
        return (case.users[sl], case.items[sl], [m[sl] for m in case.memories_h],
                [m[sl] for m in case.memories_r], [m[sl] for m in case.memories_t])

    def run(n):
        cargs = make_args(**dict(vars(margs), batch_size=n))
        t0 = time.perf_counter()
        ref = mirror_fp32.forward(cargs, pt, case.adj_entity, case.adj_relation, *feed_of(n))
        return time.perf_counter() - t0, ref

    probe_n = min(32, len(case.users))
    best_thr, best_t = None, None
    for thr in sorted({t for t in (ncpu, 64, 32, 16, 8) if 1 <= t <= ncpu}, reverse=True):
        torch.set_num_threads(thr)
        run(probe_n)
        t, _ = run(probe_n)
        if best_t is None or t < best_t:
            best_thr, best_t = thr, t
    torch.set_num_threads(best_thr)
    Bc = min(args.cpu_batch, len(case.users))
    for _ in range(args.cpu_warmup):
        run(Bc)
    times = []
    for _ in range(args.cpu_iters):
        t, ref = run(Bc)
        times.append(t)
    med = float(np.median(times))
    return {"value": Bc / med, "unit": "pairs/s", "cores": best_thr, "kind": "port",
            "sample": f"{args.cpu_warmup} warm-ups + {args.cpu_iters} timed passes (median) of {Bc} pairs of the same workload, "
                      f"torch-CPU fp32 op-by-op mirror of the TF graph (oracle/mirror_fp32.py); "
                      f"{best_thr} threads = fastest of a probe over thread counts on a {ncpu}-core host"}, ref, Bc


def l2_reference_f64(table, adj_e, adj_r, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, K):
    """The outputs of mvin_gather_attn_l2_fwd (include/mvin_hip.h; reference model.py:295-305 with aggregators.py:98-146)
    for the listed parents, evaluated with plain torch indexing in float64 on the tensors' device: the checker of the
    roofline launches (parents_per_pair = 1: q[i] belongs to parents[i]).  Returns (nagg0, nagg1) [n, D] float64."""
    import torch
    f = lambda t: t.double()
    par = parents.long()
    x1 = adj_e[par].long()                                    # [n, K]
    y = adj_e[x1].long()                                      # [n, K, K]
    p = torch.softmax(f(t0)[adj_r[x1].long()], dim=-1)        # [n, K, K]
    E = table
    c1 = f(q) @ f(W1) + f(b1)
    c2 = f(q) @ f(W2) + f(b2)
    self1 = f(E[x1]) @ f(W1) + c1[:, None, :]
    S = (p[..., None] * f(E[y])).sum(2)                       # [n, K, D]
    Z = self1 + (S @ f(W2) + p.sum(-1, keepdim=True) * c2[:, None, :]) / K
    out1 = torch.relu(Z @ f(A0) + f(a0))
    p0 = torch.softmax(f(t0)[adj_r[par].long()], dim=-1)
    p1 = torch.softmax(f(t1)[adj_r[par].long()], dim=-1)
    return (p0[..., None] * self1).sum(1) / K, (p1[..., None] * out1).sum(1) / K


def check_l2_launch(out, table, adj_e, adj_r, parents, t0, t1, W1, W2, b1, b2, q, A0, a0, K, n_check=256, seed=0):
    """Compare nagg0 / nagg1 of ``n_check`` parents sampled across a launch with l2_reference_f64:
    |delta| <= 1e-5 |ref| + 1e-6 (BASELINE.json north_star; fp32 path).  -> (max_abs_err, worst err / bound, ok)."""
    import torch
    n = parents.shape[0]
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    idx = torch.randperm(n, generator=g)[:min(n_check, n)].to(parents.device)
    r0, r1 = l2_reference_f64(table, adj_e, adj_r, parents[idx], t0, t1, W1, W2, b1, b2, q[idx], A0, a0, K)
    worst, max_abs = 0.0, 0.0
    for got, ref in ((out[0][idx].double(), r0), (out[1][idx].double(), r1)):
        err = (got - ref).abs()
        bound = 1e-5 * ref.abs() + 1e-6
        worst = max(worst, float((err / bound).max()))
        max_abs = max(max_abs, float(err.max()))
    return max_abs, worst, worst <= 1.0


def hbm_leg(dev, D, K, table_dtype, iters=10, warmup=3, n_rows=None, pairs=HBM_LEG_PAIRS, n_rel=9, seed=0, encoded=False, prj=False):
    """The dominant kernel in its HBM-bound regime: the SAME kernel as in the timed steps (same D, K, table dtype,
    projection + attention on) -- mvin_gather_attn_l2_fwd (role-split kernel), or with ``encoded`` mvin_gather_attn_l2_enc_fwd
    (packed-tile kernel over the duplicate-slot encoding of the same adjacency) -- on a ``n_rows`` x D table far larger than the
    Infinity Cache, uniform adjacency (no repeated slots: every one of the K + K^2 rows per pair is loaded), ``pairs`` parents
    per launch.  ``prj``: the PROJECTED-TABLES instance (mvin_gather_attn_l2_prj_fwd over E.W1 | E.W1.A0 | E.W2.A0, built once
    before the launches by mvin_project_tables; 4 M rows: 3 x 1.02 GB of tables), priced on ITS bytes (prj_bytes_per_pair).
    Each launch is bracketed by HIP events on the stream it is launched on (torch's current stream)."""
    import torch
    from mvin_amd import ops
    if n_rows is None:
        n_rows = HBM_LEG_PRJ_ROWS if prj else HBM_LEG_ROWS
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    dt = torch.bfloat16 if table_dtype == "bf16" else torch.float32
    table = (torch.rand((n_rows, D), device=dev, generator=g) - 0.5).to(dt)
    adj_e = torch.randint(0, n_rows, (n_rows, K), device=dev, generator=g, dtype=torch.int32)
    adj_r = torch.randint(0, n_rel, (n_rows, K), device=dev, generator=g, dtype=torch.int32)
    parents = torch.randint(0, n_rows, (pairs,), device=dev, generator=g, dtype=torch.int32)
    t0 = torch.rand(n_rel, device=dev, generator=g)
    W = (torch.rand((3, D, D), device=dev, generator=g) - 0.5) / D ** 0.5
    q = torch.rand((pairs, D), device=dev, generator=g)
    bias = torch.zeros(D, device=dev)
    args = (table, adj_e, adj_r, parents, t0, t0, W[0], W[1], bias, bias, q, W[2], bias, pairs, 1, K, D, n_rel)
    if prj:
        if table_dtype != "f32":
            raise ValueError("projected tables are fp32")
        encoded = ops.gather_attn_l2_prj_supported(D, K, True, n_rows, n_rel)
        if not encoded and not ops.gather_attn_l2_prj_supported(D, K, False, n_rows, n_rel):
            raise ValueError(f"no projected-tables kernel for D={D} K={K}")
        ws = ops.project_tables(table, W[0], W[1], bias, bias, W[2], bias, K, True)
        if encoded:
            enc_e, enc_r, cnt = ops.encode_adjacency(adj_e, adj_r)
            launch = lambda: ops.gather_attn_l2_prj(ws, enc_e, enc_r, parents, t0, t0, q, pairs, 1, K, D, n_rel, n_rows, encoded=True)   # noqa: E731
            kernel = (f"gather_attn_l2_wpp_kernel<{K}> (mvin_gather_attn_l2_prj_fwd: wave-per-parent kernel over projected tables)"
                      if ops.gather_attn_l2_wpp_supported(D, K) else
                      f"gather_attn_l2_packed_kernel<{D}, {K}, false, 4, false, true> (mvin_gather_attn_l2_prj_fwd)")
        else:
            launch = lambda: ops.gather_attn_l2_prj(ws, adj_e, adj_r, parents, t0, t0, q, pairs, 1, K, D, n_rel, n_rows, encoded=False)  # noqa: E731
            kernel = f"gather_attn_l2_d32_kernel<{K}, false, false, true> (mvin_gather_attn_l2_prj_fwd, plain adjacency)"
    elif encoded:
        enc_e, enc_r, cnt = ops.encode_adjacency(adj_e, adj_r)
        launch = lambda: ops.gather_attn_l2_enc(table, enc_e, enc_r, *args[3:])       # noqa: E731
        kernel = (f"gather_attn_l2_packed_kernel<{D}, {K}, {'true' if table_dtype == 'bf16' else 'false'}, 4, false, false> "
                  "(mvin_gather_attn_l2_enc_fwd)")
    else:
        launch = lambda: ops.gather_attn_l2(*args)[:2]                                # noqa: E731
        kernel = "gather_attn_l2_split_kernel (mvin_gather_attn_l2_fwd)"
    for _ in range(warmup):
        launch()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = launch()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    finite = bool(torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all())
    # the launch the roofline is quoted on is CHECKED: 256 parents sampled across it against a float64 evaluation
    max_abs, worst, ok = check_l2_launch(out, *args[:13], K)
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    s_ = 2 if table_dtype == "bf16" else 4
    bpp = prj_bytes_per_pair(D, K, s=s_) if prj else algorithmic_bytes_per_pair(D, K, 2, s=s_)
    del table, adj_e, adj_r
    if encoded:
        del enc_e, enc_r, cnt
    if prj:
        del ws
    torch.cuda.empty_cache()
    return {"kernel": kernel, "form": "prj" if prj else "enc" if encoded else "plain",
            "avg_launch_ms": ms, "pairs_per_launch": pairs, "bytes_per_pair": bpp,
            "bytes_per_pair_formula": ("(1 + 2K + K^2) rows (two self rows per child: T1[x], TA1[x]) + (1 + K) K 8 B of adjacency + query + score"
                                       if prj else "SURVEY.md 8(d): (1 + K + K^2) rows + (1 + K) K 8 B of adjacency + query + score"),
            "achieved": bpp * pairs / (ms * 1e-3) / 1e9, "table_rows": n_rows,
            "table_bytes": (3 if prj else 1) * n_rows * D * s_,
            "launches": iters, "outputs_finite": finite, "max_abs_err": max_abs, "worst_err_over_bound": worst,
            "verified": bool(ok and finite),
            "verified_how": "nagg0 / nagg1 of 256 parents sampled across the launch vs a torch float64 evaluation on the GPU, "
                            "|delta| <= 1e-5 |ref| + 1e-6"}


PROBE_PAIRS = 131072


def gather_probe(model, items, K, D, s_, iters=10, warmup=3):
    """A reference point for the timed region's kernel: mvin_probe_gather_l2 reads exactly the table rows
    mvin_gather_attn_l2_fwd must gather for these pairs (same entity table, same adjacency; the first
    PROBE_PAIRS pairs of the batch) with the same 16-byte lane loads, ids one round ahead, and only sums them.
    Timed with HIP events after the timed region; its bytes are the gathered rows alone."""
    import torch
    from mvin_amd import ops
    n = min(items.shape[0], PROBE_PAIRS)
    ents, _ = model.get_neighbors(items[:n])                   # mvin_expand_ids: levels 0..2 of these pairs
    l1, l2 = ents[1].contiguous().view(-1), ents[2].contiguous().view(-1)
    sums = torch.empty(n, dtype=torch.float32, device=items.device)
    for _ in range(warmup):
        ops.probe_gather_l2(model.entity_emb_matrix, l1, l2, K, sums)
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.probe_gather_l2(model.entity_emb_matrix, l1, l2, K, sums)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = float(np.mean([x.elapsed_time(y) for x, y in evs]))
    row_bytes = (K + K * K) * D * s_
    return {"avg_launch_ms": ms, "pairs_per_launch": n, "row_bytes_per_pair": row_bytes,
            "achieved": row_bytes * n / (ms * 1e-3) / 1e9, "unit": "GB/s", "launches": iters,
            "checksum_finite": bool(torch.isfinite(sums).all())}


def pmc_record(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def get_lanes(dev, n=2):
    """The process's scoring streams (mvin_amd.graph.scoring_streams: created once -- streams that share a hardware queue serialise)."""
    from mvin_amd.graph import scoring_streams
    return scoring_streams(dev, n)


def batch_sweep(model, users, items, mh, mr, mt, sizes, steps=30, warmup=5, uts=None):
    """Whole get_scores path at the reference's own batch sizes: per-pair feeds (the contents of the
    reference's feed_dict) as eager launches and replayed as one hipGraph (mvin_amd.graph.GraphedScorer), and -- ``uts``
    given -- the users feed (user_triplet_set resident, MVIN.forward_users) as eager launches.  `*_two_streams`: the same
    back-to-back batches enqueued round-robin on two HIP streams, as mvin_amd.harness.ctr_eval_device scores the batches of an
    evaluation: a pass at these sizes is a dependent chain of loads, two passes in flight overlap their waits."""
    import torch
    from mvin_amd.graph import GraphedScorer
    out = []
    steps = max(steps, 60)
    for B in sizes:
        if B > users.shape[0]:
            continue
        sl = slice(0, B)
        feed = (users[sl].contiguous(), items[sl].contiguous(), [m[sl].contiguous() for m in mh],
                [m[sl].contiguous() for m in mr], [m[sl].contiguous() for m in mt])
        rec = {"batch": B}
        lanes = get_lanes(users.device, 2)
        for ln in lanes:
            ln.wait_stream(torch.cuda.current_stream())
        for mode in ("eager", "eager_two_streams", "hipgraph", "users_feed", "users_feed_two_streams"):
            if mode.endswith("two_streams"):
                if mode.startswith("users") and uts is None:
                    continue
                one = ((lambda: model.forward_users(feed[0], feed[1], uts)) if mode.startswith("users")
                       else (lambda: model.forward_device(*feed)))
                turn = {"i": 0}

                def fn(one=one, turn=turn):          # independent batches round-robin on two streams (harness.ctr_eval_device)
                    turn["i"] ^= 1
                    with torch.cuda.stream(lanes[turn["i"]]):
                        return one()
            elif mode == "hipgraph":
                if B <= getattr(model, "small_max_batch", 0):
                    # the pass is ONE kernel launch at this size: a graph launch costs more than a kernel launch on this runtime
                    # (round 5: 41 vs 36 us at 512 pairs), there is nothing to replay
                    rec["hipgraph_note"] = "not measured: the pass is one kernel launch at this size (mvin_score_small_fwd)"
                    continue
                sc = GraphedScorer(model, B)
                sc.load(*feed)
                fn = sc.replay
            elif mode == "users_feed":
                if uts is None:
                    continue
                fn = lambda: model.forward_users(feed[0], feed[1], uts)   # noqa: E731
            else:
                fn = lambda: model.forward_device(*feed)   # noqa: E731
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            rec[mode] = {"pairs_per_s": B / dt, "us_per_step": 1e6 * dt}
        out.append(rec)
    return out


def shipped_config(a, dev, steps=10, warmup=3):
    """The settings of the reference's own run scripts (src/bash/mvin_last_fm.sh / mvin_movie.sh / mvin_az_book.sh:
    --dim 16 --neighbor_sample_size 8 --h_hop 2 --n_mix_hop 1, batch 512 / 1024) on the same dataset shape: the whole
    get_scores path at a large resident batch (users feed) and at the script's own batch size (per-pair feeds).
    Informational: BASELINE.json quotes the metric at dim 64 / fan-out 32."""
    import torch
    from mvin_amd import ops, synth
    from mvin_amd.config import make_args
    from mvin_amd.model import MVIN
    from mvin_amd.params import init_params
    d = synth.DATASETS[a.dataset]
    Bbig, Bs = 524288, (1024 if a.dataset == "MovieLens-1M" else 512)
    margs = make_args(dataset=a.dataset, dim=16, neighbor_sample_size=8, h_hop=2, n_mix_hop=1, p_hop=d["p_hop"],
                      n_memory=d["n_memory"], batch_size=Bbig)
    case = synth.dataset_case(a.dataset, K=8, B=Bbig, seed=a.seed)
    params = init_params(margs, case.n_user, case.n_entity, case.n_relation, seed=a.seed)
    model = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
    users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    uts = torch.from_numpy(case.user_triplet_set).to(dev)
    mem = [[torch.from_numpy(np.ascontiguousarray(m[:Bs])).to(dev) for m in lst]
           for lst in (case.memories_h, case.memories_r, case.memories_t)]

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    dt_big = timed(lambda: model.forward_users(users, items, uts))
    dt_small = timed(lambda: model.forward_device(users[:Bs], items[:Bs], *mem))
    L, K, D = 2, 8, 16
    return {"workload": f"{a.dataset}-shaped tables, dim=16 hop=2 n_mix_hop=1 fan-out=8 p_hop={d['p_hop']} n_memory={d['n_memory']} "
                        "(src/bash/mvin_*.sh), full get_scores path",
            "fused_kernel_variant": ops.gather_attn_l2_variant(D, K, Bbig, case.n_entity, False),
            "bytes_per_pair": algorithmic_bytes_per_pair(D, K, L),
            "rows": [{"batch": Bbig, "feed": "users", "pairs_per_s": Bbig / dt_big, "ms_per_step": 1e3 * dt_big},
                     {"batch": Bs, "feed": "pairs", "pairs_per_s": Bs / dt_small, "us_per_step": 1e6 * dt_small}]}


def uniform_adjacency_variant(a, dev, steps=5, warmup=2, pairs=262144):
    """"a uniform-random adjacency (randint(0, nE)) is the worst-case-locality variant and should also be
    reported" -- the same tables, dims and feed with adj_entity uniform over the entities (no repeated slots: every one
    of the K + K^2 rows per pair is a distinct load, the plain-adjacency kernel runs) and items uniform over the item range."""
    import torch
    from mvin_amd import synth
    from mvin_amd.config import make_args
    from mvin_amd.model import MVIN
    from mvin_amd.params import init_params
    d = synth.DATASETS[a.dataset]
    Bu = min(a.batch, pairs)
    margs = make_args(dataset=a.dataset, dim=a.dim, neighbor_sample_size=a.fanout, h_hop=a.hop, n_mix_hop=a.mix,
                      p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=Bu)
    case = synth.dataset_case(a.dataset, K=a.fanout, B=Bu, seed=a.seed, zipf=False, uniform_adj=True)
    params = init_params(margs, case.n_user, case.n_entity, case.n_relation, seed=a.seed)
    model = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params,
                 device=dev, table_dtype=a.table_dtype)
    users, items = torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev)
    uts = torch.from_numpy(case.user_triplet_set).to(dev)
    for _ in range(warmup):
        out = model.forward_users(users, items, uts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model.forward_users(users, items, uts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    L = a.hop * a.mix
    bpp = algorithmic_bytes_per_pair(a.dim, a.fanout, L, s=2 if a.table_dtype == "bf16" else 4)
    enc = model._enc_for_l2(n_parents=Bu * a.fanout ** max(L - 2, 0))
    return {"value": Bu / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt, "pairs_per_step": Bu, "steps": steps,
            "bytes_per_pair": bpp, "whole_path_bytes_rate_gbs": Bu / dt * bpp / 1e9,
            "fused_kernel": "encoded (packed-tile)" if enc is not None else "plain adjacency",
            "scores_finite": bool(torch.isfinite(out.scores).all()),
            "note": "uniform-random adjacency + uniform items, users feed, same tables / dims: nothing repeats, so the bytes "
                    "per pair ARE SURVEY 8(d)'s and whole_path_bytes_rate_gbs is a rate of bytes really gathered (from the "
                    "Infinity Cache: the table is cache-resident)"}


def c1_plumbing(a, dev, cpu_threads, warmup=3, iters=10):
    """BASELINE.json configs[0] -- MovieLens-1M-shaped tables, dim 16, ONE hop, fan-out 8, the reference's own batch of
    1 024 (mvin_movie.sh): the configuration SURVEY 8(d) makes mandatory for the CPU column.  The torch-CPU mirror of
    the TF graph and the HIP path score the SAME 1 024 pairs (3 warm-ups + 10 timed passes, median); the scores are compared."""
    import torch
    from oracle import mirror_fp32
    from mvin_amd import synth
    from mvin_amd.config import make_args
    from mvin_amd.model import MVIN
    from mvin_amd.params import init_params
    ds, B, D, K = "MovieLens-1M", 1024, 16, 8
    d = synth.DATASETS[ds]
    margs = make_args(dataset=ds, dim=D, neighbor_sample_size=K, h_hop=1, n_mix_hop=1, p_hop=d["p_hop"],
                      n_memory=d["n_memory"], batch_size=B)
    case = synth.dataset_case(ds, K=K, B=B, seed=a.seed)
    params = init_params(margs, case.n_user, case.n_entity, case.n_relation, seed=a.seed)
    pt = mirror_fp32.as_torch_params(params)
    torch.set_num_threads(cpu_threads)
    feed = (case.users, case.items, case.memories_h, case.memories_r, case.memories_t)
    for _ in range(warmup):
        ref = mirror_fp32.forward(margs, pt, case.adj_entity, case.adj_relation, *feed)
    ct = []
    for _ in range(iters):
        t0 = time.perf_counter()
        ref = mirror_fp32.forward(margs, pt, case.adj_entity, case.adj_relation, *feed)
        ct.append(time.perf_counter() - t0)
    model = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation, params=params, device=dev)
    dfeed = (torch.from_numpy(case.users).to(dev), torch.from_numpy(case.items).to(dev),
             [torch.from_numpy(m).to(dev) for m in case.memories_h], [torch.from_numpy(m).to(dev) for m in case.memories_r],
             [torch.from_numpy(m).to(dev) for m in case.memories_t])
    for _ in range(warmup):
        out = model.forward_device(*dfeed)
    torch.cuda.synchronize()
    gt = []
    for _ in range(iters):
        t0 = time.perf_counter()
        out = model.forward_device(*dfeed)
        torch.cuda.synchronize()
        gt.append(time.perf_counter() - t0)
    want = ref.scores.numpy()
    err = np.abs(out.scores.cpu().numpy() - want)
    return {"workload": f"{ds}-shaped tables (nE={case.n_entity}, nU={case.n_user}, nR={case.n_relation}), dim=16 hop=1 "
                        f"n_mix_hop=1 fan-out=8 p_hop={d['p_hop']} n_memory={d['n_memory']}, batch 1024, per-pair feeds "
                        "(BASELINE.json configs[0])",
            "bytes_per_pair": algorithmic_bytes_per_pair(D, K, 1),
            "cpu": {"pairs_per_s": B / float(np.median(ct)), "ms_per_step": 1e3 * float(np.median(ct)), "cores": cpu_threads,
                    "kind": "port"},
            "gpu": {"pairs_per_s": B / float(np.median(gt)), "us_per_step": 1e6 * float(np.median(gt)),
                    "note": "one synchronised call per step (host launch latency included)"},
            "max_abs_err": float(err.max()),
            "within_1e-5rel_1e-6abs": bool((err <= 1e-5 * np.abs(want) + 1e-6).all())}


def training_steps(margs, case, params, dev, users, items, mems, sizes=(512, 4096), steps=20):
    """ms per optimisation step (mvin_amd/training.py) eager and as one hipGraph replay per step."""
    import torch
    from mvin_amd.model import MVIN
    from mvin_amd.training import GraphedTrainer, Trainer
    out = []
    for B in sizes:
        if B > items.shape[0]:
            continue
        feed = (users[:B].contiguous(), items[:B].contiguous(), (torch.arange(B, device=dev) % 2).to(torch.float32),
                [m[:B].contiguous() for m in mems[0]], [m[:B].contiguous() for m in mems[1]],
                [m[:B].contiguous() for m in mems[2]])
        row = {"batch": B}
        for mode in ("eager", "hipgraph"):
            tm = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                      params=params, device=dev)
            tr = Trainer(tm)
            if mode == "hipgraph":
                gt = GraphedTrainer(tr, B, ids_dtype=users.dtype)
                one = lambda: gt.step(*feed)                               # noqa: E731
            else:
                one = lambda: tr.step(*feed)                               # noqa: E731
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = one()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            row[mode] = {"ms_per_step": 1e3 * dt, "pairs_per_s": B / dt, "loss_after": float(loss)}
            del tr, tm
        out.append(row)
    return {"what": "forward + loss (model.py:378-412) + backward + tf.train.AdamOptimizer rule, all HIP; hipgraph = the "
                    "whole step replayed as one graph (training.GraphedTrainer)", "steps": steps, "rows": out}


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # a line that says n_gpus = world while the caller asked for --gpus N would be read as an N-GPU measurement
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    backend = os.environ.get("MVIN_DIST_BACKEND", "nccl")
    if world > 1 or a.force_collectives:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":       # = RCCL; the production transport
            if a.launch_check:
                raise SystemExit("--launch-check is the no-GPU plumbing test: set MVIN_DIST_BACKEND=gloo")
            if torch.cuda.device_count() < world:
                raise SystemExit(f"bench.py: --gpus {world} over RCCL needs {world} visible GPUs, found "
                                 f"{torch.cuda.device_count()} (tests only: MVIN_DIST_BACKEND=gloo time-shares them)")
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:                       # tests only (explicit MVIN_DIST_BACKEND): N ranks time-sharing the GPUs of a smaller box over gloo
            if not a.launch_check:
                local_rank %= torch.cuda.device_count()
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != a.gpus or dist.get_backend() != backend:
            raise SystemExit(f"bench.py: process group has world size {dist.get_world_size()} / backend {dist.get_backend()}, "
                             f"asked for {a.gpus} / {backend}")
    if a.launch_check:
        one = torch.ones(1, dtype=torch.float64)
        if dist.is_initialized():
            dist.all_reduce(one)
        rec = {"launch_check": {"gpus_asked": a.gpus, "ranks_launched": int(one.item()),
                                "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                                "backend": dist.get_backend() if dist.is_initialized() else None,
                                "launcher": os.environ.get("MVIN_BENCH_LAUNCHER", "external")}}
        if dist.is_initialized():
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(rec), flush=True)
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from mvin_amd import synth
    from mvin_amd.config import make_args
    from mvin_amd.model import MVIN
    from mvin_amd.params import init_params

    if a.hbm_leg_only:
        leg = hbm_leg(dev, a.dim, a.fanout, a.table_dtype, iters=max(a.steps, 1), warmup=a.warmup,
                      encoded=a.hbm_leg_form != "plain" and os.environ.get("MVIN_L2_ENC", "auto") != "0", prj=a.hbm_leg_form == "prj")
        leg.update(bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", frac=leg["achieved"] / HBM_PEAK_GBS)
        print(json.dumps({"hbm_leg_only": leg}), flush=True)
        return

    d = synth.DATASETS[a.dataset]
    split = a.emulate_world if (a.emulate_world > 1 and world == 1) else world     # ranks the global batch is cut into
    if a.batch % split:
        raise SystemExit("--batch must be divisible by the number of ranks")
    Bl = a.batch // split                       # pairs this rank scores per step
    margs = make_args(dataset=a.dataset, dim=a.dim, neighbor_sample_size=a.fanout, h_hop=a.hop,
                      n_mix_hop=a.mix, p_hop=d["p_hop"], n_memory=d["n_memory"], batch_size=Bl, ablation=a.ablation)
    # one global synthetic batch (same seed everywhere); rank r scores pairs [r*Bl, (r+1)*Bl).
    # Pairs are independent: no data-path reduction across ranks.
    if a.n_entity:   # HBM-bound variant: same workload on a table far larger than the Infinity Cache
        synth.DATASETS[a.dataset] = dict(d, n_entity=a.n_entity)
        a.adj = "uniform"
    case = synth.dataset_case(a.dataset, K=a.fanout, B=a.batch, seed=a.seed,
                              zipf=(a.items == "zipf"), uniform_adj=(a.adj == "uniform"))
    if split > 1:
        # Pairs are independent, so ANY partition of the batch is valid; the ranks take contiguous slices of the batch
        # in USER order: a rank then holds all pairs of ~n_user/W users (B/n_user pairs each, as on one GPU) and key
        # addressing keeps reading a user's ripple-set rows once per user instead of once per pair.  Slices stay
        # exactly B/W pairs; at most W-1 users straddle a cut (both neighbours then read that user's rows).
        order = np.argsort(case.users, kind="stable")
        case.users, case.items = case.users[order], case.items[order]
        case.memories_h = [m[order] for m in case.memories_h]
        case.memories_r = [m[order] for m in case.memories_r]
        case.memories_t = [m[order] for m in case.memories_t]
    distinct = -(-case.n_user // split) + 1 if split > 1 else None     # users a rank expects in its slice
    sl = slice(rank * Bl, (rank + 1) * Bl)
    params = init_params(margs, case.n_user, case.n_entity, case.n_relation, seed=a.seed)
    rowshard = a.shard == "rowshard" or (a.shard == "auto" and split > 1)
    if a.emulate_world > 1 and world == 1:
        a.force_collectives = True
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    hoist_kw = {"off": False, "cached": True, "step": "step"}[a.hoist]
    if rowshard:  # the model lives in shard space; its entity table is the sharded table's working copy
        from mvin_amd.dist import ShardedMVIN, shard_rows
        shard = shard_rows(torch.from_numpy(params["entity_emb_matrix"]), rank, world)
        runner = ShardedMVIN.build(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity,
                                   case.adj_relation, params, shard, rank, world, device=dev,
                                   table_dtype=a.table_dtype, hoist=hoist_kw, always_collective=a.force_collectives)
        model = runner.model
    else:
        model = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                     params=params, device=dev, table_dtype=a.table_dtype, hoist=hoist_kw)
        runner = model
    users = torch.from_numpy(case.users[sl]).to(dev)
    items = torch.from_numpy(case.items[sl]).to(dev)
    by_user = a.feed == "users"
    if by_user:   # user_triplet_set resident; the per-pair ripple sets are assembled inside key addressing
        uts_d = torch.from_numpy(case.user_triplet_set).to(dev)
        if rowshard:
            runner.set_user_triplet_set(uts_d)
        mh = mr = mt = None
    else:
        mh = [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_h]
        mr = [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_r]
        mt = [torch.from_numpy(np.ascontiguousarray(m[sl])).to(dev) for m in case.memories_t]

    def pair_feed():
        m = synth.memories_for(case.user_triplet_set, case.users[sl])
        return [[torch.from_numpy(x).to(dev) for x in lst] for lst in m]

    scorer = None
    if a.graph and not rowshard:
        from mvin_amd.graph import GraphedScorer
        if by_user:
            mh, mr, mt = pair_feed()
            by_user = False
        scorer = GraphedScorer(model, Bl)
        scorer.load(users, items, mh, mr, mt)
    overlap = rowshard and not a.no_overlap
    if overlap:
        runner.enable_pipeline()
    state = {"i": 0}
    ukw = {"distinct_users": distinct} if (by_user and distinct) else {}

    def step():
        if scorer is not None:
            return scorer.replay()
        if not rowshard:
            if by_user:
                return model.forward_users(users, items, uts_d, distinct_users=distinct)
            return model.forward_device(users, items, mh, mr, mt)
        if not overlap:
            return runner.forward_device(users, items, mh, mr, mt, global_batch=a.batch, **ukw)
        # every step scores one batch AND performs one row exchange (for the following batch),
        # on two streams; the same synthetic batch is re-used, the exchange is not skipped
        i = state["i"]
        if i == 0:
            runner.prefetch(0, users, items, mh, mt, global_batch=a.batch)
        out = runner.forward_prefetched(i % 2, users, items, mh, mr, mt, **ukw)
        runner.prefetch((i + 1) % 2, users, items, mh, mt, global_batch=a.batch)
        state["i"] = i + 1
        return out

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # Steps are independent scoring passes: they go round-robin to `--streams` HIP streams (every step still runs its kernels in
    # order on ITS stream; the workspaces of the one-call schedule are per stream).  Not for a hipGraph replay.  The row-sharded
    # runner in its pipelined form takes exactly two: step i scores working table i % 2 on stream i % 2 while the exchange for step
    # i + 1 fills the other table on the runner's side stream (the buffer's ready / free events order the three streams); its
    # serialised form (--no-overlap: one working table) keeps one.
    if a.streams <= 0:
        from mvin_amd.harness import eval_streams
        a.streams = eval_streams(model, Bl)
    nstreams = a.streams if (a.streams > 1 and scorer is None and (not rowshard or overlap)) else 1
    if rowshard and nstreams > 1:
        nstreams = 2
    lanes = get_lanes(dev, nstreams) if nstreams > 1 else None
    if lanes:
        for ln in lanes:
            ln.wait_stream(torch.cuda.current_stream())
    plain_step = step

    def step_on(i):
        if lanes is None:
            return plain_step()
        with torch.cuda.stream(lanes[i % nstreams]):
            return plain_step()

    for i in range(max(a.warmup, nstreams)):
        out = step_on(i)
    barrier()
    # The dominant kernel is timed with HIP events around its launches inside the timed steps (model._profile).  At
    # small batches the pass is ONE native call (mvin_score_l2_fwd, no event hooks inside): there the timed steps run
    # un-instrumented and the kernel time comes from the same K steps repeated with the hooks on, after the clock stops.
    grouped_now = by_user and Bl >= model.group_min_pairs_per_user * min(case.n_user, distinct or case.n_user)
    one_call = (scorer is None and (Bl <= model.native_l2_max_batch or grouped_now)
                and model._native_l2_ok(items, None if by_user else mh, False, cap=not grouped_now))
    one_call = one_call or lanes is not None          # (several streams: the kernel is timed in the single-stream repeat, not overlapped)
    model._profile = None if one_call else []
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step_on(i)
    barrier()
    elapsed = time.perf_counter() - t0
    single_stream = None
    if lanes is not None:                              # the same K steps on ONE stream, after the clock stopped (continuity with rounds 1-4)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        barrier()
        single_stream = (time.perf_counter() - t1) / a.steps
    faithful = None
    if scorer is None and not rowshard and model.prj is None and model._prj_for_l2(Bl, Bl * a.fanout ** (a.hop * a.mix - 2)) \
            and (model._enc_for_l2(n_parents=Bl * a.fanout ** (a.hop * a.mix - 2)) is not None or model._prj_plain_ok()):
        # the same K steps with the two deepest levels in their round-4 form (W1 / W2 products per distinct child), one stream
        model.prj = False
        for _ in range(3):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            out_f = step()
        barrier()
        faithful = ((time.perf_counter() - t1) / a.steps, float((out_f.scores - out.scores).abs().max()))
        model.prj = None
    # ... and with every pair gathering its own rows: the kernels over the projected tables themselves (no per-entity aggregates), on the
    # SAME streams as the timed region.  SURVEY 7.3-c asks for the per-entity route to be a separate, labelled mode: this is the other one.
    gather_mode = None
    if scorer is None and not rowshard and model.agg is None and model.prj is None:
        enc_g = model._enc_for_l2(n_parents=Bl * a.fanout ** (a.hop * a.mix - 2))
        if model._prj_for_l2(Bl, Bl * a.fanout ** (a.hop * a.mix - 2)) and (model._fold_for(enc_g) or model._agg_for(enc_g)):
            model.agg = False
            for i in range(max(3, nstreams)):
                step_on(i)
            barrier()
            t1 = time.perf_counter()
            for i in range(a.steps):
                out_g = step_on(i)
            barrier()
            gather_mode = ((time.perf_counter() - t1) / a.steps, float((out_g.scores - out.scores).abs().max()))
            model.agg = None
    if one_call:
        model._profile = []
        model._ka_profile = []
        for _ in range(a.steps):
            step()
        barrier()
    prof = model._profile
    ka_prof = getattr(model, "_ka_profile", None) or []
    model._profile = None
    model._ka_profile = None
    ranks_counted = 1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.ones(1, dtype=torch.float64, device=dev)       # the ranks that really took part, counted by the fabric
        dist.all_reduce(cnt)
        ranks_counted = int(cnt.item())
        if ranks_counted != a.gpus:
            raise SystemExit(f"bench.py: {ranks_counted} ranks answered the all-reduce, --gpus {a.gpus}")

    if rank == 0:
        # HBM/fabric traffic of the dominant kernel comes from separate rocprofv3 --pmc passes
        # (FETCH_SIZE, WRITE_SIZE; gfx950 correction applied) archived under profiles/; it is
        # attached only when that profile was taken on exactly this workload
        traffic = None
        hoisted = a.hoist != "off" and model.hoist_supported()
        from mvin_amd import ops as _ops
        used_l2 = bool(model.fused and a.hop >= 2 and _ops.gather_attn_l2_supported(a.dim, a.fanout))
        pmc = pmc_record("pmc_latest.json")
        try:
            same = (pmc is not None and a.table_dtype == "f32" and not a.n_entity and pmc.get("bench_args") == {
                "dataset": a.dataset, "dim": a.dim, "hop": a.hop, "mix": a.mix, "fanout": a.fanout, "adj": a.adj,
                "items": a.items, "batch": a.batch})
            if same and used_l2 and world == 1 and not hoisted:
                traffic = pmc["gather_attn_l2_traffic_bytes_per_launch"] / pmc["gather_attn_l2_pairs_per_launch"] * Bl
        except (KeyError, ValueError):
            traffic = None
        L = a.hop * a.mix
        s_ = 2 if a.table_dtype == "bf16" else 4
        bpp = algorithmic_bytes_per_pair(a.dim, a.fanout, L, s=s_)
        bpp_faithful = bpp
        if hoisted:
            # bytes this mode really needs per pair: K+1 fp32 rows of the hoisted tables + the adjacency row
            # per level-(L-2) node, + (mode 'step') the table build amortised over the rank's pairs
            K, D = a.fanout, a.dim
            bpp = K ** (L - 2) * ((K + 1) * D * 4 + 2 * K * 4) + D * 4 + 4
            if a.hoist == "step":
                bpp += case.n_entity * (K * D * s_ + 2 * K * 4 + D * s_ + 4 * D * 4) / Bl
        kern_ms = [e0.elapsed_time(e1) for e0, e1 in prof]
        kern_avg_ms = float(np.mean(kern_ms)) if kern_ms else None
        achieved = (bpp * Bl / (kern_avg_ms * 1e-3) / 1e9) if kern_avg_ms else None
        emulated = a.emulate_world > 1 and world == 1
        # under --emulate-world only rank 0's share was scored: `value` is that MEASURED rate, the W-rank projection is a
        # separate field
        value = (Bl if emulated else a.batch) * a.steps / elapsed
        table_bytes = case.n_entity * a.dim * s_
        cache_resident = table_bytes <= 256 * 2 ** 20        # Infinity Cache (MI355X_MICROARCH.md)
        enc = model._enc_for_l2(n_parents=Bl * a.fanout ** (L - 2)) if (used_l2 and not hoisted) else None
        prj_now = (used_l2 and not hoisted and (enc is not None or model._prj_plain_ok())
                   and model._prj_for_l2(Bl, Bl * a.fanout ** (L - 2)))
        fold_now = bool(prj_now and L == 2 and model._fold_for(enc))
        agg_now = bool(fold_now or (prj_now and model._agg_for(enc)))
        kname = (("gather_mix_kernel (mvin_gather_mix_fwd) + per-entity table build/lookups "
                  "[entity-table mode: its own bytes per pair, not SURVEY 8(d)'s]") if hoisted
                 else "linear_mfma_kernel + entity_aggregates_kernel<%d> + score_l2_folded_kernel<%d> (mvin_fold_tables -> mvin_score_l2_folded_fwd: "
                      "EVERYTHING above key addressing -- four per-entity tables E.W1.A0 | E.W2.A0 | E.W0.A0 | E.W0.Wm0 and the aggregates H0 | G, all "
                      "rebuilt inside every step from the current parameters, then ONE launch that takes a pair from its item id and query row to "
                      "its score: six 64 x 64 products on the matrix cores around the gather of its distinct children's G rows; the same sums in "
                      "another association)" % (a.fanout, a.fanout)
                 if fold_now
                 else "entity_aggregates_kernel<%d> + gather_attn_l2_agg_kernel<%d> (mvin_entity_aggregates -> mvin_gather_attn_l2_agg_fwd: per-entity "
                      "aggregates S0 | G of the projected tables E.W1 | E.W1.A0 | E.W2.A0, all rebuilt inside every step from the current parameters; a "
                      "pair then gathers its distinct children's G rows and one S0 row -- the same sums in another association)" % (a.fanout, a.fanout)
                 if agg_now
                 else "gather_attn_l2_wpp_kernel<%d> (mvin_gather_attn_l2_prj_ordered_fwd: wave-per-parent kernel over the duplicate-slot encoding of the "
                      "adjacency, rows gathered from the projected tables E.W1 | E.W1.A0 | E.W2.A0 that mvin_project_tables rebuilds every step "
                      "-- same ids and grandchild rows per pair as mvin_gather_attn_l2_enc_fwd, no product per distinct child; parents taken in "
                      "item order: %s)" % (a.fanout, bool(model._item_order_for(Bl)))
                 if (prj_now and enc is not None and _ops.gather_attn_l2_wpp_supported(a.dim, a.fanout))
                 else "gather_attn_l2_packed_kernel<..., PRJ> (mvin_gather_attn_l2_prj_fwd: duplicate-slot encoding of the adjacency, rows "
                      "gathered from the projected tables E.W1 | E.W1.A0 | E.W2.A0 that mvin_project_tables rebuilds every step -- same ids "
                      "and grandchild rows per pair as mvin_gather_attn_l2_enc_fwd, no W1 / W2 / A0 product per distinct child)"
                 if (prj_now and enc is not None)
                 else "gather_attn_l2_d32_kernel<..., PRJ> (mvin_gather_attn_l2_prj_fwd over the plain adjacency: rows gathered from the "
                      "projected tables E.W1 | E.W1.A0 | E.W2.A0 that mvin_project_tables rebuilds every step)"
                 if prj_now
                 else "gather_attn_l2_packed_kernel (mvin_gather_attn_l2_enc_fwd: duplicate-slot encoding of the adjacency)"
                 if enc is not None
                 else "gather_attn_l2_kernel (mvin_gather_attn_l2_fwd)" if used_l2
                 else "gather_attn_kernel (mvin_gather_attn_fwd)")
        peak = L2_PEAK_GBS if cache_resident else HBM_PEAK_GBS
        # `achieved` / `frac` price the bytes the kernel MUST LOAD (for a plain adjacency that is SURVEY 8(d)'s figure; for the
        # encoded path it is replaced below by the distinct rows it fetches); the faithful-bytes rate is informational
        timed = {"bound": "l2" if cache_resident else "hbm", "kernel": kname, "achieved": achieved, "peak": peak,
                 "unit": "GB/s", "frac": (achieved / peak) if achieved else None,
                 "bytes_per_pair": bpp, "pairs_per_launch": Bl, "avg_launch_ms": kern_avg_ms,
                 "faithful_bytes_rate_gbs": achieved,
                 "table_bytes": table_bytes, "traffic": traffic,
                 "kernel_timing": ("HIP events around the kernel's launches in a repeat of the same steps after the timed "
                                   "region (the timed steps are ONE native call each, mvin_score_l2_fwd: no hooks inside)"
                                   if one_call else "HIP events around the kernel's launches inside the timed steps"),
                 "hbm_frac_from_traffic": (traffic / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                 if (traffic and kern_avg_ms) else None,
                 "whole_path_faithful_bytes_rate_gbs": value / world * bpp / 1e9,
                 "faithful_bytes_per_pair": bpp_faithful,
                 "note": ("launches inside the timed steps; the %.0f MB entity table is L2 / Infinity-Cache resident, so "
                          "the algorithmic bytes are served on-chip: priced against the %.1f TB/s L2 aggregate, NOT the HBM "
                          "peak; `traffic` = bytes per launch beyond L2 from rocprofv3 PMC passes (2*FETCH_SIZE + "
                          "WRITE_SIZE, profiles/pmc_latest.json)" % (table_bytes / 1e6, L2_PEAK_GBS / 1e3))
                 if cache_resident else "launches inside the timed steps; the table is far larger than the Infinity Cache"}
        if enc is not None and L == 2 and kern_avg_ms:
            # what the encoded path really loads: the DISTINCT slots of the item's row and of its distinct children's rows
            # (the algorithmic bytes above stay SURVEY 8(d)'s K + K^2 rows per pair: repeated slots are not re-read)
            enc_e, enc_r, cnt, frac_distinct = enc
            it = items.long()
            ch = (enc_e[it] & 0xFFFFFF).long()
            real = ((enc_r[it] >> 16) & 0xFF) > 0
            rows_loaded = float((1 + cnt[it].double() + (cnt[ch].double() * real).sum(1)).mean())
            if prj_now:       # the projected-tables form reads TWO self rows per distinct child (T1[x] and TA1[x])
                rows_loaded += float(cnt[it].double().mean())
            K_ = a.fanout
            # encoded adjacency rows read per pair: the item's and its distinct children's, K words of ids + K of relations each
            adj_rows = float((1 + cnt[it].double()).mean())
            if agg_now:
                # the aggregates form: per pair its S0 row and its distinct children's G rows, its own adjacency row, the query row and
                # the two output rows; per ENTITY and step (entity_aggregates_kernel) its adjacency row, TA1[e], T1 and TA2 of its
                # distinct slots, S0[e] and G[e] written -- shared out over the step's pairs
                rows_loaded = float((1 + cnt[it].double()).mean())
                adj_rows = 1.0
                per_entity = float(((2 * cnt.double() + 1) * a.dim * 4 + K_ * 8 + 2 * a.dim * 4).sum()) / Bl
                loaded_bpp = rows_loaded * a.dim * 4 + K_ * 8 + 3 * a.dim * 4 + per_entity
                if fold_now:
                    # folded tail, one launch: per pair H0[x], M0[x] and its distinct children's G rows, its adjacency row, its query and
                    # user_o rows, the item embedding written, the score; per ENTITY and step E[e] read and four table rows written (table
                    # build), then the aggregates kernel as above with one more self row (T0A[e])
                    rows_loaded = float((2 + cnt[it].double()).mean())
                    per_entity = float(((2 * cnt.double() + 2) * a.dim * 4 + K_ * 8 + 2 * a.dim * 4 + 5 * a.dim * 4).sum()) / Bl
                    loaded_bpp = rows_loaded * a.dim * 4 + K_ * 8 + 3 * a.dim * 4 + 8 + per_entity
                timed["aggregates_build_bytes_per_step"] = per_entity * Bl
            else:
                loaded_bpp = rows_loaded * a.dim * s_ + adj_rows * K_ * 8 + a.dim * 4 + 4
            loaded_gbs = loaded_bpp * Bl / (kern_avg_ms * 1e-3) / 1e9
            timed.update({"achieved": loaded_gbs, "frac": loaded_gbs / peak, "bytes_per_pair": loaded_bpp,
                          "rows_per_pair_faithful": 1 + K_ + K_ * K_, "rows_per_pair_loaded": rows_loaded,
                          "adjacency_rows_per_pair_loaded": adj_rows,
                          "distinct_slots_per_adjacency_row": frac_distinct * K_,
                          "folded_note": ("folded-tail form: `avg_launch_ms` covers mvin_fold_tables (parameter block, four-table build, aggregates) AND "
                                          "the scoring launch -- everything above key addressing, mvin_l2_tail_fwd's work included; MVIN_L2_FOLD=0 "
                                          "times aggregates + tail kernel, MVIN_L2_AGG=0 the kernel over the projected tables") if fold_now else None,
                          "aggregates_note": ("aggregates form: `avg_launch_ms` covers BOTH launches (aggregates over all %d entities + the per-pair "
                                              "gather); a pair's rows come from the 27 MB G table, not from its grandchildren's rows of three tables "
                                              "(rows_per_pair_loaded ~ 1 + its distinct children; the wave-per-parent kernel over the tables loaded "
                                              "~120); MVIN_L2_AGG=0 times that kernel instead" % case.n_entity) if agg_now else None,
                          "dedup_note": "the reference's sampler repeats slots whenever deg < K (data_loader_user_set.py:383-384) and the "
                                        "packed kernel fetches every DISTINCT row once: `achieved` / `frac` / `bytes_per_pair` price what it "
                                        "loads (distinct entity rows + the encoded adjacency rows + query + score); "
                                        "`faithful_bytes_rate_gbs` = SURVEY 8(d)'s bytes per pair (every slot's row) over the same time is "
                                        "informational and can exceed any peak -- it is a rate of work done, not of bytes moved"})
        if ka_prof and by_user:
            # the flash form of key addressing + user MLP inside the same (single-stream, instrumented) repeat of the timed steps:
            # a kernel that IS bound by bytes from beyond the L2 -- every user's ripple rows come from a 245 MB per-call table
            # (R_KGE[r] . E[e]) and three entity-sized ones, once per user and 32-pair tile
            prep_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ka_prof]))
            kern2_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ka_prof]))
            P_, Nm_, D_ = d["p_hop"], d["n_memory"], a.dim
            NmP = (Nm_ + 15) // 16 * 16
            ucount = torch.bincount(users, minlength=case.n_user)
            n_users_now = int((ucount > 0).sum().item())
            tiles32 = int(((ucount + 31) // 32).sum().item())
            slots64 = int(((ucount + 63) // 64).sum().item())
            row_b = D_ * 4
            # per 32-pair tile: P NmP rows of R_KGE.E + P NmP rows of E.Wmlp blocks; per 64-pair slot: NmP h-set rows + the record's three
            # id sections; per pair: the item row, its id, its index in user order, the user_o row written
            bytes_step = (tiles32 * 2 * P_ * NmP * row_b + slots64 * (NmP * row_b + (NmP + 2 * P_ * NmP) * 4)
                          + Bl * (row_b + 8 + 4 + row_b))
            tab_bytes = (case.n_relation + P_ + 1) * case.n_entity * row_b
            timed["key_addressing_kernel"] = {
                "kernel": "key_addr_flash_kernel (mvin_key_addressing_flash_fwd: attention reads + user MLP of model.py:161-240 in one barrier-free "
                          "kernel over per-call tables)",
                "avg_launch_ms": kern2_ms, "tables_build_ms": prep_ms, "tables_bytes_written_per_step": tab_bytes,
                "users_in_batch": n_users_now, "tiles_of_32_pairs": tiles32,
                "bytes_loaded_per_step": bytes_step, "bound": "hbm", "achieved": bytes_step / (kern2_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": bytes_step / (kern2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "bytes the kernel must load per step (every row counted once per tile that needs it; the 245 MB R_KGE.E table and the "
                        "82 MB of E.Wmlp blocks are far beyond the 8 x 4 MB of L2, the ripple sets are uniform random: SURVEY 8(d)) over its "
                        "HIP-event time in the instrumented single-stream repeat; rocprofv3 counters of the same kernel under profiles/r6/ "
                        "(TCC_EA0_RDREQ: ~1.2x these bytes leave the L2)"}
        if used_l2 and not hoisted and L == 2 and world == 1 and not a.no_probe and kern_avg_ms:
            gp = gather_probe(model, items, a.fanout, a.dim, s_)
            rows_gbs = gp["row_bytes_per_pair"] * Bl / (kern_avg_ms * 1e-3) / 1e9     # the fused kernel, rows only
            gp["fused_kernel_rows_gbs"] = rows_gbs
            gp["fused_kernel_frac_of_probe"] = rows_gbs / gp["achieved"]
            gp["note"] = ("mvin_probe_gather_l2: the K + K*K table rows per pair of the timed region's own pairs, read once each with "
                          "the same 16-byte lane loads by a PLAIN gather-and-sum kernel (one wave per parent, 8 loads in flight, ids "
                          "from the expanded level lists one round ahead) -- a reference point, not an upper bound; "
                          "fused_kernel_frac_of_probe = the full kernel's row bytes per second (adjacency chase, softmax, "
                          "projections, MFMA epilogues on top) over the probe's")
            timed["gather_only_probe"] = gp
        roofline = timed
        if not a.no_hbm_leg and used_l2 and not hoisted and cache_resident and L == 2:
            # the genuinely HBM-bound measurement of the kernel INSTANCE the timed steps run, live, after the timed region: the
            # projected-tables instance when that is what they take (its own leg: 4 M-row table, 3 x 1.02 GB of projected tables, its
            # own bytes per pair), the unprojected instance otherwise; the sibling instances are reported next to it
            def leg_brief(o):
                b = {k: o[k] for k in ("kernel", "form", "avg_launch_ms", "achieved", "bytes_per_pair", "pairs_per_launch", "table_rows",
                                       "table_bytes", "max_abs_err", "verified")}
                b["frac"] = o["achieved"] / HBM_PEAK_GBS
                return b

            def leg_traffic_of(leg, fname):
                pl = pmc_record(fname)
                if pl and all(pl.get(k) == v for k, v in (("dim", a.dim), ("fanout", a.fanout), ("table_rows", leg["table_rows"]),
                                                          ("pairs_per_launch", leg["pairs_per_launch"]),
                                                          ("table_dtype", a.table_dtype))) \
                        and leg["kernel"].split(" ")[0].split("<")[0] in pl.get("kernel", leg["kernel"]):     # the PMC passes timed the same kernel
                    return pl["traffic_bytes_per_launch"]
                return None
            prj_leg = bool(prj_now and a.table_dtype == "f32")
            leg = hbm_leg(dev, a.dim, a.fanout, a.table_dtype, encoded=enc is not None, prj=prj_leg)
            leg_traffic = leg_traffic_of(leg, "pmc_hbm_leg_prj.json" if prj_leg else "pmc_hbm_leg.json")
            siblings = {}
            if prj_leg and enc is not None:      # the unprojected packed-tile instance on the 16 M-row table (rounds 4-5's roofline leg)
                o = hbm_leg(dev, a.dim, a.fanout, a.table_dtype, encoded=True)
                siblings["unprojected_kernel"] = leg_brief(o)
                siblings["unprojected_kernel"]["traffic"] = leg_traffic_of(o, "pmc_hbm_leg.json")
            if enc is not None or prj_leg:       # the plain-adjacency kernel on the same 16 M-row table (round 3's roofline kernel)
                siblings["plain_adjacency_kernel"] = leg_brief(hbm_leg(dev, a.dim, a.fanout, a.table_dtype, encoded=False))
            roofline = {"bound": "hbm", "kernel": leg["kernel"], "achieved": leg["achieved"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": leg["achieved"] / HBM_PEAK_GBS, "traffic": leg_traffic,
                        "bytes_per_pair": leg["bytes_per_pair"], "bytes_per_pair_formula": leg["bytes_per_pair_formula"],
                        "pairs_per_launch": leg["pairs_per_launch"],
                        "avg_launch_ms": leg["avg_launch_ms"], "table_bytes": leg["table_bytes"],
                        "table_rows": leg["table_rows"], "launches": leg["launches"],
                        "max_abs_err": leg["max_abs_err"], "verified": leg["verified"], "verified_how": leg["verified_how"],
                        "instance_is_the_timed_regions": bool(prj_leg == bool(prj_now)) and not agg_now,
                        "instance_note": ("the timed steps take the per-entity aggregates form (timed_region.kernel): it exists where the batch holds "
                                          "more parents than ~n_entity / 6 -- on this leg's 4 M-row table the aggregates alone would cost 4 M x 65 "
                                          "rows per launch, so the gather kernel over the projected tables stays the kernel of this regime (and of "
                                          "MVIN_L2_AGG=0) and keeps the leg; the aggregates kernels' own bytes and time are under timed_region"
                                          if agg_now else None),
                        **siblings,
                        "workload": ("%s -- %s -- on a %.1f M-row synthetic entity table (%.2f GB of tables: %.0fx the Infinity Cache) with uniform "
                                     "adjacency (no repeated slots: every row of every slot is loaded), so every gathered row comes from HBM; "
                                     "measured after the timed region with HIP events around each launch on the launch stream; `traffic` = "
                                     "2*FETCH_SIZE + WRITE_SIZE per launch from profiles/%s (rocprofv3 --pmc passes over `bench.py "
                                     "--hbm-leg-only%s`)"
                                     % (leg["kernel"],
                                        "the gather kernel over the projected tables (the timed steps' tables; they then read per-entity aggregates of them)"
                                        if agg_now else
                                        "the template instance the timed steps launch" if prj_leg == bool(prj_now) else
                                        "NOT the timed region's instance (that one reads projected tables; see `timed_region.kernel`)",
                                        leg["table_rows"] / 1e6, leg["table_bytes"] / 1e9, leg["table_bytes"] / (256 * 2 ** 20),
                                        "pmc_hbm_leg_prj.json" if prj_leg else "pmc_hbm_leg.json",
                                        " --hbm-leg-form prj" if prj_leg else "")),
                        "timed_region": timed}
        rec = {
            "metric": "(user,item) pairs scored/sec @ dim=%d hop=%d fan-out=%d; %% HBM roofline"
                      % (a.dim, a.hop, a.fanout),
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.table_dtype, "data": "synthetic",
            "mode": ("per-entity aggregates (SURVEY 7.3-c route 2b, exact; every table rebuilt inside every timed step, nothing cached): the "
                     "pair-independent sums of the two deepest levels are taken per ENTITY, a pair gathers ~9 rows; the mode in which every "
                     "pair gathers its own rows is other_modes.every_pair_gathers_its_rows, and `roofline` is THAT mode's kernel"
                     if agg_now else "every pair gathers its own rows") if (used_l2 and not hoisted) else
                    ("entity-table mode (--hoist)" if hoisted else "per-level kernels"),
            "config": {"workload": f"{a.dataset}-shaped tables (nE={case.n_entity}, nU={case.n_user}, "
                                   f"nR={case.n_relation}), dim={a.dim} hop={a.hop} n_mix_hop={a.mix} "
                                   f"fan-out={a.fanout} p_hop={d['p_hop']} n_memory={d['n_memory']}, "
                                   f"full get_scores path",
                       "pairs_per_step_total": a.batch, "pairs_per_gpu_per_step": Bl,
                       "adjacency": a.adj, "items": a.items, "ablation": a.ablation, "hipgraph_replay": bool(scorer), "streams": nstreams,
                       "two_level_form": (("projected tables (E.W1 | E.W1.A0 | E.W2.A0 rebuilt inside every timed step: mvin_project_tables + "
                                           "mvin_gather_attn_l2_prj_fwd" + (("; parents in item order, mvin_order_by_key inside every step)"
                                                                             if (model._item_order_for(Bl) and not agg_now) else ")") if enc is not None
                                                                            else " over the plain adjacency)")
                                           + ("; per-entity aggregates S0 | G of those tables rebuilt inside every timed step too: mvin_entity_aggregates + "
                                              "mvin_gather_attn_l2_agg_fwd in place of mvin_gather_attn_l2_prj_fwd" if (agg_now and not fold_now) else "")
                                           if not fold_now else
                                           "folded tail over per-entity aggregates (E.W1.A0 | E.W2.A0 | E.W0.A0 | E.W0.Wm0 and H0 | G rebuilt inside every "
                                           "timed step: mvin_fold_tables; then mvin_score_l2_folded_fwd, one launch from item id to score)") if prj_now else
                                          "encoded adjacency" if enc is not None else "plain adjacency"),
                       "entity_table_dtype": a.table_dtype, "arithmetic": "f32", "entity_table_mode": a.hoist,
                       "feed_mode": "pairs" if (a.feed == "pairs" or scorer is not None) else "users",
                       "key_addressing_variant": (
                           "pairs (mvin_key_addressing_fwd: 2*P*Nm rows gathered per pair)"
                           if (a.feed == "pairs" or scorer is not None) else
                           ("grouped, flash form (mvin_key_addressing_flash_prepare + mvin_key_addressing_flash_fwd: the per-call tables R_KGE[r] . E[e] "
                            "and E . user_mlp blocks are rebuilt inside every timed step; attention reads AND the user MLP in one barrier-free kernel "
                            "over the static per-user records, each wave walking tiles of 32 pairs of one user on transposed MFMA products)"
                            if (getattr(model, "_uts_records", None) is not None and by_user and model._ka_flash_for(uts_d, model._uts_records[3], Bl)) else
                            "grouped over static per-user records (mvin_key_addressing_grouped_rec_fwd: relation buckets, tile table "
                            "and clamped ids of every user's ripple sets built once by mvin_build_user_records; a user's rows staged once "
                            "per user segment)" if getattr(model, "_uts_records", None) is not None else
                            "grouped (mvin_key_addressing_grouped_fwd: a user's rows staged once per user segment)")
                           if Bl >= model.group_min_pairs_per_user * min(case.n_user, distinct or case.n_user) else
                           "users (mvin_key_addressing_users_fwd: per pair, lists read out of user_triplet_set)"),
                       "feed": ("user_triplet_set [n_user, P, 3, n_memory] + (user, item) ids resident in HBM; the per-pair "
                                "ripple sets of train.py:117-120 are assembled inside key addressing, which groups the "
                                "batch's pairs by user" if a.feed == "users" and scorer is None else
                                "per-pair ripple-set arrays [B, n_memory] + (user, item) ids resident in HBM"),
                       "pair_split": ("contiguous slices of the batch in user order (a rank holds all pairs of ~n_user/W "
                                      "users)" if split > 1 else "none"),
                       "parallelism": (f"pairs split over {world} rank(s); entity table row-sharded (owner = id mod {world}) "
                                       f"+ all-to-all row exchange per step ({'dense' if runner.is_dense(a.batch) else 'sparse'} regime)"
                                       f"{' overlapped with scoring (2 streams, 2 working tables)' if overlap else ''}"
                                       if rowshard else
                                       (f"pairs split over {world} ranks; tables replicated" if world > 1
                                        else "single-gpu"))},
            "roofline": roofline,
        }
        # what the multi-GPU line rests on, printed so that the first real SCALE run verifies itself
        import torch.distributed as tdist
        dinfo = {"world_size": tdist.get_world_size() if tdist.is_initialized() else 1,
                 "backend": tdist.get_backend() if tdist.is_initialized() else None, "ranks_launched": ranks_counted,
                 "gpus_asked": a.gpus, "launcher": os.environ.get("MVIN_BENCH_LAUNCHER", "external" if world > 1 else "none"),
                 "production_transport": bool(world == 1 or (tdist.is_initialized() and tdist.get_backend() == "nccl"))}
        if rowshard:
            from mvin_amd.dist import exchange_wire_bytes, n_local_rows
            W_ = split
            row_b = a.dim * s_
            dense = runner.is_dense(a.batch)
            wb = exchange_wire_bytes(case.n_entity, row_b, W_, Bl, a.fanout, L)
            dinfo.update({"exchange_regime": "dense (every shard to every rank)" if dense else "sparse (fixed-capacity id buffers, no host sync)",
                          "exchange_bytes_received_per_rank_per_step": (wb["replicate"] if dense else runner.table.last_stats.get("wire_bytes_per_rank")),
                          "exchange_bytes_if_owner_side_partial_sums": wb["partial_sums"],
                          "shard_rows": n_local_rows(case.n_entity, W_), "row_bytes": row_b,
                          "collective": "all_to_all over W views of the local shard (RCCL grouped send/recv)" if dense
                                        else "all_to_all_single (ids) + all_to_all_single (rows), static equal splits",
                          "world_in_formula": W_})
        rec["distributed"] = dinfo
        if single_stream is not None:
            rec.setdefault("other_modes", {})["single_stream"] = {
                "value": (Bl if emulated else a.batch) / single_stream, "unit": "pairs/s", "ms_per_step": 1e3 * single_stream,
                "note": "the same K steps enqueued on ONE stream (how rounds 1-4 ran the line): every kernel waits for the last workgroup "
                        "of the one before; with `config.streams` streams the independent steps' kernels fill those drains"}
        if gather_mode is not None:
            rec.setdefault("other_modes", {})["every_pair_gathers_its_rows"] = {
                "value": (Bl if emulated else a.batch) / gather_mode[0], "unit": "pairs/s", "ms_per_step": 1e3 * gather_mode[0], "streams": nstreams,
                "max_abs_diff_vs_timed_scores": gather_mode[1],
                "note": "the same K steps on the same streams with MVIN.agg = False (MVIN_L2_AGG=0): the two deepest levels as the gather kernel "
                        "over the projected tables (mvin_gather_attn_l2_prj_fwd: every pair gathers the rows of its own distinct children and "
                        "grandchildren -- the kernel of `roofline`) + mvin_l2_tail_fwd, i.e. WITHOUT the per-entity sums of SURVEY 7.3-c's route "
                        "2b that `value` takes"}
        if faithful is not None:
            rec.setdefault("other_modes", {})["unprojected_two_level"] = {
                "value": (Bl if emulated else a.batch) / faithful[0], "unit": "pairs/s", "ms_per_step": 1e3 * faithful[0], "streams": 1,
                "max_abs_diff_vs_timed_scores": faithful[1],
                "note": "the same steps on one stream with mvin_gather_attn_l2_enc_fwd (W1 / W2 applied per distinct child: rounds 1-4) "
                        "instead of the projected tables; compare with other_modes.single_stream"}
        if a.emulate_world > 1 and world == 1:
            rec["projected_value"] = a.batch * a.steps / elapsed
            rec["emulated_world"] = {"world": a.emulate_world, "per_rank_pairs_per_s": Bl * a.steps / elapsed,
                                     "note": "ONE GPU scoring rank 0's share as one of W ranks would (user-sorted split, "
                                             "row-shard exchange of the whole table every step through RCCL at world size "
                                             "1, two streams); `value` = the MEASURED per-rank rate, `projected_value` = x W is a "
                                             "PROJECTION of the W-GPU line, not a measurement: no fabric is involved; the user-order "
                                             "partition of the global batch is done on the host OUTSIDE the timed steps"}
        if world == 1 and not a.no_sweep and a.hoist == "off" and not rowshard:
            smh, smr, smt = (mh, mr, mt) if mh is not None else pair_feed()   # per-pair feeds: the reference's own
            uts_sw = uts_d if a.feed == "users" else torch.from_numpy(case.user_triplet_set).to(dev)
            rec["batch_sweep"] = batch_sweep(model, users, items, smh, smr, smt, [int(x) for x in a.sweep.split(",") if x],
                                             uts=uts_sw)
        if world == 1 and a.hoist == "off" and not rowshard and scorer is None:
            # the same pass with the OTHER feed, after the timed region
            if by_user:
                pmh, pmr, pmt = pair_feed()
                alt = lambda: model.forward_device(users, items, pmh, pmr, pmt)          # noqa: E731
                key, note = "per_pair_feed", ("per-pair ripple-set arrays [B, n_memory] resident in HBM (the contents "
                                              "of the reference's feed_dict): key addressing gathers 2*P*Nm rows per "
                                              "PAIR (mvin_key_addressing_fwd)")
            else:
                uts_alt = torch.from_numpy(case.user_triplet_set).to(dev)
                alt = lambda: model.forward_users(users, items, uts_alt)                  # noqa: E731
                key, note = "user_grouped_feed", ("MVIN.forward_users: user_triplet_set resident, key addressing groups "
                                                  "the batch's pairs by user (mvin_key_addressing_grouped_fwd)")
            for _ in range(2):
                go = alt()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                go = alt()
            torch.cuda.synchronize()
            dtg = (time.perf_counter() - t1) / 5
            rec.setdefault("other_modes", {})[key] = {
                "value": a.batch / dtg, "unit": "pairs/s", "ms_per_step": 1e3 * dtg,
                "max_abs_diff_vs_timed_feed_scores": (go.scores - out.scores).abs().max().item(), "note": note}
            del go
        if world == 1 and not a.no_sweep and a.hoist == "off" and not rowshard and a.adj == "kg" and not a.n_entity:
            try:                                                          # informational, after the timed region
                rec.setdefault("other_modes", {})["uniform_adjacency"] = uniform_adjacency_variant(a, dev)
            except Exception as e:
                rec.setdefault("other_modes", {})["uniform_adjacency"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline and a.hoist == "off" and not rowshard and model.hoist_supported():
            # informational only, measured AFTER the timed region on the same inputs: the entity-table mode
            #  has its own bytes per pair and is never `value`
            hm = MVIN(margs, case.n_user, case.n_entity, case.n_relation, case.adj_entity, case.adj_relation,
                      params=params, device=dev, table_dtype=a.table_dtype, hoist=True)
            hstep = ((lambda: hm.forward_users(users, items, uts_d)) if by_user
                     else (lambda: hm.forward_device(users, items, mh, mr, mt)))
            for _ in range(2):
                ho = hstep()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                ho = hstep()
            torch.cuda.synchronize()
            dth = (time.perf_counter() - t1) / 5
            dev_err = (ho.scores - out.scores).abs().max().item()
            rec.setdefault("other_modes", {})["entity_table_mode_cached"] = {
                "value": a.batch / dth, "unit": "pairs/s", "ms_per_step": 1e3 * dth,
                "max_abs_diff_vs_faithful_scores": dev_err,
                "note": "per-entity tables hoist the pair-independent part of the two deepest levels "
                        "(SURVEY 7.3-c route 2b): different algorithmic bytes per pair, reported separately"}
            del hm, ho
        if world == 1 and not a.no_sweep and a.hoist == "off" and not rowshard and a.table_dtype == "f32":
            # informational, after the timed region: the caller AFTER the path -- forward + loss + backward +
            # Adam on a separate model instance with the same parameters, at the reference's batch sizes
            try:
                nt = min(4096, Bl)
                tmem = [[torch.from_numpy(x).to(dev) for x in lst]
                        for lst in synth.memories_for(case.user_triplet_set, case.users[sl][:nt])]
                rec.setdefault("other_modes", {})["training_step"] = training_steps(margs, case, params, dev, users[:nt],
                                                                                   items[:nt], tmem)
            except Exception as e:                                        # never lose the scoring line to this leg
                rec.setdefault("other_modes", {})["training_step"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_sweep and a.hoist == "off" and not rowshard and a.dim != 16 and a.table_dtype == "f32":
            try:                                                          # informational, after the timed region
                rec.setdefault("other_modes", {})["reference_shipped_config"] = shipped_config(a, dev)
            except Exception as e:
                rec.setdefault("other_modes", {})["reference_shipped_config"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            cb, ref, Bc = cpu_baseline(a, margs, case, params)
            rec["cpu_baseline"] = cb
            got = out.scores[:Bc].cpu().numpy()  # rank 0 owns the first pairs of the global batch
            err = np.abs(got - ref.scores.numpy())
            rec["parity_vs_cpu_sample"] = {"max_abs_err": float(err.max()),
                                           "within_1e-5rel_1e-6abs": bool((err <= 1e-5 * np.abs(ref.scores.numpy()) + 1e-6).all())}
            if not a.no_sweep:
                try:                                                      # the CPU column at C1 is mandatory
                    rec["cpu_baseline"]["c1"] = c1_plumbing(a, dev, cb["cores"])
                except Exception as e:
                    rec["cpu_baseline"]["c1"] = {"error": f"{type(e).__name__}: {e}"}
    else:
        rec = None
    if dist.is_initialized():
        dist.destroy_process_group()
    if rec is not None:
        # the JSON line goes out LAST: RCCL writes a version banner through C stdio, which would otherwise be
        # flushed after it at exit
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
