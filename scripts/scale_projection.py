#!/usr/bin/env python3
"""Run ON the GPU box: per-rank step time of the default batch scored as one of W = 1, 2, 4, 8 ranks would
(bench.py --emulate-world W: user-sorted split, row-shard exchange every step, id relabelling launch included) and the
strong-scaling efficiency it projects -> gpurun_out/scale_projection.json.  A projection on ONE GPU: no fabric."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "100"
rows = []
for W in (1, 2, 4, 8):
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-hbm-leg", "--no-sweep", "--no-cpu-baseline", "--steps", steps,
           "--warmup", "5"] + ([] if W == 1 else ["--emulate-world", str(W)])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    rows.append({"world": W, "emulated": W > 1, "pairs_per_rank": d["config"]["pairs_per_gpu_per_step"] if W == 1 else d["config"]["pairs_per_step_total"] // W,
                 "ms_per_step": d["ms_per_step"], "projected_value": d.get("projected_value", d["value"]),
                 "fused_kernel_ms": d["roofline"].get("timed_region", d["roofline"]).get("avg_launch_ms")})
t1 = rows[0]["ms_per_step"]
for r in rows:
    r["projected_strong_scaling_efficiency"] = t1 / (r["world"] * r["ms_per_step"])
rec = {"what": "ONE MI355X scoring rank 0's share of the default C3 batch (524 288 pairs) as one of W ranks would: bench.py "
               "--emulate-world W (user-sorted split, row-shard exchange of the whole table every step over RCCL at world size 1, "
               "two streams, the shard-space id relabelling launch included). A PROJECTION of the strong-scaling line (no fabric "
               "involved), not a multi-GPU measurement.", "steps": int(steps), "rows": rows}
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(root, "gpurun_out", "scale_projection.json"), "w"), indent=1)
for r in rows:
    print(r["world"], round(r["ms_per_step"], 4), round(r["projected_strong_scaling_efficiency"], 4), r["fused_kernel_ms"])
