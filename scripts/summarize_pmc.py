"""Reduce rocprofv3 --pmc counter_collection CSVs (one pass per counter) to per-kernel means and
the fused kernel's bytes-beyond-L2 per launch.  Usage: summarize_pmc.py <dir> [bench.py args...]
(the bench args are parsed only to record which workload the counters belong to).
gfx950: FETCH_SIZE / WRITE_SIZE are reported in KiB; read bytes = 2 * FETCH_SIZE * 1024
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    hbm = "--hbm-leg" in sys.argv
    if hbm:
        sys.argv.remove("--hbm-leg")
    prj = "--prj" in sys.argv      # the projected-tables leg (bench.py --hbm-leg-only --hbm-leg-form prj): hbm_prj_pmc_* -> pmc_hbm_leg_prj.json
    if prj:
        sys.argv.remove("--prj")
    out = sys.argv[1]
    saved = sys.argv
    sys.argv = ["bench.py"] + saved[2:]
    import bench
    a = bench.parse()
    sys.argv = saved
    kernels = {}
    prefix = ("hbm_prj_pmc_" if prj else "hbm_pmc_") if hbm else "pmc_"
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob(os.path.join(out, f"{prefix}{c}", "*", "*counter_collection.csv")):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != c:
                        continue
                    k = kernels.setdefault(row["Kernel_Name"], {}).setdefault(c, [])
                    k.append(float(row["Counter_Value"]))
    summ = {}
    for name, cs in kernels.items():
        if "mvin::" not in name:
            continue
        summ[name] = {}
        for c, v in cs.items():
            v = v[len(v) // 6:] if len(v) > 6 else v     # drop the warm-up launches' share
            summ[name][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    # the kernel of the two deepest levels: the folded-tail launch where the timed steps take it, the gather kernels otherwise
    fused = [k for k in summ if "score_l2_folded" in k] or [k for k in summ if "gather_attn_l2" in k]
    if hbm:
        k = summ[fused[0]]
        rec = {"command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py "
                          + " ".join(saved[2:]) + " (separate passes: FETCH_SIZE, WRITE_SIZE)",
               "dim": a.dim, "fanout": a.fanout, "table_dtype": a.table_dtype,
               "table_rows": bench.HBM_LEG_PRJ_ROWS if prj else bench.HBM_LEG_ROWS,
               "pairs_per_launch": bench.HBM_LEG_PAIRS,
               "algorithmic_bytes_per_launch": (bench.prj_bytes_per_pair(a.dim, a.fanout) if prj else bench.algorithmic_bytes_per_pair(
                   a.dim, a.fanout, 2, 2 if a.table_dtype == "bf16" else 4)) * bench.HBM_LEG_PAIRS,
               "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; gfx950 correction: read bytes = 2*FETCH_SIZE*1024 "
                        "(MI355X_MICROARCH.md, HBM section)",
               "kernel": fused[0], "counters": k,
               "traffic_bytes_per_launch": (2 * k["FETCH_SIZE"]["mean_per_launch"]
                                            + k["WRITE_SIZE"]["mean_per_launch"]) * 1024}
        with open(os.path.join(out, "pmc_hbm_leg_prj.json" if prj else "pmc_hbm_leg.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print(json.dumps({k: rec[k] for k in ("traffic_bytes_per_launch", "algorithmic_bytes_per_launch")}))
        return
    rec = {
        "command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py "
                   + " ".join(saved[2:]) + " (separate passes: FETCH_SIZE, WRITE_SIZE; scripts/collect_profiles.sh)",
        "bench_args": {"dataset": a.dataset, "dim": a.dim, "hop": a.hop, "mix": a.mix, "fanout": a.fanout,
                       "adj": a.adj, "items": a.items, "batch": a.batch},
        "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; gfx950 correction: read bytes = 2*FETCH_SIZE*1024 "
                 "(MI355X_MICROARCH.md, HBM section). Calibration on this access pattern: the 4 GiB-table "
                 "microbench (scripts/bench_l2.py, 16384 pairs) reads 2*2231080 KiB = 4.57 GB per launch vs "
                 "its algorithmic 4.58 GB.",
        "kernels": summ,
    }
    if fused:
        k = summ[fused[0]]
        rec["gather_attn_l2_traffic_bytes_per_launch"] = (2 * k["FETCH_SIZE"]["mean_per_launch"]
                                                          + k["WRITE_SIZE"]["mean_per_launch"]) * 1024
        rec["gather_attn_l2_pairs_per_launch"] = a.batch
        rec["gather_attn_l2_kernel"] = fused[0]
    with open(os.path.join(out, "pmc.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in rec if k.startswith("gather")}))


if __name__ == "__main__":
    main()
