#!/bin/bash
# Run ON the GPU box: SQ / TCP / TCC counters of the flash key-addressing kernel alone (scripts/bench_ka_flash.py, ONLY="flash alone"),
# one rocprofv3 --pmc pass per counter group, reduced to per-kernel means.  usage: scripts/pmc_flash.sh <tag> [dataset] [B]
set -u
tag=${1:-flash}; shift || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export ONLY="${ONLY:-flash alone}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/g$i" -- python "$root/scripts/${SCRIPT:-bench_ka_flash.py}" "$@" > "$out/g$i.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$root/scripts/${SCRIPT:-bench_ka_flash.py}" "$@" > "$out/stats.log" 2>&1
cd "$root"
KERNEL=${KERNEL:-key_addr_flash} python - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        if "mvin::" in row["Kernel_Name"]:
            d[row["Kernel_Name"].split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in cs.items()} for k, cs in d.items()}
json.dump(res, open(out + "/counters.json", "w"), indent=1)
for k, cs in res.items():
    if __import__("os").environ["KERNEL"] in k:
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-34s %.4g" % (c, v))
PY
f=$(ls "$out"/stats/*/*kernel_stats.csv | head -1); head -6 "$f" | cut -c1-200
