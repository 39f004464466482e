#!/bin/bash
# Run ON the GPU box: SQ / TCP / TCC counters of the default bench.py workload, one rocprofv3 --pmc pass
# per counter group (never combined with sys/hip/hsa traces), reduced to per-kernel means.
# usage: scripts/pmc_counters.sh <tag> [env VAR=..] -- [bench.py args...]
set -u
tag=${1:-latest}; shift || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bargs="--steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-sweep --streams 1 $*"     # (kernels alone: see collect_profiles.sh)
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/g$i" -- python "$root/bench.py" $bargs > "$out/g$i.log" 2>&1
done
cd "$root"
python - "$out" <<'PY'
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
    if "gather_attn_l2" in k or "key_addr" in k:
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-34s %.4g" % (c, v))
PY
