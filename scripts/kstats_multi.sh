#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel durations of the MULTI-launch schedule at one batch size (MVIN_SMALL=0), both feeds.
# usage: scripts/kstats_multi.sh <tag> <B>
tag=$1; B=$2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
o=$root/gpurun_out/ksm_$tag; mkdir -p "$o"
cd /tmp && export TMPDIR=/tmp
MVIN_SMALL=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$o" -- python "$root/scripts/bench_small_batch.py" --steps 30 --sizes $B --groups 1 --only enc --feeds pairs > "$o/log" 2>&1 < /dev/null
f=$(ls "$o"/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $tag: B=$B multi-launch"
[ -n "$f" ] && head -14 "$f" | python3 -c "
import sys, csv
for r in csv.reader(sys.stdin):
    if r[0] == 'Name': continue
    print('%-78s calls %5s avg %9.2f us' % (r[0][:78], r[1], float(r[3]) / 1e3))
"
grep "multi-launch" "$o/log"
