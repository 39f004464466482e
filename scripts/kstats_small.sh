#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel durations of the single-launch pass (scripts/bench_small_batch.py), optionally cut after
# stage N (MVIN_SMALL_DBG) -- true kernel time, not host-paired wall time.
# usage: scripts/kstats_small.sh <tag> [bench_small_batch.py args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
o=$root/gpurun_out/kss_$tag; mkdir -p "$o"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$o" -- python "$root/scripts/bench_small_batch.py" --steps 30 "$@" > "$o/log" 2>&1 < /dev/null
f=$(ls "$o"/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $tag: $* (MVIN_SMALL_DBG=${MVIN_SMALL_DBG:-0})"
[ -n "$f" ] && grep -E "score_small|Name" "$f" | python3 -c "
import sys, csv
for r in csv.reader(sys.stdin):
    if r[0] == 'Name': continue
    print('%-60s calls %5s avg %9.2f us  min %9.2f  max %9.2f' % (r[0][:60], r[1], float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3))
"
