#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of one bench.py invocation, top kernels printed.
# usage: scripts/kstats.sh <tag> [bench.py args...]   (env passes through, e.g. MVIN_KA_STREAM=0)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
o=$root/gpurun_out/ks_$tag; mkdir -p "$o"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$o" -- python "$root/bench.py" --no-hbm-leg --no-sweep --no-cpu-baseline --steps 5 --warmup 1 --streams 1 "$@" > "$o/log" 2>&1 < /dev/null
f=$(ls "$o"/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $tag: $*"
[ -n "$f" ] && head -7 "$f" | python3 -c "
import sys, csv
for r in csv.reader(sys.stdin):
    if r[0] == 'Name': continue
    print('%-70s calls %4s avg %10.1f us  %5s%%' % (r[0][:70], r[1], float(r[3]) / 1e3, r[4]))
"
grep '^{' "$o/log" | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.1f %s  %.3f ms/step' % (d['value'], d['unit'], d['ms_per_step']))
"
