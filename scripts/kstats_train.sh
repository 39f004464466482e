#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of scripts/bench_train.py (training step), every kernel printed.
# usage: scripts/kstats_train.sh <tag> [bench_train.py args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
o=$root/gpurun_out/kt_$tag; mkdir -p "$o"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$o" -- python "$root/scripts/bench_train.py" --steps 20 "$@" > "$o/log" 2>&1 < /dev/null
f=$(ls "$o"/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $tag: $*"
[ -n "$f" ] && cp "$f" "$root/gpurun_out/kt_${tag}_kernel_stats.csv" && python3 - "$f" <<'PY'
import sys, csv
rows = [r for r in csv.reader(open(sys.argv[1])) if r and r[0] != 'Name']
steps = 22.0
tot = sum(float(r[2]) for r in rows)
print('total kernel time per step: %.1f us over %d launches' % (tot / steps / 1e3, sum(int(r[1]) for r in rows) / steps))
for r in rows[:28]:
    print('%-64s calls/step %5.1f avg %8.1f us  per step %8.1f us %5s%%' % (r[0][:64], int(r[1]) / steps, float(r[3]) / 1e3, float(r[2]) / steps / 1e3, r[4]))
PY
grep '^{' "$o/log" | tail -1
