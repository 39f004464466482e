#!/bin/bash
# Run ON the GPU box: per-dispatch timeline of rank 0's step when the default batch is cut into W ranks
# (bench.py --emulate-world W).  usage: scripts/trace_world.sh <W> [rows]
W=${1:-8}; N=${2:-40}; shift; shift   # further arguments go to bench.py
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
o=$root/gpurun_out/tw_$W; mkdir -p "$o"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$o" -- python "$root/bench.py" --emulate-world "$W" "$@" --no-hbm-leg --no-sweep --no-cpu-baseline --steps 6 --warmup 2 > "$o/log" 2>&1 < /dev/null
f=$(ls "$o"/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 "$root/scripts/timeline.py" "$f" "$N" > "$o/timeline.txt"
grep '^{' "$o/log" | tail -1 > "$o/bench.json"
