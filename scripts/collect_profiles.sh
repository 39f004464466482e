#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 evidence for the default bench.py workload.
#   1. kernel trace + stats           -> gpurun_out/prof_<tag>/stats
#   2. PMC passes, one counter each   -> gpurun_out/prof_<tag>/pmc_<COUNTER>   (never combined with
#      sys/hip/hsa traces; FETCH_SIZE and WRITE_SIZE per MI355X_MICROARCH.md's HBM recipe)
#   3. scripts/summarize_pmc.py       -> gpurun_out/prof_<tag>/pmc.json
# usage: scripts/collect_profiles.sh <tag> [bench.py args...]
set -u
tag=${1:-latest}; shift || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bargs="--steps 5 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$root/bench.py" $bargs > "$out/bench_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- python "$root/bench.py" $bargs > "$out/bench_pmc_$c.log" 2>&1
done
cd "$root"
python scripts/summarize_pmc.py "$out" $bargs
f=$(ls "$out"/stats/*/*kernel_stats.csv | head -1)
cp "$f" "$out/kernel_stats.csv"
grep '^{' "$out/bench_stats.log" | tail -1 > "$out/bench.json"
head -12 "$out/kernel_stats.csv" | cut -c1-160
