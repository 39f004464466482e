#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 evidence for the bench.py numbers.
#   A. the metric workload (timed region of the default bench.py; --no-hbm-leg --no-sweep so that the
#      kernel-name averages describe ONE regime):
#        kernel trace + stats        -> gpurun_out/prof_<tag>/stats
#        PMC passes, one counter each-> gpurun_out/prof_<tag>/pmc_<COUNTER>  (never combined with
#        sys/hip/hsa traces; FETCH_SIZE and WRITE_SIZE per MI355X_MICROARCH.md's HBM recipe)
#   B. the HBM-bound leg of the dominant kernel (`bench.py --hbm-leg-only`: 4 GB table), same passes
#        -> gpurun_out/prof_<tag>/hbm_stats, hbm_pmc_<COUNTER>
#   C. scripts/summarize_pmc.py      -> pmc.json (A) and pmc_hbm_leg.json (B)
# usage: scripts/collect_profiles.sh <tag> [bench.py args...]
set -u
tag=${1:-latest}; shift || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# --streams 1: the per-kernel durations and counters are those of kernels running ALONE (the default line interleaves independent steps on
# two streams: kernels of different steps overlap, and rocprofv3 then reports inflated, mixed per-kernel figures -- 2.03 ms for the packed
# kernel against 1.69 ms un-overlapped); bench.py times the kernel the same way (single-stream repeat after the clock stops)
bargs="--steps 5 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-sweep --streams 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$root/bench.py" $bargs > "$out/bench_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- python "$root/bench.py" $bargs > "$out/bench_pmc_$c.log" 2>&1
done
hargs="--hbm-leg-only --steps 10 --warmup 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/hbm_stats" -- python "$root/bench.py" $hargs > "$out/bench_hbm_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/hbm_pmc_$c" -- python "$root/bench.py" $hargs > "$out/bench_hbm_pmc_$c.log" 2>&1
done
# B'. the same for the PROJECTED-TABLES instance (the one the default line's timed steps launch): 4 M-row table, 3 x 1.02 GB of tables
pargs="--hbm-leg-only --hbm-leg-form prj --steps 10 --warmup 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/hbm_prj_stats" -- python "$root/bench.py" $pargs > "$out/bench_hbm_prj_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/hbm_prj_pmc_$c" -- python "$root/bench.py" $pargs > "$out/bench_hbm_prj_pmc_$c.log" 2>&1
done
cd "$root"
python scripts/summarize_pmc.py "$out" $bargs
python scripts/summarize_pmc.py --hbm-leg --prj "$out" $pargs
f=$(ls "$out"/hbm_prj_stats/*/*kernel_stats.csv | head -1); cp "$f" "$out/hbm_leg_prj_kernel_stats.csv"
grep '^{' "$out/bench_hbm_prj_stats.log" | tail -1 > "$out/hbm_leg_prj_bench.json"
python scripts/summarize_pmc.py --hbm-leg "$out" $hargs
f=$(ls "$out"/stats/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"
f=$(ls "$out"/hbm_stats/*/*kernel_stats.csv | head -1); cp "$f" "$out/hbm_leg_kernel_stats.csv"
grep '^{' "$out/bench_stats.log" | tail -1 > "$out/bench.json"
grep '^{' "$out/bench_hbm_stats.log" | tail -1 > "$out/hbm_leg_bench.json"
head -8 "$out/kernel_stats.csv" | cut -c1-160
head -4 "$out/hbm_leg_kernel_stats.csv" | cut -c1-160
head -4 "$out/hbm_leg_prj_kernel_stats.csv" | cut -c1-160
