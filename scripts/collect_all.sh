#!/bin/bash
# Run ON the GPU box (via gpurun): everything DESIGN.md section 4 quotes, in one call.
#   scripts/collect_profiles.sh <tag>   kernel stats + FETCH/WRITE PMC passes, timed region and HBM leg
#   scripts/pmc_counters.sh <tag>       SQ / TCP / TCC counter groups
#   scripts/kstats.sh                   kernel stats of the per-pair feed and of configs C2 / C4 / C5
#   scripts/bench_variants.sh           one bench.py JSON line per variant
#   python bench.py                     the driver's default line
# then scripts/archive_profiles.sh <tag> <prefix> [round] (in the build container) copies the summaries into profiles/<round>/.
tag=${1:-latest}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
timeout 900 scripts/collect_profiles.sh $tag > gpurun_out/collect_$tag.log 2>&1 < /dev/null
timeout 900 scripts/pmc_counters.sh $tag > gpurun_out/pmc_counters_$tag.log 2>&1 < /dev/null
{
  scripts/kstats.sh ${tag}_c3_pairs --feed pairs
  scripts/kstats.sh ${tag}_c2 --dataset MovieLens-1M --dim 32 --fanout 16 --batch 524288
  scripts/kstats.sh ${tag}_c4 --dataset amazon-book_20core --dim 64 --fanout 64 --batch 32768
  scripts/kstats.sh ${tag}_c5 --dataset amazon-book_20core --dim 128 --hop 3 --fanout 128 --table-dtype bf16 --batch 64 --steps 3
  scripts/kstats.sh ${tag}_shipped --dim 16 --fanout 8
  scripts/kstats.sh ${tag}_shipped_b512 --dim 16 --fanout 8 --batch 512 --feed pairs --steps 50
} > gpurun_out/kstats_$tag.log 2>&1 < /dev/null
timeout 1500 scripts/bench_variants.sh > gpurun_out/bench_variants_$tag.log 2>&1 < /dev/null
timeout 600 python bench.py 2> gpurun_out/bench_default_$tag.err | grep '^{' | tail -1 > gpurun_out/bench_default_$tag.json
[ -x scripts/micro/dma_probe.bin ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o scripts/micro/dma_probe.bin scripts/micro/dma_probe.hip
timeout 120 scripts/micro/dma_probe.bin > gpurun_out/dma_probe_$tag.txt 2>&1 < /dev/null
tail -3 gpurun_out/collect_$tag.log; tail -30 gpurun_out/bench_variants_$tag.log; cat gpurun_out/kstats_$tag.log | grep -E "==|avg|value"
