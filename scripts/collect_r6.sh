#!/bin/bash
# Run ON the GPU box (via gpurun): the round-6 evidence in one call -- scripts/collect_all.sh <tag> plus the counters / kernel
# stats of the kernels round 6 added (flash key addressing: SQ / TCC counters and kernel stats; training step kernel stats).
tag=${1:-r6}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
scripts/collect_all.sh $tag > gpurun_out/collect_all_$tag.log 2>&1 < /dev/null
timeout 600 scripts/pmc_flash.sh ${tag}_flash > gpurun_out/pmc_flash_$tag.log 2>&1 < /dev/null
{
  scripts/kstats_train.sh ${tag}_b512 --batch 512
  scripts/kstats_train.sh ${tag}_b4096 --batch 4096
} > gpurun_out/kstats_train_$tag.log 2>&1 < /dev/null
SCRIPT=bench_agg.py KERNEL=score_l2_folded timeout 600 scripts/pmc_flash.sh ${tag}_fold > gpurun_out/pmc_fold_$tag.log 2>&1 < /dev/null
timeout 300 python scripts/bench_agg.py > gpurun_out/bench_agg_$tag.txt 2>&1 < /dev/null
for d in 14 8 6 0; do echo "MVIN_FOLD_DBG=$d"; MVIN_FOLD_DBG=$d timeout 200 python scripts/bench_agg.py 2>&1 | grep "score_l2_folded"; done > gpurun_out/fold_phases_$tag.txt 2>&1 < /dev/null
timeout 300 python scripts/bench_ka_flash.py > gpurun_out/bench_ka_flash_$tag.txt 2>&1 < /dev/null
timeout 300 python scripts/trace_flash.py > gpurun_out/trace_flash_$tag.txt 2>&1 < /dev/null
tail -5 gpurun_out/collect_all_$tag.log; tail -12 gpurun_out/pmc_flash_$tag.log; grep "total kernel\|^{" gpurun_out/kstats_train_$tag.log; grep "us$" gpurun_out/bench_ka_flash_$tag.txt; cat gpurun_out/bench_agg_$tag.txt gpurun_out/fold_phases_$tag.txt; tail -30 gpurun_out/pmc_fold_$tag.log
