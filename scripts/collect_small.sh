#!/bin/bash
# Run ON the GPU box: everything DESIGN.md quotes about the single-launch pass (mvin_score_small_fwd) --
#   wall / event timings per batch size and group size against the multi-launch schedule      -> small_<tag>_sweep.txt
#   true kernel durations (rocprofv3) of the launch at 512 / 1 024 pairs, product group size   -> small_<tag>_kernel_stats.txt
#   cycle stamps of workgroup 0 per stage (MVIN_SMALL_DBG=99)                                   -> small_<tag>_trace.txt
#   kernel list of the multi-launch schedule at 4 096 / 16 384 pairs                            -> small_<tag>_multi_launch_kernels.txt
tag=${1:-latest}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
python scripts/bench_small_batch.py --sizes 512,1024,2048,4096,16384 --groups 0,1,2,4,8 2>&1 | grep -v "^/opt" > gpurun_out/small_${tag}_sweep.txt
{ scripts/kstats_small.sh ${tag}_512 --sizes 512 --groups 0 --only enc --no-multi
  scripts/kstats_small.sh ${tag}_1024 --sizes 1024 --groups 0 --only enc --no-multi
  scripts/kstats_small.sh ${tag}_512_plain --sizes 512 --groups 0 --only plain --no-multi; } 2>&1 | grep -E "==|score_small" > gpurun_out/small_${tag}_kernel_stats.txt
{ echo "# B = 512, one pair per workgroup, encoded adjacency"; python scripts/trace_small.py 512 1
  echo "# B = 1024, two pairs per workgroup"; python scripts/trace_small.py 1024 2
  echo "# B = 512, plain adjacency"; python scripts/trace_small.py 512 1 plain; } 2>&1 | grep -v "^/opt" > gpurun_out/small_${tag}_trace.txt
{ scripts/kstats_multi.sh ${tag}_4096 4096; scripts/kstats_multi.sh ${tag}_16384 16384; } 2>&1 | grep -v "score_small\|^/opt" > gpurun_out/small_${tag}_multi_launch_kernels.txt
tail -3 gpurun_out/small_${tag}_sweep.txt; cat gpurun_out/small_${tag}_kernel_stats.txt
