#!/bin/bash
# Build container: copy the summaries scripts/collect_all.sh <tag> left under gpurun_out/ into profiles/<round>/<prefix>_*
# (gpurun_out/ is scratch; profiles/ is what is committed and judged).  usage: archive_profiles.sh <tag> <prefix> [round=r6]
tag=$1; p=$2; round=${3:-r6}
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root"
g=gpurun_out; d=profiles/$round; mkdir -p $d
cp $g/prof_$tag/kernel_stats.csv $d/${p}_c3_B524288_kernel_stats.csv
cp $g/prof_$tag/bench.json $d/${p}_c3_B524288_bench.json
cp $g/prof_$tag/pmc.json $d/${p}_c3_B524288_pmc.json
cp $g/prof_$tag/hbm_leg_kernel_stats.csv $d/${p}_hbm_leg_kernel_stats.csv
cp $g/prof_$tag/hbm_leg_bench.json $d/${p}_hbm_leg_bench.json
cp $g/prof_$tag/pmc_hbm_leg.json $d/${p}_hbm_leg_pmc.json
cp $g/prof_$tag/hbm_leg_prj_kernel_stats.csv $d/${p}_hbm_leg_prj_kernel_stats.csv
cp $g/prof_$tag/hbm_leg_prj_bench.json $d/${p}_hbm_leg_prj_bench.json
cp $g/prof_$tag/pmc_hbm_leg_prj.json $d/${p}_hbm_leg_prj_pmc.json
cp $g/prof_$tag/pmc_hbm_leg_prj.json profiles/pmc_hbm_leg_prj.json
cp $g/prof_$tag/pmc.json profiles/pmc_latest.json
cp $g/prof_$tag/pmc_hbm_leg.json profiles/pmc_hbm_leg.json
cp $g/pmc_$tag/counters.json $d/${p}_c3_sq_tcp_tcc_counters.json
for v in c3_pairs c2 c4 c5 shipped shipped_b512; do
  f=$(ls $g/ks_${tag}_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $d/${p}_${v}_kernel_stats.csv
done
cp $g/bench_variants.jsonl $d/${p}_bench_variants.jsonl
cp $g/bench_default_$tag.json $d/${p}_default_bench.json
cp $g/dma_probe_$tag.txt $d/${p}_gather_ceiling_dma_probe.txt
# round 6: the flash key-addressing kernel, the folded-tail kernel and the forms of the two deepest levels (scripts/collect_r6.sh)
cp $g/pmc_${tag}_flash/counters.json $d/${p}_flash_keyaddr_sq_tcc_counters.json 2>/dev/null
f=$(ls $g/pmc_${tag}_flash/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $d/${p}_flash_keyaddr_kernel_stats.csv
cp $g/pmc_${tag}_fold/counters.json $d/${p}_folded_kernel_sq_tcc_counters.json 2>/dev/null
f=$(ls $g/pmc_${tag}_fold/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $d/${p}_two_level_forms_kernel_stats.csv
cp $g/bench_agg_$tag.txt $d/${p}_two_level_forms_microbench.txt 2>/dev/null
cp $g/fold_phases_$tag.txt $d/${p}_folded_kernel_phase_ablation.txt 2>/dev/null
cp $g/bench_ka_flash_$tag.txt $d/${p}_keyaddr_forms_microbench.txt 2>/dev/null
cp $g/trace_flash_$tag.txt $d/${p}_flash_keyaddr_stage_trace.txt 2>/dev/null
for b in b512 b4096; do
  [ -f $g/kt_${tag}_${b}_kernel_stats.csv ] && cp $g/kt_${tag}_${b}_kernel_stats.csv $d/${p}_train_${b}_kernel_stats.csv
done
ls -la $d | grep " ${p}_"
