#!/bin/bash
# Run ON the GPU box: SQ / TA counters of the grouped key-addressing kernel at dim 16 (scripts/bench_ka_grouped.py), one
# rocprofv3 --pmc pass per counter group, reduced to per-kernel means.  usage: scripts/pmc_ka16.sh [dataset]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_ka16; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TA_TA_BUSY_sum TCP_GATE_EN1_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/g$i" -- python "$root/scripts/bench_ka_grouped.py" 16 "$@" > "$out/g$i.log" 2>&1
done
cd "$root"; python scripts/pmc_kernel_means.py "$out" key_addr_wave_kernel
