#!/bin/bash
# rocprofv3 evidence for one command:  tools/pmc.sh <outdir-under-gpurun_out> <command...>
#   <outdir>/stats        --kernel-trace --stats
#   <outdir>/pmc_<i>      one --pmc pass per counter group below (kernel-trace only, as MI355X_MICROARCH.md prescribes)
# Summarise with tools/pmc_summary.py <outdir> <kernel-substring>.
OUT="$1"; shift
R="${GRAFT_REPO_ROOT:-$PWD}"
cd /tmp && export TMPDIR=/tmp
cd "$R"
rm -rf "gpurun_out/$OUT"; mkdir -p "gpurun_out/$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/$OUT/stats" -- "$@" > /dev/null 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "gpurun_out/$OUT/pmc_$i" -- "$@" > "gpurun_out/$OUT/pmc_$i.log" 2>&1
  i=$((i+1))
done
find "gpurun_out/$OUT" -name "*agent_info.csv" -delete
