#!/bin/bash
# Collect the rocprofv3 evidence quoted in DESIGN.md / bench.py (run on the GPU box through gpurun):
#   gpurun_out/prof/stats      kernel trace + stats of the default bench workload
#   gpurun_out/prof/pmc_*      separate PMC passes (kernel-trace only, as the MI355X guide prescribes)
# tools/summarise_profiles.py turns them into profiles/r01_*.md / .json.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
# the stats pass traces the default timed region (100 steps + 5 warm-up) and prints the bench line of that very run
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof/bench_under_rocprof.json 2> /dev/null
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=gpurun_out/prof/pmc_$(echo $c | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- $B > /dev/null 2>&1
done
# pairing kernel
for c in "" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  if [ -z "$c" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pair_stats -- python tools/quick_pair3.py > /dev/null 2>&1
  else rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof/pair_pmc_$(echo $c | cut -d" " -f1) -- python tools/quick_pair3.py > /dev/null 2>&1; fi
done
find gpurun_out/prof -name "*agent_info.csv" -delete
ls gpurun_out/prof
