#!/bin/bash
# Collect the rocprofv3 evidence quoted in DESIGN.md / bench.py (run on the GPU box through gpurun):
#   gpurun_out/prof_msm        kernel trace + stats of the default bench timed region, then PMC passes (tools/pmc.sh)
#   gpurun_out/prof_pair       the same for 2^16 pairings (tools/run_pairing.py)
#   gpurun_out/prof_mml        ... and for a 2^18-term multi_miller_loop
#   gpurun_out/prof_wide       ... and for 2^8 pairings per call (the wide small-batch kernel)
# tools/summarise_profiles.py <round> turns them into profiles/<round>_*.md / .json.
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
tools/pmc.sh prof_msm python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
# the stats pass of prof_msm is replaced by one over the DEFAULT command, whose bench line is kept next to it
cd /tmp && export TMPDIR=/tmp && cd "$R"
rm -rf gpurun_out/prof_msm/stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_msm/stats -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_msm/bench_under_rocprof.json 2> /dev/null
tools/pmc.sh prof_pair python tools/run_pairing.py pairing 16 3
BLSGPU_PAIRING_LAYOUT=pair tools/pmc.sh prof_pair_lp python tools/run_pairing.py pairing 16 3
tools/pmc.sh prof_mml python tools/run_pairing.py mml 18 3
tools/pmc.sh prof_wide python tools/run_pairing.py pairing 8 20
# round 5: the prepared Miller loops (prep.hip.h)
tools/pmc.sh prof_mmlp python tools/run_pairing.py mmlp 18 3
tools/pmc.sh prof_eqp python tools/run_pairing.py eqp 16 3
BLSGPU_MML_IMPL=4 tools/pmc.sh prof_mmlq python tools/run_pairing.py mml 18 3
tools/pmc.sh prof_eq python tools/run_pairing.py equations 14 3
# the bulk-verification chain (kernel trace only): 2^14 signatures, keys in G1 and keys in G2
cd /tmp && export TMPDIR=/tmp && cd "$R"
rm -rf gpurun_out/prof_verify; mkdir -p gpurun_out/prof_verify
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_verify/mode0 -- python tools/run_verify.py 14 0 3 > gpurun_out/prof_verify/mode0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_verify/mode1 -- python tools/run_verify.py 14 1 3 > gpurun_out/prof_verify/mode1.log 2>&1
find gpurun_out/prof_msm gpurun_out/prof_pair gpurun_out/prof_pair_lp gpurun_out/prof_mml gpurun_out/prof_wide gpurun_out/prof_mmlp gpurun_out/prof_eqp gpurun_out/prof_mmlq gpurun_out/prof_eq gpurun_out/prof_verify -name "*agent_info.csv" -delete
du -sh gpurun_out/prof_*
