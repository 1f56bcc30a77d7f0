#!/bin/bash
# 1 / 2 / 4 / 8-GPU bench lines in one go (run on a multi-GPU MI355X node):
#   tools/run_scale.sh [extra bench.py flags]
# N = 1 is BASELINE configs[1] (2^20-point G1 MSM), N > 1 is BASELINE configs[3] (ONE 2^24-point MSM sharded N ways, strong
# scaling); add --weak for 2^20 points per GPU, or --workload mixed for BASELINE configs[4].
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  if [ "$N" -le "$NG" ]; then
    # (the line carries "ranks": the process group's own world size, the backend, lazy / eager RCCL communicator, devices visible -- and
    # "result_matches_identity": the folded point checked against [sum s_i k_i] G)
    python bench.py --gpus $N --no-cpu-baseline --no-extras "$@" | tail -1
    # the same sharded MSM from ONE process (the C library's device group; round 5)
    python bench.py --group $N --no-cpu-baseline --no-extras "$@" | tail -1
  else
    echo "{\"n_gpus\": $N, \"skipped\": \"only $NG GPU(s) visible\"}"
  fi
done
