#!/bin/bash
# compile the standalone wide harness and print the kernel's resource usage
cd "$(dirname "$0")/../.." && mkdir -p build && cd build && \
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -mllvm -amdgpu-sched-strategy=iterative-ilp -I../bls12_381_amd/csrc --save-temps=obj ../tools/experiments/wide_bench.hip -o wide_bench "$@" && \
S=wide_bench-hip-amdgcn-amd-amdhsa-gfx950.s && grep "\.vgpr_count\|vgpr_spill_count\|\.private_segment_fixed_size\|group_segment_fixed" $S | tr -d '\n' && echo && echo "scratch ops $(grep -c scratch_ $S)  mads $(grep -c v_mad_u64_u32 $S) lines $(wc -l < $S)"
