#!/usr/bin/env python3
"""per-kernel HIP-event times of the 2^20-point G1 MSM (single calls and a pipelined run): python tools/msm_kernels.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bls12_381_amd as bls
from bls12_381_amd import synthetic
n = 1 << 20
ctx = bls.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
bases = ctx.bases_from_scalars(1, synthetic.scalars(n, synthetic.SEED + 1))
d_s = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).cuda(); d_o = torch.zeros((8, 18), dtype=torch.int64, device="cuda")
call = lambda i=0: ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i % 8].data_ptr())
for _ in range(3): call(); torch.cuda.synchronize()
ts = []
for _ in range(9):
    t = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
print("single call ms", round(float(np.median(ts)), 3))
ctx.kernel_timing(True)
for _ in range(5): call(); torch.cuda.synchronize()
rep = ctx.kernel_timing_report(); ctx.kernel_timing(False)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]): print("   %-40s n=%d avg %.3f ms" % (k[:40], v["launches"], v["total_ms"] / v["launches"]))
ctx.set_pipelining(True)
for i in range(6): call(i)
ctx.join(); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(40): call(i)
ctx.join(); torch.cuda.synchronize()
print("pipelined ms/step", round(1e3 * (time.perf_counter() - t) / 40, 3))
# clocks and power under the pipelined MSM (sysfs, as bench.py samples them under the pairing kernel)
import bench
st = bench.clocks_under_load(lambda: [call(i) for i in range(8)], lambda: (ctx.join(), torch.cuda.synchronize()), launches=40)
print("under pipelined msm:", st)
ctx.set_pipelining(False)
