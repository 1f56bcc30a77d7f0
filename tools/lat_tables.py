#!/usr/bin/env python3
"""single-call latency of a 2^20-point G1 MSM: default path vs resident window-shifted tables of several widths (tools/lat_tables.py [c ...])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bls12_381_amd as bls
from bls12_381_amd import synthetic
n = 1 << 20
ctx = bls.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
bases = ctx.bases_from_scalars(1, synthetic.scalars(n, synthetic.SEED + 1))
d_s = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).cuda(); d_o = torch.zeros(18, dtype=torch.int64, device="cuda")
def med(fn, reps=11):
    for _ in range(3): fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    return float(np.median(ts))
call = lambda: ctx.msm_device(bases, d_s.data_ptr(), n, d_o.data_ptr())
print("default", round(med(call), 3)); ref = ctx.batch_normalize(1, d_o.cpu().numpy().view(np.uint64)[None, :])[0]
ctx.set_profiling(True); call(); print(ctx.last_msm_phase_ms()); ctx.set_profiling(False)
for c in [int(a) for a in sys.argv[1:]] or [16, 18, 20]:
    bases.precompute(c)
    ms = med(call)
    same = np.array_equal(ctx.batch_normalize(1, d_o.cpu().numpy().view(np.uint64)[None, :])[0], ref)
    ctx.set_profiling(True); call(); ph = ctx.last_msm_phase_ms(); ctx.set_profiling(False)
    print("tables c=%d" % c, round(ms, 3), same, {k: round(v, 3) for k, v in ph.items()})
