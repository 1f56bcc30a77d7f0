"""Phase times of one 2^20-point G1 MSM with and without resident window-shifted tables (experiment, round 5).
   usage: python tools/experiments/tables_phases.py [window bits ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bls12_381_amd as bls
from bls12_381_amd import synthetic
dev = torch.device("cuda", 0)
n = 1 << 20
ctx = bls.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
kb = synthetic.scalars(n, synthetic.SEED + 1); sb = synthetic.scalars(n, synthetic.SEED)
d_s = torch.from_numpy(sb).to(dev)
d_o = [torch.zeros(18, dtype=torch.int64, device=dev) for _ in range(4)]
def run(bases, tag):
    ctx.set_profiling(True)
    for _ in range(3):
        ctx.msm_device(bases, d_s.data_ptr(), n, d_o[0].data_ptr())
        ph = ctx.last_msm_phase_ms()
    ctx.set_profiling(False)
    ctx.set_pipelining(True)
    for i in range(8): ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(40): ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 40
    ctx.set_pipelining(False)
    print(tag, {k: round(v, 3) for k, v in ph.items()}, "pipelined ms/MSM %.3f" % (dt * 1e3), flush=True)
bases = ctx.bases_from_scalars(1, kb)
run(bases, "plain")
for wb in ([int(a) for a in sys.argv[1:]] or [16, 18, 20]):
    b2 = ctx.bases_from_scalars(1, kb)
    b2.precompute(wb)
    run(b2, "tables c=%d" % wb)
    b2.free()
