#!/usr/bin/env python3
"""time of blsgpu_fp12_product_device for n values (the fold after multi_miller_loop) -- development aid"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bls12_381_amd as bls
dev = torch.device("cuda", 0)
ctx = bls.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n0 = 1024
rs = np.random.RandomState(5)
ka = rs.randint(0, 256, size=(n0, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
g1, _ = ctx.bases_from_scalars(1, ka).download(); g2, _ = ctx.bases_from_scalars(2, ka).download()
d_g1 = torch.from_numpy(g1.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2.view(np.int64)).to(dev)
d_ml = torch.zeros((n0, 72), dtype=torch.int64, device=dev)
ctx.miller_loop_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), n0, d_ml.data_ptr())
d_out = torch.zeros(72, dtype=torch.int64, device=dev)
for n in (2, 3, 8, 16, 64, 128, 1024, 8192, 65536):
    src = d_ml.repeat((n + n0 - 1) // n0, 1)[:n].contiguous()
    def run():
        bls._lib.check(ctx.lib.blsgpu_fp12_product_device(ctx.h, src.data_ptr(), n, d_out.data_ptr()), "prod")
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); run(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("fp12 product of", n, "values:", round(1e3 * best, 3), "ms")
