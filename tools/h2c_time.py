#!/usr/bin/env python3
"""median time of hash_to_curve / checked decoding on the bench's shapes: python tools/h2c_time.py   (BLSGPU_LIB_PATH selects the library)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bls12_381_amd as bls
from bls12_381_amd import synthetic
ctx = bls.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
P = lambda t: t.data_ptr()
sync = torch.cuda.synchronize
def med(fn, warm=3, reps=11):
    for _ in range(warm): fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync(); ts.append(1e3 * (time.perf_counter() - t))
    return float(np.median(ts))
for g, logn in ((2, 16), (2, 17), (1, 16), (1, 17)):
    n = 1 << logn
    hm = torch.from_numpy(np.random.RandomState(99).randint(0, 256, size=n * 32, dtype=np.uint8)).cuda()
    ho = torch.arange(0, (n + 1) * 32, 32, dtype=torch.int64, device="cuda")
    dst = b"BLS_SIG_BLS12381G%d_XMD:SHA-256_SSWU_RO_NUL_" % g
    hd = torch.from_numpy(np.frombuffer(dst, dtype=np.uint8).copy()).cuda()
    out = torch.zeros((n, 18 * g), dtype=torch.int64, device="cuda")
    print("hash_to_G%d 2^%d  %.3f ms" % (g, logn, med(lambda: bls._lib.check(ctx.lib.blsgpu_hash_to_curve_device(ctx.h, g, P(hm), P(ho), n, P(hd), len(dst), 0, P(out)), "h2c"))))
n16 = 1 << 16
for g in (1, 2):
    xy, _ = ctx.bases_from_scalars(g, synthetic.scalars(n16, 99 if g == 1 else 100)).download()
    enc = torch.from_numpy(ctx.points_to_bytes(g, xy, None, compressed=True)).cuda()
    d_cx = torch.zeros((n16, 12 * g), dtype=torch.int64, device="cuda"); d_ci = torch.zeros(n16, dtype=torch.uint8, device="cuda"); d_ck = torch.zeros(n16, dtype=torch.uint8, device="cuda")
    print("decode G%d 2^16  %.3f ms" % (g, med(lambda: ctx.points_from_bytes_device(g, P(enc), n16, P(d_cx), P(d_ci), P(d_ck), compressed=True, checked=True))))
