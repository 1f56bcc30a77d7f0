#!/usr/bin/env python3
"""Latency of the pairing path at small batch sizes (one final exponentiation, one pairing, ...):  python tools/lat_small.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bls12_381_amd as bls
ctx = bls.default_context()
rs = np.random.RandomState(5)
def scal(n):
    a = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x3F
    return a
for n in (1, 2, 32, 1024):
    g1, i1 = ctx.bases_from_scalars(1, scal(n)).download()
    g2, i2 = ctx.bases_from_scalars(2, scal(n)).download()
    ml = ctx.miller_loop_batch(g1, i1, g2, i2)
    def t(f, reps=5):
        f(); ts = []
        for _ in range(reps):
            a = time.perf_counter(); f(); ts.append(time.perf_counter() - a)
        return 1e3 * float(np.median(ts))
    print("n=%d  miller_loop_batch %.2f ms  final_exponentiation_batch %.2f ms  pairing_batch %.2f ms  multi_miller_loop %.2f ms" % (
        n, t(lambda: ctx.miller_loop_batch(g1, i1, g2, i2)), t(lambda: ctx.final_exponentiation_batch(ml)), t(lambda: ctx.pairing_batch(g1, i1, g2, i2)),
        t(lambda: ctx.multi_miller_loop(g1, i1, g2, i2))))
