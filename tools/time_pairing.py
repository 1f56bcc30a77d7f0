#!/usr/bin/env python3
"""median time of the pairing-family calls on synthetic inputs: python tools/time_pairing.py   (BLSGPU_LIB_PATH selects the library)
   2^16 pairings, one 2^18-term multi_miller_loop, 2^14 three-term equations (+ final exponentiation), 2^14 final exponentiations"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bls12_381_amd as bls
dev = torch.device("cuda", 0)
ctx = bls.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << 16
rs = np.random.RandomState(99)
ka = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
kq = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); kq[:, 31] &= 0x3F
g1xy, _ = ctx.bases_from_scalars(1, ka).download()
g2xy, _ = ctx.bases_from_scalars(2, kq).download()
d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev)
d_gt = torch.zeros((n, 72), dtype=torch.int64, device=dev)
P = lambda t: t.data_ptr()
sync = torch.cuda.synchronize
def med(fn, warm=2, reps=9):
    for _ in range(warm): fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync(); ts.append(1e3 * (time.perf_counter() - t))
    return float(np.median(ts)), float(min(ts))
print("pairing 2^16        median %.3f ms  min %.3f" % med(lambda: ctx.pairing_batch_device(P(d_g1), P(d_g2), n, P(d_gt))))
d4g1 = d_g1.repeat(4, 1); d4g2 = d_g2.repeat(4, 1); d_one = torch.zeros(72, dtype=torch.int64, device=dev)
for m in (49152, 40000, 32768, 24576):
    print("pairing %6d      median %.3f ms  min %.3f" % ((m,) + med(lambda: ctx.pairing_batch_device(P(d_g1), P(d_g2), m, P(d_gt)))))
print("mml 2^18            median %.3f ms  min %.3f" % med(lambda: ctx.multi_miller_loop_device(P(d4g1), P(d4g2), 4 * n, P(d_one))))
ne = 1 << 14
off = torch.arange(0, 3 * ne + 1, 3, dtype=torch.int64, device=dev)
d_eq = torch.zeros((ne, 72), dtype=torch.int64, device=dev)
print("equations 2^14 x 3  median %.3f ms  min %.3f" % med(lambda: ctx.multi_miller_loop_many_device(P(d_g1), P(d_g2), P(off), ne, 3 * ne, P(d_eq), final_exp=True)))
print("final exp 2^14      median %.3f ms  min %.3f" % med(lambda: ctx.final_exponentiation_device(P(d_gt), ne, P(d_eq))))
# prepared forms: one product of 2^18 prepared terms (a table of four points), 2^14 equations with two prepared terms
try:
    tab = ctx.g2_prepare(g2xy[:4])
    qidx = torch.from_numpy((np.arange(4 * n) % 4).astype(np.uint32)).to(dev)
    print("mml prepared 2^18   median %.3f ms  min %.3f" % med(lambda: ctx.multi_miller_loop_prepared_device(P(d4g1), tab, P(qidx), 4 * n, P(d_one))))
except Exception as ex:
    print("prepared legs skipped:", ex)
# single-layer launches (one wavefront per SIMD): 2^14 pairings, 2^14 three-term equations with two prepared terms
try:
    print("pairing 2^14        median %.3f ms  min %.3f" % med(lambda: ctx.pairing_batch_device(P(d_g1), P(d_g2), 1 << 14, P(d_gt))))
    qi = np.full(3 * ne, bls.UNPREPARED, dtype=np.uint32); qi[1::3] = 0; qi[2::3] = 1
    d_qi3 = torch.from_numpy(qi.view(np.int32)).to(dev)
    print("prepared eq 2^14    median %.3f ms  min %.3f" % med(lambda: ctx.multi_miller_loop_prepared_many_device(P(d_g1), tab, P(d_qi3), P(off), ne, 3 * ne, P(d_eq), max_seg_terms=3, d_g2=P(d_g2))))
except Exception as ex:
    print("single-layer legs skipped:", ex)
