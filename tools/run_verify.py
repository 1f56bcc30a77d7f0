#!/usr/bin/env python3
"""Run the bulk-verification chain (blsgpu_bls_verify_batch_device) a few times on synthetic signatures: profiling target.
   usage: python tools/run_verify.py [log2 n = 14] [mode = 0] [reps = 3]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bls12_381_amd as bls
from bls12_381_amd import synthetic

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
ctx = bls.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << logn
gk, gs = (1, 2) if mode == 0 else (2, 1)
dst = b"BLS_SIG_BLS12381G%d_XMD:SHA-256_SSWU_RO_NUL_" % gs
skb = synthetic.scalars(n, 4242)
msgs = np.random.RandomState(7).randint(0, 256, size=(n, 32), dtype=np.uint8)
pkxy, pkinf = ctx.bases_from_scalars(gk, skb).download()
hxy, hinf = ctx.batch_normalize(gs, ctx.hash_to_curve(gs, [m.tobytes() for m in msgs], dst))
sgxy, sginf = ctx.batch_normalize(gs, ctx.mul_batch(gs, hxy, hinf, skb))
pk_b = ctx.points_to_bytes(gk, pkxy, pkinf, compressed=True).copy()
sg_b = ctx.points_to_bytes(gs, sgxy, sginf, compressed=True).copy()
d_pk, d_sg, d_m = torch.from_numpy(pk_b).to(dev), torch.from_numpy(sg_b).to(dev), torch.from_numpy(msgs).to(dev)
d_o = torch.arange(0, (n + 1) * 32, 32, dtype=torch.int64, device=dev)
d_d = torch.from_numpy(np.frombuffer(dst, dtype=np.uint8).copy()).to(dev)
d_v = torch.zeros(n, dtype=torch.uint8, device=dev)
for _ in range(reps):
    t = time.perf_counter()
    ctx.bls_verify_batch_device(mode, d_pk.data_ptr(), d_sg.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), n, d_d.data_ptr(), len(dst), d_v.data_ptr())
    torch.cuda.synchronize()
    print("verify 2^%d mode %d: %.3f ms, valid %d" % (logn, mode, 1e3 * (time.perf_counter() - t), int((d_v == 1).sum())), flush=True)
