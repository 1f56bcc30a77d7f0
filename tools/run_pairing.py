#!/usr/bin/env python3
"""Launch the pairing kernels a few times on synthetic inputs (profiling target for tools/pmc.sh).
   usage: python tools/run_pairing.py [pairing|mml|equations|mmlp|eqp] [log2 n] [reps]
   equations: 2^n three-term equations through blsgpu_multi_miller_loop_many_device (n <= 14 with the 2^16 synthetic pairs)
   mmlp: one product of 2^n terms, every term PREPARED (a table of four points); eqp: 2^n three-term equations, two terms prepared"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bls12_381_amd as bls

what = sys.argv[1] if len(sys.argv) > 1 else "pairing"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
ctx = bls.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << (16 if what in ("equations", "eqp") else min(logn, 16))
rs = np.random.RandomState(99)
ka = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
kq = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); kq[:, 31] &= 0x3F
g1xy, _ = ctx.bases_from_scalars(1, ka).download()
g2xy, _ = ctx.bases_from_scalars(2, kq).download()
rep = max(1, (1 << logn) // n) if what != "eqp" else max(1, (3 << logn) // n)
d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev).repeat(rep, 1); d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev).repeat(rep, 1)
n = n * rep
d_gt = torch.zeros((n, 72), dtype=torch.int64, device=dev)
if what in ("equations", "eqp"):
    ne, ke = 1 << logn, 3
    assert ne * ke <= n
    d_off = torch.arange(0, (ne + 1) * ke, ke, dtype=torch.int64, device=dev)
if what in ("mmlp", "eqp"):
    table = ctx.g2_prepare(g2xy[:4].copy())
    if what == "mmlp":
        d_qi = torch.from_numpy((np.arange(n) % 4).astype(np.uint32).view(np.int32)).to(dev)
    else:
        qi = np.full(ne * ke, bls.UNPREPARED, dtype=np.uint32); qi[1::3] = 0; qi[2::3] = 1
        d_qi = torch.from_numpy(qi.view(np.int32)).to(dev)
for _ in range(reps):
    if what == "mmlp":
        ctx.multi_miller_loop_prepared_device(d_g1.data_ptr(), table, d_qi.data_ptr(), n, d_gt.data_ptr())
    elif what == "eqp":
        ctx.multi_miller_loop_prepared_many_device(d_g1.data_ptr(), table, d_qi.data_ptr(), d_off.data_ptr(), ne, ne * ke, d_gt.data_ptr(), max_seg_terms=ke, d_g2=d_g2.data_ptr())
    elif what == "equations":
        ctx.multi_miller_loop_many_device(d_g1.data_ptr(), d_g2.data_ptr(), d_off.data_ptr(), ne, ne * ke, d_gt.data_ptr(), max_seg_terms=ke)
    elif what == "pairing":
        bls._lib.check(ctx.lib.blsgpu_pairing_batch_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, n, d_gt.data_ptr()), "pairing")
    else:
        bls._lib.check(ctx.lib.blsgpu_multi_miller_loop_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, n, d_gt.data_ptr()), "mml")
    torch.cuda.synchronize()
