#!/usr/bin/env python3
"""Fr NTT timing: python tools/ntt_time.py [log_n ...]  (default 2^10, 2^16, 2^20, 2^24 elements in HBM, in place; median of 20)"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bls12_381_amd as bls
ctx = bls.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for log_n in ([int(a) for a in sys.argv[1:]] or (10, 16, 20, 24)):
    n = 1 << log_n
    rs = np.random.RandomState(log_n)
    a = rs.randint(0, 2**62, size=(n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64(0x0fffffffffffffff)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    ctx.fr_ntt_device(d.data_ptr(), log_n, False); torch.cuda.synchronize()
    h = hashlib.sha256(d.cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for _ in range(20):
        t = time.perf_counter(); ctx.fr_ntt_device(d.data_ptr(), log_n, False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("2^%d  %.4f ms  sha(first transform) %s" % (log_n, 1e3 * float(np.median(ts)), h))
