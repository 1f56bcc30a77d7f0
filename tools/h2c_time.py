#!/usr/bin/env python3
"""hash-to-curve timing: python tools/h2c_time.py [log_n]  (32-byte messages, both groups; host-batch entry point, so the
figure includes the 2 MB upload and the download of the points)"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bls12_381_amd as bls
from bls12_381_amd.api import _ptr, check
ctx = bls.default_context()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
rs = np.random.RandomState(1)
blob = np.frombuffer(rs.bytes(32 * n) + b"\0", dtype=np.uint8).copy()
offs = (np.arange(n + 1, dtype=np.uint64) * 32)
dst = b"QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_"
d = np.frombuffer(dst + b"\0", dtype=np.uint8).copy()
for g in (1, 2):
    out = np.zeros((n, 18 * g), dtype=np.uint64)
    fn = ctx.lib.blsgpu_g1_hash_to_curve_batch if g == 1 else ctx.lib.blsgpu_g2_hash_to_curve_batch
    ts = []
    for _ in range(6):
        a = time.perf_counter(); check(fn(ctx.h, _ptr(blob), _ptr(offs), n, _ptr(d), len(dst), 0, _ptr(out)), "h2c"); ts.append(time.perf_counter() - a)
    print("G%d n=%d  %.3f ms  sha %s" % (g, n, 1e3 * min(ts[1:]), hashlib.sha256(out.tobytes()).hexdigest()[:12]))
