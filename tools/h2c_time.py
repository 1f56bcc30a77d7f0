#!/usr/bin/env python3
"""A/B timing of the batched hash-to-curve (round 5): the split form (two lane groups per message) against one lane / lane pair per
message, 32-byte messages resident in HBM.   usage: python tools/h2c_time.py"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bls12_381_amd as bls

dev = torch.device("cuda", 0)
sync = torch.cuda.synchronize


def make(split):
    os.environ["BLSGPU_H2C_SPLIT"] = split
    try:
        c = bls.Context(0)
    finally:
        os.environ.pop("BLSGPU_H2C_SPLIT")
    return c


def med(fn, reps=7):
    fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


ctxs = {"split": make("1"), "plain": make("0")}
rows = []
for group in (1, 2):
    dst = b"BLS_SIG_BLS12381G%d_XMD:SHA-256_SSWU_RO_NUL_" % group
    d_dst = torch.from_numpy(np.frombuffer(dst, dtype=np.uint8).copy()).to(dev)
    for logn in (10, 12, 13, 14, 15, 16, 17):
        n = 1 << logn
        m = torch.from_numpy(np.random.RandomState(logn).randint(0, 256, size=n * 32, dtype=np.uint8)).to(dev)
        off = torch.arange(0, (n + 1) * 32, 32, dtype=torch.int64, device=dev)
        row = {"group": group, "log_n": logn}
        res = {}
        for name, c in ctxs.items():
            out = torch.zeros((n, 18 * group), dtype=torch.int64, device=dev)
            sync()
            f = lambda: bls._lib.check(c.lib.blsgpu_hash_to_curve_device(c.h, group, m.data_ptr(), off.data_ptr(), n, d_dst.data_ptr(), len(dst), 0, out.data_ptr()), "h2c")
            row[name + "_ms"] = round(med(f), 4)
            res[name] = out
        row["identical"] = bool(torch.equal(res["split"], res["plain"]))
        rows.append(row)
        print(row, flush=True)
print(json.dumps({"hash_to_curve": rows}))
