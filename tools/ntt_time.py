#!/usr/bin/env python3
"""A/B timing of the Fr transform (round 5): the column-tile path (k_fr_cols + k_fr_tile) against the stage-pair passes of rounds 2-4
(BLSGPU_NTT_IMPL=stage), 2^16 .. 2^24 elements resident in HBM; outputs compared limb for limb.
   usage: python tools/ntt_time.py [log2 sizes ...]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bls12_381_amd as bls

sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 20, 22, 24]
dev = torch.device("cuda", 0)
sync = torch.cuda.synchronize


def make(impl):
    if impl:
        os.environ["BLSGPU_NTT_IMPL"] = impl
    try:
        c = bls.Context(0)
    finally:
        os.environ.pop("BLSGPU_NTT_IMPL", None)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    return c


def med(fn, reps=20):
    fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


only_new = bool(os.environ.get("NTT_TIME_ONLY_DEFAULT"))            # profiling target (tools/ntt_traffic.py): the default path alone
new = make(None)
old = new if only_new else make("stage")
rows = []
for log_n in sizes:
    n = 1 << log_n
    rs = np.random.RandomState(log_n)
    raw = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); raw[:, 31] &= 0x3F
    x = torch.from_numpy(raw.view(np.int64).reshape(n, 4).copy()).to(dev)
    a, b = x.clone(), x.clone()
    new.fr_ntt_device(a.data_ptr(), log_n); old.fr_ntt_device(b.data_ptr(), log_n); sync()
    same = bool(torch.equal(a, b))
    b0 = b.clone()
    tn = med(lambda: new.fr_ntt_device(a.data_ptr(), log_n))
    to = med(lambda: old.fr_ntt_device(b.data_ptr(), log_n))
    row = {"log_n": log_n, "column_tiles_ms": round(tn, 4), "stage_passes_ms": round(to, 4), "identical": same}
    for shape in os.environ.get("NTT_TIME_SHAPES", "").split():          # "tile log2,max depth per pass,block" variants of the column pass
        os.environ["BLSGPU_NTT_COLS"] = shape; os.environ["BLSGPU_NTT_IMPL"] = "cols"      # read when the context is created (csrc/diag.h)
        shaped = bls.Context(0)
        shaped.set_stream(torch.cuda.current_stream().cuda_stream)
        os.environ.pop("BLSGPU_NTT_COLS"); os.environ.pop("BLSGPU_NTT_IMPL")
        a.copy_(x); sync(); shaped.fr_ntt_device(a.data_ptr(), log_n); sync()
        ok = bool(torch.equal(a, b0))
        row["cols " + shape] = (round(med(lambda: shaped.fr_ntt_device(a.data_ptr(), log_n)), 4), ok)
        shaped.close()
    rows.append(row)
    print(rows[-1], flush=True)
print(json.dumps({"fr_ntt": rows}))
