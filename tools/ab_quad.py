#!/usr/bin/env python3
"""A/B of the lane-pair, quad and wide pairing kernels inside ONE build (contexts created with BLSGPU_PAIRING_LAYOUT=pair|quad|wide):
Miller loops, pairings and final exponentiations at several batch sizes, outputs hashed for bit-equality.

    python tools/ab_quad.py [log_n ...]          (default 16 14)"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bls12_381_amd as bls
    dev = torch.device("cuda", 0)
    logs = [int(a) for a in sys.argv[1:]] or [16, 14]
    nmax = 1 << max(logs)
    ctxs = {}
    for name in ("pair", "quad", "wide"):
        os.environ["BLSGPU_PAIRING_LAYOUT"] = name
        ctxs[name] = bls.Context(0)
        ctxs[name].set_stream(torch.cuda.current_stream().cuda_stream)
    os.environ.pop("BLSGPU_PAIRING_LAYOUT")
    rs = np.random.RandomState(99)
    ka = rs.randint(0, 256, size=(nmax, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
    kq = rs.randint(0, 256, size=(nmax, 32), dtype=np.uint8); kq[:, 31] &= 0x3F
    c0 = ctxs["pair"]
    g1xy, _ = c0.bases_from_scalars(1, ka).download()
    g2xy, _ = c0.bases_from_scalars(2, kq).download()
    d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev)
    d_f = torch.zeros((nmax, 72), dtype=torch.int64, device=dev)
    d_gt = torch.zeros((nmax, 72), dtype=torch.int64, device=dev)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        return 1e3 * best

    for lg in logs:
        n = 1 << lg
        row = {"log_n": lg}
        for name, ctx in ctxs.items():
            if name == "wide" and n > 2048:
                continue
            dt = timed(lambda: ctx.miller_loop_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), n, d_f.data_ptr()))
            row[name + "_miller_ms"] = round(dt, 3)
            row[name + "_miller_sha"] = hashlib.sha256(d_f[:n].cpu().numpy().tobytes()).hexdigest()[:12]
            dt = timed(lambda: bls._lib.check(ctx.lib.blsgpu_final_exponentiation_device(ctx.h, d_f.data_ptr(), n, d_gt.data_ptr()), "fe"))
            row[name + "_finalexp_ms"] = round(dt, 3)
            row[name + "_finalexp_sha"] = hashlib.sha256(d_gt[:n].cpu().numpy().tobytes()).hexdigest()[:12]
            dt = timed(lambda: ctx.pairing_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), n, d_gt.data_ptr()))
            row[name + "_pairing_ms"] = round(dt, 3)
            row[name + "_pairing_sha"] = hashlib.sha256(d_gt[:n].cpu().numpy().tobytes()).hexdigest()[:12]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
