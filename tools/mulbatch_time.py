#!/usr/bin/env python3
"""Timing of the batched variable-base scalar multiplication (device pointers, HIP-synchronised wall clock)."""
import json
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import bls12_381_amd as bls
    from bls12_381_amd import synthetic as sy
    dev = torch.device("cuda", 0)
    ctx = bls.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for group, lg in ((1, 20), (1, 16), (2, 18), (2, 14)):
        n = 1 << lg
        xy, _ = ctx.bases_from_scalars(group, sy.scalars(n, sy.SEED + 5)).download()
        s = sy.scalars(n, sy.SEED + 6)
        d_xy = torch.from_numpy(xy.view(np.int64)).to(dev); d_s = torch.from_numpy(s).to(dev)
        d_out = torch.zeros((n, 18 if group == 1 else 36), dtype=torch.int64, device=dev)
        fn = lambda: ctx.mul_batch_device(group, d_xy.data_ptr(), 0, d_s.data_ptr(), n, d_out.data_ptr())
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        print(json.dumps({"group": group, "log_n": lg, "ms": round(1e3 * best, 3), "scalar_muls_per_s": n / best}), flush=True)


if __name__ == "__main__":
    main()
