#!/usr/bin/env python3
"""A/B timing of the pairing kernels for alternative builds of libblsgpu.so.

    python tools/ab_pairing.py build/libblsgpu_a.so build/libblsgpu_b.so ...

Each library is exercised in its own subprocess (BLSGPU_LIB_PATH): 2^16 pairings, 2^18-term multi_miller_loop, with the
outputs hashed so that variants can be compared for bit-equality."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import bls12_381_amd as bls
    dev = torch.device("cuda", 0)
    ctx = bls.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << 16
    rs = np.random.RandomState(99)
    ka = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
    kq = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); kq[:, 31] &= 0x3F
    g1xy, _ = ctx.bases_from_scalars(1, ka).download()
    g2xy, _ = ctx.bases_from_scalars(2, kq).download()
    d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev)
    d_gt = torch.zeros((n, 72), dtype=torch.int64, device=dev)
    out = {}

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        return best

    def pair():
        bls._lib.check(ctx.lib.blsgpu_pairing_batch_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, n, d_gt.data_ptr()), "pairing")
    dt = timed(pair, 5)
    out["pairing_2_16_ms"] = 1e3 * dt
    out["pairings_per_s"] = n / dt
    out["pairing_sha"] = hashlib.sha256(d_gt.cpu().numpy().tobytes()).hexdigest()[:16]
    nm = 4 * n
    d_g1m, d_g2m = d_g1.repeat(4, 1), d_g2.repeat(4, 1)
    d_one = torch.zeros(72, dtype=torch.int64, device=dev)

    def mml():
        bls._lib.check(ctx.lib.blsgpu_multi_miller_loop_device(ctx.h, d_g1m.data_ptr(), None, d_g2m.data_ptr(), None, nm, d_one.data_ptr()), "mml")
    dt = timed(mml, 5)
    out["mml_2_18_ms"] = 1e3 * dt
    out["mml_terms_per_s"] = nm / dt
    out["mml_sha"] = hashlib.sha256(d_one.cpu().numpy().tobytes()).hexdigest()[:16]

    def mml16():
        bls._lib.check(ctx.lib.blsgpu_multi_miller_loop_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, n, d_one.data_ptr()), "mml")
    dt = timed(mml16, 5)
    out["mml_2_16_ms"] = 1e3 * dt
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child()
    else:
        for lib in sys.argv[1:]:
            env = dict(os.environ, AB_CHILD="1", BLSGPU_LIB_PATH=os.path.abspath(lib))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(os.path.basename(lib), line[-1] if line else ("FAILED: " + r.stderr[-800:]))
