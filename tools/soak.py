#!/usr/bin/env python3
"""Randomised differential soak on the GPU: MSM cases with skewed scalar and base distributions, sizes and window widths drawn at random, checked
through the discrete-log identity against the oracle (the helper of tests/test_gpu_parity.py); pairing bilinearity on random pairs.
   python tools/soak.py [seconds = 240] [seed = 1]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bls12_381_amd as b
import test_gpu_parity as T
from oracle import bls12_381_ref as o
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = b.default_context()
R = o.R_ORDER
def scalar(kind):
    if kind == 0: return rnd.randrange(R)
    if kind == 1: return rnd.choice([0, 1, 2, R - 1, R - 2, (1 << 254), (1 << 255) % R, (1 << 127), (1 << 128) - 1, (1 << 128), (1 << 64) - 1])
    if kind == 2: return rnd.randrange(1 << rnd.choice([1, 8, 16, 17, 32, 63, 64, 127, 128, 129, 200]))
    if kind == 3: return (R - rnd.randrange(1 << 20)) % R
    return sum(((1 << 15) + rnd.choice([0, 1, -1])) << (16 * i) for i in range(16)) % R          # digit boundaries of the 16-bit windows
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    group = rnd.choice([1, 1, 2])
    n = rnd.choice([1, 2, 3, 5, 31, 64, 65, 257, 1000, 4096, 5000]) if group == 1 else rnd.choice([1, 2, 3, 17, 64, 300, 1024])
    kk = rnd.choice([0, 0, 1, 2, 3]); sk = rnd.choice([0, 0, 1, 2, 3, 4])
    pool = [scalar(kk) for _ in range(rnd.choice([1, 2, 7, n]))]
    ks = [rnd.choice(pool) if rnd.random() < 0.5 else scalar(kk) for _ in range(n)]
    spool = [scalar(sk) for _ in range(rnd.choice([1, 3, n]))]
    ss = [rnd.choice(spool) if rnd.random() < 0.6 else scalar(sk) for _ in range(n)]
    if rnd.random() < 0.3:                       # P and -P with the same scalar, identity bases
        for i in range(0, n - 1, 2):
            if rnd.random() < 0.3: ks[i + 1] = (R - ks[i]) % R; ss[i + 1] = ss[i]
        for i in range(n):
            if rnd.random() < 0.05: ks[i] = 0
    w = rnd.choice([0, 0, 0, 4, 7, 9, 12, 13, 16])
    try:
        T._msm_case(ctx, group, ks, ss, window=w)
    except AssertionError:
        print("MISMATCH group", group, "n", n, "window", w, "kinds", kk, sk); np.save("/tmp/soak_fail.npy", np.array([ks, ss], dtype=object)); sys.exit(1)
    cases += 1
    if cases % 200 == 0:
        # the part of hash_to_curve behind the expander on random uniform bytes, both groups, against the oracle's restatement of the reference
        from oracle import h2c_ref as h
        for g in (1, 2):
            M = 1 if g == 1 else 2
            uni = np.frombuffer(bytes(rnd.getrandbits(8) for _ in range(3 * 2 * M * 64)), dtype=np.uint8).reshape(3, 2 * M * 64)
            out = ctx.hash_to_curve_from_uniform(g, uni)
            for k in range(3):
                raw = uni[k].tobytes()
                if g == 1:
                    u = [h.fp_from_okm(raw[64 * i:64 * i + 64]) for i in range(2)]
                    want = np.concatenate([T.fpw(c) for c in h.g1_clear_cofactor(o.g1_add(h.g1_map_to_curve(u[0]), h.g1_map_to_curve(u[1])))])
                else:
                    u = [(h.fp_from_okm(raw[128 * i:128 * i + 64]), h.fp_from_okm(raw[128 * i + 64:128 * i + 128])) for i in range(2)]
                    want = np.concatenate([T.fp2w(c) for c in h.g2_clear_cofactor(o.g2_add(h.g2_map_to_curve(u[0]), h.g2_map_to_curve(u[1])))])
                assert np.array_equal(out[k], want), ("hash_to_curve_from_uniform", g, k)
        h2c_checked = globals().get("h2c_checked", 0) + 6
    if cases % 25 == 0:
        # pairing bilinearity on a fresh random pair: e(aP, bQ) == e(P, Q)^(ab)
        a, bb = rnd.randrange(1, R), rnd.randrange(1, R)
        g1, f1 = ctx.bases_from_scalars(1, [a, 1]).download(); g2, f2 = ctx.bases_from_scalars(2, [bb, 1]).download()
        gt = ctx.pairing_batch(g1, f1, g2, f2)
        want = ctx.gt_mul_scalar_batch(gt[1:2], [a * bb % R])[0]
        assert np.array_equal(gt[0], want), "bilinearity"
print("soak ok:", cases, "MSM cases,", globals().get("h2c_checked", 0), "maps from uniform bytes in", round(time.time() - t0), "s")
