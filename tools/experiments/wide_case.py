#!/usr/bin/env python3
"""case file for tools/experiments/wide_bench.hip: n random pairs with their Miller values and pairings from the oracle"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bls12_381_ref as o


def fpw(x):
    return np.array(o.fp_to_mont_limbs(x), dtype=np.uint64)


def fp12w(f):
    return np.concatenate([fpw(c) for c in o.fp12_flatten(f)])


n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
r = o.SplitMix64(2024)
rows = []
for _ in range(n):
    P = o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar())); Q = o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar()))
    ml = o.miller_loop(P, Q)
    rows.append(np.concatenate([fpw(P[0]), fpw(P[1]), fpw(Q[0][0]), fpw(Q[0][1]), fpw(Q[1][0]), fpw(Q[1][1]), fp12w(ml), fp12w(o.final_exponentiation(ml))]).view(np.uint32))
with open(sys.argv[1], "wb") as f:
    f.write(np.uint32(n).tobytes()); f.write(np.concatenate(rows).tobytes())
