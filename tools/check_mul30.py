#!/usr/bin/env python3
"""Checks the {"check": "chain30", ...} line of tools/microbench_mul30.hip against Python integers:
x_{i+1} = x_{i-1} * x_i / 2^390 mod p, five steps from (a, b)."""
import json
import sys
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
for line in sys.stdin:
    if '"check"' not in line:
        continue
    d = json.loads(line)
    val = lambda l: sum(v << (30 * i) for i, v in enumerate(l))
    a, b = val(d["a"]), val(d["b"])
    rinv = pow(1 << 390, -1, P)
    for _ in range(d["steps"]):
        a, b = b, a * b * rinv % P
    got = val(d["out"])
    print("mul30 chain", "ok" if got % P == b else "MISMATCH", "(value below 2p: %s)" % (got < 2 * P))
