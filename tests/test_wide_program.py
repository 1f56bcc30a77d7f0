"""The wide pairing programs (tools/gen_wide_prog.py -> bls12_381_amd/wide_prog.bin), executed at LIMB level in Python exactly
as bls12_381_amd/csrc/wide.hip.h specifies them -- 14 x 28-bit limbs, lazily formed operands, 64-bit column sums with overflow
checks, Montgomery reduction by 2^392, weak reduction of every stored value -- and compared with the oracle.  The generator
already checks its symbolic values against the oracle; this test checks what the kernel actually reads: the ENCODED tables
(slot allocation, lane dealing, operand words, post-operations), on inputs the tables were not generated with."""
import os
import struct
import subprocess
import sys

import pytest

from oracle import bls12_381_ref as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = o.P
NL, LW = 14, 28
MASK = (1 << LW) - 1
RP = 1 << (NL * LW)
COEF = [0, 1, -1, 2, -2, 3, -3, 4, -4, 6, -6, 8, -8, 12, -12, 9]
VB = 4
INV28 = (-pow(P, -1, 1 << LW)) % (1 << LW)
P_L = [(P >> (LW * i)) & MASK for i in range(NL)]


def make_bias(c, s):
    """fe.hip.h make_bias: c * p spread over the limbs so that every limb is >= s * (2^28 - 1)"""
    r, carry = [0] * NL, 0
    for i in range(NL):
        t = P_L[i] * c + carry
        r[i] = (t & MASK) if i < NL - 1 else t
        carry = (t >> LW) if i < NL - 1 else 0
    r[0] += s << LW
    for i in range(1, NL - 1):
        r[i] += (s << LW) - s
    r[NL - 1] -= s
    assert sum(v << (LW * i) for i, v in enumerate(r)) == c * P
    return r


BIAS = make_bias(VB, 1)


def limbs(v):
    return [(v >> (LW * i)) & MASK for i in range(NL - 1)] + [v >> (LW * (NL - 1))]


def value(l):
    return sum(v << (LW * i) for i, v in enumerate(l))


def reduce_v(l):
    """fe.hip.h reduce_v: normalised limbs, value < 1024 p  ->  same residue below 2p"""
    q = l[NL - 1] // 106514
    v = value(l) - q * P
    assert 0 <= v < 2 * P
    return limbs(v)


def load_items(S, word, count):
    """lazily formed operand: sum of coefficient * slot, negative coefficients against the bias; limbs must fit 32 bits"""
    x = [0] * NL
    for i in range(count):
        f = (word[i // 2] >> (14 * (i % 2))) & 0x3FFF
        slot, k = f & 0x3FF, COEF[f >> 10]
        if k == 0:
            continue
        s = S[slot]
        assert s is not None, "read of an unwritten slot"
        for j in range(NL):
            x[j] += k * s[j] if k > 0 else (-k) * (BIAS[j] - s[j])
    assert all(0 <= v < (1 << 32) for v in x), "operand limb overflow"
    return x


def pair_count(word):
    return 2 if (word >> 24) & 0xF else 1


def run_program(prog, inputs):
    hdr = struct.unpack_from("<16I", prog, 0)
    assert hdr[0] == 0x57494445
    nrounds, nslots, nconst, n_in, n_out, c_off, r_off, d_off = hdr[1:9]
    W = struct.unpack_from("<%dI" % (len(prog) // 4), prog, 0)
    S = [None] * max(nslots, n_in, n_out)
    for i in range(nconst):
        S[W[c_off + 15 * i]] = list(W[c_off + 15 * i + 1:c_off + 15 * i + 15])
    for slot, v in inputs.items():
        S[slot] = limbs(v * RP % P)
    for r in range(nrounds):
        first, lanes = W[r_off + 2 * r], W[r_off + 2 * r + 1]
        D = [W[d_off + 8 * (first + l):d_off + 8 * (first + l) + 8] for l in range(lanes)]
        part, pend = {}, []
        # phase 1: every read of S
        for l, d in enumerate(D):
            op, nt, red, nparts, out = d[0] & 3, (d[0] >> 2) & 15, (d[0] >> 6) & 1, (d[0] >> 7) & 0x1FF, (d[0] >> 16) & 0x3FF
            if op == 1:
                acc = [0] * (2 * NL - 1)
                for t in range(nt):
                    x = load_items(S, [d[1 + 2 * t]], pair_count(d[1 + 2 * t]))
                    y = load_items(S, [d[2 + 2 * t]], pair_count(d[2 + 2 * t]))
                    for i in range(NL):
                        for j in range(NL):
                            acc[i + j] += x[i] * y[j]
                part[l] = acc
                if red:
                    post = None
                    if d[7] >> 31:
                        ps, cs, cv = d[7] & 0x3FF, COEF[(d[7] >> 10) & 15], COEF[(d[7] >> 14) & 15]
                        post = (cv, list(S[ps]), cs)
                    pend.append((l, "sop", nparts, out, post))
            elif op == 2:
                pend.append((l, "lin", load_items(S, [d[1], d[2]], nt), out, None))
            elif op == 3:
                pend.append((l, "inv", load_items(S, [d[1]], 1), out, None))
        # phase 2: every write of S
        for l, kind, arg, out, post in pend:
            if kind == "sop":
                acc = list(part[l])
                for k in range(1, arg):
                    acc = [a + b for a, b in zip(acc, part[l + k])]
                # Montgomery reduction by 2^392 on the column sums (one 64-bit accumulator per column, as fe_sop2_body does)
                c, m, res = 0, [], [0] * NL
                for k in range(NL):
                    t = acc[k] + c + sum(m[i] * P_L[k - i] for i in range(k))
                    m.append(((t & 0xFFFFFFFF) * INV28) & MASK)
                    t += m[k] * P_L[0]
                    assert t < (1 << 64) and t & MASK == 0, "column overflow"
                    c = t >> LW
                for k in range(NL, 2 * NL - 1):
                    t = acc[k] + c + sum(m[i] * P_L[k - i] for i in range(k - NL + 1, NL))
                    assert t < (1 << 64), "column overflow"
                    res[k - NL] = t & MASK
                    c = t >> LW
                res[NL - 1] = c
                v = res
                assert value(v) * RP % P == sum(a << (LW * i) for i, a in enumerate(acc)) % P
                if post:
                    cv, s, cs = post
                    t = [cv * v[j] + (cs * s[j] if cs > 0 else (-cs) * (BIAS[j] - s[j])) for j in range(NL)]
                    v = limbs(value(t))
                assert value(v) < 1024 * P
                S[out] = reduce_v(v)
            elif kind == "lin":
                assert value(arg) < 1024 * P
                S[out] = reduce_v(limbs(value(arg)))
            else:
                x = value(arg) % P
                S[out] = limbs((pow(x, -1, P) if x else 0) * RP * RP % P)           # internal form of 1/x: (x R')^-1 R'^2
    return [value(S[i]) * pow(RP, -1, P) % P for i in range(n_out)]


@pytest.fixture(scope="module")
def blob():
    path = os.path.join(ROOT, "build", "wide_prog_test.bin")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_wide_prog.py"), "--check", "--out", path])
    data = open(path, "rb").read()
    head = struct.unpack_from("<16I", data, 0)
    assert head[0] == 0x57504752 and head[1] == 2
    return {"miller": data[4 * head[2]:4 * (head[2] + head[3])], "final_exp": data[4 * head[4]:4 * (head[4] + head[5])]}


def test_encoded_wide_programs_match_oracle(blob):
    r = o.SplitMix64(4711)
    Pa = o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar()))
    Qa = o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar()))
    ins = {12: Pa[0], 13: Pa[1], 14: Qa[0][0], 15: Qa[0][1], 16: Qa[1][0], 17: Qa[1][1]}
    ml = run_program(blob["miller"], ins)
    assert ml == [v % P for v in o.fp12_flatten(o.miller_loop(Pa, Qa))]
    gt = run_program(blob["final_exp"], {i: v for i, v in enumerate(ml)})
    assert gt == [v % P for v in o.fp12_flatten(o.pairing(Pa, Qa))]
    # the final exponentiation alone, on the generator pairing's Miller value and on 1
    one = o.fp12_flatten(o.FP12_ONE)
    assert run_program(blob["final_exp"], {i: v for i, v in enumerate(one)}) == [v % P for v in one]
