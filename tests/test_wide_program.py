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
VB = 4
INV28 = (-pow(P, -1, 1 << LW)) % (1 << LW)
P_L = [(P >> (LW * i)) & MASK for i in range(NL)]
ACC_COLS = 32


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


def s5(v):
    v &= 31
    return v - 32 if v & 16 else v


def form(S, word, bias_slot):
    """wide_form: c0 * slot0 + c1 * slot1 + wb * bias slot, limb by limb; every limb must be a non-negative 32-bit number
    (the kernel computes it mod 2^32)"""
    if word == 0:
        return [0] * NL
    k0, k1, kb = s5(word >> 8), s5(word >> 21), (word >> 26) & 15
    s0, s1, sb = S[word & 0xFF], S[(word >> 13) & 0xFF], S[bias_slot]
    assert (k0 == 0 or s0 is not None) and (k1 == 0 or s1 is not None), "read of an unwritten slot"
    x = [(k0 * s0[j] if k0 else 0) + (k1 * s1[j] if k1 else 0) + kb * sb[j] for j in range(NL)]
    assert all(0 <= v < (1 << 32) for v in x), "operand limb out of range"
    return x


def mont_reduce(acc):
    """Montgomery reduction by 2^392 on the column sums (one 64-bit accumulator per column, as wide_mont_reduce does)"""
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
    assert value(res) * RP % P == sum(a << (LW * i) for i, a in enumerate(acc)) % P
    return res


def run_program(prog, inputs):
    hdr = struct.unpack_from("<16I", prog, 0)
    assert hdr[0] == 0x57494432
    nrounds, nslots, nconst, n_in, n_out, c_off, r_off, p_off, d_off, lanes, chunk, bias_slot, max_acc = hdr[1:14]
    assert nslots <= 256 and max_acc <= 128 and chunk in (4, 8)
    W = struct.unpack_from("<%dI" % (len(prog) // 4), prog, 0)
    S = [None] * 256
    for i in range(nconst):
        S[W[c_off + 15 * i]] = list(W[c_off + 15 * i + 1:c_off + 15 * i + 15])
    assert S[bias_slot] == BIAS
    for slot, v in inputs.items():
        S[slot] = limbs(v * RP % P)
    accs = [[0] * ACC_COLS for _ in range(max_acc)]
    for r in range(nrounds):
        pfirst, pcnt, rfirst, rcnt = W[r_off + 4 * r:r_off + 4 * r + 4]
        assert pcnt <= lanes and rcnt <= lanes
        pend = []
        # phase 1: every read of S; products are added to the column accumulators
        for l in range(pcnt):
            d = W[p_off + 4 * (pfirst + l):p_off + 4 * (pfirst + l) + 4]
            op = d[0] & 3
            if op == 1:
                lo, ln, a = (d[0] >> 2) & 15, (d[0] >> 6) & 15, (d[0] >> 10) & 0x3FF
                x, y = form(S, d[1], bias_slot), form(S, d[2], bias_slot)
                assert lo % 4 == 0 and ln == min(chunk, NL - lo)
                assert chunk != 4 or lo // 4 == l % 4, "the four lanes of a product must be one quad, in chunk order (DPP exchange of y)"
                for i in range(lo, min(lo + chunk, NL)):
                    for j in range(NL):
                        accs[a][i + j] += x[i] * y[j]
            elif op == 2:
                x = [u + v for u, v in zip(form(S, d[1], bias_slot), form(S, d[2], bias_slot))]
                assert all(v < (1 << 32) for v in x)
                pend.append(("lin", x, d[3] & 0xFF))
            elif op == 3:
                pend.append(("inv", form(S, d[1], bias_slot), d[3] & 0xFF))
        posts = {}
        for l in range(rcnt):
            w0, w1 = W[d_off + 2 * (rfirst + l)], W[d_off + 2 * (rfirst + l) + 1]
            assert w0 & 1
            if (w0 >> 19) & 1:
                posts[l] = list(S[(w0 >> 20) & 0xFF])
        # phase 2: every write of S
        for kind, x, out in pend:
            if kind == "lin":
                assert value(x) < 1024 * P
                S[out] = reduce_v(limbs(value(x)))
            else:
                v = value(x) % P
                S[out] = limbs((pow(v, -1, P) if v else 0) * RP * RP % P)           # internal form of 1/x: (x R')^-1 R'^2
        for l in range(rcnt):
            w0, w1 = W[d_off + 2 * (rfirst + l)], W[d_off + 2 * (rfirst + l) + 1]
            a, out = (w0 >> 1) & 0x3FF, (w0 >> 11) & 0xFF
            assert all(c < (1 << 64) for c in accs[a]) and not any(accs[a][27:])
            v = mont_reduce(accs[a][:27])
            accs[a] = [0] * ACC_COLS
            if (w0 >> 19) & 1:
                cv, cs, s = s5(w1), s5(w1 >> 5), posts[l]
                t = [cv * v[j] + (cs * s[j] if cs >= 0 else (-cs) * (BIAS[j] - s[j])) for j in range(NL)]
                assert all(0 <= c < (1 << 40) for c in t)
                v = limbs(value(t))
                assert value(v) < 1024 * P
                v = reduce_v(v)
            elif (w0 >> 28) & 1:
                v = reduce_v(v)
            assert value(v) < 2 * P, "stored value not below 2p"
            S[out] = v
    assert not any(any(a) for a in accs)
    return [value(S[i]) * pow(RP, -1, P) % P for i in range(n_out)]


@pytest.fixture(scope="module")
def blob():
    path = os.path.join(ROOT, "build", "wide_prog_test.bin")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_wide_prog.py"), "--check", "--out", path])
    data = open(path, "rb").read()
    head = struct.unpack_from("<16I", data, 0)
    assert head[0] == 0x57504752 and head[1] == 4          # (Miller loop, final exponentiation) x the two built-in configurations
    progs = [data[4 * head[2 + 2 * k]:4 * (head[2 + 2 * k] + head[3 + 2 * k])] for k in range(4)]
    return {(1024, 4): {"miller": progs[0], "final_exp": progs[1]}, (512, 8): {"miller": progs[2], "final_exp": progs[3]}}


@pytest.mark.parametrize("config", [(1024, 4), (512, 8)])
def test_encoded_wide_programs_match_oracle(blob, config):
    progs = blob[config]
    for p in progs.values():
        hdr = struct.unpack_from("<16I", p, 0)
        assert (hdr[10], hdr[11]) == config
    r = o.SplitMix64(4711 + config[0])
    Pa = o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar()))
    Qa = o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar()))
    ins = {12: Pa[0], 13: Pa[1], 14: Qa[0][0], 15: Qa[0][1], 16: Qa[1][0], 17: Qa[1][1]}
    ml = run_program(progs["miller"], ins)
    assert ml == [v % P for v in o.fp12_flatten(o.miller_loop(Pa, Qa))]
    gt = run_program(progs["final_exp"], {i: v for i, v in enumerate(ml)})
    assert gt == [v % P for v in o.fp12_flatten(o.pairing(Pa, Qa))]
    # the final exponentiation alone on 1
    one = o.fp12_flatten(o.FP12_ONE)
    assert run_program(progs["final_exp"], {i: v for i, v in enumerate(one)}) == [v % P for v in one]
