"""Big-integer models of the two scalar decompositions the MSM kernels use (bls12_381_amd/csrc/msm.hip.h:
k_glv_decompose for G1, k_gls_decompose for G2).  Each model follows its kernel branch by branch, with the kernel's own
constants parsed from csrc/consts_gen.h, so that tests/test_host_cpu.py can check on the CPU -- against plain big integers --
what the kernels are built to guarantee:

    G1:  k = s1 |k1| + s2 |k2| L  (mod r),   L = z^2,  |k1|, |k2| < 2^127      (phi(P) = -[L] P on the subgroup, g1.rs:396-437)
    G2:  k = sum_j s_j |d_j| x^j  (mod r),   x = -X,   |d_j| < 2^63            (psi(P) = [x] P on the subgroup, g2.rs:475-482)

Test infrastructure only: the product never imports this file."""
import functools
import os
import re

R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_ABS = 0xD201000000010000
L = X_ABS * X_ABS
assert L * L - L + 1 == R_ORDER

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bls12_381_amd", "csrc", "consts_gen.h")


@functools.lru_cache(maxsize=None)
def header_words(name):
    """the little-endian 32-bit word list of a `#define NAME {..}` in consts_gen.h as one integer"""
    for line in open(_HDR):
        m = re.match(r"#define %s \{([^}]*)\}" % name, line)
        if m:
            ws = [int(w.strip().rstrip("u"), 16) for w in m.group(1).split(",")]
            return sum(w << (32 * i) for i, w in enumerate(ws))
    raise KeyError(name)


def glv_model(k):
    """k_glv_decompose: returns (|k1|, neg1, |k2|, sub2, barrett_short) -- the two 127-bit magnitudes, the flag 'k1 P is
    subtracted', the flag 'k2 phi(P) is subtracted' as the kernel stores them, and whether the Barrett estimate was one short."""
    Lw, M, H = header_words("BLS_GLV_L_W"), header_words("BLS_GLV_M_W"), header_words("BLS_GLV_H_W")
    assert Lw == L and M == (1 << 256) // L and H == L >> 1
    q = (k * M) >> 256
    k1 = (k - q * L) & ((1 << 160) - 1)          # five 32-bit words, as the kernel keeps them
    k2 = q
    short = k1 >= L
    if short:
        k1 -= L
        k2 += 1
    assert k1 < L and k2 < (1 << 128)
    neg1 = neg2 = 0
    if k1 > H:
        k1 = L - k1
        neg1 = 1
        k2 += 1
    if k2 > H:
        k2 = (L - 1) - k2
        neg2 = 1
        if neg1:
            k1 += 1
        elif k1 == 0:
            k1, neg1 = 1, 1
        else:
            k1 -= 1
    return k1, neg1, k2, 0 if neg2 else 1, short


def glv_value(k1, neg1, k2, sub2):
    """the scalar the MSM effectively applies: the phi term stands for -[L] P, so 'subtract k2 phi(P)' adds k2 L"""
    return ((-k1 if neg1 else k1) + (k2 if sub2 else -k2) * L) % R_ORDER


def gls_model(k):
    """k_gls_decompose: returns [(|d_j|, subtracted_j)] for j = 0..3 as the kernel stores them."""
    X, H = X_ABS, X_ABS >> 1
    d = []
    q = k
    for _ in range(3):
        q, rem = divmod(q, X)
        d.append(rem)
    d.append(q)
    assert q < X
    neg = [0, 0, 0, 0]
    carry = 0
    for j in range(4):
        v = d[j] + carry
        if v > H:
            d[j], neg[j], carry = X - v, 1, 1
        else:
            d[j], neg[j], carry = v, 0, 0
    if carry:                                    # X^4 = x^4 = x^2 - 1 (mod r): d2 += 1, d0 -= 1 (signed)
        if neg[2]:
            if d[2] == 0:
                d[2], neg[2] = 1, 0
            else:
                d[2] -= 1
        else:
            d[2] += 1
        if neg[0]:
            d[0] += 1
        elif d[0] == 0:
            d[0], neg[0] = 1, 1
        else:
            d[0] -= 1
    # k P = d0 P - d1 psi(P) + d2 psi^2(P) - d3 psi^3(P): odd terms are subtracted when their digit is positive
    sub = [neg[0], neg[1] ^ 1, neg[2], neg[3] ^ 1]
    return list(zip(d, sub))


def gls_value(terms):
    """psi^j(P) = [x^j] P with x = -X: a stored term (|d|, subtracted) contributes -/+ |d| x^j"""
    x = -X_ABS
    return sum((-m if s else m) * x ** j for j, (m, s) in enumerate(terms)) % R_ORDER


def glv_candidates():
    """scalars at every branch of k_glv_decompose (all reduced into [0, r))"""
    H = L >> 1
    rr = R_ORDER
    cand = [0, 1, 2, rr - 1, rr - 2, 1 << 254, (1 << 255) % rr, L - 1, L, L + 1, H - 1, H, H + 1, H + 2]
    for m in (1, 2, 3, H - 1, H, H + 1, H + 2, L - 2, L - 1):
        for dlt in (-2, -1, 0, 1, 2):
            cand.append(m * L + dlt)                 # multiples of L: the Barrett estimate is one short exactly here
            cand.append(m * L + H + dlt)             # k1 at the balancing threshold
            cand.append(m * L + L - 1 + dlt)
    cand += [H * L + H, H * L + H + 1, (H + 1) * L, (H + 1) * L + H + 1, (H + 1) * L - 1, H * L + L - 1, H * L]   # k2 at the threshold, k1 = 0 corner
    return sorted({c % rr for c in cand if c >= 0})


def gls_candidates():
    X, H = X_ABS, X_ABS >> 1
    rr = R_ORDER
    cand = [0, 1, 2, rr - 1, rr - 2, H, H + 1, H - 1, X - 1, X, X + 1]
    for pw in (1, 2, 3):
        for m in (1, 2, H, H + 1, X - 1):
            for dlt in (-1, 0, 1):
                cand.append(m * X ** pw + dlt)
                cand.append(m * X ** pw + H + dlt)
    cand += [X ** 3 * (X - 1) + X ** 2 * (X - 1), (H + 1) * (1 + X + X ** 2 + X ** 3), H * (1 + X + X ** 2 + X ** 3)]
    return sorted({c % rr for c in cand if c >= 0})
