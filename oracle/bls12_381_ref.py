"""Tier-0 oracle: pure-Python big-integer restatement of the zkcrypto/bls12_381 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`bls12_381_amd/`) may import this module; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may use `oracle/`.

Every function restates the *algorithm* of the reference (crate v0.8.0, paths relative to
/root/reference) on Python integers, so that intermediate values -- not just final results -- match
the reference (projective triples, Miller-loop values, line coefficients).  Field elements are plain
integers in [0, p); the Montgomery form the reference stores (R = 2^384, src/fp.rs:83-90) only
appears in the (de)serialisation helpers `fp_to_mont_limbs` / `fp_from_mont_limbs`.

Pinned by tests/test_oracle_golden.py against the reference's own fixtures:
the four k*G golden files (src/tests/mod.rs:3-76), the RELIC pairing constant
(src/pairings.rs:359-475 == src/tests/mod.rs:78-231), the field/group KATs of src/fp.rs, src/fp2.rs,
src/g1.rs, src/g2.rs, and the module constants.
"""

# --------------------------------------------------------------------------------------------------
# constants
# --------------------------------------------------------------------------------------------------
# src/fp.rs:70-77 (MODULUS)
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
# src/scalar.rs:76-81 (MODULUS of Fr)
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
# src/lib.rs:70-74
BLS_X = 0xD201000000010000
BLS_X_IS_NEGATIVE = True
MONT_R = (1 << 384) % P          # src/fp.rs:83-90
MONT_R_INV = pow(MONT_R, -1, P)
FR_MONT_R = (1 << 256) % R_ORDER  # src/scalar.rs:155-160


def fp_to_mont_limbs(x):
    """Integer in [0,p) -> six little-endian u64 Montgomery limbs (the reference's `Fp([u64;6])`)."""
    v = (x * MONT_R) % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def fp_from_mont_limbs(limbs):
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return (v * MONT_R_INV) % P


# --------------------------------------------------------------------------------------------------
# Fp  (src/fp.rs)
# --------------------------------------------------------------------------------------------------
def fp_add(a, b): return (a + b) % P          # fp.rs:382-394
def fp_sub(a, b): return (a - b) % P          # fp.rs:421-423
def fp_neg(a): return (-a) % P                # fp.rs:397-418
def fp_mul(a, b): return (a * b) % P          # fp.rs:565-609
def fp_sqr(a): return (a * a) % P             # fp.rs:613-660


def fp_inv(a):
    """fp.rs:346-358: x^(p-2); returns None for zero (CtOption none)."""
    if a % P == 0:
        return None
    return pow(a, P - 2, P)


def fp_sqrt(a):
    """fp.rs:324-340: x^((p+1)/4), None if not a square."""
    s = pow(a, (P + 1) // 4, P)
    return s if (s * s) % P == a % P else None


def fp_lex_largest(a):
    """fp.rs:273-298: a > (p-1)/2."""
    return a > (P - 1) // 2


def fp_to_bytes(a):
    return int(a).to_bytes(48, "big")          # fp.rs:209-227


def fp_from_bytes(b):
    """fp.rs:179-205: big-endian, must be canonical."""
    v = int.from_bytes(b, "big")
    return v if v < P else None


# --------------------------------------------------------------------------------------------------
# Fp2 = Fp[u]/(u^2+1)  (src/fp2.rs);  element = (c0, c1)
# --------------------------------------------------------------------------------------------------
FP2_ZERO = (0, 0)
FP2_ONE = (1, 0)


def fp2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)      # fp2.rs:224-229
def fp2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)      # fp2.rs:231-236
def fp2_neg(a): return ((-a[0]) % P, (-a[1]) % P)                      # fp2.rs:238-243
def fp2_conj(a): return (a[0], (-a[1]) % P)                            # fp2.rs:148-153
def fp2_is_zero(a): return a[0] % P == 0 and a[1] % P == 0


def fp2_mul(a, b):
    """fp2.rs:205-222."""
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_sqr(a):
    """fp2.rs:182-203 (complex squaring)."""
    return (((a[0] + a[1]) * (a[0] - a[1])) % P, (2 * a[0] * a[1]) % P)


def fp2_mul_by_nonresidue(a):
    """fp2.rs:156-166: multiply by (u+1)."""
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


def fp2_mul_fp(a, k): return ((a[0] * k) % P, (a[1] * k) % P)


def fp2_inv(a):
    """fp2.rs:300-319."""
    t = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    if t is None:
        return None
    return ((a[0] * t) % P, (a[1] * (-t)) % P)


def fp2_pow(a, e):
    r = FP2_ONE
    for i in reversed(range(e.bit_length())):
        r = fp2_sqr(r)
        if (e >> i) & 1:
            r = fp2_mul(r, a)
    return r


def fp2_sqrt(a):
    """fp2.rs:245-295 (Algorithm 9 of eprint 2012/685)."""
    if fp2_is_zero(a):
        return FP2_ZERO
    a1 = fp2_pow(a, (P - 3) // 4)
    alpha = fp2_mul(fp2_sqr(a1), a)
    x0 = fp2_mul(a1, a)
    if alpha == ((-1) % P, 0):
        s = ((-x0[1]) % P, x0[0])
    else:
        s = fp2_mul(fp2_pow(fp2_add(alpha, FP2_ONE), (P - 1) // 2), x0)
    return s if fp2_sqr(s) == (a[0] % P, a[1] % P) else None


def fp2_lex_largest(a):
    """fp2.rs:171-180."""
    return fp_lex_largest(a[1]) or (a[1] % P == 0 and fp_lex_largest(a[0]))


# --------------------------------------------------------------------------------------------------
# Fp6 = Fp2[v]/(v^3-(u+1))  (src/fp6.rs);  element = (c0, c1, c2)
# --------------------------------------------------------------------------------------------------
FP6_ZERO = (FP2_ZERO, FP2_ZERO, FP2_ZERO)
FP6_ONE = (FP2_ONE, FP2_ZERO, FP2_ZERO)

# Frobenius coefficients, fp6.rs:159-185 / fp12.rs:149-168; recomputed from their definitions here and
# compared with the reference's literal limbs in tests/test_oracle_golden.py.
FROB6_C1 = fp2_pow((1, 1), (P - 1) // 3)
FROB6_C2 = fp2_pow((1, 1), (2 * P - 2) // 3)
FROB12_C1 = fp2_pow((1, 1), (P - 1) // 6)


def fp6_add(a, b): return tuple(fp2_add(x, y) for x, y in zip(a, b))
def fp6_sub(a, b): return tuple(fp2_sub(x, y) for x, y in zip(a, b))
def fp6_neg(a): return tuple(fp2_neg(x) for x in a)


def fp6_mul(a, b):
    """fp6.rs:200-274 (mul_interleaved) -- schoolbook over Fp2 with v^3 = u+1."""
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = fp2_add(fp2_mul(a0, b0), fp2_mul_by_nonresidue(fp2_add(fp2_mul(a1, b2), fp2_mul(a2, b1))))
    c1 = fp2_add(fp2_add(fp2_mul(a0, b1), fp2_mul(a1, b0)), fp2_mul_by_nonresidue(fp2_mul(a2, b2)))
    c2 = fp2_add(fp2_add(fp2_mul(a0, b2), fp2_mul(a1, b1)), fp2_mul(a2, b0))
    return (c0, c1, c2)


def fp6_sqr(a):
    """fp6.rs:277-291."""
    s0 = fp2_sqr(a[0])
    ab = fp2_mul(a[0], a[1])
    s1 = fp2_add(ab, ab)
    s2 = fp2_sqr(fp2_add(fp2_sub(a[0], a[1]), a[2]))
    bc = fp2_mul(a[1], a[2])
    s3 = fp2_add(bc, bc)
    s4 = fp2_sqr(a[2])
    return (fp2_add(fp2_mul_by_nonresidue(s3), s0),
            fp2_add(fp2_mul_by_nonresidue(s4), s1),
            fp2_sub(fp2_sub(fp2_add(fp2_add(s1, s2), s3), s0), s4))


def fp6_mul_by_1(a, c1):
    """fp6.rs:113-119."""
    return (fp2_mul_by_nonresidue(fp2_mul(a[2], c1)), fp2_mul(a[0], c1), fp2_mul(a[1], c1))


def fp6_mul_by_01(a, c0, c1):
    """fp6.rs:121-136."""
    a_a = fp2_mul(a[0], c0)
    b_b = fp2_mul(a[1], c1)
    t1 = fp2_add(fp2_mul_by_nonresidue(fp2_mul(a[2], c1)), a_a)
    t2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(c0, c1), fp2_add(a[0], a[1])), a_a), b_b)
    t3 = fp2_add(fp2_mul(a[2], c0), b_b)
    return (t1, t2, t3)


def fp6_mul_by_nonresidue(a):
    """fp6.rs:139-150: multiply by v."""
    return (fp2_mul_by_nonresidue(a[2]), a[0], a[1])


def fp6_frobenius(a):
    """fp6.rs:154-188."""
    return (fp2_conj(a[0]), fp2_mul(fp2_conj(a[1]), FROB6_C1), fp2_mul(fp2_conj(a[2]), FROB6_C2))


def fp6_inv(a):
    """fp6.rs:294-312."""
    c0 = fp2_sub(fp2_sqr(a[0]), fp2_mul_by_nonresidue(fp2_mul(a[1], a[2])))
    c1 = fp2_sub(fp2_mul_by_nonresidue(fp2_sqr(a[2])), fp2_mul(a[0], a[1]))
    c2 = fp2_sub(fp2_sqr(a[1]), fp2_mul(a[0], a[2]))
    tmp = fp2_mul_by_nonresidue(fp2_add(fp2_mul(a[1], c2), fp2_mul(a[2], c1)))
    tmp = fp2_add(tmp, fp2_mul(a[0], c0))
    t = fp2_inv(tmp)
    if t is None:
        return None
    return (fp2_mul(t, c0), fp2_mul(t, c1), fp2_mul(t, c2))


# --------------------------------------------------------------------------------------------------
# Fp12 = Fp6[w]/(w^2-v)  (src/fp12.rs);  element = (c0, c1)
# --------------------------------------------------------------------------------------------------
FP12_ONE = (FP6_ONE, FP6_ZERO)


def fp12_mul(a, b):
    """fp12.rs:197-214."""
    aa = fp6_mul(a[0], b[0])
    bb = fp6_mul(a[1], b[1])
    o = fp6_add(b[0], b[1])
    c1 = fp6_mul(fp6_add(a[1], a[0]), o)
    c1 = fp6_sub(fp6_sub(c1, aa), bb)
    c0 = fp6_add(fp6_mul_by_nonresidue(bb), aa)
    return (c0, c1)


def fp12_sqr(a):
    """fp12.rs:174-185."""
    ab = fp6_mul(a[0], a[1])
    c0c1 = fp6_add(a[0], a[1])
    c0 = fp6_add(fp6_mul_by_nonresidue(a[1]), a[0])
    c0 = fp6_sub(fp6_mul(c0, c0c1), ab)
    c1 = fp6_add(ab, ab)
    c0 = fp6_sub(c0, fp6_mul_by_nonresidue(ab))
    return (c0, c1)


def fp12_mul_by_014(a, c0, c1, c4):
    """fp12.rs:116-128."""
    aa = fp6_mul_by_01(a[0], c0, c1)
    bb = fp6_mul_by_1(a[1], c4)
    o = fp2_add(c1, c4)
    r1 = fp6_mul_by_01(fp6_add(a[1], a[0]), c0, o)
    r1 = fp6_sub(fp6_sub(r1, aa), bb)
    r0 = fp6_add(fp6_mul_by_nonresidue(bb), aa)
    return (r0, r1)


def fp12_conj(a): return (a[0], fp6_neg(a[1]))          # fp12.rs:136-141


def fp12_frobenius(a):
    """fp12.rs:145-171."""
    c0 = fp6_frobenius(a[0])
    c1 = fp6_frobenius(a[1])
    c1 = tuple(fp2_mul(x, FROB12_C1) for x in c1)       # Fp6 * Fp6::from(Fp2) == coefficient-wise
    return (c0, c1)


def fp12_inv(a):
    """fp12.rs:187-194."""
    t = fp6_inv(fp6_sub(fp6_sqr(a[0]), fp6_mul_by_nonresidue(fp6_sqr(a[1]))))
    if t is None:
        return None
    return (fp6_mul(a[0], t), fp6_mul(a[1], fp6_neg(t)))


def fp12_flatten(a):
    """Reference struct order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 (fp12.rs:13-16, fp6.rs:12-16)."""
    return [c for f6 in a for f2 in f6 for c in f2]


def fp12_unflatten(v):
    v = list(v)
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))


# --------------------------------------------------------------------------------------------------
# Generic short-Weierstrass group code over a field "F" (Fp for G1, Fp2 for G2)
# --------------------------------------------------------------------------------------------------
class _FpOps:
    zero, one = 0, 1
    add, sub, neg, mul, sqr, inv = staticmethod(fp_add), staticmethod(fp_sub), staticmethod(fp_neg), \
        staticmethod(fp_mul), staticmethod(fp_sqr), staticmethod(fp_inv)
    @staticmethod
    def is_zero(a): return a % P == 0
    @staticmethod
    def mul_by_3b(a): return (12 * a) % P                 # g1.rs:597-601 (b = 4)
    B = 4                                                 # g1.rs:176-183


class _Fp2Ops:
    zero, one = FP2_ZERO, FP2_ONE
    add, sub, neg, mul, sqr, inv = staticmethod(fp2_add), staticmethod(fp2_sub), staticmethod(fp2_neg), \
        staticmethod(fp2_mul), staticmethod(fp2_sqr), staticmethod(fp2_inv)
    is_zero = staticmethod(fp2_is_zero)
    B = (4, 4)                                            # g2.rs:177-194
    @staticmethod
    def mul_by_3b(a): return fp2_mul(a, (12, 12))         # g2.rs:196,650-652


def _identity(F): return (F.zero, F.one, F.zero)          # g1.rs:605-611 / g2.rs identity


def _double(F, p):
    """RCB15 Algorithm 9 (a=0): g1.rs:638-667 / g2.rs:709-738."""
    x, y, z = p
    t0 = F.sqr(y)
    z3 = F.add(t0, t0); z3 = F.add(z3, z3); z3 = F.add(z3, z3)
    t1 = F.mul(y, z)
    t2 = F.mul_by_3b(F.sqr(z))
    x3 = F.mul(t2, z3)
    y3 = F.add(t0, t2)
    z3 = F.mul(t1, z3)
    t1 = F.add(t2, t2); t2 = F.add(t1, t2)
    t0 = F.sub(t0, t2)
    y3 = F.add(x3, F.mul(t0, y3))
    t1 = F.mul(x, y)
    x3 = F.mul(t0, t1); x3 = F.add(x3, x3)
    if F.is_zero(z):
        return _identity(F)
    return (x3, y3, z3)


def _add(F, p, q):
    """RCB15 Algorithm 7 (a=0): g1.rs:670-712 / g2.rs:741-783."""
    x1, y1, z1 = p
    x2, y2, z2 = q
    t0 = F.mul(x1, x2); t1 = F.mul(y1, y2); t2 = F.mul(z1, z2)
    t3 = F.mul(F.add(x1, y1), F.add(x2, y2))
    t3 = F.sub(t3, F.add(t0, t1))
    t4 = F.mul(F.add(y1, z1), F.add(y2, z2))
    t4 = F.sub(t4, F.add(t1, t2))
    x3 = F.mul(F.add(x1, z1), F.add(x2, z2))
    y3 = F.sub(x3, F.add(t0, t2))
    x3 = F.add(t0, t0); t0 = F.add(x3, t0)
    t2 = F.mul_by_3b(t2)
    z3 = F.add(t1, t2); t1 = F.sub(t1, t2)
    y3 = F.mul_by_3b(y3)
    x3 = F.mul(t4, y3); t2 = F.mul(t3, t1); x3 = F.sub(t2, x3)
    y3 = F.mul(y3, t0); t1 = F.mul(t1, z3); y3 = F.add(t1, y3)
    t0 = F.mul(t0, t3); z3 = F.mul(z3, t4); z3 = F.add(z3, t0)
    return (x3, y3, z3)


def _add_mixed(F, p, q_aff):
    """RCB15 Algorithm 8 (a=0): g1.rs:715-752 / g2.rs:786-823.  q_aff = (x, y, infinity)."""
    x1, y1, z1 = p
    x2, y2, inf = q_aff
    t0 = F.mul(x1, x2); t1 = F.mul(y1, y2)
    t3 = F.mul(F.add(x2, y2), F.add(x1, y1))
    t3 = F.sub(t3, F.add(t0, t1))
    t4 = F.add(F.mul(y2, z1), y1)
    y3 = F.add(F.mul(x2, z1), x1)
    x3 = F.add(t0, t0); t0 = F.add(x3, t0)
    t2 = F.mul_by_3b(z1)
    z3 = F.add(t1, t2); t1 = F.sub(t1, t2)
    y3 = F.mul_by_3b(y3)
    x3 = F.mul(t4, y3); t2 = F.mul(t3, t1); x3 = F.sub(t2, x3)
    y3 = F.mul(y3, t0); t1 = F.mul(t1, z3); y3 = F.add(t1, y3)
    t0 = F.mul(t0, t3); z3 = F.mul(z3, t4); z3 = F.add(z3, t0)
    return p if inf else (x3, y3, z3)


def _neg(F, p): return (p[0], F.neg(p[1]), p[2])


def _multiply(F, p, scalar_le_bytes):
    """g1.rs:754-774 / g2.rs:825-845: MSB-first double-and-add over 255 bits of the 32-byte LE scalar."""
    acc = _identity(F)
    bits = [(byte >> i) & 1 for byte in reversed(scalar_le_bytes) for i in reversed(range(8))][1:]
    for bit in bits:
        acc = _double(F, acc)
        if bit:                                  # conditional_select(&acc, &(acc + self), bit)
            acc = _add(F, acc, p)
    return acc


def _to_affine(F, p):
    """g1.rs:49-63 / g2.rs:50-64: (x, y, infinity); identity -> (0, 1, True)."""
    zinv = F.inv(p[2])
    if zinv is None:
        return (F.zero, F.one, True)
    return (F.mul(p[0], zinv), F.mul(p[1], zinv), False)


def _from_affine(F, a):
    """g1.rs:463-471 / g2.rs: From<&Affine> for Projective."""
    return (a[0], a[1], F.zero if a[2] else F.one)


def _proj_eq(F, p, q):
    """g1.rs:479-496: cross-multiplied equality."""
    x1 = F.mul(p[0], q[2]); x2 = F.mul(q[0], p[2])
    y1 = F.mul(p[1], q[2]); y2 = F.mul(q[1], p[2])
    pz, qz = F.is_zero(p[2]), F.is_zero(q[2])
    return (pz and qz) or ((not pz) and (not qz) and x1 == x2 and y1 == y2)


def _batch_normalize(F, ps):
    """g1.rs:806-839 / g2.rs:951-984 (Montgomery's trick, identities skipped)."""
    acc = F.one
    pref = []
    for p in ps:
        pref.append(acc)
        if not F.is_zero(p[2]):
            acc = F.mul(acc, p[2])
    acc = F.inv(acc)
    out = [None] * len(ps)
    for i in reversed(range(len(ps))):
        p = ps[i]
        skip = F.is_zero(p[2])
        tmp = F.mul(pref[i], acc)
        if not skip:
            acc = F.mul(acc, p[2])
        out[i] = (F.zero, F.one, True) if skip else (F.mul(p[0], tmp), F.mul(p[1], tmp), False)
    return out


# -- G1 -------------------------------------------------------------------------------------------
# g1.rs:197-217 generator (decimal values in src/notes/design.rs:13-18)
G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    False,
)
G1_IDENTITY_AFF = (0, 1, True)


def g1_identity(): return _identity(_FpOps)
def g1_double(p): return _double(_FpOps, p)
def g1_add(p, q): return _add(_FpOps, p, q)
def g1_add_mixed(p, q): return _add_mixed(_FpOps, p, q)
def g1_neg(p): return _neg(_FpOps, p)
def g1_to_affine(p): return _to_affine(_FpOps, p)
def g1_from_affine(a): return _from_affine(_FpOps, a)
def g1_eq(p, q): return _proj_eq(_FpOps, p, q)
def g1_batch_normalize(ps): return _batch_normalize(_FpOps, ps)


def scalar_to_bytes(s):
    """scalar.rs:284-296: canonical little-endian 32 bytes of s mod r."""
    return (int(s) % R_ORDER).to_bytes(32, "little")


def g1_mul(p, s):
    """`&G1Projective * &Scalar` (g1.rs:556-562)."""
    return _multiply(_FpOps, p, scalar_to_bytes(s))


def g1_affine_mul(a, s):
    """`&G1Affine * &Scalar` (g1.rs:573-579)."""
    return g1_mul(g1_from_affine(a), s)


def g1_sum(points):
    """`Sum for G1Projective` (g1.rs:161-171)."""
    acc = g1_identity()
    for q in points:
        acc = g1_add(acc, q)
    return acc


def g1_msm(bases_aff, scalars):
    """The reference's only expression of an MSM: sum_i (P_i * s_i)."""
    return g1_sum(g1_affine_mul(b, s) for b, s in zip(bases_aff, scalars))


def g1_is_on_curve(a):
    """g1.rs:410-413."""
    return a[2] or (a[1] * a[1] - a[0] * a[0] * a[0]) % P == 4


def g1_to_uncompressed(a):
    """g1.rs:246-260."""
    x, y, inf = a
    res = bytearray(fp_to_bytes(0 if inf else x) + fp_to_bytes(0 if inf else y))
    if inf:
        res[0] |= 1 << 6
    return bytes(res)


def g1_to_compressed(a):
    """g1.rs:221-242."""
    x, y, inf = a
    res = bytearray(fp_to_bytes(0 if inf else x))
    res[0] |= 1 << 7
    if inf:
        res[0] |= 1 << 6
    if (not inf) and fp_lex_largest(y):
        res[0] |= 1 << 5
    return bytes(res)


def g1_from_uncompressed_unchecked(b):
    """g1.rs:273-322.  Returns None where the reference returns CtOption::none."""
    c, i, s = (b[0] >> 7) & 1, (b[0] >> 6) & 1, (b[0] >> 5) & 1
    x = fp_from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:48]))
    y = fp_from_bytes(bytes(b[48:96]))
    if x is None or y is None:
        return None
    ok = ((not i) or (x == 0 and y == 0)) and (not c) and (not s)
    if not ok:
        return None
    return G1_IDENTITY_AFF if i else (x, y, False)


def g1_from_compressed_unchecked(b):
    """g1.rs:336-390."""
    c, i, s = (b[0] >> 7) & 1, (b[0] >> 6) & 1, (b[0] >> 5) & 1
    x = fp_from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:48]))
    if x is None:
        return None
    if i and c and (not s) and x == 0:
        return G1_IDENTITY_AFF
    y = fp_sqrt((x * x * x + 4) % P)
    if y is None or i or not c:
        return None
    if fp_lex_largest(y) != bool(s):
        y = fp_neg(y)
    return (x, y, False)


# -- G2 -------------------------------------------------------------------------------------------
# g2.rs:210-250 generator (decimal values in src/notes/design.rs)
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
    False,
)
G2_IDENTITY_AFF = (FP2_ZERO, FP2_ONE, True)


def g2_identity(): return _identity(_Fp2Ops)
def g2_double(p): return _double(_Fp2Ops, p)
def g2_add(p, q): return _add(_Fp2Ops, p, q)
def g2_add_mixed(p, q): return _add_mixed(_Fp2Ops, p, q)
def g2_neg(p): return _neg(_Fp2Ops, p)
def g2_to_affine(p): return _to_affine(_Fp2Ops, p)
def g2_from_affine(a): return _from_affine(_Fp2Ops, a)
def g2_eq(p, q): return _proj_eq(_Fp2Ops, p, q)
def g2_batch_normalize(ps): return _batch_normalize(_Fp2Ops, ps)
def g2_mul(p, s): return _multiply(_Fp2Ops, p, scalar_to_bytes(s))       # g2.rs:609-615
def g2_affine_mul(a, s): return g2_mul(g2_from_affine(a), s)              # g2.rs:626-632


def g2_sum(points):
    acc = g2_identity()
    for q in points:
        acc = g2_add(acc, q)
    return acc


def g2_msm(bases_aff, scalars):
    return g2_sum(g2_affine_mul(b, s) for b, s in zip(bases_aff, scalars))


def g2_is_on_curve(a):
    """g2.rs:484-488."""
    x, y, inf = a
    return inf or fp2_sub(fp2_sqr(y), fp2_mul(fp2_sqr(x), x)) == (4, 4)


def g2_to_uncompressed(a):
    """g2.rs:284-299: x.c1 | x.c0 | y.c1 | y.c0."""
    x, y, inf = a
    if inf:
        x, y = FP2_ZERO, FP2_ZERO
    res = bytearray(fp_to_bytes(x[1]) + fp_to_bytes(x[0]) + fp_to_bytes(y[1]) + fp_to_bytes(y[0]))
    if inf:
        res[0] |= 1 << 6
    return bytes(res)


def g2_to_compressed(a):
    """g2.rs:254-280."""
    x, y, inf = a
    if inf:
        x = FP2_ZERO
    res = bytearray(fp_to_bytes(x[1]) + fp_to_bytes(x[0]))
    res[0] |= 1 << 7
    if inf:
        res[0] |= 1 << 6
    if (not inf) and fp2_lex_largest(y):
        res[0] |= 1 << 5
    return bytes(res)


def g2_from_uncompressed_unchecked(b):
    """g2.rs:311-380."""
    c, i, s = (b[0] >> 7) & 1, (b[0] >> 6) & 1, (b[0] >> 5) & 1
    xc1 = fp_from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:48]))
    xc0 = fp_from_bytes(bytes(b[48:96]))
    yc1 = fp_from_bytes(bytes(b[96:144]))
    yc0 = fp_from_bytes(bytes(b[144:192]))
    if None in (xc1, xc0, yc1, yc0):
        return None
    ok = ((not i) or (xc0 == 0 and xc1 == 0 and yc0 == 0 and yc1 == 0)) and (not c) and (not s)
    if not ok:
        return None
    return G2_IDENTITY_AFF if i else ((xc0, xc1), (yc0, yc1), False)


def g2_from_compressed_unchecked(b):
    """g2.rs:394-464."""
    c, i, s = (b[0] >> 7) & 1, (b[0] >> 6) & 1, (b[0] >> 5) & 1
    xc1 = fp_from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:48]))
    xc0 = fp_from_bytes(bytes(b[48:96]))
    if xc1 is None or xc0 is None:
        return None
    x = (xc0, xc1)
    if i and c and (not s) and fp2_is_zero(x):
        return G2_IDENTITY_AFF
    y = fp2_sqrt(fp2_add(fp2_mul(fp2_sqr(x), x), (4, 4)))
    if y is None or i or not c:
        return None
    if fp2_lex_largest(y) != bool(s):
        y = fp2_neg(y)
    return (x, y, False)


# --------------------------------------------------------------------------------------------------
# Pairing  (src/pairings.rs)
# --------------------------------------------------------------------------------------------------
def doubling_step(r):
    """pairings.rs:709-738 (CLN Algorithm 26).  r = [x, y, z] over Fp2 (mutated); returns line coeffs."""
    add, sub, sqr, mul, neg = fp2_add, fp2_sub, fp2_sqr, fp2_mul, fp2_neg
    tmp0 = sqr(r[0])
    tmp1 = sqr(r[1])
    tmp2 = sqr(tmp1)
    tmp3 = sub(sub(sqr(add(tmp1, r[0])), tmp0), tmp2)
    tmp3 = add(tmp3, tmp3)
    tmp4 = add(add(tmp0, tmp0), tmp0)
    tmp6 = add(r[0], tmp4)
    tmp5 = sqr(tmp4)
    zsquared = sqr(r[2])
    r[0] = sub(sub(tmp5, tmp3), tmp3)
    r[2] = sub(sub(sqr(add(r[2], r[1])), tmp1), zsquared)
    r[1] = mul(sub(tmp3, r[0]), tmp4)
    tmp2 = add(tmp2, tmp2); tmp2 = add(tmp2, tmp2); tmp2 = add(tmp2, tmp2)
    r[1] = sub(r[1], tmp2)
    tmp3 = mul(tmp4, zsquared)
    tmp3 = add(tmp3, tmp3)
    tmp3 = neg(tmp3)
    tmp6 = sub(sub(sqr(tmp6), tmp0), tmp5)
    tmp1 = add(tmp1, tmp1); tmp1 = add(tmp1, tmp1)
    tmp6 = sub(tmp6, tmp1)
    tmp0 = mul(r[2], zsquared)
    tmp0 = add(tmp0, tmp0)
    return (tmp0, tmp3, tmp6)


def addition_step(r, q):
    """pairings.rs:740-770 (CLN Algorithm 27).  q = (x, y, inf) affine over Fp2."""
    add, sub, sqr, mul, neg = fp2_add, fp2_sub, fp2_sqr, fp2_mul, fp2_neg
    qx, qy = q[0], q[1]
    zsquared = sqr(r[2])
    ysquared = sqr(qy)
    t0 = mul(zsquared, qx)
    t1 = mul(sub(sub(sqr(add(qy, r[2])), ysquared), zsquared), zsquared)
    t2 = sub(t0, r[0])
    t3 = sqr(t2)
    t4 = add(t3, t3); t4 = add(t4, t4)
    t5 = mul(t4, t2)
    t6 = sub(sub(t1, r[1]), r[1])
    t9 = mul(t6, qx)
    t7 = mul(t4, r[0])
    r[0] = sub(sub(sub(sqr(t6), t5), t7), t7)
    r[2] = sub(sub(sqr(add(r[2], t2)), zsquared), t3)
    t10 = add(qy, r[2])
    t8 = mul(sub(t7, r[0]), t6)
    t0 = mul(r[1], t5)
    t0 = add(t0, t0)
    r[1] = sub(t8, t0)
    t10 = sub(sqr(t10), ysquared)
    ztsquared = sqr(r[2])
    t10 = sub(t10, ztsquared)
    t9 = sub(add(t9, t9), t10)
    t10 = add(r[2], r[2])
    t6 = neg(t6)
    t1 = add(t6, t6)
    return (t10, t1, t9)


def ell(f, coeffs, p_aff):
    """pairings.rs:696-707."""
    c0 = fp2_mul_fp(coeffs[0], p_aff[1])
    c1 = fp2_mul_fp(coeffs[1], p_aff[0])
    return fp12_mul_by_014(f, coeffs[2], c1, c0)


# bits of BLS_X >> 1, MSB first, leading one skipped (pairings.rs:668-694)
_X_BITS = []
_found = False
for _b in reversed(range(64)):
    _i = ((BLS_X >> 1) >> _b) & 1
    if not _found:
        _found = bool(_i)
        continue
    _X_BITS.append(_i)
assert len(_X_BITS) == 62 and sum(_X_BITS) == 5


def g2_prepare(q_aff):
    """`G2Prepared::from` (pairings.rs:504-546): (infinity, 68 line coefficient triples)."""
    inf = q_aff[2]
    q = G2_GEN if inf else q_aff
    cur = [q[0], q[1], FP2_ONE]
    coeffs = []
    for bit in _X_BITS:
        coeffs.append(doubling_step(cur))
        if bit:
            coeffs.append(addition_step(cur, q))
    coeffs.append(doubling_step(cur))
    assert len(coeffs) == 68
    return (inf, coeffs)


def multi_miller_loop(terms):
    """pairings.rs:554-603.  terms = [(G1 affine, G2Prepared)].  Returns the raw MillerLoopResult Fp12."""
    f = FP12_ONE
    idx = 0

    def step(f, idx):
        for p_aff, (qinf, coeffs) in terms:
            if not (p_aff[2] or qinf):
                f = ell(f, coeffs[idx], p_aff)
        return f

    for bit in _X_BITS:
        f = step(f, idx); idx += 1
        if bit:
            f = step(f, idx); idx += 1
        f = fp12_sqr(f)
    f = step(f, idx); idx += 1
    assert idx == 68
    if BLS_X_IS_NEGATIVE:
        f = fp12_conj(f)
    return f


def miller_loop(p_aff, q_aff):
    """The unprepared Miller loop inside `pairing` (pairings.rs:607-653), incl. identity handling.
    Returns the raw MillerLoopResult Fp12 (Fp12::one() if either input is the identity)."""
    either = p_aff[2] or q_aff[2]
    if either:
        return FP12_ONE
    cur = [q_aff[0], q_aff[1], FP2_ONE]
    f = FP12_ONE
    for bit in _X_BITS:
        f = ell(f, doubling_step(cur), p_aff)
        if bit:
            f = ell(f, addition_step(cur, q_aff), p_aff)
        f = fp12_sqr(f)
    f = ell(f, doubling_step(cur), p_aff)
    if BLS_X_IS_NEGATIVE:
        f = fp12_conj(f)
    return f


def _fp4_square(a, b):
    """pairings.rs:50-62."""
    t0 = fp2_sqr(a)
    t1 = fp2_sqr(b)
    t2 = fp2_mul_by_nonresidue(t1)
    c0 = fp2_add(t2, t0)
    t2 = fp2_sqr(fp2_add(a, b))
    t2 = fp2_sub(t2, t0)
    c1 = fp2_sub(t2, t1)
    return c0, c1


def cyclotomic_square(f):
    """pairings.rs:66-112."""
    z0, z4, z3 = f[0]
    z2, z1, z5 = f[1]
    t0, t1 = _fp4_square(z0, z1)
    z0 = fp2_sub(t0, z0); z0 = fp2_add(fp2_add(z0, z0), t0)
    z1 = fp2_add(t1, z1); z1 = fp2_add(fp2_add(z1, z1), t1)
    t0, t1 = _fp4_square(z2, z3)
    t2, t3 = _fp4_square(z4, z5)
    z4 = fp2_sub(t0, z4); z4 = fp2_add(fp2_add(z4, z4), t0)
    z5 = fp2_add(t1, z5); z5 = fp2_add(fp2_add(z5, z5), t1)
    t0 = fp2_mul_by_nonresidue(t3)
    z2 = fp2_add(t0, z2); z2 = fp2_add(fp2_add(z2, z2), t0)
    z3 = fp2_sub(t2, z3); z3 = fp2_add(fp2_add(z3, z3), t2)
    return ((z0, z4, z3), (z2, z1, z5))


def cyclotomic_exp(f):
    """pairings.rs:114-132 (sic `cycolotomic_exp`): f^x then conjugate (x negative)."""
    tmp = FP12_ONE
    found_one = False
    for b in reversed(range(64)):
        i = (BLS_X >> b) & 1
        if found_one:
            tmp = cyclotomic_square(tmp)
        else:
            found_one = bool(i)
        if i:
            tmp = fp12_mul(tmp, f)
    return fp12_conj(tmp)


def final_exponentiation(f):
    """pairings.rs:48-176.  Raises to 3*(p^4-p^2+1)/r * (p^6-1)(p^2+1) -- note the factor 3."""
    t0 = f
    for _ in range(6):
        t0 = fp12_frobenius(t0)
    t1 = fp12_inv(f)
    assert t1 is not None
    t2 = fp12_mul(t0, t1)
    t1 = t2
    t2 = fp12_frobenius(fp12_frobenius(t2))
    t2 = fp12_mul(t2, t1)
    t1 = fp12_conj(cyclotomic_square(t2))
    t3 = cyclotomic_exp(t2)
    t4 = cyclotomic_square(t3)
    t5 = fp12_mul(t1, t3)
    t1 = cyclotomic_exp(t5)
    t0 = cyclotomic_exp(t1)
    t6 = cyclotomic_exp(t0)
    t6 = fp12_mul(t6, t4)
    t4 = cyclotomic_exp(t6)
    t5 = fp12_conj(t5)
    t4 = fp12_mul(t4, fp12_mul(t5, t2))
    t5 = fp12_conj(t2)
    t1 = fp12_mul(t1, t2)
    t1 = fp12_frobenius(fp12_frobenius(fp12_frobenius(t1)))
    t6 = fp12_mul(t6, t5)
    t6 = fp12_frobenius(t6)
    t3 = fp12_mul(t3, t0)
    t3 = fp12_frobenius(fp12_frobenius(t3))
    t3 = fp12_mul(t3, t1)
    t3 = fp12_mul(t3, t6)
    return fp12_mul(t3, t4)


def pairing(p_aff, q_aff):
    """pairings.rs:607-653 -> Gt (an Fp12)."""
    return final_exponentiation(miller_loop(p_aff, q_aff))


def gt_mul_scalar(g, s):
    """`&Gt * &Scalar` (pairings.rs:297-322): double-and-add in the additive notation (= power)."""
    acc = FP12_ONE
    bits = [(byte >> i) & 1 for byte in reversed(scalar_to_bytes(s)) for i in reversed(range(8))][1:]
    for bit in bits:
        acc = fp12_sqr(acc)
        if bit:
            acc = fp12_mul(acc, g)
    return acc


# --------------------------------------------------------------------------------------------------
# deterministic synthetic inputs shared by tests / bench
# --------------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def scalar(self):
        """Uniform in [0, r): 32 random bytes, top bit cleared, rejection-sampled."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < R_ORDER:
                return v


# --------------------------------------------------------------------------------------------------
# Point validation / checked deserialisation (the step in front of the hot path)
# --------------------------------------------------------------------------------------------------
BETA = pow(2, (P - 1) // 3, P)                      # g1.rs:421-428 (equals the reference literal; tests check it)
PSI_COEFF_X = fp2_inv(fp2_pow((1, 1), (P - 1) // 3))   # g2.rs:848-859
PSI_COEFF_Y = fp2_inv(fp2_pow((1, 1), (P - 1) // 2))   # g2.rs:860-880


def _mul_by_x(F, p, add, dbl, neg):
    """g1.rs:777-795 / g2.rs:914-931: multiply by BLS_X (negative)."""
    xself = _identity(F)
    x = BLS_X >> 1
    tmp = p
    while x != 0:
        tmp = dbl(tmp)
        if x % 2 == 1:
            xself = add(xself, tmp)
        x >>= 1
    return neg(xself) if BLS_X_IS_NEGATIVE else xself


def g1_mul_by_x(p): return _mul_by_x(_FpOps, p, g1_add, g1_double, g1_neg)
def g2_mul_by_x(p): return _mul_by_x(_Fp2Ops, p, g2_add, g2_double, g2_neg)


def g1_is_torsion_free(a):
    """g1.rs:401-410: endomorphism(P) == -[x^2] P."""
    m = g1_neg(g1_mul_by_x(g1_mul_by_x(g1_from_affine(a))))
    e = ((a[0] * BETA) % P, a[1], a[2])
    return g1_eq(m, g1_from_affine(e))


def g2_psi(p):
    """g2.rs:847-890."""
    return (fp2_mul(fp2_conj(p[0]), PSI_COEFF_X), fp2_mul(fp2_conj(p[1]), PSI_COEFF_Y), fp2_conj(p[2]))


def g2_is_torsion_free(a):
    """g2.rs:475-482: psi(P) == [x] P."""
    p = g2_from_affine(a)
    return g2_eq(g2_psi(p), g2_mul_by_x(p))


def g1_from_compressed(b):
    """g1.rs:326-332."""
    p = g1_from_compressed_unchecked(b)
    return p if (p is not None and g1_is_torsion_free(p)) else None


def g1_from_uncompressed(b):
    """g1.rs:264-267."""
    p = g1_from_uncompressed_unchecked(b)
    return p if (p is not None and g1_is_on_curve(p) and g1_is_torsion_free(p)) else None


def g2_from_compressed(b):
    """g2.rs:390-395."""
    p = g2_from_compressed_unchecked(b)
    return p if (p is not None and g2_is_torsion_free(p)) else None


def g2_from_uncompressed(b):
    """g2.rs:303-306."""
    p = g2_from_uncompressed_unchecked(b)
    return p if (p is not None and g2_is_on_curve(p) and g2_is_torsion_free(p)) else None


# --------------------------------------------------------------------------------------------------
# Scalar field Fr (src/scalar.rs) -- SURVEY.md 8(f) rank 3: the caller-side producer of MSM scalars.
# A `Scalar` is four little-endian u64 limbs in Montgomery form with R = 2^256 (scalar.rs:23-27,155-165);
# every operation returns the canonical representative, so parity is on those limbs.
# --------------------------------------------------------------------------------------------------
FR_S = 32                                   # scalar.rs:191  (2^S * t = r - 1, t odd)
FR_GENERATOR = 7                            # scalar.rs:99-105
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_ORDER - 1) >> FR_S, R_ORDER)     # scalar.rs:193-205 (GENERATOR^t)


def fr_to_mont_limbs(x):
    v = (int(x) % R_ORDER) * FR_MONT_R % R_ORDER
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def fr_from_mont_limbs(limbs):
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    assert v < R_ORDER, "non-canonical Scalar limbs"
    return v * pow(FR_MONT_R, -1, R_ORDER) % R_ORDER


def fr_add(a, b): return (a + b) % R_ORDER            # scalar.rs:435-449
def fr_sub(a, b): return (a - b) % R_ORDER            # scalar.rs:420-432
def fr_neg(a): return (-a) % R_ORDER                  # scalar.rs:552-568
def fr_mul(a, b): return a * b % R_ORDER              # scalar.rs:452-503 (schoolbook + montgomery_reduce :506-550)
def fr_sqr(a): return a * a % R_ORDER                 # scalar.rs:334-369
def fr_double(a): return 2 * a % R_ORDER              # scalar.rs:246-250


def fr_pow(a, e):
    """scalar.rs:371-404 (`pow` / `pow_vartime`): e is an integer exponent (the reference takes [u64; 4])."""
    return pow(a, e, R_ORDER)


def fr_inv(a):
    """scalar.rs:573-628: a^(r-2); None for zero (CtOption::none)."""
    return None if a % R_ORDER == 0 else pow(a, R_ORDER - 2, R_ORDER)


def fr_from_bytes_wide(b):
    """scalar.rs:300-331: 64 little-endian bytes reduced mod r."""
    assert len(b) == 64
    return int.from_bytes(bytes(b), "little") % R_ORDER


# ---- `Scalar` <-> bytes on the reference's own limbs -----------------------------------------------
# The three functions below follow scalar.rs limb by limb (mac / adc / sbb on 64-bit words) instead of using Python's big
# integers, so that they pin the DEVICE conversions (scalar.hip.h) independently of fr_to_mont_limbs / fr_from_mont_limbs above.
_M64 = 0xFFFFFFFFFFFFFFFF
FR_MODULUS_LIMBS = [(R_ORDER >> (64 * i)) & _M64 for i in range(4)]                    # scalar.rs:76-81
FR_INV = (-pow(R_ORDER, -1, 1 << 64)) % (1 << 64)                                       # scalar.rs:156 (asserted in tests against the literal)
FR_R2_LIMBS = [((1 << 512) % R_ORDER >> (64 * i)) & _M64 for i in range(4)]             # scalar.rs:167-172
FR_R3_LIMBS = [((1 << 768) % R_ORDER >> (64 * i)) & _M64 for i in range(4)]             # scalar.rs:174-180


def _mac(a, b, c, carry):          # util.rs:14-20: a + b * c + carry -> (low, high)
    t = a + b * c + carry
    return t & _M64, t >> 64


def _adc(a, b, carry):             # util.rs:1-6
    t = a + b + carry
    return t & _M64, t >> 64


def _sbb(a, b, borrow):            # util.rs:8-12: a - (b + (borrow >> 63)) -> (low, borrow word)
    t = a - (b + (borrow >> 63))
    return t & _M64, (t >> 64) & _M64


def scalar_limbs_sub(a, b):
    """scalar.rs:420-432 `sub`: a - b, the modulus added back under the borrow mask."""
    d, borrow = [0] * 4, 0
    for i in range(4):
        d[i], borrow = _sbb(a[i], b[i], borrow)
    out, carry = [0] * 4, 0
    for i in range(4):
        out[i], carry = _adc(d[i], FR_MODULUS_LIMBS[i] & borrow, carry)
    return out


def scalar_limbs_add(a, b):
    """scalar.rs:435-449 `add`: limb-wise sum, then `sub(&MODULUS)`."""
    d, carry = [0] * 4, 0
    for i in range(4):
        d[i], carry = _adc(a[i], b[i], carry)
    return scalar_limbs_sub(d, FR_MODULUS_LIMBS)


def scalar_montgomery_reduce(r):
    """scalar.rs:506-550: HAC 14.32 on eight limbs r[0..7], result = r / 2^256 mod q as four limbs."""
    r = list(r)
    carry2 = 0
    for i in range(4):
        k = (r[i] * FR_INV) & _M64
        _, carry = _mac(r[i], k, FR_MODULUS_LIMBS[0], 0)
        for j in range(1, 4):
            r[i + j], carry = _mac(r[i + j], k, FR_MODULUS_LIMBS[j], carry)
        r[i + 4], carry2 = _adc(r[i + 4], carry2, carry)
    return scalar_limbs_sub(r[4:8], FR_MODULUS_LIMBS)


def scalar_limbs_mul(a, b):
    """scalar.rs:452-503 `mul`: schoolbook 4 x 4 into eight limbs, then montgomery_reduce."""
    r = [0] * 8
    for i in range(4):
        carry = 0
        for j in range(4):
            r[i + j], carry = _mac(r[i + j], a[i], b[j], carry)
        r[i + 4] = carry
    return scalar_montgomery_reduce(r)


def scalar_limbs_to_bytes(limbs):
    """scalar.rs:284-296 `to_bytes`: montgomery_reduce(l0, l1, l2, l3, 0, 0, 0, 0), little-endian."""
    t = scalar_montgomery_reduce(list(limbs) + [0, 0, 0, 0])
    return b"".join(int(x).to_bytes(8, "little") for x in t)


def scalar_limbs_from_bytes(b):
    """scalar.rs:256-280 `from_bytes`: (limbs of tmp * R2, is_some) -- the value is computed either way, as in the reference."""
    assert len(b) == 32
    tmp = [int.from_bytes(bytes(b[8 * i:8 * i + 8]), "little") for i in range(4)]
    borrow = 0
    for i in range(4):
        _, borrow = _sbb(tmp[i], FR_MODULUS_LIMBS[i], borrow)
    return scalar_limbs_mul(tmp, FR_R2_LIMBS), bool(borrow & 1)


def scalar_limbs_from_bytes_wide(b):
    """scalar.rs:300-331 `from_bytes_wide` -> `from_u512`: d0 * R2 + d1 * R3."""
    assert len(b) == 64
    l = [int.from_bytes(bytes(b[8 * i:8 * i + 8]), "little") for i in range(8)]
    return scalar_limbs_add(scalar_limbs_mul(l[0:4], FR_R2_LIMBS), scalar_limbs_mul(l[4:8], FR_R3_LIMBS))


def fr_omega(log_n):
    """primitive 2^log_n-th root of unity: ROOT_OF_UNITY^(2^(S - log_n)) (how ff::PrimeField::ROOT_OF_UNITY is used)."""
    assert 0 <= log_n <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), R_ORDER)


def fr_ntt(values, inverse=False):
    """Number-theoretic transform over Fr, natural order in and out:
         forward  y_k = sum_j x_j w^(jk)          inverse  x_j = n^-1 sum_k y_k w^(-jk),   w = fr_omega(log2 n).
    The reference crate provides the field and ROOT_OF_UNITY (scalar.rs:193-205, 688-713) but no transform; this is the
    definition the GPU kernels are tested against (recursive radix-2 evaluation of exactly that sum)."""
    n = len(values)
    assert n and n & (n - 1) == 0
    w = fr_omega(n.bit_length() - 1)
    if inverse:
        w = pow(w, -1, R_ORDER)

    def rec(x, w):
        m = len(x)
        if m == 1:
            return list(x)
        w2 = w * w % R_ORDER
        ev, od = rec(x[0::2], w2), rec(x[1::2], w2)
        out = [0] * m
        t = 1
        for k in range(m // 2):
            u = od[k] * t % R_ORDER
            out[k] = (ev[k] + u) % R_ORDER
            out[k + m // 2] = (ev[k] - u) % R_ORDER
            t = t * w % R_ORDER
        return out

    y = rec([int(v) % R_ORDER for v in values], w)
    if inverse:
        ninv = pow(n, -1, R_ORDER)
        y = [v * ninv % R_ORDER for v in y]
    return y


def fr_ntt_naive(values, inverse=False):
    """the defining O(n^2) sums (used to pin fr_ntt itself on small sizes)."""
    n = len(values)
    w = fr_omega(n.bit_length() - 1)
    if inverse:
        w = pow(w, -1, R_ORDER)
    y = [sum(int(values[j]) * pow(w, j * k, R_ORDER) for j in range(n)) % R_ORDER for k in range(n)]
    if inverse:
        ninv = pow(n, -1, R_ORDER)
        y = [v * ninv % R_ORDER for v in y]
    return y
