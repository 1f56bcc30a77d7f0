"""Pin the oracle (oracle/bls12_381_ref.py) to the reference's own known-answer material.

Fixtures were extracted from /root/reference by tests/golden/make_golden.py (committed with the data):
field KATs of src/fp.rs / src/fp2.rs, group KATs of src/g1.rs / src/g2.rs, the module constants, the
four k*G golden files (src/tests/mod.rs:3-76) and the RELIC pairing constant (src/pairings.rs:359-475).
All literals are Montgomery limbs (R = 2^384); `fp_from_mont_limbs` decodes them.
"""
import os

import pytest

from oracle import bls12_381_ref as o

F = o.fp_from_mont_limbs


def fp2s(arrs):
    """consecutive limb arrays -> list of Fp2 (c0, c1)"""
    return [(F(arrs[i]), F(arrs[i + 1])) for i in range(0, len(arrs), 2)]


def test_constants(kats):
    c = kats["consts"]
    assert sum(l << (64 * i) for i, l in enumerate(c["fp.MODULUS"])) == o.P
    assert c["fp.INV"] == (-pow(o.P, -1, 1 << 64)) % (1 << 64)
    assert F(c["fp.R"]) == 1
    assert F(c["fp.R2"]) == o.MONT_R and F(c["fp.R3"]) == (o.MONT_R * o.MONT_R) % o.P
    assert F(c["g1.B"]) == 4
    assert [F(x) for x in c["g1.GENERATOR"][:2]] == [o.G1_GEN[0], o.G1_GEN[1]]
    assert fp2s(c["g2.B"]) == [(4, 4)]
    g2 = fp2s(c["g2.GENERATOR"])
    assert g2[0] == o.G2_GEN[0] and g2[1] == o.G2_GEN[1]
    assert sum(l << (64 * i) for i, l in enumerate(c["scalar.MODULUS"])) == o.R_ORDER
    assert c["lib.BLS_X"] == o.BLS_X
    # Frobenius coefficients are recomputed from their definition in the oracle
    assert o.FROB6_C1 == (0, F(c["fp6.FROBENIUS_C1"][0]))
    assert o.FROB6_C2 == (F(c["fp6.FROBENIUS_C2"][0]), 0)
    assert o.FROB12_C1 == (F(c["fp12.FROBENIUS_C1"][0]), F(c["fp12.FROBENIUS_C1"][1]))
    beta = F(c["g1.BETA"])
    assert beta != 1 and pow(beta, 3, o.P) == 1
    assert o.g1_is_on_curve(o.G1_GEN) and o.g2_is_on_curve(o.G2_GEN)


def test_fp_kats(kats):
    t = kats["tests"]
    a, b = map(F, t["fp.test_squaring"]["fp"]);               assert o.fp_sqr(a) == b
    a, b, c = map(F, t["fp.test_multiplication"]["fp"]);      assert o.fp_mul(a, b) == c
    a, b, c = map(F, t["fp.test_addition"]["fp"]);            assert o.fp_add(a, b) == c
    a, b, c = map(F, t["fp.test_subtraction"]["fp"]);         assert o.fp_sub(a, b) == c
    a, b = map(F, t["fp.test_negation"]["fp"]);               assert o.fp_neg(a) == b
    a, b = map(F, t["fp.test_inversion"]["fp"]);              assert o.fp_inv(a) == b
    assert o.fp_inv(0) is None
    a, b = map(F, t["fp.test_sqrt"]["fp"])                    # a = 4, sqrt is +-2
    assert o.fp_sqrt(a) in (b, o.fp_neg(b)) and o.fp_sqr(o.fp_sqrt(a)) == a
    x, y, z = map(F, t["fp.test_lexicographic_largest"]["fp"])
    assert not o.fp_lex_largest(x) and o.fp_lex_largest(y) and o.fp_lex_largest(z)
    assert not o.fp_lex_largest(0) and not o.fp_lex_largest(1)


def test_fp2_kats(kats):
    t = kats["tests"]
    a, b = fp2s(t["fp2.test_squaring"]["fp"]);                assert o.fp2_sqr(a) == b
    a, b, c = fp2s(t["fp2.test_multiplication"]["fp"]);       assert o.fp2_mul(a, b) == c
    a, b, c = fp2s(t["fp2.test_addition"]["fp"]);             assert o.fp2_add(a, b) == c
    a, b, c = fp2s(t["fp2.test_subtraction"]["fp"]);          assert o.fp2_sub(a, b) == c
    a, b = fp2s(t["fp2.test_negation"]["fp"]);                assert o.fp2_neg(a) == b
    a, b = fp2s(t["fp2.test_inversion"]["fp"]);               assert o.fp2_inv(a) == b
    assert o.fp2_inv((0, 0)) is None
    s = fp2s(t["fp2.test_sqrt"]["fp"])                        # (a, b, c): a.sqrt()^2 == a, b.sqrt()^2 == b, c has no root
    for v in s[:2]:
        assert o.fp2_sqr(o.fp2_sqrt(v)) == v
    assert o.fp2_sqrt(s[2]) is None


def test_tower_identities(kats):
    """src/fp6.rs:376-561 / src/fp12.rs:265-649 check algebraic identities on fixed elements; same here."""
    vals = kats["tests"]["fp6.test_arithmetic"]["fp"]
    a, b, c = [tuple(fp2s(vals[6 * i:6 * i + 6])) for i in range(3)]
    assert o.fp6_sqr(a) == o.fp6_mul(a, a) and o.fp6_sqr(b) == o.fp6_mul(b, b)
    assert o.fp6_mul(o.fp6_add(a, b), o.fp6_sqr(c)) == o.fp6_add(o.fp6_mul(o.fp6_mul(c, c), a), o.fp6_mul(o.fp6_mul(c, c), b))
    assert o.fp6_mul(o.fp6_inv(a), o.fp6_inv(b)) == o.fp6_inv(o.fp6_mul(a, b))
    assert o.fp6_mul(o.fp6_inv(a), a) == o.FP6_ONE
    assert o.fp6_mul_by_1(a, b[1]) == o.fp6_mul(a, (o.FP2_ZERO, b[1], o.FP2_ZERO))
    assert o.fp6_mul_by_01(a, b[0], b[1]) == o.fp6_mul(a, (b[0], b[1], o.FP2_ZERO))
    f = a
    for _ in range(6):
        f = o.fp6_frobenius(f)
    assert f == a
    vals = kats["tests"]["fp12.test_arithmetic"]["fp"]
    x, y, z = [o.fp12_unflatten([F(v) for v in vals[12 * i:12 * i + 12]]) for i in range(3)]
    assert o.fp12_sqr(x) == o.fp12_mul(x, x)
    assert o.fp12_mul(o.fp12_inv(x), x) == o.FP12_ONE
    assert o.fp12_mul(o.fp12_mul(x, y), z) == o.fp12_mul(x, o.fp12_mul(y, z))
    f = x
    for _ in range(12):
        f = o.fp12_frobenius(f)
    assert f == x
    assert o.fp12_frobenius(x) != x
    sparse = (y[0][0], y[0][1], y[1][1])
    full = ((sparse[0], sparse[1], o.FP2_ZERO), (o.FP2_ZERO, sparse[2], o.FP2_ZERO))
    assert o.fp12_mul_by_014(x, *sparse) == o.fp12_mul(x, full)


def test_g1_kats(kats):
    t = kats["tests"]
    gen = o.g1_from_affine(o.G1_GEN)
    d = [F(v) for v in t["g1.test_doubling"]["fp"]]           # affine 2G
    assert o.g1_to_affine(o.g1_double(gen)) == (d[0], d[1], False)
    assert o.g1_to_affine(o.g1_double(o.g1_identity()))[2]
    # degenerate-case KATs of test_projective_addition / test_mixed_addition: beta * (x,y) + (x,y)
    # degenerate case (g1.rs:1372-1417 / :1493-1539): a = 4G, b = (beta^2 * a.x, -a.y) has the same x^3
    for name in ("g1.test_projective_addition", "g1.test_mixed_addition"):
        v = [F(x) for x in t[name]["fp"]]
        beta, cx, cy = (v[-3] * v[-3]) % o.P, v[-2], v[-1]
        a = o.g1_double(o.g1_double(gen))
        b = ((a[0] * beta) % o.P, (-a[1]) % o.P, a[2])
        assert o.g1_to_affine(o.g1_add(a, b)) == (cx, cy, False)
        assert o.g1_to_affine(o.g1_add_mixed(a, o.g1_to_affine(b))) == (cx, cy, False)
    # group laws
    g2_, g3 = o.g1_double(gen), o.g1_add(o.g1_double(gen), gen)
    assert o.g1_eq(o.g1_add(gen, gen), g2_) and o.g1_eq(o.g1_add_mixed(g2_, o.G1_GEN), g3)
    assert o.g1_eq(o.g1_add(g3, o.g1_neg(g3)), o.g1_identity())
    assert o.g1_eq(o.g1_add_mixed(o.g1_identity(), o.G1_GEN), gen)
    assert o.g1_eq(o.g1_add_mixed(gen, o.G1_IDENTITY_AFF), gen)
    # (g * a) * b == g * (a * b)   (g1.rs:1558-1595)
    sa, sb = [sum(l << (64 * i) for i, l in enumerate(x)) for x in t["g1.test_projective_scalar_multiplication"]["scalar"]]
    sa, sb = (sa * pow(o.FR_MONT_R, -1, o.R_ORDER)) % o.R_ORDER, (sb * pow(o.FR_MONT_R, -1, o.R_ORDER)) % o.R_ORDER
    assert o.g1_eq(o.g1_mul(o.g1_mul(gen, sa), sb), o.g1_mul(gen, (sa * sb) % o.R_ORDER))
    # batch_normalize over identity patterns (g1.rs:1691-1727)
    pts = [gen, o.g1_identity(), o.g1_double(gen), o.g1_identity()]
    assert o.g1_batch_normalize(pts) == [o.g1_to_affine(p) for p in pts]
    assert o.g1_eq(o.g1_mul(gen, o.R_ORDER - 1), o.g1_neg(gen))


def test_g2_kats(kats):
    t = kats["tests"]
    gen = o.g2_from_affine(o.G2_GEN)
    d = fp2s(t["g2.test_doubling"]["fp"])
    assert o.g2_to_affine(o.g2_double(gen)) == (d[0], d[1], False)
    g2_, g3 = o.g2_double(gen), o.g2_add(o.g2_double(gen), gen)
    assert o.g2_eq(o.g2_add(gen, gen), g2_) and o.g2_eq(o.g2_add_mixed(g2_, o.G2_GEN), g3)
    assert o.g2_eq(o.g2_add(g3, o.g2_neg(g3)), o.g2_identity())
    sa, sb = [sum(l << (64 * i) for i, l in enumerate(x)) for x in t["g2.test_projective_scalar_multiplication"]["scalar"]]
    sa, sb = (sa * pow(o.FR_MONT_R, -1, o.R_ORDER)) % o.R_ORDER, (sb * pow(o.FR_MONT_R, -1, o.R_ORDER)) % o.R_ORDER
    assert o.g2_eq(o.g2_mul(o.g2_mul(gen, sa), sb), o.g2_mul(gen, (sa * sb) % o.R_ORDER))
    pts = [gen, o.g2_identity(), o.g2_double(gen)]
    assert o.g2_batch_normalize(pts) == [o.g2_to_affine(p) for p in pts]


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_golden_multiples(group, golden_dir):
    """src/tests/mod.rs:3-76: record k of each file is k * generator (k = 0..999) -- all four encodings."""
    usz, csz = (96, 48) if group == "g1" else (192, 96)
    unc = open(os.path.join(golden_dir, f"{group}_uncompressed_valid_test_vectors.dat"), "rb").read()
    cmp_ = open(os.path.join(golden_dir, f"{group}_compressed_valid_test_vectors.dat"), "rb").read()
    assert len(unc) == 1000 * usz and len(cmp_) == 1000 * csz
    if group == "g1":
        e, gen_aff, add_mixed, to_aff = o.g1_identity(), o.G1_GEN, o.g1_add_mixed, o.g1_to_affine
        enc_u, enc_c, dec_u, dec_c = o.g1_to_uncompressed, o.g1_to_compressed, o.g1_from_uncompressed_unchecked, o.g1_from_compressed_unchecked
    else:
        e, gen_aff, add_mixed, to_aff = o.g2_identity(), o.G2_GEN, o.g2_add_mixed, o.g2_to_affine
        enc_u, enc_c, dec_u, dec_c = o.g2_to_uncompressed, o.g2_to_compressed, o.g2_from_uncompressed_unchecked, o.g2_from_compressed_unchecked
    for k in range(1000):
        a = to_aff(e)
        ub, cb = unc[k * usz:(k + 1) * usz], cmp_[k * csz:(k + 1) * csz]
        assert enc_u(a) == ub, k
        assert enc_c(a) == cb, k
        if k % 50 == 0 or k < 4:
            assert dec_u(ub) == a and dec_c(cb) == a
        e = add_mixed(e, gen_aff)


def test_scalar_mul_matches_golden(golden_dir):
    """`multiply` (255-step double-and-add) against the golden k*G records for a few k."""
    unc = open(os.path.join(golden_dir, "g1_uncompressed_valid_test_vectors.dat"), "rb").read()
    for k in (0, 1, 2, 3, 77, 999):
        assert o.g1_to_uncompressed(o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, k))) == unc[96 * k:96 * k + 96]
    unc2 = open(os.path.join(golden_dir, "g2_uncompressed_valid_test_vectors.dat"), "rb").read()
    for k in (0, 1, 5, 998):
        assert o.g2_to_uncompressed(o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, k))) == unc2[192 * k:192 * k + 192]


def test_pairing_relic_kat(kats):
    """src/tests/mod.rs:78-231: e(G1, G2) equals the RELIC constant, and pairing == multi_miller_loop + final exp."""
    expect = o.fp12_unflatten([F(v) for v in kats["consts"]["pairings.GT_GENERATOR"]])
    got = o.pairing(o.G1_GEN, o.G2_GEN)
    assert got == expect
    ml = o.multi_miller_loop([(o.G1_GEN, o.g2_prepare(o.G2_GEN))])
    assert ml == o.miller_loop(o.G1_GEN, o.G2_GEN)
    assert o.final_exponentiation(ml) == expect


def test_pairing_properties():
    """pairings.rs:835-970: bilinearity, unitarity, multi_miller_loop with identities."""
    a, b = 0x1234567, 0x7654321
    g, h = o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, a)), o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, b))
    p = o.pairing(g, h)
    gt = o.pairing(o.G1_GEN, o.G2_GEN)
    assert p == o.gt_mul_scalar(gt, (a * b) % o.R_ORDER)
    assert p == o.pairing(o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, a * b)), o.G2_GEN)
    assert o.pairing(o.G1_IDENTITY_AFF, h) == o.FP12_ONE and o.pairing(g, o.G2_IDENTITY_AFF) == o.FP12_ONE
    # unitary: e(P, -Q) * e(P, Q) = 1, and conjugate is the inverse
    nh = (h[0], o.fp2_neg(h[1]), False)
    assert o.fp12_mul(o.pairing(g, nh), p) == o.FP12_ONE and o.fp12_conj(p) == o.pairing(g, nh)
    # multi miller loop of 3 terms incl. identities == product of pairings
    terms = [(g, h), (o.G1_IDENTITY_AFF, h), (o.G1_GEN, o.G2_GEN), (g, o.G2_IDENTITY_AFF)]
    ml = o.multi_miller_loop([(x, o.g2_prepare(y)) for x, y in terms])
    assert o.final_exponentiation(ml) == o.fp12_mul(p, gt)
    assert o.multi_miller_loop([]) == o.FP12_ONE


# ---- tier-1 C oracle (oracle/bls_oracle.c) against tier 0 and the golden vectors -----------------------------
def _fpw(x):
    import numpy as np
    return np.array(o.fp_to_mont_limbs(x), dtype=np.uint64)


def test_c_oracle_field(kats):
    import numpy as np
    from oracle import c_oracle as c
    r = o.SplitMix64(1)
    def rnd():
        v = 0
        for i in range(6):
            v |= r.next() << (64 * i)
        return v % o.P
    a = [0, 1, o.P - 1] + [rnd() for _ in range(300)]
    b = [o.P - 1, 0, o.P - 1] + [rnd() for _ in range(300)]
    A, B = np.stack([_fpw(x) for x in a]), np.stack([_fpw(x) for x in b])
    for op, fn in [(0, o.fp_mul), (1, o.fp_add), (2, o.fp_sub)]:
        assert np.array_equal(c.fp_op(op, A, B), np.stack([_fpw(fn(x, y)) for x, y in zip(a, b)]))
    assert np.array_equal(c.fp_op(3, A), np.stack([_fpw(o.fp_sqr(x)) for x in a]))
    assert np.array_equal(c.fp_op(5, A), np.stack([_fpw(o.fp_neg(x)) for x in a]))
    assert np.array_equal(c.fp_op(4, A[:20]), np.stack([_fpw(o.fp_inv(x) or 0) for x in a[:20]]))
    t = kats["tests"]["fp.test_multiplication"]["fp"]
    assert np.array_equal(c.fp_op(0, np.array([t[0]], dtype=np.uint64), np.array([t[1]], dtype=np.uint64))[0], np.array(t[2], dtype=np.uint64))


def test_c_oracle_g1(golden_dir):
    import numpy as np
    from oracle import c_oracle as c
    gen = np.concatenate([_fpw(o.G1_GEN[0]), _fpw(o.G1_GEN[1])])
    unc = open(os.path.join(golden_dir, "g1_uncompressed_valid_test_vectors.dat"), "rb").read()
    for k in (0, 1, 2, 3, 500, 999, o.R_ORDER - 1, 0x1234567890ABCDEF1234567890ABCDEF):
        s = np.frombuffer((k % o.R_ORDER).to_bytes(32, "little"), dtype=np.uint8)
        proj = c.g1_affine_mul(gen, False, s)
        # exact projective triple == tier 0 (same formulas, same order)
        want = o.g1_affine_mul(o.G1_GEN, k)
        assert [o.fp_from_mont_limbs(proj[6 * i:6 * i + 6]) for i in range(3)] == list(want)
        xy, inf = c.g1_to_affine(proj)
        aff = (o.fp_from_mont_limbs(xy[:6]), o.fp_from_mont_limbs(xy[6:]), inf)
        if k < 1000:
            assert o.g1_to_uncompressed(aff) == unc[96 * k:96 * k + 96]
    # reference-definition MSM, 1 thread and all threads, vs tier 0
    r = o.SplitMix64(9)
    ks = [r.scalar() for _ in range(24)]
    ss = [r.scalar() for _ in range(24)]
    pts = [o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, k)) for k in ks]
    xy = np.stack([np.concatenate([_fpw(p[0]), _fpw(p[1])]) for p in pts])
    sb = np.stack([np.frombuffer(s.to_bytes(32, "little"), dtype=np.uint8) for s in ss])
    want = o.g1_to_affine(o.g1_msm(pts, ss))
    for th in (1, 0):
        out, used = c.g1_msm(xy, None, sb, th)
        got = c.g1_to_affine(out)
        assert (o.fp_from_mont_limbs(got[0][:6]), o.fp_from_mont_limbs(got[0][6:]), got[1]) == want
        assert used >= 1


def test_validation_kats(kats, golden_dir):
    """src/g1.rs:1598-1640 / src/g2.rs:1862-1906 (test_is_torsion_free): a curve point outside the subgroup is
    rejected, generators and golden multiples are accepted; BETA and the psi coefficients match the literals."""
    c = kats["consts"]
    assert F(c["g1.BETA"]) == o.BETA
    v = kats["tests"]["g1.test_is_torsion_free"]["fp"]
    a = (F(v[0]), F(v[1]), False)
    assert o.g1_is_on_curve(a) and not o.g1_is_torsion_free(a)
    assert o.g1_is_torsion_free(o.G1_GEN) and o.g1_is_torsion_free(o.G1_IDENTITY_AFF)
    v = kats["tests"]["g2.test_is_torsion_free"]["fp"]
    a2 = ((F(v[0]), F(v[1])), (F(v[2]), F(v[3])), False)
    assert o.g2_is_on_curve(a2) and not o.g2_is_torsion_free(a2)
    assert o.g2_is_torsion_free(o.G2_GEN) and o.g2_is_torsion_free(o.G2_IDENTITY_AFF)
    psi = kats["tests"]["g2.test_psi"]["fp"]      # first literals of test_psi are not the coefficients; check by property
    g = o.g2_from_affine(o.G2_GEN)
    # psi is a homomorphism and psi^2(P) - [t]psi(P) + [p]P = 0 is implied by psi(P) == [x]P on the subgroup
    assert o.g2_eq(o.g2_psi(o.g2_double(g)), o.g2_double(o.g2_psi(g)))
    cmp_ = open(os.path.join(golden_dir, "g1_compressed_valid_test_vectors.dat"), "rb").read()
    for k in (0, 1, 7, 999):
        p = o.g1_from_compressed(cmp_[48 * k:48 * k + 48])
        assert p is not None and o.g1_to_compressed(p) == cmp_[48 * k:48 * k + 48]
    cmp2 = open(os.path.join(golden_dir, "g2_compressed_valid_test_vectors.dat"), "rb").read()
    for k in (0, 1, 998):
        p = o.g2_from_compressed(cmp2[96 * k:96 * k + 96])
        assert p is not None and o.g2_to_compressed(p) == cmp2[96 * k:96 * k + 96]
    # the off-subgroup point survives the unchecked decoders and is rejected by the checked ones
    enc = o.g1_to_compressed(a)
    assert o.g1_from_compressed_unchecked(enc) == a and o.g1_from_compressed(enc) is None
    assert o.g1_from_uncompressed(o.g1_to_uncompressed(a)) is None
    enc2 = o.g2_to_compressed(a2)
    assert o.g2_from_compressed_unchecked(enc2) == a2 and o.g2_from_compressed(enc2) is None
    # not on the curve
    bad = bytearray(o.g1_to_uncompressed(o.G1_GEN)); bad[95] ^= 1
    assert o.g1_from_uncompressed_unchecked(bytes(bad)) is not None and o.g1_from_uncompressed(bytes(bad)) is None


def test_scalar_field_kats(kats):
    """Fr restatement against the reference's constants and stored answers (src/scalar.rs:76-222, tests :786-1056)."""
    c = kats["consts"]
    S = o.fr_from_mont_limbs
    r = o.R_ORDER
    assert sum(l << (64 * i) for i, l in enumerate(c["scalar.MODULUS"])) == r
    assert (c["scalar.INV"] * c["scalar.MODULUS"][0]) % (1 << 64) == (1 << 64) - 1          # test_inv :818-832
    assert S(c["scalar.R"]) == 1 and o.fr_to_mont_limbs(1) == c["scalar.R"]
    assert S(c["scalar.R2"]) == pow(2, 256, r) and S(c["scalar.R3"]) == pow(2, 512, r)
    assert S(c["scalar.GENERATOR"]) == o.FR_GENERATOR == 7
    assert c["scalar.S"] == o.FR_S and (r - 1) % (1 << o.FR_S) == 0 and ((r - 1) >> o.FR_S) & 1
    root, root_inv, delta, two_inv = S(c["scalar.ROOT_OF_UNITY"]), S(c["scalar.ROOT_OF_UNITY_INV"]), S(c["scalar.DELTA"]), S(c["scalar.TWO_INV"])
    assert root == o.FR_ROOT_OF_UNITY
    # test_constants :786-816
    assert o.fr_mul(2, two_inv) == 1 and o.fr_mul(root, root_inv) == 1
    assert o.fr_pow(root, 1 << o.FR_S) == 1 and o.fr_pow(root, 1 << (o.FR_S - 1)) != 1
    assert o.fr_pow(delta, (r - 1) >> o.FR_S) == 1 and delta == pow(7, 1 << o.FR_S, r)
    # from_bytes_wide :1005-1040
    assert S(c["scalar.FROM_BYTES_WIDE_MAXIMUM"]) == o.fr_from_bytes_wide([0xFF] * 64)
    assert o.fr_from_bytes_wide([254, 255, 255, 255, 1, 0, 0, 0, 2, 72, 3, 0, 250, 183, 132, 88, 245, 79, 188, 236, 239, 79, 140, 153, 111, 5,
                                 197, 172, 89, 177, 36, 24] + [0] * 32) == S(c["scalar.R2"])
    # test_multiplication / test_squaring / test_inversion :1107-1181 walk multiples of LARGEST (= r - 1) and R2
    cur = S(c["scalar.LARGEST"])         # the largest LIMB pattern (raw limbs = r - 1), i.e. the value (r - 1) / R
    assert sum(l << (64 * i) for i, l in enumerate(c["scalar.LARGEST"])) == r - 1
    largest = cur
    for _ in range(20):
        acc = 0
        for bit in bin(cur)[2:]:
            acc = o.fr_add(acc, acc)
            if bit == "1":
                acc = o.fr_add(acc, cur)
        assert o.fr_mul(cur, cur) == acc == o.fr_sqr(cur)
        cur = o.fr_add(cur, largest)
    assert o.fr_inv(0) is None and o.fr_inv(1) == 1 and o.fr_inv(r - 1) == r - 1
    t = S(c["scalar.R2"])
    for _ in range(20):
        assert o.fr_mul(o.fr_inv(t), t) == 1
        t = o.fr_add(t, S(c["scalar.R2"]))


def test_scalar_byte_conversions_limb_level(kats):
    """The limb-level restatement of `Scalar::to_bytes` / `from_bytes` / `from_bytes_wide` (oracle: scalar_limbs_*; reference
    scalar.rs:256-331 with montgomery_reduce :506-550, mul :452-503, add :435-449, sub :420-432) against the reference's own known
    answers (scalar.rs:864-1040) and against the big-integer definitions above: it is what pins the device conversions (row a8)."""
    c = kats["consts"]
    r = o.R_ORDER
    assert o.FR_INV == c["scalar.INV"] and o.FR_MODULUS_LIMBS == c["scalar.MODULUS"]
    assert o.FR_R2_LIMBS == c["scalar.R2"] and o.FR_R3_LIMBS == c["scalar.R3"]
    r2_bytes = bytes([254, 255, 255, 255, 1, 0, 0, 0, 2, 72, 3, 0, 250, 183, 132, 88, 245, 79, 188, 236, 239, 79, 140, 153, 111, 5, 197, 172, 89, 177, 36, 24])
    neg1_bytes = bytes([0, 0, 0, 0, 255, 255, 255, 255, 254, 91, 254, 255, 2, 164, 189, 83, 5, 216, 161, 9, 8, 216, 57, 51, 72, 125, 157, 41, 83, 167, 237, 115])
    neg1 = o.scalar_limbs_sub([0, 0, 0, 0], c["scalar.R"])                       # -&Scalar::one()
    # test_to_bytes :864-896
    assert o.scalar_limbs_to_bytes([0, 0, 0, 0]) == bytes(32)
    assert o.scalar_limbs_to_bytes(c["scalar.R"]) == (1).to_bytes(32, "little")
    assert o.scalar_limbs_to_bytes(c["scalar.R2"]) == r2_bytes
    assert o.scalar_limbs_to_bytes(neg1) == neg1_bytes == (r - 1).to_bytes(32, "little")
    # test_from_bytes :898-966
    assert o.scalar_limbs_from_bytes(bytes(32)) == ([0, 0, 0, 0], True)
    assert o.scalar_limbs_from_bytes((1).to_bytes(32, "little")) == (c["scalar.R"], True)
    assert o.scalar_limbs_from_bytes(r2_bytes) == (c["scalar.R2"], True)
    assert o.scalar_limbs_from_bytes(neg1_bytes) == (neg1, True)
    mod = bytearray(neg1_bytes); mod[0] = 1
    for idx, val in ((0, 1), (0, 2), (22, 58), (31, 116)):
        bad = bytearray(mod); bad[idx] = val
        assert o.scalar_limbs_from_bytes(bytes(bad))[1] is False
    # test_from_u512_* :969-1003 and test_from_bytes_wide_* :1005-1040
    assert o.scalar_limbs_from_bytes_wide(r.to_bytes(32, "little") + bytes(32)) == [0, 0, 0, 0]
    assert o.scalar_limbs_from_bytes_wide((1).to_bytes(64, "little")) == c["scalar.R"]
    assert o.scalar_limbs_from_bytes_wide(bytes(32) + (1).to_bytes(32, "little")) == c["scalar.R2"]
    assert o.scalar_limbs_from_bytes_wide(b"\xff" * 64) == o.scalar_limbs_sub(c["scalar.R3"], c["scalar.R"]) == c["scalar.FROM_BYTES_WIDE_MAXIMUM"]
    assert o.scalar_limbs_from_bytes_wide(r2_bytes + bytes(32)) == c["scalar.R2"]
    assert o.scalar_limbs_from_bytes_wide(neg1_bytes + bytes(32)) == neg1
    # against the big-integer definitions on seeded values
    g = o.SplitMix64(0xA8)
    for _ in range(300):
        v = g.scalar()
        l = o.fr_to_mont_limbs(v)
        assert o.scalar_limbs_to_bytes(l) == v.to_bytes(32, "little") == o.scalar_to_bytes(v)
        assert o.scalar_limbs_from_bytes(v.to_bytes(32, "little")) == (l, True)
        w = (g.scalar() << 256) | (g.next() << 192) | g.scalar()
        assert o.scalar_limbs_from_bytes_wide((w % (1 << 512)).to_bytes(64, "little")) == o.fr_to_mont_limbs(o.fr_from_bytes_wide((w % (1 << 512)).to_bytes(64, "little")))
        a, b_ = o.fr_to_mont_limbs(g.scalar()), o.fr_to_mont_limbs(g.scalar())
        assert o.scalar_limbs_mul(a, b_) == o.fr_to_mont_limbs(o.fr_mul(o.fr_from_mont_limbs(a), o.fr_from_mont_limbs(b_)))
    # the C port of `to_bytes` (bench.py's host-side conversion figure) against the same definitions
    from oracle import c_oracle
    import numpy as np
    vals = [0, 1, r - 1, r - 2] + [g.scalar() for _ in range(500)]
    L = np.array([o.fr_to_mont_limbs(v) for v in vals], dtype=np.uint64)
    assert [bytes(row) for row in c_oracle.scalar_to_bytes_batch(L)] == [v.to_bytes(32, "little") for v in vals]


def test_fr_ntt_definition():
    """the recursive transform equals the defining sums; inverse undoes forward; omega has exact order n"""
    rng = o.SplitMix64(99)
    for log_n in range(0, 7):
        n = 1 << log_n
        x = [rng.scalar() for _ in range(n)]
        y = o.fr_ntt(x)
        assert y == o.fr_ntt_naive(x)
        assert o.fr_ntt(y, inverse=True) == x == o.fr_ntt_naive(y, inverse=True)
        w = o.fr_omega(log_n)
        assert pow(w, n, o.R_ORDER) == 1 and (n == 1 or pow(w, n // 2, o.R_ORDER) == o.R_ORDER - 1)
    # polynomial evaluation view: y_k = p(w^k)
    x = [rng.scalar() for _ in range(16)]
    w = o.fr_omega(4)
    assert o.fr_ntt(x)[5] == sum(c * pow(w, 5 * j, o.R_ORDER) for j, c in enumerate(x)) % o.R_ORDER


def test_hash_to_curve_vectors(kats, golden_dir):
    """oracle/h2c_ref.py against the RFC 9380 (draft-16) vectors of the reference's integration tests
    (tests/hash_to_curve_g1.rs, hash_to_curve_g2.rs, expand_msg.rs) and the SSWU answers of map_g1.rs:655-760."""
    import json
    from oracle import h2c_ref as h
    v = json.load(open(os.path.join(golden_dir, "h2c_vectors.json")))
    assert len(v["expand_msg"]) == 60
    for t in v["expand_msg"]:                      # every expander the reference tests (tests/expand_msg.rs), long DSTs included
        ex = h.XMD_SHA512 if "sha512" in t["test"] else h.XMD_SHA256 if "sha256" in t["test"] else h.XOF_SHAKE128 if "shake128" in t["test"] else h.XOF_SHAKE256
        assert "shake256" in t["test"] or ex != h.XOF_SHAKE256
        assert h.expand_message(ex, bytes.fromhex(t["msg"]), bytes.fromhex(t["dst"]), t["len_in_bytes"]).hex() == t["out"], t["test"]
    for t in v["hash_to_scalar"]:                  # `HashToField for Scalar`, map_scalar.rs:27-45
        assert "%064x" % h.scalar_from_okm(bytes.fromhex(t["okm"])) == t["out"]
    for t in v["g1"]:
        fn = h.g1_hash_to_curve if t["test"].endswith("_ro") else h.g1_encode_to_curve
        p = fn(bytes.fromhex(t["msg"]), bytes.fromhex(t["dst"]))
        a = o.g1_to_affine(p)
        assert o.g1_to_uncompressed(a).hex() == t["out"]
        assert o.g1_is_on_curve(a) and o.g1_is_torsion_free(a)
    for t in v["g2"]:
        fn = h.g2_hash_to_curve if t["test"].endswith("_ro") else h.g2_encode_to_curve
        a = o.g2_to_affine(fn(bytes.fromhex(t["msg"]), bytes.fromhex(t["dst"])))
        assert o.g2_to_uncompressed(a).hex() == t["out"]
        assert o.g2_is_torsion_free(a)
    # exceptional SSWU inputs (u = 0 and u = sqrt(-1/XI)) with the reference's stored projective answer
    F = o.fp_from_mont_limbs
    s = kats["tests"]["h2c_g1.test_simple_swu_expected"]["fp"]
    xo, yo, zo, excp = F(s[0]), F(s[1]), F(s[2]), F(s[3])
    assert h.g1_map_to_curve_simple_swu(0) == (xo, yo, zo)
    assert h.g1_map_to_curve_simple_swu(excp) == (xo, yo, zo)
    # F_2_256 is 2^256, sgn0 is the parity of the canonical value (map_g1.rs:790-806)
    assert h._consts()["F_2_256"] == pow(2, 256, o.P)
    assert h.sgn0_fp(0) == 0 and h.sgn0_fp(1) == 1 and h.sgn0_fp(o.P - 1) == 0


def test_c_oracle_pairing(kats):
    """tier-1 C towers / Miller loop / final exponentiation against the RELIC constant the reference stores
    (pairings.rs:359-475 = src/tests/mod.rs:78-231) and against the tier-0 oracle, raw Miller values included"""
    import numpy as np
    from oracle import c_oracle
    c_oracle.build()
    W = lambda v: np.array(o.fp_to_mont_limbs(v), dtype=np.uint64)
    g1w = lambda a: np.concatenate([W(a[0]), W(a[1])])
    g2w = lambda a: np.concatenate([W(a[0][0]), W(a[0][1]), W(a[1][0]), W(a[1][1])])
    f12w = lambda f: np.concatenate([W(c) for c in o.fp12_flatten(f)])
    gt, _ = c_oracle.pairing_batch(0, g1w(o.G1_GEN)[None, :], None, g2w(o.G2_GEN)[None, :], None, threads=1)
    assert np.array_equal(gt[0], np.array(kats["consts"]["pairings.GT_GENERATOR"], dtype=np.uint64).reshape(-1))
    r = o.SplitMix64(11)
    ps = [o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar())) for _ in range(3)]
    qs = [o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar())) for _ in range(3)]
    G1 = np.stack([g1w(p) for p in ps]); G2 = np.stack([g2w(q) for q in qs])
    ml, _ = c_oracle.pairing_batch(1, G1, None, G2, None)
    full, _ = c_oracle.pairing_batch(0, G1, np.array([0, 0, 1], dtype=np.uint8), G2, None)
    for i in range(3):
        m = o.miller_loop(ps[i], qs[i])
        assert np.array_equal(ml[i], f12w(m))
        want = o.FP12_ONE if i == 2 else o.final_exponentiation(m)          # identity on the G1 side -> Gt::identity
        assert np.array_equal(full[i], f12w(want))
    fe, _ = c_oracle.pairing_batch(2, ml, None, None, None)
    assert np.array_equal(fe[0], full[0])


def test_c_oracle_g2(golden_dir):
    """tier-1 G2 (complete formulas over Fp2, 255-step multiply, Sum) against tier 0 and the golden k*G2 records"""
    import numpy as np
    from oracle import c_oracle
    c_oracle.build(True)
    W = lambda v: np.array(o.fp_to_mont_limbs(v), dtype=np.uint64)
    g2w = lambda a: np.concatenate([W(a[0][0]), W(a[0][1]), W(a[1][0]), W(a[1][1])])
    raw = open(os.path.join(golden_dir, "g2_uncompressed_valid_test_vectors.dat"), "rb").read()
    gen = g2w(o.G2_GEN)
    for k in (1, 2, 7, 999):
        out, _ = c_oracle.g2_msm(gen[None, :], None, np.frombuffer(k.to_bytes(32, "little"), dtype=np.uint8)[None, :], 1)
        xy, inf = c_oracle.g2_to_affine(out)
        a = ((o.fp_from_mont_limbs(xy[0:6]), o.fp_from_mont_limbs(xy[6:12])), (o.fp_from_mont_limbs(xy[12:18]), o.fp_from_mont_limbs(xy[18:24])), inf)
        assert o.g2_to_uncompressed(a) == raw[192 * k:192 * k + 192]
    r = o.SplitMix64(3)
    ks = [r.scalar() for _ in range(2)] + [0]
    ss = [r.scalar() for _ in range(2)] + [5]
    pts = [o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, k)) for k in ks]
    XY = np.stack([g2w(p) for p in pts]); INF = np.array([1 if p[2] else 0 for p in pts], dtype=np.uint8)
    S = np.stack([np.frombuffer(s.to_bytes(32, "little"), dtype=np.uint8) for s in ss])
    out, _ = c_oracle.g2_msm(XY, INF, S, 2)
    xy, inf = c_oracle.g2_to_affine(out)
    tot = sum(k * s for k, s in zip(ks, ss)) % o.R_ORDER
    assert np.array_equal(xy, g2w(o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, tot)))) and not inf


def test_c_oracle_prepared_path_matches_tier0():
    import numpy as np
    """tier 1's `G2Prepared::from` (68 coefficient triples) and its shared-accumulator multi_miller_loop over prepared terms
    (pairings.rs:504-546, 554-603) against tier 0, incl. a term prepared on the fly, an identity, an empty segment and the final exponentiation"""
    from oracle import c_oracle
    c_oracle.build()
    r = o.SplitMix64(5)
    fpw = lambda x: np.array(o.fp_to_mont_limbs(x), dtype=np.uint64)
    f12 = lambda f: np.concatenate([fpw(c) for c in o.fp12_flatten(f)])
    Q = [o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar())) for _ in range(3)]
    P = [o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar())) for _ in range(4)]
    g2 = np.stack([np.concatenate([fpw(q[0][0]), fpw(q[0][1]), fpw(q[1][0]), fpw(q[1][1])]) for q in Q] + [np.zeros(24, dtype=np.uint64)])
    g1 = np.stack([np.concatenate([fpw(p[0]), fpw(p[1])]) for p in P])
    tabs = c_oracle.g2_prepare(g2[1:3])
    for j in (0, 1):
        want = np.array([[np.concatenate([fpw(c[0]), fpw(c[1])]) for c in tri] for tri in o.g2_prepare(Q[1 + j])[1]], dtype=np.uint64)
        assert np.array_equal(tabs[j], want)
    qi = np.array([0xffffffff, 0, 1, 1], dtype=np.uint32)
    f1 = np.array([0, 0, 0, 1], dtype=np.uint8)                     # the last term's P is the identity: skipped
    off = np.array([0, 4, 4], dtype=np.uint64)
    w = o.multi_miller_loop([(p, o.g2_prepare(q)) for p, q in zip(P[:3], Q)])
    out, _ = c_oracle.multi_miller_prepared_many(g1, f1, g2, None, qi, tabs, None, off, False, 1)
    assert np.array_equal(out[0], f12(w)) and np.array_equal(out[1], f12(o.FP12_ONE))
    out, _ = c_oracle.multi_miller_prepared_many(g1, f1, g2, None, qi, tabs, None, off, True, 2)
    assert np.array_equal(out[0], f12(o.final_exponentiation(w)))
