"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's hash-to-curve path.

Follows /root/reference/src/hash_to_curve/: expand_msg.rs (ExpandMsgXmd :230-328, DST reduction :74-95),
mod.rs (hash_to_field :32-49, hash_to_curve / encode_to_curve :86-108), map_g1.rs (from_okm :513-531, sgn0
:535-543, map_to_curve_simple_swu :550-586, iso_map :589-630), map_g2.rs (from_okm :374-378, sgn0 :382-388,
map_to_curve_simple_swu :391-454, iso_map :457-492), g1.rs clear_cofactor :800-802, g2.rs psi2 :890-912 and
clear_cofactor :938-947.  Expanders: XMD over SHA-256 / SHA-512 and XOF over SHAKE128 / SHAKE256 (expand_msg.rs:167-328; hashlib
provides the digests), and `HashToField for Scalar` (map_scalar.rs:10-25).

Pinned by tests/test_oracle_golden.py against the RFC 9380 (draft-16) vectors the reference's integration tests
hold (tests/golden/h2c_vectors.json) and its SSWU exceptional-case answers.  The isogeny / SSWU constants are the
reference's literals, read from tests/golden/ref_kats.json (numbers only, extracted by tests/golden/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import hashlib
import json
import os

from . import bls12_381_ref as o

P = o.P
_K = None


def _consts():
    global _K
    if _K is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_kats.json")
        c = json.load(open(path))["consts"]
        F = o.fp_from_mont_limbs
        k = {}
        for name in ("ISO11_XNUM", "ISO11_XDEN", "ISO11_YNUM", "ISO11_YDEN"):
            k[name] = [F(v) for v in c["h2c_g1." + name]]
        for name in ("SSWU_ELLP_A", "SSWU_ELLP_B", "SSWU_XI", "SQRT_M_XI_CUBED"):
            k["G1_" + name] = F(c["h2c_g1." + name][0])
        k["F_2_256"] = F(c["h2c_g1.F_2_256"])
        pair = lambda v: [(F(v[2 * i]), F(v[2 * i + 1])) for i in range(len(v) // 2)]
        for name in ("ISO3_XNUM", "ISO3_XDEN", "ISO3_YNUM", "ISO3_YDEN", "SSWU_ETAS"):
            k[name] = pair(c["h2c_g2." + name])
        for name in ("SSWU_ELLP_A", "SSWU_ELLP_B", "SSWU_XI", "SSWU_RV1"):
            k["G2_" + name] = pair(c["h2c_g2." + name])[0]
        _K = k
    return _K


# ---- message expansion (expand_msg.rs): the four expanders the reference's tests exercise ---------------------------------
# expander ids as in include/bls12_381_hip.h: 0 XMD:SHA-256, 1 XMD:SHA-512, 2 XOF:SHAKE128, 3 XOF:SHAKE256
XMD_SHA256, XMD_SHA512, XOF_SHAKE128, XOF_SHAKE256 = 0, 1, 2, 3
_XMD = {XMD_SHA256: (hashlib.sha256, 32, 64), XMD_SHA512: (hashlib.sha512, 64, 128)}          # digest, output bytes, block bytes
_XOF = {XOF_SHAKE128: hashlib.shake_128, XOF_SHAKE256: hashlib.shake_256}
XOF_DST_LEN = 32            # `L` = ceil(2 k / 8) with k = 128: `XofOutputLength = U32` of Fp / Fp2 / Scalar (map_g1.rs, map_g2.rs, map_scalar.rs:15-16)


def expand_message(expander, msg, dst, len_in_bytes):
    """ExpandMessage::init_expand + reading everything (expand_msg.rs): XMD :230-328 (generic in the digest), XOF :167-228;
    DST reduction :47-95."""
    msg, dst = bytes(msg), bytes(dst)
    assert len_in_bytes <= 0xFFFF
    if expander in _XMD:
        H, hs, bs = _XMD[expander]
        if len(dst) > 255:                                 # expand_msg.rs:74-95
            dst = H(b"H2C-OVERSIZE-DST-" + dst).digest()
        ell = (len_in_bytes + hs - 1) // hs
        assert ell <= 255
        dst_prime = dst + bytes([len(dst)])
        b0 = H(bytes(bs) + msg + len_in_bytes.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
        bi = H(b0 + b"\x01" + dst_prime).digest()
        out = bi
        for i in range(2, ell + 1):
            bi = H(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
            out += bi
        return out[:len_in_bytes]
    X = _XOF[expander]
    if len(dst) > 255:                                     # expand_msg.rs:47-70
        dst = X(b"H2C-OVERSIZE-DST-" + dst).digest(XOF_DST_LEN)
    return X(msg + len_in_bytes.to_bytes(2, "big") + dst + bytes([len(dst)])).digest(len_in_bytes)


def expand_message_xmd(msg, dst, len_in_bytes):
    return expand_message(XMD_SHA256, msg, dst, len_in_bytes)


def scalar_from_okm(okm):
    """`HashToField for Scalar` (map_scalar.rs:10-25): 48 big-endian bytes, zero-extended to 64 and reversed -> from_bytes_wide."""
    assert len(okm) == 48
    return o.fr_from_bytes_wide((bytes(16) + bytes(okm))[::-1])


def hash_to_field_scalar(msg, dst, count, expander=XMD_SHA256):
    """mod.rs:32-49 with T = Scalar (InputLength = 48)"""
    u = expand_message(expander, msg, dst, 48 * count)
    return [scalar_from_okm(u[48 * i:48 * i + 48]) for i in range(count)]


def fp_from_okm(okm):
    """map_g1.rs:513-531: db * 2^256 + da with db, da the two big-endian 32-byte halves."""
    assert len(okm) == 64
    return (int.from_bytes(okm[:32], "big") * _consts()["F_2_256"] + int.from_bytes(okm[32:], "big")) % P


def hash_to_field_fp(msg, dst, count, expander=XMD_SHA256):
    u = expand_message(expander, msg, dst, 64 * count)
    return [fp_from_okm(u[64 * i:64 * i + 64]) for i in range(count)]


def hash_to_field_fp2(msg, dst, count, expander=XMD_SHA256):
    u = expand_message(expander, msg, dst, 128 * count)
    return [(fp_from_okm(u[128 * i:128 * i + 64]), fp_from_okm(u[128 * i + 64:128 * i + 128])) for i in range(count)]


def sgn0_fp(a): return a & 1                                         # map_g1.rs:535-543
def sgn0_fp2(a): return (a[0] & 1) | ((a[0] == 0) & (a[1] & 1))     # map_g2.rs:382-388


# ---- G1 (map_g1.rs) ------------------------------------------------------------------------------------------
def g1_map_to_curve_simple_swu(u):
    """map_g1.rs:550-586 -> projective point on the 11-isogenous curve E'."""
    k = _consts()
    A, B, XI = k["G1_SSWU_ELLP_A"], k["G1_SSWU_ELLP_B"], k["G1_SSWU_XI"]
    usq = u * u % P
    xi_usq = XI * usq % P
    xisq_u4 = xi_usq * xi_usq % P
    nd_common = (xisq_u4 + xi_usq) % P
    x_den = A * (XI if nd_common == 0 else (-nd_common) % P) % P
    x0_num = B * (1 + nd_common) % P
    x_densq = x_den * x_den % P
    gx_den = x_densq * x_den % P
    gx0_num = ((x0_num * x0_num + A * x_densq) * x0_num + B * gx_den) % P
    u_v = gx0_num * gx_den % P
    vsq = gx_den * gx_den % P
    sqrt_candidate = u_v * pow(u_v * vsq % P, (P - 3) // 4, P) % P           # chain_pm3div4
    gx0_square = sqrt_candidate * sqrt_candidate * gx_den % P == gx0_num
    x1_num = x0_num * xi_usq % P
    y1 = k["G1_SQRT_M_XI_CUBED"] * usq % P * u % P * sqrt_candidate % P
    x_num = x0_num if gx0_square else x1_num
    y = sqrt_candidate if gx0_square else y1
    if sgn0_fp(y) ^ sgn0_fp(u):
        y = (-y) % P
    return (x_num, y * x_den % P, x_den)


def _iso_map(F, coeffs, pt, nz):
    """map_g1.rs:589-630 / map_g2.rs:457-492 (Horner in x with powers of z)."""
    x, y, z = pt
    zpows = [z]
    for _ in range(nz - 1):
        zpows.append(F.mul(zpows[-1], z))
    mapvals = []
    for coeff in coeffs:
        clast = len(coeff) - 1
        v = coeff[clast]
        for j in range(clast):
            v = F.add(F.mul(v, x), F.mul(zpows[j], coeff[clast - 1 - j]))
        mapvals.append(v)
    mapvals[1] = F.mul(mapvals[1], z)
    mapvals[2] = F.mul(mapvals[2], y)
    mapvals[3] = F.mul(mapvals[3], z)
    return (F.mul(mapvals[0], mapvals[3]), F.mul(mapvals[2], mapvals[1]), F.mul(mapvals[1], mapvals[3]))


def g1_iso_map(pt):
    k = _consts()
    return _iso_map(o._FpOps, [k["ISO11_XNUM"], k["ISO11_XDEN"], k["ISO11_YNUM"], k["ISO11_YDEN"]], pt, 15)


def g1_map_to_curve(u): return g1_iso_map(g1_map_to_curve_simple_swu(u))          # map_g1.rs:635-638


def g1_clear_cofactor(p):
    """g1.rs:800-802: self - self.mul_by_x()"""
    return o.g1_add(p, o.g1_neg(o.g1_mul_by_x(p)))


def g1_hash_to_curve(msg, dst, expander=XMD_SHA256):
    u = hash_to_field_fp(msg, dst, 2, expander)
    return g1_clear_cofactor(o.g1_add(g1_map_to_curve(u[0]), g1_map_to_curve(u[1])))     # mod.rs:86-92


def g1_encode_to_curve(msg, dst, expander=XMD_SHA256):
    return g1_clear_cofactor(g1_map_to_curve(hash_to_field_fp(msg, dst, 1, expander)[0]))           # mod.rs:103-108


# ---- G2 (map_g2.rs) --------------------------------------------------------------------------------------------
def g2_map_to_curve_simple_swu(u):
    """map_g2.rs:391-454 -> projective point on the 3-isogenous curve E2'."""
    k = _consts()
    mul, sqr, add, neg = o.fp2_mul, o.fp2_sqr, o.fp2_add, o.fp2_neg
    A, B, XI = k["G2_SSWU_ELLP_A"], k["G2_SSWU_ELLP_B"], k["G2_SSWU_XI"]
    usq = sqr(u)
    xi_usq = mul(XI, usq)
    xisq_u4 = sqr(xi_usq)
    nd_common = add(xisq_u4, xi_usq)
    x_den = mul(A, XI if nd_common == (0, 0) else neg(nd_common))
    x0_num = mul(B, add((1, 0), nd_common))
    x_densq = sqr(x_den)
    gx_den = mul(x_densq, x_den)
    gx0_num = add(mul(add(sqr(x0_num), mul(A, x_densq)), x0_num), mul(B, gx_den))
    vsq = sqr(gx_den)
    v_3 = mul(vsq, gx_den)
    v_4 = sqr(vsq)
    uv_7 = mul(mul(gx0_num, v_3), v_4)
    uv_15 = mul(uv_7, sqr(v_4))
    sqrt_candidate = mul(uv_7, o.fp2_pow(uv_15, (P * P - 9) // 16))              # chain_p2m9div16
    ok = lambda t: mul(sqr(t), gx_den) == gx0_num
    y = sqrt_candidate
    tmp = ((-sqrt_candidate[1]) % P, sqrt_candidate[0])
    if ok(tmp): y = tmp
    tmp = mul(sqrt_candidate, k["G2_SSWU_RV1"])
    if ok(tmp): y = tmp
    tmp = (tmp[1], (-tmp[0]) % P)
    if ok(tmp): y = tmp
    gx1_num = mul(mul(gx0_num, xi_usq), xisq_u4)
    sc = mul(mul(sqrt_candidate, usq), u)
    eta_found = False
    for eta in k["SSWU_ETAS"]:
        tmp = mul(sc, eta)
        if mul(sqr(tmp), gx_den) == gx1_num:
            y = tmp
            eta_found = True
    x_num = mul(x0_num, xi_usq) if eta_found else x0_num
    if sgn0_fp2(u) ^ sgn0_fp2(y):
        y = neg(y)
    return (x_num, mul(y, x_den), x_den)


def g2_iso_map(pt):
    k = _consts()
    return _iso_map(o._Fp2Ops, [k["ISO3_XNUM"], k["ISO3_XDEN"], k["ISO3_YNUM"], k["ISO3_YDEN"]], pt, 3)


def g2_map_to_curve(u): return g2_iso_map(g2_map_to_curve_simple_swu(u))


PSI2_COEFF_X = pow(pow(2, (P - 1) // 3, P), -1, P)                    # g2.rs:891-903: 1 / 2^((p-1)/3)


def g2_psi2(p):
    """g2.rs:890-912."""
    return (o.fp2_mul(p[0], (PSI2_COEFF_X, 0)), o.fp2_neg(p[1]), p[2])


def g2_clear_cofactor(p):
    """g2.rs:938-947."""
    t1 = o.g2_mul_by_x(p)
    t2 = o.g2_psi(p)
    r = g2_psi2(o.g2_double(p))
    r = o.g2_add(r, o.g2_mul_by_x(o.g2_add(t1, t2)))
    r = o.g2_add(r, o.g2_neg(t1))
    r = o.g2_add(r, o.g2_neg(t2))
    return o.g2_add(r, o.g2_neg(p))


def g2_hash_to_curve(msg, dst, expander=XMD_SHA256):
    u = hash_to_field_fp2(msg, dst, 2, expander)
    return g2_clear_cofactor(o.g2_add(g2_map_to_curve(u[0]), g2_map_to_curve(u[1])))


def g2_encode_to_curve(msg, dst, expander=XMD_SHA256):
    return g2_clear_cofactor(g2_map_to_curve(hash_to_field_fp2(msg, dst, 1, expander)[0]))
