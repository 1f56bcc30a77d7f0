"""The lane-cooperative pairing kernels, compiled for the HOST and run one thread per lane (tests/simt), against the oracle.

The GPU box is where these kernels are timed and tested at size (tests/test_gpu_parity.py); this file lets the CPU suite run
the very same device functions -- pair-lane and quad-lane towers, Miller loop, final exponentiation -- bit for bit against
the oracle, so that a change in the tower code is checked before it ever reaches a GPU.  The emulation library is test
infrastructure: it is built into build/ (git-ignored) from tests/simt/emu_pairing.cpp and is never loaded by the product."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import bls12_381_ref as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
LIB = os.path.join(ROOT, "build", "libemu_test.so")


def fpw(x):
    return np.array(o.fp_to_mont_limbs(x), dtype=np.uint64)


def fp12w(f):
    return np.concatenate([fpw(c) for c in o.fp12_flatten(f)])


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ in this image")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "tests", "simt", "emu_pairing.cpp")
    csrc = os.path.join(ROOT, "bls12_381_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tests", "simt", "hip", "hip_runtime.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([CLANG, "-O1", "-std=c++17", "-pthread", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-psabi", "-DEMU_WITH_QUAD",
                               "-I" + os.path.join(ROOT, "tests", "simt"), "-I" + csrc, src, "-o", LIB])
    return ctypes.CDLL(LIB)


def _pairs(seed, n):
    r = o.SplitMix64(seed)
    P = [o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar())) for _ in range(n)]
    Q = [o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar())) for _ in range(n)]
    g1 = np.concatenate([np.concatenate([fpw(p[0]), fpw(p[1])]) for p in P])
    g2 = np.concatenate([np.concatenate([fpw(q[0][0]), fpw(q[0][1]), fpw(q[1][0]), fpw(q[1][1])]) for q in Q])
    return P, Q, g1, g2


def test_emulated_pair_lane_pairing_matches_oracle(emu):
    """validates the emulation itself on the round-1/2 kernels: k_pairing on one quad = two lane pairs"""
    P, Q, g1, g2 = _pairs(11, 2)
    out = np.zeros(2 * 72, dtype=np.uint64)
    emu.emu_pair_pairing(1, vp(g1), vp(g2), vp(out), ctypes.c_size_t(2))
    for i in range(2):
        assert np.array_equal(out[72 * i:72 * i + 72], fp12w(o.miller_loop(P[i], Q[i])))
    emu.emu_pair_pairing(0, vp(g1), vp(g2), vp(out), ctypes.c_size_t(2))
    for i in range(2):
        assert np.array_equal(out[72 * i:72 * i + 72], fp12w(o.pairing(P[i], Q[i])))


def test_emulated_quad_miller_loop_and_pairing_match_oracle(emu):
    """quad.hip.h: raw Miller value and full pairing of one pair spread over four lanes == the reference's
    (pairings.rs:668-770, :48-176, :607-653)"""
    for seed in (12, 13):
        P, Q, g1, g2 = _pairs(seed, 1)
        out = np.zeros(72, dtype=np.uint64)
        ml = o.miller_loop(P[0], Q[0])
        emu.emu_quad_pairing(1, vp(g1), vp(g2), vp(out), ctypes.c_size_t(1))
        assert np.array_equal(out, fp12w(ml))
        emu.emu_quad_pairing(0, vp(g1), vp(g2), vp(out), ctypes.c_size_t(1))
        assert np.array_equal(out, fp12w(o.final_exponentiation(ml)))


def test_emulated_quad_tower_ops_match_oracle(emu):
    """the quad forms of Fp12 multiplication, inversion, Frobenius, conjugation, cyclotomic squaring, the compressed-squaring
    exponentiation by |x| (with its degenerate input 1) and the final exponentiation, on random field elements"""
    r = o.SplitMix64(77)

    def rfp12():
        vals = []
        for _ in range(12):
            v = 0
            for i in range(6):
                v |= r.next() << (64 * i)
            vals.append(v % o.P)
        return o.fp12_unflatten(vals)

    def run(op, a, b=None):
        out = np.zeros(72, dtype=np.uint64)
        aw = fp12w(a)
        bw = fp12w(b) if b is not None else None
        emu.emu_quad_fp12_op(op, vp(aw), vp(bw) if bw is not None else None, vp(out), ctypes.c_size_t(1))
        return out

    x, y = rfp12(), rfp12()
    assert np.array_equal(run(0, x, y), fp12w(o.fp12_mul(x, y)))
    assert np.array_equal(run(4, x), fp12w(o.fp12_inv(x)))
    assert np.array_equal(run(7, x), fp12w(o.fp12_frobenius(x)))
    assert np.array_equal(run(8, x), fp12w(o.fp12_conj(x)))
    t0 = x
    for _ in range(6):
        t0 = o.fp12_frobenius(t0)
    t2 = o.fp12_mul(t0, o.fp12_inv(x))
    t2 = o.fp12_mul(o.fp12_frobenius(o.fp12_frobenius(t2)), t2)          # in the cyclotomic subgroup
    assert np.array_equal(run(9, t2), fp12w(o.cyclotomic_square(t2)))
    assert np.array_equal(run(10, t2), fp12w(o.cyclotomic_exp(t2)))
    assert np.array_equal(run(10, o.FP12_ONE), fp12w(o.cyclotomic_exp(o.FP12_ONE)))
    out = np.zeros(72, dtype=np.uint64)
    xw = fp12w(x)
    emu.emu_quad_final_exp(vp(xw), vp(out), ctypes.c_size_t(1))
    assert np.array_equal(out, fp12w(o.final_exponentiation(x)))


def test_emulated_wide_kernel_matches_oracle():
    """wide.hip.h (one pairing per 1024-lane workgroup) interpreting the generated programs, 1024 host threads as lanes: the raw
    Miller value, the pairing and the final exponentiation alone, bit for bit against the oracle"""
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ in this image")
    lib_path = os.path.join(ROOT, "build", "libemu_wide_test.so")
    prog_path = os.path.join(ROOT, "build", "wide_prog_emu.bin")
    src = os.path.join(ROOT, "tests", "simt", "emu_pairing.cpp")
    csrc = os.path.join(ROOT, "bls12_381_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tests", "simt", "hip", "hip_runtime.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < max(os.path.getmtime(d) for d in deps):          # rebuilt only when a source changed
        subprocess.check_call([CLANG, "-O1", "-std=c++17", "-pthread", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-psabi", "-DEMU_WITH_WIDE", "-DEMU_LANES=1024",
                               "-I" + os.path.join(ROOT, "tests", "simt"), "-I" + csrc, src, "-o", lib_path])
    gen = os.path.join(ROOT, "tools", "gen_wide_prog.py")
    gdeps = [gen]
    if not os.path.exists(prog_path) or os.path.getmtime(prog_path) < max(os.path.getmtime(d) for d in gdeps):
        subprocess.check_call([os.sys.executable, gen, "--lanes", "1024", "--chunk", "4", "--out", prog_path])
    lib = ctypes.CDLL(lib_path)
    blob = np.frombuffer(open(prog_path, "rb").read(), dtype=np.uint32).copy()
    pm = ctypes.c_void_p(blob.ctypes.data + 4 * int(blob[2])); pf = ctypes.c_void_p(blob.ctypes.data + 4 * int(blob[4]))
    P, Q, g1, g2 = _pairs(31, 1)
    out = np.zeros(72, dtype=np.uint64)
    ml = o.miller_loop(P[0], Q[0])
    lib.emu_wide(1, vp(g1), vp(g2), vp(out), pm, pf)
    assert np.array_equal(out, fp12w(ml))
    lib.emu_wide(0, vp(g1), vp(g2), vp(out), pm, pf)
    assert np.array_equal(out, fp12w(o.final_exponentiation(ml)))
    mlw = fp12w(ml)
    lib.emu_wide(2, vp(mlw), None, vp(out), pm, pf)
    assert np.array_equal(out, fp12w(o.final_exponentiation(ml)))


def _g2w(q):
    return np.concatenate([fpw(q[0][0]), fpw(q[0][1]), fpw(q[1][0]), fpw(q[1][1])])


PREP_POINT_WORDS = 68 * 3 * 2 * 16


def test_emulated_g2_prepared_table_and_shared_prepared_loop_match_oracle(emu):
    """prep.hip.h: (i) the device table of a point holds the reference's 68 coefficient triples (`G2Prepared::from`, pairings.rs:504-546)
    limb for limb; (ii) the shared-accumulator loop over a segment of prepared and unprepared terms (incl. a skipped identity term) gives
    the reference's multi_miller_loop value (pairings.rs:554-603); (iii) a segment longer than one pass (kmax = 2) and the uniform-run
    mode of one long product agree with it."""
    r = o.SplitMix64(4242)
    Q = [o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar())) for _ in range(3)]
    P = [o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar())) for _ in range(4)]
    # (i) tables of Q[1], Q[2] (index 0, 1)
    tab = np.zeros(2 * PREP_POINT_WORDS // 2, dtype=np.uint64)          # u32 words, 8-byte aligned by numpy (>= 16 in practice)
    assert tab.ctypes.data % 16 == 0
    tab_inf = np.zeros(2, dtype=np.uint8)
    for j, q in enumerate(Q[1:]):
        qw = _g2w(q)
        emu.emu_g2_prepare(vp(qw), None, ctypes.c_void_p(tab.ctypes.data + 4 * PREP_POINT_WORDS * j), ctypes.c_void_p(tab_inf.ctypes.data + j))
        co = np.zeros(68 * 3 * 12, dtype=np.uint64)
        emu.emu_g2_prepared_export(ctypes.c_void_p(tab.ctypes.data + 4 * PREP_POINT_WORDS * j), vp(co))
        want = np.concatenate([np.concatenate([fpw(c[0]), fpw(c[1])]) for tri in o.g2_prepare(q)[1] for c in tri])
        assert np.array_equal(co, want), "prepared coefficients of point %d" % j
    assert not tab_inf.any()
    # (ii) segment: P0 with Q0 unprepared, P1 with table 0, P2 = identity with table 1 (skipped), P3 with table 1
    terms_p = [P[0], P[1], (0, 1, True), P[3]]
    g1 = np.concatenate([np.concatenate([fpw(p[0]), fpw(p[1])]) for p in terms_p])
    g1inf = np.array([0, 0, 1, 0], dtype=np.uint8)
    g2 = np.concatenate([_g2w(Q[0])] + [np.zeros(24, dtype=np.uint64)] * 3)
    qidx = np.array([0xffffffff, 0, 1, 1], dtype=np.uint32)
    want = o.multi_miller_loop([(P[0], o.g2_prepare(Q[0])), (P[1], o.g2_prepare(Q[1])), (P[3], o.g2_prepare(Q[2]))])

    def run(off, total, kuni, kmax):
        work = np.zeros(kmax * 65 * 4 // 2 + 2, dtype=np.uint64)
        out = np.zeros(72, dtype=np.uint64)
        status = np.zeros(1, dtype=np.uint32)
        offp = vp(off) if off is not None else None
        emu.emu_mml_prep(vp(g1), vp(g1inf), vp(g2), None, vp(qidx), vp(tab), vp(tab_inf), ctypes.c_uint32(2), offp, ctypes.c_size_t(total), kuni, kmax, vp(work), vp(out), vp(status))
        assert status[0] == 0
        return out

    off = np.array([0, 4], dtype=np.uint64)
    assert np.array_equal(run(off, 4, 0, 4), fp12w(want))
    # (iii) two passes of two terms each, multiplied in the kernel; and the uniform-run mode
    assert np.array_equal(run(off, 4, 0, 2), fp12w(want))
    assert np.array_equal(run(None, 4, 4, 4), fp12w(want))
    # an empty segment is one; an all-unprepared single term equals the plain Miller loop
    assert np.array_equal(run(np.array([2, 2], dtype=np.uint64), 4, 0, 4), fp12w(o.FP12_ONE))
    assert np.array_equal(run(np.array([0, 1], dtype=np.uint64), 4, 0, 1), fp12w(o.miller_loop(P[0], Q[0])))
