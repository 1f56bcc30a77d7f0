"""ctypes loader for the tier-1 C oracle (oracle/bls_oracle.c).  TEST INFRASTRUCTURE / CPU BASELINE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "bls_oracle.c")
OUT_DIR = os.path.join(_HERE, "_build")
LIB = os.path.join(OUT_DIR, "libblsoracle.so")
_lib = None


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        # -march=x86-64-v3 (AVX2/BMI2/ADX-capable baseline) instead of native: the .so is built in the build
        # container and travels to the GPU box, whose host CPU differs.
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.ora_g1_msm.restype = ctypes.c_int
        _lib.ora_g1_to_affine.restype = ctypes.c_int
        _lib.ora_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fp_op(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 6)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 6)
    out = np.zeros_like(a)
    load().ora_fp_op(ctypes.c_int(op), _p(a), _p(b), _p(out), ctypes.c_long(a.shape[0]))
    return out


def g1_affine_mul(xy, infinity, scalar_bytes):
    xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(12)
    s = np.ascontiguousarray(scalar_bytes, dtype=np.uint8).reshape(32)
    out = np.zeros(18, dtype=np.uint64)
    load().ora_g1_affine_mul(_p(xy), ctypes.c_int(int(infinity)), _p(s), _p(out))
    return out


def g1_to_affine(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(18)
    xy = np.zeros(12, dtype=np.uint64)
    inf = load().ora_g1_to_affine(_p(xyz), _p(xy))
    return xy, bool(inf)


def g1_msm(xy, inf, scalars, threads=0):
    """reference-definition MSM; returns (projective limbs, threads used)"""
    xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 12)
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    assert xy.shape[0] == s.shape[0]
    f = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(18, dtype=np.uint64)
    used = load().ora_g1_msm(_p(xy), None if f is None else _p(f), _p(s), ctypes.c_long(xy.shape[0]), ctypes.c_int(threads), _p(out))
    return out, used


def max_threads():
    return load().ora_max_threads()
