"""Host-side mirror of the zkcrypto/bls12_381 surface for the accelerated hot path.

Same names, argument meaning and error behaviour as the reference's operator / trait API
(/root/reference/src/lib.rs:49-83): `Scalar`, `G1Affine`, `G1Projective`, `G2Affine`, `G2Projective`, `Gt`,
`MillerLoopResult`, `G2Prepared`, `pairing`, `multi_miller_loop`, `Bls12` -- every group / pairing
operation below is executed by the HIP kernels through the C ABI (include/bls12_381_hip.h).  Values are
carried in the reference's own in-memory format (canonical Montgomery limbs, R = 2^384), as numpy uint64
arrays, so they can be handed to / taken from a Rust caller unchanged.

Only (de)serialisation (big-endian byte encodings, src/notes/serialization.rs) and `Scalar` bookkeeping
run on the host in Python integers; they are format conversion, not part of the compute path.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import BlsGpuError, check

# ---- curve constants (src/fp.rs:70-77, src/scalar.rs:76-81, src/g1.rs:197-217, src/g2.rs:210-250) ----
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_MONT_R = (1 << 384) % P
_MONT_RINV = pow(_MONT_R, -1, P)
_G1X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
_G1Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
_G2X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
_G2Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)


def fp_to_limbs(x):
    """integer in [0,p) -> 6 Montgomery limbs (np.uint64), the reference's `Fp([u64; 6])`"""
    v = (int(x) * _MONT_R) % P
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], dtype=np.uint64)


def limbs_to_fp(l):
    v = 0
    for i, w in enumerate(np.asarray(l, dtype=np.uint64).reshape(-1)[:6]):
        v |= int(w) << (64 * i)
    return (v * _MONT_RINV) % P


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _u64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(shape)


def _flags(f, n):
    if f is None:
        return None
    f = np.ascontiguousarray(f, dtype=np.uint8).reshape(-1)
    if f.shape[0] != n:
        raise ValueError("infinity flag array has the wrong length")
    return f


_R_WORDS = np.frombuffer(R_ORDER.to_bytes(32, "little"), dtype="<u8")
SCALAR_BYTES, SCALAR_MONT = 0, 1          # include/bls12_381_hip.h: BLSGPU_SCALAR_BYTES / BLSGPU_SCALAR_MONT
EXPAND_XMD_SHA256, EXPAND_XMD_SHA512, EXPAND_XOF_SHAKE128, EXPAND_XOF_SHAKE256 = 0, 1, 2, 3      # BLSGPU_EXPAND_*


def scalars_are_canonical(sb):
    """(n,32) uint8 little-endian -> True iff every row is < r (vectorised lexicographic compare, top word first)."""
    w = np.ascontiguousarray(sb).reshape(-1, 32).view("<u8")
    lt = np.zeros(len(w), dtype=bool)
    eq = np.ones(len(w), dtype=bool)
    for k in (3, 2, 1, 0):
        lt |= eq & (w[:, k] < _R_WORDS[k])
        eq &= w[:, k] == _R_WORDS[k]
    return bool(lt.all())


def scalars_to_bytes(scalars):
    """iterable of ints / Scalars / (n,32) uint8 array -> (n,32) uint8 little-endian canonical (Scalar::to_bytes)"""
    if isinstance(scalars, np.ndarray) and scalars.dtype == np.uint8:
        s = np.ascontiguousarray(scalars).reshape(-1, 32)
        if not scalars_are_canonical(s):
            raise ValueError("scalar bytes are not canonical (>= r): Scalar::from_bytes would return None (scalar.rs:256-280)")
        return s
    out = np.zeros((len(scalars), 32), dtype=np.uint8)
    for i, s in enumerate(scalars):
        v = s.value if isinstance(s, Scalar) else int(s) % R_ORDER
        out[i] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    return out


class ResidentBases:
    """Device-resident base points in the library's internal form (`blsgpu_bases`)."""

    def __init__(self, ctx, handle, group):
        self.ctx, self.handle, self.group = ctx, handle, group

    def __len__(self):
        return int(_lib.load().blsgpu_bases_len(self.handle))

    @property
    def subgroup_state(self):
        """1: every base passed the device-side is_torsion_free (or the set was built as [k]G); 2: assumed by the caller
        (Context.set_assume_subgroup); 0: some base is outside the prime-order subgroup -- the MSM then runs on plain windows
        (exact for every curve point, like the reference's `multiply`, g1.rs:754-774)."""
        return int(_lib.load().blsgpu_bases_subgroup_state(self.handle))

    def precompute(self, window_bits=0):
        """Build resident window-shifted tables (see blsgpu_bases_precompute); later MSMs use them automatically."""
        check(_lib.load().blsgpu_bases_precompute(self.ctx.h, self.handle, window_bits), "bases_precompute")
        return self

    def download(self, first=0, count=None):
        n = len(self) - first if count is None else count
        w = 12 if self.group == 1 else 24
        xy = np.zeros((n, w), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        check(_lib.load().blsgpu_bases_download(self.ctx.h, self.handle, first, n, _ptr(xy), _ptr(inf)), "bases_download")
        return xy, inf

    def free(self):
        if self.handle:
            _lib.load().blsgpu_bases_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PreparedG2Table:
    """m `G2Prepared` values resident on the device (`blsgpu_g2_prepared`): the 68 line-coefficient triples of each point
    (pairings.rs:487-546), named by index in the `*_prepared` Miller loops."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def __len__(self):
        return int(self.ctx.lib.blsgpu_g2_prepared_len(self.handle)) if self.handle else 0

    def coeffs(self, index):
        """(infinity, (68, 3, 12) u64): `G2Prepared { infinity, coeffs }` of point `index` in the reference's value format"""
        out = np.zeros((68, 3, 12), dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        check(self.ctx.lib.blsgpu_g2_prepared_coeffs(self.ctx.h, self.handle, index, _ptr(out), _ptr(inf)), "g2_prepared_coeffs")
        return bool(inf[0]), out

    def free(self):
        if self.handle:
            self.ctx.lib.blsgpu_g2_prepared_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


UNPREPARED = 0xffffffff


class Context:
    """One device + one HIP stream + scratch memory (`blsgpu_ctx`).  Not re-entrant."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_create(device, ctypes.byref(h)), "blsgpu_create")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.blsgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ------------------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        check(self.lib.blsgpu_set_stream(self.h, ctypes.c_void_p(stream_ptr) if stream_ptr else None), "set_stream")

    def synchronize(self):
        check(self.lib.blsgpu_synchronize(self.h), "synchronize")

    def set_pipelining(self, on):
        check(self.lib.blsgpu_set_pipelining(self.h, 1 if on else 0), "set_pipelining")

    def join(self, lag=0):
        check(self.lib.blsgpu_join_lag(self.h, lag), "join")

    def set_msm_window(self, c):
        """Pippenger window width in bits: 0 = automatic, else 4..16.  With the endomorphism split active (subgroup base sets)
        the width applies to the 128-bit (G1) / 64-bit (G2) sub-scalars."""
        check(self.lib.blsgpu_set_msm_window(self.h, c), "set_msm_window")

    def set_assume_subgroup(self, on):
        """Skip the per-upload `is_torsion_free` pass over uploaded bases (the caller vouches that every point is in the
        prime-order subgroup, e.g. values that came from the checked decoders).  Default off: every upload is tested and a
        set with an off-subgroup point falls back to plain windows, which match the reference's `multiply` on any curve point."""
        check(self.lib.blsgpu_set_assume_subgroup(self.h, 1 if on else 0), "set_assume_subgroup")

    def pairing_layout(self, n):
        """which kernels a pairing / Miller-loop / final-exponentiation call with n items runs: 256 = wide (one item per
        workgroup, small batches), 4 = quad, 2 = lane pair (include/bls12_381_hip.h `blsgpu_pairing_layout`)"""
        r = int(self.lib.blsgpu_pairing_layout(self.h, n))
        if r < 0:
            check(r, "pairing_layout")
        return r

    def msm_accumulate_stats(self, enable):
        """(average ms, launches) of the accumulation kernel since the last call (HIP events on its stream); then reset and
        switch the measurement off (False / 0) or on (True / 1: every launch; N: every N-th launch)"""
        avg, cnt = ctypes.c_double(), ctypes.c_uint()
        check(self.lib.blsgpu_msm_accumulate_stats(self.h, int(enable), ctypes.byref(avg), ctypes.byref(cnt)), "msm_accumulate_stats")
        return avg.value, cnt.value

    def kernel_timing(self, on):
        """record the HIP-event duration of every kernel this context launches (blsgpu_kernel_timing); read with kernel_timing_report()"""
        check(self.lib.blsgpu_kernel_timing(self.h, 1 if on else 0), "kernel_timing")

    def kernel_timing_report(self):
        """{kernel name: {"launches", "total_ms", "min_ms", "max_ms"}} in first-launch order; clears the records"""
        buf = ctypes.create_string_buffer(1 << 16)
        need = ctypes.c_size_t(0)
        check(self.lib.blsgpu_kernel_timing_report(self.h, buf, len(buf), ctypes.byref(need)), "kernel_timing_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, tot, mn, mx = line.split("\t")
            out[name] = {"launches": int(n), "total_ms": float(tot), "min_ms": float(mn), "max_ms": float(mx)}
        return out

    def set_profiling(self, on):
        check(self.lib.blsgpu_set_profiling(self.h, 1 if on else 0), "set_profiling")

    def last_msm_phase_ms(self):
        names = ["digits", "scan", "scatter", "order", "accumulate", "reduce", "combine", "total"]
        out = {}
        for i, nm in enumerate(names):
            v = ctypes.c_float()
            check(self.lib.blsgpu_last_msm_phase_ms(self.h, i, ctypes.byref(v)), "last_msm_phase_ms")
            out[nm] = v.value
        return out

    # -- bases -----------------------------------------------------------------------------------------
    def upload_bases(self, group, xy, infinity=None):
        w = 12 if group == 1 else 24
        xy = _u64(xy, (-1, w))
        n = xy.shape[0]
        inf = _flags(infinity, n)
        h = ctypes.c_void_p()
        fn = self.lib.blsgpu_g1_bases_upload if group == 1 else self.lib.blsgpu_g2_bases_upload
        check(fn(self.h, _ptr(xy), _ptr(inf), n, ctypes.byref(h)), "bases_upload")
        return ResidentBases(self, h, group)

    def bases_from_device(self, group, d_xy, d_inf, n):
        h = ctypes.c_void_p()
        fn = self.lib.blsgpu_g1_bases_from_device if group == 1 else self.lib.blsgpu_g2_bases_from_device
        check(fn(self.h, ctypes.c_void_p(d_xy), ctypes.c_void_p(d_inf) if d_inf else None, n, ctypes.byref(h)), "bases_from_device")
        return ResidentBases(self, h, group)

    def bases_from_scalars(self, group, scalars):
        s = scalars_to_bytes(scalars)
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_bases_from_scalars(self.h, group, _ptr(s), s.shape[0], ctypes.byref(h)), "bases_from_scalars")
        return ResidentBases(self, h, group)

    # -- MSM -------------------------------------------------------------------------------------------
    def msm(self, bases, scalars, first=0):
        """sum_i scalars[i] * bases[first+i] -> projective wire limbs (18 or 36 uint64)."""
        s = scalars_to_bytes(scalars)
        n = s.shape[0]
        out = np.zeros(18 if bases.group == 1 else 36, dtype=np.uint64)
        fn = self.lib.blsgpu_g1_msm if bases.group == 1 else self.lib.blsgpu_g2_msm
        check(fn(self.h, bases.handle, first, _ptr(s), n, _ptr(out)), "msm")
        return out

    def msm_device(self, bases, d_scalars, n, d_out, first=0):
        fn = self.lib.blsgpu_g1_msm_device if bases.group == 1 else self.lib.blsgpu_g2_msm_device
        check(fn(self.h, bases.handle, first, ctypes.c_void_p(d_scalars), n, ctypes.c_void_p(d_out)), "msm_device")

    # scalars as the reference stores them: (n, 4) u64 Montgomery limbs of `Scalar([u64; 4])` (scalar.rs:23-27); `to_bytes` runs on the device
    def set_scalar_form(self, form):
        """SCALAR_BYTES (0, default) or SCALAR_MONT (1) for every later scalar argument of this context (blsgpu_set_scalar_form)"""
        check(self.lib.blsgpu_set_scalar_form(self.h, int(form)), "set_scalar_form")

    def msm_mont(self, bases, scalar_limbs, first=0):
        """sum_i scalars[i] * bases[first+i] with scalars as (n, 4) u64 Montgomery limbs (`&[Scalar]` memory)."""
        s = _u64(scalar_limbs, (-1, 4))
        out = np.zeros(18 if bases.group == 1 else 36, dtype=np.uint64)
        fn = self.lib.blsgpu_g1_msm_mont if bases.group == 1 else self.lib.blsgpu_g2_msm_mont
        check(fn(self.h, bases.handle, first, _ptr(s), s.shape[0], _ptr(out)), "msm_mont")
        return out

    def msm_mont_device(self, bases, d_scalar_limbs, n, d_out, first=0):
        fn = self.lib.blsgpu_g1_msm_mont_device if bases.group == 1 else self.lib.blsgpu_g2_msm_mont_device
        check(fn(self.h, bases.handle, first, ctypes.c_void_p(d_scalar_limbs), n, ctypes.c_void_p(d_out)), "msm_mont_device")

    def mul_batch_mont(self, group, xy, infinity, scalar_limbs):
        """mul_batch with the scalars as (n, 4) u64 Montgomery limbs"""
        w = 12 if group == 1 else 24
        xy = _u64(xy, (-1, w))
        s = _u64(scalar_limbs, (xy.shape[0], 4))
        inf = _flags(infinity, xy.shape[0])
        out = np.zeros((xy.shape[0], 18 if group == 1 else 36), dtype=np.uint64)
        fn = self.lib.blsgpu_g1_mul_batch_mont if group == 1 else self.lib.blsgpu_g2_mul_batch_mont
        check(fn(self.h, _ptr(xy), _ptr(inf), _ptr(s), xy.shape[0], _ptr(out)), "mul_batch_mont")
        return out

    def mul_batch_mont_device(self, group, d_xy, d_inf, d_scalar_limbs, n, d_out):
        fn = self.lib.blsgpu_g1_mul_batch_mont_device if group == 1 else self.lib.blsgpu_g2_mul_batch_mont_device
        check(fn(self.h, ctypes.c_void_p(d_xy), ctypes.c_void_p(d_inf) if d_inf else None, ctypes.c_void_p(d_scalar_limbs), n, ctypes.c_void_p(d_out)), "mul_batch_mont_device")

    def msm_many(self, bases, scalar_sets):
        """k MSMs over the same resident bases; scalar_sets: (k, n, 32) uint8 (or a list of k scalar lists).  Returns (k, 18|36)."""
        if isinstance(scalar_sets, np.ndarray) and scalar_sets.dtype == np.uint8:
            s = np.ascontiguousarray(scalar_sets)
        else:
            s = np.stack([scalars_to_bytes(x) for x in scalar_sets])
        k, n = s.shape[0], s.shape[1]
        out = np.zeros((k, 18 if bases.group == 1 else 36), dtype=np.uint64)
        fn = self.lib.blsgpu_g1_msm_many if bases.group == 1 else self.lib.blsgpu_g2_msm_many
        check(fn(self.h, bases.handle, 0, _ptr(s), n, k, _ptr(out)), "msm_many")
        return out

    def msm_bytes(self, group, bases_uncompressed, scalars):
        """MSM on the reference's public encodings: bases = n uncompressed encodings (bytes), scalars = ints / (n,32) bytes;
        returns the uncompressed encoding of the sum."""
        size = 96 if group == 1 else 192
        buf = np.frombuffer(bytes(bases_uncompressed), dtype=np.uint8).copy() if not isinstance(bases_uncompressed, np.ndarray) else np.ascontiguousarray(bases_uncompressed, dtype=np.uint8)
        n = buf.size // size
        s = scalars_to_bytes(scalars)
        if s.shape[0] != n:
            raise ValueError("msm_bytes: bases and scalars differ in length")
        out = np.zeros(size, dtype=np.uint8)
        fn = self.lib.blsgpu_g1_msm_bytes if group == 1 else self.lib.blsgpu_g2_msm_bytes
        check(fn(self.h, _ptr(buf), _ptr(s), n, _ptr(out)), "msm_bytes")
        return out.tobytes()

    def set_bases_cache(self, entries):
        """keep the base arrays of repeated one-shot MSMs (`msm_host`, the mirrored `msm_g1` / `msm_g2`) resident: see blsgpu_set_bases_cache"""
        check(self.lib.blsgpu_set_bases_cache(self.h, int(entries)), "set_bases_cache")

    def set_bases_cache_verify(self, on):
        """recognise cached base arrays by a hash of every word (safe for buffers that are reused with other contents): blsgpu_set_bases_cache_verify"""
        check(self.lib.blsgpu_set_bases_cache_verify(self.h, 1 if on else 0), "set_bases_cache_verify")

    def msm_host(self, group, xy, infinity, scalars):
        w = 12 if group == 1 else 24
        xy = _u64(xy, (-1, w))
        s = scalars_to_bytes(scalars)
        if s.shape[0] != xy.shape[0]:
            raise ValueError("bases and scalars differ in length")
        inf = _flags(infinity, xy.shape[0])
        out = np.zeros(18 if group == 1 else 36, dtype=np.uint64)
        fn = self.lib.blsgpu_g1_msm_host if group == 1 else self.lib.blsgpu_g2_msm_host
        check(fn(self.h, _ptr(xy), _ptr(inf), _ptr(s), xy.shape[0], _ptr(out)), "msm_host")
        return out

    def mul_batch(self, group, xy, infinity, scalars):
        """out[i] = scalars[i] * points[i]: n affine points (n, 12|24) + n scalars -> (n, 18|36) projective wire limbs
        (`&G1Affine * &Scalar` element-wise, g1.rs:573-579 / g2.rs:626-632; exact for every curve point)."""
        w = 12 if group == 1 else 24
        xy = _u64(xy, (-1, w))
        s = scalars_to_bytes(scalars)
        if s.shape[0] != xy.shape[0]:
            raise ValueError("points and scalars differ in length")
        inf = _flags(infinity, xy.shape[0])
        out = np.zeros((xy.shape[0], 18 if group == 1 else 36), dtype=np.uint64)
        fn = self.lib.blsgpu_g1_mul_batch if group == 1 else self.lib.blsgpu_g2_mul_batch
        check(fn(self.h, _ptr(xy), _ptr(inf), _ptr(s), xy.shape[0], _ptr(out)), "mul_batch")
        return out

    def mul_batch_device(self, group, d_xy, d_inf, d_scalars, n, d_out):
        fn = self.lib.blsgpu_g1_mul_batch_device if group == 1 else self.lib.blsgpu_g2_mul_batch_device
        check(fn(self.h, ctypes.c_void_p(d_xy), ctypes.c_void_p(d_inf) if d_inf else None, ctypes.c_void_p(d_scalars), n, ctypes.c_void_p(d_out)), "mul_batch_device")

    # -- group helpers -----------------------------------------------------------------------------------
    def point_sum(self, group, xyz):
        w = 18 if group == 1 else 36
        xyz = _u64(xyz, (-1, w))
        out = np.zeros(w, dtype=np.uint64)
        fn = self.lib.blsgpu_g1_sum if group == 1 else self.lib.blsgpu_g2_sum
        check(fn(self.h, _ptr(xyz), xyz.shape[0], _ptr(out)), "sum")
        return out

    def point_sum_device(self, group, d_xyz, n, d_out):
        fn = self.lib.blsgpu_g1_sum_device if group == 1 else self.lib.blsgpu_g2_sum_device
        check(fn(self.h, ctypes.c_void_p(d_xyz), n, ctypes.c_void_p(d_out)), "sum_device")

    def batch_normalize(self, group, xyz):
        w = 18 if group == 1 else 36
        xyz = _u64(xyz, (-1, w))
        n = xyz.shape[0]
        xy = np.zeros((n, w * 2 // 3), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        fn = self.lib.blsgpu_g1_batch_normalize if group == 1 else self.lib.blsgpu_g2_batch_normalize
        check(fn(self.h, _ptr(xyz), n, _ptr(xy), _ptr(inf)), "batch_normalize")
        return xy, inf

    def point_op(self, group, op, a, b=None, b_inf=None):
        w = 18 if group == 1 else 36
        a = _u64(a, (-1, w))
        n = a.shape[0]
        if b is not None:
            b = _u64(b, (n, -1))
        out = np.zeros((n, w), dtype=np.uint64)
        check(self.lib.blsgpu_point_op(self.h, group, op, _ptr(a), _ptr(b), _ptr(_flags(b_inf, n)), n, _ptr(out)), "point_op")
        return out

    # -- batched (de)serialisation + validation --------------------------------------------------------------
    def points_from_bytes(self, group, data, compressed=True, checked=True):
        """data: (n, 48|96|192) uint8 or bytes.  Returns (xy, infinity, ok) like `from_compressed` & friends."""
        size = (48 if group == 1 else 96) * (1 if compressed else 2)
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        buf = buf.reshape(-1, size)
        n = buf.shape[0]
        xy = np.zeros((n, 12 if group == 1 else 24), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8); ok = np.zeros(n, dtype=np.uint8)
        fn = self.lib.blsgpu_g1_from_bytes_batch if group == 1 else self.lib.blsgpu_g2_from_bytes_batch
        check(fn(self.h, _ptr(buf), n, 1 if compressed else 0, 1 if checked else 0, _ptr(xy), _ptr(inf), _ptr(ok)), "from_bytes_batch")
        return xy, inf, ok

    def points_to_bytes(self, group, xy, infinity=None, compressed=True):
        w = 12 if group == 1 else 24
        xy = _u64(xy, (-1, w))
        n = xy.shape[0]
        size = (48 if group == 1 else 96) * (1 if compressed else 2)
        out = np.zeros((n, size), dtype=np.uint8)
        fn = self.lib.blsgpu_g1_to_bytes_batch if group == 1 else self.lib.blsgpu_g2_to_bytes_batch
        check(fn(self.h, _ptr(xy), _ptr(_flags(infinity, n)), n, 1 if compressed else 0, _ptr(out)), "to_bytes_batch")
        return out

    # -- field self-test hooks ---------------------------------------------------------------------------
    def _elem_op(self, fn, words, op, a, b):
        a = _u64(a, (-1, words))
        if b is not None:
            b = _u64(b, (a.shape[0], words))
        out = np.zeros_like(a)
        check(fn(self.h, op, _ptr(a), _ptr(b), a.shape[0], _ptr(out)), "field op")
        return out

    def fp_op(self, op, a, b=None):
        return self._elem_op(self.lib.blsgpu_fp_op, 6, op, a, b)

    def fp2_op(self, op, a, b=None):
        return self._elem_op(self.lib.blsgpu_fp2_op, 12, op, a, b)

    def fp6_op(self, op, a, b=None):
        return self._elem_op(self.lib.blsgpu_fp6_op, 36, op, a, b)

    def fp12_op(self, op, a, b=None):
        return self._elem_op(self.lib.blsgpu_fp12_op, 72, op, a, b)

    # ---- hash-to-curve (reference: src/hash_to_curve/) ----
    def hash_to_curve(self, group, msgs, dst, encode_only=False):
        """`G::hash_to_curve(msg, dst)` (or `encode_to_curve`) with ExpandMsgXmd<Sha256> for a list of byte strings;
        returns (n, 18 | 36) u64 projective points in the reference's limbs."""
        msgs = [bytes(m) for m in msgs]
        n = len(msgs)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64) if n else 0
        blob = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8).copy()
        d = np.frombuffer(bytes(dst) + b"\0", dtype=np.uint8).copy()
        out = np.zeros((n, 18 if group == 1 else 36), dtype=np.uint64)
        fn = self.lib.blsgpu_g1_hash_to_curve_batch if group == 1 else self.lib.blsgpu_g2_hash_to_curve_batch
        check(fn(self.h, _ptr(blob), _ptr(offs), n, _ptr(d), len(dst), 1 if encode_only else 0, _ptr(out)), "hash_to_curve")
        return out

    @staticmethod
    def _msgs(msgs, dst):
        msgs = [bytes(m) for m in msgs]
        offs = np.zeros(len(msgs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64) if msgs else 0
        blob = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8).copy()
        d = np.frombuffer(bytes(dst) + b"\0", dtype=np.uint8).copy()
        return len(msgs), blob, offs, d

    def hash_to_curve_expander(self, group, expander, msgs, dst, encode_only=False):
        """`hash_to_curve::<X>` / `encode_to_curve::<X>` with X chosen by `expander` (EXPAND_XMD_SHA256 / _XMD_SHA512 / _XOF_SHAKE128 / _XOF_SHAKE256:
        expand_msg.rs:167-328); returns (n, 18 | 36) u64 projective points"""
        n, blob, offs, d = self._msgs(msgs, dst)
        out = np.zeros((n, 18 if group == 1 else 36), dtype=np.uint64)
        check(self.lib.blsgpu_hash_to_curve_expander_batch(self.h, group, int(expander), _ptr(blob), _ptr(offs), n, _ptr(d), len(dst), 1 if encode_only else 0, _ptr(out)), "hash_to_curve_expander")
        return out

    def hash_to_curve_from_uniform(self, group, uniform, encode_only=False):
        """the part of hash_to_curve behind the expander (mod.rs:86-108): `uniform` = (n, count * M * 64) bytes per message (count = 1 | 2, M = 1 for G1,
        2 for G2) -> from_okm -> map_to_curve -> sum -> clear_h; returns (n, 18 | 36) u64 projective points"""
        per = (1 if encode_only else 2) * (1 if group == 1 else 2) * 64
        u = np.ascontiguousarray(np.asarray(uniform, dtype=np.uint8)).reshape(-1, per)
        out = np.zeros((u.shape[0], 18 if group == 1 else 36), dtype=np.uint64)
        check(self.lib.blsgpu_hash_to_curve_from_uniform_batch(self.h, group, _ptr(u), u.shape[0], 1 if encode_only else 0, _ptr(out)), "hash_to_curve_from_uniform")
        return out

    def expand_message(self, expander, msgs, dst, len_in_bytes):
        """`ExpandMessage::init_expand` + reading all the bytes: (n, len_in_bytes) uint8"""
        n, blob, offs, d = self._msgs(msgs, dst)
        out = np.zeros((n, len_in_bytes), dtype=np.uint8)
        check(self.lib.blsgpu_expand_message_batch(self.h, int(expander), _ptr(blob), _ptr(offs), n, _ptr(d), len(dst), len_in_bytes, _ptr(out)), "expand_message")
        return out

    def hash_to_scalar(self, expander, msgs, dst, count=1):
        """`hash_to_field::<X, Scalar>` (mod.rs:32-49, map_scalar.rs:10-25): (n, count, 4) u64 Montgomery limbs"""
        n, blob, offs, d = self._msgs(msgs, dst)
        out = np.zeros((n, count, 4), dtype=np.uint64)
        check(self.lib.blsgpu_hash_to_scalar_batch(self.h, int(expander), _ptr(blob), _ptr(offs), n, _ptr(d), len(dst), count, _ptr(out)), "hash_to_scalar")
        return out

    # ---- scalar field Fr (reference: src/scalar.rs) ----
    def fr_op(self, op, a, b=None, return_flags=False):
        """element-wise Scalar arithmetic on (n, 4) u64 Montgomery limbs; op 0 mul, 1 add, 2 sub, 3 square, 4 invert,
        5 neg, 6 double.  For invert, return_flags also returns the `is_some` bytes (0 where the input was zero)."""
        a = _u64(a, (-1, 4))
        if b is not None:
            b = _u64(b, (a.shape[0], 4))
        out = np.zeros_like(a)
        flags = np.ones(a.shape[0], dtype=np.uint8)
        check(self.lib.blsgpu_fr_op(self.h, op, _ptr(a), _ptr(b), a.shape[0], _ptr(out), _ptr(flags)), "fr_op")
        return (out, flags) if return_flags else out

    def fr_ntt(self, values, inverse=False):
        """radix-2 transform of 2^k Scalars ((n, 4) u64 Montgomery limbs), natural order in and out; see
        include/bls12_381_hip.h for the definition."""
        v = _u64(values, (-1, 4)).copy()
        n = v.shape[0]
        if n == 0 or n & (n - 1):
            raise ValueError("fr_ntt: length must be a power of two")
        check(self.lib.blsgpu_fr_ntt(self.h, _ptr(v), n.bit_length() - 1, 1 if inverse else 0), "fr_ntt")
        return v

    def fr_ntt_device(self, d_ptr, log_n, inverse=False):
        check(self.lib.blsgpu_fr_ntt_device(self.h, d_ptr, log_n, 1 if inverse else 0), "fr_ntt_device")

    def fr_to_bytes(self, limbs, return_flags=False):
        """`Scalar::to_bytes` (scalar.rs:284-296) over (n, 4) u64 Montgomery limbs -> (n, 32) uint8; flags: limbs below r"""
        a = _u64(limbs, (-1, 4))
        out = np.zeros((a.shape[0], 32), dtype=np.uint8)
        ok = np.ones(a.shape[0], dtype=np.uint8)
        check(self.lib.blsgpu_fr_to_bytes(self.h, _ptr(a), a.shape[0], _ptr(out), _ptr(ok)), "fr_to_bytes")
        return (out, ok) if return_flags else out

    def fr_from_bytes(self, data):
        """`Scalar::from_bytes` (scalar.rs:256-280) over (n, 32) uint8 -> ((n, 4) u64 Montgomery limbs, is_some bytes)"""
        b = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros((b.shape[0], 4), dtype=np.uint64)
        ok = np.ones(b.shape[0], dtype=np.uint8)
        check(self.lib.blsgpu_fr_from_bytes(self.h, _ptr(b), b.shape[0], _ptr(out), _ptr(ok)), "fr_from_bytes")
        return out, ok

    def fr_from_bytes_wide(self, data):
        """`Scalar::from_bytes_wide` (scalar.rs:300-331) over (n, 64) uint8 -> (n, 4) u64 Montgomery limbs"""
        b = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8).reshape(-1, 64)
        out = np.zeros((b.shape[0], 4), dtype=np.uint64)
        check(self.lib.blsgpu_fr_from_bytes_wide(self.h, _ptr(b), b.shape[0], _ptr(out)), "fr_from_bytes_wide")
        return out

    def fr_to_bytes_device(self, d_limbs, n, d_bytes, d_ok=None):
        check(self.lib.blsgpu_fr_to_bytes_device(self.h, ctypes.c_void_p(d_limbs), n, ctypes.c_void_p(d_bytes), ctypes.c_void_p(d_ok) if d_ok else None), "fr_to_bytes_device")

    def fr_from_bytes_device(self, d_bytes, n, d_limbs, d_ok=None):
        check(self.lib.blsgpu_fr_from_bytes_device(self.h, ctypes.c_void_p(d_bytes), n, ctypes.c_void_p(d_limbs), ctypes.c_void_p(d_ok) if d_ok else None), "fr_from_bytes_device")

    def fr_from_bytes_wide_device(self, d_bytes, n, d_limbs):
        check(self.lib.blsgpu_fr_from_bytes_wide_device(self.h, ctypes.c_void_p(d_bytes), n, ctypes.c_void_p(d_limbs)), "fr_from_bytes_wide_device")

    def fp_mul_throughput(self, iters=2000):
        v = ctypes.c_double()
        check(self.lib.blsgpu_fp_mul_throughput(self.h, iters, ctypes.byref(v)), "fp_mul_throughput")
        return v.value

    def mad_throughput(self, iters=2000):
        v = ctypes.c_double()
        check(self.lib.blsgpu_mad_throughput(self.h, iters, ctypes.byref(v)), "mad_throughput")
        return v.value

    # -- pairings ------------------------------------------------------------------------------------------
    def _pair_args(self, g1_xy, g1_inf, g2_xy, g2_inf):
        g1 = _u64(g1_xy, (-1, 12))
        g2 = _u64(g2_xy, (-1, 24))
        if g1.shape[0] != g2.shape[0]:
            raise ValueError("G1 and G2 inputs differ in length")
        n = g1.shape[0]
        return g1, _flags(g1_inf, n), g2, _flags(g2_inf, n), n

    def pairing_batch(self, g1_xy, g1_inf, g2_xy, g2_inf):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        out = np.zeros((n, 72), dtype=np.uint64)
        check(self.lib.blsgpu_pairing_batch(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), n, _ptr(out)), "pairing_batch")
        return out

    def miller_loop_batch(self, g1_xy, g1_inf, g2_xy, g2_inf):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        out = np.zeros((n, 72), dtype=np.uint64)
        check(self.lib.blsgpu_miller_loop_batch(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), n, _ptr(out)), "miller_loop_batch")
        return out

    def multi_miller_loop(self, g1_xy, g1_inf, g2_xy, g2_inf):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        out = np.zeros(72, dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), n, _ptr(out)), "multi_miller_loop")
        return out

    def multi_miller_loop_many(self, g1_xy, g1_inf, g2_xy, g2_inf, offsets, final_exp=True):
        """N independent `multi_miller_loop`s (CSR offsets over the terms) -> (N, 72): `MillerLoopResult`s, or -- final_exp --
        `Gt`s (`blsgpu_multi_miller_loop_many`; pairings.rs:554-603 once per segment)"""
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.uint64))
        if off.ndim != 1 or off.shape[0] < 1 or int(off[-1]) != n:
            raise ValueError("multi_miller_loop_many: offsets must be nseg + 1 values ending at the number of terms")
        nseg = off.shape[0] - 1
        out = np.zeros((nseg, 72), dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_many(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), _ptr(off), nseg, 1 if final_exp else 0, _ptr(out)),
              "multi_miller_loop_many")
        return out

    def multi_miller_loop_many_device(self, d_g1, d_g2, d_offsets, nseg, total_terms, d_out, max_seg_terms=0, final_exp=True, d_g1_inf=None, d_g2_inf=None):
        check(self.lib.blsgpu_multi_miller_loop_many_device(self.h, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g1_inf), ctypes.c_void_p(d_g2), ctypes.c_void_p(d_g2_inf),
                                                            ctypes.c_void_p(d_offsets), nseg, total_terms, max_seg_terms, 1 if final_exp else 0, ctypes.c_void_p(d_out)),
              "multi_miller_loop_many_device")

    # -- the widened rows with device pointers (a chain that stays in HBM) and bulk signature verification -------------
    def batch_normalize_device(self, group, d_xyz, n, d_xy, d_inf):
        fn = self.lib.blsgpu_g1_batch_normalize_device if group == 1 else self.lib.blsgpu_g2_batch_normalize_device
        check(fn(self.h, ctypes.c_void_p(d_xyz), n, ctypes.c_void_p(d_xy), ctypes.c_void_p(d_inf)), "batch_normalize_device")

    def points_from_bytes_device(self, group, d_bytes, n, d_xy, d_inf, d_ok, compressed=True, checked=True):
        fn = self.lib.blsgpu_g1_from_bytes_batch_device if group == 1 else self.lib.blsgpu_g2_from_bytes_batch_device
        check(fn(self.h, ctypes.c_void_p(d_bytes), n, 1 if compressed else 0, 1 if checked else 0, ctypes.c_void_p(d_xy), ctypes.c_void_p(d_inf), ctypes.c_void_p(d_ok)),
              "from_bytes_batch_device")

    def points_to_bytes_device(self, group, d_xy, d_inf, n, d_out, compressed=True):
        fn = self.lib.blsgpu_g1_to_bytes_batch_device if group == 1 else self.lib.blsgpu_g2_to_bytes_batch_device
        check(fn(self.h, ctypes.c_void_p(d_xy), ctypes.c_void_p(d_inf), n, 1 if compressed else 0, ctypes.c_void_p(d_out)), "to_bytes_batch_device")

    def gt_mul_scalar_batch_device(self, d_gt, d_scalars, n, d_out):
        check(self.lib.blsgpu_gt_mul_scalar_batch_device(self.h, ctypes.c_void_p(d_gt), ctypes.c_void_p(d_scalars), n, ctypes.c_void_p(d_out)), "gt_mul_scalar_batch_device")

    def gt_is_identity_device(self, d_gt, n, d_flags):
        check(self.lib.blsgpu_gt_is_identity_device(self.h, ctypes.c_void_p(d_gt), n, ctypes.c_void_p(d_flags)), "gt_is_identity_device")

    def bls_verify_batch(self, mode, pk_bytes, sig_bytes, msgs, dst):
        """Bulk BLS verification from bytes (blsgpu_bls_verify_batch): mode 0 = public keys in G1 (48 B) / signatures in G2 (96 B),
        mode 1 the other way round; msgs: list of bytes.  -> (n,) uint8: 1 valid, 0 invalid, 2 bad public key, 3 bad signature"""
        n = len(msgs)
        pk = np.ascontiguousarray(np.frombuffer(bytes(pk_bytes), dtype=np.uint8)) if not isinstance(pk_bytes, np.ndarray) else np.ascontiguousarray(pk_bytes, dtype=np.uint8).reshape(-1)
        sg = np.ascontiguousarray(np.frombuffer(bytes(sig_bytes), dtype=np.uint8)) if not isinstance(sig_bytes, np.ndarray) else np.ascontiguousarray(sig_bytes, dtype=np.uint8).reshape(-1)
        if pk.shape[0] != n * (48 if mode == 0 else 96) or sg.shape[0] != n * (96 if mode == 0 else 48):
            raise ValueError("bls_verify_batch: key / signature bytes do not match the number of messages")
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(m) for m in msgs])
        blob = np.frombuffer(b"".join(bytes(m) for m in msgs), dtype=np.uint8).copy() if int(off[-1]) else np.zeros(1, dtype=np.uint8)
        d = np.frombuffer(bytes(dst), dtype=np.uint8).copy() if len(dst) else np.zeros(1, dtype=np.uint8)
        out = np.zeros(n, dtype=np.uint8)
        check(self.lib.blsgpu_bls_verify_batch(self.h, mode, _ptr(pk), _ptr(sg), _ptr(blob), _ptr(off), n, _ptr(d), len(dst), _ptr(out)), "bls_verify_batch")
        return out

    def bls_verify_batch_device(self, mode, d_pk, d_sig, d_msgs, d_offsets, n, d_dst, dst_len, d_verdict):
        check(self.lib.blsgpu_bls_verify_batch_device(self.h, mode, ctypes.c_void_p(d_pk), ctypes.c_void_p(d_sig), ctypes.c_void_p(d_msgs), ctypes.c_void_p(d_offsets), n,
                                                      ctypes.c_void_p(d_dst), dst_len, ctypes.c_void_p(d_verdict)), "bls_verify_batch_device")

    # -- G2Prepared resident on the device (pairings.rs:487-546) and its consumers (:554-603) -----------------------
    def g2_prepare(self, g2_xy, g2_inf=None):
        g2 = _u64(g2_xy, (-1, 24))
        m = g2.shape[0]
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_g2_prepare(self.h, _ptr(g2), _ptr(_flags(g2_inf, m)) if g2_inf is not None else None, m, ctypes.byref(h)), "g2_prepare")
        return PreparedG2Table(self, h)

    def g2_prepare_device(self, d_g2, d_inf, m):
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_g2_prepare_device(self.h, ctypes.c_void_p(d_g2), ctypes.c_void_p(d_inf), m, ctypes.byref(h)), "g2_prepare_device")
        return PreparedG2Table(self, h)

    def _prepared_args(self, g1_xy, g1_inf, q_index, g2_xy, g2_inf):
        g1 = _u64(g1_xy, (-1, 12))
        n = g1.shape[0]
        qi = None
        if q_index is not None:
            qi = np.ascontiguousarray(np.asarray(q_index, dtype=np.uint32))
            if qi.shape != (n,):
                raise ValueError("q_index must name one table entry (or UNPREPARED) per term")
        g2 = None
        if g2_xy is not None:
            g2 = _u64(g2_xy, (-1, 24))
            if g2.shape[0] != n:
                raise ValueError("G1 and G2 inputs differ in length")
        f2 = _flags(g2_inf, n) if (g2 is not None and g2_inf is not None) else None
        return g1, _flags(g1_inf, n), g2, f2, qi, n

    def multi_miller_loop_prepared(self, g1_xy, g1_inf, table, q_index, g2_xy=None, g2_inf=None):
        """prod_i ML(g1[i], Q_i) with Q_i = table[q_index[i]] or -- q_index[i] = UNPREPARED -- g2[i] (pairings.rs:554-603)"""
        g1, f1, g2, f2, qi, n = self._prepared_args(g1_xy, g1_inf, q_index, g2_xy, g2_inf)
        out = np.zeros(72, dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_prepared(self.h, _ptr(g1), _ptr(f1), _ptr(g2) if g2 is not None else None, _ptr(f2) if f2 is not None else None,
                                                         _ptr(qi) if qi is not None else None, table.handle if table is not None else None, n, _ptr(out)),
              "multi_miller_loop_prepared")
        return out

    def multi_miller_loop_prepared_many(self, g1_xy, g1_inf, table, q_index, offsets, g2_xy=None, g2_inf=None, final_exp=True):
        g1, f1, g2, f2, qi, n = self._prepared_args(g1_xy, g1_inf, q_index, g2_xy, g2_inf)
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.uint64))
        if off.ndim != 1 or off.shape[0] < 1 or int(off[-1]) != n:
            raise ValueError("multi_miller_loop_prepared_many: offsets must be nseg + 1 values ending at the number of terms")
        nseg = off.shape[0] - 1
        out = np.zeros((nseg, 72), dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_prepared_many(self.h, _ptr(g1), _ptr(f1), _ptr(g2) if g2 is not None else None, _ptr(f2) if f2 is not None else None,
                                                              _ptr(qi) if qi is not None else None, table.handle if table is not None else None, _ptr(off), nseg,
                                                              1 if final_exp else 0, _ptr(out)), "multi_miller_loop_prepared_many")
        return out

    def multi_miller_loop_prepared_device(self, d_g1, table, d_q_index, n, d_out, d_g2=None, d_g1_inf=None, d_g2_inf=None):
        check(self.lib.blsgpu_multi_miller_loop_prepared_device(self.h, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g1_inf), ctypes.c_void_p(d_g2), ctypes.c_void_p(d_g2_inf),
                                                                ctypes.c_void_p(d_q_index), table.handle if table is not None else None, n, ctypes.c_void_p(d_out)),
              "multi_miller_loop_prepared_device")

    def multi_miller_loop_prepared_many_device(self, d_g1, table, d_q_index, d_offsets, nseg, total_terms, d_out, max_seg_terms=0, final_exp=True, d_g2=None, d_g1_inf=None,
                                               d_g2_inf=None):
        check(self.lib.blsgpu_multi_miller_loop_prepared_many_device(self.h, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g1_inf), ctypes.c_void_p(d_g2), ctypes.c_void_p(d_g2_inf),
                                                                     ctypes.c_void_p(d_q_index), table.handle if table is not None else None, ctypes.c_void_p(d_offsets), nseg,
                                                                     total_terms, max_seg_terms, 1 if final_exp else 0, ctypes.c_void_p(d_out)),
              "multi_miller_loop_prepared_many_device")

    def wide_status(self):
        """'' when the small-batch (wide) pairing programs are loaded, otherwise the reason they are not"""
        return (self.lib.blsgpu_wide_status(self.h) or b"").decode()

    def final_exponentiation_batch(self, f):
        f = _u64(f, (-1, 72))
        out = np.zeros_like(f)
        check(self.lib.blsgpu_final_exponentiation_batch(self.h, _ptr(f), f.shape[0], _ptr(out)), "final_exponentiation")
        return out

    def gt_mul_scalar_batch(self, gt, scalars):
        """out[i] = gt[i] * scalars[i] (`&Gt * &Scalar`, pairings.rs:297-322); scalars: ints or (n, 32) little-endian bytes"""
        gt = _u64(gt, (-1, 72))
        sb = scalars_to_bytes(scalars)
        if sb.shape[0] != gt.shape[0]:
            raise ValueError("gt_mul_scalar_batch: lengths differ")
        out = np.zeros_like(gt)
        check(self.lib.blsgpu_gt_mul_scalar_batch(self.h, _ptr(gt), _ptr(sb), gt.shape[0], _ptr(out)), "gt_mul_scalar_batch")
        return out

    # device-pointer variants (asynchronous on the context's stream); pointers are plain ints (e.g. tensor.data_ptr())
    def pairing_batch_device(self, d_g1, d_g2, n, d_out, d_g1_inf=None, d_g2_inf=None):
        check(self.lib.blsgpu_pairing_batch_device(self.h, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g1_inf), ctypes.c_void_p(d_g2), ctypes.c_void_p(d_g2_inf), n,
                                                   ctypes.c_void_p(d_out)), "pairing_batch_device")

    def miller_loop_batch_device(self, d_g1, d_g2, n, d_out, d_g1_inf=None, d_g2_inf=None):
        check(self.lib.blsgpu_miller_loop_batch_device(self.h, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g1_inf), ctypes.c_void_p(d_g2), ctypes.c_void_p(d_g2_inf), n,
                                                       ctypes.c_void_p(d_out)), "miller_loop_batch_device")

    def multi_miller_loop_device(self, d_g1, d_g2, n, d_out, d_g1_inf=None, d_g2_inf=None):
        check(self.lib.blsgpu_multi_miller_loop_device(self.h, ctypes.c_void_p(d_g1), ctypes.c_void_p(d_g1_inf), ctypes.c_void_p(d_g2), ctypes.c_void_p(d_g2_inf), n,
                                                       ctypes.c_void_p(d_out)), "multi_miller_loop_device")

    def final_exponentiation_device(self, d_in, n, d_out):
        check(self.lib.blsgpu_final_exponentiation_device(self.h, ctypes.c_void_p(d_in), n, ctypes.c_void_p(d_out)), "final_exponentiation_device")

    def fp12_product_device(self, d_in, n, d_out):
        check(self.lib.blsgpu_fp12_product_device(self.h, ctypes.c_void_p(d_in), n, ctypes.c_void_p(d_out)), "fp12_product_device")

    def fp12_product(self, f):
        f = _u64(f, (-1, 72))
        out = np.zeros(72, dtype=np.uint64)
        check(self.lib.blsgpu_fp12_product(self.h, _ptr(f), f.shape[0], _ptr(out)), "fp12_product")
        return out


class GroupBases:
    """Resident bases sharded over the members of a Group (`blsgpu_group_bases`): member k holds a contiguous slice."""

    def __init__(self, group, handle, gid):
        self.group, self.handle, self.gid = group, handle, gid

    def __len__(self):
        return int(_lib.load().blsgpu_group_bases_len(self.handle))

    def free(self):
        if self.handle:
            _lib.load().blsgpu_group_bases_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GroupPreparedG2:
    """the same `G2Prepared` table on every member of a Group (`blsgpu_group_g2_prepared`)"""

    def __init__(self, group, handle):
        self.group, self.handle = group, handle

    def __len__(self):
        return int(_lib.load().blsgpu_group_g2_prepared_len(self.handle)) if self.handle else 0

    def free(self):
        if self.handle:
            _lib.load().blsgpu_group_g2_prepared_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Group:
    """The hot path sharded over several GPUs of one node from ONE process (`blsgpu_group`, include/bls12_381_hip.h): one
    context and one host thread per listed device; partial results (one group element per member) are folded with `Sum`
    (g1.rs:161-171) / `MillerLoopResult + MillerLoopResult` (pairings.rs:179-186).  A device may be listed more than once."""

    def __init__(self, devices):
        self.lib = _lib.load()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_group_create(devs, len(devices), ctypes.byref(h)), "group_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.blsgpu_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.lib.blsgpu_group_size(self.h))

    def set_assume_subgroup(self, on):
        for k in range(len(self)):
            check(self.lib.blsgpu_set_assume_subgroup(ctypes.c_void_p(self.lib.blsgpu_group_ctx(self.h, k)), 1 if on else 0), "set_assume_subgroup")

    def upload_bases(self, group, xy, infinity=None):
        w = 12 if group == 1 else 24
        xy = _u64(xy, (-1, w))
        n = xy.shape[0]
        inf = _flags(infinity, n)
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_group_bases_upload(self.h, group, _ptr(xy), _ptr(inf), n, ctypes.byref(h)), "group_bases_upload")
        return GroupBases(self, h, group)

    def bases_from_scalars(self, group, scalars):
        sb = scalars_to_bytes(scalars)
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_group_bases_from_scalars(self.h, group, _ptr(sb), sb.shape[0], ctypes.byref(h)), "group_bases_from_scalars")
        return GroupBases(self, h, group)

    def msm(self, bases, scalars):
        sb = scalars_to_bytes(scalars)
        out = np.zeros(18 if bases.gid == 1 else 36, dtype=np.uint64)
        fn = self.lib.blsgpu_g1_msm_sharded if bases.gid == 1 else self.lib.blsgpu_g2_msm_sharded
        check(fn(self.h, bases.handle, _ptr(sb), sb.shape[0], _ptr(out)), "msm_sharded")
        return out

    # device-pointer, asynchronous MSMs on every member (the pipelined headline path from one process)
    def member_ctx(self, k):
        """raw handle of member k's context (blsgpu_group_ctx) for per-context calls through the ctypes layer"""
        return ctypes.c_void_p(self.lib.blsgpu_group_ctx(self.h, k))

    def shard_sizes(self, n):
        """points per member for n resident points (the rule of blsgpu_group_bases_*: contiguous slices, sizes differ by at most one)"""
        w = len(self)
        return [n // w + (1 if k < n % w else 0) for k in range(w)]

    def set_pipelining(self, on):
        check(self.lib.blsgpu_group_set_pipelining(self.h, 1 if on else 0), "group_set_pipelining")

    def synchronize(self):
        check(self.lib.blsgpu_group_synchronize(self.h), "group_synchronize")

    def msm_sharded_device(self, bases, d_scalars, d_partials):
        """member k: partial sum of its whole resident slice times the scalars at d_scalars[k] -> d_partials[k] (device pointers as ints,
        each in member k's device memory); only enqueues"""
        w = len(self)
        sp = (ctypes.c_void_p * w)(*[ctypes.c_void_p(int(p)) for p in d_scalars])
        op = (ctypes.c_void_p * w)(*[ctypes.c_void_p(int(p)) for p in d_partials])
        fn = self.lib.blsgpu_g1_msm_sharded_device if bases.gid == 1 else self.lib.blsgpu_g2_msm_sharded_device
        check(fn(self.h, bases.handle, sp, op), "msm_sharded_device")

    def partials_fold(self, gid, d_partials, lag=0):
        """the sum of the members' partial sums at d_partials (waiting, per member, for all but the `lag` most recent MSM calls)"""
        w = len(self)
        op = (ctypes.c_void_p * w)(*[ctypes.c_void_p(int(p)) for p in d_partials])
        out = np.zeros(18 if gid == 1 else 36, dtype=np.uint64)
        fn = self.lib.blsgpu_g1_partials_fold if gid == 1 else self.lib.blsgpu_g2_partials_fold
        check(fn(self.h, op, lag, _ptr(out)), "partials_fold")
        return out

    def pairings_sharded_device(self, mode, d_g1, d_g2, counts, d_out, d_g1_inf=None, d_g2_inf=None):
        """member k: mode 0 pairings / 1 raw Miller values / 2 its local multi_miller_loop product of counts[k] pairs at d_g1[k], d_g2[k] -> d_out[k]; enqueues only"""
        w = len(self)
        arr = lambda ps: (ctypes.c_void_p * w)(*[ctypes.c_void_p(int(p)) for p in ps]) if ps is not None else None
        cnt = (ctypes.c_size_t * w)(*[int(c) for c in counts])
        check(self.lib.blsgpu_pairings_sharded_device(self.h, mode, arr(d_g1), arr(d_g1_inf), arr(d_g2), arr(d_g2_inf), cnt, arr(d_out)), "pairings_sharded_device")

    def fp12_partials_fold_device(self, d_partials, d_out, final_exp=False):
        w = len(self)
        op = (ctypes.c_void_p * w)(*[ctypes.c_void_p(int(p)) for p in d_partials])
        check(self.lib.blsgpu_fp12_partials_fold_device(self.h, op, 1 if final_exp else 0, ctypes.c_void_p(int(d_out))), "fp12_partials_fold_device")

    # G2Prepared tables on every member; the prepared Miller loops sharded like the unprepared ones
    def g2_prepare(self, g2_xy, g2_inf=None):
        g2 = _u64(g2_xy, (-1, 24))
        m = g2.shape[0]
        h = ctypes.c_void_p()
        check(self.lib.blsgpu_group_g2_prepare(self.h, _ptr(g2), _ptr(_flags(g2_inf, m)) if g2_inf is not None else None, m, ctypes.byref(h)), "group_g2_prepare")
        return GroupPreparedG2(self, h)

    def multi_miller_loop_prepared(self, g1_xy, g1_inf, table, q_index, g2_xy=None, g2_inf=None, final_exp=False):
        g1, f1, g2, f2, qi, n = Context._prepared_args(self, g1_xy, g1_inf, q_index, g2_xy, g2_inf)
        out = np.zeros(72, dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_prepared_sharded(self.h, _ptr(g1), _ptr(f1), _ptr(g2) if g2 is not None else None, _ptr(f2) if f2 is not None else None,
                                                                 _ptr(qi) if qi is not None else None, table.handle if table is not None else None, n, 1 if final_exp else 0,
                                                                 _ptr(out)), "multi_miller_loop_prepared_sharded")
        return out

    def multi_miller_loop_prepared_many(self, g1_xy, g1_inf, table, q_index, offsets, g2_xy=None, g2_inf=None, final_exp=True):
        g1, f1, g2, f2, qi, n = Context._prepared_args(self, g1_xy, g1_inf, q_index, g2_xy, g2_inf)
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.uint64))
        if off.ndim != 1 or off.shape[0] < 1 or int(off[-1]) != n:
            raise ValueError("multi_miller_loop_prepared_many: offsets must be nseg + 1 values ending at the number of terms")
        nseg = off.shape[0] - 1
        out = np.zeros((nseg, 72), dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_prepared_many_sharded(self.h, _ptr(g1), _ptr(f1), _ptr(g2) if g2 is not None else None, _ptr(f2) if f2 is not None else None,
                                                                      _ptr(qi) if qi is not None else None, table.handle if table is not None else None, _ptr(off), nseg,
                                                                      1 if final_exp else 0, _ptr(out)), "multi_miller_loop_prepared_many_sharded")
        return out

    def partials_fold_device(self, gid, d_partials, d_out, lag=0):
        """the same fold queued on the members' streams: the sum lands at d_out (member 0's device memory), nothing is synchronised"""
        w = len(self)
        op = (ctypes.c_void_p * w)(*[ctypes.c_void_p(int(p)) for p in d_partials])
        fn = self.lib.blsgpu_g1_partials_fold_device if gid == 1 else self.lib.blsgpu_g2_partials_fold_device
        check(fn(self.h, op, lag, ctypes.c_void_p(int(d_out))), "partials_fold_device")

    _pair_args = Context._pair_args

    def pairing_batch(self, g1_xy, g1_inf, g2_xy, g2_inf):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        out = np.zeros((n, 72), dtype=np.uint64)
        check(self.lib.blsgpu_pairing_batch_sharded(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), n, _ptr(out)), "pairing_batch_sharded")
        return out

    def miller_loop_batch(self, g1_xy, g1_inf, g2_xy, g2_inf):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        out = np.zeros((n, 72), dtype=np.uint64)
        check(self.lib.blsgpu_miller_loop_batch_sharded(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), n, _ptr(out)), "miller_loop_batch_sharded")
        return out

    def multi_miller_loop(self, g1_xy, g1_inf, g2_xy, g2_inf, final_exp=False):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        out = np.zeros(72, dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_sharded(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), n, 1 if final_exp else 0, _ptr(out)), "multi_miller_loop_sharded")
        return out

    def multi_miller_loop_many(self, g1_xy, g1_inf, g2_xy, g2_inf, offsets, final_exp=True):
        g1, f1, g2, f2, n = self._pair_args(g1_xy, g1_inf, g2_xy, g2_inf)
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.uint64))
        if off.ndim != 1 or off.shape[0] < 1 or int(off[-1]) != n:
            raise ValueError("multi_miller_loop_many: offsets must be nseg + 1 values ending at the number of terms")
        nseg = off.shape[0] - 1
        out = np.zeros((nseg, 72), dtype=np.uint64)
        check(self.lib.blsgpu_multi_miller_loop_many_sharded(self.h, _ptr(g1), _ptr(f1), _ptr(g2), _ptr(f2), _ptr(off), nseg, 1 if final_exp else 0, _ptr(out)),
              "multi_miller_loop_many_sharded")
        return out


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ======================================================================================================
# value types mirroring the reference
# ======================================================================================================
class Scalar:
    """Element of Fr (src/scalar.rs).  Only `to_bytes`/`from_bytes` are on the hot path (the bit source
    of scalar multiplication, scalar.rs:284-296); arithmetic is plain Python integers mod r."""

    __slots__ = ("value",)

    def __init__(self, value=0):
        self.value = int(value) % R_ORDER

    @staticmethod
    def zero(): return Scalar(0)
    @staticmethod
    def one(): return Scalar(1)

    @staticmethod
    def from_bytes(b):
        """scalar.rs:256-280: 32 little-endian bytes, None (CtOption none) if not canonical."""
        v = int.from_bytes(bytes(b), "little")
        return Scalar(v) if v < R_ORDER else None

    @staticmethod
    def from_bytes_wide(b):
        """scalar.rs:300-331: 64 little-endian bytes reduced mod r."""
        return Scalar(int.from_bytes(bytes(b), "little"))

    def to_bytes(self): return self.value.to_bytes(32, "little")
    def __add__(self, o): return Scalar(self.value + o.value)
    def __sub__(self, o): return Scalar(self.value - o.value)
    def __neg__(self): return Scalar(-self.value)
    def __mul__(self, o):
        if isinstance(o, Scalar):
            return Scalar(self.value * o.value)
        return NotImplemented
    def invert(self): return None if self.value == 0 else Scalar(pow(self.value, -1, R_ORDER))
    def __eq__(self, o): return isinstance(o, Scalar) and self.value == o.value
    def __hash__(self): return hash(self.value)
    def __repr__(self): return f"Scalar(0x{self.value:064x})"


def _fp_bytes(l):
    return limbs_to_fp(l).to_bytes(48, "big")


def _lex_largest(v):
    return v > (P - 1) // 2


class _Group:
    """shared plumbing of the four point types; G = 1 or 2, W = u64 words per coordinate"""
    G = 1
    W = 6


class G1Affine(_Group):
    """src/g1.rs:28-32.  `xy` = 12 uint64 Montgomery limbs (x | y), `infinity` bool."""
    G, W = 1, 6

    def __init__(self, xy, infinity=False):
        self.xy = _u64(xy, (2 * self.W,)).copy()
        self.infinity = bool(infinity)

    @classmethod
    def identity(cls):
        one = fp_to_limbs(1)
        z = np.zeros(6, dtype=np.uint64)
        if cls.G == 1:
            return cls(np.concatenate([z, one]), True)
        return cls(np.concatenate([z, z, one, z]), True)

    @classmethod
    def generator(cls):
        if cls.G == 1:
            return cls(np.concatenate([fp_to_limbs(_G1X), fp_to_limbs(_G1Y)]))
        return cls(np.concatenate([fp_to_limbs(_G2X[0]), fp_to_limbs(_G2X[1]), fp_to_limbs(_G2Y[0]), fp_to_limbs(_G2Y[1])]))

    def is_identity(self): return self.infinity

    def __neg__(self):
        if self.infinity:
            # g1.rs:121-130: conditional_select(&-y, &Fp::one(), infinity) -- the identity keeps its (0, 1, inf) limbs
            return type(self)(self.xy.copy(), True)
        xy = self.xy.copy()
        for k in range(self.W // 6):
            off = self.W + 6 * k
            xy[off:off + 6] = fp_to_limbs((-limbs_to_fp(xy[off:off + 6])) % P)
        return type(self)(xy, self.infinity)

    def __eq__(self, o):
        return type(o) is type(self) and ((self.infinity and o.infinity) or
                                          (self.infinity == o.infinity and bool(np.array_equal(self.xy, o.xy))))

    def to_projective(self):
        one = fp_to_limbs(1)
        z = np.zeros(self.W, dtype=np.uint64)
        if not self.infinity:
            z[:6] = one
        return self._PROJ(np.concatenate([self.xy, z]))

    def __mul__(self, s):
        """`&G1Affine * &Scalar` (g1.rs:573-579): the batched variable-base kernel with one element."""
        ctx = default_context()
        out = ctx.mul_batch(self.G, self.xy[None, :], np.array([self.infinity], dtype=np.uint8), [s])
        return self._PROJ(out[0])

    @classmethod
    def mul_batch(cls, points, scalars):
        """points.iter().zip(scalars).map(|(p, s)| p * s) as ONE device call: [G1Projective] (`Mul` over slices)."""
        points = list(points)
        if not points:
            return []
        xy = np.stack([p.xy for p in points]); inf = np.array([p.infinity for p in points], dtype=np.uint8)
        out = default_context().mul_batch(cls.G, xy, inf, scalars)
        return [cls._PROJ(o) for o in out]

    # -- encodings (src/notes/serialization.rs; g1.rs:221-260, g2.rs:254-299) --
    def to_uncompressed(self):
        if self.G == 1:
            res = bytearray((b"\0" * 96) if self.infinity else _fp_bytes(self.xy[0:6]) + _fp_bytes(self.xy[6:12]))
        else:
            res = bytearray((b"\0" * 192) if self.infinity else
                            _fp_bytes(self.xy[6:12]) + _fp_bytes(self.xy[0:6]) + _fp_bytes(self.xy[18:24]) + _fp_bytes(self.xy[12:18]))
        if self.infinity:
            res[0] |= 1 << 6
        return bytes(res)

    def to_compressed(self):
        if self.G == 1:
            res = bytearray((b"\0" * 48) if self.infinity else _fp_bytes(self.xy[0:6]))
            big = (not self.infinity) and _lex_largest(limbs_to_fp(self.xy[6:12]))
        else:
            res = bytearray((b"\0" * 96) if self.infinity else _fp_bytes(self.xy[6:12]) + _fp_bytes(self.xy[0:6]))
            y0, y1 = limbs_to_fp(self.xy[12:18]), limbs_to_fp(self.xy[18:24])
            big = (not self.infinity) and (_lex_largest(y1) or (y1 == 0 and _lex_largest(y0)))
        res[0] |= 1 << 7
        if self.infinity:
            res[0] |= 1 << 6
        if big:
            res[0] |= 1 << 5
        return bytes(res)

    @classmethod
    def from_uncompressed_unchecked(cls, b):
        """g1.rs:273-322 / g2.rs:311-380; None where the reference returns CtOption::none."""
        b = bytes(b)
        n = 2 * cls.W // 6
        if len(b) != 48 * n:
            return None
        c, i, s = (b[0] >> 7) & 1, (b[0] >> 6) & 1, (b[0] >> 5) & 1
        vals = [int.from_bytes((bytes([b[0] & 0x1F]) + b[1:48]) if k == 0 else b[48 * k:48 * k + 48], "big") for k in range(n)]
        if any(v >= P for v in vals) or c or s or (i and any(vals)):
            return None
        if i:
            return cls.identity()
        if cls.G == 1:
            return cls(np.concatenate([fp_to_limbs(vals[0]), fp_to_limbs(vals[1])]))
        return cls(np.concatenate([fp_to_limbs(vals[1]), fp_to_limbs(vals[0]), fp_to_limbs(vals[3]), fp_to_limbs(vals[2])]))

    @classmethod
    def _decode(cls, b, compressed, checked):
        b = bytes(b)
        if len(b) != (48 if compressed else 96) * cls.G:
            return None                  # the reference takes a fixed-size array; any other input has no decoding (CtOption none)
        xy, inf, ok = default_context().points_from_bytes(cls.G, b, compressed, checked)
        return cls(xy[0], bool(inf[0])) if ok[0] else None

    @classmethod
    def from_compressed(cls, b):
        """g1.rs:326-332 / g2.rs:390-395 (decompression + subgroup check on the GPU); None = CtOption::none."""
        return cls._decode(b, True, True)

    @classmethod
    def from_compressed_unchecked(cls, b): return cls._decode(b, True, False)

    @classmethod
    def from_uncompressed(cls, b):
        """g1.rs:264-267 (on-curve + subgroup check on the GPU)."""
        return cls._decode(b, False, True)

    def __repr__(self):
        return f"{type(self).__name__}({'identity' if self.infinity else self.to_compressed().hex()})"


class G1Projective(_Group):
    """src/g1.rs:442-446.  `xyz` = 18 uint64 Montgomery limbs (X | Y | Z), x = X/Z; identity (0:1:0)."""
    G, W = 1, 6
    _AFF = G1Affine

    def __init__(self, xyz):
        self.xyz = _u64(xyz, (3 * self.W,)).copy()

    @classmethod
    def identity(cls): return cls._AFF.identity().to_projective()
    @classmethod
    def generator(cls): return cls._AFF.generator().to_projective()

    def is_identity(self):
        return not self.xyz[2 * self.W:].any()

    def to_affine(self):
        """`G1Affine::from(&G1Projective)` (g1.rs:49-63)."""
        xy, inf = default_context().batch_normalize(self.G, self.xyz[None, :])
        return self._AFF(xy[0], bool(inf[0]))

    @classmethod
    def batch_normalize(cls, points):
        """`G1Projective::batch_normalize` (g1.rs:806-839)."""
        if not points:
            return []
        xy, inf = default_context().batch_normalize(cls.G, np.stack([p.xyz for p in points]))
        return [cls._AFF(xy[i], bool(inf[i])) for i in range(len(points))]

    def __add__(self, o):
        ctx = default_context()
        if isinstance(o, self._AFF):
            return type(self)(ctx.point_op(self.G, 2, self.xyz[None, :], o.xy[None, :], np.array([o.infinity], dtype=np.uint8))[0])
        return type(self)(ctx.point_op(self.G, 0, self.xyz[None, :], o.xyz[None, :])[0])

    def __neg__(self):
        a = self.xyz.copy()
        for k in range(self.W // 6):
            off = self.W + 6 * k
            a[off:off + 6] = fp_to_limbs((-limbs_to_fp(a[off:off + 6])) % P)
        return type(self)(a)

    def __sub__(self, o): return self + (-o)
    def double(self): return type(self)(default_context().point_op(self.G, 1, self.xyz[None, :])[0])

    def __mul__(self, s):
        """`&G1Projective * &Scalar` (g1.rs:556-562)."""
        return self.to_affine() * s

    @classmethod
    def sum(cls, points):
        """`Sum for G1Projective` (g1.rs:161-171)."""
        pts = list(points)
        if not pts:
            return cls.identity()
        return cls(default_context().point_sum(cls.G, np.stack([p.xyz for p in pts])))

    @classmethod
    def hash_to_curve(cls, msg, dst):
        """`<G as HashToCurve<ExpandMsgXmd<Sha256>>>::hash_to_curve(msg, dst)` (hash_to_curve/mod.rs:86-92)"""
        return cls(default_context().hash_to_curve(cls.G, [msg], dst)[0])

    @classmethod
    def encode_to_curve(cls, msg, dst):
        """`...::encode_to_curve(msg, dst)` (hash_to_curve/mod.rs:103-108)"""
        return cls(default_context().hash_to_curve(cls.G, [msg], dst, encode_only=True)[0])

    @classmethod
    def hash_to_curve_batch(cls, msgs, dst, encode_only=False):
        out = default_context().hash_to_curve(cls.G, msgs, dst, encode_only=encode_only)
        return [cls(row) for row in out]

    def __eq__(self, o):
        """projective equality (g1.rs:479-496) decided on canonical affine forms"""
        return type(o) is type(self) and self.to_affine() == o.to_affine()

    def __repr__(self): return f"{type(self).__name__}({self.to_affine()!r})"


class G2Affine(G1Affine):
    """src/g2.rs.  `xy` = 24 uint64 limbs (x.c0 x.c1 y.c0 y.c1)."""
    G, W = 2, 12


class G2Projective(G1Projective):
    G, W = 2, 12
    _AFF = G2Affine


G1Affine._PROJ = G1Projective
G2Affine._PROJ = G2Projective


class Gt:
    """Target group element (src/pairings.rs:204-337), written additively like the reference: `+` is the
    Fp12 product, `-x` the conjugate.  `f` = 72 uint64 limbs in struct order."""

    def __init__(self, f):
        self.f = _u64(f, (72,)).copy()

    @staticmethod
    def identity():
        f = np.zeros(72, dtype=np.uint64)
        f[:6] = fp_to_limbs(1)
        return Gt(f)

    @staticmethod
    def generator():
        """pairings.rs:359-475: e(G1::generator, G2::generator)."""
        return pairing(G1Affine.generator(), G2Affine.generator())

    def __add__(self, o): return Gt(default_context().fp12_op(0, self.f[None, :], o.f[None, :])[0])
    def __neg__(self): return Gt(default_context().fp12_op(8, self.f[None, :])[0])
    def __sub__(self, o): return self + (-o)
    def double(self): return Gt(default_context().fp12_op(3, self.f[None, :])[0])

    def __mul__(self, s):
        """`&Gt * &Scalar` (pairings.rs:297-322): double-and-add over the 255 low bits."""
        v = s.value if isinstance(s, Scalar) else int(s) % R_ORDER
        return Gt(default_context().gt_mul_scalar_batch(self.f[None, :], [v])[0])

    @staticmethod
    def sum(items):
        items = list(items)
        if not items:
            return Gt.identity()
        return Gt(default_context().fp12_product(np.stack([g.f for g in items])))

    def __eq__(self, o): return isinstance(o, Gt) and bool(np.array_equal(self.f, o.f))
    def __repr__(self): return f"Gt({self.f[:2]}...)"


class MillerLoopResult:
    """src/pairings.rs:26.  Deliberately has no equality, like the reference (:21-26)."""

    def __init__(self, f):
        self.f = _u64(f, (72,)).copy()

    @staticmethod
    def default(): return MillerLoopResult(Gt.identity().f)

    def final_exponentiation(self):
        """pairings.rs:48-176."""
        return Gt(default_context().final_exponentiation_batch(self.f[None, :])[0])

    def __add__(self, o):
        """pairings.rs:179-186: product of the underlying Fp12 values."""
        return MillerLoopResult(default_context().fp12_op(0, self.f[None, :], o.f[None, :])[0])


class G2Prepared:
    """src/pairings.rs:487-546.  Opaque in the reference (private fields).  `G2Prepared(q)` keeps the affine point (its lines are
    then computed on the fly in every Miller loop, as in rounds 1-4); `G2Prepared.resident(q)` / `G2Prepared.resident_many(points)`
    do what `From<G2Affine>` does in the reference: the 68 coefficient triples are computed ONCE, into a device-resident table
    (`blsgpu_g2_prepare`), and every later `multi_miller_loop` only evaluates them."""

    def __init__(self, q, table=None, index=None):
        if not isinstance(q, G2Affine):
            raise TypeError("G2Prepared::from expects a G2Affine")
        self.q, self.table, self.index = q, table, index

    @classmethod
    def resident_many(cls, points):
        points = list(points)
        xy = np.stack([q.xy for q in points]) if points else np.zeros((0, 24), dtype=np.uint64)
        table = default_context().g2_prepare(xy, np.array([q.infinity for q in points], dtype=np.uint8))
        return [cls(q, table, i) for i, q in enumerate(points)]

    @classmethod
    def resident(cls, q):
        return cls.resident_many([q])[0]

    def coeffs(self):
        """(infinity, (68, 3, 12) u64) of a resident value"""
        if self.table is None:
            raise ValueError("G2Prepared.coeffs: not resident (use G2Prepared.resident)")
        return self.table.coeffs(self.index)


def pairing(p, q):
    """`pairing(&G1Affine, &G2Affine) -> Gt` (src/pairings.rs:607-653)."""
    out = default_context().pairing_batch(p.xy[None, :], np.array([p.infinity], dtype=np.uint8), q.xy[None, :],
                                          np.array([q.infinity], dtype=np.uint8))
    return Gt(out[0])


def multi_miller_loop(terms):
    """`multi_miller_loop(&[(&G1Affine, &G2Prepared)]) -> MillerLoopResult` (src/pairings.rs:554-603)."""
    terms = list(terms)
    n = len(terms)
    g1 = np.zeros((n, 12), dtype=np.uint64)
    g2 = np.zeros((n, 24), dtype=np.uint64)
    f1 = np.zeros(n, dtype=np.uint8)
    f2 = np.zeros(n, dtype=np.uint8)
    for i, (p, prep) in enumerate(terms):
        g1[i], f1[i], g2[i], f2[i] = p.xy, p.infinity, prep.q.xy, prep.q.infinity
    table, qi = _resident_indices([prep for _, prep in terms])
    if table is not None:
        return MillerLoopResult(default_context().multi_miller_loop_prepared(g1, f1, table, qi, g2, f2))
    return MillerLoopResult(default_context().multi_miller_loop(g1, f1, g2, f2))


def _resident_indices(preps):
    """the table shared by the resident `G2Prepared` values among `preps` and the per-term indices (terms of another table, or not
    resident, are UNPREPARED: their lines are computed on the fly from the affine point they also hold); (None, None) if none is resident"""
    table = next((p.table for p in preps if p.table is not None), None)
    if table is None:
        return None, None
    return table, np.array([p.index if p.table is table else UNPREPARED for p in preps], dtype=np.uint32)


def multi_miller_loop_many(equations, final_exp=True):
    """One `multi_miller_loop` per equation (a list of lists of `(&G1Affine, &G2Prepared)` terms) in ONE device call: the bulk
    form of the pattern `E::multi_miller_loop(&terms).final_exponentiation()` of signature verification (pairings.rs:554-603,
    817-824).  Returns a list of `Gt` (final_exp) or `MillerLoopResult`."""
    eqs = [list(e) for e in equations]
    n = sum(len(e) for e in eqs)
    g1 = np.zeros((n, 12), dtype=np.uint64)
    g2 = np.zeros((n, 24), dtype=np.uint64)
    f1 = np.zeros(n, dtype=np.uint8)
    f2 = np.zeros(n, dtype=np.uint8)
    off = np.zeros(len(eqs) + 1, dtype=np.uint64)
    i = 0
    for s, e in enumerate(eqs):
        for p, prep in e:
            g1[i], f1[i], g2[i], f2[i] = p.xy, p.infinity, prep.q.xy, prep.q.infinity
            i += 1
        off[s + 1] = i
    table, qi = _resident_indices([prep for e in eqs for _, prep in e])
    if table is not None:
        out = default_context().multi_miller_loop_prepared_many(g1, f1, table, qi, off, g2, f2, final_exp)
    else:
        out = default_context().multi_miller_loop_many(g1, f1, g2, f2, off, final_exp)
    return [Gt(v) if final_exp else MillerLoopResult(v) for v in out]


def _msm(group, bases, scalars):
    ctx = default_context()
    proj = G1Projective if group == 1 else G2Projective
    if isinstance(bases, ResidentBases):
        return proj(ctx.msm(bases, scalars))
    bases = list(bases)
    if len(bases) != len(scalars):
        raise ValueError("bases and scalars differ in length")
    w = 12 if group == 1 else 24
    xy = np.stack([b.xy for b in bases]) if bases else np.zeros((0, w), dtype=np.uint64)
    inf = np.array([b.infinity for b in bases], dtype=np.uint8)
    return proj(ctx.msm_host(group, xy, inf, scalars))


def msm_g1(bases, scalars):
    """bases.iter().zip(scalars).map(|(p, s)| p * s).sum::<G1Projective>()"""
    return _msm(1, bases, scalars)


def msm_g2(bases, scalars):
    return _msm(2, bases, scalars)


class Bls12:
    """`pairing::Engine` + `MultiMillerLoop` for BLS12-381 (src/pairings.rs:790-824)."""
    Fr, G1, G1Affine, G2, G2Affine, Gt, G2Prepared, Result = Scalar, G1Projective, G1Affine, G2Projective, G2Affine, Gt, G2Prepared, MillerLoopResult

    @staticmethod
    def pairing(p, q): return pairing(p, q)
    @staticmethod
    def multi_miller_loop(terms): return multi_miller_loop(terms)
