"""Seeded synthetic inputs for benchmarks and full-size tests.

Scalars are drawn exactly as the survey prescribes: a SplitMix64 stream, 32 random bytes per candidate, top bit
cleared, candidates >= r rejected -- i.e. uniform in [0, r), all 255 bits in play (the top window and its carry
included).  The generator is vectorised with numpy and produces the SAME sequence as the scalar-at-a-time
`oracle.bls12_381_ref.SplitMix64(seed).scalar()` (a CPU test pins that), so small prefixes can be cross-checked
against the oracle while 2^24-element vectors still take well under a second.

Nothing here touches the GPU or the oracle: it only makes bytes.
"""
import numpy as np

R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
SEED = 0xB1512381

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_R_WORDS = np.frombuffer(R_ORDER.to_bytes(32, "little"), dtype="<u8")


def splitmix64(seed, first, count):
    """outputs number first+1 .. first+count of SplitMix64(seed) as a uint64 array (output k = mix(seed + k*gamma))"""
    with np.errstate(over="ignore"):
        k = np.arange(first + 1, first + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _lt_r(w):
    lt = np.zeros(len(w), dtype=bool)
    eq = np.ones(len(w), dtype=bool)
    for k in (3, 2, 1, 0):
        lt |= eq & (w[:, k] < _R_WORDS[k])
        eq &= w[:, k] == _R_WORDS[k]
    return lt


def scalars(n, seed=SEED):
    """n scalars uniform in [0, r) as an (n, 32) uint8 array of little-endian canonical bytes (Scalar::to_bytes format)."""
    out = np.zeros((n, 4), dtype="<u8")
    have, used = 0, 0
    while have < n:
        m = max(1024, int((n - have) * 1.12) + 64)            # acceptance rate r / 2^255 = 0.906
        w = splitmix64(seed, used, 4 * m).reshape(m, 4)
        w[:, 3] &= np.uint64(0x7FFFFFFFFFFFFFFF)
        ok = np.flatnonzero(_lt_r(w))
        take = ok[:n - have]
        out[have:have + len(take)] = w[take]
        have += len(take)
        # the stream position after the last candidate that was looked at (accepted or rejected)
        used += 4 * (int(take[-1]) + 1 if have >= n and len(take) else m)
    return out.view(np.uint8).reshape(n, 32)


def to_ints(sb):
    """(n, 32) uint8 little-endian -> list of Python ints"""
    sb = np.ascontiguousarray(sb)
    return [int.from_bytes(sb[i].tobytes(), "little") for i in range(sb.shape[0])]


def dot_mod_r(a_bytes, b_bytes):
    """sum_i a_i * b_i mod r for two (n, 32) byte arrays: the discrete-log side of  MSM(s, [k_i]G) = [sum s_i k_i]G.
    Python big integers, chunked so that 2^24 terms stay in the seconds range."""
    n = a_bytes.shape[0]
    tot = 0
    step = 1 << 16
    for lo in range(0, n, step):
        a = to_ints(a_bytes[lo:lo + step]); b = to_ints(b_bytes[lo:lo + step])
        tot += sum(x * y for x, y in zip(a, b))
    return tot % R_ORDER
