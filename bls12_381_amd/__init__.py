"""bls12_381_amd -- MI355X-native multi-scalar multiplication and batched pairings for BLS12-381.

Host-side mirror of the zkcrypto/bls12_381 types for the accelerated hot path (see api.py); all compute
runs in hand-written HIP kernels for gfx950 behind the C ABI of include/bls12_381_hip.h.
"""
from ._lib import BlsGpuError, LIB_PATH, load  # noqa: F401
from .api import (  # noqa: F401
    Context, Scalar, G1Affine, G1Projective, G2Affine, G2Projective, Gt, MillerLoopResult, G2Prepared, Bls12,
    ResidentBases, pairing, multi_miller_loop, multi_miller_loop_many, msm_g1, msm_g2, default_context, Group, GroupBases, PreparedG2Table, UNPREPARED,
)
