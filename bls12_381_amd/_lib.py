"""ctypes binding of libblsgpu.so (the C ABI in include/bls12_381_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded, importing a
compute entry point raises.  (`oracle/` is test infrastructure and is never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLSGPU_LIB_PATH: A/B experiments load an alternative build of the SAME library (tools/ab_pairing.py); never a fallback
LIB_PATH = os.environ.get("BLSGPU_LIB_PATH") or os.path.join(_HERE, "libblsgpu.so")

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_int = ctypes.c_int

# name -> (restype, argtypes); must list every symbol declared in include/bls12_381_hip.h
SIGNATURES = {
    "blsgpu_create": (c_int, [c_int, ctypes.POINTER(c_vp)]),
    "blsgpu_destroy": (None, [c_vp]),
    "blsgpu_last_error": (ctypes.c_char_p, []),
    "blsgpu_device_count": (c_int, []),
    "blsgpu_set_stream": (c_int, [c_vp, c_vp]),
    "blsgpu_synchronize": (c_int, [c_vp]),
    "blsgpu_set_pipelining": (c_int, [c_vp, c_int]),
    "blsgpu_join": (c_int, [c_vp]),
    "blsgpu_join_lag": (c_int, [c_vp, c_int]),
    "blsgpu_g1_bases_upload": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_g2_bases_upload": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_g1_bases_from_device": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_g2_bases_from_device": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_bases_from_scalars": (c_int, [c_vp, c_int, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_bases_precompute": (c_int, [c_vp, c_vp, c_int]),
    "blsgpu_bases_len": (c_sz, [c_vp]),
    "blsgpu_set_assume_subgroup": (c_int, [c_vp, c_int]),
    "blsgpu_bases_subgroup_state": (c_int, [c_vp]),
    "blsgpu_pairing_layout": (c_int, [c_vp, c_sz]),
    "blsgpu_bases_download": (c_int, [c_vp, c_vp, c_sz, c_sz, c_vp, c_vp]),
    "blsgpu_bases_free": (None, [c_vp]),
    "blsgpu_g1_msm": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g1_msm_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_set_bases_cache": (c_int, [c_vp, c_int]),
    "blsgpu_set_bases_cache_verify": (c_int, [c_vp, c_int]),
    "blsgpu_set_scalar_form": (c_int, [c_vp, c_int]),
    "blsgpu_g1_msm_mont": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm_mont": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g1_msm_mont_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm_mont_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g1_mul_batch_mont": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_mul_batch_mont": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_mul_batch_mont_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_mul_batch_mont_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fr_to_bytes": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_fr_from_bytes": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_fr_from_bytes_wide": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fr_to_bytes_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_fr_from_bytes_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_fr_from_bytes_wide_device": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_msm_host": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm_host": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fr_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_fr_op_device": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_fr_ntt": (c_int, [c_vp, c_vp, c_int, c_int]),
    "blsgpu_fr_ntt_device": (c_int, [c_vp, c_vp, c_int, c_int]),
    "blsgpu_g1_hash_to_curve_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_g2_hash_to_curve_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_hash_to_curve_device": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_hash_to_curve_expander_batch": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_hash_to_curve_expander_device": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_hash_to_curve_from_uniform_batch": (c_int, [c_vp, c_int, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_hash_to_curve_from_uniform_device": (c_int, [c_vp, c_int, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_expand_message_batch": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_expand_message_device": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_hash_to_scalar_batch": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_hash_to_scalar_device": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_gt_mul_scalar_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_msm_bytes": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm_bytes": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_msm_accumulate_stats": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint)]),
    "blsgpu_g1_msm_many": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_g2_msm_many": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_g1_msm_many_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_g2_msm_many_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_sz, c_sz, c_vp]),
    "blsgpu_set_msm_window": (c_int, [c_vp, c_int]),
    "blsgpu_g1_mul_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_mul_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_mul_batch_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_mul_batch_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_sum": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_sum": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_sum_device": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_sum_device": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_batch_normalize": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_g2_batch_normalize": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_g1_from_bytes_batch": (c_int, [c_vp, c_vp, c_sz, c_int, c_int, c_vp, c_vp, c_vp]),
    "blsgpu_g2_from_bytes_batch": (c_int, [c_vp, c_vp, c_sz, c_int, c_int, c_vp, c_vp, c_vp]),
    "blsgpu_g1_to_bytes_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_g2_to_bytes_batch": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_pairing_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_miller_loop_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_multi_miller_loop": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_final_exponentiation_batch": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_multi_miller_loop_many": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_multi_miller_loop_many_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_sz, c_int, c_vp]),
    "blsgpu_wide_status": (ctypes.c_char_p, [c_vp]),
    "blsgpu_g1_batch_normalize_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_g2_batch_normalize_device": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_g1_from_bytes_batch_device": (c_int, [c_vp, c_vp, c_sz, c_int, c_int, c_vp, c_vp, c_vp]),
    "blsgpu_g2_from_bytes_batch_device": (c_int, [c_vp, c_vp, c_sz, c_int, c_int, c_vp, c_vp, c_vp]),
    "blsgpu_g1_to_bytes_batch_device": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_g2_to_bytes_batch_device": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_gt_mul_scalar_batch_device": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_gt_is_identity_device": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_bls_verify_batch": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_bls_verify_batch_device": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "blsgpu_g2_prepare": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_g2_prepare_device": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_g2_prepared_len": (c_sz, [c_vp]),
    "blsgpu_g2_prepared_free": (None, [c_vp]),
    "blsgpu_g2_prepared_coeffs": (c_int, [c_vp, c_vp, c_sz, c_vp, c_vp]),
    "blsgpu_multi_miller_loop_prepared": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_multi_miller_loop_prepared_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_multi_miller_loop_prepared_many": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_multi_miller_loop_prepared_many_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_sz, c_sz, c_int, c_vp]),
    "blsgpu_group_create": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp)]),
    "blsgpu_group_destroy": (None, [c_vp]),
    "blsgpu_group_size": (c_int, [c_vp]),
    "blsgpu_group_ctx": (c_vp, [c_vp, c_int]),
    "blsgpu_group_bases_upload": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_group_bases_from_scalars": (c_int, [c_vp, c_int, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_group_bases_len": (c_sz, [c_vp]),
    "blsgpu_group_bases_free": (None, [c_vp]),
    "blsgpu_g1_msm_sharded": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g2_msm_sharded": (c_int, [c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_g1_msm_sharded_device": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "blsgpu_g2_msm_sharded_device": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "blsgpu_g1_partials_fold": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "blsgpu_g2_partials_fold": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "blsgpu_g1_partials_fold_device": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "blsgpu_g2_partials_fold_device": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "blsgpu_pairings_sharded_device": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "blsgpu_fp12_partials_fold_device": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "blsgpu_group_set_pipelining": (c_int, [c_vp, c_int]),
    "blsgpu_group_synchronize": (c_int, [c_vp]),
    "blsgpu_group_g2_prepare": (c_int, [c_vp, c_vp, c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "blsgpu_group_g2_prepared_len": (c_sz, [c_vp]),
    "blsgpu_group_g2_prepared_free": (None, [c_vp]),
    "blsgpu_multi_miller_loop_prepared_sharded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_multi_miller_loop_prepared_many_sharded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_pairing_batch_sharded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_miller_loop_batch_sharded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_multi_miller_loop_sharded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_multi_miller_loop_many_sharded": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "blsgpu_fp12_product": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_pairing_batch_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_multi_miller_loop_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_miller_loop_batch_device": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_final_exponentiation_device": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fp12_product_device": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fp_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fp2_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fp6_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fp12_op": (c_int, [c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_point_op": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "blsgpu_fp_mul_throughput": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double)]),
    "blsgpu_mad_throughput": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double)]),
    "blsgpu_last_msm_phase_ms": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_float)]),
    "blsgpu_set_profiling": (c_int, [c_vp, c_int]),
    "blsgpu_kernel_timing": (c_int, [c_vp, c_int]),
    "blsgpu_kernel_timing_report": (c_int, [c_vp, ctypes.c_char_p, c_sz, ctypes.POINTER(c_sz)]),
}

_lib = None


class BlsGpuError(RuntimeError):
    pass


def load():
    """Load libblsgpu.so (once) and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BlsGpuError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the bls12_381_amd compute path.")
    # PyTorch-ROCm bundles its own HIP runtime; a process that uses both must load torch's copy first, otherwise torch finds
    # "No HIP GPUs" after libblsgpu.so has pulled in /opt/rocm's.  torch is optional plumbing (device tensors, streams,
    # torch.distributed): if it is installed and not yet imported, import it before the library (BLSGPU_NO_TORCH_PRELOAD=1 skips this).
    import sys
    if "torch" not in sys.modules and not os.environ.get("BLSGPU_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted from the header
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().blsgpu_last_error()
        raise BlsGpuError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
