//! `extern "C"` declarations of libblsgpu.so -- GENERATED from include/bls12_381_hip.h by tools/gen_rust_ffi.py; do not edit.
//! One entry per symbol of the C ABI; the header documents each one and cites the reference lines it replaces.
#![allow(non_camel_case_types, dead_code)]
use core::ffi::{c_char, c_int, c_uint, c_void};

/// opaque: one device + its streams and scratch (blsgpu_create / blsgpu_destroy)
#[repr(C)] pub struct BlsgpuCtx { _private: [u8; 0] }
/// opaque: bases resident in HBM (blsgpu_g1_bases_upload & co. / blsgpu_bases_free)
#[repr(C)] pub struct BlsgpuBases { _private: [u8; 0] }
/// opaque: one context per listed device (blsgpu_group_create / blsgpu_group_destroy)
#[repr(C)] pub struct BlsgpuGroup { _private: [u8; 0] }
/// opaque: bases sharded over the members of a group (blsgpu_group_bases_upload / blsgpu_group_bases_free)
#[repr(C)] pub struct BlsgpuGroupBases { _private: [u8; 0] }
/// opaque: `G2Prepared` values resident in HBM (blsgpu_g2_prepare / blsgpu_g2_prepared_free)
#[repr(C)] pub struct BlsgpuG2Prepared { _private: [u8; 0] }
/// opaque: the same `G2Prepared` table on every member of a group (blsgpu_group_g2_prepare / blsgpu_group_g2_prepared_free)
#[repr(C)] pub struct BlsgpuGroupG2Prepared { _private: [u8; 0] }

pub const BLSGPU_OK: c_int = 0;
/// scalar arguments hold `Scalar::to_bytes()` output (default) / the Montgomery limbs of `Scalar([u64; 4])` (blsgpu_set_scalar_form)
pub const BLSGPU_SCALAR_BYTES: c_int = 0;
pub const BLSGPU_SCALAR_MONT: c_int = 1;

#[link(name = "blsgpu")]
extern "C" {
    pub fn blsgpu_create(device: c_int, out: *mut *mut BlsgpuCtx) -> c_int;
    pub fn blsgpu_destroy(ctx: *mut BlsgpuCtx);
    pub fn blsgpu_last_error() -> *const c_char;
    pub fn blsgpu_device_count() -> c_int;
    pub fn blsgpu_set_stream(ctx: *mut BlsgpuCtx, hip_stream: *mut c_void) -> c_int;
    pub fn blsgpu_synchronize(ctx: *mut BlsgpuCtx) -> c_int;
    pub fn blsgpu_set_pipelining(ctx: *mut BlsgpuCtx, enabled: c_int) -> c_int;
    pub fn blsgpu_join(ctx: *mut BlsgpuCtx) -> c_int;
    pub fn blsgpu_join_lag(ctx: *mut BlsgpuCtx, lag: c_int) -> c_int;
    pub fn blsgpu_g1_bases_upload(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, n: usize, out: *mut *mut BlsgpuBases) -> c_int;
    pub fn blsgpu_g2_bases_upload(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, n: usize, out: *mut *mut BlsgpuBases) -> c_int;
    pub fn blsgpu_g1_bases_from_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, n: usize, out: *mut *mut BlsgpuBases) -> c_int;
    pub fn blsgpu_g2_bases_from_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, n: usize, out: *mut *mut BlsgpuBases) -> c_int;
    pub fn blsgpu_bases_from_scalars(ctx: *mut BlsgpuCtx, group: c_int, scalars: *const u8, n: usize, out: *mut *mut BlsgpuBases) -> c_int;
    pub fn blsgpu_bases_precompute(ctx: *mut BlsgpuCtx, b: *mut BlsgpuBases, window_bits: c_int) -> c_int;
    pub fn blsgpu_bases_len(b: *const BlsgpuBases) -> usize;
    pub fn blsgpu_set_assume_subgroup(ctx: *mut BlsgpuCtx, enabled: c_int) -> c_int;
    pub fn blsgpu_bases_subgroup_state(b: *const BlsgpuBases) -> c_int;
    pub fn blsgpu_bases_download(ctx: *mut BlsgpuCtx, b: *const BlsgpuBases, first: usize, count: usize, xy: *mut u64, infinity: *mut u8) -> c_int;
    pub fn blsgpu_bases_free(b: *mut BlsgpuBases);
    pub fn blsgpu_g1_msm(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_msm(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_msm_device(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_msm_device(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_set_scalar_form(ctx: *mut BlsgpuCtx, form: c_int) -> c_int;
    pub fn blsgpu_g1_msm_mont(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_msm_mont(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_msm_mont_device(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_msm_mont_device(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g1_msm_many(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, scalars: *const u8, n: usize, k: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_msm_many(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, scalars: *const u8, n: usize, k: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_msm_many_device(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, d_scalars: *const c_void, n: usize, k: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_msm_many_device(ctx: *mut BlsgpuCtx, bases: *const BlsgpuBases, first: usize, d_scalars: *const c_void, n: usize, k: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_set_bases_cache(ctx: *mut BlsgpuCtx, entries: c_int) -> c_int;
    pub fn blsgpu_set_bases_cache_verify(ctx: *mut BlsgpuCtx, enabled: c_int) -> c_int;
    pub fn blsgpu_g1_msm_host(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_msm_host(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_msm_bytes(ctx: *mut BlsgpuCtx, bases_uncompressed: *const u8, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn blsgpu_g2_msm_bytes(ctx: *mut BlsgpuCtx, bases_uncompressed: *const u8, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn blsgpu_set_msm_window(ctx: *mut BlsgpuCtx, c: c_int) -> c_int;
    pub fn blsgpu_g1_mul_batch(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_mul_batch(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_mul_batch_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_mul_batch_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g1_mul_batch_mont(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_mul_batch_mont(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, scalars: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_mul_batch_mont_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_mul_batch_mont_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, d_scalars: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g1_sum(ctx: *mut BlsgpuCtx, xyz: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_sum(ctx: *mut BlsgpuCtx, xyz: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_sum_device(ctx: *mut BlsgpuCtx, d_xyz: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_sum_device(ctx: *mut BlsgpuCtx, d_xyz: *const c_void, n: usize, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g1_batch_normalize(ctx: *mut BlsgpuCtx, xyz: *const u64, n: usize, xy: *mut u64, infinity: *mut u8) -> c_int;
    pub fn blsgpu_g2_batch_normalize(ctx: *mut BlsgpuCtx, xyz: *const u64, n: usize, xy: *mut u64, infinity: *mut u8) -> c_int;
    pub fn blsgpu_g1_from_bytes_batch(ctx: *mut BlsgpuCtx, bytes: *const u8, n: usize, compressed: c_int, checked: c_int, xy: *mut u64, infinity: *mut u8, ok: *mut u8) -> c_int;
    pub fn blsgpu_g2_from_bytes_batch(ctx: *mut BlsgpuCtx, bytes: *const u8, n: usize, compressed: c_int, checked: c_int, xy: *mut u64, infinity: *mut u8, ok: *mut u8) -> c_int;
    pub fn blsgpu_g1_to_bytes_batch(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, n: usize, compressed: c_int, out: *mut u8) -> c_int;
    pub fn blsgpu_g2_to_bytes_batch(ctx: *mut BlsgpuCtx, xy: *const u64, infinity: *const u8, n: usize, compressed: c_int, out: *mut u8) -> c_int;
    pub fn blsgpu_pairing_layout(ctx: *mut BlsgpuCtx, n: usize) -> c_int;
    pub fn blsgpu_wide_status(ctx: *mut BlsgpuCtx) -> *const c_char;
    pub fn blsgpu_pairing_batch(ctx: *mut BlsgpuCtx, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, n: usize, out_gt: *mut u64) -> c_int;
    pub fn blsgpu_miller_loop_batch(ctx: *mut BlsgpuCtx, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, n: usize, out_f: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop(ctx: *mut BlsgpuCtx, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, n: usize, out_f: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_many(ctx: *mut BlsgpuCtx, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, offsets: *const u64, nseg: usize, final_exp: c_int, out: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_many_device(ctx: *mut BlsgpuCtx, d_g1_xy: *const c_void, d_g1_inf: *const c_void, d_g2_xy: *const c_void, d_g2_inf: *const c_void, d_offsets: *const c_void, nseg: usize, total_terms: usize, max_seg_terms: usize, final_exp: c_int, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_final_exponentiation_batch(ctx: *mut BlsgpuCtx, in_f: *const u64, n: usize, out_gt: *mut u64) -> c_int;
    pub fn blsgpu_fp12_product(ctx: *mut BlsgpuCtx, in_f: *const u64, n: usize, out_f: *mut u64) -> c_int;
    pub fn blsgpu_gt_mul_scalar_batch(ctx: *mut BlsgpuCtx, gt: *const u64, scalars: *const u8, n: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_pairing_batch_device(ctx: *mut BlsgpuCtx, d_g1_xy: *const c_void, d_g1_inf: *const c_void, d_g2_xy: *const c_void, d_g2_inf: *const c_void, n: usize, d_out_gt: *mut c_void) -> c_int;
    pub fn blsgpu_multi_miller_loop_device(ctx: *mut BlsgpuCtx, d_g1_xy: *const c_void, d_g1_inf: *const c_void, d_g2_xy: *const c_void, d_g2_inf: *const c_void, n: usize, d_out_f: *mut c_void) -> c_int;
    pub fn blsgpu_miller_loop_batch_device(ctx: *mut BlsgpuCtx, d_g1_xy: *const c_void, d_g1_inf: *const c_void, d_g2_xy: *const c_void, d_g2_inf: *const c_void, n: usize, d_out_f: *mut c_void) -> c_int;
    pub fn blsgpu_final_exponentiation_device(ctx: *mut BlsgpuCtx, d_in_f: *const c_void, n: usize, d_out_gt: *mut c_void) -> c_int;
    pub fn blsgpu_fp12_product_device(ctx: *mut BlsgpuCtx, d_in_f: *const c_void, n: usize, d_out_f: *mut c_void) -> c_int;
    pub fn blsgpu_g2_prepare(ctx: *mut BlsgpuCtx, g2_xy: *const u64, g2_inf: *const u8, m: usize, out: *mut *mut BlsgpuG2Prepared) -> c_int;
    pub fn blsgpu_g2_prepare_device(ctx: *mut BlsgpuCtx, d_g2_xy: *const c_void, d_g2_inf: *const c_void, m: usize, out: *mut *mut BlsgpuG2Prepared) -> c_int;
    pub fn blsgpu_g2_prepared_len(p: *const BlsgpuG2Prepared) -> usize;
    pub fn blsgpu_g2_prepared_free(p: *mut BlsgpuG2Prepared);
    pub fn blsgpu_g2_prepared_coeffs(ctx: *mut BlsgpuCtx, p: *const BlsgpuG2Prepared, index: usize, out_coeffs: *mut u64, out_inf: *mut u8) -> c_int;
    pub fn blsgpu_multi_miller_loop_prepared(ctx: *mut BlsgpuCtx, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, q_index: *const u32, prepared: *const BlsgpuG2Prepared, n: usize, out_f: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_prepared_device(ctx: *mut BlsgpuCtx, d_g1_xy: *const c_void, d_g1_inf: *const c_void, d_g2_xy: *const c_void, d_g2_inf: *const c_void, d_q_index: *const c_void, prepared: *const BlsgpuG2Prepared, n: usize, d_out_f: *mut c_void) -> c_int;
    pub fn blsgpu_multi_miller_loop_prepared_many(ctx: *mut BlsgpuCtx, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, q_index: *const u32, prepared: *const BlsgpuG2Prepared, offsets: *const u64, nseg: usize, final_exp: c_int, out: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_prepared_many_device(ctx: *mut BlsgpuCtx, d_g1_xy: *const c_void, d_g1_inf: *const c_void, d_g2_xy: *const c_void, d_g2_inf: *const c_void, d_q_index: *const c_void, prepared: *const BlsgpuG2Prepared, d_offsets: *const c_void, nseg: usize, total_terms: usize, max_seg_terms: usize, final_exp: c_int, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_group_create(devices: *const c_int, ndev: c_int, out: *mut *mut BlsgpuGroup) -> c_int;
    pub fn blsgpu_group_destroy(group: *mut BlsgpuGroup);
    pub fn blsgpu_group_size(group: *const BlsgpuGroup) -> c_int;
    pub fn blsgpu_group_ctx(group: *mut BlsgpuGroup, member: c_int) -> *mut BlsgpuCtx;
    pub fn blsgpu_group_bases_upload(group: *mut BlsgpuGroup, group_id: c_int, xy: *const u64, infinity: *const u8, n: usize, out: *mut *mut BlsgpuGroupBases) -> c_int;
    pub fn blsgpu_group_bases_from_scalars(group: *mut BlsgpuGroup, group_id: c_int, scalars: *const u8, n: usize, out: *mut *mut BlsgpuGroupBases) -> c_int;
    pub fn blsgpu_group_bases_len(b: *const BlsgpuGroupBases) -> usize;
    pub fn blsgpu_group_bases_free(b: *mut BlsgpuGroupBases);
    pub fn blsgpu_g1_msm_sharded(group: *mut BlsgpuGroup, bases: *const BlsgpuGroupBases, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_msm_sharded(group: *mut BlsgpuGroup, bases: *const BlsgpuGroupBases, scalars: *const u8, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_msm_sharded_device(group: *mut BlsgpuGroup, bases: *const BlsgpuGroupBases, d_scalars: *const *const c_void, d_partials: *const *mut c_void) -> c_int;
    pub fn blsgpu_g2_msm_sharded_device(group: *mut BlsgpuGroup, bases: *const BlsgpuGroupBases, d_scalars: *const *const c_void, d_partials: *const *mut c_void) -> c_int;
    pub fn blsgpu_g1_partials_fold(group: *mut BlsgpuGroup, d_partials: *const *const c_void, lag: c_int, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_partials_fold(group: *mut BlsgpuGroup, d_partials: *const *const c_void, lag: c_int, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g1_partials_fold_device(group: *mut BlsgpuGroup, d_partials: *const *const c_void, lag: c_int, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_g2_partials_fold_device(group: *mut BlsgpuGroup, d_partials: *const *const c_void, lag: c_int, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_pairings_sharded_device(group: *mut BlsgpuGroup, mode: c_int, d_g1_xy: *const *const c_void, d_g1_inf: *const *const c_void, d_g2_xy: *const *const c_void, d_g2_inf: *const *const c_void, counts: *const usize, d_out: *const *mut c_void) -> c_int;
    pub fn blsgpu_fp12_partials_fold_device(group: *mut BlsgpuGroup, d_partials: *const *const c_void, final_exp: c_int, d_out_f: *mut c_void) -> c_int;
    pub fn blsgpu_group_set_pipelining(group: *mut BlsgpuGroup, enabled: c_int) -> c_int;
    pub fn blsgpu_group_synchronize(group: *mut BlsgpuGroup) -> c_int;
    pub fn blsgpu_pairing_batch_sharded(group: *mut BlsgpuGroup, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, n: usize, out_gt: *mut u64) -> c_int;
    pub fn blsgpu_miller_loop_batch_sharded(group: *mut BlsgpuGroup, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, n: usize, out_f: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_sharded(group: *mut BlsgpuGroup, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, n: usize, final_exp: c_int, out: *mut u64) -> c_int;
    pub fn blsgpu_group_g2_prepare(group: *mut BlsgpuGroup, g2_xy: *const u64, g2_inf: *const u8, m: usize, out: *mut *mut BlsgpuGroupG2Prepared) -> c_int;
    pub fn blsgpu_group_g2_prepared_len(p: *const BlsgpuGroupG2Prepared) -> usize;
    pub fn blsgpu_group_g2_prepared_free(p: *mut BlsgpuGroupG2Prepared);
    pub fn blsgpu_multi_miller_loop_prepared_sharded(group: *mut BlsgpuGroup, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, q_index: *const u32, prepared: *const BlsgpuGroupG2Prepared, n: usize, final_exp: c_int, out: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_prepared_many_sharded(group: *mut BlsgpuGroup, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, q_index: *const u32, prepared: *const BlsgpuGroupG2Prepared, offsets: *const u64, nseg: usize, final_exp: c_int, out: *mut u64) -> c_int;
    pub fn blsgpu_multi_miller_loop_many_sharded(group: *mut BlsgpuGroup, g1_xy: *const u64, g1_inf: *const u8, g2_xy: *const u64, g2_inf: *const u8, offsets: *const u64, nseg: usize, final_exp: c_int, out: *mut u64) -> c_int;
    pub fn blsgpu_fp_op(ctx: *mut BlsgpuCtx, op: c_int, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_fp2_op(ctx: *mut BlsgpuCtx, op: c_int, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_fp6_op(ctx: *mut BlsgpuCtx, op: c_int, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_fp12_op(ctx: *mut BlsgpuCtx, op: c_int, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_point_op(ctx: *mut BlsgpuCtx, group: c_int, op: c_int, a: *const u64, b: *const u64, b_inf: *const u8, n: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_fp_mul_throughput(ctx: *mut BlsgpuCtx, iters: c_int, muls_per_second: *mut f64) -> c_int;
    pub fn blsgpu_mad_throughput(ctx: *mut BlsgpuCtx, iters: c_int, mads_per_second: *mut f64) -> c_int;
    pub fn blsgpu_last_msm_phase_ms(ctx: *mut BlsgpuCtx, phase: c_int, ms: *mut f32) -> c_int;
    pub fn blsgpu_set_profiling(ctx: *mut BlsgpuCtx, enabled: c_int) -> c_int;
    pub fn blsgpu_msm_accumulate_stats(ctx: *mut BlsgpuCtx, enable: c_int, avg_ms: *mut f64, launches: *mut c_uint) -> c_int;
    pub fn blsgpu_kernel_timing(ctx: *mut BlsgpuCtx, enable: c_int) -> c_int;
    pub fn blsgpu_kernel_timing_report(ctx: *mut BlsgpuCtx, buf: *mut c_char, cap: usize, needed: *mut usize) -> c_int;
    pub fn blsgpu_fr_op(ctx: *mut BlsgpuCtx, op: c_int, a: *const u64, b: *const u64, n: usize, out: *mut u64, nonzero_flags: *mut u8) -> c_int;
    pub fn blsgpu_fr_op_device(ctx: *mut BlsgpuCtx, op: c_int, d_a: *const c_void, d_b: *const c_void, n: usize, d_out: *mut c_void, d_nonzero_flags: *mut c_void) -> c_int;
    pub fn blsgpu_fr_to_bytes(ctx: *mut BlsgpuCtx, scalars: *const u64, n: usize, bytes: *mut u8, ok: *mut u8) -> c_int;
    pub fn blsgpu_fr_from_bytes(ctx: *mut BlsgpuCtx, bytes: *const u8, n: usize, scalars: *mut u64, ok: *mut u8) -> c_int;
    pub fn blsgpu_fr_from_bytes_wide(ctx: *mut BlsgpuCtx, bytes: *const u8, n: usize, scalars: *mut u64) -> c_int;
    pub fn blsgpu_fr_to_bytes_device(ctx: *mut BlsgpuCtx, d_scalars: *const c_void, n: usize, d_bytes: *mut c_void, d_ok: *mut c_void) -> c_int;
    pub fn blsgpu_fr_from_bytes_device(ctx: *mut BlsgpuCtx, d_bytes: *const c_void, n: usize, d_scalars: *mut c_void, d_ok: *mut c_void) -> c_int;
    pub fn blsgpu_fr_from_bytes_wide_device(ctx: *mut BlsgpuCtx, d_bytes: *const c_void, n: usize, d_scalars: *mut c_void) -> c_int;
    pub fn blsgpu_fr_ntt(ctx: *mut BlsgpuCtx, data: *mut u64, log_n: c_int, inverse: c_int) -> c_int;
    pub fn blsgpu_fr_ntt_device(ctx: *mut BlsgpuCtx, d_data: *mut c_void, log_n: c_int, inverse: c_int) -> c_int;
    pub fn blsgpu_g1_hash_to_curve_batch(ctx: *mut BlsgpuCtx, msgs: *const u8, offsets: *const u64, n: usize, dst: *const u8, dst_len: usize, encode_only: c_int, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_g2_hash_to_curve_batch(ctx: *mut BlsgpuCtx, msgs: *const u8, offsets: *const u64, n: usize, dst: *const u8, dst_len: usize, encode_only: c_int, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_hash_to_curve_device(ctx: *mut BlsgpuCtx, group: c_int, d_msgs: *const c_void, d_offsets: *const c_void, n: usize, d_dst: *const c_void, dst_len: usize, encode_only: c_int, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_hash_to_curve_expander_batch(ctx: *mut BlsgpuCtx, group: c_int, expander: c_int, msgs: *const u8, offsets: *const u64, n: usize, dst: *const u8, dst_len: usize, encode_only: c_int, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_hash_to_curve_expander_device(ctx: *mut BlsgpuCtx, group: c_int, expander: c_int, d_msgs: *const c_void, d_offsets: *const c_void, n: usize, d_dst: *const c_void, dst_len: usize, encode_only: c_int, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_hash_to_curve_from_uniform_batch(ctx: *mut BlsgpuCtx, group: c_int, uniform: *const u8, n: usize, encode_only: c_int, out_xyz: *mut u64) -> c_int;
    pub fn blsgpu_hash_to_curve_from_uniform_device(ctx: *mut BlsgpuCtx, group: c_int, d_uniform: *const c_void, n: usize, encode_only: c_int, d_out_xyz: *mut c_void) -> c_int;
    pub fn blsgpu_expand_message_batch(ctx: *mut BlsgpuCtx, expander: c_int, msgs: *const u8, offsets: *const u64, n: usize, dst: *const u8, dst_len: usize, len_in_bytes: usize, out: *mut u8) -> c_int;
    pub fn blsgpu_expand_message_device(ctx: *mut BlsgpuCtx, expander: c_int, d_msgs: *const c_void, d_offsets: *const c_void, n: usize, d_dst: *const c_void, dst_len: usize, len_in_bytes: usize, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_hash_to_scalar_batch(ctx: *mut BlsgpuCtx, expander: c_int, msgs: *const u8, offsets: *const u64, n: usize, dst: *const u8, dst_len: usize, count: usize, out: *mut u64) -> c_int;
    pub fn blsgpu_hash_to_scalar_device(ctx: *mut BlsgpuCtx, expander: c_int, d_msgs: *const c_void, d_offsets: *const c_void, n: usize, d_dst: *const c_void, dst_len: usize, count: usize, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_g1_batch_normalize_device(ctx: *mut BlsgpuCtx, d_xyz: *const c_void, n: usize, d_xy: *mut c_void, d_infinity: *mut c_void) -> c_int;
    pub fn blsgpu_g2_batch_normalize_device(ctx: *mut BlsgpuCtx, d_xyz: *const c_void, n: usize, d_xy: *mut c_void, d_infinity: *mut c_void) -> c_int;
    pub fn blsgpu_g1_from_bytes_batch_device(ctx: *mut BlsgpuCtx, d_bytes: *const c_void, n: usize, compressed: c_int, checked: c_int, d_xy: *mut c_void, d_infinity: *mut c_void, d_ok: *mut c_void) -> c_int;
    pub fn blsgpu_g2_from_bytes_batch_device(ctx: *mut BlsgpuCtx, d_bytes: *const c_void, n: usize, compressed: c_int, checked: c_int, d_xy: *mut c_void, d_infinity: *mut c_void, d_ok: *mut c_void) -> c_int;
    pub fn blsgpu_g1_to_bytes_batch_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, n: usize, compressed: c_int, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_g2_to_bytes_batch_device(ctx: *mut BlsgpuCtx, d_xy: *const c_void, d_infinity: *const c_void, n: usize, compressed: c_int, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_gt_mul_scalar_batch_device(ctx: *mut BlsgpuCtx, d_gt: *const c_void, d_scalars: *const c_void, n: usize, d_out: *mut c_void) -> c_int;
    pub fn blsgpu_gt_is_identity_device(ctx: *mut BlsgpuCtx, d_gt: *const c_void, n: usize, d_flags: *mut c_void) -> c_int;
    pub fn blsgpu_bls_verify_batch(ctx: *mut BlsgpuCtx, mode: c_int, pk_bytes: *const u8, sig_bytes: *const u8, msgs: *const u8, offsets: *const u64, n: usize, dst: *const u8, dst_len: usize, verdict: *mut u8) -> c_int;
    pub fn blsgpu_bls_verify_batch_device(ctx: *mut BlsgpuCtx, mode: c_int, d_pk_bytes: *const c_void, d_sig_bytes: *const c_void, d_msgs: *const c_void, d_offsets: *const c_void, n: usize, d_dst: *const c_void, dst_len: usize, d_verdict: *mut c_void) -> c_int;
}
