//! `src/hip.rs` for a fork of zkcrypto/bls12_381 v0.8.0 -- the limb-level MI355X back end behind the crate's own types.
//!
//! Add `mod hip;` (behind a `hip` cargo feature) to src/lib.rs and copy `../src/ffi.rs` next to this file as `src/hip_ffi.rs`.
//! The crate denies `unsafe_code` (src/lib.rs:17); a `deny` lint may be lifted per module, which the first line below does.
//! Values cross the C ABI in the crate's own in-memory format -- `Fp([u64; 6])` canonical Montgomery limbs (src/fp.rs:15),
//! struct order for Fp2 / Fp6 / Fp12 (src/fp2.rs:11-14, src/fp6.rs:12-16, src/fp12.rs:13-16), `Scalar::to_bytes` for scalars
//! (src/scalar.rs:284-296), (X : Y : Z) with identity (0 : 1 : 0) for projective points (src/g1.rs:605-611) -- so the binding
//! is copies of limbs plus an explicit infinity byte (the `Choice` field of the affine types is not `repr(C)`, src/g1.rs:28-32).
//!
//! What is forwarded (see in-tree/README.md for the four call sites that change):
//!   `impl Engine for Bls12 { fn pairing }`                    src/pairings.rs:795-806   -> hip::pairing
//!   `impl MultiMillerLoop for Bls12 { fn multi_miller_loop }` src/pairings.rs:817-824   -> hip::multi_miller_loop
//!   `impl pairing::MillerLoopResult { fn final_exponentiation }` :808-814               -> hip::final_exponentiation
//!   slice-level `G1Projective::msm` / `G2Projective::msm`       (new; the element-wise `Mul` src/g1.rs:556-594 and `Sum`
//!                                                               :161-171 keep their CPU meaning for single elements)
//!   `G1Projective::batch_normalize` for n >= 4096              src/g1.rs:806-839         -> hip::batch_normalize_g1
#![allow(unsafe_code)]

use crate::fp::Fp;
use crate::fp2::Fp2;
use crate::fp6::Fp6;
use crate::fp12::Fp12;
use crate::hip_ffi as ffi;
use crate::{G1Affine, G1Projective, G2Affine, G2Projective, Gt, MillerLoopResult, Scalar};
use alloc::vec::Vec;
use core::ffi::c_int;

/// With the `hip` feature `G2Prepared` (opaque: private fields, src/pairings.rs:498-501) holds the affine point; the GPU
/// recomputes the 68 line-coefficient triples on the fly instead of reading 19 584 B per point from memory.
/// `G2PreparedHip::from(q)` keeps the affine point (right for a point met once: its lines are computed inside the Miller kernel);
/// `G2PreparedHip::resident_many(points)` does what the reference's `From<G2Affine> for G2Prepared` does (src/pairings.rs:504-546) --
/// the 68 coefficient triples are computed ONCE, into a device-resident table (`blsgpu_g2_prepare`) -- and every later
/// `multi_miller_loop` only evaluates them (verification keys, fixed generators).
pub struct PreparedTable(*mut ffi::BlsgpuG2Prepared);
unsafe impl Send for PreparedTable {}
unsafe impl Sync for PreparedTable {}
impl Drop for PreparedTable { fn drop(&mut self) { unsafe { ffi::blsgpu_g2_prepared_free(self.0) } } }
#[derive(Clone)]
pub struct G2PreparedHip { pub(crate) q: G2Affine, pub(crate) resident: Option<(alloc::sync::Arc<PreparedTable>, u32)> }
impl core::fmt::Debug for G2PreparedHip {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result { write!(f, "G2PreparedHip({:?}, resident: {})", self.q, self.resident.is_some()) }
}
impl From<G2Affine> for G2PreparedHip { fn from(q: G2Affine) -> Self { G2PreparedHip { q, resident: None } } }
impl G2PreparedHip {
    /// one device table for all of `points`; falls back to the plain form when no device is available
    pub fn resident_many(points: &[G2Affine]) -> Vec<G2PreparedHip> {
        let table = with_ctx(|ctx| {
            let (g2, f2) = g2_wire(points);
            let mut h = core::ptr::null_mut();
            ok(unsafe { ffi::blsgpu_g2_prepare(ctx, g2.as_ptr(), f2.as_ptr(), points.len(), &mut h) })?;
            Some(alloc::sync::Arc::new(PreparedTable(h)))
        });
        points.iter().enumerate().map(|(i, q)| G2PreparedHip { q: *q, resident: table.as_ref().map(|t| (t.clone(), i as u32)) }).collect()
    }
}
/// the table shared by the resident terms (the first one found) and the per-term indices (`UNPREPARED` for the others)
const UNPREPARED: u32 = 0xffff_ffff;
fn resident_indices(preps: &[&G2PreparedHip]) -> Option<(alloc::sync::Arc<PreparedTable>, Vec<u32>)> {
    let table = preps.iter().find_map(|p| p.resident.as_ref().map(|r| r.0.clone()))?;
    let qi = preps.iter().map(|p| match &p.resident { Some((t, i)) if alloc::sync::Arc::ptr_eq(t, &table) => *i, _ => UNPREPARED }).collect();
    Some((table, qi))
}

/// Process-wide context (one device).  `None` = no GPU / creation failed: every caller below then falls back to the CPU path,
/// which keeps the infallible signatures of the reference.
struct Ctx(*mut ffi::BlsgpuCtx);
unsafe impl Send for Ctx {}
static CTX: spin::Mutex<Option<Ctx>> = spin::Mutex::new(None);     // any lock works; the context is single-stream

fn with_ctx<R>(f: impl FnOnce(*mut ffi::BlsgpuCtx) -> Option<R>) -> Option<R> {
    let mut g = CTX.lock();
    if g.is_none() {
        let mut h = core::ptr::null_mut();
        if unsafe { ffi::blsgpu_create(0, &mut h) } != ffi::BLSGPU_OK { return None; }
        // every scalar argument of this module is `&[Scalar]` memory (see scalar_bytes)
        if unsafe { ffi::blsgpu_set_scalar_form(h, ffi::BLSGPU_SCALAR_MONT) } != ffi::BLSGPU_OK { unsafe { ffi::blsgpu_destroy(h) }; return None; }
        *g = Some(Ctx(h));
    }
    f(g.as_ref().unwrap().0)
}
fn ok(rc: c_int) -> Option<()> { (rc == ffi::BLSGPU_OK).then_some(()) }

// ---- limbs <-> crate types ---------------------------------------------------------------------------------------------
fn fp(l: &[u64]) -> Fp { Fp::from_raw_unchecked([l[0], l[1], l[2], l[3], l[4], l[5]]) }       // src/fp.rs:302
fn fp2(l: &[u64]) -> Fp2 { Fp2 { c0: fp(&l[0..6]), c1: fp(&l[6..12]) } }
fn fp6(l: &[u64]) -> Fp6 { Fp6 { c0: fp2(&l[0..12]), c1: fp2(&l[12..24]), c2: fp2(&l[24..36]) } }
fn fp12(l: &[u64]) -> Fp12 { Fp12 { c0: fp6(&l[0..36]), c1: fp6(&l[36..72]) } }
fn put_fp(out: &mut Vec<u64>, a: &Fp) { out.extend_from_slice(&a.0); }
fn put_fp2(out: &mut Vec<u64>, a: &Fp2) { put_fp(out, &a.c0); put_fp(out, &a.c1); }
fn put_fp12(out: &mut Vec<u64>, a: &Fp12) {
    for c6 in [&a.c0, &a.c1] { for c2 in [&c6.c0, &c6.c1, &c6.c2] { put_fp2(out, c2); } }
}
fn g1_wire(points: &[G1Affine]) -> (Vec<u64>, Vec<u8>) {
    let (mut xy, mut inf) = (Vec::with_capacity(points.len() * 12), Vec::with_capacity(points.len()));
    for p in points { put_fp(&mut xy, &p.x); put_fp(&mut xy, &p.y); inf.push(bool::from(p.is_identity()) as u8); }
    (xy, inf)
}
fn g2_wire(points: &[G2Affine]) -> (Vec<u64>, Vec<u8>) {
    let (mut xy, mut inf) = (Vec::with_capacity(points.len() * 24), Vec::with_capacity(points.len()));
    for p in points { put_fp2(&mut xy, &p.x); put_fp2(&mut xy, &p.y); inf.push(bool::from(p.is_identity()) as u8); }
    (xy, inf)
}
/// The scalars exactly as the slice holds them: four u64 Montgomery limbs each (src/scalar.rs:23-27).  The library context is
/// switched to BLSGPU_SCALAR_MONT when it is created (below), so `Scalar::to_bytes` -- one `montgomery_reduce` per scalar,
/// src/scalar.rs:284-296 -- runs on the device inside the kernels that decompose the scalars; no per-element host work, no copy.
/// Needs `#[repr(transparent)]` on `pub struct Scalar(pub(crate) [u64; 4])` (a one-line addition in the fork, src/scalar.rs:23);
/// the assertions keep a layout surprise from going unnoticed.
const _: () = assert!(core::mem::size_of::<Scalar>() == 32 && core::mem::align_of::<Scalar>() == 8);
fn scalar_bytes(scalars: &[Scalar]) -> &[u8] {
    unsafe { core::slice::from_raw_parts(scalars.as_ptr() as *const u8, scalars.len() * 32) }
}

// ---- MSM ----------------------------------------------------------------------------------------------------------------
/// `bases.iter().zip(scalars).map(|(p, s)| p * s).sum::<G1Projective>()`  (src/g1.rs:573-579, 754-774, 161-171)
pub fn msm_g1(bases: &[G1Affine], scalars: &[Scalar]) -> G1Projective {
    assert_eq!(bases.len(), scalars.len());
    let gpu = with_ctx(|ctx| {
        let ((xy, inf), s) = (g1_wire(bases), scalar_bytes(scalars));
        let mut out = [0u64; 18];
        ok(unsafe { ffi::blsgpu_g1_msm_host(ctx, xy.as_ptr(), inf.as_ptr(), s.as_ptr(), bases.len(), out.as_mut_ptr()) })?;
        Some(G1Projective { x: fp(&out[0..6]), y: fp(&out[6..12]), z: fp(&out[12..18]) })
    });
    gpu.unwrap_or_else(|| bases.iter().zip(scalars).map(|(p, s)| p * s).sum())
}
/// the same over G2 (src/g2.rs:626-632, 825-845, 162-172)
pub fn msm_g2(bases: &[G2Affine], scalars: &[Scalar]) -> G2Projective {
    assert_eq!(bases.len(), scalars.len());
    let gpu = with_ctx(|ctx| {
        let ((xy, inf), s) = (g2_wire(bases), scalar_bytes(scalars));
        let mut out = [0u64; 36];
        ok(unsafe { ffi::blsgpu_g2_msm_host(ctx, xy.as_ptr(), inf.as_ptr(), s.as_ptr(), bases.len(), out.as_mut_ptr()) })?;
        Some(G2Projective { x: fp2(&out[0..12]), y: fp2(&out[12..24]), z: fp2(&out[24..36]) })
    });
    gpu.unwrap_or_else(|| bases.iter().zip(scalars).map(|(p, s)| p * s).sum())
}
/// `points.iter().zip(scalars).map(|(p, s)| p * s)` collected -- N variable-base scalar multiplications in one device call
/// (`Mul<&Scalar> for &G1Affine`, src/g1.rs:573-579 -> `multiply` :754-774); exact for every curve point
pub fn mul_batch_g1(points: &[G1Affine], scalars: &[Scalar]) -> Vec<G1Projective> {
    assert_eq!(points.len(), scalars.len());
    let gpu = with_ctx(|ctx| {
        let ((xy, inf), s) = (g1_wire(points), scalar_bytes(scalars));
        let mut out = alloc::vec![0u64; points.len() * 18];
        ok(unsafe { ffi::blsgpu_g1_mul_batch(ctx, xy.as_ptr(), inf.as_ptr(), s.as_ptr(), points.len(), out.as_mut_ptr()) })?;
        Some(out.chunks_exact(18).map(|c| G1Projective { x: fp(&c[0..6]), y: fp(&c[6..12]), z: fp(&c[12..18]) }).collect::<Vec<_>>())
    });
    gpu.unwrap_or_else(|| points.iter().zip(scalars).map(|(p, s)| p * s).collect())
}
/// the same over G2 (src/g2.rs:626-632, 825-845)
pub fn mul_batch_g2(points: &[G2Affine], scalars: &[Scalar]) -> Vec<G2Projective> {
    assert_eq!(points.len(), scalars.len());
    let gpu = with_ctx(|ctx| {
        let ((xy, inf), s) = (g2_wire(points), scalar_bytes(scalars));
        let mut out = alloc::vec![0u64; points.len() * 36];
        ok(unsafe { ffi::blsgpu_g2_mul_batch(ctx, xy.as_ptr(), inf.as_ptr(), s.as_ptr(), points.len(), out.as_mut_ptr()) })?;
        Some(out.chunks_exact(36).map(|c| G2Projective { x: fp2(&c[0..12]), y: fp2(&c[12..24]), z: fp2(&c[24..36]) }).collect::<Vec<_>>())
    });
    gpu.unwrap_or_else(|| points.iter().zip(scalars).map(|(p, s)| p * s).collect())
}
/// `Sum` over a slice of projective points (src/g1.rs:161-171) on the device (used to fold per-GPU partial sums)
pub fn sum_g1(points: &[G1Projective]) -> G1Projective {
    let gpu = with_ctx(|ctx| {
        let mut xyz = Vec::with_capacity(points.len() * 18);
        for p in points { put_fp(&mut xyz, &p.x); put_fp(&mut xyz, &p.y); put_fp(&mut xyz, &p.z); }
        let mut out = [0u64; 18];
        ok(unsafe { ffi::blsgpu_g1_sum(ctx, xyz.as_ptr(), points.len(), out.as_mut_ptr()) })?;
        Some(G1Projective { x: fp(&out[0..6]), y: fp(&out[6..12]), z: fp(&out[12..18]) })
    });
    gpu.unwrap_or_else(|| points.iter().sum())
}
/// `G1Projective::batch_normalize` (src/g1.rs:806-839): Montgomery's trick on the GPU; `q.len()` must equal `p.len()`
pub fn batch_normalize_g1(p: &[G1Projective], q: &mut [G1Affine]) {
    assert_eq!(p.len(), q.len());
    let done = with_ctx(|ctx| {
        let mut xyz = Vec::with_capacity(p.len() * 18);
        for a in p { put_fp(&mut xyz, &a.x); put_fp(&mut xyz, &a.y); put_fp(&mut xyz, &a.z); }
        let (mut xy, mut inf) = (alloc::vec![0u64; p.len() * 12], alloc::vec![0u8; p.len()]);
        ok(unsafe { ffi::blsgpu_g1_batch_normalize(ctx, xyz.as_ptr(), p.len(), xy.as_mut_ptr(), inf.as_mut_ptr()) })?;
        for (i, out) in q.iter_mut().enumerate() {
            *out = if inf[i] != 0 { G1Affine::identity() } else {
                G1Affine { x: fp(&xy[12 * i..12 * i + 6]), y: fp(&xy[12 * i + 6..12 * i + 12]), infinity: subtle::Choice::from(0u8) }
            };
        }
        Some(())
    });
    if done.is_none() { G1Projective::batch_normalize_cpu(p, q); }     // the existing body of src/g1.rs:806-839, renamed
}

// ---- pairings -----------------------------------------------------------------------------------------------------------
/// Batched `pairing` (src/pairings.rs:607-653): out[i] = e(p[i], q[i]); identities give `Gt::identity()` as in :636-651.
pub fn pairing_batch(p: &[G1Affine], q: &[G2Affine]) -> Vec<Gt> {
    assert_eq!(p.len(), q.len());
    if p.len() < GPU_MIN_PAIRINGS { return p.iter().zip(q).map(|(a, b)| crate::pairings::pairing_cpu(a, b)).collect(); }
    let gpu = with_ctx(|ctx| {
        let ((g1, f1), (g2, f2)) = (g1_wire(p), g2_wire(q));
        let mut out = alloc::vec![0u64; p.len() * 72];
        ok(unsafe { ffi::blsgpu_pairing_batch(ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), p.len(), out.as_mut_ptr()) })?;
        Some(out.chunks_exact(72).map(|c| Gt(fp12(c))).collect::<Vec<_>>())
    });
    gpu.unwrap_or_else(|| p.iter().zip(q).map(|(a, b)| crate::pairings::pairing_cpu(a, b)).collect())
}
/// Below these batch sizes the crate's own CPU code is at least as fast as a device round trip.  One call with 1..256 items takes
/// the library's wide path (one item per 1024-lane workgroup): ~1.15 ms for pairings, ~0.4 ms for Miller loops, ~0.8 ms for
/// final exponentiations, whatever the count (bench.py `pairing_small_batches`); one pairing on a host core is ~1.1-1.3 ms, one
/// Miller loop ~0.45 ms, one final exponentiation ~0.7 ms (`per_op_ns`).  So ONE item stays on the CPU (a tie, minus
/// the copies) and the device pays from two items on.
pub const GPU_MIN_PAIRINGS: usize = 2;
pub const GPU_MIN_MILLER_TERMS: usize = 2;
/// `pairing` for one pair (what `Engine::pairing` forwards to): the CPU path of the crate, unchanged (src/pairings.rs:607-653,
/// renamed `pairing_cpu`) -- a single pair never goes to the device.  Callers with many pairs use `pairing_batch`.
pub fn pairing(p: &G1Affine, q: &G2Affine) -> Gt { crate::pairings::pairing_cpu(p, q) }

/// `multi_miller_loop` (src/pairings.rs:554-603); terms with an identity are skipped (:566-569); no terms give `default()`.
pub fn multi_miller_loop(terms: &[(&G1Affine, &G2PreparedHip)]) -> MillerLoopResult {
    if terms.len() < GPU_MIN_MILLER_TERMS { return crate::pairings::multi_miller_loop_cpu(terms); }
    let gpu = with_ctx(|ctx| {
        let p: Vec<G1Affine> = terms.iter().map(|t| *t.0).collect();
        let q: Vec<G2Affine> = terms.iter().map(|t| t.1.q).collect();
        let ((g1, f1), (g2, f2)) = (g1_wire(&p), g2_wire(&q));
        let mut out = [0u64; 72];
        let preps: Vec<&G2PreparedHip> = terms.iter().map(|t| t.1).collect();
        match resident_indices(&preps) {
            Some((table, qi)) => ok(unsafe { ffi::blsgpu_multi_miller_loop_prepared(ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), qi.as_ptr(), table.0, terms.len(),
                                                                                    out.as_mut_ptr()) })?,
            None => ok(unsafe { ffi::blsgpu_multi_miller_loop(ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), terms.len(), out.as_mut_ptr()) })?,
        }
        Some(MillerLoopResult(fp12(&out)))
    });
    gpu.unwrap_or_else(|| crate::pairings::multi_miller_loop_cpu(terms))
}
/// `MillerLoopResult::final_exponentiation` (src/pairings.rs:48-176) of ONE value: the crate's CPU code (~0.7 ms on a host
/// core against ~0.8 ms for one value on the device); `final_exponentiation_batch` is the device entry point
pub fn final_exponentiation(f: &MillerLoopResult) -> Gt { f.final_exponentiation_cpu() }
pub fn final_exponentiation_batch(fs: &[MillerLoopResult]) -> Vec<Gt> {
    if fs.len() < GPU_MIN_PAIRINGS { return fs.iter().map(|f| f.final_exponentiation_cpu()).collect(); }
    let gpu = with_ctx(|ctx| {
        let mut inp = Vec::with_capacity(72 * fs.len());
        for f in fs { put_fp12(&mut inp, &f.0); }
        let mut out = alloc::vec![0u64; 72 * fs.len()];
        ok(unsafe { ffi::blsgpu_final_exponentiation_batch(ctx, inp.as_ptr(), fs.len(), out.as_mut_ptr()) })?;
        Some(out.chunks_exact(72).map(|c| Gt(fp12(c))).collect::<Vec<_>>())
    });
    gpu.unwrap_or_else(|| fs.iter().map(|f| f.final_exponentiation_cpu()).collect())
}
/// `MillerLoopResult + MillerLoopResult` over a slice (src/pairings.rs:179-186): fold of per-GPU partial products
pub fn miller_product(parts: &[MillerLoopResult]) -> MillerLoopResult {
    let gpu = with_ctx(|ctx| {
        let mut inp = Vec::with_capacity(parts.len() * 72);
        for p in parts { put_fp12(&mut inp, &p.0); }
        let mut out = [0u64; 72];
        ok(unsafe { ffi::blsgpu_fp12_product(ctx, inp.as_ptr(), parts.len(), out.as_mut_ptr()) })?;
        Some(MillerLoopResult(fp12(&out)))
    });
    gpu.unwrap_or_else(|| parts.iter().fold(MillerLoopResult::default(), |a, b| a + b))
}

/// One `multi_miller_loop(terms).final_exponentiation()` per equation in ONE device call -- the bulk form of what generic
/// `E: MultiMillerLoop` verifiers do once per signature (src/pairings.rs:554-603, 817-824, 48-176).  Equations with no terms give
/// `Gt::identity()`.
pub fn multi_miller_loop_many(equations: &[&[(&G1Affine, &G2PreparedHip)]]) -> Vec<Gt> {
    let cpu = || equations.iter().map(|e| crate::pairings::multi_miller_loop_cpu(e).final_exponentiation_cpu()).collect::<Vec<_>>();
    if equations.len() < GPU_MIN_PAIRINGS { return cpu(); }
    let gpu = with_ctx(|ctx| {
        let p: Vec<G1Affine> = equations.iter().flat_map(|e| e.iter().map(|t| *t.0)).collect();
        let q: Vec<G2Affine> = equations.iter().flat_map(|e| e.iter().map(|t| t.1.q)).collect();
        let mut off = Vec::with_capacity(equations.len() + 1);
        off.push(0u64);
        for e in equations { off.push(off[off.len() - 1] + e.len() as u64); }
        let ((g1, f1), (g2, f2)) = (g1_wire(&p), g2_wire(&q));
        let mut out = alloc::vec![0u64; equations.len() * 72];
        let preps: Vec<&G2PreparedHip> = equations.iter().flat_map(|e| e.iter().map(|t| t.1)).collect();
        match resident_indices(&preps) {
            Some((table, qi)) => ok(unsafe { ffi::blsgpu_multi_miller_loop_prepared_many(ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), qi.as_ptr(), table.0, off.as_ptr(),
                                                                                         equations.len(), 1, out.as_mut_ptr()) })?,
            None => ok(unsafe { ffi::blsgpu_multi_miller_loop_many(ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), off.as_ptr(), equations.len(), 1, out.as_mut_ptr()) })?,
        }
        Some(out.chunks_exact(72).map(|c| Gt(fp12(c))).collect::<Vec<_>>())
    });
    gpu.unwrap_or_else(cpu)
}

// ---- all GPUs of the node from this process ------------------------------------------------------------------------------
/// One library context (and, inside the library, one host thread) per listed device.  MSMs, batches of pairings and
/// `multi_miller_loop`s are dealt to the members in contiguous slices; the members' partial results -- one group element each
/// -- are folded with the crate's own `Sum` (src/g1.rs:161-171) / `MillerLoopResult + MillerLoopResult` (src/pairings.rs:179-186),
/// followed by ONE final exponentiation.  `None` from the constructors = no usable device: callers keep the CPU path.
pub struct Group(*mut ffi::BlsgpuGroup);
unsafe impl Send for Group {}
impl Drop for Group { fn drop(&mut self) { unsafe { ffi::blsgpu_group_destroy(self.0) } } }
impl Group {
    pub fn new(devices: &[c_int]) -> Option<Group> {
        let mut h = core::ptr::null_mut();
        ok(unsafe { ffi::blsgpu_group_create(devices.as_ptr(), devices.len() as c_int, &mut h) })?;
        let g = Group(h);                                        // (dropped -- and the group destroyed -- if a member refuses the setting)
        for k in 0..devices.len() as c_int {                     // scalars travel as `&[Scalar]` memory on every member (see scalar_bytes)
            ok(unsafe { ffi::blsgpu_set_scalar_form(ffi::blsgpu_group_ctx(h, k), ffi::BLSGPU_SCALAR_MONT) })?;
        }
        Some(g)
    }
    /// every device the HIP runtime shows
    pub fn all_devices() -> Option<Group> {
        let n = unsafe { ffi::blsgpu_device_count() };
        if n <= 0 { return None; }
        let d: Vec<c_int> = (0..n).collect();
        Group::new(&d)
    }
    pub fn len(&self) -> usize { unsafe { ffi::blsgpu_group_size(self.0) as usize } }
    /// `bases.iter().zip(scalars).map(|(p, s)| p * s).sum::<G1Projective>()` over all members
    pub fn msm_g1(&self, bases: &[G1Affine], scalars: &[Scalar]) -> G1Projective {
        assert_eq!(bases.len(), scalars.len());
        let gpu = (|| {
            let ((xy, inf), s) = (g1_wire(bases), scalar_bytes(scalars));
            let mut b = core::ptr::null_mut();
            ok(unsafe { ffi::blsgpu_group_bases_upload(self.0, 1, xy.as_ptr(), inf.as_ptr(), bases.len(), &mut b) })?;
            let mut out = [0u64; 18];
            let rc = unsafe { ffi::blsgpu_g1_msm_sharded(self.0, b, s.as_ptr(), bases.len(), out.as_mut_ptr()) };
            unsafe { ffi::blsgpu_group_bases_free(b) };
            ok(rc)?;
            Some(G1Projective { x: fp(&out[0..6]), y: fp(&out[6..12]), z: fp(&out[12..18]) })
        })();
        gpu.unwrap_or_else(|| bases.iter().zip(scalars).map(|(p, s)| p * s).sum())
    }
    pub fn msm_g2(&self, bases: &[G2Affine], scalars: &[Scalar]) -> G2Projective {
        assert_eq!(bases.len(), scalars.len());
        let gpu = (|| {
            let ((xy, inf), s) = (g2_wire(bases), scalar_bytes(scalars));
            let mut b = core::ptr::null_mut();
            ok(unsafe { ffi::blsgpu_group_bases_upload(self.0, 2, xy.as_ptr(), inf.as_ptr(), bases.len(), &mut b) })?;
            let mut out = [0u64; 36];
            let rc = unsafe { ffi::blsgpu_g2_msm_sharded(self.0, b, s.as_ptr(), bases.len(), out.as_mut_ptr()) };
            unsafe { ffi::blsgpu_group_bases_free(b) };
            ok(rc)?;
            Some(G2Projective { x: fp2(&out[0..12]), y: fp2(&out[12..24]), z: fp2(&out[24..36]) })
        })();
        gpu.unwrap_or_else(|| bases.iter().zip(scalars).map(|(p, s)| p * s).sum())
    }
    /// out[i] = e(p[i], q[i]), index slices per member
    pub fn pairing_batch(&self, p: &[G1Affine], q: &[G2Affine]) -> Vec<Gt> {
        assert_eq!(p.len(), q.len());
        let gpu = (|| {
            let ((g1, f1), (g2, f2)) = (g1_wire(p), g2_wire(q));
            let mut out = alloc::vec![0u64; p.len() * 72];
            ok(unsafe { ffi::blsgpu_pairing_batch_sharded(self.0, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), p.len(), out.as_mut_ptr()) })?;
            Some(out.chunks_exact(72).map(|c| Gt(fp12(c))).collect::<Vec<_>>())
        })();
        gpu.unwrap_or_else(|| p.iter().zip(q).map(|(a, b)| crate::pairings::pairing_cpu(a, b)).collect())
    }
    /// `multi_miller_loop(terms)` with the terms dealt to the members (raw `MillerLoopResult`: apply `final_exponentiation` once)
    pub fn multi_miller_loop(&self, terms: &[(&G1Affine, &G2PreparedHip)]) -> MillerLoopResult {
        let gpu = (|| {
            let p: Vec<G1Affine> = terms.iter().map(|t| *t.0).collect();
            let q: Vec<G2Affine> = terms.iter().map(|t| t.1.q).collect();
            let ((g1, f1), (g2, f2)) = (g1_wire(&p), g2_wire(&q));
            let mut out = [0u64; 72];
            ok(unsafe { ffi::blsgpu_multi_miller_loop_sharded(self.0, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), terms.len(), 0, out.as_mut_ptr()) })?;
            Some(MillerLoopResult(fp12(&out)))
        })();
        gpu.unwrap_or_else(|| crate::pairings::multi_miller_loop_cpu(terms))
    }
}

// ---- trait forwarding (replaces the bodies at src/pairings.rs:795-824) --------------------------------------------------
#[cfg(feature = "hip")]
impl pairing::Engine for crate::Bls12 {
    type Fr = Scalar;
    type G1 = G1Projective;
    type G1Affine = G1Affine;
    type G2 = G2Projective;
    type G2Affine = G2Affine;
    type Gt = Gt;
    fn pairing(p: &Self::G1Affine, q: &Self::G2Affine) -> Self::Gt { pairing(p, q) }
}
#[cfg(feature = "hip")]
impl pairing::MillerLoopResult for MillerLoopResult {
    type Gt = Gt;
    fn final_exponentiation(&self) -> Self::Gt { final_exponentiation(self) }
}
#[cfg(feature = "hip")]
impl pairing::MultiMillerLoop for crate::Bls12 {
    type G2Prepared = G2PreparedHip;
    type Result = MillerLoopResult;
    fn multi_miller_loop(terms: &[(&Self::G1Affine, &Self::G2Prepared)]) -> Self::Result { multi_miller_loop(terms) }
}
/// slice-level helpers next to the element-wise operators (`Mul<&Scalar>` src/g1.rs:556-594 and `Sum` :161-171 are unchanged)
impl G1Projective {
    pub fn msm(bases: &[G1Affine], scalars: &[Scalar]) -> G1Projective { msm_g1(bases, scalars) }
    /// `Mul<&Scalar>` over slices: out[i] = points[i] * scalars[i]
    pub fn mul_slices(points: &[G1Affine], scalars: &[Scalar]) -> Vec<G1Projective> { mul_batch_g1(points, scalars) }
    pub fn sum_slice(points: &[G1Projective]) -> G1Projective { sum_g1(points) }
}
impl G2Projective {
    pub fn msm(bases: &[G2Affine], scalars: &[Scalar]) -> G2Projective { msm_g2(bases, scalars) }
    pub fn mul_slices(points: &[G2Affine], scalars: &[Scalar]) -> Vec<G2Projective> { mul_batch_g2(points, scalars) }
}
