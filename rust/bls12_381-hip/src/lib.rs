//! MI355X back end for the hot path of `bls12_381` through the C ABI of `libblsgpu.so` (include/bls12_381_hip.h).
//!
//! This crate is the OUT-OF-TREE binding: it can only see what `bls12_381` 0.8 exports, so group elements cross the
//! boundary in the crate's public encodings (`to_uncompressed` / `from_uncompressed_unchecked`, `Scalar::to_bytes`) and
//! target-group values -- `Gt` has no public constructor or encoding -- stay as the 72 canonical Montgomery limbs the
//! reference keeps inside `Gt(Fp12)` (`GtLimbs`; equality of limbs is equality of group elements because `Fp` is always
//! fully reduced, src/fp.rs:361-379).  The limb-level module that returns the crate's own `Gt` / `MillerLoopResult` and
//! forwards `pairing::Engine` / `MultiMillerLoop` for `Bls12` lives in `in-tree/hip.rs` (it needs `pub(crate)` fields).
//!
//! Reference semantics, by entry point:
//!   msm_g1 / msm_g2           bases.iter().zip(scalars).map(|(p, s)| p * s).sum()      src/g1.rs:573-579,754-774,161-171
//!   pairing_batch             pairing(p_i, q_i) for every i                             src/pairings.rs:607-653
//!   multi_miller_loop         multi_miller_loop(&[(p_i, prepared q_i)])                 src/pairings.rs:554-603
//!   final_exponentiation      MillerLoopResult::final_exponentiation                    src/pairings.rs:48-176
//!   batch_normalize_g1        G1Projective::batch_normalize                             src/g1.rs:806-839
//!   multi_miller_loop_many    multi_miller_loop(terms_s).final_exponentiation() for every equation s    src/pairings.rs:554-603, 48-176
//!   GpuGroup::*               the same operations sharded over the GPUs of the node (folds: `Sum`, `MillerLoopResult +`)
//! Hot-path functions of the reference are infallible; here a HIP failure or a bad argument is an `Err(Error)` and the
//! caller decides (fall back to the CPU expression above, or propagate).  The library itself never computes on the CPU.
#![allow(clippy::missing_safety_doc)]

pub mod ffi;

use bls12_381::{G1Affine, G1Projective, G2Affine, G2Projective, Scalar};
use core::ffi::{c_int, CStr};
use group::Curve;

/// Error of a library call: the status code and `blsgpu_last_error()`.
#[derive(Debug, Clone)]
pub struct Error { pub code: i32, pub message: String }

fn check(rc: c_int) -> Result<(), Error> {
    if rc == ffi::BLSGPU_OK { return Ok(()); }
    let message = unsafe {
        let p = ffi::blsgpu_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    Err(Error { code: rc, message })
}

/// One context = one device with its streams and scratch memory.  Not `Sync`: use one per host thread.
pub struct Gpu { ctx: *mut ffi::BlsgpuCtx }
unsafe impl Send for Gpu {}

impl Gpu {
    pub fn new(device: i32) -> Result<Gpu, Error> {
        let mut ctx = core::ptr::null_mut();
        check(unsafe { ffi::blsgpu_create(device, &mut ctx) })?;
        Ok(Gpu { ctx })
    }
    pub fn device_count() -> i32 { unsafe { ffi::blsgpu_device_count() } }
    pub fn raw(&self) -> *mut ffi::BlsgpuCtx { self.ctx }
    /// waits for everything queued on this context; also reports a non-canonical scalar seen by an asynchronous MSM
    pub fn synchronize(&self) -> Result<(), Error> { check(unsafe { ffi::blsgpu_synchronize(self.ctx) }) }
}
impl Drop for Gpu { fn drop(&mut self) { unsafe { ffi::blsgpu_destroy(self.ctx) } } }

/// The 72 canonical Montgomery limbs of a `Gt` / `MillerLoopResult` in struct order c0.c0.c0 .. c1.c2.c1 (src/fp12.rs:13-16).
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub struct GtLimbs(pub [u64; 72]);

fn scalar_bytes(scalars: &[Scalar]) -> Vec<u8> {
    let mut s = Vec::with_capacity(scalars.len() * 32);
    for k in scalars { s.extend_from_slice(&k.to_bytes()); }          // src/scalar.rs:284-296: canonical, little endian
    s
}
fn g1_bytes(points: &[G1Affine]) -> Vec<u8> {
    let mut b = Vec::with_capacity(points.len() * 96);
    for p in points { b.extend_from_slice(&p.to_uncompressed()); }    // src/g1.rs:246-260
    b
}
fn g2_bytes(points: &[G2Affine]) -> Vec<u8> {
    let mut b = Vec::with_capacity(points.len() * 192);
    for p in points { b.extend_from_slice(&p.to_uncompressed()); }    // src/g2.rs:284-299
    b
}

/// Drop-in for `bases.iter().zip(scalars).map(|(p, s)| p * s).sum::<G1Projective>()`.
pub fn msm_g1(gpu: &Gpu, bases: &[G1Affine], scalars: &[Scalar]) -> Result<G1Projective, Error> {
    assert_eq!(bases.len(), scalars.len());
    let (b, s) = (g1_bytes(bases), scalar_bytes(scalars));
    let mut out = [0u8; 96];
    check(unsafe { ffi::blsgpu_g1_msm_bytes(gpu.ctx, b.as_ptr(), s.as_ptr(), bases.len(), out.as_mut_ptr()) })?;
    // the library returns a valid encoding of a subgroup point; `unchecked` skips a second subgroup check
    let aff = Option::<G1Affine>::from(G1Affine::from_uncompressed_unchecked(&out)).expect("libblsgpu returned an invalid G1 encoding");
    Ok(G1Projective::from(aff))
}

/// Drop-in for the same expression over G2 (src/g2.rs:626-632,825-845,162-172).
pub fn msm_g2(gpu: &Gpu, bases: &[G2Affine], scalars: &[Scalar]) -> Result<G2Projective, Error> {
    assert_eq!(bases.len(), scalars.len());
    let (b, s) = (g2_bytes(bases), scalar_bytes(scalars));
    let mut out = [0u8; 192];
    check(unsafe { ffi::blsgpu_g2_msm_bytes(gpu.ctx, b.as_ptr(), s.as_ptr(), bases.len(), out.as_mut_ptr()) })?;
    let aff = Option::<G2Affine>::from(G2Affine::from_uncompressed_unchecked(&out)).expect("libblsgpu returned an invalid G2 encoding");
    Ok(G2Projective::from(aff))
}

/// Decode uncompressed encodings into the wire limbs of the ABI (x | y Montgomery limbs + infinity bytes) on the GPU.
fn g1_wire(gpu: &Gpu, points: &[G1Affine]) -> Result<(Vec<u64>, Vec<u8>), Error> {
    let n = points.len();
    let bytes = g1_bytes(points);
    let (mut xy, mut inf, mut ok) = (vec![0u64; n * 12], vec![0u8; n], vec![0u8; n]);
    check(unsafe { ffi::blsgpu_g1_from_bytes_batch(gpu.ctx, bytes.as_ptr(), n, 0, 0, xy.as_mut_ptr(), inf.as_mut_ptr(), ok.as_mut_ptr()) })?;
    debug_assert!(ok.iter().all(|&o| o == 1));
    Ok((xy, inf))
}
fn g2_wire(gpu: &Gpu, points: &[G2Affine]) -> Result<(Vec<u64>, Vec<u8>), Error> {
    let n = points.len();
    let bytes = g2_bytes(points);
    let (mut xy, mut inf, mut ok) = (vec![0u64; n * 24], vec![0u8; n], vec![0u8; n]);
    check(unsafe { ffi::blsgpu_g2_from_bytes_batch(gpu.ctx, bytes.as_ptr(), n, 0, 0, xy.as_mut_ptr(), inf.as_mut_ptr(), ok.as_mut_ptr()) })?;
    debug_assert!(ok.iter().all(|&o| o == 1));
    Ok((xy, inf))
}
/// Drop-in for `points.iter().zip(scalars).map(|(p, s)| p * s).collect::<Vec<G1Projective>>()` -- `Mul<&Scalar>` over
/// slices (src/g1.rs:573-579, 754-774): decode on the device, N variable-base multiplications in one launch, one batched
/// affine conversion, encode; exact for every curve point (no subgroup precondition).
pub fn mul_batch_g1(gpu: &Gpu, points: &[G1Affine], scalars: &[Scalar]) -> Result<Vec<G1Projective>, Error> {
    assert_eq!(points.len(), scalars.len());
    let n = points.len();
    let ((xy, inf), s) = (g1_wire(gpu, points)?, scalar_bytes(scalars));
    let (mut xyz, mut axy, mut ainf, mut enc) = (vec![0u64; n * 18], vec![0u64; n * 12], vec![0u8; n], vec![0u8; n * 96]);
    check(unsafe { ffi::blsgpu_g1_mul_batch(gpu.ctx, xy.as_ptr(), inf.as_ptr(), s.as_ptr(), n, xyz.as_mut_ptr()) })?;
    check(unsafe { ffi::blsgpu_g1_batch_normalize(gpu.ctx, xyz.as_ptr(), n, axy.as_mut_ptr(), ainf.as_mut_ptr()) })?;
    check(unsafe { ffi::blsgpu_g1_to_bytes_batch(gpu.ctx, axy.as_ptr(), ainf.as_ptr(), n, 0, enc.as_mut_ptr()) })?;
    Ok(enc.chunks_exact(96).map(|c| {
        let mut b = [0u8; 96]; b.copy_from_slice(c);
        G1Projective::from(Option::<G1Affine>::from(G1Affine::from_uncompressed_unchecked(&b)).expect("libblsgpu returned an invalid G1 encoding"))
    }).collect())
}
/// the same over G2 (src/g2.rs:626-632, 825-845)
pub fn mul_batch_g2(gpu: &Gpu, points: &[G2Affine], scalars: &[Scalar]) -> Result<Vec<G2Projective>, Error> {
    assert_eq!(points.len(), scalars.len());
    let n = points.len();
    let ((xy, inf), s) = (g2_wire(gpu, points)?, scalar_bytes(scalars));
    let (mut xyz, mut axy, mut ainf, mut enc) = (vec![0u64; n * 36], vec![0u64; n * 24], vec![0u8; n], vec![0u8; n * 192]);
    check(unsafe { ffi::blsgpu_g2_mul_batch(gpu.ctx, xy.as_ptr(), inf.as_ptr(), s.as_ptr(), n, xyz.as_mut_ptr()) })?;
    check(unsafe { ffi::blsgpu_g2_batch_normalize(gpu.ctx, xyz.as_ptr(), n, axy.as_mut_ptr(), ainf.as_mut_ptr()) })?;
    check(unsafe { ffi::blsgpu_g2_to_bytes_batch(gpu.ctx, axy.as_ptr(), ainf.as_ptr(), n, 0, enc.as_mut_ptr()) })?;
    Ok(enc.chunks_exact(192).map(|c| {
        let mut b = [0u8; 192]; b.copy_from_slice(c);
        G2Projective::from(Option::<G2Affine>::from(G2Affine::from_uncompressed_unchecked(&b)).expect("libblsgpu returned an invalid G2 encoding"))
    }).collect())
}
fn split72(flat: Vec<u64>) -> Vec<GtLimbs> {
    flat.chunks_exact(72).map(|c| { let mut a = [0u64; 72]; a.copy_from_slice(c); GtLimbs(a) }).collect()
}

/// Batched `pairing` (src/pairings.rs:607-653): out[i] = e(p[i], q[i]); an identity on either side gives Gt::identity().
pub fn pairing_batch(gpu: &Gpu, p: &[G1Affine], q: &[G2Affine]) -> Result<Vec<GtLimbs>, Error> {
    assert_eq!(p.len(), q.len());
    let n = p.len();
    let ((g1, f1), (g2, f2)) = (g1_wire(gpu, p)?, g2_wire(gpu, q)?);
    let mut out = vec![0u64; n * 72];
    check(unsafe { ffi::blsgpu_pairing_batch(gpu.ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), n, out.as_mut_ptr()) })?;
    Ok(split72(out))
}

/// `multi_miller_loop` (src/pairings.rs:554-603) over (P_i, Q_i): the raw `MillerLoopResult` limbs.  The reference's
/// `G2Prepared` caches 68 line triples per point (19 584 B); the GPU recomputes the lines from the affine point instead.
pub fn multi_miller_loop(gpu: &Gpu, terms: &[(&G1Affine, &G2Affine)]) -> Result<GtLimbs, Error> {
    let p: Vec<G1Affine> = terms.iter().map(|t| *t.0).collect();
    let q: Vec<G2Affine> = terms.iter().map(|t| *t.1).collect();
    let ((g1, f1), (g2, f2)) = (g1_wire(gpu, &p)?, g2_wire(gpu, &q)?);
    let mut out = [0u64; 72];
    check(unsafe { ffi::blsgpu_multi_miller_loop(gpu.ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), terms.len(), out.as_mut_ptr()) })?;
    Ok(GtLimbs(out))
}

/// `G2Prepared` values resident on the device (src/pairings.rs:487-546: the 68 line-coefficient triples of FIXED G2 arguments, computed
/// once by `blsgpu_g2_prepare`); the prepared Miller loops name them by index.
pub struct PreparedG2<'a> { gpu: &'a Gpu, handle: *mut ffi::BlsgpuG2Prepared, len: usize }
impl<'a> PreparedG2<'a> {
    pub fn new(gpu: &'a Gpu, points: &[G2Affine]) -> Result<Self, Error> {
        let (g2, f2) = g2_wire(gpu, points)?;
        let mut handle = core::ptr::null_mut();
        check(unsafe { ffi::blsgpu_g2_prepare(gpu.ctx, g2.as_ptr(), f2.as_ptr(), points.len(), &mut handle) })?;
        Ok(PreparedG2 { gpu, handle, len: points.len() })
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    /// `multi_miller_loop` over terms (P_i, Q_i) with Q_i = table[i_q] (`Ok(index)`) or a fresh point (`Err(&G2Affine)`, lines on the fly)
    pub fn multi_miller_loop(&self, terms: &[(&G1Affine, Result<u32, &G2Affine>)]) -> Result<GtLimbs, Error> {
        let p: Vec<G1Affine> = terms.iter().map(|t| *t.0).collect();
        let q: Vec<G2Affine> = terms.iter().map(|t| match t.1 { Ok(_) => G2Affine::identity(), Err(q) => *q }).collect();
        let qi: Vec<u32> = terms.iter().map(|t| match t.1 { Ok(i) => i, Err(_) => u32::MAX }).collect();
        let ((g1, f1), (g2, f2)) = (g1_wire(self.gpu, &p)?, g2_wire(self.gpu, &q)?);
        let mut out = [0u64; 72];
        check(unsafe { ffi::blsgpu_multi_miller_loop_prepared(self.gpu.ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), qi.as_ptr(), self.handle, terms.len(), out.as_mut_ptr()) })?;
        Ok(GtLimbs(out))
    }
}
impl Drop for PreparedG2<'_> { fn drop(&mut self) { unsafe { ffi::blsgpu_g2_prepared_free(self.handle) } } }

/// N independent `multi_miller_loop(terms).final_exponentiation()` in ONE device call (bulk signature verification: one product of
/// k pairings per equation, src/pairings.rs:554-603, 817-824, 48-176): the `Gt` limbs of every equation; `final_exp = false` returns
/// the raw `MillerLoopResult` limbs.  An equation without terms gives `Gt::identity()` / `MillerLoopResult::default()`.
pub fn multi_miller_loop_many(gpu: &Gpu, equations: &[&[(&G1Affine, &G2Affine)]], final_exp: bool) -> Result<Vec<GtLimbs>, Error> {
    let p: Vec<G1Affine> = equations.iter().flat_map(|e| e.iter().map(|t| *t.0)).collect();
    let q: Vec<G2Affine> = equations.iter().flat_map(|e| e.iter().map(|t| *t.1)).collect();
    let mut off = Vec::with_capacity(equations.len() + 1);
    off.push(0u64);
    for e in equations { off.push(off[off.len() - 1] + e.len() as u64); }
    let ((g1, f1), (g2, f2)) = (g1_wire(gpu, &p)?, g2_wire(gpu, &q)?);
    let mut out = vec![0u64; equations.len() * 72];
    check(unsafe { ffi::blsgpu_multi_miller_loop_many(gpu.ctx, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), off.as_ptr(), equations.len(), final_exp as c_int, out.as_mut_ptr()) })?;
    Ok(split72(out))
}

/// `MillerLoopResult::final_exponentiation` (src/pairings.rs:48-176) for a batch of raw Miller values.
pub fn final_exponentiation(gpu: &Gpu, f: &[GtLimbs]) -> Result<Vec<GtLimbs>, Error> {
    let flat: Vec<u64> = f.iter().flat_map(|g| g.0).collect();
    let mut out = vec![0u64; flat.len()];
    check(unsafe { ffi::blsgpu_final_exponentiation_batch(gpu.ctx, flat.as_ptr(), f.len(), out.as_mut_ptr()) })?;
    Ok(split72(out))
}

/// `MillerLoopResult + MillerLoopResult` / `Gt + Gt` folded over a slice (src/pairings.rs:179-186): how partial products of a
/// sharded `multi_miller_loop` are combined before the single final exponentiation.
pub fn fp12_product(gpu: &Gpu, f: &[GtLimbs]) -> Result<GtLimbs, Error> {
    let flat: Vec<u64> = f.iter().flat_map(|g| g.0).collect();
    let mut out = [0u64; 72];
    check(unsafe { ffi::blsgpu_fp12_product(gpu.ctx, flat.as_ptr(), f.len(), out.as_mut_ptr()) })?;
    Ok(GtLimbs(out))
}

/// `&Gt * &Scalar` (src/pairings.rs:297-322) for n pairs.
pub fn gt_mul_scalar(gpu: &Gpu, gt: &[GtLimbs], scalars: &[Scalar]) -> Result<Vec<GtLimbs>, Error> {
    assert_eq!(gt.len(), scalars.len());
    let flat: Vec<u64> = gt.iter().flat_map(|g| g.0).collect();
    let s = scalar_bytes(scalars);
    let mut out = vec![0u64; flat.len()];
    check(unsafe { ffi::blsgpu_gt_mul_scalar_batch(gpu.ctx, flat.as_ptr(), s.as_ptr(), gt.len(), out.as_mut_ptr()) })?;
    Ok(split72(out))
}

/// `G1Projective::batch_normalize` (src/g1.rs:806-839).  Out of tree the projective limbs are not reachable, so the points
/// travel as affine encodings already; this wrapper exists for symmetry and simply maps `to_affine` (`Curve::to_affine`).
/// The limb-level version (Montgomery's trick on the GPU, `blsgpu_g1_batch_normalize`) is in `in-tree/hip.rs`.
pub fn batch_normalize_g1(points: &[G1Projective]) -> Vec<G1Affine> {
    let mut out = vec![G1Affine::identity(); points.len()];
    G1Projective::batch_normalize(points, &mut out);
    out
}

/// Resident bases (e.g. an SRS): uploaded once, reused by any number of MSMs (`blsgpu_g1_bases_upload`, `blsgpu_g1_msm`).
pub struct ResidentG1<'a> { gpu: &'a Gpu, handle: *mut ffi::BlsgpuBases, len: usize }
impl<'a> ResidentG1<'a> {
    pub fn upload(gpu: &'a Gpu, bases: &[G1Affine]) -> Result<Self, Error> {
        let (xy, inf) = g1_wire(gpu, bases)?;
        let mut handle = core::ptr::null_mut();
        check(unsafe { ffi::blsgpu_g1_bases_upload(gpu.ctx, xy.as_ptr(), inf.as_ptr(), bases.len(), &mut handle) })?;
        Ok(ResidentG1 { gpu, handle, len: bases.len() })
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    /// sum_i scalars[i] * bases[first + i]; the result comes back as projective wire limbs and is normalised on the GPU
    pub fn msm(&self, first: usize, scalars: &[Scalar]) -> Result<G1Projective, Error> {
        let s = scalar_bytes(scalars);
        let mut xyz = [0u64; 18];
        check(unsafe { ffi::blsgpu_g1_msm(self.gpu.ctx, self.handle, first, s.as_ptr(), scalars.len(), xyz.as_mut_ptr()) })?;
        let (mut xy, mut inf, mut enc) = ([0u64; 12], [0u8; 1], [0u8; 96]);
        check(unsafe { ffi::blsgpu_g1_batch_normalize(self.gpu.ctx, xyz.as_ptr(), 1, xy.as_mut_ptr(), inf.as_mut_ptr()) })?;
        check(unsafe { ffi::blsgpu_g1_to_bytes_batch(self.gpu.ctx, xy.as_ptr(), inf.as_ptr(), 1, 0, enc.as_mut_ptr()) })?;
        let aff = Option::<G1Affine>::from(G1Affine::from_uncompressed_unchecked(&enc)).expect("libblsgpu returned an invalid G1 encoding");
        Ok(G1Projective::from(aff))
    }
}
impl Drop for ResidentG1<'_> { fn drop(&mut self) { unsafe { ffi::blsgpu_bases_free(self.handle) } } }

/// Every listed GPU of the node behind one handle (`blsgpu_group_*`): MSMs, batches of pairings and `multi_miller_loop`s are dealt to
/// the members in contiguous slices -- one context and one host thread per member inside the library -- and the members' partial results
/// (one group element each) are folded with the reference's own operators, `Sum` (src/g1.rs:161-171) and `MillerLoopResult +
/// MillerLoopResult` (src/pairings.rs:179-186), followed by ONE final exponentiation.  The decoders run on member 0's context.
pub struct GpuGroup { group: *mut ffi::BlsgpuGroup, first: Gpu }
unsafe impl Send for GpuGroup {}
impl Drop for GpuGroup { fn drop(&mut self) { unsafe { ffi::blsgpu_group_destroy(self.group) } } }
impl GpuGroup {
    pub fn new(devices: &[i32]) -> Result<GpuGroup, Error> {
        let mut group = core::ptr::null_mut();
        check(unsafe { ffi::blsgpu_group_create(devices.as_ptr(), devices.len() as c_int, &mut group) })?;
        // a context of its own for the byte decoders (the members' contexts belong to the group's worker threads during a call)
        match Gpu::new(devices[0]) {
            Ok(first) => Ok(GpuGroup { group, first }),
            Err(e) => { unsafe { ffi::blsgpu_group_destroy(group) }; Err(e) }
        }
    }
    pub fn all_devices() -> Result<GpuGroup, Error> {
        let d: Vec<i32> = (0..Gpu::device_count().max(1)).collect();
        GpuGroup::new(&d)
    }
    pub fn len(&self) -> usize { unsafe { ffi::blsgpu_group_size(self.group) as usize } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    /// `bases.iter().zip(scalars).map(|(p, s)| p * s).sum::<G1Projective>()` sharded over the members
    pub fn msm_g1(&self, bases: &[G1Affine], scalars: &[Scalar]) -> Result<G1Projective, Error> {
        assert_eq!(bases.len(), scalars.len());
        let ((xy, inf), s) = (g1_wire(&self.first, bases)?, scalar_bytes(scalars));
        let mut gb = core::ptr::null_mut();
        check(unsafe { ffi::blsgpu_group_bases_upload(self.group, 1, xy.as_ptr(), inf.as_ptr(), bases.len(), &mut gb) })?;
        let mut xyz = [0u64; 18];
        let rc = unsafe { ffi::blsgpu_g1_msm_sharded(self.group, gb, s.as_ptr(), bases.len(), xyz.as_mut_ptr()) };
        unsafe { ffi::blsgpu_group_bases_free(gb) };
        check(rc)?;
        let (mut axy, mut ainf, mut enc) = ([0u64; 12], [0u8; 1], [0u8; 96]);
        check(unsafe { ffi::blsgpu_g1_batch_normalize(self.first.ctx, xyz.as_ptr(), 1, axy.as_mut_ptr(), ainf.as_mut_ptr()) })?;
        check(unsafe { ffi::blsgpu_g1_to_bytes_batch(self.first.ctx, axy.as_ptr(), ainf.as_ptr(), 1, 0, enc.as_mut_ptr()) })?;
        Ok(G1Projective::from(Option::<G1Affine>::from(G1Affine::from_uncompressed_unchecked(&enc)).expect("libblsgpu returned an invalid G1 encoding")))
    }
    /// out[i] = e(p[i], q[i]), index slices per member
    pub fn pairing_batch(&self, p: &[G1Affine], q: &[G2Affine]) -> Result<Vec<GtLimbs>, Error> {
        assert_eq!(p.len(), q.len());
        let ((g1, f1), (g2, f2)) = (g1_wire(&self.first, p)?, g2_wire(&self.first, q)?);
        let mut out = vec![0u64; p.len() * 72];
        check(unsafe { ffi::blsgpu_pairing_batch_sharded(self.group, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), p.len(), out.as_mut_ptr()) })?;
        Ok(split72(out))
    }
    /// `multi_miller_loop(terms).final_exponentiation()`: member-local products, one fold, ONE final exponentiation
    pub fn multi_miller_loop_final_exp(&self, terms: &[(&G1Affine, &G2Affine)]) -> Result<GtLimbs, Error> {
        let p: Vec<G1Affine> = terms.iter().map(|t| *t.0).collect();
        let q: Vec<G2Affine> = terms.iter().map(|t| *t.1).collect();
        let ((g1, f1), (g2, f2)) = (g1_wire(&self.first, &p)?, g2_wire(&self.first, &q)?);
        let mut out = [0u64; 72];
        check(unsafe { ffi::blsgpu_multi_miller_loop_sharded(self.group, g1.as_ptr(), f1.as_ptr(), g2.as_ptr(), f2.as_ptr(), terms.len(), 1, out.as_mut_ptr()) })?;
        Ok(GtLimbs(out))
    }
}

/// `Curve::to_affine` convenience for callers that want affine results.
pub fn to_affine_g1(p: &G1Projective) -> G1Affine { p.to_affine() }
