/* bls12_381_hip.h -- C ABI of the MI355X-native BLS12-381 hot path (libblsgpu.so).
 *
 * This is the drop-in boundary for the data-parallel path of zkcrypto/bls12_381 v0.8.0: multi-scalar
 * multiplication over G1/G2 and batched Miller-loop / final-exponentiation pairings.  The reference has
 * no FFI (it is `#![deny(unsafe_code)]`, src/lib.rs:17); each entry point below names the Rust
 * operator / trait method it replaces.  INTEGRATION.md shows the `extern "C"` block and the wrapper a
 * maintainer would add on the Rust side.
 *
 * Wire formats (all little-endian, plain memory, caller-owned):
 *   Fp      6 x u64   canonical Montgomery limbs, R = 2^384       (`Fp([u64;6])`, src/fp.rs:11-15)
 *   Fp2     c0 | c1                                               (src/fp2.rs:11-14)
 *   Fp12    c0.c0.c0 c0.c0.c1 c0.c1.c0 ... c1.c2.c1 (72 u64)      (src/fp12.rs:13-16, src/fp6.rs:12-16)
 *   Scalar  32 bytes, little-endian canonical integer in [0, r)   (`Scalar::to_bytes`, src/scalar.rs:284-296)
 *   G1 affine      x | y          (12 u64)  + out-of-band infinity byte   (src/g1.rs:28-32)
 *   G1 projective  X | Y | Z      (18 u64), x = X/Z, identity (0:1:0)     (src/g1.rs:442-446,605-611)
 *   G2 affine      x.c0 x.c1 y.c0 y.c1 (24 u64) + infinity byte           (src/g2.rs)
 *   G2 projective  36 u64
 * Results are exact group / field elements in canonical limbs: after affine conversion (G1/G2) or as they
 * are (Gt) they are bit-identical to what the reference computes on the same inputs.
 *
 * Every function returns BLSGPU_OK (0) or a negative error code; nothing is retained from caller buffers
 * after return.  A context owns one device, one HIP stream and its scratch memory; it is not re-entrant
 * (use one context per host thread).  The rule is enforced: an entry point called while another host thread is inside an
 * entry point of the SAME context returns BLSGPU_ERR_ARG ("the context is in use by another host thread") and touches
 * nothing; contexts are independent of one another (tests: two contexts on two threads; one context from two threads).
 */
#ifndef BLS12_381_HIP_H
#define BLS12_381_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLSGPU_OK 0
#define BLSGPU_ERR_HIP (-1)      /* a HIP runtime call failed (blsgpu_last_error gives the text) */
#define BLSGPU_ERR_ARG (-2)      /* bad argument (NULL pointer, length mismatch, unsupported size) */
#define BLSGPU_ERR_NODEV (-3)    /* no usable gfx950 device */

typedef struct blsgpu_ctx blsgpu_ctx;
typedef struct blsgpu_bases blsgpu_bases;   /* device-resident base points in the library's internal form */

/* ---- context ------------------------------------------------------------------------------------ */
int blsgpu_create(int device, blsgpu_ctx** out);
void blsgpu_destroy(blsgpu_ctx* ctx);
const char* blsgpu_last_error(void);
int blsgpu_device_count(void);
/* Use an existing HIP stream (e.g. torch's current stream) for all work of this context; NULL restores
 * the context's own stream. */
int blsgpu_set_stream(blsgpu_ctx* ctx, void* hip_stream);
/* Wait for everything queued on the context.  Scalars must be canonical (< r; `Scalar::to_bytes` produces nothing else): a
 * synchronous entry point (`*_msm`, `*_msm_many`, `*_msm_bytes`, `*_msm_host`, `*_mul_batch`) that meets one returns
 * BLSGPU_ERR_ARG ITSELF -- its verdict travels with its result; the asynchronous `*_device` entry points cannot, so their
 * verdict is kept in a sticky flag that THIS call reports (BLSGPU_ERR_ARG) and clears.  A valid synchronous call is never blamed
 * for an earlier asynchronous one. */
int blsgpu_synchronize(blsgpu_ctx* ctx);
/* MSM pipelining.  An MSM is three phases with different bottlenecks: digit sort (LDS atomics), bucket
 * accumulation (integer VALU) and a latency-bound tail (bucket reduction + window combine: a few wavefronts
 * for ~ms).  The library runs them on internal streams chained by events.  By default the context's stream
 * waits for the tail, so results are ordered like any other work on that stream.  With pipelining on,
 * `*_msm_device` only records a dependency on the work already queued on the context's stream (the producer
 * of the scalars) and returns; sort, accumulation and tail of up to four calls then overlap one another.
 * The caller must keep each call's scalars and output buffer untouched, and use `blsgpu_join` (stream-level
 * wait, no host sync) or `blsgpu_synchronize` before consuming results. */
int blsgpu_set_pipelining(blsgpu_ctx* ctx, int enabled);
int blsgpu_join(blsgpu_ctx* ctx);
/* As blsgpu_join, but leaves the `lag` most recent MSM calls in flight (lag = 1: wait for everything except
 * the call just enqueued -- the pattern "enqueue MSM i, consume result i-1"). */
int blsgpu_join_lag(blsgpu_ctx* ctx, int lag);

/* ---- resident bases ------------------------------------------------------------------------------- */
/* Upload n affine points (host memory).  `infinity` may be NULL (no identities).
 * Replaces holding a `&[G1Affine]` / `&[G2Affine]` on the Rust side. */
int blsgpu_g1_bases_upload(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, size_t n, blsgpu_bases** out);
int blsgpu_g2_bases_upload(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, size_t n, blsgpu_bases** out);
/* Same, but `xy` / `infinity` are device pointers (e.g. torch tensors' data_ptr). */
int blsgpu_g1_bases_from_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, size_t n, blsgpu_bases** out);
int blsgpu_g2_bases_from_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, size_t n, blsgpu_bases** out);
/* bases[i] = [k_i] * generator for n 32-byte LE scalars (device-side fixed-base multiplication; used to
 * build synthetic inputs and SRS-style tables).  `group` is 1 or 2. */
int blsgpu_bases_from_scalars(blsgpu_ctx* ctx, int group, const uint8_t* scalars, size_t n, blsgpu_bases** out);
/* Optional, for bases that are reused (an SRS): build resident window-shifted tables [2^(c*w)] P_i (W x the memory;
 * W = ceil(256/c), c = window_bits, 0 -> 20).  Later MSMs over these bases put every window into ONE bucket set:
 * no window combine, a W-times smaller bucket reduction, fewer windows.  Requires n * W <= 2^24.  Results are the
 * same group elements.  The default width is the tuned one (2^20 points: 2.5 ms per pipelined MSM against 2.8 without tables); 16 works
 * as well, the widths between and 21 are correct but 1.3-1.6x slower (the sort's bin geometry). */
int blsgpu_bases_precompute(blsgpu_ctx* ctx, blsgpu_bases* b, int window_bits);
size_t blsgpu_bases_len(const blsgpu_bases* b);
/* Subgroup contract of the MSM.  The reference's `multiply` (src/g1.rs:754-774, src/g2.rs:825-845) is plain
 * double-and-add and therefore defined for EVERY curve point, including the on-curve points outside the prime-order
 * subgroup that `from_uncompressed_unchecked` / `from_compressed_unchecked` (src/g1.rs:273-322, 336-390) hand out.  The
 * fast MSM path splits scalars with the endomorphisms phi (G1) / psi (G2), whose eigenvalue relations hold only on the
 * subgroup.  Every upload therefore runs the reference's own `is_on_curve() & is_torsion_free()` (src/g1.rs:396-416,
 * src/g2.rs:475-489) over the set once, on the device: sets that pass keep endomorphism images and take the fast path;
 * a set with ANY point outside the subgroup runs on plain 256-bit windows with complete addition formulas, which
 * compute sum [s_i]P_i for arbitrary curve points exactly as the reference's `multiply` + `Sum` do.  Either way the
 * result is the reference's group element.  Bases built by blsgpu_bases_from_scalars are multiples of the generator and
 * skip the test.  A caller that already knows its points are in the subgroup (e.g. `G1Affine` values that came from
 * the checked decoders) may skip the per-upload test with blsgpu_set_assume_subgroup(ctx, 1); with that flag set,
 * results for off-subgroup inputs are unspecified.  Default: 0 (test every upload).
 * blsgpu_bases_subgroup_state: 1 = verified (or built as [k]G), 2 = assumed by the caller, 0 = at least one point is
 * outside the subgroup (plain windows), 3 = not tested: the set is beyond the size the split supports (G2: more than 2^22
 * points) and runs on plain windows in any case.
 * The ONE-SHOT entry points (blsgpu_g{1,2}_msm_host, blsgpu_g{1,2}_msm_bytes) upload a set for a single MSM: the test would
 * cost several times the MSM it speeds up, so they skip it and run on plain windows (exact for every curve point, state 3)
 * unless blsgpu_set_assume_subgroup(ctx, 1) is set, in which case they take the fast path untested.
 * Synchronisation: an upload that runs the test (`*_bases_upload`, `*_bases_from_device` with the default
 * assume_subgroup = 0) reads the verdict back and therefore SYNCHRONISES the context's stream (blsgpu_set_stream: the
 * caller's stream) -- it blocks the host and cannot be captured into a graph.  With blsgpu_set_assume_subgroup(ctx, 1)
 * `*_bases_from_device` only enqueues work and returns. */
int blsgpu_set_assume_subgroup(blsgpu_ctx* ctx, int enabled);
int blsgpu_bases_subgroup_state(const blsgpu_bases* b);
/* Read points [first, first+count) back in wire format (xy: count*12 or count*24 u64; infinity: count bytes). */
int blsgpu_bases_download(blsgpu_ctx* ctx, const blsgpu_bases* b, size_t first, size_t count, uint64_t* xy, uint8_t* infinity);
void blsgpu_bases_free(blsgpu_bases* b);

/* ---- multi-scalar multiplication ------------------------------------------------------------------- */
/* out = sum_{i<n} scalars[i] * bases[first + i]   as a projective point (18 / 36 u64).
 * Replaces  bases.iter().zip(scalars).map(|(p, s)| p * s).sum::<G1Projective>()
 * (`Mul<&Scalar> for &G1Affine` src/g1.rs:573-579 -> `multiply` :754-774, `Sum` :161-171; G2: src/g2.rs:626-632,
 * 825-845, 162-172).  n = 0 yields the identity (0:1:0). */
int blsgpu_g1_msm(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const uint8_t* scalars, size_t n, uint64_t out_xyz[18]);
int blsgpu_g2_msm(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const uint8_t* scalars, size_t n, uint64_t out_xyz[36]);
/* Asynchronous variants: scalars and the result live in device memory, work is enqueued on the
 * context's stream and the call returns without synchronising. */
int blsgpu_g1_msm_device(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, void* d_out_xyz);
int blsgpu_g2_msm_device(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, void* d_out_xyz);
/* Scalars as the reference STORES them.  Every entry point above (and blsgpu_*_mul_batch*, blsgpu_gt_mul_scalar_batch*, the `*_many`,
 * `*_host` and `*_sharded` MSMs) reads n x 32 bytes per scalar vector; what the 32 bytes are is the context's scalar form:
 *   BLSGPU_SCALAR_BYTES (default)  the canonical little-endian integer, `Scalar::to_bytes()` (src/scalar.rs:284-296), the bit source
 *                                  of `multiply` (src/g1.rs:559-561);
 *   BLSGPU_SCALAR_MONT             the four u64 Montgomery limbs of `Scalar([u64; 4])` (src/scalar.rs:23-27), i.e. the memory of a
 *                                  `&[Scalar]`: `to_bytes` = one `montgomery_reduce` (src/scalar.rs:506-550) then runs on the device,
 *                                  fused into the kernels that decompose the scalars, and an in-tree `msm(&[G1Affine], &[Scalar])`
 *                                  passes its slice as it is (2^20 host-side `to_bytes` calls cost several times the MSM).
 * Limbs / bytes that no `Scalar` can hold (>= r) are reported like non-canonical bytes (BLSGPU_ERR_ARG from the synchronous entry
 * points / blsgpu_synchronize).  blsgpu_set_scalar_form changes the form for all later calls on the context (members of a device
 * group: through blsgpu_group_ctx); the `*_mont` entry points are the four MSMs above with BLSGPU_SCALAR_MONT for that one call.
 * blsgpu_g{1,2}_msm_bytes always take `to_bytes()` output. */
#define BLSGPU_SCALAR_BYTES 0
#define BLSGPU_SCALAR_MONT 1
int blsgpu_set_scalar_form(blsgpu_ctx* ctx, int form);
int blsgpu_g1_msm_mont(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const uint64_t* scalars, size_t n, uint64_t out_xyz[18]);
int blsgpu_g2_msm_mont(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int blsgpu_g1_msm_mont_device(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, void* d_out_xyz);
int blsgpu_g2_msm_mont_device(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, void* d_out_xyz);
/* k MSMs over the same resident bases (k scalar vectors of n x 32 bytes back to back -> k projective results back to back),
 * e.g. commitments to k polynomials under one SRS.  Synchronous like blsgpu_g1_msm, but the k calls run through the
 * library's pipeline (sort / accumulation / tail of consecutive MSMs overlap): the sustained rate of the asynchronous API
 * without managing streams. */
int blsgpu_g1_msm_many(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const uint8_t* scalars, size_t n, size_t k, uint64_t* out_xyz);
int blsgpu_g2_msm_many(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const uint8_t* scalars, size_t n, size_t k, uint64_t* out_xyz);
int blsgpu_g1_msm_many_device(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, size_t k, void* d_out_xyz);
int blsgpu_g2_msm_many_device(blsgpu_ctx* ctx, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, size_t k, void* d_out_xyz);
/* Opt-in cache for callers that pass the SAME base array to the one-shot entry points again and again (a drop-in `msm(&bases, &scalars)`
 * over an SRS: the reference's surface has no resident handle).  entries = 0 (default) switches it off and drops what is cached.  An
 * array is recognised by its length and a fingerprint of 64 evenly spaced points, so the caller promises NOT to modify an array it
 * passes again.  First sight: the one-shot path as always.  Second sight: the set is uploaded as resident bases (subgroup test,
 * endomorphism images) and kept; later calls only move their scalars and run on the path of blsgpu_g1_msm. */
int blsgpu_set_bases_cache(blsgpu_ctx* ctx, int entries);
/* HAZARD of the cache above: with the default fingerprint a caller that reuses a same-length buffer and changes only points that are
 * not among the 65 sampled ones gets an MSM over the STALE resident bases, with no error.  blsgpu_set_bases_cache_verify(ctx, 1) makes the
 * cache recognise an array by a hash of every word instead (four host threads, ~3 ms per 2^20 G1 points per call): use it unless the
 * arrays are known to be immutable (an SRS loaded once).  Switching the mode drops what is cached. */
int blsgpu_set_bases_cache_verify(blsgpu_ctx* ctx, int enabled);
/* One-shot convenience: upload, multiply, free.  No subgroup test and no endomorphism split (plain windows: exact for every
 * curve point) unless blsgpu_set_assume_subgroup(ctx, 1) -- see the subgroup contract above.  For a set used more than once
 * upload it (blsgpu_g1_bases_upload) and call blsgpu_g1_msm. */
int blsgpu_g1_msm_host(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, const uint8_t* scalars, size_t n, uint64_t out_xyz[18]);
int blsgpu_g2_msm_host(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, const uint8_t* scalars, size_t n, uint64_t out_xyz[36]);
/* The same on the reference's PUBLIC encodings, for a wrapper crate that cannot reach limbs (`pub(crate)`, src/g1.rs:28-32):
 * bases = n x 96 (G1) / n x 192 (G2) bytes of `to_uncompressed()` (decoded like `from_uncompressed_unchecked`,
 * src/g1.rs:271-322), scalars = n x 32 bytes of `Scalar::to_bytes()`, out = `to_uncompressed()` of the affine sum. */
int blsgpu_g1_msm_bytes(blsgpu_ctx* ctx, const uint8_t* bases_uncompressed, const uint8_t* scalars, size_t n, uint8_t out[96]);
int blsgpu_g2_msm_bytes(blsgpu_ctx* ctx, const uint8_t* bases_uncompressed, const uint8_t* scalars, size_t n, uint8_t out[192]);
/* Window width c (bits) used by Pippenger; 0 = automatic, otherwise 4..16 (the LDS counting sort keys on 8 + 7 bits of
 * |digit|; wider windows exist only with resident tables, blsgpu_bases_precompute, which carry their own width).  With the
 * endomorphism split active (see the subgroup contract above) the width applies to the 128-bit (G1) / 64-bit (G2)
 * sub-scalars: c = 16 then means 8 (G1) / 4 (G2) windows instead of 16. */
int blsgpu_set_msm_window(blsgpu_ctx* ctx, int c);

/* ---- batched variable-base scalar multiplication ------------------------------------------------------------ */
/* out[i] = scalars[i] * points[i] for n independent (point, scalar) pairs: n affine points in, n projective points out
 * (18 / 36 u64 each).  Replaces n evaluations of `&G1Affine * &Scalar` / `&G2Affine * &Scalar` (src/g1.rs:556-594 ->
 * `multiply` :754-774; src/g2.rs:609-647, 825-845; the "scalar multiplication" points of benches/groups.rs:44,89,113,158).
 * Signed 4-bit windows over complete addition formulas: exact for every curve point (identity, scalars 0 and r - 1, points
 * outside the prime-order subgroup) -- no subgroup precondition.  The projective representative differs from the one the
 * reference's double-and-add produces; the group element (affine coordinates) is the same.  `infinity` may be NULL.
 * With blsgpu_set_assume_subgroup(ctx, 1) -- the caller vouches that every point lies in the prime-order subgroup, e.g. values
 * from the checked decoders -- the scalars are split with the endomorphisms as in the MSM (G1: two 127-bit halves, half the
 * doublings; G2: four 63-bit digits over psi, a quarter of the doublings); results for off-subgroup points are then
 * unspecified, as for the MSM. */
int blsgpu_g1_mul_batch(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, const uint8_t* scalars, size_t n, uint64_t* out_xyz);
int blsgpu_g2_mul_batch(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, const uint8_t* scalars, size_t n, uint64_t* out_xyz);
/* Same with device pointers, asynchronous on the context's stream. */
int blsgpu_g1_mul_batch_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, const void* d_scalars, size_t n, void* d_out_xyz);
int blsgpu_g2_mul_batch_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, const void* d_scalars, size_t n, void* d_out_xyz);
/* The same four with the scalars as Montgomery limbs (BLSGPU_SCALAR_MONT for that one call; `Mul<&Scalar>` calls `to_bytes` per
 * product, src/g1.rs:556-562). */
int blsgpu_g1_mul_batch_mont(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, const uint64_t* scalars, size_t n, uint64_t* out_xyz);
int blsgpu_g2_mul_batch_mont(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, const uint64_t* scalars, size_t n, uint64_t* out_xyz);
int blsgpu_g1_mul_batch_mont_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, const void* d_scalars, size_t n, void* d_out_xyz);
int blsgpu_g2_mul_batch_mont_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, const void* d_scalars, size_t n, void* d_out_xyz);

/* ---- group helpers ----------------------------------------------------------------------------------- */
/* out = sum of n projective points (`Sum for G1Projective`, src/g1.rs:161-171) -- the fold used after the
 * cross-GPU all-gather of per-rank partial results. */
int blsgpu_g1_sum(blsgpu_ctx* ctx, const uint64_t* xyz, size_t n, uint64_t out_xyz[18]);
int blsgpu_g2_sum(blsgpu_ctx* ctx, const uint64_t* xyz, size_t n, uint64_t out_xyz[36]);
/* Same with device pointers, asynchronous on the context's stream (no host round trip after the all-gather). */
int blsgpu_g1_sum_device(blsgpu_ctx* ctx, const void* d_xyz, size_t n, void* d_out_xyz);
int blsgpu_g2_sum_device(blsgpu_ctx* ctx, const void* d_xyz, size_t n, void* d_out_xyz);
/* Projective -> affine for n points (`G1Projective::batch_normalize`, src/g1.rs:806-839; `G1Affine::from`,
 * :49-63).  Identity maps to x = 0, y = 1 (Montgomery one), infinity = 1. */
int blsgpu_g1_batch_normalize(blsgpu_ctx* ctx, const uint64_t* xyz, size_t n, uint64_t* xy, uint8_t* infinity);
int blsgpu_g2_batch_normalize(blsgpu_ctx* ctx, const uint64_t* xyz, size_t n, uint64_t* xy, uint8_t* infinity);

/* ---- batched point (de)serialisation and validation (the step in front of the hot path) ------------------ */
/* Decode n points from the reference's byte encodings (src/notes/serialization.rs): 48/96 B compressed or 96/192 B
 * uncompressed.  checked = 0: `from_compressed_unchecked` / `from_uncompressed_unchecked` (src/g1.rs:273-322, 336-390);
 * checked = 1: `from_compressed` (subgroup check, :326-332) / `from_uncompressed` (on-curve + subgroup, :264-267).
 * ok[i] = 1 exactly where the reference returns CtOption::some; rejected entries decode to the identity. */
int blsgpu_g1_from_bytes_batch(blsgpu_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int checked, uint64_t* xy, uint8_t* infinity, uint8_t* ok);
int blsgpu_g2_from_bytes_batch(blsgpu_ctx* ctx, const uint8_t* bytes, size_t n, int compressed, int checked, uint64_t* xy, uint8_t* infinity, uint8_t* ok);
/* Encode n affine points: `to_compressed` / `to_uncompressed` (src/g1.rs:221-260, src/g2.rs:254-299). */
int blsgpu_g1_to_bytes_batch(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, size_t n, int compressed, uint8_t* out);
int blsgpu_g2_to_bytes_batch(blsgpu_ctx* ctx, const uint64_t* xy, const uint8_t* infinity, size_t n, int compressed, uint8_t* out);

/* ---- pairings ------------------------------------------------------------------------------------------ */
/* Which kernels a call with n items (pairings, Miller loops or final exponentiations) runs on this context:
 *   256 = wide  (one item per 1024- or 512-lane workgroup: the small-batch latency path, n <= 1536; needs the generated
 *                program file `wide_prog.bin` next to the library -- build step of __graft_entry__.py -- or at
 *                $BLSGPU_WIDE_PROG),
 *     4 = quad  (one item per four lanes: the throughput path),
 *     2 = lane pair (the layout of rounds 1-2, kept for A/B runs).
 * BLSGPU_PAIRING_LAYOUT=wide|quad|pair|auto in the environment at blsgpu_create fixes the choice; with "wide" a missing
 * or mismatching program file makes the pairing entry points FAIL (BLSGPU_ERR_ARG) instead of silently running another
 * kernel; with "auto" (the default) the quad kernels take every size then, this function says so, and the library prints ONE
 * line to stderr the first time it happens (a single pairing costs ~6 ms instead of ~1.1 ms then).  Any other value of the
 * variable makes blsgpu_create fail with BLSGPU_ERR_ARG.  All layouts return limb-identical results. */
int blsgpu_pairing_layout(blsgpu_ctx* ctx, size_t n);
/* "" when the wide programs are loaded on this context, otherwise the reason they are not (file missing, stale format
 * version, generated for another kernel configuration, library path unknown in a static link: set $BLSGPU_WIDE_PROG). */
const char* blsgpu_wide_status(blsgpu_ctx* ctx);      /* (the pointer is valid until the next call on this context) */
/* out[i] = pairing(g1[i], g2[i]) for n independent pairs (`pairing`, src/pairings.rs:607-653; 72 u64 each).
 * An identity on either side yields Gt::identity() = Fp12::one(), as the reference does. */
int blsgpu_pairing_batch(blsgpu_ctx* ctx, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t n, uint64_t* out_gt);
/* out[i] = raw Miller-loop value of pair i (a `MillerLoopResult`, src/pairings.rs:26) -- bit-identical to the
 * reference's because the same line formulas are used (src/pairings.rs:696-770). */
int blsgpu_miller_loop_batch(blsgpu_ctx* ctx, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t n, uint64_t* out_f);
/* out = prod_i ML(g1[i], g2[i])  (`multi_miller_loop`, src/pairings.rs:554-603; pairs with an identity are
 * skipped).  n = 0 yields Fp12::one() (`MillerLoopResult::default`, :28-32). */
int blsgpu_multi_miller_loop(blsgpu_ctx* ctx, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t n, uint64_t out_f[72]);
/* N independent `multi_miller_loop`s in one call -- the shape of bulk signature verification, where generic `E: MultiMillerLoop`
 * callers evaluate one product of k pairings per equation (src/pairings.rs:554-603, 817-824; the test pattern :871-921).
 * Segment s = terms offsets[s] .. offsets[s+1] of the g1 / g2 arrays (CSR: nseg + 1 non-decreasing offsets, offsets[0] = 0).
 *   final_exp = 0: out[s] = multi_miller_loop(terms of s)                        (a `MillerLoopResult`, 72 u64)
 *   final_exp != 0: out[s] = multi_miller_loop(terms of s).final_exponentiation() (a `Gt`, 72 u64)
 * Terms with an identity on either side are skipped as the reference does (:566-569); an empty segment yields
 * `MillerLoopResult::default()` = Fp12::one() (:28-32), whose final exponentiation is `Gt::identity()`.  One batched Miller
 * kernel over all terms, one segmented Fp12 product, one batched final exponentiation; few terms take the wide path; from
 * 49 152 segments of at most 8 terms on, every segment shares one accumulator (the reference's own schedule) instead.  For
 * ONE product over very many terms blsgpu_multi_miller_loop is the faster entry point. */
int blsgpu_multi_miller_loop_many(blsgpu_ctx* ctx, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, const uint64_t* offsets,
                                  size_t nseg, int final_exp, uint64_t* out);
/* Device-pointer variant, asynchronous on the context's stream: points, flags, offsets (nseg + 1 u64) and the output in device
 * memory.  `total_terms` = offsets[nseg] and `max_seg_terms` = an upper bound of the segment lengths (0 = unknown) are passed
 * by value because the host cannot read the offsets without a synchronisation; offsets beyond total_terms are clamped, and a
 * segment longer than a max_seg_terms <= 8 the caller passed is reported by the next blsgpu_synchronize (BLSGPU_ERR_ARG; the
 * value of that segment is unspecified) -- the sticky flag of the asynchronous entry points, see blsgpu_synchronize. */
int blsgpu_multi_miller_loop_many_device(blsgpu_ctx* ctx, const void* d_g1_xy, const void* d_g1_inf, const void* d_g2_xy, const void* d_g2_inf, const void* d_offsets,
                                         size_t nseg, size_t total_terms, size_t max_seg_terms, int final_exp, void* d_out);
/* out[i] = final_exponentiation(in[i])  (`MillerLoopResult::final_exponentiation`, src/pairings.rs:48-176). */
int blsgpu_final_exponentiation_batch(blsgpu_ctx* ctx, const uint64_t* in_f, size_t n, uint64_t* out_gt);
/* out = prod of n Fp12 values (`MillerLoopResult + MillerLoopResult`, src/pairings.rs:179-186; `Gt + Gt`). */
int blsgpu_fp12_product(blsgpu_ctx* ctx, const uint64_t* in_f, size_t n, uint64_t out_f[72]);
/* `&Gt * &Scalar` (src/pairings.rs:297-322) for n (element, scalar) pairs: out[i] = gt[i] "times" scalars[i], i.e. the
 * Fp12 power by the canonical little-endian 32-byte scalar (double-and-add over its 255 low bits). */
int blsgpu_gt_mul_scalar_batch(blsgpu_ctx* ctx, const uint64_t* gt, const uint8_t* scalars, size_t n, uint64_t* out);
/* Device-pointer variants (inputs/outputs in device memory, asynchronous on the context's stream). */
int blsgpu_pairing_batch_device(blsgpu_ctx* ctx, const void* d_g1_xy, const void* d_g1_inf, const void* d_g2_xy, const void* d_g2_inf, size_t n, void* d_out_gt);
int blsgpu_multi_miller_loop_device(blsgpu_ctx* ctx, const void* d_g1_xy, const void* d_g1_inf, const void* d_g2_xy, const void* d_g2_inf, size_t n, void* d_out_f);
/* raw Miller values of n pairs, final exponentiation of n values and the product of n values, device to device: the pieces a
 * sharded `multi_miller_loop` needs around its single exchange (rank-local product -> all-gather -> fold -> ONE final
 * exponentiation; src/pairings.rs:179-186, :48-176) and that a sharded batch of independent pairings keeps on the device. */
int blsgpu_miller_loop_batch_device(blsgpu_ctx* ctx, const void* d_g1_xy, const void* d_g1_inf, const void* d_g2_xy, const void* d_g2_inf, size_t n, void* d_out_f);
int blsgpu_final_exponentiation_device(blsgpu_ctx* ctx, const void* d_in_f, size_t n, void* d_out_gt);
int blsgpu_fp12_product_device(blsgpu_ctx* ctx, const void* d_in_f, size_t n, void* d_out_f);

/* ---- G2Prepared: line coefficients of FIXED G2 arguments resident on the device ----------------------------------------- */
/* `G2Prepared` (src/pairings.rs:487-502) holds the 68 coefficient triples of a point's doubling / addition steps, computed once
 * (`From<G2Affine> for G2Prepared`, :504-546), so that every later `multi_miller_loop` only EVALUATES them at P (:554-603 with
 * `ell`, :696-707): ~2.9 k field multiplications per term instead of ~6.9 k.  A `blsgpu_g2_prepared` is a device-resident table
 * of m such points in the library's limb form (26 112 B per point); verification keys (Groth16: three of four G2 arguments fixed)
 * and fixed generators (BLS) are prepared once and named by index afterwards.  The identity keeps its flag and is skipped by the
 * consumers as the reference skips it (:566-569).  The handle belongs to the context's device and outlives the context's calls;
 * free it with blsgpu_g2_prepared_free (after the work that reads it has completed). */
typedef struct blsgpu_g2_prepared blsgpu_g2_prepared;
int blsgpu_g2_prepare(blsgpu_ctx* ctx, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t m, blsgpu_g2_prepared** out);
/* the same with the points (24 u64 each) and flags in device memory; asynchronous on the context's stream */
int blsgpu_g2_prepare_device(blsgpu_ctx* ctx, const void* d_g2_xy, const void* d_g2_inf, size_t m, blsgpu_g2_prepared** out);
size_t blsgpu_g2_prepared_len(const blsgpu_g2_prepared* p);
void blsgpu_g2_prepared_free(blsgpu_g2_prepared* p);
/* The stored coefficients of point `index` in the reference's own value format -- `coeffs: Vec<(Fp2, Fp2, Fp2)>`, 68 triples of
 * 3 x 12 u64 canonical Montgomery limbs (2 448 u64) -- and its `infinity` flag: what an in-tree `G2Prepared` would hold, and the
 * parity hook against the oracle's `G2Prepared::from`. */
int blsgpu_g2_prepared_coeffs(blsgpu_ctx* ctx, const blsgpu_g2_prepared* p, size_t index, uint64_t* out_coeffs, uint8_t* out_inf);
/* `multi_miller_loop(&[(&G1Affine, &G2Prepared)])` (src/pairings.rs:554-603) with prepared and unprepared terms mixed: term i pairs
 * g1[i] with table point q_index[i], or -- q_index[i] = BLSGPU_UNPREPARED (or q_index = NULL: every term) -- with g2[i], whose
 * lines are computed on the fly as in blsgpu_multi_miller_loop (g2 may be NULL when every term is prepared).  out = the raw
 * `MillerLoopResult`, limb-identical to the unprepared entry points' and to the reference's.  An index outside the table is an
 * argument error (host-pointer forms) / reported by the next blsgpu_synchronize (device-pointer forms, where the term is skipped). */
#define BLSGPU_UNPREPARED 0xffffffffu
int blsgpu_multi_miller_loop_prepared(blsgpu_ctx* ctx, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, const uint32_t* q_index,
                                      const blsgpu_g2_prepared* prepared, size_t n, uint64_t out_f[72]);
int blsgpu_multi_miller_loop_prepared_device(blsgpu_ctx* ctx, const void* d_g1_xy, const void* d_g1_inf, const void* d_g2_xy, const void* d_g2_inf, const void* d_q_index,
                                             const blsgpu_g2_prepared* prepared, size_t n, void* d_out_f);
/* N independent products in one call (blsgpu_multi_miller_loop_many with prepared terms): segment s = terms offsets[s] ..
 * offsets[s+1]; every segment runs the reference's own schedule on one accumulator -- per step each term multiplies its line in,
 * one squaring for all -- so a verification equation with k terms pays 62 squarings, not 62 k, and a prepared term only its
 * `ell`s.  final_exp as in blsgpu_multi_miller_loop_many.  Segments of any length are accepted (more than 8 terms: several passes,
 * multiplied together); for ONE long product use blsgpu_multi_miller_loop_prepared. */
int blsgpu_multi_miller_loop_prepared_many(blsgpu_ctx* ctx, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, const uint32_t* q_index,
                                           const blsgpu_g2_prepared* prepared, const uint64_t* offsets, size_t nseg, int final_exp, uint64_t* out);
int blsgpu_multi_miller_loop_prepared_many_device(blsgpu_ctx* ctx, const void* d_g1_xy, const void* d_g1_inf, const void* d_g2_xy, const void* d_g2_inf, const void* d_q_index,
                                                  const blsgpu_g2_prepared* prepared, const void* d_offsets, size_t nseg, size_t total_terms, size_t max_seg_terms,
                                                  int final_exp, void* d_out);

/* ---- device groups: the path sharded over the GPUs of one node from ONE process ------------------------------------- */
/* "Large MSMs and pairing batches shard embarrassingly across the 8 GPUs" (SURVEY.md 8e): a group is one context per listed
 * device; every `*_sharded` call deals its independent terms to the members in contiguous slices (sizes differ by at most
 * one), runs each member on its own host thread, and folds the members' partial results -- ONE group element each: 144 B
 * G1, 288 B G2, 576 B Fp12 -- on member 0 with the reference's own operators: `Sum for G1Projective` (src/g1.rs:161-171,
 * src/g2.rs:162-172) and `MillerLoopResult + MillerLoopResult` (src/pairings.rs:179-186), then ONE final exponentiation
 * (:48-176).  Inside one process those few hundred bytes travel through host memory; no collective library is involved
 * (one process per GPU over RCCL is bls12_381_amd/distributed.py + bench.py --gpus N).  Results are the same group / field
 * elements as the single-context entry points'.  `devices` may name a device more than once (logical members on one GPU).
 * A group is driven by one host thread at a time; it owns one persistent worker thread per member beyond the first (created by
 * blsgpu_group_create, joined by blsgpu_group_destroy: no thread is started per call).  blsgpu_group_ctx gives access to a member's
 * context for the per-context settings (blsgpu_set_assume_subgroup, blsgpu_set_msm_window, ...).  The host-pointer `*_sharded`
 * calls are synchronous; the `*_sharded_device` MSMs below are asynchronous and pipelined. */
typedef struct blsgpu_group blsgpu_group;
typedef struct blsgpu_group_bases blsgpu_group_bases;   /* resident bases, member k holds points [lo_k, hi_k) on its device */
int blsgpu_group_create(const int* devices, int ndev, blsgpu_group** out);
void blsgpu_group_destroy(blsgpu_group* group);
int blsgpu_group_size(const blsgpu_group* group);
blsgpu_ctx* blsgpu_group_ctx(blsgpu_group* group, int member);
/* Shard n affine points (wire format as blsgpu_g1_bases_upload; `group_id` = 1 | 2) / the multiples [k_i]G over the members. */
int blsgpu_group_bases_upload(blsgpu_group* group, int group_id, const uint64_t* xy, const uint8_t* infinity, size_t n, blsgpu_group_bases** out);
int blsgpu_group_bases_from_scalars(blsgpu_group* group, int group_id, const uint8_t* scalars, size_t n, blsgpu_group_bases** out);
size_t blsgpu_group_bases_len(const blsgpu_group_bases* b);
void blsgpu_group_bases_free(blsgpu_group_bases* b);
/* out = sum_{i<n} scalars[i] * bases[i], n <= blsgpu_group_bases_len: every member runs a complete MSM over its slice,
 * member 0 adds the partial sums (blsgpu_g1_msm / blsgpu_g1_sum per member; same contracts, incl. canonical scalars). */
int blsgpu_g1_msm_sharded(blsgpu_group* group, const blsgpu_group_bases* bases, const uint8_t* scalars, size_t n, uint64_t out_xyz[18]);
int blsgpu_g2_msm_sharded(blsgpu_group* group, const blsgpu_group_bases* bases, const uint8_t* scalars, size_t n, uint64_t out_xyz[36]);
/* Device-pointer, asynchronous form for pipelined use (the headline path on every GPU of the node): member k multiplies its WHOLE
 * resident slice by the scalars at d_scalars[k] and writes its partial sum (18 / 36 u64) to d_partials[k], both in ITS device's memory;
 * the call only enqueues on every member (one persistent host thread per member does the enqueueing).  With
 * blsgpu_group_set_pipelining(group, 1) up to four calls per member overlap as described at blsgpu_set_pipelining.
 * blsgpu_g{1,2}_partials_fold(group, d_partials, lag, out) then waits -- per member, without holding up the `lag` most recent calls --
 * for the call whose partial sums `d_partials` name, and adds them on member 0 (`Sum`, src/g1.rs:161-171): the pattern is "enqueue MSM
 * i, fold MSM i - 2".  blsgpu_group_synchronize waits for everything and reports the sticky verdict of asynchronous calls. */
int blsgpu_g1_msm_sharded_device(blsgpu_group* group, const blsgpu_group_bases* bases, const void* const* d_scalars, void* const* d_partials);
int blsgpu_g2_msm_sharded_device(blsgpu_group* group, const blsgpu_group_bases* bases, const void* const* d_scalars, void* const* d_partials);
int blsgpu_g1_partials_fold(blsgpu_group* group, const void* const* d_partials, int lag, uint64_t out_xyz[18]);
int blsgpu_g2_partials_fold(blsgpu_group* group, const void* const* d_partials, int lag, uint64_t out_xyz[36]);
/* The fold without a host round trip: every member queues a copy of its partial sum to member 0's device behind the MSM it belongs to,
 * member 0's stream adds the w points into d_out (device memory of member 0, 18 / 36 u64); nothing is synchronised -- the fold of MSM
 * i - 3 runs under the accumulation of the later MSMs, on the members' own fold streams (an MSM's front waits for what is queued on its
 * context's stream, so the fold is kept off it).  At most four folds may be outstanding; an MSM whose output buffer one of the last eight
 * folds still has to read waits for that fold's copy, so a pipelined caller rotates at least eight partial-sum buffers per member (with
 * four, every MSM waits for the fold three calls back).  d_out is valid after blsgpu_group_synchronize. */
int blsgpu_g1_partials_fold_device(blsgpu_group* group, const void* const* d_partials, int lag, void* d_out_xyz);
int blsgpu_g2_partials_fold_device(blsgpu_group* group, const void* const* d_partials, int lag, void* d_out_xyz);
/* The pairing entry points in the same form: member k works on ITS arrays (device pointers in its memory, counts[k] items; the infinity
 * arrays may be NULL) and only enqueues.  mode 0: d_out[k] = counts[k] pairings (`Gt`), mode 1: raw Miller values -- the outputs stay sharded,
 * nothing to fold; mode 2: d_out[k] = the member-local `multi_miller_loop` of its terms (ONE Fp12 value), and
 * blsgpu_fp12_partials_fold_device(group, d_partials, final_exp, d_out) multiplies the members' values on member 0
 * (`MillerLoopResult + MillerLoopResult`, src/pairings.rs:179-186) and applies the ONE final exponentiation if asked; d_out (72 u64 in
 * member 0's memory) is valid after blsgpu_group_synchronize. */
int blsgpu_pairings_sharded_device(blsgpu_group* group, int mode, const void* const* d_g1_xy, const void* const* d_g1_inf, const void* const* d_g2_xy,
                                   const void* const* d_g2_inf, const size_t* counts, void* const* d_out);
int blsgpu_fp12_partials_fold_device(blsgpu_group* group, const void* const* d_partials, int final_exp, void* d_out_f);
int blsgpu_group_set_pipelining(blsgpu_group* group, int enabled);
int blsgpu_group_synchronize(blsgpu_group* group);
/* n independent pairings / raw Miller values: index slices, each member writes its slice of `out` (n x 72 u64); nothing to fold. */
int blsgpu_pairing_batch_sharded(blsgpu_group* group, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t n, uint64_t* out_gt);
int blsgpu_miller_loop_batch_sharded(blsgpu_group* group, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t n, uint64_t* out_f);
/* `multi_miller_loop` over n terms: member-local products of index slices, folded by member 0; final_exp != 0 applies the ONE
 * final exponentiation (out = a `Gt`), final_exp = 0 returns the `MillerLoopResult`. */
int blsgpu_multi_miller_loop_sharded(blsgpu_group* group, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t n, int final_exp,
                                     uint64_t out[72]);
/* Prepared G2 arguments for a group: the same m points prepared on EVERY member (blsgpu_g2_prepare per member), and the prepared Miller
 * loops sharded like the unprepared ones (terms / segments in contiguous slices, partial products folded on member 0, ONE final
 * exponentiation where asked). */
typedef struct blsgpu_group_g2_prepared blsgpu_group_g2_prepared;
int blsgpu_group_g2_prepare(blsgpu_group* group, const uint64_t* g2_xy, const uint8_t* g2_inf, size_t m, blsgpu_group_g2_prepared** out);
size_t blsgpu_group_g2_prepared_len(const blsgpu_group_g2_prepared* p);
void blsgpu_group_g2_prepared_free(blsgpu_group_g2_prepared* p);
int blsgpu_multi_miller_loop_prepared_sharded(blsgpu_group* group, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf, const uint32_t* q_index,
                                              const blsgpu_group_g2_prepared* prepared, size_t n, int final_exp, uint64_t out[72]);
int blsgpu_multi_miller_loop_prepared_many_sharded(blsgpu_group* group, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf,
                                                   const uint32_t* q_index, const blsgpu_group_g2_prepared* prepared, const uint64_t* offsets, size_t nseg, int final_exp,
                                                   uint64_t* out);
/* blsgpu_multi_miller_loop_many with the SEGMENTS dealt to the members in contiguous slices. */
int blsgpu_multi_miller_loop_many_sharded(blsgpu_group* group, const uint64_t* g1_xy, const uint8_t* g1_inf, const uint64_t* g2_xy, const uint8_t* g2_inf,
                                          const uint64_t* offsets, size_t nseg, int final_exp, uint64_t* out);

/* ---- field self-test hooks (parity tests of the arithmetic core against the oracle) ---------------------- */
/* out[i] = a[i] op b[i] over n Fp elements in wire format; op: 0 mul, 1 add, 2 sub, 3 square(a), 4 invert(a), 5 neg(a). */
int blsgpu_fp_op(blsgpu_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Same over Fp2 (12 u64 per element); op: 0 mul, 1 add, 2 sub, 3 square, 4 invert, 5 neg, 6 mul_by_nonresidue. */
int blsgpu_fp2_op(blsgpu_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Same over Fp6 (36 u64 per element: c0 | c1 | c2, src/fp6.rs:12-16); op: 0 mul (fp6.rs:200-274), 3 square (:277-291), 4 invert
 * (:294-312), 5 mul_by_nonresidue (:139-150), 7 frobenius_map (:154-188), 11 mul_by_1 with c1 = b.c1 (:113-119), 12 mul_by_01 with
 * (c0, c1) = (b.c0, b.c1) (:121-136). */
int blsgpu_fp6_op(blsgpu_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Same over Fp12 (72 u64 per element); op: 0 mul, 3 square, 4 invert, 7 frobenius_map, 8 conjugate, 9 cyclotomic_square,
 * 10 the final exponentiation's power by |x| followed by conjugation (`cycolotomic_exp`, src/pairings.rs:114-132; quad layout only). */
int blsgpu_fp12_op(blsgpu_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Point ops over n pairs in wire format; op: 0 add (proj+proj), 1 double (a), 2 add_mixed (proj + affine b). group 1|2. */
int blsgpu_point_op(blsgpu_ctx* ctx, int group, int op, const uint64_t* a, const uint64_t* b, const uint8_t* b_inf, size_t n, uint64_t* out);
/* Throughput probe: `iters` dependent Fp multiplications in every lane of a full-chip launch; returns
 * multiplications per second (the arithmetic roofline evidence quoted by bench.py). */
int blsgpu_fp_mul_throughput(blsgpu_ctx* ctx, int iters, double* muls_per_second);
/* Same for a stream of independent v_mad_u64_u32 (the peak MAC32 rate used as roofline denominator). */
int blsgpu_mad_throughput(blsgpu_ctx* ctx, int iters, double* mads_per_second);
/* Duration (ms) of the most recent launch of the named phase inside the last MSM (HIP events on the
 * context's stream): 0 digits+hist, 1 scan, 2 scatter, 3 order, 4 accumulate, 5 reduce, 6 combine, 7 total. */
int blsgpu_last_msm_phase_ms(blsgpu_ctx* ctx, int phase, float* ms);
int blsgpu_set_profiling(blsgpu_ctx* ctx, int enabled);
/* Live duration of the bucket-accumulation kernel (the dominant kernel of an MSM) measured with HIP events on the stream
 * it is launched on, also while calls are pipelined: returns the average over the launches since the previous call and
 * their number, then resets the statistics and switches the measurement off (enable = 0) or on: enable = N > 0 times every
 * N-th launch (two event records per timed launch cost ~0.05-0.1 ms of queue time in a pipelined run, so benchmarks sample). */
int blsgpu_msm_accumulate_stats(blsgpu_ctx* ctx, int enable, double* avg_ms, unsigned* launches);
/* Diagnostics: the duration of EVERY kernel the context's entry points launch, measured with HIP events on the stream each kernel is
 * launched on (a call's kernels run on up to four library streams, which a caller-side event never sees).  blsgpu_kernel_timing(ctx, 1)
 * starts recording (and clears earlier records), (ctx, 0) stops; blsgpu_kernel_timing_report waits for the recorded launches and writes one
 * line per kernel name in first-launch order -- "name<TAB>launches<TAB>total_ms<TAB>min_ms<TAB>max_ms\n" -- into buf (at most cap - 1
 * characters + NUL; the full length goes to *needed, which may be NULL), then clears the records.  Two event records per launch: meant
 * for one call at a time (bench.py's `kernel_ms` rows), not for pipelined production runs. */
int blsgpu_kernel_timing(blsgpu_ctx* ctx, int enable);
int blsgpu_kernel_timing_report(blsgpu_ctx* ctx, char* buf, size_t cap, size_t* needed);

/* ---- scalar field Fr (the producer side of the MSM's scalars) ----------------------- */
/* A scalar is the reference's `Scalar([u64; 4])`: four little-endian u64 Montgomery limbs (R = 2^256), canonical
 * (scalar.rs:23-27).  Element-wise vector operation over n scalars; op: 0 mul (scalar.rs:452-503), 1 add (:435-449),
 * 2 sub (:420-432), 3 square (:334-369), 4 invert (:573-628; a zero input yields 0 and nonzero_flags[i] = 0, the
 * reference's CtOption::none), 5 neg (:552-568), 6 double (:246-250).  `b` is ignored for unary ops,
 * `nonzero_flags` (n bytes) may be NULL. */
int blsgpu_fr_op(blsgpu_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, uint8_t* nonzero_flags);
int blsgpu_fr_op_device(blsgpu_ctx* ctx, int op, const void* d_a, const void* d_b, size_t n, void* d_out, void* d_nonzero_flags);
/* `Scalar` <-> bytes over vectors of n scalars:
 *   to_bytes         Montgomery limbs -> 32 canonical little-endian bytes, `Scalar::to_bytes` (src/scalar.rs:284-296: one
 *                    `montgomery_reduce`, :506-550); ok[i] = 0 where the limbs are not below r (no `Scalar` holds them);
 *   from_bytes       32 bytes -> Montgomery limbs (multiplication by R2), ok[i] = 0 where `Scalar::from_bytes` returns
 *                    CtOption::none: the integer is not below r (src/scalar.rs:256-280);
 *   from_bytes_wide  64 bytes -> the 512-bit little-endian integer mod r as d0 R2 + d1 R3 (src/scalar.rs:300-331); always defined.
 * `ok` (n bytes) may be NULL.  The device-pointer forms are asynchronous on the context's stream, e.g.
 * blsgpu_fr_ntt_device -> blsgpu_g1_msm_mont_device needs no conversion at all, and blsgpu_fr_to_bytes_device feeds the byte-form
 * entry points. */
int blsgpu_fr_to_bytes(blsgpu_ctx* ctx, const uint64_t* scalars, size_t n, uint8_t* bytes, uint8_t* ok);
int blsgpu_fr_from_bytes(blsgpu_ctx* ctx, const uint8_t* bytes, size_t n, uint64_t* scalars, uint8_t* ok);
int blsgpu_fr_from_bytes_wide(blsgpu_ctx* ctx, const uint8_t* bytes, size_t n, uint64_t* scalars);
int blsgpu_fr_to_bytes_device(blsgpu_ctx* ctx, const void* d_scalars, size_t n, void* d_bytes, void* d_ok);
int blsgpu_fr_from_bytes_device(blsgpu_ctx* ctx, const void* d_bytes, size_t n, void* d_scalars, void* d_ok);
int blsgpu_fr_from_bytes_wide_device(blsgpu_ctx* ctx, const void* d_bytes, size_t n, void* d_scalars);
/* Radix-2 number-theoretic transform of 2^log_n scalars, in place, natural order in and out:
 *   forward  y_k = sum_j x_j w^(jk),   inverse  x_j = n^-1 sum_k y_k w^(-jk),   w = ROOT_OF_UNITY^(2^(32 - log_n))
 * with the reference's ROOT_OF_UNITY / S = 32 (scalar.rs:191-205, exported through ff::PrimeField :703-712).
 * The reference crate has no transform of its own; this is the operation its callers build on those constants.
 * log_n in [0, 28]. */
int blsgpu_fr_ntt(blsgpu_ctx* ctx, uint64_t* data, int log_n, int inverse);
int blsgpu_fr_ntt_device(blsgpu_ctx* ctx, void* d_data, int log_n, int inverse);

/* ---- hash-to-curve (the step in front of multi_miller_loop in bulk signature checks) ---- */
/* `<G as HashToCurve<ExpandMsgXmd<Sha256>>>::hash_to_curve(msg, dst)` / `encode_to_curve` (encode_only != 0) for n
 * messages (src/hash_to_curve/mod.rs:86-108; expand_msg.rs:230-328; map_g1.rs:513-638; map_g2.rs:374-504;
 * g1.rs:800-802, g2.rs:938-947).  `msgs` holds the messages back to back, message i = bytes offsets[i] ..
 * offsets[i+1] (n + 1 offsets).  `dst` of any length (longer than 255 bytes: reduced as expand_msg.rs:74-95 does).
 * out_xyz: n projective points (X : Y : Z), 18 (G1) / 36 (G2) u64 each -- the reference's own coordinates. */
int blsgpu_g1_hash_to_curve_batch(blsgpu_ctx* ctx, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                  int encode_only, uint64_t* out_xyz);
int blsgpu_g2_hash_to_curve_batch(blsgpu_ctx* ctx, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                  int encode_only, uint64_t* out_xyz);
/* Same with messages, offsets and DST (<= 255 bytes) already in device memory; group = 1 | 2. */
int blsgpu_hash_to_curve_device(blsgpu_ctx* ctx, int group, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len,
                                int encode_only, void* d_out_xyz);
/* The reference's hash-to-curve is generic in the expander: `hash_to_curve::<X>` with X = ExpandMsgXmd<H> for a fixed-output digest H
 * (src/hash_to_curve/expand_msg.rs:230-328) or ExpandMsgXof<H> for an extendable-output function (:167-228); its tests run Sha256, Sha512,
 * Shake128 and Shake256 (tests/expand_msg.rs).  The entry points above are X = ExpandMsgXmd<Sha256>, fused into the hashing kernels (the
 * BLS-signature suites); the `_expander` forms take the choice as an argument, expand into uniform bytes and map from those -- for
 * BLSGPU_EXPAND_XMD_SHA256 with results limb-identical to the fused kernels.  A DST longer than 255 bytes is reduced as
 * expand_msg.rs:47-95 does (XMD: the digest of the salted tag; XOF: its first 32 output bytes) by the host-pointer forms; the
 * device-pointer forms take a DST of at most 255 bytes.  group = 1 | 2. */
#define BLSGPU_EXPAND_XMD_SHA256 0
#define BLSGPU_EXPAND_XMD_SHA512 1
#define BLSGPU_EXPAND_XOF_SHAKE128 2
#define BLSGPU_EXPAND_XOF_SHAKE256 3
int blsgpu_hash_to_curve_expander_batch(blsgpu_ctx* ctx, int group, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst,
                                        size_t dst_len, int encode_only, uint64_t* out_xyz);
int blsgpu_hash_to_curve_expander_device(blsgpu_ctx* ctx, int group, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst,
                                         size_t dst_len, int encode_only, void* d_out_xyz);
/* The part of `hash_to_curve` / `encode_to_curve` BEHIND the expander (hash_to_curve/mod.rs:86-108 after `hash_to_field`'s
 * `expand_message` call): per message count x M x 64 uniform bytes (count = 1 for encode_only, else 2; M = 1 for G1, 2 for G2) are reduced
 * as `from_okm` does (map_g1.rs:512-531, map_g2.rs:362-380), mapped (`map_to_curve`: simplified SWU + isogeny), added and cleared of the
 * cofactor.  For callers with an expander of their own; with the bytes of blsgpu_expand_message_* it gives the limbs of
 * blsgpu_hash_to_curve_expander_*.  uniform = n x count x M x 64 bytes, out_xyz = n projective points in wire limbs -- the reference's own
 * coordinates, except that an identity result (the two field elements of a message negatives of each other) is (0 : Y : 0) with a Y of the
 * formulas' making where the reference's `double` substitutes the literal (0 : 1 : 0) (g1.rs:666, g2.rs:737). */
int blsgpu_hash_to_curve_from_uniform_batch(blsgpu_ctx* ctx, int group, const uint8_t* uniform, size_t n, int encode_only, uint64_t* out_xyz);
int blsgpu_hash_to_curve_from_uniform_device(blsgpu_ctx* ctx, int group, const void* d_uniform, size_t n, int encode_only, void* d_out_xyz);
/* `ExpandMessage::init_expand(msg, dst, len_in_bytes)` followed by reading all `len_in_bytes` bytes, for n messages: out = n x len_in_bytes
 * uniform bytes.  len_in_bytes <= 65535, and at most 255 digest blocks for the XMD expanders (the reference panics beyond, :181-183, :263-268). */
int blsgpu_expand_message_batch(blsgpu_ctx* ctx, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                size_t len_in_bytes, uint8_t* out);
int blsgpu_expand_message_device(blsgpu_ctx* ctx, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len,
                                 size_t len_in_bytes, void* d_out);
/* `hash_to_field::<X, Scalar>` (src/hash_to_curve/mod.rs:32-49 with `HashToField for Scalar`, map_scalar.rs:10-25: 48 uniform bytes per
 * element, zero-extended to 64 and read as `Scalar::from_bytes_wide`): `count` scalars per message, as the reference's `Scalar` (four u64
 * Montgomery limbs): out = n x count x 4 u64 -- ready for blsgpu_g{1,2}_msm_mont* / blsgpu_fr_*. */
int blsgpu_hash_to_scalar_batch(blsgpu_ctx* ctx, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                size_t count, uint64_t* out);
int blsgpu_hash_to_scalar_device(blsgpu_ctx* ctx, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len,
                                 size_t count, void* d_out);


/* ---- the widened rows with inputs and outputs in device memory: a chain that stays on the GPU ---------------- */
/* Device-pointer twins of blsgpu_g{1,2}_batch_normalize, *_from_bytes_batch, *_to_bytes_batch and blsgpu_gt_mul_scalar_batch:
 * same kernels, same formats, asynchronous on the context's stream, nothing crosses PCIe.  With them the stages of bulk
 * verification -- decode + subgroup check (src/g1.rs:273-416, src/g2.rs:330-489), hash-to-curve (map_g2.rs:374-504), normalise
 * (g2.rs:951-984), multi_miller_loop (pairings.rs:554-603), compare with the identity -- hand their results to one another in
 * HBM: blsgpu_hash_to_curve_device writes PROJECTIVE points, *_batch_normalize_device turns them into the AFFINE + infinity
 * form the Miller entry points take.  `d_infinity` of *_batch_normalize_device and `d_ok` / `d_infinity` of *_from_bytes_batch_device
 * are outputs (n bytes each, required). */
int blsgpu_g1_batch_normalize_device(blsgpu_ctx* ctx, const void* d_xyz, size_t n, void* d_xy, void* d_infinity);
int blsgpu_g2_batch_normalize_device(blsgpu_ctx* ctx, const void* d_xyz, size_t n, void* d_xy, void* d_infinity);
int blsgpu_g1_from_bytes_batch_device(blsgpu_ctx* ctx, const void* d_bytes, size_t n, int compressed, int checked, void* d_xy, void* d_infinity, void* d_ok);
int blsgpu_g2_from_bytes_batch_device(blsgpu_ctx* ctx, const void* d_bytes, size_t n, int compressed, int checked, void* d_xy, void* d_infinity, void* d_ok);
int blsgpu_g1_to_bytes_batch_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, size_t n, int compressed, void* d_out);
int blsgpu_g2_to_bytes_batch_device(blsgpu_ctx* ctx, const void* d_xy, const void* d_infinity, size_t n, int compressed, void* d_out);
int blsgpu_gt_mul_scalar_batch_device(blsgpu_ctx* ctx, const void* d_gt, const void* d_scalars, size_t n, void* d_out);
/* d_flags[i] = 1 if gt[i] == Gt::identity() (= Fp12::one(), src/pairings.rs:211-218), else 0: the verdict of an equation
 * prod e(P_j, Q_j) == 1 without bringing 576 B per equation to the host.  (d_gt must be 16-byte aligned, as device allocations are.) */
int blsgpu_gt_is_identity_device(blsgpu_ctx* ctx, const void* d_gt, size_t n, void* d_flags);
/* Bulk BLS signature verification, bytes in -> verdict bytes out, every stage on the device (one upload, one download in the
 * host-pointer form): checked decoding of the compressed public keys and signatures, hash_to_curve of the messages (message i =
 * bytes offsets[i] .. offsets[i+1] of `msgs`; `dst` <= 255 bytes in the device form), normalisation, ONE multi_miller_loop +
 * final exponentiation per signature, comparison with Gt::identity().
 *   mode 0 (public keys in G1, 48 B; signatures in G2, 96 B):  e(pk, H(m)) * e(-G1, sig) == 1   <=>  e(pk, H(m)) == e(G1, sig)
 *   mode 1 (signatures in G1, 48 B; public keys in G2, 96 B):  e(sig, -G2) * e(H(m), pk) == 1   <=>  e(sig, G2) == e(H(m), pk),
 *           with -G2 a `G2Prepared` resident in the context (built at first use).
 * verdict[i]: 1 = the equation holds, 0 = it does not, 2 = the public key is not a valid encoding of a subgroup point
 * (`from_compressed` -> None), 3 = the signature is not.  Identity points decode successfully and take part as the reference's
 * `pairing` treats them (an identity on either side contributes Gt::identity()); rejecting an identity public key is the caller's
 * policy (`KeyValidate`).  This is the reference's own operations composed, not a new scheme: src/g1.rs:336-390, src/g2.rs:390-489,
 * src/hash_to_curve/map_g2.rs:374-504 (map_g1.rs:513-638), src/g2.rs:951-984, src/pairings.rs:554-603, :48-176. */
int blsgpu_bls_verify_batch(blsgpu_ctx* ctx, int mode, const uint8_t* pk_bytes, const uint8_t* sig_bytes, const uint8_t* msgs, const uint64_t* offsets, size_t n,
                            const uint8_t* dst, size_t dst_len, uint8_t* verdict);
int blsgpu_bls_verify_batch_device(blsgpu_ctx* ctx, int mode, const void* d_pk_bytes, const void* d_sig_bytes, const void* d_msgs, const void* d_offsets, size_t n,
                                   const void* d_dst, size_t dst_len, void* d_verdict);

#ifdef __cplusplus
}
#endif
#endif /* BLS12_381_HIP_H */
