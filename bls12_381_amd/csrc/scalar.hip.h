// scalar.hip.h -- the scalar field element `Scalar([u64; 4])` of the reference as the kernels see it, and its conversions.
//
// Reference: /root/reference/src/scalar.rs -- Montgomery form with R = 2^256 (:23-27, :155-165), MODULUS :76-81, INV :156,
// R2 :167-172, R3 :174-180; add :435-449, sub :420-432, mul :452-503 + montgomery_reduce :506-550;
// to_bytes :284-296 (one montgomery_reduce), from_bytes :256-280 (range check, then * R2), from_bytes_wide / from_u512 :300-331
// (d0 * R2 + d1 * R3).  SURVEY.md 8 row a8: the MSM consumes `Scalar::to_bytes()`; with the loader below every scalar-consuming
// kernel takes EITHER the 32 canonical little-endian bytes OR the four Montgomery limbs exactly as `&[Scalar]` memory holds them,
// so a caller never converts on the host (2^20 `to_bytes` calls cost more CPU time than the MSM costs GPU time).
// No kernels here: this header is shared by msm.hip.h, mulbatch.hip.h, pairing.hip.h and fr.hip.h.
#pragma once
#include "fe.hip.h"

namespace bls {

struct Fr { u32 l[8]; };

struct FrWords { u32 w[8]; };
constexpr FrWords FR_MOD_C = {BLS_FR_MOD_W};   // scalar.rs:76-81
#define FR_MOD (FR_MOD_C.w)
constexpr u32 FR_INV32 = BLS_FR_INV32;         // -r^-1 mod 2^32 (low word of scalar.rs:156 INV)

DEV Fr fr_zero() { Fr r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
DEV Fr fr_load(const u32* p) {
  const uint4* v = reinterpret_cast<const uint4*>(p);
  uint4 a = v[0], b = v[1];
  Fr r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
DEV void fr_store(u32* p, const Fr& a) {
  uint4* v = reinterpret_cast<uint4*>(p);
  v[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
  v[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
}
DEV bool fr_is_zero(const Fr& a) { u32 t = 0; for (int i = 0; i < 8; i++) t |= a.l[i]; return t == 0; }
// a - r if a >= r (a < 2r, possibly with a carry bit out of the top word)
DEV Fr fr_cond_sub(const Fr& a, u32 carry) {
  Fr s; int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { int64_t d = (int64_t)a.l[i] - FR_MOD[i] + br; s.l[i] = (u32)d; br = d >> 32; }
  const bool take = (int64_t)carry + br >= 0;        // no borrow overall
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = take ? s.l[i] : a.l[i];
  return r;
}
DEV Fr fr_add(const Fr& a, const Fr& b) {
  Fr t; u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { u64 x = (u64)a.l[i] + b.l[i] + c; t.l[i] = (u32)x; c = x >> 32; }
  return fr_cond_sub(t, (u32)c);
}
DEV Fr fr_sub(const Fr& a, const Fr& b) {
  Fr t; int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { int64_t d = (int64_t)a.l[i] - b.l[i] + br; t.l[i] = (u32)d; br = d >> 32; }
  const u32 m = (u32)br;                              // all ones if a < b: add r back
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { u64 x = (u64)t.l[i] + (FR_MOD[i] & m) + c; t.l[i] = (u32)x; c = x >> 32; }
  return t;
}
DEV Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
// CIOS Montgomery product, canonical result
DEV Fr fr_mul(const Fr& a, const Fr& b) {
  u32 t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { u64 x = (u64)a.l[j] * b.l[i] + t[j] + c; t[j] = (u32)x; c = x >> 32; }
    u64 x = (u64)t[8] + c; t[8] = (u32)x; t[9] = (u32)(x >> 32);
    const u32 m = t[0] * FR_INV32;
    c = ((u64)m * FR_MOD[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { u64 y = (u64)m * FR_MOD[j] + t[j] + c; t[j - 1] = (u32)y; c = y >> 32; }
    x = (u64)t[8] + c; t[7] = (u32)x; t[8] = t[9] + (u32)(x >> 32);
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  return fr_cond_sub(r, t[8]);
}
DEV Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }
DEV Fr fr_one() {
  // R mod r (scalar.rs:159-164)
  constexpr FrWords k = {BLS_FR_ONE_W};
  Fr r;
  for (int i = 0; i < 8; i++) r.l[i] = k.w[i];
  return r;
}

// ---- Montgomery limbs <-> canonical integer ---------------------------------------------------------------------------
// scalar.rs:284-296 `to_bytes`: montgomery_reduce(a0..a3, 0, 0, 0, 0) = a / 2^256 mod r, canonical (eight words = the 32 LE bytes)
DEV Fr fr_from_mont(const Fr& a) {
  u32 t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = a.l[i];
  u32 top = 0;                                       // the carry word above t[7]
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u32 m = t[0] * FR_INV32;
    u64 c = ((u64)m * FR_MOD[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { u64 y = (u64)m * FR_MOD[j] + t[j] + c; t[j - 1] = (u32)y; c = y >> 32; }
    u64 x = (u64)top + c; t[7] = (u32)x; top = (u32)(x >> 32);
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  return fr_cond_sub(r, top);
}
// the words are below r: what `Scalar` limbs always are, and what `from_bytes` demands of its input (scalar.rs:265-274)
DEV bool fr_words_below_r(const u32* s) {
  bool lt = false, eq = true;
#pragma unroll
  for (int i = 7; i >= 0; i--) { lt = lt || (eq && s[i] < FR_MOD[i]); eq = eq && s[i] == FR_MOD[i]; }
  return lt;
}
// scalar.rs:256-280 `from_bytes`: ok = the integer is canonical; the value is integer * R2 / R either way (as the reference computes it)
DEV Fr fr_to_mont(const Fr& canon) {
  constexpr FrWords k = {BLS_FR_R2_W};
  Fr r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.l[i] = k.w[i];
  return fr_mul(r2, canon);                          // the full-width operand of the CIOS loop must be < r: R2 is, the input need not be
}
// scalar.rs:300-331 `from_bytes_wide` = from_u512: d0 * R2 + d1 * R3 for the two 256-bit halves (any values below 2^256)
DEV Fr fr_from_wide(const Fr& d0, const Fr& d1) {
  constexpr FrWords k2 = {BLS_FR_R2_W}, k3 = {BLS_FR_R3_W};
  Fr r2, r3;
#pragma unroll
  for (int i = 0; i < 8; i++) { r2.l[i] = k2.w[i]; r3.l[i] = k3.w[i]; }
  return fr_add(fr_mul(r2, d0), fr_mul(r3, d1));
}

// ---- the one way a kernel reads a scalar ------------------------------------------------------------------------------------
// k[0..7] = the canonical integer of scalar i, little-endian words.  form SCALAR_BYTES: `scalars` holds 32 canonical LE bytes per
// scalar (`Scalar::to_bytes()`); form SCALAR_MONT: it holds the four u64 Montgomery limbs of a `Scalar` (the memory of a `&[Scalar]`),
// reduced here.  Returns false for an input no `Scalar` can hold (bytes or limbs >= r): callers report it through the sticky flag.
constexpr int SCALAR_BYTES = 0, SCALAR_MONT = 1;
DEV bool scalar_load(const u32* __restrict__ scalars, size_t i, int form, u32* k) {
  Fr a = fr_load(scalars + i * 8);
  const bool ok = fr_words_below_r(a.l);
  if (form == SCALAR_MONT) a = fr_from_mont(a);
#pragma unroll
  for (int j = 0; j < 8; j++) k[j] = a.l[j];
  return ok;
}

}  // namespace bls
