// fr.cuh -- the scalar field Fr of BLS12-381 and a radix-2 number-theoretic transform over it.
//
// Reference: /root/reference/src/scalar.rs -- `Scalar([u64; 4])` in Montgomery form with R = 2^256 (:23-27,
// :155-165), add :435-449, sub :420-432, neg :552-568, mul :452-503 + montgomery_reduce :506-550, square
// :334-369, pow :371-404, invert :573-628 (a^(r-2)), ROOT_OF_UNITY (a 2^32-th root of unity) :193-205,
// S = 32 :191.  SURVEY.md 8(f) rank 3: vectors of scalars are what a prover feeds the MSM, and their
// transforms are the step before it.
//
// Elements stay in the reference's own representation end to end: eight little-endian u32 words = the four
// u64 Montgomery limbs, always canonical (< r), so there is no conversion on the way in or out.  Fr work is
// HBM-bound (one multiplication per 64 bytes moved in a butterfly), so the arithmetic is a plain 8 x 32-bit
// CIOS Montgomery product; the effort goes into touching every element as few times as possible:
//   * stages whose butterflies span more than one tile run two at a time (radix-4 passes over global memory),
//   * the last FR_TILE_LOG stages run on a tile resident in LDS, and the bit-reversal that a decimation-in-
//     frequency schedule leaves behind is folded into that kernel's store.
// The transform (natural order in and out):  y_k = sum_j x_j w^(jk),  w = ROOT_OF_UNITY^(2^(32 - log n));
// the inverse uses w^-1 and scales by n^-1.  The reference crate has no transform; oracle/bls12_381_ref.py
// fr_ntt states the definition the kernels are tested against.
#pragma once
#include "fe.cuh"

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
// a^e, e = 8 little-endian u32 words (scalar.rs:371-404), square-and-multiply from the top bit
DEVNI Fr fr_pow(const Fr& a, const u32* e) {
  Fr r = fr_one();
  for (int i = 255; i >= 0; i--) {
    r = fr_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = fr_mul(r, a);
  }
  return r;
}
// a^(r-2); 0 for a = 0 (the reference returns CtOption::none, scalar.rs:573-628 -- callers get a flag)
DEVNI Fr fr_inv(const Fr& a) {
  constexpr FrWords rm2 = {BLS_FR_RM2_W};
  u32 e[8];
  for (int i = 0; i < 8; i++) e[i] = rm2.w[i];
  return fr_pow(a, e);
}
// small powers of two of the exponent: w^(2^k)
DEV Fr fr_pow2k(Fr a, int k) { for (int i = 0; i < k; i++) a = fr_sqr(a); return a; }
// a^n for a 64-bit n
DEVNI Fr fr_pow_u64(const Fr& a, u64 n) {
  Fr r = fr_one();
  for (int i = 63; i >= 0; i--) { r = fr_sqr(r); if ((n >> i) & 1ull) r = fr_mul(r, a); }
  return r;
}

// ROOT_OF_UNITY (scalar.rs:200-205), Montgomery limbs as u32 words
DEV Fr fr_root_of_unity() {
  constexpr FrWords k = {BLS_FR_ROOT_W};
  Fr r;
  for (int i = 0; i < 8; i++) r.l[i] = k.w[i];
  return r;
}

// ---- element-wise vector operations --------------------------------------------------------------------
// op 0 mul, 1 add, 2 sub, 3 square, 4 invert (flag[i] = 0 for a zero input, as CtOption::none), 5 neg, 6 double
__global__ void __launch_bounds__(256) k_fr_op(int op, const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out,
                                               uint8_t* __restrict__ flag, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x = fr_load(a + i * 8);
  Fr y = b ? fr_load(b + i * 8) : x;
  Fr r;
  switch (op) {
    case 0: r = fr_mul(x, y); break;
    case 1: r = fr_add(x, y); break;
    case 2: r = fr_sub(x, y); break;
    case 3: r = fr_sqr(x); break;
    case 4: r = fr_inv(x); if (flag) flag[i] = fr_is_zero(x) ? 0 : 1; break;
    case 5: r = fr_neg(x); break;
    default: r = fr_add(x, x); break;
  }
  fr_store(out + i * 8, r);
}

// ---- twiddle table: tw[j] = w^j, j < n/2 ------------------------------------------------------------------
// each lane starts from w^(64 t) (one exponentiation) and walks 64 consecutive powers
constexpr int FR_TW_RUN = 64;
__global__ void __launch_bounds__(256) k_fr_twiddles(u32* __restrict__ tw, int log_n, int inverse) {
  const size_t half = (size_t)1 << (log_n - 1);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j0 = t * FR_TW_RUN;
  if (j0 >= half) return;
  Fr w = fr_pow2k(fr_root_of_unity(), 32 - log_n);
  if (inverse) w = fr_inv(w);
  Fr cur = fr_pow_u64(w, (u64)j0);
  for (int k = 0; k < FR_TW_RUN && j0 + k < half; k++) { fr_store(tw + (j0 + k) * 8, cur); cur = fr_mul(cur, w); }
}

// ---- decimation-in-frequency stages over global memory ------------------------------------------------------
// One radix-2 DIF stage with half-span h on a length-n vector: for each block of 2h elements,
//   (a, b) = (x[i], x[i+h])  ->  x[i] = a + b,  x[i+h] = (a - b) w^(i * n/(2h))
// k_fr_stage2 performs TWO consecutive stages (h and h/2) on four elements per lane.
// (src and dst may be the same buffer: every lane reads its own elements before it writes them)
__global__ void __launch_bounds__(256) k_fr_stage1(const u32* src, u32* x, const u32* __restrict__ tw, int log_n, int log_h) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t half = (size_t)1 << (log_n - 1);
  if (t >= half) return;
  const size_t h = (size_t)1 << log_h;
  const size_t i = t & (h - 1), blk = t >> log_h;
  const size_t p = (blk << (log_h + 1)) + i;
  Fr a = fr_load(src + p * 8), b = fr_load(src + (p + h) * 8);
  Fr w = fr_load(tw + (i << (log_n - 1 - log_h)) * 8);
  fr_store(x + p * 8, fr_add(a, b));
  fr_store(x + (p + h) * 8, fr_mul(fr_sub(a, b), w));
}
__global__ void __launch_bounds__(256) k_fr_stage2(const u32* src, u32* x, const u32* __restrict__ tw, int log_n, int log_h) {
  // stages with half-spans h = 2^log_h and h/2; lane t owns elements p, p + h/2, p + h, p + 3h/2
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t quarter = (size_t)1 << (log_n - 2);
  if (t >= quarter) return;
  const size_t q = (size_t)1 << (log_h - 1);           // h / 2
  const size_t i = t & (q - 1), blk = t >> (log_h - 1);
  const size_t p = (blk << (log_h + 1)) + i;
  Fr a0 = fr_load(src + p * 8), a1 = fr_load(src + (p + q) * 8), a2 = fr_load(src + (p + 2 * q) * 8), a3 = fr_load(src + (p + 3 * q) * 8);
  const int sh = log_n - 1 - log_h;                      // twiddle stride of the first stage
  Fr w0 = fr_load(tw + (i << sh) * 8), w1 = fr_load(tw + ((i + q) << sh) * 8);
  Fr w2 = fr_load(tw + (i << (sh + 1)) * 8);            // second stage: index i (mod q), stride doubled
  // stage h: pairs (a0, a2), (a1, a3)
  Fr b0 = fr_add(a0, a2), b2 = fr_mul(fr_sub(a0, a2), w0);
  Fr b1 = fr_add(a1, a3), b3 = fr_mul(fr_sub(a1, a3), w1);
  // stage h/2: pairs (b0, b1), (b2, b3)
  fr_store(x + p * 8, fr_add(b0, b1));
  fr_store(x + (p + q) * 8, fr_mul(fr_sub(b0, b1), w2));
  fr_store(x + (p + 2 * q) * 8, fr_add(b2, b3));
  fr_store(x + (p + 3 * q) * 8, fr_mul(fr_sub(b2, b3), w2));
}

// ---- the last stages on a tile in LDS + bit-reversed store ---------------------------------------------------
constexpr int FR_TILE_LOG = 10;                  // 1024 elements = 32 KB of LDS per workgroup
// Runs the stages with half-spans 2^(tl-1) ... 1 on each aligned tile of 2^tl elements (tl = min(FR_TILE_LOG,
// log_n)), then writes element p of the (bit-reversed) result to its natural position bitrev(p), optionally
// scaled (the inverse transform's n^-1).  `x` is read, `y` written (they differ: the permutation is not in place).
__global__ void __launch_bounds__(256) k_fr_tile(const u32* __restrict__ x, u32* __restrict__ y, const u32* __restrict__ tw, int log_n,
                                                 int tl, const u32* __restrict__ scale) {
  extern __shared__ u32 lds[];                   // 2^tl elements, word-interleaved: word k of element e at lds[k * 2^tl + e]
  const int T = 1 << tl;
  const size_t base = (size_t)blockIdx.x << tl;
  for (int e = threadIdx.x; e < T; e += blockDim.x) {
    Fr v = fr_load(x + (base + e) * 8);
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * T + e] = v.l[k];
  }
  __syncthreads();
  for (int lh = tl - 1; lh >= 0; lh--) {
    const int h = 1 << lh;
    for (int t = threadIdx.x; t < T / 2; t += blockDim.x) {
      const int i = t & (h - 1), p = ((t >> lh) << (lh + 1)) + i;
      Fr a, b;
#pragma unroll
      for (int k = 0; k < 8; k++) { a.l[k] = lds[k * T + p]; b.l[k] = lds[k * T + p + h]; }
      Fr w = fr_load(tw + ((size_t)i << (log_n - 1 - lh)) * 8);
      Fr s = fr_add(a, b), d = fr_mul(fr_sub(a, b), w);
#pragma unroll
      for (int k = 0; k < 8; k++) { lds[k * T + p] = s.l[k]; lds[k * T + p + h] = d.l[k]; }
    }
    __syncthreads();
  }
  Fr sc; if (scale) sc = fr_load(scale);
  for (int e = threadIdx.x; e < T; e += blockDim.x) {
    Fr v;
#pragma unroll
    for (int k = 0; k < 8; k++) v.l[k] = lds[k * T + e];
    if (scale) v = fr_mul(v, sc);
    const size_t p = base + e;
    const size_t r = (size_t)(__brevll((unsigned long long)p) >> (64 - log_n));
    fr_store(y + r * 8, v);
  }
}
// n^-1 in Montgomery form (n = 2^log_n): (2^-1)^log_n
__global__ void k_fr_ninv(u32* __restrict__ out, int log_n) {
  if (threadIdx.x || blockIdx.x) return;
  constexpr FrWords k = {BLS_FR_TWO_INV_W};        // TWO_INV (scalar.rs:183-188)
  Fr h;
  for (int i = 0; i < 8; i++) h.l[i] = k.w[i];
  Fr r = fr_one();
  for (int i = 0; i < log_n; i++) r = fr_mul(r, h);
  fr_store(out, r);
}

}  // namespace bls
