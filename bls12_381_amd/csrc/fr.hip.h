// fr.hip.h -- the scalar field Fr of BLS12-381 and a radix-2 number-theoretic transform over it.
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
#include "scalar.hip.h"

namespace bls {

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
  if (n == 0) return r;
  for (int i = 63 - __clzll((unsigned long long)n); i >= 0; i--) { r = fr_sqr(r); if ((n >> i) & 1ull) r = fr_mul(r, a); }
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

// ---- `Scalar` <-> bytes for whole vectors (scalar.rs:256-331) --------------------------------------------------------------
// op 0 to_bytes: Montgomery limbs -> 32 canonical LE bytes (ok[i] = the limbs were below r, i.e. a value `Scalar` can hold);
// op 1 from_bytes: 32 LE bytes -> Montgomery limbs, ok[i] = 0 where the reference returns CtOption::none (integer >= r);
// op 2 from_bytes_wide: 64 LE bytes -> Montgomery limbs of the 512-bit integer mod r (always defined).  `ok` may be NULL.
__global__ void __launch_bounds__(256) k_fr_convert(int op, const u32* __restrict__ in, u32* __restrict__ out, uint8_t* __restrict__ ok, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr r; bool good = true;
  if (op == 2) {
    r = fr_from_wide(fr_load(in + i * 16), fr_load(in + i * 16 + 8));
  } else {
    const Fr a = fr_load(in + i * 8);
    good = fr_words_below_r(a.l);
    r = op == 0 ? fr_from_mont(a) : fr_to_mont(a);
  }
  fr_store(out + i * 8, r);
  if (ok) ok[i] = good ? 1 : 0;
}

// ---- twiddle table: tw[j] = 2^5 w^j, j < n/2 ---------------------------------------------------------------
// each lane starts from w^(64 t) (one exponentiation) and walks 64 consecutive powers
DEV size_t fr_tw_off(int lh) { return ((size_t)1 << lh) - 1; }      // element offset of level lh in the twiddle buffer (see below)
constexpr int FR_TW_RUN = 64;
__global__ void __launch_bounds__(256) k_fr_twiddles(u32* __restrict__ tw, int log_n, int inverse) {
  const size_t half = (size_t)1 << (log_n - 1);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j0 = t * FR_TW_RUN;
  if (j0 >= half) return;
  Fr w = fr_pow2k(fr_root_of_unity(), 32 - log_n);
  if (inverse) w = fr_pow_u64(w, ((u64)1 << log_n) - 1);          // w^-1 = w^(n-1): a short power instead of an inversion per lane
  Fr cur = fr_pow_u64(w, (u64)j0);
  for (int k = 0; k < 5; k++) cur = fr_add(cur, cur);          // the table holds w^j * 2^5 (see frl_mul)
  u32* top = tw + fr_tw_off(log_n - 1) * 8;
  for (int k = 0; k < FR_TW_RUN && j0 + k < half; k++) { fr_store(top + (j0 + k) * 8, cur); cur = fr_mul(cur, w); }
}

// compact per-stage tables: level lh (half-span 2^lh) holds 2^5 w_{2^(lh+1)}^i, i < 2^lh, at element offset 2^lh - 1, so
// that every stage reads its twiddles with unit stride (the full table is level log_n - 1)
__global__ void __launch_bounds__(256) k_fr_tw_levels(u32* __restrict__ tw, int log_n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // t + 1 in [1, 2^(log_n-1)): level = floor(log2(t+1))
  const int top = log_n - 1;
  if (t + 1 >= ((size_t)1 << top)) return;
  const int lh = 63 - __clzll((unsigned long long)(t + 1));
  const size_t i = (t + 1) - ((size_t)1 << lh);
  fr_store(tw + t * 8, fr_load(tw + (fr_tw_off(top) + (i << (top - lh))) * 8));
}

// ---- lazy 9 x 29-bit arithmetic for the butterflies -------------------------------------------------------------------
// Inside the transform kernels an element is nine 29-bit limbs (three spare bits per word) and the Montgomery factor is
// 2^261: a column of the product is at most 9 + 9 partial products below 2^58 times a small slack, so it accumulates in
// one 64-bit register with no carries (171 v_mad_u64_u32 against ~500 instructions for the 8 x 32-bit CIOS product), and
// additions / subtractions are nine carry-free VOP2 instructions.  The twiddles are stored multiplied by 2^5, so that
// frl_mul(x, w') = x w 2^5 / 2^261 = x w / 2^256 is still the reference's Montgomery product.  Bounds are tracked by hand
// ("A": limbs < A * 2^29, "V": value < V * r):
//   load        A1 V2   (memory between passes holds values in [0, 2r): r < 2^255, so 2r still fits eight words)
//   R-stage     a + b -> A2 V4 (kept);  (a + 4r - b) w' : A3 V6 -> product A1 V2
//   F-stage     inputs up to A2 V4:  a + b -> A4 V8 -> frl_reduce -> A1 V2;  (a + 8r - b) w' : A5 V12 -> A1 V2
// Column bound of the product: 9 * A * 2^58 + 9 * 2^58 < 2^64 needs A <= 6; value bound: a w' / 2^261 + r < 2r needs
// V <= 70 (2^261 / r = 70.6).
struct FrL { u32 l[9]; };
constexpr u32 M29 = (1u << 29) - 1;
struct FrL9 { u32 w[9]; };
constexpr FrL9 FR_MOD_L = {BLS_FR_MOD_L29};
DEV FrL frl_unpack(const Fr& a) {                  // eight 32-bit words -> nine 29-bit limbs
  FrL r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
    u64 two = (u64)a.l[k] | (k + 1 < 8 ? (u64)a.l[k + 1] << 32 : 0ull);
    r.l[i] = (u32)(two >> sh) & M29;
  }
  return r;
}
DEV Fr frl_pack(const FrL& a) {                    // normalised limbs (A1), value < 2^256
  Fr r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, i = bit / 29, sh = bit - 29 * i;       // word k starts inside limb i at bit sh
    u32 v = a.l[i] >> sh;
    if (i + 1 < 9) v |= a.l[i + 1] << (29 - sh);
    if (i + 2 < 9 && 58 - sh < 32) v |= a.l[i + 2] << (58 - sh);
    r.l[k] = v;
  }
  return r;
}
DEV FrL frl_load(const u32* p) { return frl_unpack(fr_load(p)); }
DEV FrL frl_add(const FrL& a, const FrL& b) { FrL r; for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i]; return r; }
template <int WHICH> DEV FrL frl_sub(const FrL& a, const FrL& b) {        // a + K r - b with the bias matching b's bounds
  constexpr FrL9 bias = WHICH == 1 ? FrL9{BLS_FR_BIAS_4_1} : FrL9{BLS_FR_BIAS_8_2};
  FrL r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + bias.w[i] - b.l[i];
  return r;
}
// (a * b) / 2^261 mod r, result A1 V2; a: A <= 6 / (A of b), b: A1 canonical twiddle (see the bound table above)
DEV FrL frl_mul(const FrL& a, const FrL& b) {
  u32 m[9];
  FrL r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * FR_MOD_L.w[k - i];
    m[k] = (0u - (u32)acc) & M29;                   // -r^-1 mod 2^29 = -1 (r = 1 mod 2^32)
    acc += (u64)m[k] * FR_MOD_L.w[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * FR_MOD_L.w[k - i];
    r.l[k - 9] = (u32)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (u32)acc;
  return r;
}
// A <= 4, V <= 8  ->  A1, value in [0, 2r): carry propagation, then q = floor(top / ((r >> 232) + 1)) <= floor(x / r)
// undershoots by at most one, so x - q r < 2r
DEV FrL frl_reduce(const FrL& a) {
  FrL t; u32 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { u32 v = a.l[i] + c; t.l[i] = v & M29; c = v >> 29; }
  t.l[8] = a.l[8] + c;
  const u32 q = t.l[8] / BLS_FR_TOP_P1;
  FrL r; int64_t cc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { int64_t v = (int64_t)t.l[i] - (int64_t)((u64)q * FR_MOD_L.w[i]) + cc; r.l[i] = (u32)v & M29; cc = v >> 29; }
  r.l[8] = (u32)((int64_t)t.l[8] - (int64_t)((u64)q * FR_MOD_L.w[8]) + cc);
  return r;
}
// A1, value < 2r  ->  canonical eight words
DEV Fr frl_canon(const FrL& a) { return fr_cond_sub(frl_pack(a), 0); }

// ---- decimation-in-frequency stages over global memory ------------------------------------------------------
// One radix-2 DIF stage with half-span h on a length-n vector: for each block of 2h elements,
//   (a, b) = (x[i], x[i+h])  ->  x[i] = a + b,  x[i+h] = (a - b) w^(i * n/(2h))
// k_fr_stage2 performs TWO consecutive stages (h and h/2) on four elements per lane.  Values in memory between
// passes are in [0, 2r) (only the last kernel canonicalises).
// (src and dst may be the same buffer: every lane reads its own elements before it writes them)
__global__ void __launch_bounds__(256) k_fr_stage1(const u32* src, u32* x, const u32* __restrict__ tw, int log_n, int log_h) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t half = (size_t)1 << (log_n - 1);
  if (t >= half) return;
  const size_t h = (size_t)1 << log_h;
  const size_t i = t & (h - 1), blk = t >> log_h;
  const size_t p = (blk << (log_h + 1)) + i;
  FrL a = frl_load(src + p * 8), b = frl_load(src + (p + h) * 8);
  FrL w = frl_load(tw + (fr_tw_off(log_h) + i) * 8);
  fr_store(x + p * 8, frl_pack(frl_reduce(frl_add(a, b))));
  fr_store(x + (p + h) * 8, frl_pack(frl_mul(frl_sub<1>(a, b), w)));
}
__global__ void __launch_bounds__(256) k_fr_stage2(const u32* src, u32* x, const u32* __restrict__ tw, int log_n, int log_h) {
  // stages with half-spans h = 2^log_h and h/2; lane t owns elements p, p + h/2, p + h, p + 3h/2
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t quarter = (size_t)1 << (log_n - 2);
  if (t >= quarter) return;
  const size_t q = (size_t)1 << (log_h - 1);           // h / 2
  const size_t i = t & (q - 1), blk = t >> (log_h - 1);
  const size_t p = (blk << (log_h + 1)) + i;
  FrL a0 = frl_load(src + p * 8), a1 = frl_load(src + (p + q) * 8), a2 = frl_load(src + (p + 2 * q) * 8), a3 = frl_load(src + (p + 3 * q) * 8);
  FrL w0 = frl_load(tw + (fr_tw_off(log_h) + i) * 8), w1 = frl_load(tw + (fr_tw_off(log_h) + i + q) * 8);
  FrL w2 = frl_load(tw + (fr_tw_off(log_h - 1) + i) * 8);          // second stage: half-span q, same offset i
  // stage h (R): pairs (a0, a2), (a1, a3)
  FrL b0 = frl_add(a0, a2), b2 = frl_mul(frl_sub<1>(a0, a2), w0);
  FrL b1 = frl_add(a1, a3), b3 = frl_mul(frl_sub<1>(a1, a3), w1);
  // stage h/2 (F): pairs (b0, b1), (b2, b3)
  fr_store(x + p * 8, frl_pack(frl_reduce(frl_add(b0, b1))));
  fr_store(x + (p + q) * 8, frl_pack(frl_mul(frl_sub<2>(b0, b1), w2)));
  fr_store(x + (p + 2 * q) * 8, frl_pack(frl_reduce(frl_add(b2, b3))));
  fr_store(x + (p + 3 * q) * 8, frl_pack(frl_mul(frl_sub<2>(b2, b3), w2)));
}

// ---- the last stages on a tile in LDS + bit-reversed store ---------------------------------------------------
#ifndef FR_TILE_WAVES
#define FR_TILE_WAVES 4
#endif
constexpr int FR_TILE_LOG = 10;                  // 1024 elements x 9 limbs = 36 KB of LDS per workgroup
// Runs the stages with half-spans 2^(tl-1) ... 1 on each aligned tile of 2^tl elements (tl = min(FR_TILE_LOG,
// log_n)), then writes element p of the (bit-reversed) result to its natural position bitrev(p), optionally
// scaled (the inverse transform's n^-1), in canonical form.  `x` is read, `y` written (they differ: the permutation
// is not in place).  Stages alternate R (sums kept unreduced) and F (sums reduced), see the bound table above.
__global__ void __launch_bounds__(256, FR_TILE_WAVES) k_fr_tile(const u32* __restrict__ x, u32* __restrict__ y, const u32* __restrict__ tw, int log_n,
                                                 int tl, const u32* __restrict__ scale) {
  extern __shared__ u32 lds[];                   // 2^tl elements, limb-interleaved: limb k of element e at lds[k * 2^tl + e]
  const int T = 1 << tl;
  const size_t base = (size_t)blockIdx.x << tl;
  for (int e = threadIdx.x; e < T; e += blockDim.x) {
    FrL v = frl_load(x + (base + e) * 8);
#pragma unroll
    for (int k = 0; k < 9; k++) lds[k * T + e] = v.l[k];
  }
  __syncthreads();
  // Two stages per round trip through LDS: lane t owns the four elements p, p + q, p + 2q, p + 3q (q = half-span of the second
  // stage) and runs an R stage (sums kept unreduced) and an F stage (sums reduced) on them in registers, exactly like
  // k_fr_stage2 -- half the LDS traffic and half the barriers of one stage per pass.  An odd tile depth leaves one R stage.
  int lh = tl - 1;
  for (; lh >= 1; lh -= 2) {
    const int q = 1 << (lh - 1);
    for (int t = threadIdx.x; t < T / 4; t += blockDim.x) {
      const int i = t & (q - 1), p = ((t >> (lh - 1)) << (lh + 1)) + i;
      FrL a0, a1, a2, a3;
#pragma unroll
      for (int k = 0; k < 9; k++) { a0.l[k] = lds[k * T + p]; a1.l[k] = lds[k * T + p + q]; a2.l[k] = lds[k * T + p + 2 * q]; a3.l[k] = lds[k * T + p + 3 * q]; }
      FrL w0 = frl_load(tw + (fr_tw_off(lh) + i) * 8), w1 = frl_load(tw + (fr_tw_off(lh) + i + q) * 8);
      FrL w2 = frl_load(tw + (fr_tw_off(lh - 1) + i) * 8);
      // stage with half-span 2q (R): pairs (a0, a2), (a1, a3)
      FrL b0 = frl_add(a0, a2), b2 = frl_mul(frl_sub<1>(a0, a2), w0);
      FrL b1 = frl_add(a1, a3), b3 = frl_mul(frl_sub<1>(a1, a3), w1);
      // stage with half-span q (F): pairs (b0, b1), (b2, b3)
      FrL c0 = frl_reduce(frl_add(b0, b1)), c1 = frl_mul(frl_sub<2>(b0, b1), w2);
      FrL c2 = frl_reduce(frl_add(b2, b3)), c3 = frl_mul(frl_sub<2>(b2, b3), w2);
#pragma unroll
      for (int k = 0; k < 9; k++) { lds[k * T + p] = c0.l[k]; lds[k * T + p + q] = c1.l[k]; lds[k * T + p + 2 * q] = c2.l[k]; lds[k * T + p + 3 * q] = c3.l[k]; }
    }
    __syncthreads();
  }
  if (lh == 0) {                                   // odd depth: one more stage (R), half-span 1
    for (int t = threadIdx.x; t < T / 2; t += blockDim.x) {
      const int p = t << 1;
      FrL a, b;
#pragma unroll
      for (int k = 0; k < 9; k++) { a.l[k] = lds[k * T + p]; b.l[k] = lds[k * T + p + 1]; }
      FrL w = frl_load(tw + (fr_tw_off(0)) * 8);
      FrL s2 = frl_add(a, b), d = frl_mul(frl_sub<1>(a, b), w);
#pragma unroll
      for (int k = 0; k < 9; k++) { lds[k * T + p] = s2.l[k]; lds[k * T + p + 1] = d.l[k]; }
    }
    __syncthreads();
  }
  FrL sc;
  if (scale) sc = frl_load(scale);
  for (int e = threadIdx.x; e < T; e += blockDim.x) {
    FrL v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = lds[k * T + e];
    if (scale) v = frl_mul(v, sc);                // A <= 2 V <= 4 -> A1 V2
    else v = frl_reduce(v);
    const size_t p = base + e;
    const size_t r = (size_t)(__brevll((unsigned long long)p) >> (64 - log_n));
    fr_store(y + r * 8, frl_canon(v));
  }
}
// ---- the TOP stages on column tiles in LDS (round 5) -------------------------------------------------------------------------------
// The stages with half-spans 2^lh_top ... 2^(lh_top-d+1) only couple elements whose indices differ in bits [lh_top-d+1, lh_top]: a
// "column" of 2^d elements S = 2^ls apart (ls = lh_top - d + 1).  A workgroup takes K = 2^lk ADJACENT columns -- K * 32 contiguous
// bytes per row, 2 KB for the 2^20 transform -- into LDS (element (row, col) at tile index row * K + col, limb-interleaved like
// k_fr_tile), runs all d stages there with the radix-4 schedule of k_fr_tile (R stage + F stage per round trip, twiddles read from the
// per-level tables at the element's GLOBAL offset), and writes the tile back where it came from with values in [0, 2r).  One pass over
// the data instead of d / 2: the 2^20 transform is two such passes of five stages + k_fr_tile (ten stages) = three passes instead of
// six, the 2^24 transform 2 x 7 stages + k_fr_tile instead of eight passes (the shape is the host's: api_aux.hip::blsgpu_fr_ntt_device).
// Block b = (hi, chunk): hi = b >> (ls - lk) selects the aligned block of 2^(lh_top+1) elements, chunk the K columns inside a stride.
constexpr int FR_COLS_LOG = 12;                    // largest tile the host may ask for: 4096 elements x 9 limbs = 144 KB of the CU's 160 KB
constexpr int FR_COLS_BLOCK = 1024;
__global__ void __launch_bounds__(FR_COLS_BLOCK) k_fr_cols(const u32* src, u32* dst, const u32* __restrict__ tw, int lh_top, int d, int lk) {
  extern __shared__ u32 lds[];
  const int T = 1 << (d + lk), K = 1 << lk;
  const int ls = lh_top - d + 1;
  const size_t lo = ((size_t)blockIdx.x & (((size_t)1 << (ls - lk)) - 1)) << lk;
  const size_t base = (((size_t)blockIdx.x >> (ls - lk)) << (lh_top + 1)) + lo;
  for (int e = threadIdx.x; e < T; e += blockDim.x) {
    FrL v = frl_load(src + (base + ((size_t)(e >> lk) << ls) + (e & (K - 1))) * 8);
#pragma unroll
    for (int k = 0; k < 9; k++) lds[k * T + e] = v.l[k];
  }
  __syncthreads();
  int s = d;                                       // stages left; the next one has local half-span 2^(s-1) * K, global 2^(s-1) * S
  for (; s >= 2; s -= 2) {
    const int lq = s - 2 + lk, q = 1 << lq;        // local half-span of the SECOND stage of this round trip
    const int gh = s - 1 + ls;                     // log2 of the global half-span of the first
    for (int t = threadIdx.x; t < T / 4; t += blockDim.x) {
      const int i = t & (q - 1), p = ((t >> lq) << (lq + 2)) + i;
      const size_t gi = ((size_t)(i >> lk) << ls) + lo + (size_t)(i & (K - 1));      // offset of p inside the second stage's global half-span
      FrL a0, a1, a2, a3;
#pragma unroll
      for (int k = 0; k < 9; k++) { a0.l[k] = lds[k * T + p]; a1.l[k] = lds[k * T + p + q]; a2.l[k] = lds[k * T + p + 2 * q]; a3.l[k] = lds[k * T + p + 3 * q]; }
      FrL w0 = frl_load(tw + (fr_tw_off(gh) + gi) * 8), w1 = frl_load(tw + (fr_tw_off(gh) + gi + ((size_t)1 << (gh - 1))) * 8);
      FrL w2 = frl_load(tw + (fr_tw_off(gh - 1) + gi) * 8);
      FrL b0 = frl_add(a0, a2), b2 = frl_mul(frl_sub<1>(a0, a2), w0);
      FrL b1 = frl_add(a1, a3), b3 = frl_mul(frl_sub<1>(a1, a3), w1);
      FrL c0 = frl_reduce(frl_add(b0, b1)), c1 = frl_mul(frl_sub<2>(b0, b1), w2);
      FrL c2 = frl_reduce(frl_add(b2, b3)), c3 = frl_mul(frl_sub<2>(b2, b3), w2);
#pragma unroll
      for (int k = 0; k < 9; k++) { lds[k * T + p] = c0.l[k]; lds[k * T + p + q] = c1.l[k]; lds[k * T + p + 2 * q] = c2.l[k]; lds[k * T + p + 3 * q] = c3.l[k]; }
    }
    __syncthreads();
  }
  if (s == 1) {                                    // odd depth: one more stage (R), local half-span K, global half-span S
    for (int t = threadIdx.x; t < T / 2; t += blockDim.x) {
      const int i = t & (K - 1), p = ((t >> lk) << (lk + 1)) + i;
      FrL a, b;
#pragma unroll
      for (int k = 0; k < 9; k++) { a.l[k] = lds[k * T + p]; b.l[k] = lds[k * T + p + K]; }
      FrL w = frl_load(tw + (fr_tw_off(ls) + lo + (size_t)i) * 8);
      FrL s2 = frl_add(a, b), df = frl_mul(frl_sub<1>(a, b), w);
#pragma unroll
      for (int k = 0; k < 9; k++) { lds[k * T + p] = s2.l[k]; lds[k * T + p + K] = df.l[k]; }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < T; e += blockDim.x) {
    FrL v;
#pragma unroll
    for (int k = 0; k < 9; k++) v.l[k] = lds[k * T + e];
    fr_store(dst + (base + ((size_t)(e >> lk) << ls) + (e & (K - 1))) * 8, frl_pack(frl_reduce(v)));
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
  for (int k = 0; k < 5; k++) r = fr_add(r, r);               // pre-scaled by 2^5 like the twiddles
  fr_store(out, r);
}

}  // namespace bls
