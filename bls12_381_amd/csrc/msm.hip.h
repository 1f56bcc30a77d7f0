// msm.hip.h -- Pippenger multi-scalar multiplication  sum_i s_i * P_i  for G1 and G2 (templated on the
// field policy).
//
// The reference has no MSM routine; it defines the result as  points.zip(scalars).map(|(p,s)| p*s).sum()
// i.e. 255-step double-and-add per point (src/g1.rs:754-774, src/g2.rs:825-845) folded with `Sum`
// (src/g1.rs:161-171).  The same group element is produced here with the bucket method:
//
//   0. split       G1: GLV, k = k1 + k2 z^2 with 127-bit halves over the bases and their images under the endomorphism
//                  (k_glv_decompose); G2: four 63-bit digits of base |x| over the images under psi^j (k_gls_decompose);
//                  also the canonical-scalar check.  Plain 256-bit scalars when the images are not resident / too many.
//   1.-3. sort     signed c-bit digits (DigitIter); two-level counting sort of (sign, point index) by (window, |digit|):
//                  coarse 8 bits through per-tile LDS histograms, fine 7 bits per region in LDS (k_sort_hist / scan /
//                  scatter / fine); beyond 2^24 entries per window set the global-atomic fallback (k_msm_digits, k_msm_scatter)
//   4. items       buckets cut into work items of <= cap entries (cap ~ 4x the mean load), sorted by length (descending) so
//                  that the 64 lanes of a wavefront walk items of equal length and no lane walks a long bucket
//   5. accumulate  one lane per item (G2: one lane PAIR): gathers its points (128 B / 256 B records) and adds them with the
//                  XYZZ mixed addition of curve.hip.h, exceptional cases exact -- this is >90% of the arithmetic; partial
//                  sums of buckets that were cut are folded by k_msm_heavy
//   6. reduce      sum_k k * B_k per window by chunked running sums, fan 8; the two running sums of a chain on two lanes
//                  (bottom level) / two 8-lane teams (above); the T records of all levels tree-summed by multi-job launches
//   7. combine     Horner over the levels (3 doublings + 1 addition each) and over the windows (c doublings + 1 addition)
//
// Data layout in HBM: bases are converted once at upload to the internal field form and stored as
// 128-byte (G1) / 256-byte (G2) records, so a gather is 8 / 16 aligned 16-byte loads per lane; scalars
// are read coalesced exactly once; the sorted index array and the bucket array are the only large
// temporaries (4 B per (point, window) and 176 / 336 B per bucket).
#pragma once
#include <type_traits>
#include "convert.hip.h"
#include "scalar.hip.h"
#include "limits.h"
#include "team.hip.h"
#include "pairlane.hip.h"

namespace bls {

// ---- resident storage records ---------------------------------------------------------------------
template <class F> struct Store;
template <> struct Store<FpPolicy> {
  static constexpr int AFF_WORDS = 32;    // x[14] y[14] inf pad[3]          (128 B)
  static constexpr int PROJ_WORDS = 44;   // x[14] y[14] z[14] pad[2]        (176 B)
  static constexpr int EL = NL;
  static DEV void ld(const u32* w, fe1& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) a.l[i] = w[i];
  }
  template <class T> static DEV void st(u32* w, const T& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) w[i] = a.l[i];
  }
  static DEV void ldw(const u32* w, fe& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) a.l[i] = w[i];
  }
};
template <> struct Store<Fp2Policy> {
  static constexpr int AFF_WORDS = 64;    // x0 x1 y0 y1 (14 each) inf pad[7]  (256 B)
  static constexpr int PROJ_WORDS = 84;   // 6 x 14                            (336 B)
  static constexpr int EL = 2 * NL;
  static DEV void ld(const u32* w, fe2_1& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) { a.c0.l[i] = w[i]; a.c1.l[i] = w[NL + i]; }
  }
  template <class T> static DEV void st(u32* w, const T& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) { w[i] = a.c0.l[i]; w[NL + i] = a.c1.l[i]; }
  }
  static DEV void ldw(const u32* w, fe2& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) { a.c0.l[i] = w[i]; a.c1.l[i] = w[NL + i]; }
  }
};

template <class F> DEV void load_aff(const u32* rec, Aff<F>& q, bool& inf) {
  constexpr int EL = Store<F>::EL;
  // 16-byte vector loads of the whole record
  u32 w[2 * EL + 4];
  const uint4* v = reinterpret_cast<const uint4*>(rec);
#pragma unroll
  for (int i = 0; i < (2 * EL + 4) / 4; i++) {
    uint4 t = v[i];
    w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
  }
  Store<F>::ld(w, q.x);
  Store<F>::ld(w + EL, q.y);
  inf = w[2 * EL] != 0;
}
// the same with the identity flag left as the stored word: a prefetching loop must not test it before the record is needed
// (the comparison is a use of the load: `bool` made the accumulation loop wait for the record it had just requested)
template <class F> DEV void load_aff_word(const u32* rec, Aff<F>& q, u32& flag) {
  constexpr int EL = Store<F>::EL;
  u32 w[2 * EL + 4];
  const uint4* v = reinterpret_cast<const uint4*>(rec);
#pragma unroll
  for (int i = 0; i < (2 * EL + 4) / 4; i++) {
    uint4 t = v[i];
    w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
  }
  Store<F>::ld(w, q.x);
  Store<F>::ld(w + EL, q.y);
  flag = w[2 * EL];
}
template <class F> DEV void load_proj(const u32* rec, Proj<F>& p) {
  constexpr int EL = Store<F>::EL;
  Store<F>::ldw(rec, p.x); Store<F>::ldw(rec + EL, p.y); Store<F>::ldw(rec + 2 * EL, p.z);
}
template <class F> DEV void store_proj(u32* rec, const Proj<F>& p) {
  constexpr int EL = Store<F>::EL;
  Store<F>::st(rec, p.x); Store<F>::st(rec + EL, p.y); Store<F>::st(rec + 2 * EL, p.z);
}

// negate an affine point's y lazily, keeping one static type for both signs
DEV Fe<2, 2> cond_neg(const fe1& y, bool n) { return select(n, neg(y), (Fe<2, 2>)y); }
DEV Fe2<2, 2> cond_neg(const fe2_1& y, bool n) { return select(n, neg(y), (Fe2<2, 2>)y); }

// mixed addition with a sign-adjusted affine operand (same formula as pt_add_mixed; y has bound <2,2>)
template <class F, class YT>
DEV Proj<F> pt_add_mixed_y(const Proj<F>& p, const typename F::aff_elem& qx, const YT& qy) {
  auto t0 = mul(p.x, qx);
  auto t1 = mul(p.y, qy);
  auto t3 = mul(norm(add(qx, qy)), add(p.x, p.y));
  auto t4 = add(t0, t1);
  auto t3b = norm(sub(t3, t4));
  auto t4b = norm(add(mul(qy, p.z), p.y));
  auto y3 = norm(add(mul(qx, p.z), p.x));
  auto t0b = norm(add(dbl(t0), t0));
  auto t2 = F::mul_by_3b(p.z);
  auto z3 = norm(add(t1, t2));
  auto t1b = norm(sub(t1, t2));
  auto y3b = F::mul_by_3b(y3);
  auto x3 = mul(t4b, y3b);
  auto t2b = mul(t3b, t1b);
  auto x3b = sub(t2b, x3);
  auto y3c = mul(y3b, t0b);
  auto t1c = mul(t1b, z3);
  auto y3d = add(t1c, y3c);
  auto t0c = mul(t0b, t3b);
  auto z3b = mul(z3, t4b);
  auto z3c = add(z3b, t0c);
  Proj<F> r;
  r.x = F::st(x3b); r.y = F::st(y3d); r.z = F::st(z3c);
  return r;
}

// The reference's Scalar is always canonical (Scalar::from_bytes rejects values >= r, scalar.rs:256-280), and the digit
// recoding relies on it: scalars < r < 2^255 leave the top window a spare bit, so no carry leaves it.  A raw 32-byte input
// that is NOT canonical is reported through a sticky device flag (blsgpu_synchronize / the synchronous MSM entry points
// return BLSGPU_ERR_ARG) instead of silently producing s*P for some window widths and (s - 2^256)*P for others.
DEV bool scalar_is_canonical(const u32* s) { return fr_words_below_r(s); }
// `&[Scalar]` memory (Montgomery limbs, scalar.hip.h) -> the canonical words the digit kernels read: used where no decomposition
// kernel touches the scalars first (plain 256-bit windows); with the endomorphism splits the reduction is fused into k_glv_decompose /
// k_gls_decompose.  scalar.rs:284-296.
__global__ void __launch_bounds__(256) k_scalars_from_mont(const u32* __restrict__ scalars, u32* __restrict__ out, int n, u32* __restrict__ status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr k;
  if (!scalar_load(scalars, (size_t)i, SCALAR_MONT, k.l)) atomicOr(status, 1u);
  fr_store(out + (size_t)i * 8, k);
}

// ---- 1. digits + histogram -------------------------------------------------------------------------
// ent[w * n + i] = global bucket (w * nbw + |d| - 1) | sign << 31, or 0xffffffff for a zero digit;
// rank[w * n + i] = arrival order of the entry inside its bucket (the value returned by the histogram
// atomic), which makes the later scatter a plain permutation with no second round of atomics.
__global__ void __launch_bounds__(256) k_msm_digits(const u32* __restrict__ scalars, u32* __restrict__ ent, u32* __restrict__ rank,
                                                    u32* __restrict__ hist, int n, int c, int nwin, u32* __restrict__ status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[9];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  uint4 a = sp[0], b = sp[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; s[8] = 0;
  if (!scalar_is_canonical(s)) atomicOr(status, 1u);
  const u32 nbw = 1u << (c - 1);
  const u32 mask = (1u << c) - 1;
  u32 carry = 0;
  for (int w = 0; w < nwin; w++) {
    int bit = w * c, lo = bit >> 5, sh = bit & 31;
    u32 raw = 0;
    if (lo < 8) {
      u64 t = ((u64)s[lo + 1] << 32 | s[lo]) >> sh;
      raw = (u32)t & mask;
    }
    raw += carry;
    u32 neg = raw > nbw;
    u32 mag = neg ? ((1u << c) - raw) : raw;
    carry = neg;
    u32 e = 0xffffffffu, rk = 0;
    if (mag) {
      u32 gb = (u32)w * nbw + (mag - 1);
      e = gb | (neg << 31);
      rk = atomicAdd(&hist[gb], 1u);
    }
    ent[(size_t)w * n + i] = e;
    rank[(size_t)w * n + i] = rk;
  }
}

// ---- 0. resident window-shifted tables (optional, fixed bases) ---------------------------------------------
// table[w * n + i] = affine([2^(c w)] P_i): with them every window of every scalar lands in ONE set of
// 2^(c-1) buckets -- no Horner over the windows, a 16x smaller bucket reduction, and c can grow to 20 (13
// windows instead of 16).  Costs W x the resident memory (1.7 GB for 2^20 points: nothing against 288 GB) and
// one pass of c doublings + an inversion per entry at upload.
template <class F>
__global__ void __launch_bounds__(256) k_bases_precompute(const u32* __restrict__ rec, u32* __restrict__ table, size_t n, int c, int nwin) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int EL = Store<F>::EL, AW = Store<F>::AFF_WORDS;
  Aff<F> q; bool inf;
  load_aff<F>(rec + i * AW, q, inf);
  Proj<F> p; p.x = F::st(q.x); p.y = F::st(q.y); p.z = inf ? F::zero() : F::one();
  for (int w = 0; w < nwin; w++) {
    u32* r = table + ((size_t)w * n + i) * AW;
    if (w > 0) for (int k = 0; k < c; k++) p = pt_double<F>(p);
    bool zz = is_zero(p.z);
    auto zi = inv(p.z);
    auto x = canon_any(mul(p.x, zi));
    auto y = canon_any(mul(p.y, zi));
    Store<F>::st(r, x); Store<F>::st(r + EL, y);
    r[2 * EL] = zz ? 1 : 0;
    for (int j = 2 * EL + 1; j < AW; j++) r[j] = 0;
  }
}

// ---- 1'. two-level counting sort (default path, n <= 2^24, c <= 16) ------------------------------------
// The (window, |digit|) key is split into a coarse part (<= 8 bits) and a fine part (<= 7 bits).
//   k_sort_hist     per tile of scalars: LDS histogram over (window, coarse) -> global counters
//   k_sort_scan     exclusive scan of the nwin * ncoarse counters (one block)
//   k_sort_scatter  per tile: reserve a slice of every (window, coarse) region with one global atomic per
//                   non-empty counter, rank the tile's entries with LDS atomics, write packed entries
//                   (index:24 | sign:1 | fine:7)
//   k_sort_fine     one block per (window, coarse) region: LDS histogram over the fine key, scan, place ->
//                   `sorted` grouped by bucket and the bucket offsets `offs` (no global scan over the buckets)
// Compared with one global atomic + one random 4-byte gather/scatter per entry, almost all atomics are LDS
// atomics and the random traffic stays inside a 16 KB region.
// tile / block size of the coarse sort passes: every tile costs one global atomic per non-empty (window, coarse) counter in
// each pass, so larger tiles mean fewer of them and longer runs per region (measured at 2^21 half-scalars: 2048/256 -> 0.40 ms for
// hist + scan + scatter, 4096/512 -> 0.29 ms, 8192/1024 -> 0.30 ms, 16384/1024 -> 0.37 ms)
#ifndef BLS_SORT_TILE
#define BLS_SORT_TILE 4096
#endif
#ifndef BLS_SORT_THREADS
#define BLS_SORT_THREADS 512
#endif
constexpr int SORT_TILE = BLS_SORT_TILE;         // scalars per block in k_sort_hist / k_sort_scatter
constexpr int SORT_THREADS = BLS_SORT_THREADS;   // threads per block of those two kernels
constexpr int SORT_MAX_COUNTERS = 8192;         // nwin * ncoarse upper bound (LDS: 32 KB)

// Signed c-bit digits of a scalar, lowest window first.  WORDS = 8: a canonical 32-byte scalar.  WORDS = 4: one half of a GLV
// decomposition (k_glv_decompose below): a 127-bit magnitude with the sign of its contribution in bit 127.  WORDS = 2: one of the
// four 63-bit digits of the G2 decomposition (k_gls_decompose), sign in bit 63.  The scalar is kept
// as a shift register (funnel shifts with static register indices), so the iterator lives entirely in VGPRs.
template <int WORDS> struct DigitIter {
  u32 s[WORDS]; u32 carry, sign; int c; u32 nbw, mask;
  DEV void init(const u32* src, size_t i, int c_) {
    if constexpr (WORDS == 2) {
      uint2 a = *reinterpret_cast<const uint2*>(src + i * 2);
      s[0] = a.x; s[1] = a.y; sign = s[1] >> 31; s[1] &= 0x7fffffffu;
    } else {
      const uint4* sp = reinterpret_cast<const uint4*>(src + i * WORDS);
      uint4 a = sp[0];
      s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
      if constexpr (WORDS == 8) { uint4 b = sp[1]; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; sign = 0; }
      else { sign = s[3] >> 31; s[3] &= 0x7fffffffu; }
    }
    carry = 0; c = c_; nbw = 1u << (c - 1); mask = (1u << c) - 1;
  }
  DEV bool canonical() const { if constexpr (WORDS == 8) return scalar_is_canonical(s); else return true; }
  // next window (call once per window, in order): magnitude (0 = no entry) and the sign of the entry
  DEV void next(u32& mag, u32& neg) {
    u32 raw = (s[0] & mask) + carry;
#pragma unroll
    for (int j = 0; j < WORDS - 1; j++) s[j] = (s[j] >> c) | (s[j + 1] << (32 - c));
    s[WORDS - 1] >>= c;
    u32 over = raw > nbw;
    mag = over ? ((1u << c) - raw) : raw;
    carry = over;
    neg = over ^ sign;
  }
};

// ---- GLV decomposition (G1) --------------------------------------------------------------------------------------------
// phi(x, y) = (BETA x, y) is the endomorphism of g1.rs:421-437; the reference's subgroup check (:396-405) states
// phi(P) = -[z^2] P.  With L = z^2 (128 bits; r = L^2 - L + 1) a scalar k < r splits as k = k1 + k2 L and, using L^2 = L - 1
// (mod r), into BALANCED halves |k1|, |k2| <= L/2 + 1 < 2^126.5:
//     k P = k1 P + k2 [L] P = sign(k1) |k1| P  -  sign(k2) |k2| phi(P).
// The MSM then runs over 2n points (the resident bases and their images under phi, stored next to them) with 127-bit
// scalars: the same number of bucket additions, but HALF the windows -- half the buckets to reduce and half the doublings
// in the window combine.  out[i] = |k1|, out[n + i] = |k2| as four words each; bit 127 = 1 if the term is SUBTRACTED.
typedef unsigned __int128 u128;
// k (eight words, canonical) -> |k1|, |k2| as four words each; bit 127 of k1 = 1 if the P term is SUBTRACTED, bit 127 of k2 = 1 if the
// phi(P) term is SUBTRACTED
DEV void glv_split(const u32* k, u32* k1o, u32* k2o) {
  constexpr u32 Lw[4] = BLS_GLV_L_W, Mw[5] = BLS_GLV_M_W, Hw[4] = BLS_GLV_H_W;
  // q = floor(k M / 2^256) in {floor(k / L) - 1, floor(k / L)}   (M = floor(2^256 / L))
  u32 q[5];
  u128 acc = 0;
#pragma unroll
  for (int col = 0; col < 13; col++) {
#pragma unroll
    for (int x = 0; x < 8; x++) { int y = col - x; if (y >= 0 && y < 5) acc += (u64)k[x] * Mw[y]; }
    if (col >= 8) q[col - 8] = (u32)acc;
    acc >>= 32;
  }
  // k1 = k - q L  (five words are enough: k1 < 2L)
  u32 t[5];
  acc = 0;
#pragma unroll
  for (int col = 0; col < 5; col++) {
#pragma unroll
    for (int x = 0; x < 4; x++) { int y = col - x; if (y >= 0 && y < 4) acc += (u64)q[x] * Lw[y]; }
    t[col] = (u32)acc;
    acc >>= 32;
  }
  u32 k1[5];
  {
    int64_t br = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { int64_t d = (int64_t)k[j] - (int64_t)t[j] + br; k1[j] = (u32)d; br = d >> 32; }
  }
  auto ge4 = [](const u32* x, const u32* y) { bool gt = false, eq = true; for (int j = 3; j >= 0; j--) { gt = gt || (eq && x[j] > y[j]); eq = eq && x[j] == y[j]; } return gt || eq; };
  auto gt4 = [](const u32* x, const u32* y) { bool gt = false, eq = true; for (int j = 3; j >= 0; j--) { gt = gt || (eq && x[j] > y[j]); eq = eq && x[j] == y[j]; } return gt; };
  auto sub4 = [](u32* r, const u32* x, const u32* y) { int64_t br = 0; for (int j = 0; j < 4; j++) { int64_t d = (int64_t)x[j] - (int64_t)y[j] + br; r[j] = (u32)d; br = d >> 32; } };
  auto addsmall4 = [](u32* x, int v) { int64_t cy = v; for (int j = 0; j < 4; j++) { int64_t d = (int64_t)x[j] + cy; x[j] = (u32)d; cy = d >> 32; } };
  u32 k2[4] = {q[0], q[1], q[2], q[3]};
  if (k1[4] != 0 || ge4(k1, Lw)) { sub4(k1, k1, Lw); addsmall4(k2, 1); }          // the Barrett estimate was one short
  // balance: k1 in (-L/2 - 1, L/2], k2 in (-L/2, L/2]
  u32 neg1 = 0, neg2 = 0;
  if (gt4(k1, Hw)) { sub4(k1, Lw, k1); neg1 = 1; addsmall4(k2, 1); }              // k1 - L, carried into k2
  if (gt4(k2, Hw)) {
    u32 lm1[4] = {Lw[0], Lw[1], Lw[2], Lw[3]}; addsmall4(lm1, -1);
    sub4(k2, lm1, k2); neg2 = 1;                                                  // k2 L = (k2 - L + 1) L - 1  (mod r)
    bool z1 = (k1[0] | k1[1] | k1[2] | k1[3]) == 0;
    if (neg1) addsmall4(k1, 1); else if (z1) { k1[0] = 1; neg1 = 1; } else addsmall4(k1, -1);
  }
  // k P = (neg1 ? -1 : 1) |k1| P  +  k2 (-phi(P)):  the phi term is subtracted when k2 > 0
  u32 sub2 = neg2 ? 0u : 1u;
  k1o[0] = k1[0]; k1o[1] = k1[1]; k1o[2] = k1[2]; k1o[3] = k1[3] | (neg1 << 31);
  k2o[0] = k2[0]; k2o[1] = k2[1]; k2o[2] = k2[2]; k2o[3] = k2[3] | (sub2 << 31);
}
__global__ void __launch_bounds__(256) k_glv_decompose(const u32* __restrict__ scalars, u32* __restrict__ out, int n, u32* __restrict__ status, int form) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[8];
  if (!scalar_load(scalars, (size_t)i, form, k)) atomicOr(status, 1u);
  u32 k1[4], k2[4];
  glv_split(k, k1, k2);
  uint4* o = reinterpret_cast<uint4*>(out);
  o[i] = make_uint4(k1[0], k1[1], k1[2], k1[3]);
  o[(size_t)n + i] = make_uint4(k2[0], k2[1], k2[2], k2[3]);
}
// the images under phi of resident G1 bases: (BETA x, y), same flag
__global__ void __launch_bounds__(256) k_bases_endo(const u32* __restrict__ rec, u32* __restrict__ endo, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int AW = Store<FpPolicy>::AFF_WORDS;
  constexpr PLimbs kb = {BLS_BETA};
  const u32* r = rec + i * AW; u32* e = endo + i * AW;
  fe1 x; Store<FpPolicy>::ld(r, x);
  fe1 bx = canon(mul(x, fe1_const(kb)));
  Store<FpPolicy>::st(e, bx);
  for (int j = NL; j < AW; j++) e[j] = r[j];
}

// ---- four-dimensional decomposition (G2) ------------------------------------------------------------------------------
// psi (g2.rs:847-890: the untwist-Frobenius-twist endomorphism) acts on the order-r subgroup of the twist as multiplication by
// the curve parameter x = -X, X = 0xd201000000010000 -- that is the reference's own subgroup test psi(P) == [x] P
// (g2.rs:475-482) -- and r = x^4 - x^2 + 1.  A scalar k < r < X^4 has four base-X digits k = d0 + d1 X + d2 X^2 + d3 X^3, so
//     k P = d0 P  -  d1 psi(P)  +  d2 psi^2(P)  -  d3 psi^3(P).
// Digits are balanced into (-X/2, X/2] (a carry out of d3 is folded back with x^4 = x^2 - 1 (mod r): d2 += 1, d0 -= 1), which
// leaves |d_i| <= X/2 + 1 < 2^63: the top 16-bit window never exceeds 0x6901, so the signed recoding needs no extra window.
// The MSM then runs over 4n points (every base with its three images, interleaved in one resident array) with 63-bit scalars:
// the same 16 n bucket additions, but FOUR windows instead of sixteen -- a quarter of the buckets to reduce and 48 instead of
// 240 doublings in the window combine.  out[4 i + j] = |d_j| (two words), bit 63 = 1 if the term is SUBTRACTED.
// Division by X: Knuth's algorithm D on 32-bit digits (X is normalised: its top bit is set), two corrections at most.
DEV void gls_divmod_x(u32* u, int m, u32* q) {        // u: m + 2 words (top word 0 on entry), q: m words; remainder left in u[0..1]
  constexpr u64 X = 0xd201000000010000ull;
  constexpr u32 v1 = (u32)(X >> 32);
  for (int j = m - 1; j >= 0; j--) {
    u64 hi = ((u64)u[j + 2] << 32) | u[j + 1];
    u64 qh = hi / v1;
    if (qh > 0xffffffffull) qh = 0xffffffffull;
    // t = u[j+2 : j] - qh * X as a signed 128-bit value
    __int128 t = ((__int128)(((unsigned __int128)u[j + 2] << 64) | ((unsigned __int128)u[j + 1] << 32) | u[j])) - (__int128)((unsigned __int128)qh * X);
    while (t < 0) { t += (__int128)X; qh--; }
    q[j] = (u32)qh;
    u[j] = (u32)(u64)t; u[j + 1] = (u32)((u64)t >> 32); u[j + 2] = 0;
  }
}
// k (TEN words: the canonical scalar in k[0..7], k[8] = k[9] = 0; clobbered) -> the four digits |d_j| < 2^63 and, in sub[j], whether the
// term d_j psi^j(P) is SUBTRACTED
DEV void gls_split(u32* k, u64* d_out, u32* sub) {
  constexpr u64 X = 0xd201000000010000ull, H = X >> 1;
  u64 d[5];
  u32 q0[8], q1[6];
  gls_divmod_x(k, 7, q0);                       // k = q0 X + d0,  q0 < 2^192
  d[0] = ((u64)k[1] << 32) | k[0];
  u32 w[8];
#pragma unroll
  for (int j = 0; j < 6; j++) w[j] = q0[j];
  w[6] = 0; w[7] = 0;
  gls_divmod_x(w, 5, q1);                       // q0 = q1 X + d1,  q1 < 2^128
  d[1] = ((u64)w[1] << 32) | w[0];
  u32 z[6];
#pragma unroll
  for (int j = 0; j < 4; j++) z[j] = q1[j];
  z[4] = 0; z[5] = 0;
  u32 q2[4];
  gls_divmod_x(z, 3, q2);                       // q1 = d3 X + d2,  d3 < X
  d[2] = ((u64)z[1] << 32) | z[0];
  d[3] = ((u64)q2[1] << 32) | q2[0];
  // balance (digits as signed 65-bit values: magnitude + sign)
  u32 neg[4];
  u64 carry = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    u64 v = d[j] + carry;                       // <= X: no wrap
    if (v > H) { d[j] = X - v; neg[j] = 1; carry = 1; } else { d[j] = v; neg[j] = 0; carry = 0; }
  }
  if (carry) {                                  // X^4 = x^4 = x^2 - 1 (mod r):  d2 += 1, d0 -= 1   (signed)
    if (neg[2]) { if (d[2] == 0) { d[2] = 1; neg[2] = 0; } else d[2] -= 1; } else d[2] += 1;
    if (neg[0]) d[0] += 1; else if (d[0] == 0) { d[0] = 1; neg[0] = 1; } else d[0] -= 1;
  }
  // k P = d0 P - d1 psi(P) + d2 psi^2(P) - d3 psi^3(P):  odd terms are subtracted when their digit is positive
  sub[0] = neg[0]; sub[1] = neg[1] ^ 1u; sub[2] = neg[2]; sub[3] = neg[3] ^ 1u;
  d_out[0] = d[0]; d_out[1] = d[1]; d_out[2] = d[2]; d_out[3] = d[3];
}
__global__ void __launch_bounds__(256) k_gls_decompose(const u32* __restrict__ scalars, u32* __restrict__ out, int n, u32* __restrict__ status, int form) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[10];
  if (!scalar_load(scalars, (size_t)i, form, k)) atomicOr(status, 1u);
  k[8] = 0; k[9] = 0;
  u64 d[4]; u32 sb[4];
  gls_split(k, d, sb);
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  o[0] = make_uint4((u32)d[0], (u32)(d[0] >> 32) | (sb[0] << 31), (u32)d[1], (u32)(d[1] >> 32) | (sb[1] << 31));
  o[1] = make_uint4((u32)d[2], (u32)(d[2] >> 32) | (sb[2] << 31), (u32)d[3], (u32)(d[3] >> 32) | (sb[3] << 31));
}
// the images psi^j(P), j = 0..3, of resident G2 bases, interleaved (four 256-byte records per point):
// psi(x, y) = (conj(x) cx, conj(y) cy) (g2.rs:847-890), psi^2(x, y) = (x k2, -y) (:891-912), psi^3 = psi o psi^2
__global__ void __launch_bounds__(256) k_bases_endo_g2(const u32* __restrict__ rec, u32* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int AW = Store<Fp2Policy>::AFF_WORDS, EL = Store<Fp2Policy>::EL;
  constexpr PLimbs zero = {{0}}, px1 = {BLS_PSI_X_1}, py0 = {BLS_PSI_Y_0}, py1 = {BLS_PSI_Y_1}, k2 = {BLS_PSI2_X};
  fe2_1 cx, cy;
  cx.c0 = fe1_const(zero); cx.c1 = fe1_const(px1); cy.c0 = fe1_const(py0); cy.c1 = fe1_const(py1);
  const u32* r = rec + i * AW;
  u32* o = out + 4 * i * AW;
  fe2_1 x, y; Store<Fp2Policy>::ld(r, x); Store<Fp2Policy>::ld(r + EL, y);
  const u32 flag = r[2 * EL];
  auto put = [&](int j, const fe2_1& px, const fe2_1& py) {
    u32* e = o + j * AW;
    Store<Fp2Policy>::st(e, px); Store<Fp2Policy>::st(e + EL, py);
    e[2 * EL] = flag;
    for (int t = 2 * EL + 1; t < AW; t++) e[t] = 0;
  };
  auto canon2 = [](const auto& v) { fe2_1 c; c.c0 = canon(v.c0); c.c1 = canon(v.c1); return c; };
  auto psi = [&](const fe2_1& px, const fe2_1& py, fe2_1& ox, fe2_1& oy) {
    ox = canon2(mul(conj(px), cx)); oy = canon2(mul(conj(py), cy));
  };
  put(0, x, y);
  fe2_1 x1, y1; psi(x, y, x1, y1); put(1, x1, y1);
  fe2_1 x2 = canon2(mul_fp(x, fe1_const(k2))), y2 = canon2(neg(y));
  put(2, x2, y2);
  fe2_1 x3, y3; psi(x2, y2, x3, y3); put(3, x3, y3);
}

// merged != 0 (resident window-shifted tables): all windows share ONE bucket set, the window only selects the table
template <int WORDS>
__global__ void __launch_bounds__(SORT_THREADS) k_sort_hist(const u32* __restrict__ scalars, u32* __restrict__ ghist, int n, int c, int nwin,
                                                   int fine_bits, int ncoarse, int merged, u32* __restrict__ status) {
  extern __shared__ u32 lh[];
  const int nc = (merged ? 1 : nwin) * ncoarse;
  for (int i = threadIdx.x; i < nc; i += SORT_THREADS) lh[i] = 0;
  __syncthreads();
  for (int k = 0; k < SORT_TILE / SORT_THREADS; k++) {
    int i = blockIdx.x * SORT_TILE + k * SORT_THREADS + threadIdx.x;
    if (i < n) {
      DigitIter<WORDS> d; d.init(scalars, i, c);
      if (!d.canonical()) atomicOr(status, 1u);
      for (int w = 0; w < nwin; w++) {
        u32 mag, neg; d.next(mag, neg);
        if (mag) atomicAdd(&lh[(merged ? 0 : w) * ncoarse + ((mag - 1) >> fine_bits)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += SORT_THREADS) if (lh[i]) atomicAdd(&ghist[i], lh[i]);
}
// exclusive scan of nc <= 8192 counters; gbase[nc] = total.  Also does the per-call zeroing that would
// otherwise be separate memset launches: the reservation cursors, the item-control words, and the counters
// themselves (ready for the next call; they are zeroed once at allocation).
__global__ void __launch_bounds__(1024) k_sort_scan(u32* __restrict__ ghist, u32* __restrict__ gbase, u32* __restrict__ gcur, int nc,
                                                    u32* __restrict__ ctrl, int nctrl) {
  __shared__ u32 sh[SORT_MAX_COUNTERS];
  __shared__ u32 part[1024];
  const int per = (nc + 1023) / 1024;
  u32 t = 0;
  for (int j = 0; j < per; j++) { int i = threadIdx.x * per + j; u32 v = i < nc ? ghist[i] : 0; if (i < nc) sh[i] = v; t += v; }
  part[threadIdx.x] = t;
  __syncthreads();
  for (int st = 1; st < 1024; st <<= 1) {
    u32 x = (int)threadIdx.x >= st ? part[threadIdx.x - st] : 0;
    __syncthreads();
    part[threadIdx.x] += x;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - t;
  for (int j = 0; j < per; j++) { int i = threadIdx.x * per + j; if (i < nc) { gbase[i] = run; gcur[i] = 0; ghist[i] = 0; run += sh[i]; } }
  if (threadIdx.x == 1023) gbase[nc] = part[1023];
  for (int i = threadIdx.x; i < nctrl; i += 1024) ctrl[i] = 0;
}
template <int WORDS>
__global__ void __launch_bounds__(SORT_THREADS) k_sort_scatter(const u32* __restrict__ scalars, const u32* __restrict__ gbase, u32* __restrict__ gcur,
                                                      u32* __restrict__ coarse_out, int n, int c, int nwin, int fine_bits, int ncoarse,
                                                      int merged, u32 stride) {
  extern __shared__ u32 lh[];          // [nc] counts, then reused as running local ranks; [nc] bases
  const int nc = (merged ? 1 : nwin) * ncoarse;
  u32* lbase = lh + nc;
  for (int i = threadIdx.x; i < nc; i += SORT_THREADS) lh[i] = 0;
  __syncthreads();
  for (int k = 0; k < SORT_TILE / SORT_THREADS; k++) {
    int i = blockIdx.x * SORT_TILE + k * SORT_THREADS + threadIdx.x;
    if (i < n) {
      DigitIter<WORDS> d; d.init(scalars, i, c);
      for (int w = 0; w < nwin; w++) {
        u32 mag, neg; d.next(mag, neg);
        if (mag) atomicAdd(&lh[(merged ? 0 : w) * ncoarse + ((mag - 1) >> fine_bits)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += SORT_THREADS) {
    u32 cnt = lh[i];
    lbase[i] = cnt ? gbase[i] + atomicAdd(&gcur[i], cnt) : 0;
    lh[i] = 0;
  }
  __syncthreads();
  const u32 fmask = (1u << fine_bits) - 1;
  for (int k = 0; k < SORT_TILE / SORT_THREADS; k++) {
    int i = blockIdx.x * SORT_TILE + k * SORT_THREADS + threadIdx.x;
    if (i < n) {
      DigitIter<WORDS> d; d.init(scalars, i, c);
      for (int w = 0; w < nwin; w++) {
        u32 mag, neg; d.next(mag, neg);
        if (mag) {
          int ci = (merged ? 0 : w) * ncoarse + ((mag - 1) >> fine_bits);
          u32 r = atomicAdd(&lh[ci], 1u);
          u32 idx = merged ? (u32)w * stride + (u32)i : (u32)i;          // < 2^24 (checked by the host)
          coarse_out[lbase[ci] + r] = (idx << 8) | (neg << 7) | ((mag - 1) & fmask);
        }
      }
    }
  }
}
// one block per (window, coarse) region
__global__ void __launch_bounds__(256) k_sort_fine(const u32* __restrict__ coarse_in, const u32* __restrict__ gbase, u32* __restrict__ sorted,
                                                   u32* __restrict__ offs, int fine_bits, int nc) {
  __shared__ u32 cnt[128];
  __shared__ u32 base[128];
  const int region = blockIdx.x;
  const u32 beg = gbase[region], end = gbase[region + 1];
  const int nfine = 1 << fine_bits;
  if (threadIdx.x < 128) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (u32 j = beg + threadIdx.x; j < end; j += 256) atomicAdd(&cnt[coarse_in[j] & 127u], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = beg;
    for (int f = 0; f < nfine; f++) { base[f] = run; run += cnt[f]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < nfine) { offs[(size_t)region * nfine + threadIdx.x] = base[threadIdx.x]; cnt[threadIdx.x] = 0; }
  if (region == nc - 1 && threadIdx.x == 0) offs[(size_t)nc * nfine] = end;
  __syncthreads();
  for (u32 j = beg + threadIdx.x; j < end; j += 256) {
    u32 e = coarse_in[j];
    u32 f = e & 127u;
    u32 r = atomicAdd(&cnt[f], 1u);
    sorted[base[f] + r] = (e >> 8) | ((e & 0x80u) << 24);
  }
}

// ---- 2. exclusive scan (three small kernels; <= 2^22 elements) --------------------------------------
__global__ void __launch_bounds__(256) k_scan_block_sums(const u32* __restrict__ in, u32* __restrict__ bsum, int n) {
  __shared__ u32 sh[256];
  int base = blockIdx.x * 1024;
  u32 t = 0;
  for (int j = 0; j < 4; j++) {
    int idx = base + threadIdx.x * 4 + j;
    if (idx < n) t += in[idx];
  }
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(1024) k_scan_top(u32* __restrict__ bsum, int nb) {
  // single block: exclusive scan of up to 4096 block sums
  __shared__ u32 sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 1024) sh[i] = i < nb ? bsum[i] : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (int i = 0; i < nb; i++) { u32 v = sh[i]; sh[i] = run; run += v; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 1024) bsum[i] = sh[i];
}
__global__ void __launch_bounds__(256) k_scan_apply(const u32* __restrict__ in, const u32* __restrict__ bsum,
                                                    u32* __restrict__ out, int n) {
  __shared__ u32 sh[256];
  int base = blockIdx.x * 1024;
  u32 v[4]; u32 t = 0;
  for (int j = 0; j < 4; j++) {
    int idx = base + threadIdx.x * 4 + j;
    v[j] = idx < n ? in[idx] : 0;
    t += v[j];
  }
  sh[threadIdx.x] = t;
  __syncthreads();
  // Hillis-Steele inclusive scan over 256 partials
  for (int s = 1; s < 256; s <<= 1) {
    u32 x = (int)threadIdx.x >= s ? sh[threadIdx.x - s] : 0;
    __syncthreads();
    sh[threadIdx.x] += x;
    __syncthreads();
  }
  u32 run = bsum[blockIdx.x] + sh[threadIdx.x] - t;
  for (int j = 0; j < 4; j++) {
    int idx = base + threadIdx.x * 4 + j;
    if (idx < n) out[idx] = run;
    run += v[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = run;   // total in out[n]
}

// ---- 3. scatter ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_msm_scatter(const u32* __restrict__ ent, const u32* __restrict__ rank, const u32* __restrict__ offs,
                                                     u32* __restrict__ sorted, int n, size_t total) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  u32 e = ent[t];
  if (e == 0xffffffffu) return;
  u32 gb = e & 0x7fffffffu;
  u32 i = (u32)(t % (size_t)n);
  sorted[offs[gb] + rank[t]] = i | (e & 0x80000000u);
}

// ---- 4. work items: buckets cut into chunks of <= ITEM_CAP entries, sorted by length (descending) ----------
// A lane of the accumulation kernel walks ONE item.  Cutting bounds the walk (a single digit value shared
// by many scalars -- or the narrow top window -- would otherwise serialise a whole bucket on one lane) and
// sorting by length makes the 64 lanes of a wavefront finish together.  Buckets with one item write their
// sum straight into the bucket array; the others ("heavy") write partial sums that k_msm_heavy folds.
constexpr int ITEM_BINS = ITEM_CAP_MAX + 1;       // bin = cap - len  (bin 0 = longest)
constexpr int HEAVY_SMALL = 16;                   // heavy buckets with <= this many partials are folded by one lane
struct ItemDesc { u32 start, len, dest; };        // entries [start, start+len) of `sorted`; dest = record index
// ctrl[0] = #partial records handed out, ctrl[1] = #heavy buckets, ctrl[2] = #items, ctrl[3] = #heavy buckets folded by a whole block
// heavy[0 .. ctrl[1]) = (bucket, first partial record, #partials); heavy[nb .. nb + ctrl[3]).x = indices of the block-folded ones
// Each block handles ITEM_BLOCK_BUCKETS buckets: the per-block LDS histograms are flushed with one global atomic per
// non-empty bin, and all blocks hit the same few dozen bins (lengths cluster around the mean load), so fewer, fatter blocks
// mean proportionally fewer contended atomics (256 buckets per block: 55 + 48 us for 2^18 buckets; 2048: see DESIGN.md).
constexpr int ITEM_PER_THREAD = 8;
constexpr int ITEM_BLOCK_BUCKETS = 256 * ITEM_PER_THREAD;
__global__ void __launch_bounds__(256) k_item_count(const u32* __restrict__ offs, u32* __restrict__ bins, u32* __restrict__ ctrl, int nb, u32 cap) {
  __shared__ u32 cnt[ITEM_BINS];
  for (u32 i = threadIdx.x; i <= cap; i += 256) cnt[i] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEM_PER_THREAD; k++) {
    int b = blockIdx.x * ITEM_BLOCK_BUCKETS + k * 256 + threadIdx.x;
    if (b < nb) {
      u32 load = offs[b + 1] - offs[b];
      u32 full = load / cap, rem = load - full * cap;
      if (full) atomicAdd(&cnt[0], full);
      if (rem || !full) atomicAdd(&cnt[cap - rem], 1u);
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i <= cap; i += 256) if (cnt[i]) atomicAdd(&bins[i], cnt[i]);
}
__global__ void __launch_bounds__(256) k_item_scan(u32* __restrict__ bins, u32* __restrict__ ctrl, u32 cap) {
  __shared__ u32 sh[ITEM_BINS];
  for (u32 i = threadIdx.x; i <= cap; i += 256) sh[i] = bins[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (u32 i = 0; i <= cap; i++) { u32 v = sh[i]; sh[i] = run; run += v; }
    ctrl[2] = run;
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i <= cap; i += 256) bins[i] = sh[i];      // exclusive bin bases
}
__global__ void __launch_bounds__(256) k_item_fill(const u32* __restrict__ offs, const u32* __restrict__ bins, u32* __restrict__ bcur,
                                                   u32* __restrict__ ctrl, ItemDesc* __restrict__ items, uint4* __restrict__ heavy, int nb, u32 cap) {
  __shared__ u32 cnt[ITEM_BINS];
  __shared__ u32 base[ITEM_BINS];
  const u32 ITEM_CAP = cap;
  for (u32 i = threadIdx.x; i <= cap; i += 256) cnt[i] = 0;
  __syncthreads();
  u32 beg[ITEM_PER_THREAD], full[ITEM_PER_THREAD], rem[ITEM_PER_THREAD], r_full[ITEM_PER_THREAD], r_rem[ITEM_PER_THREAD];
#pragma unroll
  for (int k = 0; k < ITEM_PER_THREAD; k++) {
    int b = blockIdx.x * ITEM_BLOCK_BUCKETS + k * 256 + threadIdx.x;
    beg[k] = 0; full[k] = 0; rem[k] = 0; r_full[k] = 0; r_rem[k] = 0;
    if (b < nb) {
      beg[k] = offs[b]; u32 load = offs[b + 1] - beg[k];
      full[k] = load / ITEM_CAP; rem[k] = load - full[k] * ITEM_CAP;
      if (full[k]) r_full[k] = atomicAdd(&cnt[0], full[k]);
      if (rem[k] || !full[k]) r_rem[k] = atomicAdd(&cnt[ITEM_CAP - rem[k]], 1u);
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i <= cap; i += 256) base[i] = cnt[i] ? bins[i] + atomicAdd(&bcur[i], cnt[i]) : 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEM_PER_THREAD; k++) {
    int b = blockIdx.x * ITEM_BLOCK_BUCKETS + k * 256 + threadIdx.x;
    if (b < nb) {
      const bool has_rem = rem[k] || !full[k];
      u32 nitems = full[k] + (has_rem ? 1u : 0u);
      u32 dest0 = (u32)b;
      if (nitems > 1) {
        dest0 = (u32)nb + atomicAdd(&ctrl[0], nitems);
        u32 h = atomicAdd(&ctrl[1], 1u);
        heavy[h] = make_uint4((u32)b, dest0, nitems, 0);
        if (nitems > (u32)HEAVY_SMALL) heavy[(u32)nb + atomicAdd(&ctrl[3], 1u)].x = h;      // the few block-folded buckets, listed apart
      }
      for (u32 j = 0; j < full[k]; j++) {
        ItemDesc d; d.start = beg[k] + j * ITEM_CAP; d.len = ITEM_CAP; d.dest = dest0 + j;
        items[base[0] + r_full[k] + j] = d;
      }
      if (has_rem) {
        ItemDesc d; d.start = beg[k] + full[k] * ITEM_CAP; d.len = rem[k]; d.dest = dest0 + full[k];
        items[base[ITEM_CAP - rem[k]] + r_rem[k]] = d;
      }
    }
  }
}

// ---- 5. bucket accumulation ---------------------------------------------------------------------------
// Point index e (31 bits): records [0, nsplit) live in `bases`, records [nsplit, ...) in `bases2` (the images under the GLV
// endomorphism; nsplit = 0xffffffff when there is no second array).
#ifndef BLS_ACC_BLOCK
#define BLS_ACC_BLOCK 256
#endif
// wave priority of the accumulation kernels: above the sort / item kernels of the neighbouring calls (priority 0), below the tails
// (3).  A/B on one box, default bench: 0 -> 3.67-3.69, 1 -> 3.73, 2 -> 3.71, 3 -> 3.55 *10^8 scalar-muls/s; 64- or 128-lane blocks make
// the launch itself faster (2.59-2.67 vs 2.73 ms) and the pipeline slower (3.53-3.60), 512-lane blocks lose 20 %
#ifndef BLS_ACC_PRIO
#define BLS_ACC_PRIO 1
#endif
// diagnostic build only (-DBLS_ACC_TRACE, tools/acc_trace.py): every wavefront of the accumulation records its start and end on the
// constant-rate clock, its hardware id and its item length, so that the occupancy of the chip over the launch can be drawn
#ifdef BLS_ACC_TRACE
__device__ unsigned long long* g_acc_trace = nullptr;
__device__ unsigned int g_acc_seq = 0;             // != 0: sequence mode (records appended across launches, BLS_ACC_TRACE_SEQ_CAP of them)
constexpr unsigned int BLS_ACC_TRACE_SEQ_CAP = 1u << 18;
#endif
template <class F>
__global__ void __launch_bounds__(BLS_ACC_BLOCK) k_msm_accumulate(const u32* __restrict__ bases, const u32* __restrict__ bases2, u32 nsplit,
                                                        const u32* __restrict__ sorted,
                                                        const ItemDesc* __restrict__ items, const u32* __restrict__ ctrl,
                                                        u32* __restrict__ records) {
  if (BLS_ACC_PRIO) __builtin_amdgcn_s_setprio(BLS_ACC_PRIO);
#ifdef BLS_ACC_TRACE
  unsigned long long* const trace = g_acc_trace;
  const unsigned long long trace_t0 = wall_clock64(), trace_c0 = clock64();
#endif
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ctrl[2]) return;
  ItemDesc d = items[t];
  auto rec_of = [&](u32 e) -> const u32* {
    u32 idx = e & 0x7fffffffu;
    return idx < nsplit ? bases + (size_t)idx * Store<F>::AFF_WORDS : bases2 + (size_t)(idx - nsplit) * Store<F>::AFF_WORDS;
  };
  Xyzz<F> acc;
  acc.x = F::zero(); acc.y = F::zero(); acc.zz = F::zero(); acc.zzz = F::zero();
  bool acc_inf = true;
  // software pipeline: while addition j runs, the record of entry j+1 and the index of entry j+2 are in flight
  // (two dependent loads per entry; with two wavefronts per SIMD nothing else hides their latency).  The prefetches are
  // UNCONDITIONAL (indices clamped to the item's last entry, whose record is then fetched once more and dropped): behind a
  // branch the backend cannot count the loads in flight and waits for all of them where the current record is first used --
  // i.e. right after issuing them (round 6: that wait stood in the loop since round 2).  The identity flag stays a word for
  // the same reason: a `bool` is a comparison, a use of the load.
  const u32 end = d.start + d.len;
  if (d.len) {
    const u32 last = end - 1;
    u32 e = sorted[d.start];
    u32 e_next = sorted[d.start + 1 < end ? d.start + 1 : last];
    Aff<F> q; u32 inf;
    load_aff_word<F>(rec_of(e), q, inf);
    for (u32 j = d.start; j < end; j++) {
#ifdef BLS_ACC_TRACE
      if (trace && !g_acc_seq && (threadIdx.x & 63) == 0 && ((t >> 6) & 255) == 0 && j - d.start < 250) trace[6 * 16384 + (size_t)(t >> 14) * 256 + (j - d.start)] = wall_clock64();
#endif
      Aff<F> qn; u32 infn;
      load_aff_word<F>(rec_of(e_next), qn, infn);
      const u32 e_next2 = sorted[j + 2 < end ? j + 2 : last];
      if (inf == 0) {                                       // identity base: contributes nothing
        auto qy = cond_neg(q.y, (e >> 31) != 0);
        // G1: the ten field products are inlined (one ~35 KB straight-line body; measured 8% faster than calls even
        // at 2 waves/SIMD).  G2 keeps the out-of-line Fp2 products (its body would not fit the instruction cache).
        if constexpr (std::is_same<F, FpPolicy>::value) acc = xyzz_add_mixed_inl(acc, acc_inf, q.x, qy);
        else acc = xyzz_add_mixed<F>(acc, acc_inf, q.x, qy);
      }
      q = qn; inf = infn; e = e_next; e_next = e_next2;
    }
  }
  store_proj<F>(records + (size_t)d.dest * Store<F>::PROJ_WORDS, xyzz_to_proj<F>(acc, acc_inf));
#ifdef BLS_ACC_TRACE
  if (trace && (threadIdx.x & 63) == 0) {
    size_t slot = (size_t)(t >> 6);
    if (g_acc_seq) { unsigned int q = atomicAdd(&g_acc_seq, 1u); if (q >= BLS_ACC_TRACE_SEQ_CAP) return; slot = q; }
    unsigned long long* o = trace + 6 * slot;
    o[4] = trace_c0; o[5] = clock64();
    o[0] = trace_t0; o[1] = wall_clock64(); o[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32); o[3] = d.len | ((unsigned long long)((size_t)ctrl >> 8) << 32);
  }
#endif
}
// G2 accumulation with every Fp2 value spread over a lane pair (pairlane.hip.h): lane 2k works on the c0 coefficients
// and lane 2k+1 on the c1 coefficients of chain k.  Same items, same records, same formula.
#ifndef BLS_G2ACC_BLOCK
#define BLS_G2ACC_BLOCK 256
#endif
__global__ void __launch_bounds__(BLS_G2ACC_BLOCK, 2) k_msm_accumulate_g2pair(const u32* __restrict__ bases, const u32* __restrict__ sorted,
                                                               const ItemDesc* __restrict__ items, const u32* __restrict__ ctrl,
                                                               u32* __restrict__ records) {
  typedef Fp2PairPolicy F;
  constexpr int AW = Store<Fp2Policy>::AFF_WORDS, PW = Store<Fp2Policy>::PROJ_WORDS;
  if (BLS_ACC_PRIO) __builtin_amdgcn_s_setprio(BLS_ACC_PRIO);
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (t >= ctrl[2]) return;                     // both lanes of a pair leave together
  const u32 par = threadIdx.x & 1;
  ItemDesc d = items[t];
  // this lane's halves of a record: x coefficient `par`, y coefficient `par`, and the identity flag
  auto load_half = [&](u32 e, FeP<1, 1>& qx, FeP<1, 1>& qy, u32& flag) {
    const u32* rec = bases + (size_t)(e & 0x7fffffffu) * AW;
    const uint2* px = reinterpret_cast<const uint2*>(rec + par * NL);
    const uint2* py = reinterpret_cast<const uint2*>(rec + 2 * NL + par * NL);
#pragma unroll
    for (int i = 0; i < NL / 2; i++) {
      uint2 a = px[i], b = py[i];
      qx.v.l[2 * i] = a.x; qx.v.l[2 * i + 1] = a.y; qy.v.l[2 * i] = b.x; qy.v.l[2 * i + 1] = b.y;
    }
    flag = rec[4 * NL];
  };
  Xyzz<F> acc;
  acc.x = F::zero(); acc.y = F::zero(); acc.zz = F::zero(); acc.zzz = F::zero();
  bool acc_inf = true, slow = false;
  // software pipeline as in the G1 kernel: while addition j runs, the record of entry j+1 and the index of entry j+2 are in
  // flight (with the four images of every base resident the gathers range over 1 GB -- past the Infinity Cache)
  const u32 end = d.start + d.len;
  u32 j = d.start;
  u32 e = d.len ? sorted[d.start] : 0;
  u32 e_next = d.len > 1 ? sorted[d.start + 1] : 0;
  FeP<1, 1> qxn, qyn; u32 flagn = 1;
  if (d.len) load_half(e, qxn, qyn, flagn);
  for (; j < end; j++) {
    const FeP<1, 1> qx = qxn, qy1 = qyn;
    const u32 flag = flagn, e_cur = e;
    u32 e_next2 = 0;
    if (j + 1 < end) load_half(e_next, qxn, qyn, flagn);
    if (j + 2 < end) e_next2 = sorted[j + 2];
    e = e_next; e_next = e_next2;
    if (flag != 0) continue;                    // identity base (same decision in both lanes)
    FeP<2, 2> qy = select((e_cur >> 31) != 0, neg(qy1), (FeP<2, 2>)qy1);
    if (acc_inf) { acc_inf = false; acc = xyzz_from_affine<F>(qx, qy); continue; }
    // madd-2008-s, generic case only; when P = U2 - X MAY be zero (one-limb filter on both coefficients, decided identically
    // in both lanes) the pair leaves the loop and finishes the chain with the complete mixed addition
    auto U2 = mul(qx, acc.zz);
    auto S2 = mul(qy, acc.zzz);
    auto P = norm(sub(U2, acc.x));
    auto R = norm(sub(S2, acc.y));
    {
      bool m = maybe_zero(P.v);
      bool pm = partner_flag(m);
      if (m && pm) { slow = true; e = e_cur; break; }          // entry j is NOT consumed
    }
    auto PP = sqr(P);
    auto PPP = mul(P, PP);
    auto Q = mul(acc.x, PP);
    auto X3 = norm(sub(sqr(R), add(PPP, dbl(Q))));
    auto Y3 = sub(mul(R, norm(sub(Q, X3))), mul(acc.y, PPP));
    auto ZZ3 = mul(acc.zz, PP);
    auto ZZZ3 = mul(acc.zzz, PPP);
    acc.x = F::st(X3); acc.y = F::st(Y3); acc.zz = F::st(ZZ3); acc.zzz = F::st(ZZZ3);
  }
  Proj<F> pr = xyzz_to_proj<F>(acc, acc_inf);
  if (slow) {
    for (; j < end; j++) {
      u32 e2 = sorted[j];
      FeP<1, 1> qx, qy1; u32 flag;
      load_half(e2, qx, qy1, flag);
      if (flag != 0) continue;
      FeP<2, 2> qy = select((e2 >> 31) != 0, neg(qy1), (FeP<2, 2>)qy1);
      pr = pt_add_mixed_y<F>(pr, qx, qy);
    }
  }
  u32* o = records + (size_t)d.dest * PW + par * NL;
#pragma unroll
  for (int i = 0; i < NL; i++) { o[i] = pr.x.v.l[i]; o[2 * NL + i] = pr.y.v.l[i]; o[4 * NL + i] = pr.z.v.l[i]; }
}

// fold the partial sums of the heavy buckets: one lane per bucket when it has few partials (blocks [0, small_blocks): the host sizes them
// to one lane per bucket of the call -- with an item cap below the mean load MOST buckets are cut in two or three) ...
constexpr int HEAVY_BIG_BLOCKS = 512;
template <class F>
DEV void msm_heavy_small(const uint4* __restrict__ heavy, const u32* __restrict__ ctrl, u32* __restrict__ records, u32 small_blocks) {
  constexpr int PW = Store<F>::PROJ_WORDS;
  u32 nh = ctrl[1];
  for (u32 h = blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += small_blocks * blockDim.x) {
    uint4 d = heavy[h];
    if (d.z > HEAVY_SMALL) continue;
    const u32* part = records + (size_t)d.y * PW;
    Proj<F> acc; load_proj<F>(part, acc);
    for (u32 k = 1; k < d.z; k++) { Proj<F> e; load_proj<F>(part + (size_t)k * PW, e); acc = pt_add<F>(acc, e); }
    store_proj<F>(records + (size_t)d.x * PW, acc);
  }
}
// ... and one block per bucket (in-place tree, fan 8) when it has many.  These buckets are listed apart (heavy[nb + i].x): walking the whole
// heavy list for them cost 0.3 ms when every bucket of a 2^18-bucket call was on it (512 dependent loads per block).
template <class F>
DEV void msm_heavy_big(const uint4* __restrict__ heavy, const u32* __restrict__ ctrl, u32* __restrict__ records, u32 nb, u32 small_blocks) {
  constexpr int PW = Store<F>::PROJ_WORDS;
  const u32 nbig = ctrl[3];
  for (u32 i = blockIdx.x - small_blocks; i < nbig; i += gridDim.x - small_blocks) {
    uint4 d = heavy[heavy[nb + i].x];
    u32* part = records + (size_t)d.y * PW;
    u32 n = d.z;
    for (u32 stride = 1; stride < n; stride *= 8) {
      u32 groups = (n + stride * 8 - 1) / (stride * 8);
      for (u32 g = threadIdx.x; g < groups; g += blockDim.x) {
        size_t i0 = (size_t)g * stride * 8;
        Proj<F> acc; load_proj<F>(part + i0 * PW, acc);
        for (int k = 1; k < 8; k++) {
          size_t j = i0 + (size_t)k * stride;
          if (j < n) { Proj<F> e; load_proj<F>(part + j * PW, e); acc = pt_add<F>(acc, e); }
        }
        store_proj<F>(part + i0 * PW, acc);
      }
      __threadfence();      // partials written by other waves of this block must be visible (L1 is not coherent)
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      Proj<F> acc; load_proj<F>(part, acc);
      store_proj<F>(records + (size_t)d.x * PW, acc);
    }
    __syncthreads();
  }
}

template <class F>
__global__ void __launch_bounds__(256) k_msm_heavy(const uint4* __restrict__ heavy, const u32* __restrict__ ctrl, u32* __restrict__ records, u32 nb, u32 small_blocks) {
  if (ctrl[1] == 0) return;                                  // no bucket was cut
  if (blockIdx.x < small_blocks) msm_heavy_small<F>(heavy, ctrl, records, small_blocks);
  else msm_heavy_big<F>(heavy, ctrl, records, nb, small_blocks);
}

// ---- 6. weighted bucket reduction -----------------------------------------------------------------------
// One level:  elements E[seg][0..n) (PROJ records), weight(j) = j + off.  Chain (seg, g) handles chunk
// j in [g*M, (g+1)*M):  R = sum E_j,  T = sum (j - g*M + off) * E_j   (running sums, high index first).
// Then  wsum(seg) = sum_g T_g + M * sum_g g * R_g.
// G1 bottom level with the two running sums of a chain on TWO lanes: lane 2t keeps R (run += E_i), lane 2t+1 keeps T
// (tot += run), one step behind -- the value of `run` travels to the partner lane by DPP.  M + 1 dependent additions per chain
// instead of 2 M (the level is latency-bound: 2^15 chains of 16 complete additions on a chip with 2^16 wavefront slots' worth
// of lanes).  Same group elements; the projective representatives differ from the one-lane form only by additions of the identity.
// the bottom level is the one bulky kernel of the tail (2^16 lanes): at the tails' wave priority (3) it pushes the next call's
// accumulation aside; at the accumulation's own priority the pipelined rate is 1.3 % higher (3.70-3.72 vs 3.64-3.66 *10^8, three
// alternating runs on one box) and a single call is unchanged
#ifndef BLS_WSUM_PRIO
#define BLS_WSUM_PRIO 1
#endif
#ifndef BLS_WSUM_BLOCK
#define BLS_WSUM_BLOCK 256
#endif
__global__ void __launch_bounds__(BLS_WSUM_BLOCK) k_wsum_level_pair(const u32* __restrict__ E, u32* __restrict__ Rout, u32* __restrict__ Tout,
                                                         int nseg, int n, int M, int off) {
  typedef FpPolicy F;
  constexpr int PW = Store<F>::PROJ_WORDS;
  __builtin_amdgcn_s_setprio(BLS_WSUM_PRIO);
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int t = gid >> 1;
  const bool isT = (gid & 1) != 0;
  int G = n / M;
  if (t >= nseg * G) return;                       // both lanes of a pair leave together
  int seg = t / G, g = t - seg * G;
  const u32* base = E + ((size_t)seg * n + (size_t)g * M) * PW;
  Proj<F> acc = pt_identity<F>();
  for (int s = 0; s <= M; s++) {
    Proj<F> prev;                                  // the partner's sum after step s - 1 (both lanes execute the exchange)
    prev.x = partner(acc.x); prev.y = partner(acc.y); prev.z = partner(acc.z);
    Proj<F> e = pt_identity<F>();
    if (!isT) { if (s < M) load_proj<F>(base + (size_t)(M - 1 - s) * PW, e); }
    else if (s >= 1 && ((M - s) > 0 || off)) e = prev;
    acc = pt_add<F>(acc, e);
  }
  store_proj<F>((isT ? Tout : Rout) + (size_t)t * PW, acc);
}
// G2 bottom level over lane pairs (pairlane.hip.h): chain t on lanes 2t / 2t+1, c0 / c1 coefficients; same records, same sums.
// Half the registers of the one-lane form, and twice the lanes: the 2^16 chains of a 2^20-point MSM fill two wavefronts per SIMD.
DEV void load_proj_pair(const u32* rec, u32 par, Proj<Fp2PairPolicy>& p) {
#pragma unroll
  for (int i = 0; i < NL; i++) { p.x.v.l[i] = rec[par * NL + i]; p.y.v.l[i] = rec[2 * NL + par * NL + i]; p.z.v.l[i] = rec[4 * NL + par * NL + i]; }
}
DEV void store_proj_pair(u32* rec, u32 par, const Proj<Fp2PairPolicy>& p) {
#pragma unroll
  for (int i = 0; i < NL; i++) { rec[par * NL + i] = p.x.v.l[i]; rec[2 * NL + par * NL + i] = p.y.v.l[i]; rec[4 * NL + par * NL + i] = p.z.v.l[i]; }
}
__global__ void __launch_bounds__(256, 2) k_wsum_level_g2pair(const u32* __restrict__ E, u32* __restrict__ Rout, u32* __restrict__ Tout,
                                                              int nseg, int n, int M, int off) {
  typedef Fp2PairPolicy F;
  constexpr int PW = Store<Fp2Policy>::PROJ_WORDS;
  __builtin_amdgcn_s_setprio(3);
  int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  const u32 par = threadIdx.x & 1;
  int G = n / M;
  if (t >= nseg * G) return;
  int seg = t / G, g = t - seg * G;
  const u32* base = E + ((size_t)seg * n + (size_t)g * M) * PW;
  Proj<F> run = pt_identity<F>(), tot = pt_identity<F>();
  for (int i = M - 1; i >= 0; i--) {
    Proj<F> e; load_proj_pair(base + (size_t)i * PW, par, e);
    run = pt_add<F>(run, e);
    if (i > 0 || off) tot = pt_add<F>(tot, run);
  }
  store_proj_pair(Rout + (size_t)t * PW, par, run);
  store_proj_pair(Tout + (size_t)t * PW, par, tot);
}
// tree sum: out[seg][g] = sum of M consecutive records
template <class F>
__global__ void __launch_bounds__(256) k_tree_sum(const u32* __restrict__ E, u32* __restrict__ out, int nseg, int n, int M) {
  __builtin_amdgcn_s_setprio(3);      // latency-bound tail: win VALU arbitration against co-resident bulk waves
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int G = (n + M - 1) / M;
  if (t >= nseg * G) return;
  int seg = t / G, g = t - seg * G;
  Proj<F> acc = pt_identity<F>();
  for (int i = g * M; i < (g + 1) * M && i < n; i++) {
    Proj<F> e; load_proj<F>(E + ((size_t)seg * n + i) * Store<F>::PROJ_WORDS, e);
    acc = pt_add<F>(acc, e);
  }
  store_proj<F>(out + (size_t)t * Store<F>::PROJ_WORDS, acc);
}
// acc[seg] = 2^k * x[seg] + y[seg] (k doublings) exists only as a team kernel (k_shift_add_team) below.

// ---- 6'. team (8 lanes per chain) versions of the reduction kernels, used when a level has too few chains
// to fill the chip with one lane each.  All control flow around the team operations is block-uniform.
template <class F>
__global__ void __launch_bounds__(256) k_wsum_level_team(const u32* __restrict__ E, u32* __restrict__ Rout, u32* __restrict__ Tout,
                                                         int nseg, int n, int M, int off) {
  extern __shared__ u32 team_lds[];
  __builtin_amdgcn_s_setprio(3);
  const int tl = threadIdx.x & (TEAM - 1);
  u32* mbox = team_lds + (threadIdx.x / TEAM) * TEAM_SLOTS * TeamTraits<F>::WORDS;
  int G = n / M, total = nseg * G;
  int t = (blockIdx.x * blockDim.x + threadIdx.x) / TEAM;
  bool live = t < total;
  if (!live) t = total - 1;
  int seg = t / G, g = t - seg * G;
  const u32* base = E + ((size_t)seg * n + (size_t)g * M) * Store<F>::PROJ_WORDS;
  Proj<F> run = pt_identity<F>(), tot = pt_identity<F>();
  for (int i = M - 1; i >= 0; i--) {
    Proj<F> e; load_proj<F>(base + (size_t)i * Store<F>::PROJ_WORDS, e);
    run = pt_add_team<F>(run, e, mbox, tl);
    if (i > 0 || off) tot = pt_add_team<F>(tot, run, mbox, tl);
  }
  if (live && tl == 0) {
    store_proj<F>(Rout + (size_t)t * Store<F>::PROJ_WORDS, run);
    store_proj<F>(Tout + (size_t)t * Store<F>::PROJ_WORDS, tot);
  }
}
// the same split for the team form: teams 2t (R) and 2t+1 (T) of a block share a chain; R's running sum is handed over
// through LDS after every step.  M + 1 team additions per chain instead of 2 M.
template <class F>
__global__ void __launch_bounds__(256) k_wsum_level_team2(const u32* __restrict__ E, u32* __restrict__ Rout, u32* __restrict__ Tout,
                                                          int nseg, int n, int M, int off) {
  extern __shared__ u32 team_lds[];
  __builtin_amdgcn_s_setprio(3);
  constexpr int PW = Store<F>::PROJ_WORDS, W = TeamTraits<F>::WORDS;
  const int tl = threadIdx.x & (TEAM - 1);
  const int team = threadIdx.x / TEAM;
  const bool isT = (team & 1) != 0;
  u32* mbox = team_lds + team * TEAM_SLOTS * W;
  u32* xch = team_lds + (blockDim.x / TEAM) * TEAM_SLOTS * W + (team >> 1) * 3 * W;       // R's sum, written by team 2t, read by team 2t+1
  int G = n / M, total = nseg * G;
  int t = (blockIdx.x * blockDim.x + threadIdx.x) / (2 * TEAM);
  bool live = t < total;
  if (!live) t = total - 1;
  int seg = t / G, g = t - seg * G;
  const u32* base = E + ((size_t)seg * n + (size_t)g * M) * PW;
  Proj<F> acc = pt_identity<F>();
  for (int s = 0; s <= M; s++) {
    Proj<F> e = pt_identity<F>();
    if (!isT) { if (s < M) load_proj<F>(base + (size_t)(M - 1 - s) * PW, e); }
    else if (s >= 1 && ((M - s) > 0 || off)) { TeamTraits<F>::get(xch, e.x); TeamTraits<F>::get(xch + W, e.y); TeamTraits<F>::get(xch + 2 * W, e.z); }
    acc = pt_add_team<F>(acc, e, mbox, tl);        // (block barriers inside: every read of xch above precedes the write below)
    if (!isT && tl == 0) { TeamTraits<F>::put(xch, acc.x); TeamTraits<F>::put(xch + W, acc.y); TeamTraits<F>::put(xch + 2 * W, acc.z); }
    __syncthreads();
  }
  if (live && tl == 0) store_proj<F>((isT ? Tout : Rout) + (size_t)t * PW, acc);
}
template <class F>
__global__ void __launch_bounds__(256) k_tree_sum_team(const u32* __restrict__ E, u32* __restrict__ out, int nseg, int n, int M) {
  extern __shared__ u32 team_lds[];
  __builtin_amdgcn_s_setprio(3);
  const int tl = threadIdx.x & (TEAM - 1);
  u32* mbox = team_lds + (threadIdx.x / TEAM) * TEAM_SLOTS * TeamTraits<F>::WORDS;
  int G = (n + M - 1) / M, total = nseg * G;
  int t = (blockIdx.x * blockDim.x + threadIdx.x) / TEAM;
  bool live = t < total;
  if (!live) t = total - 1;
  int seg = t / G, g = t - seg * G;
  Proj<F> acc = pt_identity<F>();
  for (int k = 0; k < M; k++) {
    int i = g * M + k;
    Proj<F> e = pt_identity<F>();
    if (i < n) load_proj<F>(E + ((size_t)seg * n + i) * Store<F>::PROJ_WORDS, e);
    acc = pt_add_team<F>(acc, e, mbox, tl);
  }
  if (live && tl == 0) store_proj<F>(out + (size_t)t * Store<F>::PROJ_WORDS, acc);
}
// Several independent tree-sum passes in ONE launch (the T sums of different reduction levels: level l's records become
// available when level l has run, and every level needs log8(G_l) passes -- one launch per step carries the next pass of
// every level that still has one, so the whole T tree finishes while the R chain is still running).
constexpr int TREE_JOBS_MAX = 8;
struct TreeJobs {
  const u32* in[TREE_JOBS_MAX]; u32* out[TREE_JOBS_MAX];
  int n[TREE_JOBS_MAX], M[TREE_JOBS_MAX], G[TREE_JOBS_MAX];
  int first_team[TREE_JOBS_MAX + 1];          // teams of job j: [first_team[j], first_team[j + 1])
  int njobs, nseg;
  int maxM;                                   // largest fan-in of the launch: every team runs this many (barrier-carrying) steps
};
template <class F>
__global__ void __launch_bounds__(256) k_tree_sum_team_multi(TreeJobs J) {
  extern __shared__ u32 team_lds[];
  __builtin_amdgcn_s_setprio(3);
  const int tl = threadIdx.x & (TEAM - 1);
  u32* mbox = team_lds + (threadIdx.x / TEAM) * TEAM_SLOTS * TeamTraits<F>::WORDS;
  const int total = J.first_team[J.njobs];
  int t = (blockIdx.x * blockDim.x + threadIdx.x) / TEAM;
  bool live = t < total;
  if (!live) t = total - 1;
  int j = 0;
#pragma unroll
  for (int k = 1; k < TREE_JOBS_MAX; k++) if (k < J.njobs && t >= J.first_team[k]) j = k;
  const int local = t - J.first_team[j];
  const int G = J.G[j], n = J.n[j], M = J.M[j];
  const int seg = local / G, g = local - seg * G;
  const u32* E = J.in[j];
  Proj<F> acc = pt_identity<F>();
  // pt_add_team carries block-wide barriers and teams of different jobs share a block: the trip count is the launch-wide
  // maximum, jobs with a smaller fan-in add identities for the surplus steps (the complete formulas take them like any point)
  for (int k = 0; k < J.maxM; k++) {
    int i = g * M + k;
    Proj<F> e = pt_identity<F>();
    if (k < M && i < n) load_proj<F>(E + ((size_t)seg * n + i) * Store<F>::PROJ_WORDS, e);
    acc = pt_add_team<F>(acc, e, mbox, tl);
  }
  if (live && tl == 0) store_proj<F>(J.out[j] + (size_t)local * Store<F>::PROJ_WORDS, acc);
}
template <class F>
__global__ void __launch_bounds__(256) k_shift_add_team(const u32* __restrict__ x, const u32* __restrict__ y, u32* __restrict__ out, int nseg, int k) {
  extern __shared__ u32 team_lds[];
  __builtin_amdgcn_s_setprio(3);
  const int tl = threadIdx.x & (TEAM - 1);
  u32* mbox = team_lds + (threadIdx.x / TEAM) * TEAM_SLOTS * TeamTraits<F>::WORDS;
  int seg = (blockIdx.x * blockDim.x + threadIdx.x) / TEAM;
  bool live = seg < nseg;
  if (!live) seg = nseg - 1;
  Proj<F> a, b;
  load_proj<F>(x + (size_t)seg * Store<F>::PROJ_WORDS, a);
  load_proj<F>(y + (size_t)seg * Store<F>::PROJ_WORDS, b);
  for (int i = 0; i < k; i++) a = pt_double_team<F>(a, mbox, tl);
  a = pt_add_team<F>(a, b, mbox, tl);
  if (live && tl == 0) store_proj<F>(out + (size_t)seg * Store<F>::PROJ_WORDS, a);
}
// ---- 7. window combine: Horner over the windows (c doublings + 1 addition each) by ONE team
// (launch with a single block of TEAM threads)
template <class F>
__global__ void __launch_bounds__(64) k_msm_combine_team(const u32* __restrict__ wsums, u32* __restrict__ out, int nwin, int c) {
  extern __shared__ u32 team_lds[];
  __builtin_amdgcn_s_setprio(3);
  const int tl = threadIdx.x & (TEAM - 1);
  u32* mbox = team_lds;
  Proj<F> acc;
  load_proj<F>(wsums + (size_t)(nwin - 1) * Store<F>::PROJ_WORDS, acc);
  for (int w = nwin - 2; w >= 0; w--) {
    for (int i = 0; i < c; i++) acc = pt_double_team<F>(acc, mbox, tl);
    Proj<F> s; load_proj<F>(wsums + (size_t)w * Store<F>::PROJ_WORDS, s);
    acc = pt_add_team<F>(acc, s, mbox, tl);
  }
  if (tl == 0) store_proj<F>(out, acc);
}
// sum of n records by one team (cross-rank fold)
template <class F>
__global__ void __launch_bounds__(64) k_proj_sum_team(const u32* __restrict__ rec, u32* __restrict__ out, size_t n) {
  extern __shared__ u32 team_lds[];
  const int tl = threadIdx.x & (TEAM - 1);
  Proj<F> acc = pt_identity<F>();
  for (size_t i = 0; i < n; i++) { Proj<F> p; load_proj<F>(rec + i * Store<F>::PROJ_WORDS, p); acc = pt_add_team<F>(acc, p, team_lds, tl); }
  if (tl == 0) store_proj<F>(out, acc);
}

}  // namespace bls
