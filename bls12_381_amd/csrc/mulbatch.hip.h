// mulbatch.hip.h -- batched variable-base scalar multiplication: N (point, scalar) pairs in, N points out.
//
// Reference: the crate's unit operation `&G1Affine * &Scalar` / `&G1Projective * &Scalar` (src/g1.rs:556-594 -> `multiply`
// :754-774: 255 x (double, add, select) over `Scalar::to_bytes()`), the same for G2 (src/g2.rs:609-647, 825-845); the
// criterion points "G1 scalar multiplication" etc. of benches/groups.rs:44,89,113,158.  The reference's result is a group
// element; its projective representative depends on the addition chain, so results are compared after affine conversion
// (SURVEY.md "five facts" 2) -- as for the MSM.
//
// One scalar multiplication per lane (G1) or per lane pair (G2, pairlane.hip.h), signed 4-bit fixed windows:
//   table [1..8] P built with 4 doublings + 3 additions, kept in per-lane scratch (8 x 42 words; the run-time index makes
//   it a gather from scratch, ~1 % of the time), then 64 windows of 4 doublings + ONE complete addition each (a zero digit
//   adds the identity, which the complete formulas of curve.hip.h take like any other point: no divergence).
// 256 doublings x 8 + 67 additions x 12 = 2 852 field multiplications against the reference's 255 x (8 + 12) = 5 100.
// The complete RCB formulas make the result exact for EVERY curve point (identity, points outside the subgroup that the
// unchecked decoders hand out, scalars 0 and r - 1): no endomorphism is used here, so there is no subgroup precondition.
#pragma once
#include "msm.hip.h"
#include "h2c.hip.h"            // pt_psi / pt_psi2 (g2.rs:847-912) for the G2 fast path

namespace bls {

template <class F> struct MbIO;
template <> struct MbIO<FpPolicy> {
  static constexpr int LANES = 1, WW = 12;
  static DEV FpPolicy::elem load(const u32* w) { return FpPolicy::st(fe_from_ref(w)); }
  template <class T> static DEV void save(const T& a, u32* w) { fe_to_ref(a, w); }
};
template <> struct MbIO<Fp2PairPolicy> {
  static constexpr int LANES = 2, WW = 24;
  static DEV Fp2PairPolicy::elem load(const u32* w) { Fp2PairPolicy::elem r; r.v = (Fe<1, VS2>)fe_from_ref(w + (lane_is_c1() ? 12 : 0)); return r; }
  template <class T> static DEV void save(const T& a, u32* w) { fe_to_ref(a.v, w + (lane_is_c1() ? 12 : 0)); }
};

// out[i] = [scalars[i]] (xy[i], inf[i])   as a projective point in wire limbs (X | Y | Z)
template <class F>
__global__ void __launch_bounds__(256, 2)
k_mul_batch(const u32* __restrict__ xy, const uint8_t* __restrict__ inf, const u32* __restrict__ scalars, u32* __restrict__ out, size_t n,
            u32* __restrict__ status, int form) {
  constexpr int LANES = MbIO<F>::LANES, WW = MbIO<F>::WW;
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
  if (i >= n) return;
  u32 s[8];
  if (!scalar_load(scalars, i, form, s)) atomicOr(status, 1u);          // no `Scalar` has such bytes (scalar.rs:256-280): reported, result unspecified
  // signed digits d_w in [-8, 8], k = sum d_w 16^w; k < 2^255 leaves the top window at most 7 + carry: no 65th window
  u32 mag[8], sgn[2] = {0, 0};
  {
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < 64; w++) {
      u32 d = ((s[w >> 3] >> ((w & 7) * 4)) & 15u) + carry;
      const u32 over = d > 8u ? 1u : 0u;
      d = over ? 16u - d : d;
      carry = over;
      if ((w & 7) == 0) mag[w >> 3] = 0;
      mag[w >> 3] |= d << ((w & 7) * 4);
      sgn[w >> 5] |= over << (w & 31);
    }
  }
  Proj<F> tab[8];
  {
    Proj<F> p;
    p.x = MbIO<F>::load(xy + i * 2 * WW); p.y = MbIO<F>::load(xy + i * 2 * WW + WW);
    p.z = (inf && inf[i]) ? F::zero() : F::one();
    tab[0] = p;
    tab[1] = pt_double<F>(p);
    tab[2] = pt_add<F>(tab[1], p);
    tab[3] = pt_double<F>(tab[1]);
    tab[4] = pt_add<F>(tab[3], p);
    tab[5] = pt_double<F>(tab[2]);
    tab[6] = pt_add<F>(tab[5], p);
    tab[7] = pt_double<F>(tab[3]);
  }
  Proj<F> acc = pt_identity<F>();
#pragma nounroll
  for (int w = 63; w >= 0; w--) {
    if (w != 63) { acc = pt_double<F>(acc); acc = pt_double<F>(acc); acc = pt_double<F>(acc); acc = pt_double<F>(acc); }
    const u32 d = (mag[w >> 3] >> ((w & 7) * 4)) & 15u;
    const bool neg_d = (sgn[w >> 5] >> (w & 31)) & 1u;
    Proj<F> t = tab[d ? d - 1 : 0];
    if (!d) t = pt_identity<F>();
    t.y = select(neg_d, F::st(neg(t.y)), t.y);
    acc = pt_add<F>(acc, t);
  }
  u32* o = out + i * 3 * WW;
  MbIO<F>::save(acc.x, o); MbIO<F>::save(acc.y, o + WW); MbIO<F>::save(acc.z, o + 2 * WW);
}

// G1 fast path for points the caller vouches for (prime-order subgroup: blsgpu_set_assume_subgroup): the GLV split of the MSM
// (msm.hip.h glv_split: k P = +-|k1| P -+ |k2| phi(P) with 127-bit halves, phi(X : Y : Z) = (BETA X : Y : Z), g1.rs:421-437) turns
// the 64 windows into 32 with TWO additions each over ONE table -- the image of a table entry is one multiplication by BETA:
// 128 doublings x 8 + 67 additions x 12 + 32 = 1 860 field multiplications against 2 852.  Outside the subgroup phi(P) is not
// -[z^2] P, so unverified inputs keep the kernel above.
__global__ void __launch_bounds__(256, 2)
k_mul_batch_glv(const u32* __restrict__ xy, const uint8_t* __restrict__ inf, const u32* __restrict__ scalars, u32* __restrict__ out, size_t n,
                u32* __restrict__ status, int form) {
  typedef FpPolicy F;
  constexpr int WW = 12;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[8];
  if (!scalar_load(scalars, i, form, s)) atomicOr(status, 1u);
  u32 h[2][4];
  glv_split(s, h[0], h[1]);
  const u32 flip[2] = {h[0][3] >> 31, h[1][3] >> 31};        // the whole term is subtracted
  h[0][3] &= 0x7fffffffu; h[1][3] &= 0x7fffffffu;
  // signed digits in [-8, 8] of both halves; |k_j| < 2^126.5 leaves the top window at most 5 + carry: no 33rd window
  u32 mag[2][4], sgn[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < 32; w++) {
      u32 d = ((h[j][w >> 3] >> ((w & 7) * 4)) & 15u) + carry;
      const u32 over = d > 8u ? 1u : 0u;
      d = over ? 16u - d : d;
      carry = over;
      if ((w & 7) == 0) mag[j][w >> 3] = 0;
      mag[j][w >> 3] |= d << ((w & 7) * 4);
      sgn[j] |= over << w;
    }
  }
  Proj<F> tab[8];
  {
    Proj<F> p;
    p.x = MbIO<F>::load(xy + i * 2 * WW); p.y = MbIO<F>::load(xy + i * 2 * WW + WW);
    p.z = (inf && inf[i]) ? F::zero() : F::one();
    tab[0] = p;
    tab[1] = pt_double<F>(p);
    tab[2] = pt_add<F>(tab[1], p);
    tab[3] = pt_double<F>(tab[1]);
    tab[4] = pt_add<F>(tab[3], p);
    tab[5] = pt_double<F>(tab[2]);
    tab[6] = pt_add<F>(tab[5], p);
    tab[7] = pt_double<F>(tab[3]);
  }
  constexpr PLimbs kb = {BLS_BETA};
  Proj<F> acc = pt_identity<F>();
#pragma nounroll
  for (int w = 31; w >= 0; w--) {
    if (w != 31) { acc = pt_double<F>(acc); acc = pt_double<F>(acc); acc = pt_double<F>(acc); acc = pt_double<F>(acc); }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const u32 d = (mag[j][w >> 3] >> ((w & 7) * 4)) & 15u;
      const bool neg_d = (((sgn[j] >> w) & 1u) ^ flip[j]) != 0;
      Proj<F> t = tab[d ? d - 1 : 0];
      if (!d) t = pt_identity<F>();
      if (j) t.x = F::st(mul(t.x, fe1_const(kb)));
      t.y = select(neg_d, F::st(neg(t.y)), t.y);
      acc = pt_add<F>(acc, t);
    }
  }
  u32* o = out + i * 3 * WW;
  MbIO<F>::save(acc.x, o); MbIO<F>::save(acc.y, o + WW); MbIO<F>::save(acc.z, o + 2 * WW);
}

// G2 fast path for vouched points: the four-dimensional split of the MSM (msm.hip.h gls_split: k P = d0 P - d1 psi(P) + d2 psi^2(P) -
// d3 psi^3(P) with |d_j| < 2^63, psi = the untwist-Frobenius-twist endomorphism of g2.rs:847-912) turns the 64 windows into 16 with FOUR
// additions each over ONE table -- the image of a table entry under psi^j is two multiplications by constants and conjugations:
// 64 doublings + 71 additions instead of 256 + 67.  One multiplication per lane pair (pairlane.hip.h), like k_mul_batch<Fp2PairPolicy>.
__global__ void __launch_bounds__(256, 2)
k_mul_batch_gls(const u32* __restrict__ xy, const uint8_t* __restrict__ inf, const u32* __restrict__ scalars, u32* __restrict__ out, size_t n,
                u32* __restrict__ status, int form) {
  typedef Fp2PairPolicy F;
  constexpr int WW = 24;
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 2;
  if (i >= n) return;
  u32 k[10];
  if (!scalar_load(scalars, i, form, k)) atomicOr(status, 1u);
  k[8] = 0; k[9] = 0;
  u64 dg[4]; u32 flip[4];
  gls_split(k, dg, flip);
  // signed digits in [-8, 8] of the four 63-bit digits; |d_j| <= X/2 + 1 leaves the top window at most 6 + carry: no 17th window
  u32 mag[4][2], sgn[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    u32 carry = 0;
    sgn[j] = 0; mag[j][0] = 0; mag[j][1] = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
      u32 d = ((u32)(dg[j] >> (4 * w)) & 15u) + carry;
      const u32 over = d > 8u ? 1u : 0u;
      d = over ? 16u - d : d;
      carry = over;
      mag[j][w >> 3] |= d << ((w & 7) * 4);
      sgn[j] |= over << w;
    }
  }
  u64 magp[4];
#pragma unroll
  for (int j = 0; j < 4; j++) magp[j] = (u64)mag[j][0] | ((u64)mag[j][1] << 32);
  Proj<F> tab[8];
  {
    Proj<F> p;
    p.x = MbIO<F>::load(xy + i * 2 * WW); p.y = MbIO<F>::load(xy + i * 2 * WW + WW);
    p.z = (inf && inf[i]) ? F::zero() : F::one();
    tab[0] = p;
    tab[1] = pt_double<F>(p);
    tab[2] = pt_add<F>(tab[1], p);
    tab[3] = pt_double<F>(tab[1]);
    tab[4] = pt_add<F>(tab[3], p);
    tab[5] = pt_double<F>(tab[2]);
    tab[6] = pt_add<F>(tab[5], p);
    tab[7] = pt_double<F>(tab[3]);
  }
  Proj<F> acc = pt_identity<F>();
#pragma nounroll
  for (int w = 15; w >= 0; w--) {
    if (w != 15) { acc = pt_double<F>(acc); acc = pt_double<F>(acc); acc = pt_double<F>(acc); acc = pt_double<F>(acc); }
#pragma nounroll
    for (int j = 0; j < 4; j++) {
      const u32 d = (u32)(magp[j] >> (4 * w)) & 15u;
      const bool neg_d = (((sgn[j] >> w) & 1u) ^ flip[j]) != 0;
      Proj<F> t = tab[d ? d - 1 : 0];
      if (!d) t = pt_identity<F>();
      if (j & 1) t = pt_psi<F>(t);
      if (j & 2) t = pt_psi2<F>(t);
      t.y = select(neg_d, F::st(neg(t.y)), t.y);
      acc = pt_add<F>(acc, t);
    }
  }
  u32* o = out + i * 3 * WW;
  MbIO<F>::save(acc.x, o); MbIO<F>::save(acc.y, o + WW); MbIO<F>::save(acc.z, o + 2 * WW);
}

}  // namespace bls
