// team.hip.h -- lane-cooperative point arithmetic for the latency-bound tails.
//
// The last phases of an MSM (upper levels of the bucket reduction, the Horner combine over the windows, the
// cross-rank fold) are dependent chains of a few hundred point operations with almost no parallelism across
// chains; executed one chain per lane, each complete addition costs ~13 us because a lone wavefront issues
// its 12 field multiplications back to back.  The complete RCB formulas have only two dependent layers of
// independent products (6 + 6 for an addition, 4 + 4 for a doubling), so a TEAM of 8 adjacent lanes keeps the
// operands replicated in registers, lets lane k compute product k of the layer, and exchanges the results
// through an LDS mailbox: 2 multiplication latencies per point operation instead of 8-12.
// Same formulas as curve.hip.h (g1.rs:638-667, :670-712), hence the same projective triples.
#pragma once
#include "curve.hip.h"

namespace bls {

constexpr int TEAM = 8;                      // lanes per team (adjacent lanes of one wavefront)
constexpr int TEAM_SLOTS = 6;                // products per layer (max)

template <class F> struct TeamTraits;
template <> struct TeamTraits<FpPolicy> {
  static constexpr int WORDS = NL;
  template <class T> static DEV void put(u32* m, const T& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) m[i] = a.l[i];
  }
  template <class T> static DEV void get(const u32* m, T& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) a.l[i] = m[i];
  }
};
template <> struct TeamTraits<Fp2Policy> {
  static constexpr int WORDS = 2 * NL;
  template <class T> static DEV void put(u32* m, const T& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) { m[i] = a.c0.l[i]; m[NL + i] = a.c1.l[i]; }
  }
  template <class T> static DEV void get(const u32* m, T& a) {
#pragma unroll
    for (int i = 0; i < NL; i++) { a.c0.l[i] = m[i]; a.c1.l[i] = m[NL + i]; }
  }
};
template <class F> constexpr int team_lds_words(int threads) { return threads / TEAM * TEAM_SLOTS * TeamTraits<F>::WORDS; }

// static type of a product of two OpT operands (limbs normalised, tight value bound)
template <class OpT> struct ProdOf { typedef decltype(norm(mul(OpT(), OpT()))) type; };

// K independent products a[k]*b[k], one per lane of the team; every lane gets all K results.
// `mbox` points at this team's TEAM_SLOTS * WORDS words of LDS.  Contains block barriers: must be called by
// every thread of the block.  OpT is a common (widened) static bound of all operands of the layer.
template <class F, int K, class OpT>
DEV void team_mul(const OpT (&a)[K], const OpT (&b)[K], typename ProdOf<OpT>::type (&out)[K], u32* mbox, int tl) {
  OpT sa = a[0], sb = b[0];
#pragma unroll
  for (int k = 1; k < K; k++) { sa = select(tl == k, a[k], sa); sb = select(tl == k, b[k], sb); }
  typename ProdOf<OpT>::type p = norm(mul(sa, sb));
  __syncthreads();                            // previous readers of the mailbox are done
  if (tl < K) TeamTraits<F>::put(mbox + tl * TeamTraits<F>::WORDS, p);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) TeamTraits<F>::get(mbox + k * TeamTraits<F>::WORDS, out[k]);
}

// RCB15 Algorithm 7 with the 12 products in two team layers
template <class F>
DEV Proj<F> pt_add_team(const Proj<F>& p, const Proj<F>& q, u32* mbox, int tl) {
  typedef typename F::elem E;
  typedef typename F::op_sum S;
  typedef typename F::op_wide W;
  S a1[6] = {(S)p.x, (S)p.y, (S)p.z, (S)add(p.x, p.y), (S)add(p.y, p.z), (S)add(p.x, p.z)};
  S b1[6] = {(S)q.x, (S)q.y, (S)q.z, (S)add(q.x, q.y), (S)add(q.y, q.z), (S)add(q.x, q.z)};
  typename ProdOf<S>::type r[6];
  team_mul<F, 6, S>(a1, b1, r, mbox, tl);
  // r = {t0, t1, t2, t3, t4, x3}
  auto t3 = norm(sub(r[3], add(r[0], r[1])));
  auto t4 = norm(sub(r[4], add(r[1], r[2])));
  auto y3 = norm(sub(r[5], add(r[0], r[2])));
  auto t0 = norm(add(dbl(r[0]), r[0]));
  auto t2 = F::mul_by_3b(r[2]);
  auto z3 = norm(add(r[1], t2));
  auto t1 = norm(sub(r[1], t2));
  auto y3b = F::mul_by_3b(y3);
  W a2[6] = {(W)t4, (W)t3, (W)y3b, (W)t1, (W)t0, (W)z3};
  W b2[6] = {(W)y3b, (W)t1, (W)t0, (W)z3, (W)t3, (W)t4};
  typename ProdOf<W>::type s[6];
  team_mul<F, 6, W>(a2, b2, s, mbox, tl);
  // s = {t4*y3, t3*t1, y3*t0, t1*z3, t0*t3, z3*t4}
  Proj<F> o;
  o.x = F::st(sub(s[1], s[0]));
  o.y = F::st(add(s[3], s[2]));
  o.z = F::st(add(s[5], s[4]));
  return o;
}

// RCB15 Algorithm 9 with the 8 products in two team layers
template <class F>
DEV Proj<F> pt_double_team(const Proj<F>& p, u32* mbox, int tl) {
  typedef typename F::elem E;
  typedef typename F::op_wide W;
  E a1[4] = {p.y, p.y, p.z, p.x};
  E b1[4] = {p.y, p.z, p.z, p.y};
  typename ProdOf<E>::type r[4];
  team_mul<F, 4, E>(a1, b1, r, mbox, tl);
  // r = {t0 = y^2, t1 = y z, z^2, x y}
  auto z3 = norm(mul_small<8>(r[0]));
  auto t2 = F::mul_by_3b(r[2]);
  auto y3 = norm(add(r[0], t2));
  auto t2b = norm(mul_small<3>(t2));
  auto t0b = norm(sub(r[0], t2b));
  W a2[4] = {(W)t2, (W)r[1], (W)t0b, (W)t0b};
  W b2[4] = {(W)z3, (W)z3, (W)y3, (W)r[3]};
  typename ProdOf<W>::type s[4];
  team_mul<F, 4, W>(a2, b2, s, mbox, tl);
  // s = {x3 = t2 z3, z3' = t1 z3, t0 y3, t0 (x y)}
  Proj<F> o;
  o.x = F::st(dbl(s[3]));
  o.y = F::st(add(s[0], s[2]));
  o.z = F::st(s[1]);
  return o;
}

}  // namespace bls
