// pairlane.hip.h -- Fp2 arithmetic with one element spread over a PAIR of adjacent lanes.
//
// A G2 bucket accumulator in XYZZ form is 4 Fp2 = 112 registers per lane, and the out-of-line Fp2 product
// needs another ~120: the one-lane-per-chain G2 accumulation kernel needs 444 registers, i.e. one wavefront
// per SIMD, which issues VALU work at half rate (profiles/r01_microbench_valu.md).  Here lane 2k holds the c0
// coefficients and lane 2k+1 the c1 coefficients of every Fp2 value of chain k.  Linear operations are
// lane-local; a product c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0 (src/fp2.rs:205-222) becomes ONE sum of two
// products per lane after swapping operands with the partner lane (v_mov_b32 dpp quad_perm:[1,0,3,2]), a
// square (src/fp2.rs:182-203) one multiplication per lane.  Register use drops to that of the G1 kernel, two
// wavefronts fit a SIMD, and the formulas (curve.hip.h, generic over the field policy) are reused unchanged.
#pragma once
#include "curve.hip.h"

namespace bls {

template <int A, int V> struct FeP { Fe<A, V> v; static constexpr int kA = A, kV = V;
  template <int A2, int V2> DEV operator FeP<A2, V2>() const { FeP<A2, V2> r; r.v = v; return r; } };

DEV bool lane_is_c1() { return (threadIdx.x & 1) != 0; }
#ifdef BLS_DPP_OLD_ZERO
DEV u32 dpp_swap1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false); }
#else
// every lane is written (row / bank masks 0xF, a quad permutation has no invalid source lane), so the "old" operand is dead: passing
// the source itself instead of 0 spares the v_mov_b32 that materialised the zero in front of every exchange
DEV u32 dpp_swap1(u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true); }
#endif
// RULE: never subtract an exchanged value -- exchange the negation and add.  The compiler's DPP-combine pass folds
// `b - dpp(a)` into `v_subrev_u32_dpp d, a, b`, and on this toolchain / gfx950 the *rev* forms do not compute what LLVM assumes:
// measured with inline assembly, `v_subrev_u32_dpp d, x, y quad_perm:[1,0,3,2]` returns dpp(y) - x (v_lshlrev_b32_dpp is off
// in the same way; v_mov / v_add / v_sub / v_and / v_xor with DPP behave as documented).  Whether a subtraction was folded
// depended on register allocation, so any unrelated edit of a kernel could silently change results (round-2 notes in
// DESIGN.md).  `a + partner(neg(a))` costs the same instructions as the folded subtraction, keeps the same static bounds and
// can only fold into v_add_u32_dpp; __graft_entry__.check_isa() rejects a library containing a v_subrev_*_dpp.
// Also: every exchange must be executed by BOTH lanes of a pair -- never inside a lane-dependent conditional expression
// (a DPP read of a lane that is masked off returns 0).
template <int A, int V> DEV Fe<A, V> partner(const Fe<A, V>& a) {
  Fe<A, V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = dpp_swap1(a.l[i]);
  return r;
}
DEV bool partner_flag(bool f) { return dpp_swap1(f ? 1u : 0u) != 0; }

// ---- lane-local linear operations ---------------------------------------------------------------------------
template <int A1, int V1, int A2, int V2> DEV auto add(const FeP<A1, V1>& a, const FeP<A2, V2>& b) { FeP<A1 + A2, V1 + V2> r; r.v = add(a.v, b.v); return r; }
template <int A1, int V1, int A2, int V2> DEV auto sub(const FeP<A1, V1>& a, const FeP<A2, V2>& b) { FeP<A1 + A2 + 1, V1 + V2 + 1> r; r.v = sub(a.v, b.v); return r; }
template <int A, int V> DEV auto neg(const FeP<A, V>& a) { FeP<A + 1, V + 1> r; r.v = neg(a.v); return r; }
template <int A, int V> DEV auto dbl(const FeP<A, V>& a) { FeP<2 * A, 2 * V> r; r.v = dbl(a.v); return r; }
template <int K, int A, int V> DEV auto mul_small(const FeP<A, V>& a) { FeP<K * A, K * V> r; r.v = mul_small<K>(a.v); return r; }
template <int A, int V> DEV auto norm(const FeP<A, V>& a) { FeP<1, V> r; r.v = norm(a.v); return r; }
template <int V> DEV auto reduce_v(const FeP<1, V>& a) { FeP<1, 2> r; r.v = reduce_v(a.v); return r; }
template <int A, int V> DEV auto select(bool c, const FeP<A, V>& a, const FeP<A, V>& b) { FeP<A, V> r; r.v = select(c, a.v, b.v); return r; }

// (a0 + a1 u)(1 + u) = (a0 - a1) + (a0 + a1) u
template <int A, int V> DEV auto mul_by_nonresidue(const FeP<A, V>& a) {
  // c0 lane: a_me - a_other, c1 lane: a_me + a_other -- ONE exchange: the c1 lane sends its negation, the c0 lane itself (round 6: a
  // negation, a select and an addition that can take the DPP operand, 3 instructions per limb; the form that built both a_me - a_other
  // and a_me + a_other and selected took 6)
  typedef Fe<A + 1, V + 1> ST;
  const ST send = select(lane_is_c1(), neg(a.v), (ST)a.v);
  FeP<2 * A + 1, 2 * V + 1> r;
  r.v = add(a.v, partner(send));
  return r;
}

constexpr int pair_mul_v(int v1, int v2) { return 1 + (v1 * v2 + (v1 + 1) * v2 + V_DIV - 1) / V_DIV; }

// Fp2 product: one sum of two products per lane.  With a / b this lane's coefficients and a' / b' the partner's:
//   c0 lane:  a0 b0 - a1 b1 = a * b + (-a') * b'        c1 lane:  a0 b1 + a1 b0 = a' * b + a * b'
template <int A1, int V1, int A2, int V2>
DEV FeP<1, pair_mul_v(V1, V2)> mul_inl(const FeP<A1, V1>& a, const FeP<A2, V2>& b) {
  static_assert(A1 * A2 + (A1 + 1) * A2 + 1 <= MAX_A_PROD + 1, "pair-lane fe2 mul: limb bound too large, norm() an operand");
  const bool c1 = lane_is_c1();
  auto ao = partner(a.v); auto bo = partner(b.v);
  auto x0 = select(c1, ao, a.v);
  Fe<A1 + 1, V1 + 1> x1 = select(c1, (Fe<A1 + 1, V1 + 1>)a.v, partner(neg(a.v)));
  FeP<1, pair_mul_v(V1, V2)> r;
  r.v = from_v16<pair_mul_v(V1, V2)>(fe_sop2_body(to_v16(x0), to_v16(b.v), to_v16(x1), to_v16(bo)));
  return r;
}
// Fp2 square: c0 = (a0 + a1)(a0 - a1), c1 = 2 a0 a1 -- one multiplication per lane
template <int A, int V>
DEV auto sqr_inl(const FeP<A, V>& a) {
  static_assert(2 * A * (2 * A + 1) <= MAX_A_PROD, "pair-lane fe2 sqr: norm() the operand");
  const bool c1 = lane_is_c1();
  auto ao = partner(a.v);
  typedef Fe<2 * A, 2 * V> XT;
  typedef Fe<2 * A + 1, 2 * V + 1> YT;
  typedef Fe<A + 1, V + 1> NT;
  XT x = add(ao, select(c1, ao, a.v));               // c1 lane: 2 a0 ; c0 lane: a0 + a1
  // y: c0 lane a0 - a1, c1 lane a1 -- the c1 lane sends its negation, the c0 lane sends zero, and both ADD what they receive (one select
  // on the sender's side instead of a subtraction and a select on the receiver's)
  NT zero;
#pragma unroll
  for (int i = 0; i < NL; i++) zero.l[i] = 0;
  YT y = add(a.v, partner(select(c1, neg(a.v), zero)));
  FeP<1, mul_v(2 * V, 2 * V + 1)> r;
  r.v = mul_inl(x, y);
  return r;
}
// both coefficients zero (decided identically in both lanes of the pair)
template <int A, int V> DEV bool is_zero_fast(const FeP<A, V>& a) {
  bool m = maybe_zero(a.v);
  bool pm = partner_flag(m);                    // both lanes of the pair execute the exchange
  if (!(m && pm)) return false;                 // ... and take the same branch
  bool z = is_zero(a.v);
  bool pz = partner_flag(z);
  return z && pz;
}


// ---- operations of the pairing code (pairing.hip.h) ---------------------------------------------------------
// out-of-line product / square with fixed operand bounds (limbs <= 2 (2^28-1), value < 160 p): the pairing code
// has hundreds of call sites
constexpr int FEP_IN_A = 2, FEP_IN_V = 160;
DEVNI v16 fep_mul_raw(v16 a, v16 b) {
  constexpr PLimbs bias = make_bias(FEP_IN_V + 1, FEP_IN_A);
  const bool c1 = lane_is_c1();
  v16 x0, x1, bo;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    // both lanes execute every exchange; the c0 lane needs -a' and receives it as the partner's (bias - a): no subtraction
    // of an exchanged value (see the rule at dpp_swap1)
    const u32 ao = dpp_swap1(a[i]), nao = dpp_swap1(bias.l[i] - a[i]);
    bo[i] = dpp_swap1(b[i]);
    x0[i] = c1 ? ao : a[i];
    x1[i] = c1 ? a[i] : nao;
  }
  x0[14] = x0[15] = x1[14] = x1[15] = bo[14] = bo[15] = 0;
  return fe_sop2_body(x0, b, x1, bo);
}
constexpr int fep_mul_v(int v1, int v2) { return 1 + (v1 * v2 + (FEP_IN_V + 1) * v2 + V_DIV - 1) / V_DIV; }
template <int A1, int V1, int A2, int V2>
DEV FeP<1, fep_mul_v(V1, V2)> mul_ni(const FeP<A1, V1>& a, const FeP<A2, V2>& b) {
  static_assert(V1 <= FEP_IN_V && V2 <= FEP_IN_V, "pair-lane fe2 mul: operand value bound too large");
  FeP<1, fep_mul_v(V1, V2)> r;
  if constexpr (A1 <= FEP_IN_A && A2 <= FEP_IN_A) r.v = from_v16<fep_mul_v(V1, V2)>(fep_mul_raw(to_v16(a.v), to_v16(b.v)));
  else r.v = from_v16<fep_mul_v(V1, V2)>(fep_mul_raw(to_v16(norm(a.v)), to_v16(norm(b.v))));
  return r;
}
template <int A, int V>
DEV auto sqr_ni(const FeP<A, V>& a) {
  const bool c1 = lane_is_c1();
  auto ao = partner(a.v);
  typedef Fe<2 * A, 2 * V> XT;
  typedef Fe<2 * A + 1, 2 * V + 1> YT;
  typedef Fe<A + 1, V + 1> NT;
  XT x = add(ao, select(c1, ao, a.v));               // c1 lane: 2 a0 ; c0 lane: a0 + a1
  NT zero;
#pragma unroll
  for (int i = 0; i < NL; i++) zero.l[i] = 0;
  YT y = add(a.v, partner(select(c1, neg(a.v), zero)));     // c0 lane: a0 - a1, c1 lane: a1 (see sqr_inl)
  auto p = mulx(x, y);
  FeP<1, decltype(p)::kV> r; r.v = p;
  return r;
}
// the product / square the generic curve formulas pick up: out of line by default (the ten products of a mixed
// addition inlined are ~63 KB of code, more than the instruction cache)
#ifdef BLS_PAIR_MUL_INLINE
template <int A1, int V1, int A2, int V2> DEV auto mul(const FeP<A1, V1>& a, const FeP<A2, V2>& b) { return mul_inl(a, b); }
template <int A, int V> DEV auto sqr(const FeP<A, V>& a) { return sqr_inl(a); }
#else
template <int A1, int V1, int A2, int V2> DEV auto mul(const FeP<A1, V1>& a, const FeP<A2, V2>& b) { return mul_ni(a, b); }
template <int A, int V> DEV auto sqr(const FeP<A, V>& a) { return sqr_ni(a); }
#endif
// a0 - a1 u
template <int A, int V> DEV auto conj(const FeP<A, V>& a) {
  FeP<A + 1, V + 1> r; r.v = select(lane_is_c1(), neg(a.v), (Fe<A + 1, V + 1>)a.v); return r;
}
// (a0 + a1 u) u = -a1 + a0 u
template <int A, int V> DEV auto mul_by_u(const FeP<A, V>& a) {
  // c0 lane: -a1 (the c1 lane sends its negation), c1 lane: a0 (the c0 lane sends itself): one exchange
  typedef Fe<A + 1, V + 1> ST;
  FeP<A + 1, V + 1> r; r.v = partner(select(lane_is_c1(), neg(a.v), (ST)a.v)); return r;
}
// Fp2 x Fp (k identical in both lanes)
template <int A1, int V1, int A2, int V2> DEV auto mul_fp(const FeP<A1, V1>& a, const Fe<A2, V2>& k) {
  FeP<1, mul_v(V1, V2)> r; r.v = mul(a.v, k); return r;
}
// 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 + a1^2); both lanes run the same base-field inversion
template <int A, int V> DEV auto inv(const FeP<A, V>& a) {
  auto s = sqr(a.v);
  auto n = add(s, partner(s));
  auto t = inv(n);
  typedef Fe<decltype(t)::kA + 1, decltype(t)::kV + 1> TT;
  TT ts = select(lane_is_c1(), neg(t), (TT)t);
  auto p = mul(a.v, ts);
  static_assert(decltype(p)::kV <= 2, "pair-lane fe2 inv bound");
  FeP<1, 2> r; r.v = p; return r;
}

// exact zero test (both coefficients), decided identically in both lanes
template <int A, int V> DEV bool is_zero(const FeP<A, V>& a) { bool z = is_zero(a.v); return z && partner_flag(z); }

// ---- field policy: G2 over lane pairs --------------------------------------------------------------------
struct Fp2PairPolicy {
  typedef FeP<1, VS2> elem;
  typedef FeP<1, 1> aff_elem;
  template <int A, int V> static DEV elem st(const FeP<A, V>& a) {
    elem r;
    if constexpr (V <= VS2) r.v = norm(a.v); else r.v = reduce_v(norm(a.v));
    return r;
  }
  // 3b' * a = 12 (1 + u) a     (g2.rs:196,650-652)
  template <int A, int V> static DEV auto mul_by_3b(const FeP<A, V>& a) {
    auto t = norm(mul_by_nonresidue(norm(a)));
    return reduce_v(norm(mul_small<12>(t)));
  }
  static DEV elem zero() { elem r; r.v = (Fe<1, VS2>)fe_zero(); return r; }
  static DEV elem one() { elem r; r.v = select(lane_is_c1(), (Fe<1, VS2>)fe_zero(), (Fe<1, VS2>)fe_one()); return r; }
};

}  // namespace bls
