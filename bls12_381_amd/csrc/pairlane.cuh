// pairlane.cuh -- Fp2 arithmetic with one element spread over a PAIR of adjacent lanes.
//
// A G2 bucket accumulator in XYZZ form is 4 Fp2 = 112 registers per lane, and the out-of-line Fp2 product
// needs another ~120: the one-lane-per-chain G2 accumulation kernel needs 444 registers, i.e. one wavefront
// per SIMD, which issues VALU work at half rate (profiles/r01_microbench_valu.md).  Here lane 2k holds the c0
// coefficients and lane 2k+1 the c1 coefficients of every Fp2 value of chain k.  Linear operations are
// lane-local; a product c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0 (src/fp2.rs:205-222) becomes ONE sum of two
// products per lane after swapping operands with the partner lane (v_mov_b32 dpp quad_perm:[1,0,3,2]), a
// square (src/fp2.rs:182-203) one multiplication per lane.  Register use drops to that of the G1 kernel, two
// wavefronts fit a SIMD, and the formulas (curve.cuh, generic over the field policy) are reused unchanged.
#pragma once
#include "curve.cuh"

namespace bls {

template <int A, int V> struct FeP { Fe<A, V> v; static constexpr int kA = A, kV = V;
  template <int A2, int V2> DEV operator FeP<A2, V2>() const { FeP<A2, V2> r; r.v = v; return r; } };

DEV bool lane_is_c1() { return (threadIdx.x & 1) != 0; }
DEV u32 dpp_swap1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false); }
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
  auto o = partner(a.v);
  FeP<2 * A + 1, 2 * V + 1> r;
  // c0 lane: a_me - a_other ; c1 lane: a_other + a_me
  auto m = sub(a.v, o);
  auto p = add(a.v, o);
  r.v = select(lane_is_c1(), (Fe<2 * A + 1, 2 * V + 1>)p, m);
  return r;
}

constexpr int pair_mul_v(int v1, int v2) { return 1 + (v1 * v2 + (v1 + 1) * v2 + V_DIV - 1) / V_DIV; }

// Fp2 product: one sum of two products per lane
template <int A1, int V1, int A2, int V2>
DEV FeP<1, pair_mul_v(V1, V2)> mul(const FeP<A1, V1>& a, const FeP<A2, V2>& b) {
  static_assert(A1 * A2 + (A1 + 1) * A2 + 1 <= MAX_A_PROD + 1, "pair-lane fe2 mul: limb bound too large, norm() an operand");
  const bool c1 = lane_is_c1();
  auto ao = partner(a.v); auto bo = partner(b.v);
  // a0 / a1 / b0 / b1 as seen from this lane
  auto a0 = select(c1, ao, a.v), a1 = select(c1, a.v, ao);
  auto b0 = select(c1, bo, b.v), b1 = select(c1, b.v, bo);
  // c0 lane: a0 b0 + (-a1) b1 ; c1 lane: a0 b1 + a1 b0
  Fe<A1 + 1, V1 + 1> t1 = select(c1, (Fe<A1 + 1, V1 + 1>)a1, neg(a1));
  auto y0 = select(c1, b1, b0), y1 = select(c1, b0, b1);
  FeP<1, pair_mul_v(V1, V2)> r;
  r.v = from_v16<pair_mul_v(V1, V2)>(fe_sop2_body(to_v16(a0), to_v16(y0), to_v16(t1), to_v16(y1)));      // inlined: 4 vector operands do not fit the call ABI
  return r;
}
// Fp2 square: c0 = (a0 + a1)(a0 - a1), c1 = 2 a0 a1 -- one multiplication per lane
template <int A, int V>
DEV auto sqr(const FeP<A, V>& a) {
  static_assert(2 * A * (2 * A + 1) <= MAX_A_PROD, "pair-lane fe2 sqr: norm() the operand");
  const bool c1 = lane_is_c1();
  auto ao = partner(a.v);
  auto a0 = select(c1, ao, a.v), a1 = select(c1, a.v, ao);
  typedef Fe<2 * A, 2 * V> XT;
  typedef Fe<2 * A + 1, 2 * V + 1> YT;
  XT x = select(c1, dbl(a0), add(a0, a1));
  YT y = select(c1, (YT)a1, sub(a0, a1));
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

// ---- field policy: G2 over lane pairs --------------------------------------------------------------------
struct Fp2PairPolicy {
  typedef FeP<1, VS2> elem;
  typedef FeP<1, 1> aff_elem;
  template <int A, int V> static DEV elem st(const FeP<A, V>& a) {
    elem r;
    if constexpr (V <= VS2) r.v = norm(a.v); else r.v = reduce_v(norm(a.v));
    return r;
  }
  static DEV elem zero() { elem r; r.v = (Fe<1, VS2>)fe_zero(); return r; }
  static DEV elem one() { elem r; r.v = select(lane_is_c1(), (Fe<1, VS2>)fe_zero(), (Fe<1, VS2>)fe_one()); return r; }
};

}  // namespace bls
