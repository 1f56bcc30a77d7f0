// fp2.hip.h -- Fp2 = Fp[u]/(u^2+1) on the lazy 14x28 representation.
//
// Reference: /root/reference/src/fp2.rs  mul :205-222, square :182-203, add/sub/neg :224-243,
// mul_by_nonresidue :156-166, conjugate :148-153, invert :300-319.  The reference computes the two
// coefficients of a product as interleaved sums of products; mathematically the same field elements
// are produced here with 3 base-field multiplications (Karatsuba) whose subtractions are carry-free.
#pragma once
#include "fe.hip.h"

namespace bls {

template <int A, int V>
struct Fe2 {
  Fe<A, V> c0, c1;
  template <int A2, int V2>
  DEV operator Fe2<A2, V2>() const { Fe2<A2, V2> r; r.c0 = c0; r.c1 = c1; return r; }
};

template <int A1, int V1, int A2, int V2>
DEV auto add(const Fe2<A1, V1>& a, const Fe2<A2, V2>& b) {
  Fe2<A1 + A2, V1 + V2> r; r.c0 = add(a.c0, b.c0); r.c1 = add(a.c1, b.c1); return r;
}
template <int A1, int V1, int A2, int V2>
DEV auto sub(const Fe2<A1, V1>& a, const Fe2<A2, V2>& b) {
  Fe2<A1 + A2 + 1, V1 + V2 + 1> r; r.c0 = sub(a.c0, b.c0); r.c1 = sub(a.c1, b.c1); return r;
}
template <int A, int V>
DEV auto neg(const Fe2<A, V>& a) { Fe2<A + 1, V + 1> r; r.c0 = neg(a.c0); r.c1 = neg(a.c1); return r; }
template <int A, int V>
DEV auto dbl(const Fe2<A, V>& a) { Fe2<2 * A, 2 * V> r; r.c0 = dbl(a.c0); r.c1 = dbl(a.c1); return r; }
template <int K, int A, int V>
DEV auto mul_small(const Fe2<A, V>& a) {
  Fe2<K * A, K * V> r; r.c0 = mul_small<K>(a.c0); r.c1 = mul_small<K>(a.c1); return r;
}
template <int A, int V>
DEV auto norm(const Fe2<A, V>& a) { Fe2<1, V> r; r.c0 = norm(a.c0); r.c1 = norm(a.c1); return r; }
template <int V>
DEV auto reduce_v(const Fe2<1, V>& a) { Fe2<1, 2> r; r.c0 = reduce_v(a.c0); r.c1 = reduce_v(a.c1); return r; }
template <int A, int V>
DEV auto conj(const Fe2<A, V>& a) { Fe2<A + 1, V + 1> r; r.c0 = a.c0; r.c1 = neg(a.c1); return r; }

// (a0 + a1 u)(u + 1) = (a0 - a1) + (a0 + a1) u      (fp2.rs:156-166)
template <int A, int V>
DEV auto mul_by_nonresidue(const Fe2<A, V>& a) {
  Fe2<2 * A + 1, 2 * V + 1> r; r.c0 = sub(a.c0, a.c1); r.c1 = add(a.c0, a.c1); return r;
}

// c0 = a0 b0 - a1 b1 ; c1 = a0 b1 + a1 b0, each coefficient ONE sum-of-products with a single
// Montgomery reduction (fe2_mul_raw), as the reference does (fp2.rs:205-222).  Output limbs normalised.
constexpr int fe2_mul_v(int v1, int v2) { return 1 + (v1 * v2 + (FE2_IN_V + 1) * v2 + V_DIV - 1) / V_DIV; }
template <int A1, int V1, int A2, int V2>
DEV Fe2<1, fe2_mul_v(V1, V2)> mul(const Fe2<A1, V1>& a, const Fe2<A2, V2>& b) {
  static_assert(V1 <= FE2_IN_V && V2 <= FE2_IN_V, "fe2 mul: operand value bound too large");
  V16x2 t;
  if constexpr (A1 <= FE2_IN_A && A2 <= FE2_IN_A) {
    t = fe2_mul_raw(to_v16(a.c0), to_v16(a.c1), to_v16(b.c0), to_v16(b.c1));
  } else {
    auto an = norm(a); auto bn = norm(b);
    t = fe2_mul_raw(to_v16(an.c0), to_v16(an.c1), to_v16(bn.c0), to_v16(bn.c1));
  }
  Fe2<1, fe2_mul_v(V1, V2)> r;
  r.c0 = from_v16<fe2_mul_v(V1, V2)>(t.c0);
  r.c1 = from_v16<fe2_mul_v(V1, V2)>(t.c1);
  return r;
}

// Fp2 x Fp (both coefficients scaled)
template <int A1, int V1, int A2, int V2>
DEV auto mul_fp(const Fe2<A1, V1>& a, const Fe<A2, V2>& k) {
  Fe2<1, mul_v(V1, V2)> r; r.c0 = mul(a.c0, k); r.c1 = mul(a.c1, k); return r;
}

// complex squaring (fp2.rs:182-203): c0 = (a0+a1)(a0-a1), c1 = 2 a0 a1
template <int A, int V>
DEV auto sqr(const Fe2<A, V>& a) {
  auto s = add(a.c0, a.c1);
  auto d = sub(a.c0, a.c1);
  auto c0 = mulx(s, d);
  auto c1 = mulx(dbl(a.c0), a.c1);
  constexpr int VO = decltype(c1)::kV > decltype(c0)::kV ? decltype(c1)::kV : decltype(c0)::kV;
  Fe2<1, VO> r; r.c0 = c0; r.c1 = c1;
  return r;
}

template <int A, int V>
DEV auto select(bool c, const Fe2<A, V>& a, const Fe2<A, V>& b) {
  Fe2<A, V> r; r.c0 = select(c, a.c0, b.c0); r.c1 = select(c, a.c1, b.c1); return r;
}
template <int A, int V>
DEV bool is_zero(const Fe2<A, V>& a) { return is_zero(a.c0) & is_zero(a.c1); }
template <int A, int V>
DEV bool is_zero_fast(const Fe2<A, V>& a) { return maybe_zero(a.c0) && maybe_zero(a.c1) && is_zero(a); }

// 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 + a1^2)        (fp2.rs:300-319)
template <int A, int V>
DEV auto inv(const Fe2<A, V>& a) {
  auto n = add(sqr(a.c0), sqr(a.c1));
  auto t = inv(n);
  Fe2<1, 2> r;
  auto c0 = mul(a.c0, t);
  auto c1 = mul(a.c1, neg(t));
  static_assert(decltype(c0)::kV <= 2 && decltype(c1)::kV <= 2, "fe2 inv bound");
  r.c0 = c0; r.c1 = c1;
  return r;
}

// storage forms
typedef Fe2<1, 1> fe2_1;        // canonical input
constexpr int VS2 = 32;
typedef Fe2<1, VS2> fe2;        // working value
template <int A, int V>
DEV fe2 store2(const Fe2<A, V>& a) {
  fe2 r;
  if constexpr (V <= VS2) { r.c0 = norm(a.c0); r.c1 = norm(a.c1); }
  else { r.c0 = reduce_v(norm(a.c0)); r.c1 = reduce_v(norm(a.c1)); }
  return r;
}
DEV fe2 fe2_zero() { fe2 r; r.c0 = (Fe<1, VS2>)fe_zero(); r.c1 = (Fe<1, VS2>)fe_zero(); return r; }
DEV fe2 fe2_one() { fe2 r; r.c0 = (Fe<1, VS2>)fe_one(); r.c1 = (Fe<1, VS2>)fe_zero(); return r; }

// u (c0 + c1 u) = -c1 + c0 u
template <int A, int V> DEV auto mul_by_u(const Fe2<A, V>& a) { Fe2<A + 1, V + 1> r; r.c0 = neg(a.c1); r.c1 = a.c0; return r; }

}  // namespace bls
