// curve.hip.h -- G1 / G2 point arithmetic: the reference's complete RCB15 formulas (a = 0) written once
// over a field policy (Fp for G1, Fp2 for G2).
//
// Reference: /root/reference/src/g1.rs  double :638-667 (Alg. 9), add :670-712 (Alg. 7),
// add_mixed :715-752 (Alg. 8), mul_by_3b :597-601, identity (0:1:0) :605-611;  src/g2.rs the same
// code over Fp2 (:709-738, :741-783, :786-823) with mul_by_3b = multiplication by 3b' = 12 + 12u
// (:196, :650-652).  Homogeneous projective (X:Y:Z), x = X/Z; the formulas have no exceptional cases
// (P+P, P-P, identity operands), which is what keeps Pippenger's bucket accumulation free of
// divergence on a 64-wide wavefront.  Bounds of the lazy field arithmetic are proven by the types.
#pragma once
#include "fp2.hip.h"

namespace bls {

// ---- field policies ------------------------------------------------------------------------------
struct FpPolicy {
  typedef fe elem;          // working storage (value < 12p)
  typedef fe1 aff_elem;     // canonical affine coordinate
  typedef Fe<2, 2 * VS> op_sum;     // sum of two stored values (first product layer of the team formulas)
  typedef Fe<1, 88> op_wide;        // any normalised intermediate of the complete formulas (second layer)
  template <class T> static DEV elem st(const T& a) { return store(a); }
  static DEV elem zero() { return fe_zero(); }
  static DEV elem one() { return fe_one(); }
  // 3b * a = 12 a     (g1.rs:597-601)
  template <int A, int V> static DEV auto mul_by_3b(const Fe<A, V>& a) {
    if constexpr (12 * A <= 15) return norm(mul_small<12>(a));
    else return norm(mul_small<12>(norm(a)));
  }
};
struct Fp2Policy {
  typedef fe2 elem;
  typedef fe2_1 aff_elem;
  typedef Fe2<2, 2 * VS2> op_sum;
  typedef Fe2<1, 88> op_wide;
  template <class T> static DEV elem st(const T& a) { return store2(a); }
  static DEV elem zero() { return fe2_zero(); }
  static DEV elem one() { return fe2_one(); }
  // 3b' * a = (12 + 12u)(a0 + a1 u) = 12(a0 - a1) + 12(a0 + a1) u     (g2.rs:196,650-652)
  template <int A, int V> static DEV auto mul_by_3b(const Fe2<A, V>& a) {
    auto n = norm(a);
    auto t = norm(mul_by_nonresidue(n));       // <1, 2V+1>
    return reduce_v(norm(mul_small<12>(t)));  // back below 2p: keeps every Fp2 product input small
  }
};

template <class F> struct Aff { typename F::aff_elem x, y; };
template <class F> struct Proj { typename F::elem x, y, z; };

template <class F> DEV Proj<F> pt_identity() {
  Proj<F> r; r.x = F::zero(); r.y = F::one(); r.z = F::zero(); return r;
}

// RCB15 Algorithm 8 (mixed addition).  q_inf mirrors conditional_select(&tmp, self, rhs.is_identity()).
template <class F>
DEV Proj<F> pt_add_mixed(const Proj<F>& p, const Aff<F>& q, bool q_inf) {
  auto t0 = mul(p.x, q.x);
  auto t1 = mul(p.y, q.y);
  auto t3 = mul(add(q.x, q.y), add(p.x, p.y));
  auto t4 = add(t0, t1);
  auto t3b = norm(sub(t3, t4));
  auto t4b = norm(add(mul(q.y, p.z), p.y));
  auto y3 = norm(add(mul(q.x, p.z), p.x));
  auto t0b = norm(add(dbl(t0), t0));          // 3 t0
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
  if (q_inf) r = p;
  return r;
}

// RCB15 Algorithm 7 (projective + projective)
template <class F>
DEV Proj<F> pt_add(const Proj<F>& p, const Proj<F>& q) {
  auto t0 = mul(p.x, q.x);
  auto t1 = mul(p.y, q.y);
  auto t2 = mul(p.z, q.z);
  auto t3 = mul(add(p.x, p.y), add(q.x, q.y));
  auto t3b = norm(sub(t3, add(t0, t1)));
  auto t4 = mul(add(p.y, p.z), add(q.y, q.z));
  auto t4b = norm(sub(t4, add(t1, t2)));
  auto x3 = mul(add(p.x, p.z), add(q.x, q.z));
  auto y3 = norm(sub(x3, add(t0, t2)));
  auto t0b = norm(add(dbl(t0), t0));
  auto t2b = F::mul_by_3b(t2);
  auto z3 = norm(add(t1, t2b));
  auto t1b = norm(sub(t1, t2b));
  auto y3b = F::mul_by_3b(y3);
  auto x3b = mul(t4b, y3b);
  auto t2c = mul(t3b, t1b);
  auto x3c = sub(t2c, x3b);
  auto y3c = mul(y3b, t0b);
  auto t1c = mul(t1b, z3);
  auto y3d = add(t1c, y3c);
  auto t0c = mul(t0b, t3b);
  auto z3b = mul(z3, t4b);
  auto z3c = add(z3b, t0c);
  Proj<F> r;
  r.x = F::st(x3c); r.y = F::st(y3d); r.z = F::st(z3c);
  return r;
}

// RCB15 Algorithm 9 (doubling).  The reference's final select-to-identity for Z = 0 is implied:
// with Z = 0 the formula yields (0 : Y^3-ish : 0), the same point (X:Y:0 is the identity for any Y != 0).
template <class F>
DEV Proj<F> pt_double(const Proj<F>& p) {
  auto t0 = sqr(p.y);
  auto z3 = norm(mul_small<8>(t0));
  auto t1 = mul(p.y, p.z);
  auto t2 = F::mul_by_3b(sqr(p.z));
  auto x3 = mul(t2, z3);
  auto y3 = norm(add(t0, t2));
  auto z3b = mul(t1, z3);
  auto t2b = norm(mul_small<3>(t2));
  auto t0b = norm(sub(t0, t2b));
  auto y3b = mul(t0b, y3);
  auto y3c = add(x3, y3b);
  auto t1b = mul(p.x, p.y);
  auto x3b = mul(t0b, t1b);
  auto x3c = dbl(x3b);
  Proj<F> r;
  r.x = F::st(x3c); r.y = F::st(y3c); r.z = F::st(z3b);
  return r;
}

template <class F> DEV Proj<F> pt_neg(const Proj<F>& p) {
  Proj<F> r; r.x = p.x; r.y = F::st(neg(p.y)); r.z = p.z; return r;
}
template <class F> DEV Proj<F> pt_select(bool c, const Proj<F>& a, const Proj<F>& b) {
  Proj<F> r; r.x = select(c, a.x, b.x); r.y = select(c, a.y, b.y); r.z = select(c, a.z, b.z); return r;
}

// ---- extended Jacobian (XYZZ) accumulator for bucket sums -------------------------------------------
// x = X/ZZ, y = Y/ZZZ (ZZ^3 = ZZZ^2).  Mixed addition madd-2008-s costs 8M + 2S (9.6 M-equivalents with
// the dedicated squaring) against the 11M of the complete formula -- the only place it matters is the
// bucket-accumulation kernel, which performs >90% of an MSM's multiplications.  Unlike the RCB formulas
// it has exceptional cases (accumulator at infinity, P = Q, P = -Q); they are detected with a one-limb
// filter and handled exactly, so the group element produced is the same.
template <class F> struct Xyzz { typename F::elem x, y, zz, zzz; };

template <class F, class XT, class YT>
DEV Xyzz<F> xyzz_from_affine(const XT& qx, const YT& qy) {
  Xyzz<F> r; r.x = F::st(qx); r.y = F::st(qy); r.zz = F::one(); r.zzz = F::one(); return r;
}
// 2 * (qx, qy)  (mdbl-2008-s-1, a = 0)
template <class F, class XT, class YT>
DEV Xyzz<F> xyzz_double_affine(const XT& qx, const YT& qy) {
  auto U = norm(dbl(qy));
  auto V = sqr(U);
  auto W = mul(U, V);
  auto S = mul(qx, V);
  auto M = norm(mul_small<3>(sqr(qx)));
  auto X3 = norm(sub(sqr(M), dbl(S)));
  auto Y3 = sub(mul(M, norm(sub(S, X3))), mul(W, qy));
  Xyzz<F> r; r.x = F::st(X3); r.y = F::st(Y3); r.zz = F::st(V); r.zzz = F::st(W); return r;
}
// acc + (qx, qy); `inf` is the accumulator's identity flag (in/out)
template <class F, class XT, class YT>
DEV Xyzz<F> xyzz_add_mixed(const Xyzz<F>& p, bool& inf, const XT& qx, const YT& qy) {
  if (inf) { inf = false; return xyzz_from_affine<F>(qx, qy); }
  auto U2 = mul(qx, p.zz);
  auto S2 = mul(qy, p.zzz);
  auto P = norm(sub(U2, p.x));
  auto R = norm(sub(S2, p.y));
  if (is_zero_fast(P)) {                       // same x: doubling or cancellation (never taken on random input)
    if (is_zero_fast(R)) return xyzz_double_affine<F>(qx, qy);
    inf = true;
    return p;
  }
  auto PP = sqr(P);
  auto PPP = mul(P, PP);
  auto Q = mul(p.x, PP);
  auto X3 = norm(sub(sqr(R), add(PPP, dbl(Q))));
  auto Y3 = sub(mul(R, norm(sub(Q, X3))), mul(p.y, PPP));
  auto ZZ3 = mul(p.zz, PP);
  auto ZZZ3 = mul(p.zzz, PPP);
  Xyzz<F> r; r.x = F::st(X3); r.y = F::st(Y3); r.zz = F::st(ZZ3); r.zzz = F::st(ZZZ3); return r;
}
// G1 variant with the ten field products inlined (one straight-line ~35 KB body, no call overhead)
template <class XT, class YT>
DEV Xyzz<FpPolicy> xyzz_add_mixed_inl(const Xyzz<FpPolicy>& p, bool& inf, const XT& qx, const YT& qy) {
  typedef FpPolicy F;
  if (inf) { inf = false; return xyzz_from_affine<F>(qx, qy); }
  auto U2 = mul_inl(qx, p.zz);
  auto S2 = mul_inl(qy, p.zzz);
  auto P = sub(U2, p.x);          // limbs <= 3 * 2^28: still inside the multiplier's column bound
  auto R = sub(S2, p.y);
  if (is_zero_fast(P)) {
    if (is_zero_fast(R)) return xyzz_double_affine<F>(qx, qy);
    inf = true;
    return p;
  }
  auto PP = sqr_inl(P);
  auto PPP = mul_inl(P, PP);
  auto Q = mul_inl(p.x, PP);
  auto X3 = norm(sub(sqr_inl(R), add(PPP, dbl(Q))));
  auto Y3 = sop2_inl(R, norm(sub(Q, X3)), neg(p.y), PPP);       // R (Q - X3) - Y1 PPP with one reduction
  auto ZZ3 = mul_inl(p.zz, PP);
  auto ZZZ3 = mul_inl(p.zzz, PPP);
  Xyzz<F> r; r.x = F::st(X3); r.y = F::st(Y3); r.zz = F::st(ZZ3); r.zzz = F::st(ZZZ3); return r;
}
// XYZZ -> homogeneous projective (X*ZZZ : Y*ZZ : ZZ*ZZZ); identity -> (0:1:0)
template <class F>
DEV Proj<F> xyzz_to_proj(const Xyzz<F>& p, bool inf) {
  if (inf) return pt_identity<F>();
  Proj<F> r;
  r.x = F::st(mul(p.x, p.zzz)); r.y = F::st(mul(p.y, p.zz)); r.z = F::st(mul(p.zz, p.zzz));
  return r;
}

typedef Aff<FpPolicy> G1Aff;
typedef Proj<FpPolicy> G1Proj;
typedef Aff<Fp2Policy> G2Aff;
typedef Proj<Fp2Policy> G2Proj;

}  // namespace bls
