// pairing.hip.h -- Fp6 / Fp12 towers, Miller loop and final exponentiation.
//
// Reference: /root/reference/src/fp6.rs (mul :200-274, square :277-291, mul_by_1 :113-119, mul_by_01
// :121-136, mul_by_nonresidue :139-150, frobenius_map :154-188, invert :294-312), src/fp12.rs (mul
// :197-214, square :174-185, mul_by_014 :116-128, conjugate :136-141, frobenius_map :145-171, invert
// :187-194) and src/pairings.rs (miller_loop :668-694, ell :696-707, doubling_step :709-738,
// addition_step :740-770, final_exponentiation :48-176 with fp4_square / cyclotomic_square /
// cycolotomic_exp, pairing :607-653).
//
// The Miller loop uses the reference's own line formulas and schedule (bits of BLS_X >> 1, 63 doubling
// and 5 addition steps, conjugate at the end), so the raw Miller value is the same Fp12 element the
// reference's `multi_miller_loop` produces; the final exponentiation follows the reference's chain and
// therefore raises to 3 (p^4 - p^2 + 1) / r exactly as it does.  Field products are computed with
// different (Karatsuba / sum-of-products) groupings than the reference's, which changes nothing: every
// value is an exact field element and is serialised only after full reduction.
//
// The loop schedule is a compile-time constant, so all 64 lanes of a wavefront stay converged.  The code is
// generic over the Fp2 element type E:
//   E = fe2          one pairing per lane (Fp2 = two Fe in one lane; Fp12 = 168 registers),
//   E = fp2p         one pairing per PAIR of lanes (pairlane.hip.h: lane 2k holds every c0 coefficient, lane 2k+1
//                    every c1; Fp12 = 84 registers per lane), the form the kernels use.
// Tower elements live in the storage form (limbs normalised, value < 32p); the out-of-line helpers take them by
// reference, i.e. operands are staged in per-lane scratch and streamed through the VGPRs.
#pragma once
#include "convert.hip.h"
#include "scalar.hip.h"
#include "limits.h"
#include "pairlane.hip.h"

namespace bls {

template <class E> struct Fp6T { E c0, c1, c2; };
template <class E> struct Fp12T { Fp6T<E> c0, c1; };
#ifndef BLS_VSP
#define BLS_VSP 64
#endif
constexpr int VSP = BLS_VSP;          // value bound of a stored pair-lane element (sums of two still fit the product's operand bound)
typedef FeP<1, VSP> fp2p;             // working Fp2 value over a lane pair

// ---- element traits: what differs between the one-lane and the pair-lane Fp2 ------------------------------
template <class E> struct E2;
template <> struct E2<fe2> {
  static constexpr int LANES = 1;                 // lanes per tower element
  static DEV fe2 zero() { return fe2_zero(); }
  static DEV fe2 one() { return fe2_one(); }
  static DEV fe2 konst(const PLimbs& k0, const PLimbs& k1) { fe2 K; K.c0 = (Fe<1, VS2>)fe1_const(k0); K.c1 = (Fe<1, VS2>)fe1_const(k1); return K; }
  static DEV fe2 load(const u32* w) { return (fe2)fe2_from_ref(w); }
  static DEV void save(const fe2& a, u32* w) { fe2_to_ref(a, w); }
};
template <> struct E2<fp2p> {
  static constexpr int LANES = 2;
  static DEV fp2p zero() { fp2p r; r.v = (Fe<1, VSP>)fe_zero(); return r; }
  static DEV fp2p one() { fp2p r; r.v = select(lane_is_c1(), (Fe<1, VSP>)fe_zero(), (Fe<1, VSP>)fe_one()); return r; }
  static DEV fp2p konst(const PLimbs& k0, const PLimbs& k1) { fp2p K; K.v = select(lane_is_c1(), (Fe<1, VSP>)fe1_const(k1), (Fe<1, VSP>)fe1_const(k0)); return K; }
  static DEV fp2p load(const u32* w) { fp2p r; r.v = (Fe<1, VSP>)fe_from_ref(w + (lane_is_c1() ? 12 : 0)); return r; }
  static DEV void save(const fp2p& a, u32* w) { fe_to_ref(a.v, w + (lane_is_c1() ? 12 : 0)); }
};
template <int A, int V> DEV fe2 st2(const Fe2<A, V>& a) { return store2(a); }
template <int A, int V> DEV fp2p st2(const FeP<A, V>& a) {
  fp2p r;
  if constexpr (V <= VSP) r.v = norm(a.v); else r.v = reduce_v(norm(a.v));
  return r;
}
template <int A1, int V1, int A2, int V2> DEV auto pmul(const Fe2<A1, V1>& a, const Fe2<A2, V2>& b) { return mul(a, b); }
template <int A1, int V1, int A2, int V2> DEV auto pmul(const FeP<A1, V1>& a, const FeP<A2, V2>& b) { return mul_ni(a, b); }
template <int A, int V> DEV auto psqr(const Fe2<A, V>& a) { return sqr(a); }
template <int A, int V> DEV auto psqr(const FeP<A, V>& a) { return sqr_ni(a); }
// (a0 + a1 u) u = -a1 + a0 u

#ifndef BLS_PAIRING_BLOCK
#define BLS_PAIRING_BLOCK 256
#endif
#ifndef BLS_PAIRING_WAVES
#define BLS_PAIRING_WAVES 2
#endif
constexpr int PAIRING_BLOCK = BLS_PAIRING_BLOCK;       // two lanes per pairing
constexpr int PAIRING_WAVES = BLS_PAIRING_WAVES;       // wavefronts per SIMD the register budget is set for
constexpr int FP12_PROD_FAN = 8;                      // widest fan of the product tree (levels that fill the chip); narrow levels use 2

#define S2(x) st2(x)
// Inlining policy.  The hot loops (Miller loop, the run of compressed cyclotomic squarings) keep their state in NON-ESCAPING
// locals and inline the tower glue around the out-of-line Fp2 products, so the register allocator owns f, R and the line (a
// by-reference call pins them to per-lane scratch: profiles/r01_pairing_pmc.md measured ~450 KB of scratch traffic per
// pairing that way).  Code executed a few times per pairing (final exponentiation outside the squaring runs, inversions,
// Frobenius maps) stays out of line and by reference so that the kernels fit the instruction cache.
#ifdef BLS_TOWER_OUTLINE
#define HOT DEVNI
#else
#define HOT DEV
#endif

template <class E> DEV Fp6T<E> fp6_zero() { Fp6T<E> r; r.c0 = E2<E>::zero(); r.c1 = E2<E>::zero(); r.c2 = E2<E>::zero(); return r; }
template <class E> DEV Fp6T<E> fp6_one() { Fp6T<E> r; r.c0 = E2<E>::one(); r.c1 = E2<E>::zero(); r.c2 = E2<E>::zero(); return r; }
template <class E> DEV Fp12T<E> fp12_one() { Fp12T<E> r; r.c0 = fp6_one<E>(); r.c1 = fp6_zero<E>(); return r; }

template <class E> DEV Fp6T<E> fp6_add(const Fp6T<E>& a, const Fp6T<E>& b) {
  Fp6T<E> r; r.c0 = S2(add(a.c0, b.c0)); r.c1 = S2(add(a.c1, b.c1)); r.c2 = S2(add(a.c2, b.c2)); return r;
}
template <class E> DEV Fp6T<E> fp6_sub(const Fp6T<E>& a, const Fp6T<E>& b) {
  Fp6T<E> r; r.c0 = S2(sub(a.c0, b.c0)); r.c1 = S2(sub(a.c1, b.c1)); r.c2 = S2(sub(a.c2, b.c2)); return r;
}
template <class E> DEV Fp6T<E> fp6_neg(const Fp6T<E>& a) { Fp6T<E> r; r.c0 = S2(neg(a.c0)); r.c1 = S2(neg(a.c1)); r.c2 = S2(neg(a.c2)); return r; }
// multiply by v  (fp6.rs:139-150)
template <class E> DEV Fp6T<E> fp6_mul_by_nonresidue(const Fp6T<E>& a) {
  Fp6T<E> r; r.c0 = S2(mul_by_nonresidue(a.c2)); r.c1 = a.c0; r.c2 = a.c1; return r;
}

// Karatsuba over Fp2 (6 Fp2 products); same element as fp6.rs:200-274
template <class E> DEV void fp6_mul(Fp6T<E>& r, const Fp6T<E>& a, const Fp6T<E>& b) {
  auto v0 = pmul(a.c0, b.c0);
  auto v1 = pmul(a.c1, b.c1);
  auto v2 = pmul(a.c2, b.c2);
  auto t12 = pmul(add(a.c1, a.c2), add(b.c1, b.c2));
  auto t01 = pmul(add(a.c0, a.c1), add(b.c0, b.c1));
  auto t02 = pmul(add(a.c0, a.c2), add(b.c0, b.c2));
  auto x12 = norm(sub(sub(t12, v1), v2));                 // a1 b2 + a2 b1
  auto c0 = add(v0, mul_by_nonresidue(x12));
  auto x01 = norm(sub(sub(t01, v0), v1));                 // a0 b1 + a1 b0
  auto c1 = add(x01, mul_by_nonresidue(v2));
  auto x02 = norm(sub(sub(t02, v0), v2));                 // a0 b2 + a2 b0
  auto c2 = add(x02, v1);
  r.c0 = S2(c0); r.c1 = S2(c1); r.c2 = S2(c2);
}
// fp6.rs:277-291
template <class E> DEV void fp6_sqr_inl(Fp6T<E>& r, const Fp6T<E>& a) {
  auto s0 = psqr(a.c0);
  auto ab = pmul(a.c0, a.c1);
  auto s1 = dbl(ab);
  auto s2 = psqr(norm(add(sub(a.c0, a.c1), a.c2)));
  auto bc = pmul(a.c1, a.c2);
  auto s3 = dbl(bc);
  auto s4 = psqr(a.c2);
  auto c0 = add(mul_by_nonresidue(norm(s3)), s0);
  auto c1 = add(mul_by_nonresidue(s4), s1);
  auto c2 = sub(sub(norm(add(add(s1, s2), s3)), s0), s4);
  r.c0 = S2(c0); r.c1 = S2(c1); r.c2 = S2(c2);
}
template <class E> DEVNI void fp6_sqr(Fp6T<E>& r, const Fp6T<E>& a) { fp6_sqr_inl(r, a); }
// fp6.rs:113-119
template <class E> DEV void fp6_mul_by_1(Fp6T<E>& r, const Fp6T<E>& a, const E& c1) {
  auto t0 = pmul(a.c2, c1);
  auto t1 = pmul(a.c0, c1);
  auto t2 = pmul(a.c1, c1);
  r.c0 = S2(mul_by_nonresidue(t0)); r.c1 = S2(t1); r.c2 = S2(t2);
}
// fp6.rs:121-136
template <class E> DEV void fp6_mul_by_01(Fp6T<E>& r, const Fp6T<E>& a, const E& c0, const E& c1) {
  auto a_a = pmul(a.c0, c0);
  auto b_b = pmul(a.c1, c1);
  auto t1 = add(mul_by_nonresidue(pmul(a.c2, c1)), a_a);
  auto t2 = sub(sub(pmul(add(c0, c1), add(a.c0, a.c1)), a_a), b_b);
  auto t3 = add(pmul(a.c2, c0), b_b);
  r.c0 = S2(t1); r.c1 = S2(t2); r.c2 = S2(t3);
}
// fp6.rs:154-188
template <class E> DEVNI void fp6_frobenius(Fp6T<E>& r, const Fp6T<E>& a) {
  constexpr PLimbs k1 = {BLS_FROB6_C1_1}, k2 = {BLS_FROB6_C2_0};
  auto c0 = conj(a.c0);
  auto c1 = norm(conj(a.c1));
  auto c2 = norm(conj(a.c2));
  // c1 * (0 + k1 u) = (c1 u) k1
  fe1 K1 = fe1_const(k1), K2 = fe1_const(k2);
  auto m1 = mul_fp(norm(mul_by_u(c1)), K1);
  auto m2 = mul_fp(c2, K2);
  r.c0 = S2(c0); r.c1 = S2(m1); r.c2 = S2(m2);
}
// fp6.rs:294-312
template <class E> DEVNI void fp6_inv(Fp6T<E>& r, const Fp6T<E>& a) {
  auto c0 = norm(sub(psqr(a.c0), mul_by_nonresidue(pmul(a.c1, a.c2))));
  auto c1 = norm(sub(mul_by_nonresidue(psqr(a.c2)), pmul(a.c0, a.c1)));
  auto c2 = norm(sub(psqr(a.c1), pmul(a.c0, a.c2)));
  auto t = norm(add(mul_by_nonresidue(norm(add(pmul(a.c1, c2), pmul(a.c2, c1)))), pmul(a.c0, c0)));
  auto ti = inv(t);
  r.c0 = S2(pmul(ti, c0)); r.c1 = S2(pmul(ti, c1)); r.c2 = S2(pmul(ti, c2));
}

// ---- Fp12T<E> --------------------------------------------------------------------------------------------
// fp12.rs:197-214
template <class E> DEVNI void fp12_mul(Fp12T<E>& r, const Fp12T<E>& a, const Fp12T<E>& b) {
  Fp6T<E> aa, bb, t;
  fp6_mul(aa, a.c0, b.c0);
  fp6_mul(bb, a.c1, b.c1);
  Fp6T<E> o = fp6_add(b.c0, b.c1);
  Fp6T<E> s = fp6_add(a.c1, a.c0);
  fp6_mul(t, s, o);
  r.c1 = fp6_sub(fp6_sub(t, aa), bb);
  r.c0 = fp6_add(fp6_mul_by_nonresidue(bb), aa);
}
// fp12.rs:174-185
#ifndef BLS_SQR_COMPLEX
// The same field element as fp12.rs:174-185 by three Fp6 SQUARINGS instead of the reference's two Fp6 products:
//   (c0 + c1 w)^2 = (c0^2 + v c1^2) + ((c0 + c1)^2 - c0^2 - c1^2) w.
// In the lane-pair form both cost 7056 multiply-adds per lane (9 Fp2 squarings at one Fp product each + 6 Fp2 products at
// two), but a squaring has ONE Fp6 operand: the live set while the accumulator of the Miller loop is squared is ~225
// registers instead of ~320, i.e. far fewer spills to per-lane scratch.
template <class E> HOT void fp12_sqr_hot(Fp12T<E>& r, const Fp12T<E>& a) {
  Fp6T<E> s = fp6_add(a.c0, a.c1);
  Fp6T<E> v0, v1, t;
  fp6_sqr_inl(v0, a.c0);
  fp6_sqr_inl(v1, a.c1);
  fp6_sqr_inl(t, s);
  r.c1 = fp6_sub(fp6_sub(t, v0), v1);
  r.c0 = fp6_add(fp6_mul_by_nonresidue(v1), v0);
}
#else
template <class E> HOT void fp12_sqr_hot(Fp12T<E>& r, const Fp12T<E>& a) {
  Fp6T<E> ab, t;
  fp6_mul(ab, a.c0, a.c1);
  Fp6T<E> c0c1 = fp6_add(a.c0, a.c1);
  Fp6T<E> c0 = fp6_add(fp6_mul_by_nonresidue(a.c1), a.c0);
  fp6_mul(t, c0, c0c1);
  t = fp6_sub(t, ab);
  r.c1 = fp6_add(ab, ab);
  r.c0 = fp6_sub(t, fp6_mul_by_nonresidue(ab));
}
#endif
template <class E> DEVNI void fp12_sqr(Fp12T<E>& r, const Fp12T<E>& a) { fp12_sqr_hot(r, a); }
// fp12.rs:116-128
template <class E> HOT void fp12_mul_by_014(Fp12T<E>& r, const Fp12T<E>& a, const E& c0, const E& c1, const E& c4) {
  Fp6T<E> aa, bb, t;
  fp6_mul_by_01(aa, a.c0, c0, c1);
  fp6_mul_by_1(bb, a.c1, c4);
  E o = S2(add(c1, c4));
  Fp6T<E> s = fp6_add(a.c1, a.c0);
  fp6_mul_by_01(t, s, c0, o);
  r.c1 = fp6_sub(fp6_sub(t, aa), bb);
  r.c0 = fp6_add(fp6_mul_by_nonresidue(bb), aa);
}
template <class E> DEV void fp12_conj(Fp12T<E>& r, const Fp12T<E>& a) { Fp6T<E> n = fp6_neg(a.c1); r.c0 = a.c0; r.c1 = n; }
// fp12.rs:145-171
template <class E> DEVNI void fp12_frobenius(Fp12T<E>& r, const Fp12T<E>& a) {
  constexpr PLimbs k0 = {BLS_FROB12_C1_0}, k1 = {BLS_FROB12_C1_1};
  Fp6T<E> c0, c1;
  fp6_frobenius(c0, a.c0);
  fp6_frobenius(c1, a.c1);
  E K = E2<E>::konst(k0, k1);
  r.c0 = c0;
  r.c1.c0 = S2(pmul(c1.c0, K)); r.c1.c1 = S2(pmul(c1.c1, K)); r.c1.c2 = S2(pmul(c1.c2, K));
}
// fp12.rs:187-194
template <class E> DEVNI void fp12_inv(Fp12T<E>& r, const Fp12T<E>& a) {
  Fp6T<E> s0, s1, t, ti;
  fp6_sqr(s0, a.c0);
  fp6_sqr(s1, a.c1);
  t = fp6_sub(s0, fp6_mul_by_nonresidue(s1));
  fp6_inv(ti, t);
  fp6_mul(r.c0, a.c0, ti);
  Fp6T<E> nt = fp6_neg(ti);
  fp6_mul(r.c1, a.c1, nt);
}

// ---- wire I/O ------------------------------------------------------------------------------------------
template <class E> DEV void fp12_load(Fp12T<E>& f, const u32* w) {
  E* e[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int i = 0; i < 6; i++) *e[i] = E2<E>::load(w + 24 * i);
}
template <class E> DEV void fp12_save(const Fp12T<E>& f, u32* w) {
  const E* e[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
#pragma unroll
  for (int i = 0; i < 6; i++) E2<E>::save(*e[i], w + 24 * i);
}

// ---- Miller loop -------------------------------------------------------------------------------------
template <class E> struct G2JacT { E x, y, z; };       // the pairing's running point R (pairings.rs:709-770 operate on G2Projective fields)
template <class E> struct LineT { E a, b, c; };        // the (Fp2, Fp2, Fp2) coefficient triple

// pairings.rs:709-738 (CLN Algorithm 26)
template <class E> HOT void doubling_step(G2JacT<E>& r, LineT<E>& l) {
  auto tmp0 = psqr(r.x);
  auto tmp1 = psqr(r.y);
  auto tmp2 = psqr(tmp1);
  auto tmp3 = norm(sub(sub(psqr(add(tmp1, r.x)), tmp0), tmp2));
  auto tmp3d = norm(dbl(tmp3));
  auto tmp4 = norm(add(dbl(tmp0), tmp0));
  auto tmp6 = add(r.x, tmp4);
  auto tmp5 = psqr(tmp4);
  auto zsq = psqr(r.z);
  auto rx = norm(sub(sub(tmp5, tmp3d), tmp3d));
  auto rz = sub(sub(psqr(add(r.z, r.y)), tmp1), zsq);
  auto ry = pmul(norm(sub(tmp3d, rx)), tmp4);
  auto tmp2o = norm(mul_small<8>(tmp2));
  auto ryo = sub(ry, tmp2o);
  auto t3 = pmul(tmp4, zsq);
  auto t3n = neg(norm(dbl(t3)));
  auto t6 = sub(sub(psqr(norm(tmp6)), tmp0), tmp5);
  auto t1q = norm(mul_small<4>(tmp1));
  auto t6o = sub(norm(t6), t1q);
  E rzs = S2(rz);
  auto t0 = pmul(rzs, zsq);
  r.x = S2(rx); r.y = S2(ryo); r.z = rzs;
  l.a = S2(dbl(t0)); l.b = S2(t3n); l.c = S2(t6o);
}
// pairings.rs:740-770 (CLN Algorithm 27)
template <class E> HOT void addition_step(G2JacT<E>& r, const E& qx, const E& qy, LineT<E>& l) {
  auto zsq = psqr(r.z);
  auto ysq = psqr(qy);
  auto t0 = pmul(zsq, qx);
  auto t1 = pmul(norm(sub(sub(psqr(add(qy, r.z)), ysq), zsq)), zsq);
  auto t2 = norm(sub(t0, r.x));
  auto t3 = psqr(t2);
  auto t4 = norm(mul_small<4>(t3));
  auto t5 = pmul(t4, t2);
  auto t6 = norm(sub(sub(t1, r.y), r.y));
  auto t9 = pmul(t6, qx);
  auto t7 = pmul(t4, r.x);
  auto rx = norm(sub(sub(sub(psqr(t6), t5), t7), t7));
  auto rz = sub(sub(psqr(add(r.z, t2)), zsq), t3);
  E rzs = S2(rz);
  auto t10 = add(qy, rzs);
  auto t8 = pmul(norm(sub(t7, rx)), t6);
  auto t0b = pmul(r.y, t5);
  auto ry = sub(t8, norm(dbl(t0b)));
  auto t10b = sub(psqr(t10), ysq);
  auto ztsq = psqr(rzs);
  auto t10c = sub(norm(t10b), ztsq);
  auto t9b = sub(norm(dbl(t9)), norm(t10c));
  auto t10d = dbl(rzs);
  auto t6n = neg(t6);
  auto t1b = dbl(norm(t6n));
  r.x = S2(rx); r.y = S2(ry); r.z = rzs;
  l.a = S2(t10d); l.b = S2(t1b); l.c = S2(t9b);
}
// pairings.rs:696-707
template <class E> HOT void ell(Fp12T<E>& f, const LineT<E>& l, const fe1& px, const fe1& py) {
  E c0 = S2(mul_fp(l.a, py));
  E c1 = S2(mul_fp(l.b, px));
  fp12_mul_by_014(f, f, l.c, c1, c0);          // in place: every read of the input precedes the first write
}

// bits of BLS_X >> 1 below the leading one, MSB first (pairings.rs:671-685): 62 iterations, 5 set bits
constexpr unsigned long long X_HALF = 0xd201000000010000ull >> 1;

template <class E> DEVNI void addition_step_ell(Fp12T<E>& f, G2JacT<E>& r, const fe1& px, const fe1& py, const E& qx, const E& qy) {
  LineT<E> l;
  addition_step(r, qx, qy, l);
  ell(f, l, px, py);
}
// LDS parking lot: while the accumulator f (84 registers per lane) is being squared or multiplied by a line, the running point
// R (42) and P (28) are not needed, and vice versa -- but the register allocator sees them all live across the whole loop and
// spills whatever does not fit to per-lane scratch, whose hot footprint (512 lanes x ~2 KB per CU) misses the 4 MB L2.  The 70
// words are parked in LDS instead (160 KB / 8 wavefronts = 320 B per lane; word w of lane t at [w * PAIRING_BLOCK + t]:
// conflict-free), which takes them out of the allocator's hands.  The loop is inlined into its single caller, which owns the
// LDS array, so `park` is an LDS address by construction (ds_read / ds_write, not flat).
constexpr int PARK_WORDS = 5 * NL;          // R.x R.y R.z (one coefficient per lane of the pair) + px py
template <int V> DEV void park_put(u32* park, int slot, const Fe<1, V>& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) park[(slot * NL + i) * PAIRING_BLOCK] = a.l[i];
}
template <int V> DEV void park_get(const u32* park, int slot, Fe<1, V>& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) a.l[i] = park[(slot * NL + i) * PAIRING_BLOCK];
}
template <class E> DEV void miller_loop(Fp12T<E>& fout, const fe1& px_, const fe1& py_, const E& qx_, const E& qy_, u32* park) {
  {
    const E one = E2<E>::one();
    park_put(park, 0, qx_.v); park_put(park, 1, qy_.v); park_put(park, 2, one.v);
    park_put(park, 3, px_); park_put(park, 4, py_);
  }
  Fp12T<E> f = fp12_one<E>();               // never escapes: the register allocator owns it
  for (int b = 61; b >= -1; b--) {          // bit 62 is the leading one; b = -1: the final doubling step (pairings.rs:686-687)
    LineT<E> l;
    {
      G2JacT<E> r;
      park_get(park, 0, r.x.v); park_get(park, 1, r.y.v); park_get(park, 2, r.z.v);
      doubling_step(r, l);
      park_put(park, 0, r.x.v); park_put(park, 1, r.y.v); park_put(park, 2, r.z.v);
    }
    {
      fe1 px, py;
      park_get(park, 3, px); park_get(park, 4, py);
      ell(f, l, px, py);
    }
    if (b < 0) break;
    if ((X_HALF >> b) & 1) {
      // 5 of the 62 iterations: out of line (keeps the hot loop inside the instruction cache); the copies confine the
      // address-taken objects to this branch
      Fp12T<E> ft = f; G2JacT<E> rt;
      park_get(park, 0, rt.x.v); park_get(park, 1, rt.y.v); park_get(park, 2, rt.z.v);
      addition_step_ell(ft, rt, px_, py_, qx_, qy_);
      f = ft;
      park_put(park, 0, rt.x.v); park_put(park, 1, rt.y.v); park_put(park, 2, rt.z.v);
    }
    fp12_sqr_hot(f, f);                     // in place
  }
  f.c1 = fp6_neg(f.c1);                     // conjugate: BLS_X_IS_NEGATIVE
  fout = f;
}

// pairings.rs:554-603 as the reference schedules it: ONE accumulator for K terms -- per bit every term contributes its
// line(s) and the accumulator is squared once, instead of K accumulators each paying the 62 squarings.  Same element
// as the product of the K separate Miller values (f <- f^2 * prod l_k = prod (f_k^2 l_k)).  Identity terms are skipped
// (:566-569).  Per-term state (P, Q in internal form and the running point R) lives in per-lane scratch.
template <class E> struct MmlTerm { fe1 px, py; E qx, qy; G2JacT<E> r; bool skip; };
template <class E> DEVNI void multi_miller_shared(Fp12T<E>& fout, MmlTerm<E>* t, int K) {
  Fp12T<E> f = fp12_one<E>();               // non-escaping, like miller_loop's
  for (int b = 61; b >= -1; b--) {          // b = -1: the final doubling step
    for (int k = 0; k < K; k++) {
      fair_tick(1);
      if (t[k].skip) continue;
      G2JacT<E> r = t[k].r;
      const fe1 px = t[k].px, py = t[k].py;
      LineT<E> l;
      doubling_step(r, l);
      fair_tick(1);
      ell(f, l, px, py);
      t[k].r = r;
      if (b >= 0 && ((X_HALF >> b) & 1)) {
        Fp12T<E> ft = f;
        addition_step_ell(ft, t[k].r, t[k].px, t[k].py, t[k].qx, t[k].qy);
        f = ft;
      }
    }
    if (b < 0) break;
    fair_tick(1);
    fp12_sqr_hot(f, f);
  }
  f.c1 = fp6_neg(f.c1);
  fout = f;
}

// ---- final exponentiation ------------------------------------------------------------------------------
// pairings.rs:50-62
template <class E> DEV void fp4_square(E& c0, E& c1, const E& a, const E& b) {
  auto t0 = psqr(a);
  auto t1 = psqr(b);
  auto t2 = mul_by_nonresidue(t1);
  c0 = S2(add(t2, t0));
  auto t3 = psqr(add(a, b));
  c1 = S2(sub(sub(t3, t0), t1));
}
// pairings.rs:66-112
template <class E> DEVNI void cyclotomic_square(Fp12T<E>& r, const Fp12T<E>& f) {
  E z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  E t0, t1, t2, t3;
  fp4_square(t0, t1, z0, z1);
  E nz0 = S2(add(dbl(norm(sub(t0, z0))), t0));
  E nz1 = S2(add(dbl(norm(add(t1, z1))), t1));
  fp4_square(t0, t1, z2, z3);
  fp4_square(t2, t3, z4, z5);
  E nz4 = S2(add(dbl(norm(sub(t0, z4))), t0));
  E nz5 = S2(add(dbl(norm(add(t1, z5))), t1));
  E t0b = S2(mul_by_nonresidue(t3));
  E nz2 = S2(add(dbl(norm(add(t0b, z2))), t0b));
  E nz3 = S2(add(dbl(norm(sub(t2, z3))), t2));
  r.c0.c0 = nz0; r.c0.c1 = nz4; r.c0.c2 = nz3;
  r.c1.c0 = nz2; r.c1.c1 = nz1; r.c1.c2 = nz5;
}
// pairings.rs:114-132 (`cycolotomic_exp`): f^|x| then conjugate -- the reference's square-and-multiply schedule
template <class E> DEVNI void cyclotomic_exp_plain(Fp12T<E>& r, const Fp12T<E>& f) {
  constexpr unsigned long long X = 0xd201000000010000ull;
  Fp12T<E> tmp = f;                       // the leading one: tmp = one * f
  for (int b = 62; b >= 0; b--) {
    cyclotomic_square(tmp, tmp);      // in place (inputs are copied to locals first)
    if ((X >> b) & 1) fp12_mul(tmp, tmp, f);
  }
  fp12_conj(r, tmp);
}
// The same power with Karabina's compressed squarings (eprint 2010/542).  In the notation of cyclotomic_square an element
// of the cyclotomic subgroup is (A, B, C) = ((z0,z1), (z2,z3), (z4,z5)) over Fp4 = Fp2[s]/(s^2 - xi), and the squaring
// rule B' = 3 s C^2 + 2 conj(B), C' = 3 B^2 - 2 conj(C) does not involve A: a run of squarings can carry (B, C) only
// (two Fp4 squarings instead of three) and recover A where a power is needed,
//     z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2)      [z2 = 0:  z1 = 2 z4 z5 / z3]         z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1,
// with ONE shared inversion for the six powers f^(2^i), i in {16, 48, 57, 60, 62, 63}, whose product is f^|x|.  Both
// routes compute the same field element, so the result is bit-identical; `false` is returned (nothing written) in the
// degenerate case z2 = z3 = 0 (e.g. f = 1), which the caller sends down the plain route.
template <class E> struct CycC { E z2, z3, z4, z5; };
template <class E> HOT void cyclotomic_square_compressed(CycC<E>& r, const CycC<E>& c) {
  E z2 = c.z2, z3 = c.z3, z4 = c.z4, z5 = c.z5;
  E t0, t1, t2, t3;
  fp4_square(t0, t1, z2, z3);
  fp4_square(t2, t3, z4, z5);
  E nz4 = S2(add(dbl(norm(sub(t0, z4))), t0));
  E nz5 = S2(add(dbl(norm(add(t1, z5))), t1));
  E t0b = S2(mul_by_nonresidue(t3));
  E nz2 = S2(add(dbl(norm(add(t0b, z2))), t0b));
  E nz3 = S2(add(dbl(norm(sub(t2, z3))), t2));
  r.z2 = nz2; r.z3 = nz3; r.z4 = nz4; r.z5 = nz5;
}
constexpr int CYC_NSNAP = 6;
template <class E> DEVNI bool cyclotomic_exp_compressed(Fp12T<E>& r, const Fp12T<E>& f) {
  constexpr unsigned long long X = 0xd201000000010000ull;
  CycC<E> c; c.z2 = f.c1.c0; c.z3 = f.c0.c2; c.z4 = f.c0.c1; c.z5 = f.c1.c2;      // never escapes: lives in registers across the run
  CycC<E> snap[CYC_NSNAP];
  int ns = 0;
  for (int i = 1; i <= 63; i++) {
    CycC<E> n;
    cyclotomic_square_compressed(n, c);
    c = n;
    if ((X >> i) & 1) snap[ns++] = c;
  }
  // denominators and their shared inverse (Montgomery's trick)
  E den[CYC_NSNAP], pre[CYC_NSNAP];
  bool z2zero[CYC_NSNAP];
  bool ok = true;
  for (int j = 0; j < CYC_NSNAP; j++) {
    z2zero[j] = is_zero_fast(snap[j].z2);
    if (z2zero[j]) { den[j] = snap[j].z3; ok = ok && !is_zero_fast(snap[j].z3); }
    else den[j] = S2(mul_small<4>(snap[j].z2));
  }
  if (!ok) return false;
  pre[0] = den[0];
  for (int j = 1; j < CYC_NSNAP; j++) pre[j] = S2(pmul(pre[j - 1], den[j]));
  E run = S2(inv(pre[CYC_NSNAP - 1]));
  Fp12T<E> acc;
  for (int j = CYC_NSNAP - 1; j >= 0; j--) {
    E dinv = j ? S2(pmul(run, pre[j - 1])) : run;
    if (j) run = S2(pmul(run, den[j]));
    const CycC<E>& q = snap[j];
    E num;
    if (z2zero[j]) num = S2(dbl(pmul(q.z4, q.z5)));
    else num = S2(sub(add(mul_by_nonresidue(psqr(q.z5)), mul_small<3>(psqr(q.z4))), dbl(q.z3)));
    E z1 = S2(pmul(num, dinv));
    E w = S2(sub(add(dbl(psqr(z1)), pmul(q.z2, q.z5)), mul_small<3>(pmul(q.z3, q.z4))));
    E z0 = S2(add(mul_by_nonresidue(w), E2<E>::one()));
    Fp12T<E> g;
    g.c0.c0 = z0; g.c0.c1 = q.z4; g.c0.c2 = q.z3;
    g.c1.c0 = q.z2; g.c1.c1 = z1; g.c1.c2 = q.z5;
    if (j == CYC_NSNAP - 1) acc = g; else fp12_mul(acc, acc, g);
  }
  fp12_conj(r, acc);
  return true;
}
template <class E> DEV void cyclotomic_exp(Fp12T<E>& r, const Fp12T<E>& f) {
  if (!cyclotomic_exp_compressed(r, f)) cyclotomic_exp_plain(r, f);
}
// pairings.rs:134-173
template <class E> DEVNI void final_exponentiation(Fp12T<E>& out, const Fp12T<E>& fin) {
  // every Fp12T<E> helper tolerates r aliasing an input (inputs are consumed before the first write)
  Fp12T<E> t0, t1, t2, t3, t4, t5, t6;
  fp12_frobenius(t0, fin);
  for (int i = 0; i < 5; i++) fp12_frobenius(t0, t0);
  fp12_inv(t1, fin);
  fp12_mul(t2, t0, t1);
  t1 = t2;
  fp12_frobenius(t2, t2); fp12_frobenius(t2, t2);
  fp12_mul(t2, t2, t1);
  cyclotomic_square(t1, t2); fp12_conj(t1, t1);
  cyclotomic_exp(t3, t2);
  cyclotomic_square(t4, t3);
  fp12_mul(t5, t1, t3);
  cyclotomic_exp(t1, t5);
  cyclotomic_exp(t0, t1);
  cyclotomic_exp(t6, t0);
  fp12_mul(t6, t6, t4);
  cyclotomic_exp(t4, t6);
  fp12_conj(t5, t5);
  fp12_mul(t5, t5, t2);            // t5 * t2 (t5 is overwritten next anyway)
  fp12_mul(t4, t4, t5);
  fp12_conj(t5, t2);
  fp12_mul(t1, t1, t2);
  fp12_frobenius(t1, t1); fp12_frobenius(t1, t1); fp12_frobenius(t1, t1);
  fp12_mul(t6, t6, t5);
  fp12_frobenius(t6, t6);
  fp12_mul(t3, t3, t0);
  fp12_frobenius(t3, t3); fp12_frobenius(t3, t3);
  fp12_mul(t3, t3, t1);
  fp12_mul(t3, t3, t6);
  fp12_mul(out, t3, t4);
}

// ---- kernels ---------------------------------------------------------------------------------------------
// Every kernel works on lane pairs: pairing / product chain i lives on lanes 2i and 2i+1 of the grid.
typedef fp2p PE;
constexpr int PL = E2<PE>::LANES;
#define PAIR_KERNEL __global__ void __launch_bounds__(PAIRING_BLOCK, PAIRING_WAVES)

// mode 0: out[i] = pairing(g1[i], g2[i]);  mode 1: out[i] = raw Miller loop value.
// Identity on either side -> Fp12::one() (pairings.rs:636-651; multi_miller_loop skips such terms :566-569).
PAIR_KERNEL k_pairing(int mode, const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2,
                      const uint8_t* __restrict__ g2inf, u32* __restrict__ out, size_t n) {
  __shared__ u32 park_lds[PARK_WORDS * PAIRING_BLOCK];          // 70 KB per 256-thread workgroup: see miller_loop
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (i >= n) return;
  bool ident = (g1inf && g1inf[i]) || (g2inf && g2inf[i]);
  Fp12T<PE> f;
  if (ident) {
    f = fp12_one<PE>();
  } else {
    fe1 px = fe_from_ref(g1 + i * 24), py = fe_from_ref(g1 + i * 24 + 12);
    PE qx = E2<PE>::load(g2 + i * 48), qy = E2<PE>::load(g2 + i * 48 + 24);
    miller_loop(f, px, py, qx, qy, park_lds + threadIdx.x);
    if (mode == 0) { Fp12T<PE> g; final_exponentiation(g, f); f = g; }
  }
  fp12_save(f, out + i * 144);
}
// out[j] = Miller value of terms [j*K, (j+1)*K) with shared squarings (the partial products are then multiplied up)
PAIR_KERNEL k_multi_miller_shared(const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2,
                                  const uint8_t* __restrict__ g2inf, u32* __restrict__ out, size_t n, int K) {
  fair_init();
  size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  const size_t groups = (n + K - 1) / K;
  if (j >= groups) return;
  MmlTerm<PE> t[MML_MAX_K];
  for (int k = 0; k < K; k++) {
    const size_t i = j * K + k;
    t[k].skip = i >= n || (g1inf && g1inf[i]) || (g2inf && g2inf[i]);
    if (t[k].skip) continue;
    t[k].px = fe_from_ref(g1 + i * 24); t[k].py = fe_from_ref(g1 + i * 24 + 12);
    t[k].qx = E2<PE>::load(g2 + i * 48); t[k].qy = E2<PE>::load(g2 + i * 48 + 24);
    t[k].r.x = t[k].qx; t[k].r.y = t[k].qy; t[k].r.z = E2<PE>::one();
  }
  Fp12T<PE> f;
  multi_miller_shared(f, t, K);
  fp12_save(f, out + j * 144);
}
// out[s] = multi_miller_loop(terms off[s] .. off[s + 1]) with ONE shared accumulator per segment (the reference's own schedule,
// pairings.rs:554-603: per bit every term adds its line, one squaring for all): segments of at most MML_MAX_K terms (longer ones are
// clamped: the host dispatches here only when its bound holds), empty segments give one.  Used by blsgpu_multi_miller_loop_many for
// MANY short segments, where a lane pair per segment fills the chip and (k - 1) / k of the 62 squarings per term disappear.
PAIR_KERNEL k_multi_miller_seg(const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2, const uint8_t* __restrict__ g2inf,
                               const unsigned long long* __restrict__ off, size_t nseg, size_t total, u32* __restrict__ out, u32* __restrict__ status) {
  fair_init();
  size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (j >= nseg) return;
  size_t beg = (size_t)off[j], end = (size_t)off[j + 1];
  if (end > total) end = total;
  if (beg > end) beg = end;
  if (end - beg > (size_t)MML_MAX_K) atomicOr(status, 2u);          // the caller's bound on the segment length does not hold: reported, that segment's value is unspecified
  int K = (int)(end - beg < (size_t)MML_MAX_K ? end - beg : (size_t)MML_MAX_K);
  MmlTerm<PE> t[MML_MAX_K];
  for (int k = 0; k < K; k++) {
    const size_t i = beg + k;
    t[k].skip = (g1inf && g1inf[i]) || (g2inf && g2inf[i]);
    if (t[k].skip) continue;
    t[k].px = fe_from_ref(g1 + i * 24); t[k].py = fe_from_ref(g1 + i * 24 + 12);
    t[k].qx = E2<PE>::load(g2 + i * 48); t[k].qy = E2<PE>::load(g2 + i * 48 + 24);
    t[k].r.x = t[k].qx; t[k].r.y = t[k].qy; t[k].r.z = E2<PE>::one();
  }
  Fp12T<PE> f;
  multi_miller_shared(f, t, K);
  fp12_save(f, out + j * 144);
}
PAIR_KERNEL k_final_exp(const u32* __restrict__ in, u32* __restrict__ out, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (i >= n) return;
  Fp12T<PE> f, g;
  fp12_load(f, in + i * 144);
  final_exponentiation(g, f);
  fp12_save(g, out + i * 144);
}
// out[j] = product of in[j*fan .. min(n, (j+1)*fan))
PAIR_KERNEL k_fp12_prod(const u32* __restrict__ in, u32* __restrict__ out, size_t n, size_t m, int fan) {
  size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (j >= m) return;
  size_t beg = j * (size_t)fan, end = beg + fan < n ? beg + fan : n;
  Fp12T<PE> acc; fp12_load(acc, in + beg * 144);
  for (size_t i = beg + 1; i < end; i++) {
    Fp12T<PE> x, t; fp12_load(x, in + i * 144);
    fp12_mul(t, acc, x); acc = t;
  }
  fp12_save(acc, out + j * 144);
}
// `&Gt * &Scalar` (pairings.rs:297-322): double-and-add over the 255 low bits of the scalar's little-endian bytes,
// most significant first; out[i] = gt[i]^(scalar[i]) (the target group is written additively in the reference)
PAIR_KERNEL k_gt_mul_scalar(const u32* __restrict__ gt, const u32* __restrict__ scalars, u32* __restrict__ out, size_t n, int form) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (i >= n) return;
  Fp12T<PE> base, acc = fp12_one<PE>(), t;
  fp12_load(base, gt + i * 144);
  u32 s[8];
  scalar_load(scalars, i, form, s);             // bytes or `Scalar` limbs (scalar.hip.h)
  for (int bit = 254; bit >= 0; bit--) {
    fp12_sqr(acc, acc);
    if ((s[bit >> 5] >> (bit & 31)) & 1u) { fp12_mul(t, acc, base); acc = t; }
  }
  fp12_save(acc, out + i * 144);
}
__global__ void k_fp12_one(u32* out) {
  if (threadIdx.x < PL && blockIdx.x == 0) { Fp12T<PE> f = fp12_one<PE>(); fp12_save(f, out); }
}
// parity hook for the Fp6 layer (fp6.rs): op 0 mul (:200-274), 3 square (:277-291), 4 invert (:294-312), 5 mul_by_nonresidue
// (:139-150), 7 frobenius_map (:154-188), 11 mul_by_1 (b.c1; :113-119), 12 mul_by_01 (b.c0, b.c1; :121-136).  36 u64 per element.
PAIR_KERNEL k_fp6_op(int op, const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (i >= n) return;
  Fp6T<PE> x, y, r;
  x.c0 = E2<PE>::load(a + i * 72); x.c1 = E2<PE>::load(a + i * 72 + 24); x.c2 = E2<PE>::load(a + i * 72 + 48);
  if (b) { y.c0 = E2<PE>::load(b + i * 72); y.c1 = E2<PE>::load(b + i * 72 + 24); y.c2 = E2<PE>::load(b + i * 72 + 48); } else y = x;
  switch (op) {
    case 0: fp6_mul(r, x, y); break;
    case 3: fp6_sqr(r, x); break;
    case 4: fp6_inv(r, x); break;
    case 5: r = fp6_mul_by_nonresidue(x); break;
    case 7: fp6_frobenius(r, x); break;
    case 11: fp6_mul_by_1(r, x, y.c1); break;
    default: fp6_mul_by_01(r, x, y.c0, y.c1); break;
  }
  E2<PE>::save(r.c0, out + i * 72); E2<PE>::save(r.c1, out + i * 72 + 24); E2<PE>::save(r.c2, out + i * 72 + 48);
}
PAIR_KERNEL k_fp12_op(int op, const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / PL;
  if (i >= n) return;
  Fp12T<PE> x, y, r;
  fp12_load(x, a + i * 144);
  if (b) fp12_load(y, b + i * 144); else y = x;
  switch (op) {
    case 0: fp12_mul(r, x, y); break;
    case 3: fp12_sqr(r, x); break;
    case 4: fp12_inv(r, x); break;
    case 7: fp12_frobenius(r, x); break;
    case 8: fp12_conj(r, x); break;
    default: cyclotomic_square(r, x); break;
  }
  fp12_save(r, out + i * 144);
}

}  // namespace bls
