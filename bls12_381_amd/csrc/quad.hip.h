// quad.hip.h -- one pairing on a QUAD of adjacent lanes: the distributed-state form of the Fp12 tower.
//
// Reference semantics as in pairing.hip.h: src/fp6.rs, src/fp12.rs (mul :197-214, square :174-185, mul_by_014 :116-128),
// src/pairings.rs (miller_loop :668-694, ell :696-707, doubling_step :709-738, addition_step :740-770,
// final_exponentiation :48-176).  Every value computed here is the same field element the reference computes, so raw Miller
// values and Gt results stay bit-identical after canonicalisation.
//
// Why.  In the lane-PAIR form (pairlane.hip.h / pairing.hip.h) a pairing's state is f 84 + R 42 + P 28 + line 42 words per
// lane plus ~130 words of working set for an Fp6-level operation: more than the 256 registers + 80 LDS words a lane owns at
// two wavefronts per SIMD, so ~2 KB of hot state per lane lives in scratch and misses the L2 (profiles/r02_pairing_pmc.md:
// ~550x the algorithmic HBM traffic, a quarter of the wave cycles waiting).  Here the accumulator f is DISTRIBUTED: lanes
// 4q, 4q+1 (pair A) hold f.c0 and lanes 4q+2, 4q+3 (pair B) hold f.c1, each as three pair-lane Fp2 values (21 words per lane
// less than half of the pair form's 84).  Both pairs execute ONE instruction stream (SIMT); what differs is the operands:
//
//   * a product SLOT is one Fp2 product issued by both pairs at once, operands chosen per pair (selB);
//   * linear work is cheap and simply replicated wherever both pairs need the value (R, the line);
//   * results move between the pairs with one DPP move per word (quad_perm:[2,3,0,1]); the SENDER prepares what the
//     receiver adds (negations, multiplications by xi), because a subtraction of an exchanged value could be folded into
//     v_subrev_u32_dpp, which miscomputes on this toolchain (pairlane.hip.h).
//
// Slot counts per Miller iteration (Fp2 products per lane; the pair form needs twice the ideal figure on half the lanes):
//   Fp12 squaring   6   complex form: A computes c0 c1, B computes (c0 + c1)(c0 + v c1)            (ideal 6)
//   line * f        7   the 13 products of mul_by_014 dealt 7 / 6                                  (ideal 6.5)
//   doubling step   4 squarings + 2 products, + 1 for the new y and the line scaling               (ideal ~5.5)
// i.e. ~95 % of the lanes' multiply-adds are useful, and everything lives in registers.
#pragma once
#include "pairing.hip.h"

namespace bls {

#ifndef BLS_QUAD_BLOCK
#define BLS_QUAD_BLOCK 256
#endif
constexpr int QUAD_BLOCK = BLS_QUAD_BLOCK;
constexpr int QL = 4;                        // lanes per pairing

DEV bool lane_is_B() { return (threadIdx.x & 2) != 0; }
// the value of the lane with the same coefficient in the other pair of the quad
DEV u32 dpp_xpair(u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true); }
template <int A, int V> DEV FeP<A, V> xpair(const FeP<A, V>& a) {
  FeP<A, V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v.l[i] = dpp_xpair(a.v.l[i]);
  return r;
}
template <int A, int V> DEV Fe<A, V> xpair(const Fe<A, V>& a) {
  Fe<A, V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = dpp_xpair(a.l[i]);
  return r;
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }
// role select: pair B takes b, pair A takes a (static bounds: the wider of the two)
template <int A1, int V1, int A2, int V2> DEV auto selB(bool B, const FeP<A1, V1>& b, const FeP<A2, V2>& a) {
  FeP<cmax(A1, A2), cmax(V1, V2)> r;
  r.v = select(B, (Fe<cmax(A1, A2), cmax(V1, V2)>)b.v, (Fe<cmax(A1, A2), cmax(V1, V2)>)a.v);
  return r;
}
// bring a value under the static value bound VT with normalised limbs: a carry pass, plus a weak reduction only if the
// static bound demands one
template <int VT, int A, int V> DEV FeP<1, VT> fit(const FeP<A, V>& a) {
  FeP<1, VT> r;
  if constexpr (V <= VT) r.v = norm(a.v); else r.v = reduce_v(norm(a.v));
  return r;
}
// products of the hot loops: inlined sum-of-two-products per lane, operands renormalised only when the column bound asks.
// Every product is fenced for the instruction scheduler (QFENCE): left alone it interleaves the independent products of a
// tower operation for instruction-level parallelism and the live ranges of their operands blow the register file
// (measured statically: 6 144 scratch loads per Miller iteration without the fences).
#ifndef QFENCE
#define QFENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// c_L = X B0 + Y B1 with X this lane's coefficient of a, B0 / B1 the pair's (b0, b1) broadcast to both lanes, and Y the partner's
// coefficient of a, which the c1 lane SENDS negated: c0 = a0 b0 + (-a1) b1, c1 = a1 b0 + a0 b1 -- three DPP moves, one
// subtraction and one select per limb (mul_inl of pairlane.hip.h needs two selects)
DEV u32 dpp_bcast0(u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xA0 /* quad_perm:[0,0,2,2] */, 0xF, 0xF, true); }
DEV u32 dpp_bcast1(u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, true); }
template <int A1, int V1, int A2, int V2>
DEV FeP<1, pair_mul_v(V1, V2)> qmul_core(const FeP<A1, V1>& a, const FeP<A2, V2>& b) {
  static_assert(A1 * A2 + (A1 + 1) * A2 + 1 <= MAX_A_PROD + 1, "quad fe2 mul: limb bound too large, norm() an operand");
  const bool c1 = lane_is_c1();
  const Fe<A1 + 1, V1 + 1> send = select(c1, neg(a.v), (Fe<A1 + 1, V1 + 1>)a.v);
  Fe<A1 + 1, V1 + 1> y; Fe<A2, V2> b0, b1;
#pragma unroll
  for (int i = 0; i < NL; i++) { y.l[i] = dpp_swap1(send.l[i]); b0.l[i] = dpp_bcast0(b.v.l[i]); b1.l[i] = dpp_bcast1(b.v.l[i]); }
  FeP<1, pair_mul_v(V1, V2)> r;
  r.v = from_v16<pair_mul_v(V1, V2)>(fe_sop2_body(to_v16(a.v), to_v16(b0), to_v16(y), to_v16(b1)));
  return r;
}
template <int A1, int V1, int A2, int V2> DEV auto qmul_(const FeP<A1, V1>& a, const FeP<A2, V2>& b) {
  if constexpr ((2 * A1 + 1) * A2 + 1 <= MAX_A_PROD + 1) return qmul_core(a, b);
  else if constexpr (3 * A2 + 1 <= MAX_A_PROD + 1) return qmul_core(norm(a), b);
  else if constexpr ((2 * A1 + 1) + 1 <= MAX_A_PROD + 1) return qmul_core(a, norm(b));
  else return qmul_core(norm(a), norm(b));
}
template <int A1, int V1, int A2, int V2> DEV auto qmul(const FeP<A1, V1>& a, const FeP<A2, V2>& b) {
  QFENCE(); auto r = qmul_(a, b); QFENCE(); return r;
}
template <int A, int V> DEV auto qsqr(const FeP<A, V>& a) {
  QFENCE();
  if constexpr (2 * A * (2 * A + 1) <= MAX_A_PROD) { auto r = sqr_inl(a); QFENCE(); return r; }
  else { auto r = sqr_inl(norm(a)); QFENCE(); return r; }
}
// Fp2 x Fp (the Fp value may differ between the pairs, it is the same in the two lanes of a pair)
template <int A1, int V1, int A2, int V2> DEV auto qmul_fp(const FeP<A1, V1>& a, const Fe<A2, V2>& k) {
  FeP<1, mul_v(V1, V2)> r;
  QFENCE();
  if constexpr (A1 * A2 <= MAX_A_PROD) r.v = mul_inl(a.v, k); else r.v = mul_inl(norm(a.v), k);
  QFENCE();
  return r;
}

// ---- Fp12 over a quad: this pair's half ------------------------------------------------------------------------------
template <int V> struct Q6 { FeP<1, V> c0, c1, c2; };            // an Fp6 value held by a pair
template <int V> struct Q12 { Q6<V> h; };                        // pair A: c0, pair B: c1 of an Fp12 value
template <int V> DEV Q6<V> xpair6(const Q6<V>& a) { Q6<V> r; r.c0 = xpair(a.c0); r.c1 = xpair(a.c1); r.c2 = xpair(a.c2); return r; }
constexpr int VQ = 2;        // accumulator after a squaring (weakly reduced)
constexpr int VQM = 48;      // accumulator after line multiplications (no reduction needed below this bound)

DEV Q12<VQ> q12_one() {
  Q12<VQ> r;
  constexpr PLimbs k1 = {BLS_ONE_MONT}, k0 = {{0}};
  FeP<1, VQ> one; one.v = select(lane_is_c1() || lane_is_B(), (Fe<1, VQ>)fe1_const(k0), (Fe<1, VQ>)fe1_const(k1));
  FeP<1, VQ> zero; zero.v = (Fe<1, VQ>)fe1_const(k0);
  r.h.c0 = one; r.h.c1 = zero; r.h.c2 = zero;
  return r;
}

// Karatsuba over Fp2, six product slots, operands with normalised limbs (same element as fp6.rs:200-274)
template <int VA, int VB> DEV auto q6_mul(const Q6<VA>& a, const Q6<VB>& b) {
  auto v0 = qmul(a.c0, b.c0);
  auto v1 = qmul(a.c1, b.c1);
  auto v2 = qmul(a.c2, b.c2);
  auto t12 = qmul(add(a.c1, a.c2), add(b.c1, b.c2));
  auto c0 = add(v0, mul_by_nonresidue(sub(sub(t12, v1), v2)));      // v0 + xi (a1 b2 + a2 b1); limbs stay below 15 * 2^28 unnormalised
  auto t01 = qmul(add(a.c0, a.c1), add(b.c0, b.c1));
  auto c1 = add(sub(sub(t01, v0), v1), mul_by_nonresidue(v2));      // a0 b1 + a1 b0 + xi a2 b2
  auto t02 = qmul(add(a.c0, a.c2), add(b.c0, b.c2));
  auto c2 = add(sub(sub(t02, v0), v2), v1);                         // a0 b2 + a2 b0 + a1 b1
  struct R { decltype(c0) c0; decltype(c1) c1; decltype(c2) c2; };
  R r{c0, c1, c2};
  return r;
}

// f <- f^2, complex form (fp12.rs:174-185):  ab = c0 c1;  c0' = (c0 + c1)(c0 + v c1) - ab - v ab;  c1' = 2 ab.
// Pair A multiplies c0 by c1 while pair B multiplies (c0 + c1) by (c0 + v c1): six slots, every lane busy.
template <int VI> DEV Q12<VQ> q12_sqr(const Q12<VI>& f) {
  const bool B = lane_is_B();
  const Q6<VI>& own = f.h;
  const Q6<VI> oth = xpair6(own);                          // A: c1, B: c0
  // X = B ? c0 + c1 : c0        Y = B ? c0 + v c1 : c1        (v (a0, a1, a2) = (xi a2, a0, a1))
  Q6<2 * VI> X;
  X.c0 = norm(selB(B, add(own.c0, oth.c0), own.c0));
  X.c1 = norm(selB(B, add(own.c1, oth.c1), own.c1));
  X.c2 = norm(selB(B, add(own.c2, oth.c2), own.c2));
  Q6<3 * VI + 1> Y;
  Y.c0 = norm(selB(B, add(oth.c0, mul_by_nonresidue(own.c2)), oth.c0));
  Y.c1 = norm(selB(B, add(oth.c1, own.c0), oth.c1));
  Y.c2 = norm(selB(B, add(oth.c2, own.c1), oth.c2));
  auto M = q6_mul(X, Y);                                   // A: ab      B: (c0 + c1)(c0 + v c1)
  // A prepares what B has to add: Z = -(ab + v ab)
  auto m0 = norm(M.c0);
  auto m1 = norm(M.c1);
  auto m2 = norm(M.c2);
  auto r0 = xpair(norm(neg(add(m0, mul_by_nonresidue(m2)))));
  auto r1 = xpair(norm(neg(add(m1, m0))));
  auto r2 = xpair(norm(neg(add(m2, m1))));
  // A: c1' = 2 ab        B: c0' = M + Z
  auto n0 = selB(B, add(m0, r0), dbl(m0));
  auto n1 = selB(B, add(m1, r1), dbl(m1));
  auto n2 = selB(B, add(m2, r2), dbl(m2));
  // ... and the halves change places (A must hold c0')
  Q12<VQ> g;
  g.h.c0 = xpair(fit<VQ>(n0)); g.h.c1 = xpair(fit<VQ>(n1)); g.h.c2 = xpair(fit<VQ>(n2));
  return g;
}

// f <- f * (c0 + c1 v + c4 v w)   (fp12.rs:116-128 with fp6.rs:113-136): the thirteen Fp2 products dealt to the two pairs.
//   a = f.c0 (pair A), b = f.c1 (pair B), S = a + b, o = c1 + c4
//   slot   pair A                     pair B
//    1     a0 c0                      b2 c4
//    2     a1 c1                      b0 c4
//    3     a2 c1                      b1 c4
//    4     (a0 + a1)(c0 + c1)         S1 o
//    5     a2 c0                      S2 o
//    6     S0 c0                      S2 c0
//    7     (S0 + S1)(c0 + o)          (the same: both pairs need it)
//   aa = a * (c0, c1) = (xi T3 + T1, T4 - T1 - T2, T5 + T2)   [A]       bb = b * c4 = (xi T1, T2, T3)   [B]
//   t  = S * (c0, o)  = (xi T5[B] + T6[A], T7 - T6[A] - T4[B], T6[B] + T4[B])
//   new c0 = v bb + aa  -> pair A;       new c1 = t - aa - bb  -> pair B;    one exchange of three values each way.
template <int VI, int V0, int V1, int V4>
DEV Q12<VQM> q12_mul_by_014(const Q12<VI>& f, const FeP<1, V0>& c0, const FeP<1, V1>& c1, const FeP<1, V4>& c4) {
  const bool B = lane_is_B();
  const Q6<VI>& own = f.h;
  // (the other pair's coefficients are fetched where they are used -- one DPP move per word -- instead of being held for the whole
  // routine: 42 fewer live registers)
  // every product is folded into the three local sums (A: aa, B: the part of the new c1 that B computes) as soon as it
  // exists, so that at most one product result is live besides them
  auto T1 = qmul(selB(B, own.c2, own.c0), selB(B, c4, c0));
  auto T2 = qmul(selB(B, own.c0, own.c1), selB(B, c4, c1));
  auto T3 = qmul(selB(B, own.c1, own.c2), selB(B, c4, c1));
  // linear work stays unnormalised while the limbs stay below 15 * 2^28 (the static bounds check it); values are
  // renormalised where they are sent or stored
  auto xT3 = mul_by_nonresidue(T3);                                  // B sends (xi T3, xi T1, T2)
  auto W1 = mul_by_nonresidue(T1);
  auto l0 = selB(B, neg(T1), add(T1, xT3));                          // A: aa0 = xi T3 + T1          B: -T1            (xi applied at the end)
  auto l1 = selB(B, neg(T2), neg(add(T1, T2)));                      // A: -T1 - T2                  B: -T2
  auto l2 = selB(B, neg(T3), T2);                                    // A: T2                        B: -T3
  auto o = add(c1, c4);
  auto S1 = add(own.c1, xpair(own.c1));
  auto T4 = qmul(selB(B, S1, add(own.c0, own.c1)), selB(B, o, add(c0, c1)));
  auto l1b = add(l1, selB(B, neg(T4), T4));                          // A: aa1 = T4 - T1 - T2        B: -T4 - T2
  auto l2b = selB(B, add(l2, T4), l2);                               //                              B: T4 - T3
  auto S2 = add(own.c2, xpair(own.c2));
  auto T5 = qmul(selB(B, S2, own.c2), selB(B, o, c0));
  auto l0b = norm(selB(B, add(l0, T5), l0));                         //                              B: T5 - T1
  auto l2c = selB(B, l2b, add(l2b, T5));                             // A: aa2 = T5 + T2
  auto S0 = add(own.c0, xpair(own.c0));
  auto T6 = qmul(selB(B, S2, S0), c0);
  auto l2d = norm(selB(B, add(l2c, T6), l2c));                       //                              B: bn2 = T6 + T4 - T3
  auto l1n = norm(l1b);
  // A sends U = (T6 - aa0, -T6 - aa1, -aa2)
  auto U0 = sub(T6, l0b);
  auto U1 = neg(add(T6, l1n));
  auto U2 = neg(l2d);
  auto T7 = qmul(norm(add(S0, S1)), norm(add(c0, o)));
  auto l1c = selB(B, add(l1n, T7), l1n);                             //                              B: bn1 = T7 - T4 - T2
  auto l0c = selB(B, mul_by_nonresidue(l0b), l0b);                   //                              B: bn0 = xi (T5 - T1)
  auto g0 = xpair(selB(B, xT3, U0));
  auto g1 = xpair(selB(B, W1, U1));
  auto g2 = xpair(selB(B, T2, U2));
  Q12<VQM> r;
  r.h.c0 = fit<VQM>(add(l0c, g0));
  r.h.c1 = fit<VQM>(add(l1c, g1));
  r.h.c2 = fit<VQM>(add(l2d, g2));
  return r;
}

// ---- Miller loop -------------------------------------------------------------------------------------------------------
// R and the line are REPLICATED on both pairs (linear work costs the same whether one pair or both need it); only the
// products are dealt out.  pairings.rs:709-738.
typedef FeP<1, VSP> QR;                       // stored coordinate of the running point
struct QJac { QR x, y, z; };
struct QLin { QR a, b, c; };                  // the reference's coefficient triple (tmp0, tmp3, tmp6)
DEV void q_doubling_step(QJac& r, QLin& l) {
  const bool B = lane_is_B();
  // level 1: x^2 | y^2 ;  z^2 | (z + y)^2
  auto S1 = qsqr(selB(B, r.y, r.x));
  auto S2 = qsqr(selB(B, add(r.z, r.y), r.z));
  auto S1x = xpair(S1); auto S2x = xpair(S2);
  auto tmp0 = selB(B, S1x, S1);
  auto tmp1 = selB(B, S1, S1x);
  auto zsq = selB(B, S2x, S2);
  auto zy2 = selB(B, S2, S2x);
  auto tmp4 = add(dbl(tmp0), tmp0);
  QR rz = fit<VSP>(sub(sub(zy2, tmp1), zsq));
  // level 2: tmp4^2 | tmp1^2 ;  (x + tmp4)^2 | (tmp1 + x)^2 ;  tmp4 zsq | rz zsq
  auto S3 = qsqr(selB(B, tmp1, tmp4));
  auto S4 = qsqr(selB(B, add(tmp1, r.x), add(r.x, tmp4)));
  auto M1 = qmul(selB(B, rz, tmp4), zsq);
  auto S3x = xpair(S3); auto S4x = xpair(S4); auto M1x = xpair(M1);
  auto tmp5 = selB(B, S3x, S3);
  auto tmp2 = selB(B, S3, S3x);
  auto t6s = selB(B, S4x, S4);
  auto t3s = selB(B, S4, S4x);
  auto t3 = selB(B, M1x, M1);
  auto t0 = selB(B, M1, M1x);
  auto tmp3d = norm(dbl(sub(sub(t3s, tmp0), tmp2)));
  auto rx = sub(sub(tmp5, tmp3d), tmp3d);
  // level 3 (both pairs: the running point stays replicated)
  auto ry = qmul(sub(tmp3d, rx), tmp4);
  auto ryo = sub(ry, mul_small<8>(tmp2));
  auto t6 = sub(sub(sub(t6s, tmp0), tmp5), mul_small<4>(tmp1));
  r.x = fit<VSP>(rx); r.y = fit<VSP>(ryo); r.z = rz;
  l.a = fit<VSP>(dbl(t0)); l.b = fit<VSP>(neg(dbl(t3))); l.c = fit<VSP>(t6);
}
// pairings.rs:696-707: the line evaluated at P -- c4 = l.a * py on pair A, c1 = l.b * px on pair B, one Fp product per lane
// (pp = py on pair A, px on pair B) -- and multiplied into f
template <int VI> DEV Q12<VQM> q_ell(const Q12<VI>& f, const QLin& l, const fe1& pp) {
  const bool B = lane_is_B();
  auto LS = qmul_fp(selB(B, l.b, l.a), pp);
  auto LSx = xpair(LS);
  auto c1 = selB(B, LS, LSx);
  auto c4 = selB(B, LSx, LS);
  return q12_mul_by_014(f, l.c, c1, c4);
}
// the five addition steps (pairings.rs:740-770; CLN Algorithm 27) run replicated on both pairs -- 15 products at the pair
// form's cost, five times per pairing -- with the same inlined products as everything else in the loop: a call here would
// force the loop-carried state through memory around it
DEV void q_addition_step(QJac& r, const QR& qx, const QR& qy, QLin& l) {
  auto zsq = qsqr(r.z);
  auto ysq = qsqr(qy);
  auto t0 = qmul(zsq, qx);
  auto t1 = qmul(norm(sub(sub(qsqr(add(qy, r.z)), ysq), zsq)), zsq);
  auto t2 = norm(sub(t0, r.x));
  auto t3 = qsqr(t2);
  auto t4 = norm(mul_small<4>(t3));
  auto t5 = qmul(t4, t2);
  auto t6 = norm(sub(sub(t1, r.y), r.y));
  auto t9 = qmul(t6, qx);
  auto t7 = qmul(t4, r.x);
  auto rx = norm(sub(sub(sub(qsqr(t6), t5), t7), t7));
  QR rz = fit<VSP>(sub(sub(qsqr(add(r.z, t2)), zsq), t3));
  auto t10 = add(qy, rz);
  auto t8 = qmul(norm(sub(t7, rx)), t6);
  auto t0b = qmul(r.y, t5);
  auto ry = sub(t8, norm(dbl(t0b)));
  auto t10b = sub(qsqr(t10), ysq);
  auto ztsq = qsqr(rz);
  auto t10c = sub(norm(t10b), ztsq);
  auto t9b = sub(norm(dbl(t9)), norm(t10c));
  r.x = fit<VSP>(rx); r.y = fit<VSP>(ry); r.z = rz;
  l.a = fit<VSP>(dbl(rz)); l.b = fit<VSP>(dbl(norm(neg(t6)))); l.c = fit<VSP>(t9b);
}
// LDS parking lot (as in pairing.hip.h): the running point R and the lane's coordinate of P are needed only inside the
// doubling / addition step and the line evaluation; parked in LDS (word w of lane t at [w * QUAD_BLOCK + t], conflict-free)
// they are out of the register allocator's hands while f is multiplied and squared.  56 words per lane = 56 KB per block.
constexpr int QPARK_WORDS = 4 * NL;
template <int V> DEV void qpark_put(u32* park, int slot, const Fe<1, V>& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) park[(slot * NL + i) * QUAD_BLOCK] = a.l[i];
}
template <int V> DEV void qpark_get(const u32* park, int slot, Fe<1, V>& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) a.l[i] = park[(slot * NL + i) * QUAD_BLOCK];
}
DEV void qpark_put_r(u32* park, const QJac& r) { qpark_put(park, 0, r.x.v); qpark_put(park, 1, r.y.v); qpark_put(park, 2, r.z.v); }
DEV void qpark_get_r(const u32* park, QJac& r) { qpark_get(park, 0, r.x.v); qpark_get(park, 1, r.y.v); qpark_get(park, 2, r.z.v); }
// g2w: the wire words of Q (the affine coordinates are re-read for each of the five addition steps instead of living in
// registers through the loop)
DEV void q_miller_loop(Q12<VQM>& fout, const fe1& px, const fe1& py, const u32* __restrict__ g2w, u32* park) {
  const bool B = lane_is_B();
  {
    QJac r; r.x = E2<QR>::load(g2w); r.y = E2<QR>::load(g2w + 24); r.z = E2<QR>::one();
    qpark_put_r(park, r);
    qpark_put(park, 3, select(B, px, py));
  }
  Q12<VQ> f = q12_one();
  Q12<VQM> g;
  for (int b = 61; b >= -1; b--) {            // bit 62 is the leading one; b = -1: the final doubling step (pairings.rs:686-687)
    QLin l;
    fair_tick();
    // while R is in registers its three LDS slots hold f (g in the addition step): the accumulator is out of the register allocator's
    // hands exactly where it is not used (round 4: 561 -> 276 scratch instructions in the kernel, 84 more LDS instructions per iteration)
    {
      QJac r; qpark_get_r(park, r);
      qpark_put(park, 0, f.h.c0.v); qpark_put(park, 1, f.h.c1.v); qpark_put(park, 2, f.h.c2.v);
      q_doubling_step(r, l);
      qpark_get(park, 0, f.h.c0.v); qpark_get(park, 1, f.h.c1.v); qpark_get(park, 2, f.h.c2.v);
      qpark_put_r(park, r);
    }
    fair_tick();
    { fe1 pp; qpark_get(park, 3, pp); g = q_ell(f, l, pp); }
    if (b < 0) break;
    if ((X_HALF >> b) & 1) {
      QJac r; qpark_get_r(park, r);
      qpark_put(park, 0, g.h.c0.v); qpark_put(park, 1, g.h.c1.v); qpark_put(park, 2, g.h.c2.v);
      const QR qx = E2<QR>::load(g2w), qy = E2<QR>::load(g2w + 24);
      q_addition_step(r, qx, qy, l);
      qpark_get(park, 0, g.h.c0.v); qpark_get(park, 1, g.h.c1.v); qpark_get(park, 2, g.h.c2.v);
      qpark_put_r(park, r);
      fe1 pp; qpark_get(park, 3, pp);
      g = q_ell(g, l, pp);
    }
    fair_tick();
    f = q12_sqr(g);
  }
  // conjugate (BLS_X_IS_NEGATIVE): c1 -> -c1 on pair B
  fout.h.c0 = fit<VQM>(selB(B, neg(g.h.c0), g.h.c0));
  fout.h.c1 = fit<VQM>(selB(B, neg(g.h.c1), g.h.c1));
  fout.h.c2 = fit<VQM>(selB(B, neg(g.h.c2), g.h.c2));
}

// ---- wire I/O: pair A reads / writes words 0..71 (c0), pair B words 72..143 (c1) ------------------------------------------
template <int V> DEV void q12_save(const Q12<V>& f, u32* w) {
  u32* p = w + (lane_is_B() ? 72 : 0);
  E2<fp2p>::save((fp2p)fit<VSP>(f.h.c0), p); E2<fp2p>::save((fp2p)fit<VSP>(f.h.c1), p + 24); E2<fp2p>::save((fp2p)fit<VSP>(f.h.c2), p + 48);
}
DEV Q12<VQ> q12_load(const u32* w) {
  const u32* p = w + (lane_is_B() ? 72 : 0);
  Q12<VQ> f;
  f.h.c0 = fit<VQ>(E2<fp2p>::load(p)); f.h.c1 = fit<VQ>(E2<fp2p>::load(p + 24)); f.h.c2 = fit<VQ>(E2<fp2p>::load(p + 48));
  return f;
}

// ---- final exponentiation ------------------------------------------------------------------------------------------------
// pairings.rs:48-176.  The code that runs a few times per pairing works on the pair's half in the stored form of
// pairing.hip.h (Fp6T<QR>: the per-pair Fp6 routines there are reused as they are -- every pair runs them on ITS half) and
// stays out of line; only the run of compressed cyclotomic squarings, ~90 % of the multiply-adds, is inlined.
struct QC12 { Fp6T<QR> h; };                             // pair A: c0, pair B: c1
DEV Fp6T<QR> xpair6(const Fp6T<QR>& a) { Fp6T<QR> r; r.c0 = xpair(a.c0); r.c1 = xpair(a.c1); r.c2 = xpair(a.c2); return r; }
DEV Fp6T<QR> sel6(bool B, const Fp6T<QR>& b, const Fp6T<QR>& a) {
  Fp6T<QR> r; r.c0 = selB(B, b.c0, a.c0); r.c1 = selB(B, b.c1, a.c1); r.c2 = selB(B, b.c2, a.c2); return r;
}
// fp12.rs:197-214 over a quad, nine product slots: a0 b0 | a1 b1 by the per-pair Karatsuba routine (six slots), then the six
// products of (a0 + a1)(b0 + b1) dealt three to a pair; one exchange of three values each way.
//   with s = a0 + a1, t = b0 + b1:   A: P1 = s0 t0, P2 = s1 t1, P3 = (s1 + s2)(t1 + t2)      B: P1 = s2 t2, P2 = (s0 + s1)(t0 + t1), P3 = (s0 + s2)(t0 + t2)
//   s t = (P1 + xi (P3 - P2), -P1 - P2, P2 - P1)[A]  +  (-xi P1, P2 + xi P1, P3 - P1)[B]
DEVNI void qc_mul(QC12& r, const QC12& a, const QC12& b) {
  const bool B = lane_is_B();
  Fp6T<QR> aa;
  fp6_mul(aa, a.h, b.h);                                 // A: a0 b0      B: a1 b1
  const Fp6T<QR> sa = fp6_add(a.h, xpair6(a.h)), sb = fp6_add(b.h, xpair6(b.h));
  auto P1 = pmul(selB(B, sa.c2, sa.c0), selB(B, sb.c2, sb.c0));
  auto P2 = pmul(selB(B, add(sa.c0, sa.c1), sa.c1), selB(B, add(sb.c0, sb.c1), sb.c1));
  auto P3 = pmul(selB(B, add(sa.c0, sa.c2), add(sa.c1, sa.c2)), selB(B, add(sb.c0, sb.c2), add(sb.c1, sb.c2)));
  auto xP1 = norm(mul_by_nonresidue(P1));
  // this pair's share of s t
  auto sh0 = norm(selB(B, neg(xP1), add(P1, mul_by_nonresidue(norm(sub(P3, P2))))));
  auto sh1 = norm(selB(B, add(P2, xP1), neg(norm(add(P1, P2)))));
  auto sh2 = norm(selB(B, sub(P3, P1), sub(P2, P1)));
  // A sends its share minus a0 b0 (B assembles c1 = s t - a0 b0 - a1 b1); B sends v (a1 b1) (A assembles c0 = a0 b0 + v a1 b1)
  auto g0 = xpair(norm(selB(B, mul_by_nonresidue(aa.c2), sub(sh0, aa.c0))));
  auto g1 = xpair(norm(selB(B, aa.c0, sub(sh1, aa.c1))));
  auto g2 = xpair(norm(selB(B, aa.c1, sub(sh2, aa.c2))));
  r.h.c0 = S2(add(selB(B, sub(sh0, aa.c0), aa.c0), g0));
  r.h.c1 = S2(add(selB(B, sub(sh1, aa.c1), aa.c1), g1));
  r.h.c2 = S2(add(selB(B, sub(sh2, aa.c2), aa.c2), g2));
}
// fp12.rs:136-141
DEV void qc_conj(QC12& r, const QC12& a) { r.h = sel6(lane_is_B(), fp6_neg(a.h), a.h); }
// fp12.rs:145-171: Fp6 Frobenius on either half, then the c1 half times (u + 1)^((p - 1) / 6)
DEVNI void qc_frobenius(QC12& r, const QC12& a) {
  constexpr PLimbs k0 = {BLS_FROB12_C1_0}, k1 = {BLS_FROB12_C1_1};
  const bool B = lane_is_B();
  Fp6T<QR> t;
  fp6_frobenius(t, a.h);
  const QR K = E2<QR>::konst(k0, k1);
  r.h.c0 = selB(B, S2(pmul(t.c0, K)), t.c0); r.h.c1 = selB(B, S2(pmul(t.c1, K)), t.c1); r.h.c2 = selB(B, S2(pmul(t.c2, K)), t.c2);
}
// fp12.rs:187-194: (c0 + c1 w)^-1 = (c0 - c1 w) / (c0^2 - v c1^2)
DEVNI void qc_inv(QC12& r, const QC12& a) {
  const bool B = lane_is_B();
  Fp6T<QR> s, t, ti;
  fp6_sqr(s, a.h);                                       // A: c0^2       B: c1^2
  const Fp6T<QR> mine = sel6(B, fp6_neg(fp6_mul_by_nonresidue(s)), s);      // B: -v c1^2
  t = fp6_add(mine, xpair6(mine));                        // both pairs: c0^2 - v c1^2
  fp6_inv(ti, t);
  const Fp6T<QR> k = sel6(B, fp6_neg(ti), ti);
  fp6_mul(r.h, a.h, k);
}
// pairings.rs:66-112 with z0 = c0.c0, z4 = c0.c1, z3 = c0.c2 on pair A and z2 = c1.c0, z1 = c1.c1, z5 = c1.c2 on pair B:
// the six plain squares are local (three slots), the three squares of sums take two more.
DEVNI void qc_cyc_sqr(QC12& r, const QC12& f) {
  const bool B = lane_is_B();
  const QR m0 = f.h.c0, m1 = f.h.c1, m2 = f.h.c2;         // A: z0 z4 z3    B: z2 z1 z5
  const QR o0 = xpair(m0), o1 = xpair(m1), o2 = xpair(m2);
  auto q0 = psqr(m0);                                     // A: z0^2        B: z2^2
  auto q1 = psqr(m1);                                     // A: z4^2        B: z1^2
  auto q2 = psqr(m2);                                     // A: z3^2        B: z5^2
  // sums: z0 + z1 = A.m0 + B.m1, z2 + z3 = B.m0 + A.m2, z4 + z5 = A.m1 + B.m2
  auto s01 = selB(B, add(o0, m1), add(m0, o1));
  auto s23 = selB(B, add(m0, o2), add(o0, m2));
  auto s45 = selB(B, add(o1, m2), add(m1, o2));
  auto w1 = psqr(selB(B, s23, s01));                      // A: (z0 + z1)^2     B: (z2 + z3)^2
  auto w2 = psqr(s45);                                    // both
  // pair A's squares are SUBTRACTED on pair B, pair B's are added on pair A: A sends negatives (never subtract an exchanged value)
  const auto x0 = xpair(norm(selB(B, q0, neg(q0))));
  const auto x1 = xpair(norm(selB(B, q1, neg(q1))));
  const auto x2 = xpair(norm(selB(B, q2, neg(q2))));
  const auto xw = xpair(w1);
  // fp4_square(a, b) = (xi b^2 + a^2, (a + b)^2 - a^2 - b^2)
  //   (t0, t1) of (z0, z1): a^2 = A.q0, b^2 = B.q1, sum^2 = A.w1        nz0 = 3 t0 - 2 z0 [A]      nz1 = 3 t1 + 2 z1 [B]
  //   (t0, t1) of (z2, z3): a^2 = B.q0, b^2 = A.q2, sum^2 = B.w1        nz4 = 3 t0 - 2 z4 [A]      nz5 = 3 t1 + 2 z5 [B]
  //   (t2, t3) of (z4, z5): a^2 = A.q1, b^2 = B.q2, sum^2 = w2          nz2 = 3 xi t3 + 2 z2 [B]   nz3 = 3 t2 - 2 z3 [A]
  // pair A (x = B's squares):   nz0 = 3 (xi x1 + q0) - 2 m0;   nz4 = 3 (xi q2 + x0) - 2 m1;   nz3 = 3 (xi x2 + q1) - 2 m2
  // pair B (x = -A's squares):  nz2 = 3 xi (w2 + x1 - q2) + 2 m0;   nz1 = 3 (xw + x0 - q1) + 2 m1;   nz5 = 3 (w1 - q0 + x2) + 2 m2
  auto tA0 = add(mul_by_nonresidue(x1), q0);
  auto tA1 = add(mul_by_nonresidue(q2), x0);
  auto tA2 = add(mul_by_nonresidue(x2), q1);
  auto tB0 = mul_by_nonresidue(norm(sub(add(w2, x1), q2)));
  auto tB1 = sub(add(xw, x0), q1);
  auto tB2 = add(sub(w1, q0), x2);
  auto e0 = norm(selB(B, tB0, tA0)); auto e1 = norm(selB(B, tB1, tA1)); auto e2 = norm(selB(B, tB2, tA2));
  // A: 3 t - 2 z = 2 (t - z) + t        B: 3 t + 2 z = 2 (t + z) + t
  r.h.c0 = S2(add(dbl(norm(selB(B, add(e0, m0), sub(e0, m0)))), e0));
  r.h.c1 = S2(add(dbl(norm(selB(B, add(e1, m1), sub(e1, m1)))), e1));
  r.h.c2 = S2(add(dbl(norm(selB(B, add(e2, m2), sub(e2, m2)))), e2));
}
// One compressed squaring (Karabina; see cyclotomic_exp_compressed in pairing.hip.h): pair A carries (z2, z3), pair B
// (z4, z5); each pair squares ITS Fp4 element (three slots, all lanes busy) and sends the result across:
//   A: (nz2, nz3) = (3 xi t3 + 2 z2, 3 t2 - 2 z3) with (t2, t3) from B      B: (nz4, nz5) = (3 t0 - 2 z4, 3 t1 + 2 z5) with (t0, t1) from A
typedef FeP<1, VQ> QZ;
// the new (a, b) with limbs normalised and the value bounds the arithmetic produced: the caller decides where a value reduction goes
template <int VA, int VB> DEV auto q_cyc_sqr_compressed_raw(const FeP<1, VA>& a, const FeP<1, VB>& b) {
  const bool B = lane_is_B();
  auto t0 = qsqr(a);
  auto t1 = qsqr(b);
  auto t3s = qsqr(add(a, b));
  auto u0 = add(mul_by_nonresidue(t1), t0);               // this pair's fp4_square: (u0, u1), limbs unnormalised
  auto u1 = sub(sub(t3s, t0), t1);
  // what the other pair needs: A wants (xi t3, t2) = (xi u1, u0) of B; B wants (t0, t1) = (u0, u1) of A
  auto r0 = xpair(norm(selB(B, mul_by_nonresidue(u1), u0)));
  auto r1 = xpair(norm(selB(B, u0, u1)));
  // A: nz2 = 3 r0 + 2 z2, nz3 = 3 r1 - 2 z3        B: nz4 = 3 r0 - 2 z4, nz5 = 3 r1 + 2 z5      (the sign is chosen on the SMALL operand: one negation
  // and one select, then two shift-and-add instructions per limb; round 6 -- the sum and the difference were both formed and selected before)
  auto na = norm(add(mul_small<3>(r0), dbl(selB(B, neg(a), a))));
  auto nb = norm(add(mul_small<3>(r1), dbl(selB(B, b, neg(b)))));
  struct R { decltype(na) a; decltype(nb) b; };
  return R{na, nb};
}
DEV void q_cyc_sqr_compressed(QZ& a, QZ& b) {
  auto r = q_cyc_sqr_compressed_raw(a, b);
  a = fit<VQ>(r.a); b = fit<VQ>(r.b);
}
// TWO squarings with ONE value reduction per coordinate: the first leaves its results unreduced (value bounds 57 p / 30 p from inputs below 2 p),
// the second squares those (the static bounds of the products still hold: they are checked at compile time) and reduces.  Saves two of the
// four reduce_v of a pair of squarings (~64 instructions each, all 64-bit VOP3 work).
DEV void q_cyc_sqr_compressed_x2(QZ& a, QZ& b) {
  auto r1 = q_cyc_sqr_compressed_raw(a, b);
  auto r2 = q_cyc_sqr_compressed_raw(r1.a, r1.b);
  a = fit<VQ>(r2.a); b = fit<VQ>(r2.b);
}
// f^|x| conjugated (pairings.rs:114-132), |x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16: 57 compressed squarings with the
// states after 16 and 48 of them parked in LDS, the three powers decompressed with ONE shared inversion, the powers 2^60,
// 2^62, 2^63 by six plain cyclotomic squarings from there, and the product of the six.  Same field element as the
// reference's square-and-multiply; `false` in the degenerate case z2 = z3 = 0 of a compressed state (e.g. f = 1).
constexpr int QSNAP_WORDS = 2 * NL;
DEV void qsnap_put(u32* park, int j, const QZ& a, const QZ& b) { qpark_put(park, 2 * j, a.v); qpark_put(park, 2 * j + 1, b.v); }
DEV void qsnap_get(const u32* park, int j, QZ& a, QZ& b) { qpark_get(park, 2 * j, a.v); qpark_get(park, 2 * j + 1, b.v); }
DEVNI bool q_cyc_exp_compressed(QC12& r, const QC12& f, u32* park) {
  const bool B = lane_is_B();
  // pair A takes (z2, z3) = (c1.c0, c0.c2), pair B (z4, z5) = (c0.c1, c1.c2)
  QZ a = fit<VQ>(xpair(selB(B, f.h.c0, f.h.c1))), b = fit<VQ>(f.h.c2);
  for (int i = 2; i <= 56; i += 2) {          // 28 double steps (the snapshots fall on even counts), then the 57th squaring
    fair_tick();
    q_cyc_sqr_compressed_x2(a, b);
    if (i == 16) qsnap_put(park, 0, a, b);
    if (i == 48) qsnap_put(park, 1, a, b);
  }
  q_cyc_sqr_compressed(a, b);
  // decompression (both pairs, replicated): z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2)  [z2 = 0: 2 z4 z5 / z3],
  // z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1
  QR z2[3], z3[3], z4[3], z5[3], den[3], pre[3];
  bool z2zero[3];
  bool ok = true;
  for (int j = 0; j < 3; j++) {
    QZ sa, sb;
    if (j < 2) qsnap_get(park, j, sa, sb); else { sa = a; sb = b; }
    const QZ oa = xpair(sa), ob = xpair(sb);
    z2[j] = (QR)selB(B, oa, sa); z3[j] = (QR)selB(B, ob, sb); z4[j] = (QR)selB(B, sa, oa); z5[j] = (QR)selB(B, sb, ob);
    z2zero[j] = is_zero_fast(z2[j]);
    if (z2zero[j]) { den[j] = z3[j]; ok = ok && !is_zero_fast(z3[j]); }
    else den[j] = S2(mul_small<4>(z2[j]));
  }
  if (!ok) return false;
  pre[0] = den[0];
  for (int j = 1; j < 3; j++) pre[j] = S2(pmul(pre[j - 1], den[j]));
  QR run = S2(inv(pre[2]));
  QC12 g[3];
  for (int j = 2; j >= 0; j--) {
    QR dinv = j ? S2(pmul(run, pre[j - 1])) : run;
    if (j) run = S2(pmul(run, den[j]));
    QR num;
    if (z2zero[j]) num = S2(dbl(pmul(z4[j], z5[j])));
    else num = S2(sub(add(mul_by_nonresidue(psqr(z5[j])), mul_small<3>(psqr(z4[j]))), dbl(z3[j])));
    QR z1 = S2(pmul(num, dinv));
    QR w = S2(sub(add(dbl(psqr(z1)), pmul(z2[j], z5[j])), mul_small<3>(pmul(z3[j], z4[j]))));
    QR z0 = S2(add(mul_by_nonresidue(w), E2<QR>::one()));
    // A: c0 = (z0, z4, z3)      B: c1 = (z2, z1, z5)
    g[j].h.c0 = selB(B, z2[j], z0); g[j].h.c1 = selB(B, z1, z4[j]); g[j].h.c2 = selB(B, z5[j], z3[j]);
  }
  // g[0] = f^(2^16), g[1] = f^(2^48), g[2] = f^(2^57); then 2^60, 2^62, 2^63 by plain cyclotomic squarings
  QC12 acc, t = g[2];
  qc_mul(acc, g[0], g[1]);
  qc_mul(acc, acc, t);
  for (int i = 58; i <= 63; i++) {
    qc_cyc_sqr(t, t);
    if (i == 60 || i == 62 || i == 63) qc_mul(acc, acc, t);
  }
  qc_conj(r, acc);
  return true;
}
// the reference's schedule (pairings.rs:114-132), for the degenerate inputs of the compressed route
DEVNI void q_cyc_exp_plain(QC12& r, const QC12& f) {
  constexpr unsigned long long X = 0xd201000000010000ull;
  QC12 tmp = f;
  for (int b = 62; b >= 0; b--) {
    qc_cyc_sqr(tmp, tmp);
    if ((X >> b) & 1) qc_mul(tmp, tmp, f);
  }
  qc_conj(r, tmp);
}
DEV void q_cyc_exp(QC12& r, const QC12& f, u32* park) {
  if (!q_cyc_exp_compressed(r, f, park)) q_cyc_exp_plain(r, f);
}
// pairings.rs:134-173 (every helper tolerates r aliasing an input)
DEV void q_final_exponentiation(QC12& out, const QC12& fin, u32* park) {
  QC12 t0, t1, t2, t3, t4, t5, t6;
  qc_conj(t0, fin);                                      // f^(p^6): six Frobenius maps = conjugation (fp12.rs:145-171 applied six times)
  qc_inv(t1, fin);
  qc_mul(t2, t0, t1);
  t1 = t2;
  qc_frobenius(t2, t2); qc_frobenius(t2, t2);
  qc_mul(t2, t2, t1);
  qc_cyc_sqr(t1, t2); qc_conj(t1, t1);
  q_cyc_exp(t3, t2, park);
  qc_cyc_sqr(t4, t3);
  qc_mul(t5, t1, t3);
  q_cyc_exp(t1, t5, park);
  q_cyc_exp(t0, t1, park);
  q_cyc_exp(t6, t0, park);
  qc_mul(t6, t6, t4);
  q_cyc_exp(t4, t6, park);
  qc_conj(t5, t5);
  qc_mul(t5, t5, t2);
  qc_mul(t4, t4, t5);
  qc_conj(t5, t2);
  qc_mul(t1, t1, t2);
  qc_frobenius(t1, t1); qc_frobenius(t1, t1); qc_frobenius(t1, t1);
  qc_mul(t6, t6, t5);
  qc_frobenius(t6, t6);
  qc_mul(t3, t3, t0);
  qc_frobenius(t3, t3); qc_frobenius(t3, t3);
  qc_mul(t3, t3, t1);
  qc_mul(t3, t3, t6);
  qc_mul(out, t3, t4);
}

// ---- kernels: pairing i lives on lanes 4i .. 4i+3 of the grid ---------------------------------------------------------------
#ifndef BLS_QUAD_WAVES
#define BLS_QUAD_WAVES 2
#endif
#define QUAD_KERNEL __global__ void __launch_bounds__(QUAD_BLOCK, BLS_QUAD_WAVES)

template <int V> DEV QC12 q12_to_cold(const Q12<V>& f) { QC12 r; r.h.c0 = fit<VSP>(f.h.c0); r.h.c1 = fit<VSP>(f.h.c1); r.h.c2 = fit<VSP>(f.h.c2); return r; }
DEV void qc_save(const QC12& f, u32* w) {
  u32* p = w + (lane_is_B() ? 72 : 0);
  E2<QR>::save(f.h.c0, p); E2<QR>::save(f.h.c1, p + 24); E2<QR>::save(f.h.c2, p + 48);
}
DEV QC12 qc_load(const u32* w) {
  const u32* p = w + (lane_is_B() ? 72 : 0);
  QC12 f; f.h.c0 = E2<QR>::load(p); f.h.c1 = E2<QR>::load(p + 24); f.h.c2 = E2<QR>::load(p + 48);
  return f;
}
// mode 0: out[i] = pairing(g1[i], g2[i]);  mode 1: out[i] = raw Miller loop value.
// Identity on either side -> Fp12::one() (pairings.rs:636-651; multi_miller_loop skips such terms :566-569).
QUAD_KERNEL k_pairing_quad(int mode, const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2,
                           const uint8_t* __restrict__ g2inf, u32* __restrict__ out, size_t n) {
  fair_init();
  __shared__ u32 park_lds[QPARK_WORDS * QUAD_BLOCK];
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / QL;
  if (i >= n) return;
  const bool ident = (g1inf && g1inf[i]) || (g2inf && g2inf[i]);
  QC12 f;
  if (ident) {
    f = q12_to_cold(q12_one());
  } else {
    fe1 px = fe_from_ref(g1 + i * 24), py = fe_from_ref(g1 + i * 24 + 12);
    Q12<VQM> m;
    q_miller_loop(m, px, py, g2 + i * 48, park_lds + threadIdx.x);
    f = q12_to_cold(m);
    if (mode == 0) { QC12 g; q_final_exponentiation(g, f, park_lds + threadIdx.x); f = g; }
  }
  qc_save(f, out + i * 144);
}
QUAD_KERNEL k_final_exp_quad(const u32* __restrict__ in, u32* __restrict__ out, size_t n) {
  fair_init();
  __shared__ u32 park_lds[QPARK_WORDS * QUAD_BLOCK];
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / QL;
  if (i >= n) return;
  QC12 f = qc_load(in + i * 144), g;
  q_final_exponentiation(g, f, park_lds + threadIdx.x);
  qc_save(g, out + i * 144);
}
// out[j] = product of in[j*fan .. min(n, (j+1)*fan)): the levels of the Fp12 product tree (`MillerLoopResult + MillerLoopResult`,
// pairings.rs:179-186) that do not fill the chip -- one multiplication of a lone quad is ~half the latency of a lone lane pair's
QUAD_KERNEL k_fp12_prod_quad(const u32* __restrict__ in, u32* __restrict__ out, size_t n, size_t m, int fan) {
  size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / QL;
  if (j >= m) return;
  size_t beg = j * (size_t)fan, end = beg + fan < n ? beg + fan : n;
  QC12 acc = qc_load(in + beg * 144);
  for (size_t i = beg + 1; i < end; i++) {
    QC12 x = qc_load(in + i * 144), t;
    qc_mul(t, acc, x); acc = t;
  }
  qc_save(acc, out + j * 144);
}
// N independent products (`multi_miller_loop` once per CSR segment, pairings.rs:554-603: the accumulator of segment s is the product of
// its terms' Miller values; an empty segment gives `MillerLoopResult::default()` = one, :28-32).  Segment s = values
// [off[s], off[s + 1]) of `in`, cut into `parts` runs of equal length: quad (s, c) multiplies run c and writes out[s * parts + c].
// parts = 1 for short segments (one quad walks the segment); parts = 32 when segments may be long, followed by k_fp12_prod_quad with
// fan = 32 over the partial products.  Offsets beyond `total` are clamped (the device-pointer entry point cannot validate them).
QUAD_KERNEL k_fp12_prod_seg_quad(const u32* __restrict__ in, const unsigned long long* __restrict__ off, size_t nseg, size_t total, int parts,
                                 u32* __restrict__ out) {
  const size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / QL;
  if (q >= nseg * (size_t)parts) return;
  const size_t sgm = q / (size_t)parts, c = q % (size_t)parts;
  size_t beg = (size_t)off[sgm], end = (size_t)off[sgm + 1];
  if (end > total) end = total;
  if (beg > end) beg = end;
  const size_t run = (end - beg + (size_t)parts - 1) / (size_t)parts;
  size_t pb = beg + c * run, pe = pb + run;
  if (pb > end) pb = end;
  if (pe > end) pe = end;
  QC12 acc;
  if (pb == pe) acc = q12_to_cold(q12_one());
  else {
    acc = qc_load(in + pb * 144);
    for (size_t i = pb + 1; i < pe; i++) {
      QC12 x = qc_load(in + i * 144), t;
      qc_mul(t, acc, x); acc = t;
    }
  }
  qc_save(acc, out + q * 144);
}
// parity hook: op 0 mul, 4 invert, 7 frobenius_map, 8 conjugate, 9 cyclotomic_square, 10 cyclotomic exponentiation (f^|x| conjugated)
QUAD_KERNEL k_fp12_op_quad(int op, const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n) {
  __shared__ u32 park_lds[QPARK_WORDS * QUAD_BLOCK];
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / QL;
  if (i >= n) return;
  QC12 x = qc_load(a + i * 144), y = b ? qc_load(b + i * 144) : x, r;
  switch (op) {
    case 0: qc_mul(r, x, y); break;
    case 4: qc_inv(r, x); break;
    case 7: qc_frobenius(r, x); break;
    case 8: qc_conj(r, x); break;
    case 9: qc_cyc_sqr(r, x); break;
    default: q_cyc_exp(r, x, park_lds + threadIdx.x); break;
  }
  qc_save(r, out + i * 144);
}

}  // namespace bls
