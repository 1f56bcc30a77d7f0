// h2c.hip.h -- batched hash-to-curve: what stands in front of `multi_miller_loop` when BLS
// signatures are verified in bulk.  One message per lane: expand_message_xmd(SHA-256) -> hash_to_field ->
// simplified SWU onto the isogenous curve -> isogeny -> (sum of the two points) -> cofactor clearing.
//
// Reference: /root/reference/src/hash_to_curve/  expand_msg.rs:230-328 (ExpandMsgXmd; DST reduction :74-95),
// mod.rs:32-49 (hash_to_field), :86-108 (hash_to_curve / encode_to_curve); map_g1.rs:513-531 (from_okm), :535-543
// (sgn0), :550-586 (map_to_curve_simple_swu), :589-630 (iso_map); map_g2.rs:374-378, :382-388, :391-454, :457-492;
// src/g1.rs:800-802 and src/g2.rs:847-947 (psi, psi2, clear_cofactor).  The formulas are the reference's, step for
// step, so even the projective coordinates of the result are the reference's (tests compare X, Y and Z); its
// fixed addition chains (chain.rs) are replaced by windowed exponentiation with the same exponents.
// XMD:SHA-256 (the BLS-signature suites) is fused into the kernels here; the other expanders of the reference (XMD:SHA-512, XOF:SHAKE128 /
// SHAKE256) go through expand.hip.h::k_expand_message and k_hash_to_curve_uniform below.
#pragma once
#include "codec.hip.h"
#include "pairlane.hip.h"
#include "expand.hip.h"        // SHA-256 / SHA-512 / SHAKE and the expanders other than XMD:SHA-256

namespace bls {

// expand_message_xmd (expand_msg.rs:247-328): ell blocks of 8 big-endian words into `out` (dst already <= 255 bytes)
DEVNI void h2c_expand_xmd(const uint8_t* __restrict__ msg, size_t mlen, const uint8_t* __restrict__ dst, u32 dlen, u32 len_in_bytes,
                          int ell, u32* out) {
  Sha256 s;
  u32 b0[8], bi[8];
  sha_init(s);
  for (int i = 0; i < 64; i++) sha_put(s, 0);                         // Z_pad: one block of zeros
  for (size_t i = 0; i < mlen; i++) sha_put(s, msg[i]);
  sha_put(s, (uint8_t)(len_in_bytes >> 8)); sha_put(s, (uint8_t)len_in_bytes); sha_put(s, 0);
  for (u32 i = 0; i < dlen; i++) sha_put(s, dst[i]);
  sha_put(s, (uint8_t)dlen);
  sha_finish(s, b0);
  for (int k = 1; k <= ell; k++) {
    u32 x[8];
    for (int j = 0; j < 8; j++) x[j] = k == 1 ? b0[j] : (b0[j] ^ bi[j]);
    sha_init(s);
    sha_put_words(s, x, 8);
    sha_put(s, (uint8_t)k);
    for (u32 i = 0; i < dlen; i++) sha_put(s, dst[i]);
    sha_put(s, (uint8_t)dlen);
    sha_finish(s, bi);
    for (int j = 0; j < 8; j++) out[8 * (k - 1) + j] = bi[j];
  }
}
// map_g1.rs:513-531: 64 uniform bytes (16 big-endian words) -> db * 2^256 + da
DEV fe h2c_from_okm(const u32* W) {
  constexpr PLimbs f256 = {BLS_H2C_F_2_256};
  u32 hi[12], lo[12];
#pragma unroll
  for (int j = 0; j < 8; j++) { hi[j] = W[7 - j]; lo[j] = W[15 - j]; }
#pragma unroll
  for (int j = 8; j < 12; j++) { hi[j] = 0; lo[j] = 0; }
  return store(add(mul(fe_from_plain(hi), fe1_const(f256)), fe_from_plain(lo)));
}
// sgn0: parity of the canonical integer (map_g1.rs:535-543, map_g2.rs:382-388)
DEV bool h2c_sgn0(const fe& a) { u32 w[12]; fe_to_plain(a, w); return (w[0] & 1u) != 0; }
DEV bool h2c_sgn0(const fe2& a) {
  u32 w0[12], w1[12]; fe_to_plain(a.c0, w0); fe_to_plain(a.c1, w1);
  u32 z = 0;
  for (int i = 0; i < 12; i++) z |= w0[i];
  return ((w0[0] & 1u) != 0) || (z == 0 && (w1[0] & 1u) != 0);
}

DEV bool h2c_sgn0(const FeP<1, VS2>& a) {                            // same rule with the coefficients on two lanes
  u32 w[12]; fe_to_plain(a.v, w);
  u32 z = 0;
  for (int i = 0; i < 12; i++) z |= w[i];
  const bool p_me = (w[0] & 1u) != 0, z_me = z == 0;
  const bool p_o = partner_flag(p_me), z_o = partner_flag(z_me);
  return lane_is_c1() ? (p_o || (z_o && p_me)) : (p_me || (z_me && p_o));
}

// ---- constant tables (internal Montgomery form, generated from the reference's literals) ----------------------
__device__ const u32 H2C_ISO11_T[55][NL] = BLS_H2C_ISO11;
__device__ const u32 H2C_ISO3_T[30][NL] = BLS_H2C_ISO3;
__device__ const u32 H2C_G2_T[16][NL] = BLS_H2C_G2_CONSTS;          // A, B, XI, RV1, ETAS[4], each (c0, c1)
DEV fe h2c_row(const u32 (*t)[NL], int i) {
  fe1 r;
#pragma unroll
  for (int k = 0; k < NL; k++) r.l[k] = t[i][k];
  return (fe)r;
}
DEV fe2 h2c_row2(const u32 (*t)[NL], int i) {
  fe2 r;
#pragma unroll
  for (int k = 0; k < NL; k++) { r.c0.l[k] = t[2 * i][k]; r.c1.l[k] = t[2 * i + 1][k]; }
  return r;
}

// Fp2 policy glue: the G2 path below is written once over F2 in {Fp2Policy (one lane), Fp2PairPolicy (lane pair)}
template <class F2> struct H2c2;
template <> struct H2c2<Fp2Policy> {
  typedef Fp2Policy G;                     // the group-level policy whose records / wire format are used
  static constexpr int LANES = 1;
  static DEV fe2 row2(const u32 (*t)[NL], int i) { return h2c_row2(t, i); }
  static DEV fe2 from_okm(const u32* W) { fe2 r; r.c0 = (Fe<1, VS2>)h2c_from_okm(W); r.c1 = (Fe<1, VS2>)h2c_from_okm(W + 16); return r; }
  static DEV fe2 konst(const PLimbs& k0, const PLimbs& k1) { fe2 K; K.c0 = (Fe<1, VS2>)fe1_const(k0); K.c1 = (Fe<1, VS2>)fe1_const(k1); return K; }
  template <class T> static DEV void save(const T& a, u32* w) { Wire<Fp2Policy>::save(a, w); }
};
template <> struct H2c2<Fp2PairPolicy> {
  typedef Fp2Policy G;
  static constexpr int LANES = 2;
  static DEV FeP<1, VS2> row2(const u32 (*t)[NL], int i) { FeP<1, VS2> r; r.v = (Fe<1, VS2>)h2c_row(t, 2 * i + (lane_is_c1() ? 1 : 0)); return r; }
  static DEV FeP<1, VS2> from_okm(const u32* W) { FeP<1, VS2> r; r.v = (Fe<1, VS2>)h2c_from_okm(W + (lane_is_c1() ? 16 : 0)); return r; }
  static DEV FeP<1, VS2> konst(const PLimbs& k0, const PLimbs& k1) { FeP<1, VS2> K; K.v = select(lane_is_c1(), (Fe<1, VS2>)fe1_const(k1), (Fe<1, VS2>)fe1_const(k0)); return K; }
  template <class T> static DEV void save(const T& a, u32* w) { fe_to_ref(a.v, w + (lane_is_c1() ? 12 : 0)); }
};

// ---- exponentiations ---------------------------------------------------------------------------------------------
// a^((p-3)/4)  (chain_pm3div4): (p-3)/4 = (p+1)/4 - 1, i.e. the square-root exponent with the last multiplication left out;
// computed directly with the generic windowed power (codec.hip.h, exponent selector 2)
DEV fe h2c_pow_pm3div4(const fe& a) { return (fe)from_v16<2>(fe_pow_raw(to_v16(a), 2)); }
// a^((p^2-9)/16) in Fp2 (chain_p2m9div16): 4-bit fixed windows over the 762-bit exponent
template <class F2> DEVNI void h2c_pow_p2m9div16(typename F2::elem& r, const typename F2::elem& a) {
  typedef typename F2::elem E;
  constexpr u64 e[12] = BLS_EXP_P2_MINUS_9_DIV_16_U64;
  E tab[15];
  tab[0] = a;
#pragma nounroll
  for (int i = 1; i < 15; i++) tab[i] = F2::st(mul(tab[i - 1], a));
  E acc = F2::one();
  bool started = false;
#pragma nounroll
  for (int w = 191; w >= 0; w--) {
    fair_tick();
    u32 d = (u32)(e[w >> 4] >> ((w & 15) * 4)) & 15u;
    if (started) { acc = F2::st(sqr(acc)); acc = F2::st(sqr(acc)); acc = F2::st(sqr(acc)); acc = F2::st(sqr(acc)); }
    if (d) { acc = started ? F2::st(mul(acc, tab[d - 1])) : tab[d - 1]; started = true; }
  }
  r = acc;
}

// ---- simplified SWU onto the isogenous curves -----------------------------------------------------------------------
// map_g1.rs:550-586
DEVNI void h2c_sswu_g1(Proj<FpPolicy>& out, const fe& u) {
  typedef FpPolicy F;
  constexpr PLimbs la = {BLS_H2C_G1_SSWU_ELLP_A}, lb = {BLS_H2C_G1_SSWU_ELLP_B}, lxi = {BLS_H2C_G1_SSWU_XI}, lsq = {BLS_H2C_G1_SQRT_M_XI_CUBED};
  const fe A = (fe)fe1_const(la), B = (fe)fe1_const(lb), XI = (fe)fe1_const(lxi), SQ = (fe)fe1_const(lsq);
  fe usq = F::st(sqr(u));
  fe xi_usq = F::st(mul(XI, usq));
  fe xisq_u4 = F::st(sqr(xi_usq));
  fe nd_common = F::st(add(xisq_u4, xi_usq));
  fe x_den = F::st(mul(A, select(is_zero(nd_common), XI, F::st(neg(nd_common)))));
  fe x0_num = F::st(mul(B, F::st(add(fe_one(), nd_common))));
  fe x_densq = F::st(sqr(x_den));
  fe gx_den = F::st(mul(x_densq, x_den));
  fe gx0_num = F::st(add(mul(F::st(add(sqr(x0_num), mul(A, x_densq))), x0_num), mul(B, gx_den)));
  fe u_v = F::st(mul(gx0_num, gx_den));
  fe vsq = F::st(sqr(gx_den));
  fe cand = F::st(mul(u_v, h2c_pow_pm3div4(F::st(mul(u_v, vsq)))));
  const bool gx0_square = el_eq(F::st(mul(F::st(sqr(cand)), gx_den)), gx0_num);
  fe x1_num = F::st(mul(x0_num, xi_usq));
  fe y1 = F::st(mul(F::st(mul(F::st(mul(SQ, usq)), u)), cand));
  fe x_num = select(gx0_square, x0_num, x1_num);
  fe y = select(gx0_square, cand, y1);
  if (h2c_sgn0(y) != h2c_sgn0(u)) y = F::st(neg(y));
  out.x = x_num; out.y = F::st(mul(y, x_den)); out.z = x_den;
}
// map_g2.rs:391-454
template <class F2> DEVNI void h2c_sswu_g2(Proj<F2>& out, const typename F2::elem& u) {
  typedef typename F2::elem E;
  typedef H2c2<F2> H;
  const E A = H::row2(H2C_G2_T, 0), B = H::row2(H2C_G2_T, 1), XI = H::row2(H2C_G2_T, 2), RV1 = H::row2(H2C_G2_T, 3);
  E usq = F2::st(sqr(u));
  E xi_usq = F2::st(mul(XI, usq));
  E xisq_u4 = F2::st(sqr(xi_usq));
  E nd_common = F2::st(add(xisq_u4, xi_usq));
  E x_den = F2::st(mul(A, select(is_zero(nd_common), XI, F2::st(neg(nd_common)))));
  E x0_num = F2::st(mul(B, F2::st(add(F2::one(), nd_common))));
  E x_densq = F2::st(sqr(x_den));
  E gx_den = F2::st(mul(x_densq, x_den));
  E gx0_num = F2::st(add(mul(F2::st(add(sqr(x0_num), mul(A, x_densq))), x0_num), mul(B, gx_den)));
  E vsq = F2::st(sqr(gx_den));
  E v_3 = F2::st(mul(vsq, gx_den));
  E v_4 = F2::st(sqr(vsq));
  E uv_7 = F2::st(mul(F2::st(mul(gx0_num, v_3)), v_4));
  E uv_15 = F2::st(mul(uv_7, F2::st(sqr(v_4))));
  E pw; h2c_pow_p2m9div16<F2>(pw, uv_15);
  E cand = F2::st(mul(uv_7, pw));
  // the candidate times each fourth root of unity (1, u, RV1 (1+u), RV1 (1-u))
  E y = cand;
  E tmp = F2::st(mul_by_u(cand));
  if (el_eq(F2::st(mul(F2::st(sqr(tmp)), gx_den)), gx0_num)) y = tmp;
  tmp = F2::st(mul(cand, RV1));
  if (el_eq(F2::st(mul(F2::st(sqr(tmp)), gx_den)), gx0_num)) y = tmp;
  tmp = F2::st(neg(F2::st(mul_by_u(tmp))));                       // (c1, -c0) = -u (c0 + c1 u)
  if (el_eq(F2::st(mul(F2::st(sqr(tmp)), gx_den)), gx0_num)) y = tmp;
  E gx1_num = F2::st(mul(F2::st(mul(gx0_num, xi_usq)), xisq_u4));
  E sc = F2::st(mul(F2::st(mul(cand, usq)), u));
  bool eta_found = false;
#pragma nounroll
  for (int k = 0; k < 4; k++) {
    E t = F2::st(mul(sc, H::row2(H2C_G2_T, 4 + k)));
    bool found = el_eq(F2::st(mul(F2::st(sqr(t)), gx_den)), gx1_num);
    if (found) y = t;
    eta_found = eta_found || found;
  }
  E x_num = eta_found ? F2::st(mul(x0_num, xi_usq)) : x0_num;
  if (h2c_sgn0(u) != h2c_sgn0(y)) y = F2::st(neg(y));
  out.x = x_num; out.y = F2::st(mul(y, x_den)); out.z = x_den;
}

// ---- the two maps of a message at once, on a lane pair (round 6) ---------------------------------------------------------------
// hash_to_curve maps TWO field elements per message (mod.rs:86-96).  The reference finds the square root of g(x) = U/V by raising u v^15 to
// (p^2 - 9)/16 in Fp2 (192 windows of four Fp2 squarings and a multiplication) and trying eight roots of unity; what it RETURNS does not depend
// on the method: x = x0 when U/V is a square in Fp2, else x1 = xi u^2 x0; y = the root of g(x) whose sgn0 equals sgn0(u) (a field element has
// two roots, the sign rule picks one); and y = 0, x = x1 whenever gx1_num = 0 (u = 0 or U = 0: every candidate of the reference's second loop is
// zero and "matches").  Any exact square root serves, and one built from BASE-FIELD exponentiations lets the two lanes of a pair work on the
// two maps at the same time -- lane 0 on the first field element, lane 1 on the second -- instead of sharing every product of one Fp2 chain:
//   n = norm(U V), e = n^((p+1)/4):  e^2 = n  <=>  U/V is a square;  otherwise e^2 = -n and norm(gx1_num V) = n norm(xi)^3 norm(u)^6 has the
//   root e sqrt(-125) norm(u)^3  (xi = -(2 + u), norm 5; -125 is a residue mod p);
//   sqrt(a), a = U V or gx1_num V, from the root s of its norm as codec.hip.h fe2_sqrt does: X = 2 (a0 + s) (a0 when a1 = 0),
//   g = X^((p-3)/4), f = g X, f g = +-1 the quadratic character of X (so 1/f = +-g):  f^2 = X: (f/2, a1/f),  f^2 = -X: (a1/f, f/2);
//   y = sqrt(a) / V, with 1/V = conj(V) / norm(V) and the base-field inverse taken in the same per-lane phase.
// The Fp2 arithmetic before and after stays pair-cooperative.  A single map built this way is SLOWER than the reference's chain on lane pairs
// (both lanes would run the same base-field chain: 8.7 -> 9.9 ms); two maps side by side: hash-to-G2 of 2^16 messages 8.7 -> see
// profiles/r06_fair_tick.md.  Checked against the oracle's restatement of the reference on random and degenerate u and by every vector.
struct SswuMid { FeP<1, VS2> u, xi_usq, x_den, x0_num, gx_den, a0v, a1v; fe N, NV; bool g1zero; };
DEV fe h2c_pnorm(const FeP<1, VS2>& z) { const fe own = store(z.v); auto q = sqr(own); return store(add(q, partner(q))); }      // norm, in both lanes
DEV void h2c_sswu_g2_pre(SswuMid& m, const FeP<1, VS2>& u) {
  typedef Fp2PairPolicy F2;
  typedef F2::elem E;
  typedef H2c2<F2> H;
  const E A = H::row2(H2C_G2_T, 0), B = H::row2(H2C_G2_T, 1), XI = H::row2(H2C_G2_T, 2);
  E usq = F2::st(sqr(u));
  E xi_usq = F2::st(mul(XI, usq));
  E xisq_u4 = F2::st(sqr(xi_usq));
  E nd_common = F2::st(add(xisq_u4, xi_usq));
  E x_den = F2::st(mul(A, select(is_zero(nd_common), XI, F2::st(neg(nd_common)))));
  E x0_num = F2::st(mul(B, F2::st(add(F2::one(), nd_common))));
  E x_densq = F2::st(sqr(x_den));
  E gx_den = F2::st(mul(x_densq, x_den));
  E gx0_num = F2::st(add(mul(F2::st(add(sqr(x0_num), mul(A, x_densq))), x0_num), mul(B, gx_den)));
  E gx1_num = F2::st(mul(F2::st(mul(gx0_num, xi_usq)), xisq_u4));
  m.u = u; m.xi_usq = xi_usq; m.x_den = x_den; m.x0_num = x0_num; m.gx_den = gx_den;
  m.g1zero = is_zero(gx1_num);
  m.a0v = F2::st(mul(gx0_num, gx_den)); m.a1v = F2::st(mul(gx1_num, gx_den));
  m.N = h2c_pnorm(u); m.NV = h2c_pnorm(gx_den);
}
struct SswuRoot { fe lo, hi, vinv; bool qr, sq0; };
// base field only: a = (A00, A01) = U V, a' = (A10, A11) = gx1_num V, N = norm(u), NV = norm(V)
DEV void h2c_sswu_g2_root(SswuRoot& r, const fe& A00, const fe& A01, const fe& A10, const fe& A11, const fe& N, const fe& NV) {
  constexpr PLimbs csq = {BLS_SQRT_M125_MONT}, half = {BLS_TWO_INV_MONT};
  const fe n = store(add(sqr(A00), sqr(A01)));
  bool sq0;
  const fe e = (fe)fe_sqrt(n, sq0);
  const fe N3 = store(mul(store(sqr(N)), N));
  const fe s1 = store(mul(store(mul(e, fe1_const(csq))), N3));
  const fe s = select(sq0, e, s1);
  const fe a0 = select(sq0, A00, A10), a1 = select(sq0, A01, A11);
  const bool real = is_zero(a1);
  const fe X = select(real, a0, store(dbl(add(a0, s))));
  const fe g = h2c_pow_pm3div4(X);
  const fe f = store(mul(g, X));
  const bool qr = fe_eq(store(mul(f, g)), fe_one()) || is_zero(X);
  const fe h = store(mul(f, fe1_const(half)));
  const fe w = store(mul(a1, select(qr, g, store(neg(g)))));
  r.lo = select(real, f, h); r.hi = select(real, fe_zero(), w);
  r.vinv = (fe)inv(NV);
  r.qr = qr; r.sq0 = sq0;
}
DEV void h2c_sswu_g2_post(Proj<Fp2PairPolicy>& out, const SswuMid& m, const fe& lo, const fe& hi, const fe& vinv, bool qr, bool sq0) {
  typedef Fp2PairPolicy F2;
  typedef F2::elem E;
  const bool c1 = lane_is_c1();
  E r; r.v = (Fe<1, VS2>)select(qr != c1, lo, hi);                         // lane c0: qr ? lo : hi;  lane c1: qr ? hi : lo
  E vi; vi.v = (Fe<1, VS2>)store(mul(select(c1, store(neg(m.gx_den.v)), store(m.gx_den.v)), vinv));      // 1/V = conj(V) / norm(V)
  E y = F2::st(mul(r, vi));
  if (m.g1zero) y = F2::zero();
  const bool eta_found = !sq0 || m.g1zero;
  E x_num = eta_found ? F2::st(mul(m.x0_num, m.xi_usq)) : m.x0_num;
  if (h2c_sgn0(m.u) != h2c_sgn0(y)) y = F2::st(neg(y));
  out.x = x_num; out.y = F2::st(mul(y, m.x_den)); out.z = m.x_den;
}
DEVNI void h2c_sswu_g2_dual(Proj<Fp2PairPolicy>& o0, Proj<Fp2PairPolicy>& o1, const FeP<1, VS2>& u0, const FeP<1, VS2>& u1) {
  SswuMid m0, m1;
  h2c_sswu_g2_pre(m0, u0);
  h2c_sswu_g2_pre(m1, u1);
  const bool c1 = lane_is_c1();
  // lane L works on map L: its own coefficient of map L's value, and the other coefficient from the partner (which sends what THIS lane's map needs)
  const fe a0_own = select(c1, store(m1.a0v.v), store(m0.a0v.v));
  const fe a0_oth = partner(select(c1, store(m0.a0v.v), store(m1.a0v.v)));
  const fe a1_own = select(c1, store(m1.a1v.v), store(m0.a1v.v));
  const fe a1_oth = partner(select(c1, store(m0.a1v.v), store(m1.a1v.v)));
  // lane 0 owns the c0 coefficient, lane 1 the c1 coefficient
  SswuRoot rt;
  h2c_sswu_g2_root(rt, select(c1, a0_oth, a0_own), select(c1, a0_own, a0_oth), select(c1, a1_oth, a1_own), select(c1, a1_own, a1_oth),
                   select(c1, m1.N, m0.N), select(c1, m1.NV, m0.NV));
  // map 0's root was computed by lane 0, map 1's by lane 1: each lane receives the other map's
  const fe lo_x = partner(rt.lo), hi_x = partner(rt.hi), vinv_x = partner(rt.vinv);
  const bool qr_x = partner_flag(rt.qr), sq_x = partner_flag(rt.sq0);
  h2c_sswu_g2_post(o0, m0, select(c1, lo_x, rt.lo), select(c1, hi_x, rt.hi), select(c1, vinv_x, rt.vinv), c1 ? qr_x : rt.qr, c1 ? sq_x : rt.sq0);
  h2c_sswu_g2_post(o1, m1, select(c1, rt.lo, lo_x), select(c1, rt.hi, hi_x), select(c1, rt.vinv, vinv_x), c1 ? rt.qr : qr_x, c1 ? rt.sq0 : sq_x);
}

// ---- isogenies (map_g1.rs:589-630, map_g2.rs:457-492): Horner in x with powers of z --------------------------------
template <class F> struct H2cIso;
template <> struct H2cIso<FpPolicy> {
  static constexpr int NZ = 15;
  static constexpr int LEN[4] = {12, 11, 16, 16};
  static DEV fe coeff(int base, int k) { return h2c_row(H2C_ISO11_T, base + k); }
};
template <> struct H2cIso<Fp2Policy> {
  static constexpr int NZ = 3;
  static constexpr int LEN[4] = {4, 3, 4, 4};
  static DEV fe2 coeff(int base, int k) { return h2c_row2(H2C_ISO3_T, base + k); }
};
template <> struct H2cIso<Fp2PairPolicy> {
  static constexpr int NZ = 3;
  static constexpr int LEN[4] = {4, 3, 4, 4};
  static DEV FeP<1, VS2> coeff(int base, int k) { return H2c2<Fp2PairPolicy>::row2(H2C_ISO3_T, base + k); }
};
template <class F>
DEVNI void h2c_iso_map(Proj<F>& out, const Proj<F>& in) {
  typedef typename F::elem E;
  typedef H2cIso<F> I;
  E zpows[I::NZ];
  zpows[0] = in.z;
#pragma nounroll
  for (int j = 1; j < I::NZ; j++) zpows[j] = F::st(mul(zpows[j - 1], in.z));
  E mapvals[4];
  int base = 0;
#pragma nounroll
  for (int idx = 0; idx < 4; idx++) {
    const int clast = I::LEN[idx] - 1;
    E v = I::coeff(base, clast);
#pragma nounroll
    for (int j = 0; j < clast; j++) v = F::st(add(mul(v, in.x), mul(zpows[j], I::coeff(base, clast - 1 - j))));
    mapvals[idx] = v;
    base += I::LEN[idx];
  }
  mapvals[1] = F::st(mul(mapvals[1], in.z));
  mapvals[2] = F::st(mul(mapvals[2], in.y));
  mapvals[3] = F::st(mul(mapvals[3], in.z));
  out.x = F::st(mul(mapvals[0], mapvals[3]));
  out.y = F::st(mul(mapvals[2], mapvals[1]));
  out.z = F::st(mul(mapvals[1], mapvals[3]));
}

// ---- cofactor clearing ------------------------------------------------------------------------------------------------
// g1.rs:800-802: self - [x] self
DEVNI void h2c_clear_cofactor(Proj<FpPolicy>& out, const Proj<FpPolicy>& p) {
  Proj<FpPolicy> t;
  pt_mul_by_x<FpPolicy>(t, p);
  out = pt_add<FpPolicy>(p, pt_neg<FpPolicy>(t));
}
// g2.rs:847-890 / :890-912
template <class F2> DEV Proj<F2> pt_psi(const Proj<F2>& p) {
  constexpr PLimbs zero = {{0}}, px1 = {BLS_PSI_X_1}, py0 = {BLS_PSI_Y_0}, py1 = {BLS_PSI_Y_1};
  const typename F2::elem cx = H2c2<F2>::konst(zero, px1), cy = H2c2<F2>::konst(py0, py1);
  Proj<F2> s;
  s.x = F2::st(mul(F2::st(conj(p.x)), cx));
  s.y = F2::st(mul(F2::st(conj(p.y)), cy));
  s.z = F2::st(conj(p.z));
  return s;
}
template <class F2> DEV Proj<F2> pt_psi2(const Proj<F2>& p) {
  constexpr PLimbs k = {BLS_PSI2_X};
  Proj<F2> s;
  s.x = F2::st(mul_fp(p.x, fe1_const(k)));
  s.y = F2::st(neg(p.y));
  s.z = p.z;
  return s;
}
// g2.rs:938-947
template <class F2> DEVNI void h2c_clear_cofactor_g2(Proj<F2>& out, const Proj<F2>& p) {
  Proj<F2> t1, t2 = pt_psi<F2>(p), t3;
  pt_mul_by_x<F2>(t1, p);
  Proj<F2> r = pt_psi2<F2>(pt_double<F2>(p));
  pt_mul_by_x<F2>(t3, pt_add<F2>(t1, t2));
  r = pt_add<F2>(r, t3);
  r = pt_add<F2>(r, pt_neg<F2>(t1));
  r = pt_add<F2>(r, pt_neg<F2>(t2));
  out = pt_add<F2>(r, pt_neg<F2>(p));
}

template <class F> struct H2cField;
template <> struct H2cField<FpPolicy> {
  static constexpr int M = 1, LANES = 1;
  static DEV fe from_okm(const u32* W) { return h2c_from_okm(W); }
  static DEV void sswu(Proj<FpPolicy>& o, const fe& u) { h2c_sswu_g1(o, u); }
  static DEV void clear(Proj<FpPolicy>& o, const Proj<FpPolicy>& p) { h2c_clear_cofactor(o, p); }
  template <class T> static DEV void save(const T& a, u32* w) { Wire<FpPolicy>::save(a, w); }
};
template <> struct H2cField<Fp2Policy> {
  static constexpr int M = 2, LANES = 1;
  static DEV fe2 from_okm(const u32* W) { return H2c2<Fp2Policy>::from_okm(W); }
  static DEV void sswu(Proj<Fp2Policy>& o, const fe2& u) { h2c_sswu_g2<Fp2Policy>(o, u); }
  static DEV void clear(Proj<Fp2Policy>& o, const Proj<Fp2Policy>& p) { h2c_clear_cofactor_g2<Fp2Policy>(o, p); }
  template <class T> static DEV void save(const T& a, u32* w) { Wire<Fp2Policy>::save(a, w); }
};
template <> struct H2cField<Fp2PairPolicy> {
  static constexpr int M = 2, LANES = 2;
  static DEV FeP<1, VS2> from_okm(const u32* W) { return H2c2<Fp2PairPolicy>::from_okm(W); }
  static DEV void sswu(Proj<Fp2PairPolicy>& o, const FeP<1, VS2>& u) { h2c_sswu_g2<Fp2PairPolicy>(o, u); }
  static DEV void clear(Proj<Fp2PairPolicy>& o, const Proj<Fp2PairPolicy>& p) { h2c_clear_cofactor_g2<Fp2PairPolicy>(o, p); }
  template <class T> static DEV void save(const T& a, u32* w) { H2c2<Fp2PairPolicy>::save(a, w); }
};

// mod.rs:86-108 behind the expander: the uniform bytes `ub` (big-endian words) -> field elements -> SSWU -> isogeny -> (sum) -> cofactor clearing -> o
template <class F> DEV void h2c_map_and_store(const u32* ub, int encode_only, u32* o) {
  constexpr int M = H2cField<F>::M, WW = M * 12;
  Proj<F> q, t;
  if constexpr (std::is_same<F, Fp2PairPolicy>::value) {
    if (!encode_only) {
      // both field elements at once: the base-field chains of the two maps run on the two lanes (h2c_sswu_g2_dual)
      Proj<F> t1, q1;
      h2c_sswu_g2_dual(t, t1, H2cField<F>::from_okm(ub), H2cField<F>::from_okm(ub + 16 * M));
      h2c_iso_map<F>(q, t);
      h2c_iso_map<F>(q1, t1);
      q = pt_add<F>(q, q1);
    } else {
      H2cField<F>::sswu(t, H2cField<F>::from_okm(ub));
      h2c_iso_map<F>(q, t);
    }
  } else {
  H2cField<F>::sswu(t, H2cField<F>::from_okm(ub));
  h2c_iso_map<F>(q, t);
  if (!encode_only) {
    Proj<F> q1;
    H2cField<F>::sswu(t, H2cField<F>::from_okm(ub + 16 * M));
    h2c_iso_map<F>(q1, t);
    q = pt_add<F>(q, q1);
  }
  }
  Proj<F> r;
  H2cField<F>::clear(r, q);
  H2cField<F>::save(r.x, o);
  H2cField<F>::save(r.y, o + WW);
  H2cField<F>::save(r.z, o + 2 * WW);
}

// mod.rs:86-108.  msgs = the messages back to back, offs[i] .. offs[i+1] the bytes of message i; dst <= 255 bytes.
// out[i] = projective (X : Y : Z) in wire limbs.  encode_only: one field element (the *_NU_ suites).
// F = FpPolicy: one message per lane; F = Fp2PairPolicy: one message per lane PAIR (both lanes hash the message, each keeps
// its coefficient of every Fp2 value); F = Fp2Policy: the one-lane G2 form (kept for cross-checking).
template <class F>
__global__ void __launch_bounds__(H2cField<F>::LANES == 2 ? 256 : 64, H2cField<F>::LANES == 2 ? 2 : 1)
k_hash_to_curve(const uint8_t* __restrict__ msgs, const unsigned long long* __restrict__ offs, size_t n, const uint8_t* __restrict__ dst, u32 dlen,
                int encode_only, u32* __restrict__ out) {
  fair_init();
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / H2cField<F>::LANES;
  if (i >= n) return;
  constexpr int M = H2cField<F>::M, WW = M * 12;
  const int count = encode_only ? 1 : 2;
  const int ell = count * M * 2;
  u32 ub[64];
  h2c_expand_xmd(msgs + offs[i], (size_t)(offs[i + 1] - offs[i]), dst, dlen, (u32)(ell * 32), ell, ub);
  h2c_map_and_store<F>(ub, encode_only, out + i * 3 * WW);
}
// the same from uniform bytes that another expander produced (expand.hip.h::k_expand_message: XMD:SHA-512, XOF:SHAKE128 / SHAKE256 -- or XMD:SHA-256,
// which then gives the very limbs of the fused kernel): message i owns count * M * 64 bytes at uniform + i * that, count = encode_only ? 1 : 2
template <class F>
__global__ void __launch_bounds__(H2cField<F>::LANES == 2 ? 256 : 64, H2cField<F>::LANES == 2 ? 2 : 1)
k_hash_to_curve_uniform(const uint8_t* __restrict__ uniform, size_t n, int encode_only, u32* __restrict__ out) {
  fair_init();
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / H2cField<F>::LANES;
  if (i >= n) return;
  constexpr int M = H2cField<F>::M, WW = M * 12;
  const int words = (encode_only ? 1 : 2) * M * 16;
  const uint8_t* u = uniform + i * (size_t)words * 4;
  u32 ub[64];
  for (int k = 0; k < words; k++) ub[k] = ((u32)u[4 * k] << 24) | ((u32)u[4 * k + 1] << 16) | ((u32)u[4 * k + 2] << 8) | (u32)u[4 * k + 3];      // big-endian words, as h2c_expand_xmd leaves them
  h2c_map_and_store<F>(ub, encode_only, out + i * 3 * WW);
}

// ---- small batches (round 5): the two maps of one message on two lane groups ---------------------------------------------------------
// hash_to_curve maps TWO field elements u0, u1 to the isogenous curve independently (mod.rs:93-99) before it adds the images and clears
// the cofactor; each map carries a square-root-ratio power (~560 / ~2 100 multiplications of the ~2 400 / ~8 700 per hash).  With few
// messages the chip is not full (2^14 hashes to G2 = 512 wavefronts on 1 024 SIMDs) and a hash costs its LATENCY: here message i takes
// 2 x LANES lanes, group 0 maps u0 and group 1 maps u1 side by side, group 1 hands its image over with one DPP move per word
// (LANES = 1: quad_perm:[1,0,3,2], LANES = 2: quad_perm:[2,3,0,1]) and retires; group 0 adds, clears the cofactor and stores.  Same
// field elements in the same order as k_hash_to_curve, so the projective limbs are identical.  The host picks it for batches that leave
// the chip under-filled (api_aux.hip::h2c_launch); `BLSGPU_H2C_SPLIT=0|1` forces either form.
template <int CTRL> DEV u32 h2c_dpp(u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, true); }
template <int CTRL, int A, int V> DEV Fe<A, V> h2c_partner(const Fe<A, V>& a) {
  Fe<A, V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = h2c_dpp<CTRL>(a.l[i]);
  return r;
}
template <int CTRL, int A, int V> DEV FeP<A, V> h2c_partner(const FeP<A, V>& a) { FeP<A, V> r; r.v = h2c_partner<CTRL>(a.v); return r; }
template <class F>
__global__ void __launch_bounds__(H2cField<F>::LANES == 2 ? 256 : 64, H2cField<F>::LANES == 2 ? 2 : 1)
k_hash_to_curve_split(const uint8_t* __restrict__ msgs, const unsigned long long* __restrict__ offs, size_t n, const uint8_t* __restrict__ dst, u32 dlen,
                      u32* __restrict__ out) {
  fair_init();
  constexpr int L = H2cField<F>::LANES, CTRL = L == 2 ? 0x4E : 0xB1;
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / (2 * L);
  if (i >= n) return;
  const bool second = (threadIdx.x & L) != 0;
  constexpr int M = H2cField<F>::M, WW = M * 12;
  const int ell = 2 * M * 2;
  u32 ub[64];
  h2c_expand_xmd(msgs + offs[i], (size_t)(offs[i + 1] - offs[i]), dst, dlen, (u32)(ell * 32), ell, ub);
  Proj<F> q, t;
  H2cField<F>::sswu(t, H2cField<F>::from_okm(ub + (second ? 16 * M : 0)));
  h2c_iso_map<F>(q, t);
  Proj<F> q1;
  q1.x = h2c_partner<CTRL>(q.x); q1.y = h2c_partner<CTRL>(q.y); q1.z = h2c_partner<CTRL>(q.z);
  if (second) return;
  q = pt_add<F>(q, q1);
  Proj<F> r;
  H2cField<F>::clear(r, q);
  H2cField<F>::save(r.x, out + i * 3 * WW);
  H2cField<F>::save(r.y, out + i * 3 * WW + WW);
  H2cField<F>::save(r.z, out + i * 3 * WW + 2 * WW);
}

}  // namespace bls
