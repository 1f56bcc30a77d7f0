// codec.hip.h -- batched point (de)serialisation and validation: the step in FRONT of the hot path
// (SURVEY.md 8f rank 1).  Real MSM / pairing inputs arrive as compressed bytes; decompression (a square
// root per point) and the subgroup check (a scalar multiplication by the curve parameter per point) are
// themselves data-parallel, one point per lane.
//
// Reference: encodings src/notes/serialization.rs:1-29; G1 src/g1.rs:221-260 (to_compressed /
// to_uncompressed), :264-322 (from_uncompressed[_unchecked]), :326-390 (from_compressed[_unchecked]),
// :401-410 (is_torsion_free: endomorphism(P) == -[x^2]P), :412-416 (is_on_curve), :777-795 (mul_by_x);
// G2 src/g2.rs:254-299, :303-380, :390-464, :475-489, :847-890 (psi), :914-931 (mul_by_x);
// Fp::sqrt src/fp.rs:324-340, Fp2::sqrt src/fp2.rs:245-295 (any root: see fe2_sqrt), lexicographically_largest src/fp.rs:273-298,
// src/fp2.rs:171-180, Fp::from_bytes / to_bytes src/fp.rs:179-227.
// Outputs are the reference's values: a decoded point is returned in wire limbs with its infinity flag and
// an `ok` byte that is 1 exactly where the reference returns `CtOption::some`.
#pragma once
#include "convert.hip.h"

namespace bls {

// x^e for a fixed 384-bit exponent (4-bit fixed window, the table is indexed at run time and lives in scratch)
DEVNI v16 fe_pow_raw(v16 xin, int which) {
  // which: 0 = p - 2, 1 = (p + 1) / 4 (square root), 2 = (p - 3) / 4 (hash-to-curve's chain_pm3div4)
  constexpr u64 e_inv[6] = BLS_P_MINUS_2_U64, e_sqrt[6] = BLS_EXP_SQRT_U64, e_pm3[6] = BLS_EXP_P_MINUS_3_DIV_4_U64;
  fe x = (fe)from_v16<VS>(xin);
  fe tab[15];
  tab[0] = x;
#pragma nounroll
  for (int i = 1; i < 15; i++) tab[i] = (fe)mul(tab[i - 1], x);
  fe acc = fe_one();
  bool started = false;
#pragma nounroll
  for (int w = 95; w >= 0; w--) {
    fair_tick();
    u64 word = which == 0 ? e_inv[w >> 4] : which == 1 ? e_sqrt[w >> 4] : e_pm3[w >> 4];
    u32 d = (u32)(word >> ((w & 15) * 4)) & 15u;
    if (started) { acc = (fe)sqr(acc); acc = (fe)sqr(acc); acc = (fe)sqr(acc); acc = (fe)sqr(acc); }
    if (d) {
      fe t = tab[d - 1];
      acc = started ? (fe)mul(acc, t) : t;
      started = true;
    }
  }
  return to_v16(acc);
}
// fp.rs:324-340: candidate root a^((p+1)/4); `ok` = it squares back to a
template <int A, int V>
DEV fe2p fe_sqrt(const Fe<A, V>& a, bool& ok) {
  static_assert(V <= VS, "fe_sqrt: reduce first");
  fe2p s = from_v16<2>(fe_pow_raw(to_v16(norm(a)), 1));
  ok = fe_eq(sqr(s), a);
  return s;
}

// A square root in Fp2 = Fp[u]/(u^2 + 1).  The reference computes one by two Fp2 exponentiations (fp2.rs:245-295, Algorithm 9 of eprint
// 2012/685: a^((p-3)/4), then (alpha + 1)^((p-1)/2); ~2 700 base-field multiplications); the decoders below only ever use a root up to
// SIGN -- `from_compressed` picks y or -y by the sign flag (g2.rs:436-452) -- so any root gives the reference's point, and this one
// costs two BASE-FIELD exponentiations and an inversion (~1 100 multiplications; round 5).  With a = a0 + a1 u, n = sqrt(a0^2 + a1^2)
// (the norm is a square in Fp exactly when a is one in Fp2) and D = 2 (a0 + n): e = D^((p+1)/4) squares to D or to -D (p = 3 mod 4), and
//     e^2 =  D:  sqrt(a) = e/2 + (a1/e) u            e^2 = -D:  sqrt(a) = a1/e + (e/2) u
// (write delta = (a0 + n)/2, delta' = (a0 - n)/2: delta + delta' = a0, delta delta' = -(a1/2)^2, so exactly one of them is a square and
// the root is sqrt(delta) + a1 / (2 sqrt(delta)) u or sqrt(delta') + ... with sqrt(delta') = (a1/2) / sqrt(-delta)).  a1 = 0: the same
// exponentiation on a0 itself gives (e, 0) or (0, e).  `ok` = the result squares back to a, as in the reference.
DEV fe2 fe2_sqrt(const fe2& a, bool& ok) {
  if (is_zero(a)) { ok = true; return fe2_zero(); }
  constexpr PLimbs half = {BLS_TWO_INV_MONT};
  const fe a0 = store(a.c0), a1 = store(a.c1);
  const bool real = is_zero(a1);
  bool okn, qr;
  const fe n = (fe)fe_sqrt(store(add(sqr(a0), sqr(a1))), okn);
  const fe X = select(real, a0, store(dbl(add(a0, n))));
  const fe e = (fe)fe_sqrt(X, qr);
  const fe h = store(mul(e, fe1_const(half)));
  const fe w = store(mul(a1, inv(e)));
  const fe lo = select(real, e, h), hi = select(real, fe_zero(), w);
  fe2 s;
  s.c0 = (Fe<1, VS2>)select(qr, lo, hi);
  s.c1 = (Fe<1, VS2>)select(qr, hi, lo);
  fe2 sq = store2(sqr(s));
  ok = fe_eq(sq.c0, a.c0) && fe_eq(sq.c1, a.c1);
  return s;
}

// ---- byte <-> field ------------------------------------------------------------------------------------
// 48 big-endian bytes (flag bits already masked) -> canonical internal element; false if the integer is >= p
DEV bool fe_from_be(const uint8_t* b, fe1& out) {
  constexpr u32 pw[12] = BLS_P_WORDS;
  u32 w[12];
#pragma unroll
  for (int j = 0; j < 12; j++) {
    const uint8_t* q = b + 44 - 4 * j;
    w[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
  // canonical iff w < p  (fp.rs:190-199)
  bool lt = false, eq = true;
#pragma unroll
  for (int j = 11; j >= 0; j--) { lt = lt || (eq && w[j] < pw[j]); eq = eq && (w[j] == pw[j]); }
  out = fe_from_plain(w);
  return lt;
}
template <int A, int V>
DEV void fe_to_be(const Fe<A, V>& a, uint8_t* b, u32* plain_out = nullptr) {
  u32 w[12];
  fe_to_plain(a, w);
#pragma unroll
  for (int j = 0; j < 12; j++) {
    uint8_t* q = b + 44 - 4 * j;
    q[0] = (uint8_t)(w[j] >> 24); q[1] = (uint8_t)(w[j] >> 16); q[2] = (uint8_t)(w[j] >> 8); q[3] = (uint8_t)w[j];
    if (plain_out) plain_out[j] = w[j];
  }
}
// fp.rs:273-298: the canonical integer is > (p-1)/2
template <int A, int V>
DEV bool fe_lex_largest(const Fe<A, V>& a) {
  constexpr u32 hw[12] = BLS_HALF_P_WORDS;
  u32 w[12];
  fe_to_plain(a, w);
  bool gt = false, eq = true;
#pragma unroll
  for (int j = 11; j >= 0; j--) { gt = gt || (eq && w[j] > hw[j]); eq = eq && (w[j] == hw[j]); }
  return gt;
}
// fp2.rs:171-180
template <int A, int V>
DEV bool fe2_lex_largest(const Fe2<A, V>& a) { return fe_lex_largest(a.c1) || (is_zero(a.c1) && fe_lex_largest(a.c0)); }

// ---- policy glue ---------------------------------------------------------------------------------------------
template <class F> struct Codec;
template <> struct Codec<FpPolicy> {
  static constexpr int COORD_BYTES = 48;
  static DEV bool parse(const uint8_t* b, uint8_t first, fe1& out) {
    uint8_t tmp[48];
#pragma unroll
    for (int i = 0; i < 48; i++) tmp[i] = b[i];
    tmp[0] = first;
    return fe_from_be(tmp, out);
  }
  template <class T> static DEV void emit(const T& a, uint8_t* b) { fe_to_be(a, b); }
  template <class T> static DEV bool lex_largest(const T& a) { return fe_lex_largest(a); }
  static DEV fe sqrt(const fe& a, bool& ok) { return (fe)fe_sqrt(a, ok); }
  static DEV fe b_coeff() { constexpr PLimbs c = {BLS_FOUR_MONT}; return fe_const(c); }
};
template <> struct Codec<Fp2Policy> {
  static constexpr int COORD_BYTES = 96;
  // c1 first, then c0 (g2.rs:284-299)
  static DEV bool parse(const uint8_t* b, uint8_t first, fe2_1& out) {
    uint8_t tmp[48];
#pragma unroll
    for (int i = 0; i < 48; i++) tmp[i] = b[i];
    tmp[0] = first;
    bool ok1 = fe_from_be(tmp, out.c1);
    bool ok0 = fe_from_be(b + 48, out.c0);
    return ok0 && ok1;
  }
  template <class T> static DEV void emit(const T& a, uint8_t* b) { fe_to_be(a.c1, b); fe_to_be(a.c0, b + 48); }
  template <class T> static DEV bool lex_largest(const T& a) { return fe2_lex_largest(a); }
  static DEV fe2 sqrt(const fe2& a, bool& ok) { return fe2_sqrt(a, ok); }
  static DEV fe2 b_coeff() { constexpr PLimbs c = {BLS_FOUR_MONT}; fe2 r; r.c0 = (Fe<1, VS2>)fe_const(c); r.c1 = r.c0; return r; }
};

template <class T1, class T2> DEV bool el_eq(const T1& a, const T2& b) { return is_zero(sub(a, b)); }

// y^2 == x^3 + b   (g1.rs:412-416, g2.rs:484-489)
template <class F, class XT, class YT>
DEV bool on_curve(const XT& x, const YT& y) {
  auto lhs = sqr(y);
  auto rhs = add(mul(sqr(x), x), Codec<F>::b_coeff());
  return el_eq(lhs, rhs);
}

// multiply by the (negative) curve parameter x   (g1.rs:777-795, g2.rs:914-931)
template <class F>
DEVNI void pt_mul_by_x(Proj<F>& out, const Proj<F>& p) {
  constexpr u64 XH = 0xd201000000010000ull >> 1;
  Proj<F> xself = pt_identity<F>(), tmp = p;
  for (int i = 0; i < 63; i++) {
    fair_tick();
    tmp = pt_double<F>(tmp);
    if ((XH >> i) & 1) xself = pt_add<F>(xself, tmp);
  }
  out = pt_neg<F>(xself);
}
// projective equality (g1.rs:479-496)
template <class F>
DEV bool pt_eq(const Proj<F>& a, const Proj<F>& b) {
  bool az = is_zero(a.z), bz = is_zero(b.z);
  bool xe = el_eq(mul(a.x, b.z), mul(b.x, a.z));
  bool ye = el_eq(mul(a.y, b.z), mul(b.y, a.z));
  return (az && bz) || (!az && !bz && xe && ye);
}
template <class F> DEV Proj<F> proj_from_affine(const typename F::elem& x, const typename F::elem& y, bool inf) {
  Proj<F> p; p.x = x; p.y = y; p.z = inf ? F::zero() : F::one(); return p;
}
// g1.rs:401-410
DEV bool torsion_free(const fe& x, const fe& y, bool inf) {
  constexpr PLimbs bl = {BLS_BETA};
  Proj<FpPolicy> p = proj_from_affine<FpPolicy>(x, y, inf), t, m;
  pt_mul_by_x<FpPolicy>(t, p);
  pt_mul_by_x<FpPolicy>(m, t);
  m = pt_neg<FpPolicy>(m);
  Proj<FpPolicy> e = proj_from_affine<FpPolicy>(store(mul(x, fe1_const(bl))), y, inf);
  return pt_eq<FpPolicy>(m, e);
}
// g2.rs:475-482 with psi of g2.rs:847-890
DEV bool torsion_free(const fe2& x, const fe2& y, bool inf) {
  constexpr PLimbs px1 = {BLS_PSI_X_1}, py0 = {BLS_PSI_Y_0}, py1 = {BLS_PSI_Y_1};
  Proj<Fp2Policy> p = proj_from_affine<Fp2Policy>(x, y, inf), m;
  pt_mul_by_x<Fp2Policy>(m, p);
  fe2 cx; cx.c0 = (Fe<1, VS2>)fe_zero(); cx.c1 = (Fe<1, VS2>)fe1_const(px1);
  fe2 cy; cy.c0 = (Fe<1, VS2>)fe1_const(py0); cy.c1 = (Fe<1, VS2>)fe1_const(py1);
  Proj<Fp2Policy> s;
  s.x = store2(mul(store2(conj(p.x)), cx));
  s.y = store2(mul(store2(conj(p.y)), cy));
  s.z = store2(conj(p.z));
  return pt_eq<Fp2Policy>(s, m);
}

template <class F> DEV void emit_identity_wire(u32* xy) {
  constexpr int WW = Wire<F>::WORDS;
  Wire<F>::save(F::zero(), xy);
  Wire<F>::save(F::one(), xy + WW);
}

// ---- kernels ---------------------------------------------------------------------------------------------------
// mode bit 0: compressed input; bit 1: checked variant (from_compressed / from_uncompressed), else *_unchecked
template <class F>
DEV void point_decode_one(size_t i, const uint8_t* __restrict__ in, int mode, u32* __restrict__ xy, uint8_t* __restrict__ inf_out, uint8_t* __restrict__ ok_out) {
  constexpr int CB = Codec<F>::COORD_BYTES, WW = Wire<F>::WORDS;
  const bool compressed = mode & 1, checked = (mode & 2) != 0;
  const uint8_t* b = in + i * (compressed ? CB : 2 * CB);
  const uint8_t first = b[0];
  const bool cflag = (first >> 7) & 1, iflag = (first >> 6) & 1, sflag = (first >> 5) & 1;
  typename F::aff_elem x1, y1;
  bool okx = Codec<F>::parse(b, first & 0x1f, x1);
  typename F::elem x = F::st(x1), y = F::zero();
  bool ok, inf = false;
  if (compressed) {
    if (iflag && cflag && !sflag && is_zero(x)) {       // g1.rs:356-364
      ok = okx; inf = true;
    } else {
      bool oks;
      typename F::elem rhs = F::st(add(mul(sqr(x), x), Codec<F>::b_coeff()));
      typename F::elem r = Codec<F>::sqrt(rhs, oks);
      bool flip = Codec<F>::lex_largest(r) != sflag;
      y = flip ? F::st(neg(r)) : r;
      ok = okx && oks && !iflag && cflag;
    }
  } else {
    bool oky = Codec<F>::parse(b + CB, b[CB], y1);
    y = F::st(y1);
    bool zero_xy = is_zero(x) && is_zero(y);
    ok = okx && oky && (!iflag || zero_xy) && !cflag && !sflag;      // g1.rs:306-318
    inf = iflag;
    if (checked && ok && !inf) ok = on_curve<F>(x, y);               // g1.rs:264-267
  }
  if (checked && ok) ok = torsion_free(x, y, inf);
  if (inf || !ok) emit_identity_wire<F>(xy + i * 2 * WW);
  else { Wire<F>::save(x, xy + i * 2 * WW); Wire<F>::save(y, xy + i * 2 * WW + WW); }
  inf_out[i] = (inf || !ok) ? 1 : 0;
  ok_out[i] = ok ? 1 : 0;
}
template <class F>
__global__ void __launch_bounds__(128) k_point_decode(const uint8_t* __restrict__ in, size_t n, int mode, u32* __restrict__ xy,
                                                      uint8_t* __restrict__ inf_out, uint8_t* __restrict__ ok_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  point_decode_one<F>(i, in, mode, xy, inf_out, ok_out);
}
template <class F>
__global__ void __launch_bounds__(128) k_point_encode(const u32* __restrict__ xy, const uint8_t* __restrict__ inf_in, size_t n, int compressed,
                                                      uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int CB = Codec<F>::COORD_BYTES, WW = Wire<F>::WORDS;
  const bool inf = inf_in && inf_in[i];
  uint8_t* o = out + i * (compressed ? CB : 2 * CB);
  if (inf) {
    for (int j = 0; j < (compressed ? CB : 2 * CB); j++) o[j] = 0;
    o[0] = compressed ? 0xc0 : 0x40;
    return;
  }
  auto x = Wire<F>::load(xy + i * 2 * WW);
  auto y = Wire<F>::load(xy + i * 2 * WW + WW);
  Codec<F>::emit(x, o);
  if (compressed) {
    uint8_t f = 0x80;
    if (Codec<F>::lex_largest(y)) f |= 0x20;
    o[0] |= f;
  } else {
    Codec<F>::emit(y, o + CB);
  }
}

}  // namespace bls
