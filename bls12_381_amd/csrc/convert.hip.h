// convert.hip.h -- wire format <-> internal form.
//
// Wire format = the reference's in-memory value layout: an Fp is six little-endian u64 limbs of the
// canonical representative of x * 2^384 mod p (src/fp.rs:11-15, R at :83-90) -- read here as twelve
// u32 words.  Internal form = x * 2^392 mod p on 14 x 28-bit limbs (fe.hip.h).  One Montgomery
// multiplication by a constant converts in either direction; outputs are fully reduced, so the bytes
// handed back are exactly the reference's limbs.
#pragma once
#include "curve.hip.h"

namespace bls {

// 12 x u32 (saturated) -> 14 x 28-bit limbs, no arithmetic
DEV fe1 unpack_words(const u32* w) {
  fe1 r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    int bit = LW * i, lo = bit >> 5, sh = bit & 31;
    u64 t = (u64)w[lo] >> sh;
    if (sh + LW > 32 && lo + 1 < 12) t |= (u64)w[lo + 1] << (32 - sh);
    r.l[i] = (u32)t & LMASK;
  }
  return r;
}
// 14 x 28-bit normalised limbs (value < 2^384) -> 12 x u32
DEV void pack_words(const fe1& a, u32* w) {
#pragma unroll
  for (int j = 0; j < 12; j++) {
    int bit = 32 * j, i = bit / LW, sh = bit - i * LW;   // limb i holds bits [28 i, 28 i + 28)
    u64 t = (u64)a.l[i] >> sh;
    int got = LW - sh;
    if (i + 1 < NL) t |= (u64)a.l[i + 1] << got;
    if (got + LW < 32 && i + 2 < NL) t |= (u64)a.l[i + 2] << (got + LW);
    w[j] = (u32)t;
  }
}

// reference Montgomery limbs (R = 2^384) -> internal (canonical, V = 1)
DEV fe1 fe_from_ref(const u32* w) {
  constexpr PLimbs k = {BLS_TO_INTERNAL};
  return canon(mul(unpack_words(w), fe1_const(k)));
}
// internal (any bound) -> reference Montgomery limbs, canonical
template <int A, int V>
DEV void fe_to_ref(const Fe<A, V>& a, u32* w) {
  constexpr PLimbs k = {BLS_TO_REF};
  pack_words(canon(mul(norm(a), fe1_const(k))), w);
}
// plain canonical integer (12 words) -> internal
DEV fe1 fe_from_plain(const u32* w) {
  constexpr PLimbs k = {BLS_FROM_INT};
  return canon(mul(unpack_words(w), fe1_const(k)));
}
template <int A, int V>
DEV void fe_to_plain(const Fe<A, V>& a, u32* w) {
  constexpr PLimbs k = {BLS_PLAIN_ONE};
  pack_words(canon(mul(norm(a), fe1_const(k))), w);
}

DEV fe2_1 fe2_from_ref(const u32* w) { fe2_1 r; r.c0 = fe_from_ref(w); r.c1 = fe_from_ref(w + 12); return r; }
template <int A, int V>
DEV void fe2_to_ref(const Fe2<A, V>& a, u32* w) { fe_to_ref(a.c0, w); fe_to_ref(a.c1, w + 12); }

template <int A, int V> DEV fe1 canon_any(const Fe<A, V>& a) { return canon(a); }
template <int A, int V> DEV fe2_1 canon_any(const Fe2<A, V>& a) { fe2_1 r; r.c0 = canon(a.c0); r.c1 = canon(a.c1); return r; }

// policy-generic element I/O used by the templated kernels (WORDS = u32 words per element on the wire)
template <class F> struct Wire;
template <> struct Wire<FpPolicy> {
  static constexpr int WORDS = 12;
  static DEV fe1 load(const u32* w) { return fe_from_ref(w); }
  template <class T> static DEV void save(const T& a, u32* w) { fe_to_ref(a, w); }
};
template <> struct Wire<Fp2Policy> {
  static constexpr int WORDS = 24;
  static DEV fe2_1 load(const u32* w) { return fe2_from_ref(w); }
  template <class T> static DEV void save(const T& a, u32* w) { fe2_to_ref(a, w); }
};

}  // namespace bls
