// fe.hip.h -- base field Fp of BLS12-381 on the gfx950 integer VALU.
//
// Reference semantics: /root/reference/src/fp.rs (add :382-394, sub :421-423, neg :397-418,
// mul :565-609, square :613-660, invert :346-358).  The reference keeps 6x64-bit *saturated* limbs in
// Montgomery form with R = 2^384 and fully reduces after every operation.  That layout is a poor fit
// for CDNA4: measured on MI355X (profiles/r01_microbench_valu.md) v_mad_u64_u32 issues at half rate
// (4 clk/wave) and *every* carry-producing / VOP3 instruction costs the same 4 clk, so a saturated
// 12x32 CIOS spends as long on v_addc as on multiplies (fp_mul: 5.6e10/s).  Here an element is
//
//     14 limbs x 28 bits (unsaturated), value = sum l[i] * 2^(28 i), Montgomery factor R' = 2^392
//
// so a column of the product-scanning multiplication accumulates up to 28 partial products in ONE
// 64-bit register with no carry handling at all: a Montgomery multiplication is 392 v_mad_u64_u32 +
// 14 v_mul_lo_u32 + ~55 full-rate and/shift ops (7.2e10/s in a dependent chain = 63% of the measured
// v_mad_u64_u32 peak counted as 300 canonical MAC32).  Additions and subtractions are limb-wise and
// carry-free ("lazy"); limbs are renormalised only where a bound requires it.
//
// Bounds are tracked IN THE TYPE:  Fe<A,V> promises  l[i] <= A*(2^28-1) for i<13  and
// value < V*p.  mul() static_asserts  A1*A2 <= 17  (column sum < 2^64) and V1*V2 <= 2048
// (result < 2p because p < 2^381 = R'/2^11); sub() picks the multiple of p that keeps every limb
// non-negative.  A formula that could overflow does not compile.
//
// The library converts between the reference's wire format (canonical 6x64 limbs, R = 2^384) and this
// form at the C-ABI boundary only (convert.hip.h); results are canonical and bit-identical to the
// reference because Fp elements are compared/serialised only after full reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "consts_gen.h"

namespace bls {

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v16 __attribute__((ext_vector_type(16)));   // ABI carrier: passes in 16 VGPRs across calls

#define DEV __device__ __forceinline__
#ifndef DEVNI                     // a translation unit built for another occupancy gives its out-of-line routines the matching register budget
#define DEVNI __device__ __noinline__
#endif

// Issue arbitration between the wavefronts of a SIMD is strict oldest-first (profiles/r06_acc_trace.md): of two resident wavefronts with equal
// work the older one runs at its own pace (84 % of the issue slots), the younger gets the rest and then runs ALONE at 84 % for most of its
// life.  The pair fair_init() / fair_tick() makes the LAST TWO wavefronts a SIMD will see share it evenly -- they alternate their user priority
// (1 / 2) with a time slice of the shader-cycle counter and the parity of their hardware slot, so that they retire together and nothing runs
// alone -- while the wavefronts before them keep a constant higher priority (3), i.e. the plain order.  "Layer" l of a launch = wavefronts
// [1024 l, 1024 (l + 1)): with equal work the dispatcher gives every SIMD one wavefront of each layer, in order.  For n layers the launch then
// takes about n W instead of (n + 0.16) W (W = one wavefront's work; n = 3: 3.04 W against 3.13 W).
//   fair_init()  at kernel entry: decides from the launch shape (2 .. 8 layers; one layer has nothing to share with, beyond eight the end
//                effect is below 2 %) and parks the decision in the wavefront's own priority: 0 = untouched, 3 = early layer, 1 = alternating.
//   fair_tick()  at the step boundaries of the long loops: scalar code only (the priority is read back from the STATUS register), nothing
//                unless the wavefront alternates.  A starved wavefront must reach a tick to raise itself, so slices are long (2^21 cycles;
//                2^22 where a wavefront lives ~15 ms).  Kernels that never call fair_init() stay at priority 0 and their ticks do nothing.
// Measured, same box (profiles/r06_fair_tick.md): 2^18-term multi_miller_loop 31.0 -> 28.8 ms, 2^16 pairings 19.8 -> 19.4 ms, hash-to-G2 of
// 2^16 messages 9.5 -> 8.7 ms.  Results do not depend on priorities.
#ifndef BLS_FAIR
#define BLS_FAIR 1
#endif
#ifndef BLS_FAIR_SLICE_LOG
#define BLS_FAIR_SLICE_LOG 21
#endif
constexpr unsigned BLS_CHIP_SIMDS = 1024;          // MI355X: 256 CUs x 4 SIMDs (the only target of this library)
__device__ __forceinline__ void fair_init() {
#if BLS_FAIR
  const unsigned wpb = blockDim.x >> 6;                                           // wavefronts per block (1024 is a multiple of it)
  const unsigned layers = (gridDim.x * wpb + BLS_CHIP_SIMDS - 1) / BLS_CHIP_SIMDS;
  if (layers < 2 || layers > 8) return;
  const unsigned mine = (blockIdx.x * wpb) / BLS_CHIP_SIMDS;
  if (mine + 2 < layers) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
#endif
}
__device__ __forceinline__ void fair_tick(int longer = 0) {
#if BLS_FAIR
  const unsigned p = __builtin_amdgcn_s_getreg((2 << 0) | (3 << 6) | (1 << 11));       // STATUS.USER_PRIO (bits 4:3)
  if (p - 1u > 1u) return;                                                             // only the alternating wavefronts (1 or 2)
  const unsigned t = (unsigned)(__builtin_readcyclecounter() >> (BLS_FAIR_SLICE_LOG + longer));
  const unsigned w = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));      // HW_ID.wave_id: the wavefront's slot on its SIMD
  if ((t ^ w) & 1u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
#else
  (void)longer;
#endif
}
constexpr int NL = 14;
constexpr int LW = 28;
constexpr u32 LMASK = (1u << LW) - 1;
constexpr int MAX_A_PROD = 17;     // 14*(A1*A2+1)*2^56 + carry < 2^64
constexpr int V_DIV = 2520;        // floor(2^392 / p): a*b/R' < p * V1*V2 / 2520
constexpr int MAX_V = 1024;        // value bound representable with a 28-bit top limb
constexpr int mul_v(int v1, int v2) { return 1 + (v1 * v2 + V_DIV - 1) / V_DIV; }

struct PLimbs { u32 l[NL]; };
constexpr PLimbs P_L = {BLS_P_LIMBS};

// c*p spread so that every limb is >= s*(2^28-1): limb0 += s*2^28, limbs 1..12 += s*2^28 - s, limb13 -= s.
constexpr PLimbs make_bias(int c, int s) {
  PLimbs r{};
  u64 carry = 0;
  for (int i = 0; i < NL; i++) {
    u64 t = (u64)P_L.l[i] * (u64)c + carry;
    r.l[i] = (i < NL - 1) ? (u32)(t & LMASK) : (u32)t;
    carry = (i < NL - 1) ? (t >> LW) : 0;
  }
  r.l[0] += (u32)s << LW;
  for (int i = 1; i < NL - 1; i++) r.l[i] += ((u32)s << LW) - (u32)s;
  r.l[NL - 1] -= (u32)s;
  return r;
}

template <int A, int V>
struct Fe {
  u32 l[NL];
  static constexpr int kA = A, kV = V;
  // widening (never narrowing) conversion is free
  template <int A2, int V2>
  DEV operator Fe<A2, V2>() const {
    static_assert(A2 >= A && V2 >= V, "Fe bound narrowing needs norm()/mul()");
    Fe<A2, V2> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = l[i];
    return r;
  }
};

// Storage forms used in structs/arrays (limbs normalised):
//   fe1  canonical inputs (value < p);   fe  working values (value < 12p; every point formula in
//   curve.hip.h maps coordinates < 12p to coordinates < 12p without a value reduction -- build with
//   -DBLS_STRICT_STORE to have the compiler prove it);   fe2p  output of a mul with small inputs.
constexpr int VS = 12;
typedef Fe<1, VS> fe;
typedef Fe<1, 1> fe1;
typedef Fe<1, 2> fe2p;

template <int A, int V>
DEV v16 to_v16(const Fe<A, V>& a) {
  v16 r;
#pragma unroll
  for (int i = 0; i < NL; i++) r[i] = a.l[i];
  r[14] = 0; r[15] = 0;
  return r;
}
template <int V = 2>
DEV Fe<1, V> from_v16(v16 a) {
  Fe<1, V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a[i];
  return r;
}

// ---------------------------------------------------------------------------------------------
// Montgomery multiplication / squaring (product scanning, one 64-bit accumulator per column).
// ---------------------------------------------------------------------------------------------
DEV v16 fe_mul_body(v16 a, v16 b) {
  constexpr PLimbs p = P_L;
  u32 m[NL];
  v16 r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * p.l[k - i];
    m[k] = ((u32)acc * BLS_INV28) & LMASK;
    acc += (u64)m[k] * p.l[0];
    acc >>= LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)a[i] * b[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)m[i] * p.l[k - i];
    r[k - NL] = (u32)acc & LMASK;
    acc >>= LW;
  }
  r[NL - 1] = (u32)acc;
  r[14] = 0; r[15] = 0;
  return r;
}

DEV v16 fe_sqr_body(v16 a) {
  constexpr PLimbs p = P_L;
  u32 m[NL], a2[NL];
  v16 r;
#pragma unroll
  for (int i = 0; i < NL; i++) a2[i] = a[i] << 1;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (u64)a[i] * a2[k - i];
    if ((k & 1) == 0) acc += (u64)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * p.l[k - i];
    m[k] = ((u32)acc * BLS_INV28) & LMASK;
    acc += (u64)m[k] * p.l[0];
    acc >>= LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; i++) acc += (u64)a[i] * a2[k - i];
    if ((k & 1) == 0) acc += (u64)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)m[i] * p.l[k - i];
    r[k - NL] = (u32)acc & LMASK;
    acc >>= LW;
  }
  r[NL - 1] = (u32)acc;
  r[14] = 0; r[15] = 0;
  return r;
}

// Out-of-line copies: ~3.5 KB of straight-line code each, shared by every caller so the hot loops stay
// inside the 64 KB instruction cache.  Arguments/results travel in VGPRs (v16 carrier).
#ifndef BLS_FE_INLINE
DEVNI v16 fe_mul_raw(v16 a, v16 b) { return fe_mul_body(a, b); }
DEVNI v16 fe_sqr_raw(v16 a) { return fe_sqr_body(a); }
#else
DEV v16 fe_mul_raw(v16 a, v16 b) { return fe_mul_body(a, b); }
DEV v16 fe_sqr_raw(v16 a) { return fe_sqr_body(a); }
#endif

// result < p * (1 + V1*V2/2520)
template <int A1, int V1, int A2, int V2>
DEV Fe<1, mul_v(V1, V2)> mul(const Fe<A1, V1>& a, const Fe<A2, V2>& b) {
  static_assert(A1 * A2 <= MAX_A_PROD, "fe mul: limb bound too large, norm() an operand");
  static_assert(mul_v(V1, V2) <= MAX_V, "fe mul: value bound too large");
  return from_v16<mul_v(V1, V2)>(fe_mul_raw(to_v16(a), to_v16(b)));
}
// force-inlined variants (used by the one kernel whose whole body is a single formula and fits the I-cache)
template <int A1, int V1, int A2, int V2>
DEV Fe<1, mul_v(V1, V2)> mul_inl(const Fe<A1, V1>& a, const Fe<A2, V2>& b) {
  static_assert(A1 * A2 <= MAX_A_PROD, "fe mul: limb bound too large, norm() an operand");
  static_assert(mul_v(V1, V2) <= MAX_V, "fe mul: value bound too large");
  return from_v16<mul_v(V1, V2)>(fe_mul_body(to_v16(a), to_v16(b)));
}
template <int A, int V>
DEV Fe<1, mul_v(V, V)> sqr_inl(const Fe<A, V>& a) {
  static_assert(A * A <= MAX_A_PROD, "fe sqr: limb bound too large");
  return from_v16<mul_v(V, V)>(fe_sqr_body(to_v16(a)));
}

// mul that renormalises an operand only when the static limb bounds require it
template <int A1, int V1, int A2, int V2>
DEV Fe<1, mul_v(V1, V2)> mulx(const Fe<A1, V1>& a, const Fe<A2, V2>& b) {
  if constexpr (A1 * A2 <= MAX_A_PROD) return mul(a, b);
  else if constexpr (A1 >= A2 && A2 <= MAX_A_PROD) return mul(norm(a), b);
  else if constexpr (A1 <= MAX_A_PROD) return mul(a, norm(b));
  else return mul(norm(a), norm(b));
}
template <int A, int V>
DEV Fe<1, mul_v(V, V)> sqr(const Fe<A, V>& a) {
  static_assert(A * A <= MAX_A_PROD, "fe sqr: limb bound too large");
  static_assert(mul_v(V, V) <= MAX_V, "fe sqr: value bound too large");
  return from_v16<mul_v(V, V)>(fe_sqr_raw(to_v16(a)));
}

// ---------------------------------------------------------------------------------------------
// Sum of two products with ONE Montgomery reduction: (a0*b0 + a1*b1) / R'  (the analogue of the
// reference's Fp::sum_of_products<2>, fp.rs:430-484).  Column bound: 14*(A00*A01 + A10*A11 + 1) < 256.
// ---------------------------------------------------------------------------------------------
DEV v16 fe_sop2_body(v16 a0, v16 b0, v16 a1, v16 b1) {
  constexpr PLimbs p = P_L;
  u32 m[NL];
  v16 r;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a0[i] * b0[k - i];
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a1[i] * b1[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (u64)m[i] * p.l[k - i];
    m[k] = ((u32)acc * BLS_INV28) & LMASK;
    acc += (u64)m[k] * p.l[0];
    acc >>= LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)a0[i] * b0[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)a1[i] * b1[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) acc += (u64)m[i] * p.l[k - i];
    r[k - NL] = (u32)acc & LMASK;
    acc >>= LW;
  }
  r[NL - 1] = (u32)acc;
  r[14] = 0; r[15] = 0;
  return r;
}

// typed, inlined form: a0*b0 + a1*b1 with one reduction
constexpr int sop2_v(int v00, int v01, int v10, int v11) { return 1 + (v00 * v01 + v10 * v11 + V_DIV - 1) / V_DIV; }
template <int A00, int V00, int A01, int V01, int A10, int V10, int A11, int V11>
DEV Fe<1, sop2_v(V00, V01, V10, V11)> sop2_inl(const Fe<A00, V00>& a0, const Fe<A01, V01>& b0, const Fe<A10, V10>& a1, const Fe<A11, V11>& b1) {
  static_assert(A00 * A01 + A10 * A11 + 1 <= MAX_A_PROD + 1, "fe sop2: limb bound too large, norm() an operand");
  static_assert(sop2_v(V00, V01, V10, V11) <= MAX_V, "fe sop2: value bound too large");
  return from_v16<sop2_v(V00, V01, V10, V11)>(fe_sop2_body(to_v16(a0), to_v16(b0), to_v16(a1), to_v16(b1)));
}

// Fp2 product core: c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0, two reductions in total.
// Contract (checked by the typed wrapper in fp2.hip.h): every operand has limbs <= 2*(2^28-1) and
// value < FE2_IN_V * p.  -a1 is formed against the fixed multiple (FE2_IN_V+1)*p.
constexpr int FE2_IN_A = 2;
constexpr int FE2_IN_V = 128;
struct V16x2 { v16 c0, c1; };
DEVNI V16x2 fe2_mul_raw(v16 a0, v16 a1, v16 b0, v16 b1) {
  constexpr PLimbs bias = make_bias(FE2_IN_V + 1, FE2_IN_A);
  v16 na1;
#pragma unroll
  for (int i = 0; i < NL; i++) na1[i] = bias.l[i] - a1[i];
  na1[14] = 0; na1[15] = 0;
  V16x2 r;
  r.c0 = fe_sop2_body(a0, b0, na1, b1);
  r.c1 = fe_sop2_body(a0, b1, a1, b0);
  return r;
}

// ---------------------------------------------------------------------------------------------
// carry-free linear operations
// ---------------------------------------------------------------------------------------------
template <int A1, int V1, int A2, int V2>
DEV Fe<A1 + A2, V1 + V2> add(const Fe<A1, V1>& a, const Fe<A2, V2>& b) {
  static_assert(A1 + A2 <= 15, "fe add: limb overflow");
  Fe<A1 + A2, V1 + V2> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}

// a - b + (V2+1)*p  (the multiple of p is spread over the limbs so that no limb goes negative)
template <int A1, int V1, int A2, int V2>
DEV Fe<A1 + A2 + 1, V1 + V2 + 1> sub(const Fe<A1, V1>& a, const Fe<A2, V2>& b) {
  static_assert(A1 + A2 + 1 <= 15, "fe sub: limb overflow");
  static_assert(V1 + V2 + 1 <= MAX_V, "fe sub: value bound too large");
  constexpr PLimbs bias = make_bias(V2 + 1, A2);
  Fe<A1 + A2 + 1, V1 + V2 + 1> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + bias.l[i] - b.l[i];
  return r;
}

template <int A, int V>
DEV Fe<A + 1, V + 1> neg(const Fe<A, V>& a) {
  constexpr PLimbs bias = make_bias(V + 1, A);
  Fe<A + 1, V + 1> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = bias.l[i] - a.l[i];
  return r;
}

template <int A, int V>
DEV Fe<2 * A, 2 * V> dbl(const Fe<A, V>& a) {
  static_assert(2 * A <= 15, "fe dbl: limb overflow");
  Fe<2 * A, 2 * V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
  return r;
}

// multiply by a small constant K (limb-wise)
template <int K, int A, int V>
DEV Fe<K * A, K * V> mul_small(const Fe<A, V>& a) {
  static_assert(K * A <= 15, "fe mul_small: limb overflow");
  Fe<K * A, K * V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] * (u32)K;
  return r;
}

// carry propagation: limbs back below 2^28, value unchanged
template <int A, int V>
DEV Fe<1, V> norm(const Fe<A, V>& a) {
  Fe<1, V> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) {
    u32 t = a.l[i] + c;
    r.l[i] = t & LMASK;
    c = t >> LW;
  }
  r.l[NL - 1] = a.l[NL - 1] + c;
  return r;
}

// weak value reduction: normalised limbs, value < V*p  ->  same residue, value < 2p.
// q = floor(top / ceil(p/2^364)) never exceeds floor(value/p) and undershoots it by < 1.01, so
// 0 <= value - q*p < 2p (proof in DESIGN.md "lazy field arithmetic").  ~250 clk, used where a small
// constant multiple (the 3b = 12 of the complete formulas) would otherwise inflate the bounds.
template <int V>
DEV Fe<1, 2> reduce_v(const Fe<1, V>& a) {
  static_assert(V <= MAX_V, "reduce_v: bound");
  if constexpr (V <= 2) {
    return (Fe<1, 2>)a;
  } else {
    constexpr PLimbs p = P_L;
    u32 q = a.l[NL - 1] / 106514u;      // ceil(p / 2^364)
    Fe<1, 2> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
      int64_t t = (int64_t)a.l[i] - (int64_t)((u64)q * p.l[i]) + c;
      r.l[i] = (u32)t & LMASK;
      c = t >> LW;
    }
    r.l[NL - 1] = (u32)((int64_t)a.l[NL - 1] - (int64_t)((u64)q * p.l[NL - 1]) + c);
    return r;
  }
}

// bring a value into the working storage form (limbs normalised, value < 8p); a weak reduction is
// inserted only when the static bound requires one
template <int A, int V>
DEV fe store(const Fe<A, V>& a) {
  if constexpr (V <= VS) {
    if constexpr (A == 1) return (fe)a;
    else return (fe)norm(a);
  } else {
#ifdef BLS_STRICT_STORE
    static_assert(V <= VS, "store: value bound exceeds the storage bound (strict build)");
#endif
    return (fe)reduce_v(norm(a));
  }
}

DEV fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = 0;
  return r;
}
DEV fe fe_const(const PLimbs& c) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c.l[i];
  return r;
}
DEV fe fe_one() { constexpr PLimbs c = {BLS_ONE_MONT}; return fe_const(c); }
DEV fe1 fe1_const(const PLimbs& c) {
  fe1 r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c.l[i];
  return r;
}

template <int A, int V>
DEV Fe<A, V> select(bool c, const Fe<A, V>& a, const Fe<A, V>& b) {   // c ? a : b
  Fe<A, V> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// ---------------------------------------------------------------------------------------------
// canonical reduction (only at outputs / zero tests)
// ---------------------------------------------------------------------------------------------
// value < 2p, limbs normalised  ->  canonical representative in [0,p)
DEV fe2p canon2p(const fe2p& a) {
  constexpr PLimbs p = P_L;
  fe2p s;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) {
    int32_t d = (int32_t)a.l[i] - (int32_t)p.l[i] + c;
    s.l[i] = (u32)d & LMASK;
    c = d >> LW;
  }
  int32_t top = (int32_t)a.l[NL - 1] - (int32_t)p.l[NL - 1] + c;
  s.l[NL - 1] = (u32)top;
  return select(top < 0, a, s);
}

// any Fe -> canonical [0,p) with normalised limbs (still in internal Montgomery form)
template <int A, int V>
DEV fe1 canon(const Fe<A, V>& a) {
  fe2p r;
  if constexpr (V <= 2) {
    r = canon2p((fe2p)norm(a));
  } else {
    static_assert(mul_v(V, 1) <= 2, "canon: value bound too large");
    constexpr PLimbs one = {BLS_ONE_MONT};
    fe1 o;
#pragma unroll
    for (int i = 0; i < NL; i++) o.l[i] = one.l[i];
    r = canon2p(mul(norm(a), o));
  }
  fe1 c;
#pragma unroll
  for (int i = 0; i < NL; i++) c.l[i] = r.l[i];
  return c;
}

template <int A, int V>
DEV bool is_zero(const Fe<A, V>& a) {
  fe1 c = canon(a);
  u32 t = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) t |= c.l[i];
  return t == 0;
}

// cheap filter for "is this value a multiple of p": its low 28 bits must equal those of k*p for some k < V.
// False positives have probability ~V/2^28; a hit is confirmed with the exact (slow) is_zero.
template <int A, int V>
DEV bool maybe_zero(const Fe<A, V>& a) {
  constexpr PLimbs p = P_L;
  u32 l0 = a.l[0] & LMASK;
  bool hit = false;
#pragma unroll
  for (int k = 0; k < V; k++) hit |= (l0 == (((u32)k * p.l[0]) & LMASK));
  return hit;
}
template <int A, int V>
DEV bool is_zero_fast(const Fe<A, V>& a) { return maybe_zero(a) && is_zero(a); }

template <int A1, int V1, int A2, int V2>
DEV bool fe_eq(const Fe<A1, V1>& a, const Fe<A2, V2>& b) {
  fe1 x = canon(a), y = canon(b);
  u32 t = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) t |= x.l[i] ^ y.l[i];
  return t == 0;
}

// ---------------------------------------------------------------------------------------------
// inversion by x^(p-2)  (reference: fp.rs:346-358 pow_vartime; 4-bit fixed window here).  Kept as the
// independent cross-check of the safegcd inversion below (self-test op); ~480 field multiplications.
// ---------------------------------------------------------------------------------------------
DEVNI v16 fe_inv_fermat_raw(v16 xin) {
  constexpr u64 e[6] = BLS_P_MINUS_2_U64;
  fe x = (fe)from_v16<2>(xin);
  // table x^1..x^15, indexed with a run-time digit: it lives in per-lane scratch, not in 210 registers
  fe tab[15];
  tab[0] = x;
#pragma nounroll
  for (int i = 1; i < 15; i++) tab[i] = (fe)mul(tab[i - 1], x);
  fe acc = fe_one();
  bool started = false;
#pragma nounroll
  for (int w = 95; w >= 0; w--) {
    u32 d = (u32)(e[w >> 4] >> ((w & 15) * 4)) & 15u;
    if (started) {
      acc = (fe)sqr(acc); acc = (fe)sqr(acc); acc = (fe)sqr(acc); acc = (fe)sqr(acc);
    }
    if (d) {
      // uniform across the wave: exponent is a constant
      fe t = tab[d - 1];
      acc = started ? (fe)mul(acc, t) : t;
      started = true;
    }
  }
  return to_v16(acc);
}
// ---------------------------------------------------------------------------------------------
// inversion by the Bernstein-Yang "safegcd" divstep iteration (eprint 2019/266), the fixed-iteration form
// with eta = -delta: 40 batches of 28 divsteps on the low limb, each batch folded into a 2x2 transition
// matrix t (entries |.| <= 2^28) that is then applied to the full-size (f, g) and, modulo p, to (d, e).
// 40 * 28 = 1120 >= (49 * 381 + 57) / 17 = 1102 divsteps, the proven bound for 381-bit inputs, so every
// lane runs exactly the same instruction stream.  Cost ~ 34 k ALU operations, i.e. ~ 36 field
// multiplications (the exponentiation above: ~ 480).  The reference inverts by exponentiation
// (fp.rs:346-358); the result is the same field element, 0 for input 0.
// Signed limbs: 14 x 28 bits, low 13 in [0, 2^28), the top limb carries the sign.
// ---------------------------------------------------------------------------------------------
struct SgMat { int32_t u, v, q, r; };
DEV SgMat sg_divsteps(int32_t& eta, u32 f, u32 g) {
  int32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll
  for (int i = 0; i < LW; i++) {
    int32_t c1 = eta >> 31;                    // eta < 0
    int32_t c2 = -(int32_t)(g & 1u);           // g odd
    u32 x = (f ^ (u32)c1) - (u32)c1;           // +-f, +-u, +-v
    int32_t y = (u ^ c1) - c1, z = (v ^ c1) - c1;
    g += x & (u32)c2; q += y & c2; r += z & c2;
    c1 &= c2;
    eta = (eta ^ c1) - (c1 + 1);
    f += g & (u32)c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  SgMat t; t.u = u; t.v = v; t.q = q; t.r = r;
  return t;
}
// The same 28 divsteps in VARIABLE time (the shape of libsecp256k1's modinv32 `divsteps_30_var`): runs of divisions by two are
// taken in one step (count of trailing zeros), and when g is odd up to six of its low bits are cancelled at once by adding the
// right multiple w of f (w = -g / f mod 2^k from a Newton inverse of f: f itself is its own inverse mod 8, one step gives six
// bits), with k capped by eta + 1 -- the point where the constant-time sequence would swap f and g -- so that the transition
// matrix, and with it every later value, is exactly that of sg_divsteps.  ~6 iterations of ~22 instructions per batch instead
// of 28 x 17; lanes of a wavefront iterate until the slowest is done.
DEV SgMat sg_divsteps_var(int32_t& eta_io, u32 f, u32 g) {
  u32 u = 1, v = 0, q = 0, r = 1;
  int32_t eta = eta_io;
  int i = LW;
  for (;;) {
    const int zeros = __builtin_ctz(g | (0xFFFFFFFFu << i));
    g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
    if (i == 0) break;
    if (eta < 0) {
      u32 t;
      eta = -eta;
      t = f; f = g; g = 0u - t;
      t = u; u = q; q = 0u - t;
      t = v; v = r; r = 0u - t;
    }
    int limit = eta + 1 > i ? i : eta + 1;
    if (limit > 6) limit = 6;
    const u32 m = (1u << limit) - 1u;
    u32 fi = f;                                   // f^-1 mod 2^3 (f odd)
    fi *= 2u - f * fi;                            // mod 2^6
    const u32 w = (0u - g * fi) & m;
    g += f * w; q += u * w; r += v * w;
  }
  eta_io = eta;
  SgMat t; t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return t;
}
// (f, g) <- t (f, g) / 2^28   (exact)
DEV void sg_update_fg(int32_t* f, int32_t* g, const SgMat& t) {
  int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
  int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
  cf >>= LW; cg >>= LW;
#pragma unroll
  for (int i = 1; i < NL; i++) {
    cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
    cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
    f[i - 1] = (int32_t)((u32)cf & LMASK); cf >>= LW;
    g[i - 1] = (int32_t)((u32)cg & LMASK); cg >>= LW;
  }
  f[NL - 1] = (int32_t)cf; g[NL - 1] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^28 mod p, both kept in (-2p, p)
DEV void sg_update_de(int32_t* d, int32_t* e, const SgMat& t) {
  constexpr PLimbs p = P_L;
  const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
  int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
  md -= (int32_t)((BLS_PINV28 * (u32)cd + (u32)md) & LMASK);
  me -= (int32_t)((BLS_PINV28 * (u32)ce + (u32)me) & LMASK);
  cd += (int64_t)md * (int32_t)p.l[0]; ce += (int64_t)me * (int32_t)p.l[0];
  cd >>= LW; ce >>= LW;
#pragma unroll
  for (int i = 1; i < NL; i++) {
    cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)md * (int32_t)p.l[i];
    ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)me * (int32_t)p.l[i];
    d[i - 1] = (int32_t)((u32)cd & LMASK); cd >>= LW;
    e[i - 1] = (int32_t)((u32)ce & LMASK); ce >>= LW;
  }
  d[NL - 1] = (int32_t)cd; e[NL - 1] = (int32_t)ce;
}
// d <- (d + (add_p ? p : 0)) with optional negation first; carries propagated
DEV void sg_fix(int32_t* d, int32_t neg_mask, bool add_when_negative) {
  constexpr PLimbs p = P_L;
  int32_t c = 0;
  // negate (two's complement over the limb vector) if neg_mask
#pragma unroll
  for (int i = 0; i < NL; i++) {
    int32_t t = ((d[i] ^ neg_mask) - neg_mask) + c;
    if (i < NL - 1) { d[i] = (int32_t)((u32)t & LMASK); c = t >> LW; } else d[i] = t;
  }
  if (add_when_negative) {
    const int32_t m = d[NL - 1] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      int32_t t = d[i] + ((int32_t)p.l[i] & m) + c;
      if (i < NL - 1) { d[i] = (int32_t)((u32)t & LMASK); c = t >> LW; } else d[i] = t;
    }
  }
}
DEVNI v16 fe_inv_raw(v16 xin) {
  constexpr PLimbs p = P_L;
  fe1 a = canon(from_v16<VS>(xin));           // g must be the canonical integer (the divstep bound needs |g| <= f)
  int32_t f[NL], g[NL], d[NL], e[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) { f[i] = (int32_t)p.l[i]; g[i] = (int32_t)a.l[i]; d[i] = 0; e[i] = 0; }
  e[0] = 1;
  int32_t eta = -1;
#pragma nounroll
  for (int b = 0; b < 40; b++) {
    u32 fl = (u32)f[0] | ((u32)f[1] << LW), gl = (u32)g[0] | ((u32)g[1] << LW);
    SgMat t = sg_divsteps_var(eta, fl, gl);
    sg_update_de(d, e, t);
    sg_update_fg(f, g, t);
    // once g = 0 every further batch is the matrix (2^28, 0; 0, 1): f and d no longer change.  40 batches is the proven bound
    // ((49 * 381 + 57) / 17 divsteps); a typical input is done after 28-30.  The exit is WAVE-UNIFORM (taken when every active lane
    // is done: a per-lane exit makes the compiler keep a second copy of f, g, d, e for the lanes that left and spill), which is
    // what matters where ONE lane inverts on a critical path (wide.hip.h).
    u32 gz = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) gz |= (u32)g[i];
    if (__all(gz == 0)) break;
  }
  // g = 0, f = +-1 (or +-p for input 0, where d = 0): inverse = sign(f) * d, brought into [0, p)
  sg_fix(d, 0, true);                         // (-2p, p) -> (-p, p)
  sg_fix(d, f[NL - 1] >> 31, true);           // negate if f < 0, then -> [0, p)
  fe1 r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = (u32)d[i];
  // r = (x R')^-1  ->  x^-1 R'
  constexpr PLimbs k = {BLS_R3};
  return to_v16(mul(r, fe1_const(k)));
}
// 1/x; returns 0 for x == 0 (the reference's `invert().unwrap_or(Fp::zero())` use, g1.rs:51)
template <int A, int V>
DEV fe2p inv(const Fe<A, V>& a) {
  static_assert(V <= VS, "fe inv: reduce first");
  return from_v16<2>(fe_inv_raw(to_v16(norm(a))));
}

}  // namespace bls
