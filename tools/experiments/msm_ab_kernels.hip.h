// tools/experiments/msm_ab_kernels.hip.h -- NOT part of the product.
//
// Three bucket-accumulation / reduction kernels that were built, proven bit-exact on the GPU and measured SLOWER than what ships
// (rounds 2; DESIGN.md section 9 has the numbers).  They lived in msm.hip.h behind getenv hooks until round 3; they are kept here
// as the record of those experiments.  To rebuild one, include this file after msm.hip.h in a scratch translation unit and
// launch it in place of k_msm_accumulate<FpPolicy> / k_wsum_level_pair (same arguments as the removed call sites:
// bases, images, split index, sorted entries, items, ctrl, records).
//
//   k_msm_accumulate_g1      generic-case fast loop at three wavefronts per SIMD, LDS-DMA record staging      (7 % slower)
//   k_msm_accumulate_g1pair  madd-2008-s over a lane pair, five lock-step products per lane                   (12 % slower)
//   k_wsum_level<F>          one lane per chain of the bottom reduction level (2 M dependent additions)       (reduce 0.79 vs 0.57 ms)
#pragma once
#include "../../bls12_381_amd/csrc/msm.hip.h"

namespace bls {

// G1 accumulation, round-2 form: THREE wavefronts per SIMD.
//
// What kept k_msm_accumulate<FpPolicy> at 241 registers (two wavefronts per SIMD, where a wavefront can start a multiply-add
// only every other issue slot and ~37% of the issue cycles are lost, profiles/r02_msm_pmc.md) was not the addition formula but
// (a) the exceptional cases of madd-2008-s -- doubling / cancellation code with out-of-line calls in the middle of the loop
// body, whose live ranges and call-clobbered registers the allocator had to plan for on the hot path -- and (b) the next
// base record prefetched into 28 registers.  Here
//   * the fast loop handles only the generic case; when the one-limb filter says P = U2 - X MAY be zero (probability
//     ~15 / 2^28 per addition on random input, certain for duplicates and +-pairs) the lane leaves the loop and finishes its
//     chain with the reference's complete mixed addition (RCB15 Alg. 8, g1.rs:715-752) -- same group element;
//   * the next record is prefetched by the LDS-DMA path of gfx950 (global_load_lds_dwordx4: HBM -> LDS without passing
//     through registers), 8 x 16 B per lane into a per-wavefront staging area, and read back with ds_read_b128 at the top
//     of the next iteration.
// Hot loop: 168 registers, no scratch, 32 KB LDS per 256-lane block (three blocks per CU).
#ifndef BLS_ACC_WAVES
#define BLS_ACC_WAVES 3
#endif
#ifndef BLS_ACC_PREFETCH
#define BLS_ACC_PREFETCH 1          // 1: LDS-DMA staging of the next record; 2: next record prefetched into registers; 0: index prefetch only
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;
__global__ void __launch_bounds__(256, BLS_ACC_WAVES) k_msm_accumulate_g1(const u32* __restrict__ bases, const u32* __restrict__ bases2, u32 nsplit,
                                                                           const u32* __restrict__ sorted, const ItemDesc* __restrict__ items,
                                                                           const u32* __restrict__ ctrl, u32* __restrict__ records) {
  typedef FpPolicy F;
  constexpr int AW = Store<F>::AFF_WORDS;
#if BLS_ACC_PREFETCH == 1
  __shared__ uint4 stage[4][AW / 4][64];          // [wavefront][16-byte chunk of the record][lane]
  const u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#endif
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ctrl[2]) return;
  ItemDesc d = items[t];
  auto rec_of = [&](u32 e) -> const u32* {
    u32 idx = e & 0x7fffffffu;
    return idx < nsplit ? bases + (size_t)idx * AW : bases2 + (size_t)(idx - nsplit) * AW;
  };
#if BLS_ACC_PREFETCH == 1
  auto fetch = [&](u32 e) {
    const u32* r = rec_of(e);
#pragma unroll
    for (int c = 0; c < AW / 4; c++)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(r + 4 * c), (lds_void_t*)&stage[wv][c][0], 16, 0, 0);
  };
#endif
  fe X = fe_zero(), Y = fe_zero(), ZZ = fe_zero(), ZZZ = fe_zero();
  bool acc_inf = true, slow = false;
  const u32 end = d.start + d.len;
  u32 j = d.start;
  u32 e = d.len ? sorted[d.start] : 0;
  u32 e_next = d.len > 1 ? sorted[d.start + 1] : 0;
#if BLS_ACC_PREFETCH == 1
  if (d.len) fetch(e);
#elif BLS_ACC_PREFETCH == 2
  Aff<F> qn; bool infn = true;
  if (d.len) load_aff<F>(rec_of(e), qn, infn);
#endif
  for (; j < end; j++) {
    Aff<F> q; bool inf;
#if BLS_ACC_PREFETCH == 1
    {
      u32 w[AW];
#pragma unroll
      for (int c = 0; c < AW / 4; c++) { uint4 v = stage[wv][c][ln]; w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w; }
      Store<F>::ld(w, q.x); Store<F>::ld(w + NL, q.y); inf = w[2 * NL] != 0;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the staging area has been read before the DMA overwrites it
#elif BLS_ACC_PREFETCH == 2
    q = qn; inf = infn;
#else
    load_aff<F>(rec_of(e), q, inf);
#endif
    const u32 e_cur = e;
    u32 e_next2 = 0;
#if BLS_ACC_PREFETCH == 1
    if (j + 1 < end) fetch(e_next);
#elif BLS_ACC_PREFETCH == 2
    if (j + 1 < end) load_aff<F>(rec_of(e_next), qn, infn);
#endif
    if (j + 2 < end) e_next2 = sorted[j + 2];
    e = e_next; e_next = e_next2;
    if (inf) continue;                            // identity base: contributes nothing
    auto qy = cond_neg(q.y, (e_cur >> 31) != 0);
    if (acc_inf) { acc_inf = false; X = F::st(q.x); Y = F::st(qy); ZZ = fe_one(); ZZZ = fe_one(); continue; }
    auto P = sub(mul_inl(q.x, ZZ), X);            // limbs <= 3 * 2^28: still inside the multiplier's column bound
    auto R = sub(mul_inl(qy, ZZZ), Y);
    if (maybe_zero(P)) { slow = true; e = e_cur; break; }      // entry j is NOT consumed
    auto PP = sqr_inl(P);
    auto PPP = mul_inl(P, PP);
    auto Q = mul_inl(X, PP);
    auto X3 = norm(sub(sqr_inl(R), add(PPP, dbl(Q))));
    auto Y3 = sop2_inl(R, norm(sub(Q, X3)), neg(Y), PPP);      // R (Q - X3) - Y1 PPP with one reduction
    ZZ = F::st(mul_inl(ZZ, PP));
    ZZZ = F::st(mul_inl(ZZZ, PPP));
    X = F::st(X3); Y = F::st(Y3);
  }
  Xyzz<F> acc; acc.x = X; acc.y = Y; acc.zz = ZZ; acc.zzz = ZZZ;
  Proj<F> pr = xyzz_to_proj<F>(acc, acc_inf);
  if (slow) {
    for (; j < end; j++) {
      u32 e2 = sorted[j];
      Aff<F> q; bool inf;
      load_aff<F>(rec_of(e2), q, inf);
      if (inf) continue;
      pr = pt_add_mixed_y<F>(pr, q.x, cond_neg(q.y, (e2 >> 31) != 0));
    }
  }
  store_proj<F>(records + (size_t)d.dest * Store<F>::PROJ_WORDS, pr);
}


// G1 accumulation over LANE PAIRS.  One lane per bucket chain needs 241 registers (two wavefronts per SIMD), and with two
// wavefronts the VALU issue port idles ~37% of the time (profiles/r02_msm_pmc.md: a wavefront can issue a multiply-add only
// every other slot, so any bubble in one wavefront is lost).  Here lane 2k (the "x lane") owns X and ZZ of chain k's extended
// Jacobian accumulator and lane 2k+1 (the "y lane") owns Y and ZZZ; madd-2008-s splits into FIVE multiplications per lane
// executed in lock step, operands exchanged with the partner by DPP (quad_perm [1,0,3,2]):
//     step      x lane                       y lane
//     1 mul     U2 = X2 ZZ,   P = U2 - X     S2 = Y2 ZZZ,  R = S2 - Y
//     2 sqr     PP = P^2                     RR = R^2                          exchange: PP <-> RR, X <-> Y
//     3 mul     PPP = P PP                   Q = X PP                          exchange: PPP <-> Q
//     4 mul     ZZ' = ZZ PP                  ZZZ' = ZZZ PPP                    both: X3 = RR - PPP - 2Q
//     5 mul     T = Y PPP                    W = R (Q - X3)                    exchange: T -> y lane;  Y' = W - T,  X' = X3
// 4 x 406 + 315 multiply-adds per lane (3 878 per addition against 4 074 for the one-lane form), half the registers, three
// wavefronts per SIMD.  Same items, same records, same exceptional-case handling (both lanes of a pair take every branch together).
#ifndef BLS_G1PAIR_WAVES
#define BLS_G1PAIR_WAVES 3
#endif
__global__ void __launch_bounds__(256, BLS_G1PAIR_WAVES) k_msm_accumulate_g1pair(const u32* __restrict__ bases, const u32* __restrict__ bases2, u32 nsplit,
                                                                                const u32* __restrict__ sorted, const ItemDesc* __restrict__ items,
                                                                                const u32* __restrict__ ctrl, u32* __restrict__ records) {
  typedef FpPolicy F;
  constexpr int AW = Store<F>::AFF_WORDS, PW = Store<F>::PROJ_WORDS;
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (t >= ctrl[2]) return;                     // both lanes of a pair leave together
  const bool isA = (threadIdx.x & 1) == 0;
  ItemDesc d = items[t];
  fe c0 = fe_zero(), z = fe_zero();             // x lane: X, ZZ      y lane: Y, ZZZ
  bool acc_inf = true;
  const u32 end = d.start + d.len;
  u32 e_next = d.len ? sorted[d.start] : 0;
  for (u32 j = d.start; j < end; j++) {
    const u32 e = e_next;
    if (j + 1 < end) e_next = sorted[j + 1];
    const u32 idx = e & 0x7fffffffu;
    const u32* rec = idx < nsplit ? bases + (size_t)idx * AW : bases2 + (size_t)(idx - nsplit) * AW;
    fe1 q;                                      // x lane: x2      y lane: y2
    {
      const uint2* p = reinterpret_cast<const uint2*>(rec + (isA ? 0 : NL));      // 56-byte halves of the 128-byte record
#pragma unroll
      for (int i = 0; i < NL / 2; i++) { uint2 v = p[i]; q.l[2 * i] = v.x; q.l[2 * i + 1] = v.y; }
    }
    if (rec[2 * NL] != 0) continue;             // identity base (same decision in both lanes)
    const bool ng = (e >> 31) != 0;
    Fe<2, 2> q2 = select(ng && !isA, neg(q), (Fe<2, 2>)q);
    if (acc_inf) { c0 = F::st(q2); z = fe_one(); acc_inf = false; continue; }
    auto m = mul_inl(q2, z);                    // U2 | S2
    auto dd = sub(m, c0);                       // P | R        (limbs <= 3 * 2^28: inside the multiplier's column bound)
    {
      bool mz = maybe_zero(dd);
      bool pmz = partner_flag(mz);
      if (isA ? mz : pmz) {                     // P may be zero: same x -- doubling or cancellation (never on random input)
        bool ez = is_zero(dd);
        bool pez = partner_flag(ez);
        const bool p_zero = isA ? ez : pez, r_zero = isA ? pez : ez;
        if (p_zero) {
          if (r_zero) {
            auto pq = partner((Fe<2, 2>)q2);
            fe1 qx = isA ? q : canon(pq);
            Fe<2, 2> qy = select(isA, (Fe<2, 2>)norm(pq), q2);
            Xyzz<F> r2 = xyzz_double_affine<F>(qx, qy);
            c0 = select(isA, r2.x, r2.y); z = select(isA, r2.zz, r2.zzz);
          } else {
            acc_inf = true;
          }
          continue;
        }
      }
    }
    auto ee = sqr_inl(dd);                      // PP | RR                                   Fe<1,2>
    auto pe = partner(ee);                      // RR | PP
    auto pc0 = partner(c0);                     // Y  | X
    typedef Fe<3, 15> W1;
    auto f = mul_inl(select(isA, (W1)dd, (W1)pc0), select(isA, ee, pe));        // PPP | Q    Fe<1,2>
    auto pf = partner(f);                       // Q | PPP
    auto zn = mul_inl(z, select(isA, ee, pf));  // ZZ PP | ZZZ PPP
    auto rr = select(isA, pe, ee), ppp = select(isA, f, pf), qq = select(isA, pf, f);
    auto x3 = norm(sub(rr, add(ppp, dbl(qq))));                                 // Fe<1,9> in both lanes
    typedef Fe<1, 12> W2;
    auto g = mul_inl(select(isA, (W1)pc0, (W1)dd), select(isA, (W2)ppp, (W2)norm(sub(qq, x3))));      // Y PPP | R (Q - X3)
    auto npg = partner(neg(g));                 // the y lane needs W - T: the x lane sends -T (never subtract an exchanged value)
    typedef Fe<3, 9> W3;
    c0 = F::st(select(isA, (W3)x3, (W3)add(g, npg)));
    z = F::st(zn);
  }
  // XYZZ -> (X ZZZ : Y ZZ : ZZ ZZZ); identity -> (0 : 1 : 0)
  u32* o = records + (size_t)d.dest * PW;
  fe cx, cz;
  if (acc_inf) {
    cx = isA ? fe_zero() : fe_one(); cz = fe_zero();
  } else {
    auto pz = partner(z);
    cx = F::st(mul(c0, pz)); cz = F::st(mul(z, pz));
  }
#pragma unroll
  for (int i = 0; i < NL; i++) o[(isA ? 0 : NL) + i] = cx.l[i];
  if (isA) {
#pragma unroll
    for (int i = 0; i < NL; i++) o[2 * NL + i] = cz.l[i];
  }
}


// ---- 6. weighted bucket reduction -----------------------------------------------------------------------
// One level:  elements E[seg][0..n) (PROJ records), weight(j) = j + off.  Thread (seg, g) handles chunk
// j in [g*M, (g+1)*M):  R = sum E_j,  T = sum (j - g*M + off) * E_j   (running sums, high index first).
// Then  wsum(seg) = sum_g T_g + M * sum_g g * R_g.
template <class F>
__global__ void __launch_bounds__(256) k_wsum_level(const u32* __restrict__ E, u32* __restrict__ Rout, u32* __restrict__ Tout,
                                                    int nseg, int n, int M, int off) {
  __builtin_amdgcn_s_setprio(3);      // latency-bound tail: win VALU arbitration against co-resident bulk waves
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int G = n / M;
  if (t >= nseg * G) return;
  int seg = t / G, g = t - seg * G;
  const u32* base = E + ((size_t)seg * n + (size_t)g * M) * Store<F>::PROJ_WORDS;
  Proj<F> run = pt_identity<F>(), tot = pt_identity<F>();
  for (int i = M - 1; i >= 0; i--) {
    Proj<F> e; load_proj<F>(base + (size_t)i * Store<F>::PROJ_WORDS, e);
    run = pt_add<F>(run, e);
    if (i > 0 || off) tot = pt_add<F>(tot, run);
  }
  store_proj<F>(Rout + (size_t)t * Store<F>::PROJ_WORDS, run);
  store_proj<F>(Tout + (size_t)t * Store<F>::PROJ_WORDS, tot);
}

}  // namespace bls
