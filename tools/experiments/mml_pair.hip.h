// mml_pair.hip.h -- EXPERIMENT (round 5), not part of the product: the lane-pair twin of k_mml_prep_quad (prep.hip.h).
//
// Question (round-4 review, item 2): k_multi_miller_shared leaves the per-term state `MmlTerm t[K]` to the register allocator
// (4 680 B scratch, 70.9 GB of HBM-side traffic per 2^18-term launch, 24 % of the wave cycles in s_waitcnt) -- does placing that
// state by hand (per-term data in a coalesced work area, five of the accumulator's six coefficients parked in LDS while a doubling /
// addition step runs) make the kernel faster?
// Answer, measured on MI355X (tools/mml_time.py, 2^18 unprepared terms, K = 4, product tree included, one box):
//     k_multi_miller_shared  30.8 ms        k_mml_pair (this file)  31.3 ms        k_mml_prep_quad  31.8 ms
// with 2 280 B of scratch instead of 4 680 B.  The three kernels issue the same multiply-adds; the spill traffic was latency the
// second wavefront already hid (the same finding as for the quad pairing kernel in rounds 3-4).  K = 2 / 8: 34.1 / 36.6 ms.
// With every term prepared the quad kernel wins (17.2 ms against 19.0 ms), and for 2^14 three-term equations a lane pair per
// equation fills a quarter of the chip (10.2 ms against 5.4 ms).  NOTE: the PREPARED branch of this kernel matched the oracle in
// the SIMT emulation but NOT on hardware (tests of round 5, before removal); it was not debugged because the kernel is slower than
// the quad form in every configuration.  To build it: include this file behind prep.hip.h inside namespace-free code.
#pragma once
#include "../../bls12_381_amd/csrc/prep.hip.h"

namespace bls {

// ---- the lane-pair twin: one accumulator on TWO lanes, explicit placement of the per-term state -------------------------------------
// k_multi_miller_shared (pairing.hip.h) keeps `MmlTerm t[K]` -- P, Q and the running point of every term -- as a per-lane local array
// and leaves its placement to the register allocator: 4 680 B of scratch, read-only coordinates written out and read back every
// iteration, 939x the algorithmic traffic and a quarter of the wave cycles in s_waitcnt (profiles/r04_mml_pmc.md).  Here the state
// is placed by hand, as in the quad kernel above: per-term data (P in internal form, the running point) in the coalesced work area,
// Q re-read from its input array at the five addition steps, and while a doubling / addition step runs five of the six coefficients
// of the accumulator wait in LDS (70 words per lane, the budget of two wavefronts per SIMD) -- only f.c0.c2, the running point and the
// step's temporaries are in registers then.  Prepared terms are accepted exactly as in the quad kernel (same table, same lane roles).
DEV void pair_park_f(u32* park, const Fp12T<PE>& f) {
  park_put(park, 0, f.c1.c0.v); park_put(park, 1, f.c1.c1.v); park_put(park, 2, f.c1.c2.v); park_put(park, 3, f.c0.c0.v); park_put(park, 4, f.c0.c1.v);
}
DEV void pair_unpark_f(const u32* park, Fp12T<PE>& f) {
  park_get(park, 0, f.c1.c0.v); park_get(park, 1, f.c1.c1.v); park_get(park, 2, f.c1.c2.v); park_get(park, 3, f.c0.c0.v); park_get(park, 4, f.c0.c1.v);
}
DEV void pair_work_put_r(const MmlpWork& w, size_t gt, int k, const G2JacT<PE>& r) {
  uint4* p = w.rr + (size_t)k * 12 * w.stride + gt;
  work_put(p, w.stride, r.x.v); work_put(p + 4 * w.stride, w.stride, r.y.v); work_put(p + 8 * w.stride, w.stride, r.z.v);
}
DEV void pair_work_get_r(const MmlpWork& w, size_t gt, int k, G2JacT<PE>& r) {
  const uint4* p = w.rr + (size_t)k * 12 * w.stride + gt;
  work_get(p, w.stride, r.x.v); work_get(p + 4 * w.stride, w.stride, r.y.v); work_get(p + 8 * w.stride, w.stride, r.z.v);
}
// one pass over the terms [beg, beg + K): the conjugated Miller value (pp holds px then py: 8 slots of four words per term)
DEV void mml_pair_pass(Fp12T<PE>& fout, const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2, const uint8_t* __restrict__ g2inf,
                       const u32* __restrict__ qidx, const u32* __restrict__ tab, const uint8_t* __restrict__ tab_inf, u32 tab_n, size_t beg, int K,
                       const MmlpWork& w, size_t gt, u32* park, u32* __restrict__ status) {
  int live = 0;
  for (int k = 0; k < K; k++) {
    const size_t i = beg + k;
    u32 idx = qidx ? qidx[i] : PREP_NONE;
    bool skip = g1inf && g1inf[i];
    if (idx == PREP_NONE) skip = skip || (g2inf && g2inf[i]);
    else if (idx >= tab_n) { atomicOr(status, 4u); skip = true; }
    else skip = skip || tab_inf[idx] != 0;
    if (!skip) {
      uint4* pp = w.pp + (size_t)k * 8 * w.stride + gt;
      work_put(pp, w.stride, fe_from_ref(g1 + i * 24)); work_put(pp + 4 * w.stride, w.stride, fe_from_ref(g1 + i * 24 + 12));
      if (idx == PREP_NONE) {
        G2JacT<PE> r; r.x = E2<PE>::load(g2 + i * 48); r.y = E2<PE>::load(g2 + i * 48 + 24); r.z = E2<PE>::one();
        pair_work_put_r(w, gt, k, r);
      }
      live++;
    }
    w.meta[(size_t)k * w.stride + gt] = skip ? PREP_SKIP : idx;
  }
  Fp12T<PE> f = fp12_one<PE>();
  if (!live) { fout = f; return; }
#pragma nounroll
  for (int s = 0; s < PREP_STEPS; s++) {
    const bool is_add = prep_is_add(s);
    if (!is_add && s > 0) fp12_sqr_hot(f, f);
    for (int k = 0; k < K; k++) {
      const u32 meta = w.meta[(size_t)k * w.stride + gt];
      if (meta == PREP_SKIP) continue;
      LineT<PE> l;
      if (meta == PREP_NONE) {
        G2JacT<PE> r; pair_work_get_r(w, gt, k, r);
        pair_park_f(park, f);
        if (is_add) {
          const size_t i = beg + k;
          const PE qx = E2<PE>::load(g2 + i * 48), qy = E2<PE>::load(g2 + i * 48 + 24);
          addition_step(r, qx, qy, l);
        } else {
          doubling_step(r, l);
        }
        pair_work_put_r(w, gt, k, r);
        pair_unpark_f(park, f);
      } else {
        const u32* e = tab + (size_t)meta * PREP_POINT_WORDS + (size_t)s * 3 * 2 * PREP_LW;
        l.a = prep_load(e); l.b = prep_load(e + 2 * PREP_LW); l.c = prep_load(e + 4 * PREP_LW);
      }
      fe1 px, py;
      const uint4* pp = w.pp + (size_t)k * 8 * w.stride + gt;
      work_get(pp, w.stride, px); work_get(pp + 4 * w.stride, w.stride, py);
      ell(f, l, px, py);
    }
  }
  f.c1 = fp6_neg(f.c1);                       // conjugate: BLS_X_IS_NEGATIVE
  fout = f;
}
// the kernel: segment s on lanes 2s, 2s + 1 (off / kuni / kmax as in k_mml_prep_quad; work area: [kmax] meta | [kmax][8] pp | [kmax][12] running points)
PAIR_KERNEL k_mml_pair(const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2, const uint8_t* __restrict__ g2inf,
                       const u32* __restrict__ qidx, const u32* __restrict__ tab, const uint8_t* __restrict__ tab_inf, u32 tab_n,
                       const unsigned long long* __restrict__ off, size_t nseg, size_t total, int kuni, int kmax,
                       u32* __restrict__ wmeta, uint4* __restrict__ wpp, uint4* __restrict__ wrr, u32* __restrict__ out, u32* __restrict__ status) {
  __shared__ u32 park_lds[PARK_WORDS * PAIRING_BLOCK];
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t q = gt / PL;
  if (q >= nseg) return;
  size_t beg, end;
  if (off) { beg = (size_t)off[q]; end = (size_t)off[q + 1]; }
  else { beg = q * (size_t)kuni; end = beg + (size_t)kuni; }
  if (end > total) end = total;
  if (beg > end) beg = end;
  MmlpWork w; w.meta = wmeta; w.pp = wpp; w.rr = wrr; w.stride = (size_t)gridDim.x * blockDim.x;
  u32* park = park_lds + threadIdx.x;
  Fp12T<PE> acc;
  bool have = false;
  do {
    const int K = (int)(end - beg < (size_t)kmax ? end - beg : (size_t)kmax);
    Fp12T<PE> part;
    mml_pair_pass(part, g1, g1inf, g2, g2inf, qidx, tab, tab_inf, tab_n, beg, K, w, gt, park, status);
    if (have) { Fp12T<PE> t; fp12_mul(t, acc, part); acc = t; } else { acc = part; have = true; }
    beg += (size_t)K;
  } while (beg < end);
  fp12_save(acc, out + q * 144);
}

}  // namespace bls
