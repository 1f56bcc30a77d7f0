// prep.hip.h -- `G2Prepared` as a device-resident table of line coefficients, and the Miller loops that consume it.
//
// Reference: /root/reference/src/pairings.rs -- `G2Prepared` (:487-502: `infinity` + `coeffs: Vec<(Fp2, Fp2, Fp2)>`), `From<G2Affine>
// for G2Prepared` (:504-546: the 68 coefficient triples of the 63 doubling and 5 addition steps, computed ONCE per point; the identity
// keeps the generator's coefficients and its flag), and `multi_miller_loop` (:554-603), which only EVALUATES the stored lines at P
// (`ell`, :696-707) -- the doubling / addition steps on the twist are not repeated per call.  Round 1-4 kept `G2Prepared` an opaque
// holder of the affine point and re-walked the running point in every call; for verification-key-shaped workloads (Groth16: three of
// four G2 arguments fixed; BLS with a fixed generator) that is ~1.5-2x the per-term work of the reference.
//
// Layout.  One table entry = one coefficient triple in the library's internal limb form (14 x 28-bit limbs, pairlane.hip.h: the c0
// and c1 coefficient of an Fp2 value side by side), 16 words per lane-coefficient so that a lane fetches its half with aligned
// 16-byte loads:  tab[((point * 68 + step) * 3 + coef) * 32 + (c1 ? 16 : 0) + limb],  26 112 B per point.  A wavefront whose quads
// all use the same prepared point reads two 64-byte segments per load instruction (broadcast); the table of a verification key stays
// in the L2.
//
// The consuming kernel is the quad layout of quad.hip.h (one accumulator on four lanes) with ONE accumulator per SEGMENT (the
// reference's own schedule: per step every term multiplies its line into f, one squaring for all) and terms that are either prepared
// (the line is loaded) or not (the running point lives in a coalesced per-quad work area and takes the doubling / addition step as
// before).  Every value is the field element the unprepared path computes -- the stored triples are the outputs of the very same
// q_doubling_step / q_addition_step -- so raw Miller values stay limb-identical to both the unprepared kernels and the oracle.
#pragma once
#include "quad.hip.h"
#include "limits.h"

namespace bls {

constexpr int PREP_STEPS = 68;                      // 63 doubling + 5 addition steps (pairings.rs:516-546)
constexpr int PREP_LW = 16;                         // words per lane-coefficient (14 limbs + 2 of padding)
constexpr size_t PREP_POINT_WORDS = (size_t)PREP_STEPS * 3 * 2 * PREP_LW;
constexpr u32 PREP_SKIP = 0xfffffffeu;              // (internal) the term is skipped: identity on either side (pairings.rs:566-569)

// which of the 68 steps are addition steps: walking the bits of BLS_X >> 1 below the leading one, a set bit appends an addition step
// behind that iteration's doubling step (pairings.rs:671-687)
struct StepMask { unsigned long long lo; u32 hi; };
constexpr StepMask prep_add_mask() {
  StepMask m{0, 0};
  int s = 0;
  for (int b = 61; b >= 0; b--) {
    s++;                                            // the doubling step of this iteration
    if ((X_HALF >> b) & 1) { if (s < 64) m.lo |= 1ull << s; else m.hi |= 1u << (s - 64); s++; }
  }
  return m;                                         // (step 67 is the final doubling step)
}
DEV bool prep_is_add(int s) {
  constexpr StepMask m = prep_add_mask();
  return s < 64 ? ((m.lo >> s) & 1) != 0 : ((m.hi >> (s - 64)) & 1u) != 0;
}

// ---- table I/O: this lane's coefficient (c0 or c1) of one Fp2 value --------------------------------------------------------------
DEV void prep_store(u32* e, const QR& a) {
  uint4* p = reinterpret_cast<uint4*>(e + (lane_is_c1() ? PREP_LW : 0));
  p[0] = make_uint4(a.v.l[0], a.v.l[1], a.v.l[2], a.v.l[3]);
  p[1] = make_uint4(a.v.l[4], a.v.l[5], a.v.l[6], a.v.l[7]);
  p[2] = make_uint4(a.v.l[8], a.v.l[9], a.v.l[10], a.v.l[11]);
  p[3] = make_uint4(a.v.l[12], a.v.l[13], 0u, 0u);
}
DEV QR prep_load(const u32* e) {
  const uint4* p = reinterpret_cast<const uint4*>(e + (lane_is_c1() ? PREP_LW : 0));
  const uint4 a = p[0], b = p[1], c = p[2];
  const uint2 d = *reinterpret_cast<const uint2*>(p + 3);
  QR r;
  r.v.l[0] = a.x; r.v.l[1] = a.y; r.v.l[2] = a.z; r.v.l[3] = a.w;
  r.v.l[4] = b.x; r.v.l[5] = b.y; r.v.l[6] = b.z; r.v.l[7] = b.w;
  r.v.l[8] = c.x; r.v.l[9] = c.y; r.v.l[10] = c.z; r.v.l[11] = c.w;
  r.v.l[12] = d.x; r.v.l[13] = d.y;
  return r;
}

// tab[first + i] <- the 68 coefficient triples of Q_i (g2: affine wire coordinates, 48 words per point); one quad per point, the
// running point replicated on both pairs as in the Miller loop, pair A writes.  The identity keeps the generator's coefficients
// (pairings.rs:506-509) and its flag.
QUAD_KERNEL k_g2_prepare_quad(const u32* __restrict__ g2, const uint8_t* __restrict__ g2inf, size_t n, u32* __restrict__ tab, uint8_t* __restrict__ tab_inf) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / QL;
  if (i >= n) return;
  const bool B = lane_is_B();
  const bool ident = g2inf && g2inf[i];
  QR qx, qy;
  if (ident) {
    constexpr PLimbs x0 = {BLS_G2_GEN_X0}, x1 = {BLS_G2_GEN_X1}, y0 = {BLS_G2_GEN_Y0}, y1 = {BLS_G2_GEN_Y1};
    qx = E2<QR>::konst(x0, x1); qy = E2<QR>::konst(y0, y1);
  } else {
    qx = E2<QR>::load(g2 + i * 48); qy = E2<QR>::load(g2 + i * 48 + 24);
  }
  QJac r; r.x = qx; r.y = qy; r.z = E2<QR>::one();
  u32* base = tab + i * PREP_POINT_WORDS;
#pragma nounroll
  for (int s = 0; s < PREP_STEPS; s++) {
    QLin l;
    if (prep_is_add(s)) q_addition_step(r, qx, qy, l); else q_doubling_step(r, l);
    if (!B) {
      u32* e = base + (size_t)s * 3 * 2 * PREP_LW;
      prep_store(e, l.a); prep_store(e + 2 * PREP_LW, l.b); prep_store(e + 4 * PREP_LW, l.c);
    }
  }
  if ((threadIdx.x & 3) == 0) tab_inf[i] = ident ? 1 : 0;
}
// the stored triples in the reference's own value format (`coeffs: Vec<(Fp2, Fp2, Fp2)>`: 68 x 3 x 12 u64 Montgomery limbs): one lane
// pair per step of point `index`
__global__ void __launch_bounds__(256) k_g2_prepared_export(const u32* __restrict__ tab, size_t index, u32* __restrict__ out) {
  const int s = (int)((blockIdx.x * blockDim.x + threadIdx.x) / 2);
  if (s >= PREP_STEPS) return;
  const u32* e = tab + index * PREP_POINT_WORDS + (size_t)s * 3 * 2 * PREP_LW;
#pragma unroll
  for (int c = 0; c < 3; c++) E2<QR>::save(prep_load(e + c * 2 * PREP_LW), out + ((size_t)s * 3 + c) * 24);
}

// ---- per-quad work area of the shared loop (coalesced: element [slot][thread]) ----------------------------------------------------
// meta[k]: PREP_SKIP, PREP_NONE or the term's table index; pp[k]: this lane's coordinate of P_k in internal form (py on pair A, px on
// pair B: what q_ell multiplies the line by); rr[k]: the running point of an unprepared term (x, y, z: this lane's coefficient)
struct MmlpWork { u32* meta; uint4* pp; uint4* rr; size_t stride; };
template <int V> DEV void work_put(uint4* p, size_t stride, const Fe<1, V>& a) {
  p[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
  p[stride] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
  p[2 * stride] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
  p[3 * stride] = make_uint4(a.l[12], a.l[13], 0u, 0u);
}
template <int V> DEV void work_get(const uint4* p, size_t stride, Fe<1, V>& a) {
  const uint4 x = p[0], y = p[stride], z = p[2 * stride], w = p[3 * stride];
  a.l[0] = x.x; a.l[1] = x.y; a.l[2] = x.z; a.l[3] = x.w;
  a.l[4] = y.x; a.l[5] = y.y; a.l[6] = y.z; a.l[7] = y.w;
  a.l[8] = z.x; a.l[9] = z.y; a.l[10] = z.z; a.l[11] = z.w;
  a.l[12] = w.x; a.l[13] = w.y;
}
DEV void work_put_r(const MmlpWork& w, size_t gt, int k, const QJac& r) {
  uint4* p = w.rr + (size_t)k * 12 * w.stride + gt;
  work_put(p, w.stride, r.x.v); work_put(p + 4 * w.stride, w.stride, r.y.v); work_put(p + 8 * w.stride, w.stride, r.z.v);
}
DEV void work_get_r(const MmlpWork& w, size_t gt, int k, QJac& r) {
  const uint4* p = w.rr + (size_t)k * 12 * w.stride + gt;
  work_get(p, w.stride, r.x.v); work_get(p + 4 * w.stride, w.stride, r.y.v); work_get(p + 8 * w.stride, w.stride, r.z.v);
}
DEV Q12<VQM> q12_widen(const Q12<VQ>& f) { Q12<VQM> g; g.h.c0 = f.h.c0; g.h.c1 = f.h.c1; g.h.c2 = f.h.c2; return g; }

// One pass of the shared loop over the terms [beg, beg + K) of a segment, K <= kmax <= MMLP_MAX_K (pairings.rs:554-603): the conjugated
// Miller value of those terms.  The 68 steps are walked as ONE loop -- square before every doubling step but the first, then every
// term's line: loaded (prepared) or produced by the doubling / addition step of its running point (unprepared) -- so that the
// accumulator code (q_ell, q12_sqr) exists once in the instruction stream.
DEV QC12 mmlp_pass(const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2, const uint8_t* __restrict__ g2inf,
                   const u32* __restrict__ qidx, const u32* __restrict__ tab, const uint8_t* __restrict__ tab_inf, u32 tab_n, size_t beg, int K,
                   const MmlpWork& w, size_t gt, u32* park, u32* __restrict__ status) {
  const bool B = lane_is_B();
  int live = 0;
  for (int k = 0; k < K; k++) {
    const size_t i = beg + k;
    u32 idx = qidx ? qidx[i] : PREP_NONE;
    bool skip = g1inf && g1inf[i];
    if (idx == PREP_NONE) skip = skip || (g2inf && g2inf[i]);
    else if (idx >= tab_n) { atomicOr(status, 4u); skip = true; }          // an index outside the table: reported, the term contributes nothing
    else skip = skip || tab_inf[idx] != 0;
    if (!skip) {
      const fe1 px = fe_from_ref(g1 + i * 24), py = fe_from_ref(g1 + i * 24 + 12);
      work_put(w.pp + (size_t)k * 4 * w.stride + gt, w.stride, select(B, px, py));
      if (idx == PREP_NONE) {
        QJac r; r.x = E2<QR>::load(g2 + i * 48); r.y = E2<QR>::load(g2 + i * 48 + 24); r.z = E2<QR>::one();
        work_put_r(w, gt, k, r);
      }
      live++;
    }
    w.meta[(size_t)k * w.stride + gt] = skip ? PREP_SKIP : idx;
  }
  if (!live) return q12_to_cold(q12_one());
  Q12<VQM> g = q12_widen(q12_one());
#pragma nounroll
  for (int s = 0; s < PREP_STEPS; s++) {
    const bool is_add = prep_is_add(s);
    if (!is_add && s > 0) g = q12_widen(q12_sqr(g));
    for (int k = 0; k < K; k++) {
      const u32 meta = w.meta[(size_t)k * w.stride + gt];
      fair_tick(1);
      if (meta == PREP_SKIP) continue;
      fe1 pp; work_get(w.pp + (size_t)k * 4 * w.stride + gt, w.stride, pp);
      QLin l;
      if (meta == PREP_NONE) {
        // the running point comes into registers, the accumulator waits in LDS meanwhile (as in q_miller_loop)
        QJac r; work_get_r(w, gt, k, r);
        qpark_put(park, 0, g.h.c0.v); qpark_put(park, 1, g.h.c1.v); qpark_put(park, 2, g.h.c2.v);
        if (is_add) {
          const size_t i = beg + k;
          const QR qx = E2<QR>::load(g2 + i * 48), qy = E2<QR>::load(g2 + i * 48 + 24);
          q_addition_step(r, qx, qy, l);
        } else {
          q_doubling_step(r, l);
        }
        qpark_get(park, 0, g.h.c0.v); qpark_get(park, 1, g.h.c1.v); qpark_get(park, 2, g.h.c2.v);
        work_put_r(w, gt, k, r);
      } else {
        const u32* e = tab + (size_t)meta * PREP_POINT_WORDS + (size_t)s * 3 * 2 * PREP_LW;
        l.a = prep_load(e); l.b = prep_load(e + 2 * PREP_LW); l.c = prep_load(e + 4 * PREP_LW);
      }
      g = q_ell(g, l, pp);
    }
  }
  // conjugate (BLS_X_IS_NEGATIVE): c1 -> -c1 on pair B
  Q12<VQM> o;
  o.h.c0 = fit<VQM>(selB(B, neg(g.h.c0), g.h.c0));
  o.h.c1 = fit<VQM>(selB(B, neg(g.h.c1), g.h.c1));
  o.h.c2 = fit<VQM>(selB(B, neg(g.h.c2), g.h.c2));
  return q12_to_cold(o);
}

// out[s] = multi_miller_loop(terms of segment s), terms prepared or not (pairings.rs:554-603).  Segment s = terms [off[s], off[s + 1])
// (offsets clamped to `total`), or -- off == nullptr -- the run [s * kuni, (s + 1) * kuni) of ONE long product whose partial values the
// caller multiplies up.  One quad per segment; a segment longer than kmax takes ceil(len / kmax) passes, multiplied together here.
QUAD_KERNEL k_mml_prep_quad(const u32* __restrict__ g1, const uint8_t* __restrict__ g1inf, const u32* __restrict__ g2, const uint8_t* __restrict__ g2inf,
                            const u32* __restrict__ qidx, const u32* __restrict__ tab, const uint8_t* __restrict__ tab_inf, u32 tab_n,
                            const unsigned long long* __restrict__ off, size_t nseg, size_t total, int kuni, int kmax,
                            u32* __restrict__ wmeta, uint4* __restrict__ wpp, uint4* __restrict__ wrr, u32* __restrict__ out, u32* __restrict__ status) {
  fair_init();
  __shared__ u32 park_lds[QPARK_WORDS * QUAD_BLOCK];
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t q = gt / QL;
  if (q >= nseg) return;
  size_t beg, end;
  if (off) { beg = (size_t)off[q]; end = (size_t)off[q + 1]; }
  else { beg = q * (size_t)kuni; end = beg + (size_t)kuni; }
  if (end > total) end = total;
  if (beg > end) beg = end;
  MmlpWork w; w.meta = wmeta; w.pp = wpp; w.rr = wrr; w.stride = (size_t)gridDim.x * blockDim.x;
  u32* park = park_lds + threadIdx.x;
  QC12 acc;
  bool have = false;
  do {
    const int K = (int)(end - beg < (size_t)kmax ? end - beg : (size_t)kmax);
    QC12 part = mmlp_pass(g1, g1inf, g2, g2inf, qidx, tab, tab_inf, tab_n, beg, K, w, gt, park, status);
    if (have) { QC12 t; qc_mul(t, acc, part); acc = t; } else { acc = part; have = true; }
    beg += (size_t)K;
  } while (beg < end);
  qc_save(acc, out + q * 144);
}

}  // namespace bls
