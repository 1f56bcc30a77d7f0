// abi_kernels.hip.h -- small kernels at the C-ABI level: conversions between the wire format (6 x 64-bit Montgomery limbs, reference
// layout) and the resident records, affine conversion, batch_normalize, fixed-base multiples of the generators.  Templates only.
#pragma once
#include "msm.hip.h"
#include "codec.hip.h"
#include "generators.hip.h"

using namespace bls;

// ---------------------------------------------------------------------------------------------------
// small kernels living at the ABI level
// ---------------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_bases_import(const u32* __restrict__ xy, const uint8_t* __restrict__ inf, u32* __restrict__ rec, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int WW = Wire<F>::WORDS, EL = Store<F>::EL;
  u32* r = rec + i * Store<F>::AFF_WORDS;
  auto x = Wire<F>::load(xy + i * 2 * WW);
  auto y = Wire<F>::load(xy + i * 2 * WW + WW);
  Store<F>::st(r, x); Store<F>::st(r + EL, y);
  r[2 * EL] = inf ? (inf[i] != 0) : 0;
  for (int j = 2 * EL + 1; j < Store<F>::AFF_WORDS; j++) r[j] = 0;
}
// bad[0] += number of records that are not the identity and fail `is_on_curve() & is_torsion_free()` (g1.rs:396-416,
// g2.rs:475-489): such a set keeps no endomorphism images (see blsgpu_bases::subgroup)
template <class F>
__global__ void __launch_bounds__(128) k_bases_subgroup_check(const u32* __restrict__ rec, size_t n, u32* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> q; bool inf;
  load_aff<F>(rec + i * Store<F>::AFF_WORDS, q, inf);
  typename F::elem x = F::st(q.x), y = F::st(q.y);
  // (the flag travels into torsion_free as a run-time value: with a literal `false` this toolchain's backend aborts on the
  // folded identity test -- "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_private_base")
  bool ok = inf || on_curve<F>(x, y);
  if (ok) ok = torsion_free(x, y, inf);
  if (!ok) atomicAdd(bad, 1u);
}
template <class F>
__global__ void __launch_bounds__(256) k_bases_export(const u32* __restrict__ rec, u32* __restrict__ xy, uint8_t* __restrict__ inf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int WW = Wire<F>::WORDS;
  Aff<F> q; bool f;
  load_aff<F>(rec + i * Store<F>::AFF_WORDS, q, f);
  Wire<F>::save(q.x, xy + i * 2 * WW);
  Wire<F>::save(q.y, xy + i * 2 * WW + WW);
  inf[i] = f ? 1 : 0;
}
template <class F>
__global__ void __launch_bounds__(256) k_proj_import(const u32* __restrict__ xyz, u32* __restrict__ rec, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int WW = Wire<F>::WORDS, EL = Store<F>::EL;
  u32* r = rec + i * Store<F>::PROJ_WORDS;
  Store<F>::st(r, Wire<F>::load(xyz + i * 3 * WW));
  Store<F>::st(r + EL, Wire<F>::load(xyz + i * 3 * WW + WW));
  Store<F>::st(r + 2 * EL, Wire<F>::load(xyz + i * 3 * WW + 2 * WW));
}
template <class F>
__global__ void __launch_bounds__(256) k_proj_export(const u32* __restrict__ rec, u32* __restrict__ xyz, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int WW = Wire<F>::WORDS;
  Proj<F> p; load_proj<F>(rec + i * Store<F>::PROJ_WORDS, p);
  Wire<F>::save(p.x, xyz + i * 3 * WW);
  Wire<F>::save(p.y, xyz + i * 3 * WW + WW);
  Wire<F>::save(p.z, xyz + i * 3 * WW + 2 * WW);
}
// projective record -> affine wire (one inversion per point; identity -> (0, 1, inf))   g1.rs:49-63
template <class F>
__global__ void __launch_bounds__(256) k_proj_to_affine(const u32* __restrict__ rec, u32* __restrict__ xy, uint8_t* __restrict__ inf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int WW = Wire<F>::WORDS;
  Proj<F> p; load_proj<F>(rec + i * Store<F>::PROJ_WORDS, p);
  bool zz = is_zero(p.z);
  auto zi = inv(p.z);
  auto x = mul(p.x, zi);
  auto y = mul(p.y, zi);
  auto one = F::one(); auto zero = F::zero();
  if (zz) {
    Wire<F>::save(zero, xy + i * 2 * WW);
    Wire<F>::save(one, xy + i * 2 * WW + WW);
  } else {
    Wire<F>::save(x, xy + i * 2 * WW);
    Wire<F>::save(y, xy + i * 2 * WW + WW);
  }
  inf[i] = zz ? 1 : 0;
}
// Same conversion with Montgomery's trick, as the reference's batch_normalize does (g1.rs:806-839): lane t owns the
// points t, t+T, t+2T, ... (K per lane), multiplies their non-zero z's into a running product while saving the
// prefixes, inverts ONCE, and walks back.  5 multiplications per point + one inversion per K points instead of one
// inversion (~410 multiplications) per point.
// K is the host's: 32 for large arrays, fewer for arrays that would otherwise leave the chip to a few wavefronts walking long chains
// (2^14 points: 4 per lane = 64 wavefronts, 0.15 ms, instead of 8 wavefronts and 1.1 ms); the affine values do not depend on it.
constexpr int NORMALIZE_K = 32;
static inline int normalize_k(size_t n) { size_t k = n >> 16; return k < 4 ? 4 : k > NORMALIZE_K ? NORMALIZE_K : (int)k; }
template <class F>
__global__ void __launch_bounds__(256) k_batch_normalize(const u32* __restrict__ rec, u32* __restrict__ pref, u32* __restrict__ xy,
                                                         uint8_t* __restrict__ inf, size_t n, size_t T, int K) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  constexpr int WW = Wire<F>::WORDS, EL = Store<F>::EL, PW = Store<F>::PROJ_WORDS;
  typename F::elem acc = F::one();
  for (int k = 0; k < K; k++) {
    size_t i = t + (size_t)k * T;
    if (i >= n) break;
    typename F::elem z; Store<F>::ldw(rec + i * PW + 2 * EL, z);
    Store<F>::st(pref + i * EL, acc);
    if (!is_zero(z)) acc = F::st(mul(acc, z));
  }
  typename F::elem ai = F::st(inv(acc));
  for (int k = K - 1; k >= 0; k--) {
    size_t i = t + (size_t)k * T;
    if (i >= n) continue;
    Proj<F> p; load_proj<F>(rec + i * PW, p);
    typename F::elem pr; Store<F>::ldw(pref + i * EL, pr);
    bool zz = is_zero(p.z);
    if (zz) {
      Wire<F>::save(F::zero(), xy + i * 2 * WW);
      Wire<F>::save(F::one(), xy + i * 2 * WW + WW);
    } else {
      auto zi = mul(pr, ai);
      ai = F::st(mul(ai, p.z));
      Wire<F>::save(mul(p.x, zi), xy + i * 2 * WW);
      Wire<F>::save(mul(p.y, zi), xy + i * 2 * WW + WW);
    }
    inf[i] = zz ? 1 : 0;
  }
}

template <class F>
__global__ void __launch_bounds__(256) k_bases_from_scalars(const u32* __restrict__ scalars, u32* __restrict__ rec, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int EL = Store<F>::EL;
  Aff<F> g = generator<F>();
  Proj<F> acc = pt_identity<F>();
  for (int w = 7; w >= 0; w--) {
    u32 word = scalars[i * 8 + w];
    for (int b = 31; b >= 0; b--) {
      acc = pt_double<F>(acc);
      Proj<F> t = pt_add_mixed<F>(acc, g, false);
      acc = pt_select(((word >> b) & 1) != 0, t, acc);
    }
  }
  bool zz = is_zero(acc.z);
  auto zi = inv(acc.z);
  auto x = canon_any(mul(acc.x, zi));
  auto y = canon_any(mul(acc.y, zi));
  if (zz) { x = canon_any(F::zero()); y = canon_any(F::one()); }   // G1Affine::identity() = (0, 1, inf)
  u32* r = rec + i * Store<F>::AFF_WORDS;
  Store<F>::st(r, x); Store<F>::st(r + EL, y);
  r[2 * EL] = zz ? 1 : 0;
  for (int j = 2 * EL + 1; j < Store<F>::AFF_WORDS; j++) r[j] = 0;
}

// The same multiples from a resident table (fixed-base comb, the `WnafGroup`-style use of a fixed generator: g1.rs:988-1005,
// g2.rs:1133-1149): table[w * 256 + d] = affine([d * 2^(8 w)] G) for the 32 bytes of a scalar -- built once per context by the kernel
// above from 8 192 one-byte scalars -- so a multiple is 32 complete mixed additions (352 field multiplications + the affine
// conversion) instead of 255 doublings and additions (4 845): rec[i] = affine([k_i] G), any 256-bit k_i, the same canonical record.
template <class F>
__global__ void __launch_bounds__(256) k_fixed_base(const u32* __restrict__ scalars, const u32* __restrict__ table, u32* __restrict__ rec, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int EL = Store<F>::EL, AW = Store<F>::AFF_WORDS;
  u32 s[8];
  {
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
    uint4 a = sp[0], b = sp[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  }
  Proj<F> acc = pt_identity<F>();
#pragma nounroll
  for (int w = 0; w < 32; w++) {
    const u32 d = (s[w >> 2] >> ((w & 3) * 8)) & 255u;
    Aff<F> q; bool inf;
    load_aff<F>(table + ((size_t)w * 256 + d) * AW, q, inf);
    acc = pt_add_mixed<F>(acc, q, inf);
  }
  bool zz = is_zero(acc.z);
  auto zi = inv(acc.z);
  auto x = canon_any(mul(acc.x, zi));
  auto y = canon_any(mul(acc.y, zi));
  if (zz) { x = canon_any(F::zero()); y = canon_any(F::one()); }
  u32* r = rec + i * AW;
  Store<F>::st(r, x); Store<F>::st(r + EL, y);
  r[2 * EL] = zz ? 1 : 0;
  for (int j = 2 * EL + 1; j < AW; j++) r[j] = 0;
}

template <class F>
__global__ void k_store_identity(u32* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) store_proj<F>(out, pt_identity<F>());
}

