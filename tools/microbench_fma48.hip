// tools/microbench_fma48.hip -- data-gated experiment: is the FP64 pipe a better multiplier than v_mad_u64_u32
// for the 381-bit Montgomery product?  MI355X issues v_fma_f64 at the rate of v_mad_u64_u32 (16 lanes per clock and SIMD; part 1
// measures both), and an FMA covers a 48 x 48-bit limb product where the integer instruction covers 28 x 28 -- IF the 96-bit product
// can be had in few instructions.  This file holds the best formulation found, bit-exact, in the harness of microbench_mul30.hip:
//
//   8 limbs of 48 bits held as integer-valued doubles, Montgomery factor R = 2^384 (the reference's own: src/fp.rs:487-609 -- the
//   kernel computes exactly `Fp::mul` up to the final conditional subtraction), rounding mode of the wavefront set to
//   round-toward-zero (MODE register) so that
//       H' = fma(x, y, H)          with H = 2^100 + (multiple of 2^48): H' = H + floor(x y / 2^48) 2^48   EXACTLY (one chain per column)
//       lo = fma(x, y, H - H')     = x y mod 2^48                                                             EXACTLY
//       L += lo                    (a column holds <= 16 low parts and <= 16 high parts of < 2^48: < 2^53, exact)
//   i.e. FOUR FP64 instructions per limb product (FMA, SUB, FMA, ADD), 64 + 64 limb products per Montgomery product, plus per
//   column the quotient digit (3), the split of the column sum into limb and carry (6): ~630 FP64 instructions, every one of them
//   on the half-rate pipe, against 406 v_mad_u64_u32 + ~140 others (a third of them full rate) for the shipped 14 x 28-bit product.
//   (The variant of Emmart / Zheng / Weems -- hi = fma(x, y, 2^104), lo = fma(x, y, 2^104 + 2^52 - hi), bit patterns accumulated
//   with INTEGER adds -- needs five instructions per limb product here, because a 64-bit integer add is one half-rate instruction
//   (v_lshl_add_u64) on this chip, not two full-rate ones.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bls12_381_amd/csrc tools/microbench_fma48.hip -o build/mbfma && build/mbfma
// prints one JSON line per measurement: issue rates (part 1), the bit-exactness verdict over 2^20 operand pairs against a host
// restatement of fp.rs's 6 x 64-bit CIOS (part 2), and the dependent-chain rates at 2 / 4 / 8 wavefronts per SIMD next to the
// shipped fe_mul_body (part 3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "fe.hip.h"
using namespace bls;

// ---- constants: p in 48-bit limbs, -p^-1 mod 2^48 -----------------------------------------------------------------------------
constexpr int NF = 8;
struct PD { double l[NF]; };
constexpr u64 p48_limb(int j) {
  u64 v = 0;
  for (int b = 0; b < 48; b++) {
    int bit = j * 48 + b, i = bit / LW, s = bit % LW;
    if (i < NL) v |= (u64)((P_L.l[i] >> s) & 1u) << b;
  }
  return v;
}
constexpr u64 make_inv48() {            // -p^-1 mod 2^48 by Newton iteration
  u64 p0 = p48_limb(0), x = 1;
  for (int i = 0; i < 7; i++) x = x * (2 - p0 * x);
  return (0 - x) & ((1ull << 48) - 1);
}
constexpr u64 INV48 = make_inv48();
__device__ __constant__ double P48[NF] = {(double)p48_limb(0), (double)p48_limb(1), (double)p48_limb(2), (double)p48_limb(3),
                                          (double)p48_limb(4), (double)p48_limb(5), (double)p48_limb(6), (double)p48_limb(7)};

#define C100 1267650600228229401496703205376.0     /* 2^100 */
#define C48 281474976710656.0                      /* 2^48 */
#define CM48 3.552713678800501e-15                 /* 2^-48 */

// round-toward-zero for FP64 in this wavefront: MODE[3:2] = 3  (hwreg id 1, offset 2, width 2)
// (written as inline assembly: the compiler's mode-register pass tracks __builtin_amdgcn_s_setreg and RESTORES round-to-nearest in
// front of the first FP64 instruction, because plain floating-point IR is defined under the default rounding mode; the volatile
// asm is invisible to that pass, and the FP64 instructions below are ordered after it by their data dependence on the loads)
__device__ __forceinline__ void set_rz() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory"); }

struct F48 { double l[NF]; };
// one limb product folded into the column's (H, L) pair: four FP64 instructions
#define TERM(x, y) do { const double Hn_ = __builtin_fma((x), (y), H); const double d_ = H - Hn_; L += __builtin_fma((x), (y), d_); H = Hn_; } while (0)
__device__ __forceinline__ F48 mul48(const F48& a, const F48& b) {
  double m[NF];
  F48 r;
  double carry = 0.0;
#pragma unroll
  for (int k = 0; k < 2 * NF; k++) {
    double H = C100, L = carry;
#pragma unroll
    for (int i = 0; i < NF; i++) {
      const int j = k - i;
      if (j >= 0 && j < NF) TERM(a.l[i], b.l[j]);
    }
#pragma unroll
    for (int i = 0; i < NF; i++) {
      const int j = k - i;
      if (i < k && j >= 0 && j < NF) TERM(m[i], P48[j]);
    }
    if (k < NF) {
      // quotient digit: m_k = (L mod 2^48) * (-p^-1) mod 2^48
      const double q48 = (L + C100) - C100;                   // floor(L / 2^48) 2^48 (round toward zero)
      const double rlo = L - q48;
      const double Hm = __builtin_fma(rlo, (double)INV48, C100);
      m[k] = __builtin_fma(rlo, (double)INV48, C100 - Hm);
      TERM(m[k], P48[0]);                                      // makes the column divisible by 2^48
      carry = ((H - C100) + L) * CM48;
    } else {
      const double q48 = (L + C100) - C100;
      r.l[k - NF] = L - q48;
      carry = ((H - C100) + q48) * CM48;
    }
  }
  return r;                                                    // < 2p for inputs < 2p (a b / R < 0.4 p, m p / R < p); the top carry is zero
}

template <int WAVES> __global__ void __launch_bounds__(256, WAVES) k_chain48(double* __restrict__ out, const double* __restrict__ in, int iters) {
  set_rz();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  F48 a, b;
  for (int j = 0; j < NF; j++) { a.l[j] = in[(tid & 255) * 16 + j]; b.l[j] = in[(tid & 255) * 16 + 8 + j]; }
  for (int it = 0; it < iters; it++) { F48 r = mul48(a, b); a = b; b = r; }
  for (int j = 0; j < NF; j++) out[(size_t)tid * NF + j] = b.l[j];
}
template <int WAVES> __global__ void __launch_bounds__(256, WAVES) k_chain28(u32* __restrict__ out, const u32* __restrict__ in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  v16 a, b;
  for (int j = 0; j < NL; j++) { a[j] = in[(tid & 255) * 32 + j] & LMASK; b[j] = in[(tid & 255) * 32 + 16 + j] & LMASK; }
  a[NL - 1] &= 0xffff; b[NL - 1] &= 0xffff; a[14] = a[15] = b[14] = b[15] = 0;
  for (int it = 0; it < iters; it++) { v16 r = fe_mul_body(a, b); a = b; b = r; }
  for (int j = 0; j < NL; j++) out[(size_t)tid * 16 + j] = b[j];
}
// n independent products out[i] = a[i] b[i] / 2^384 (value below 2p), for the bit-exactness check
__global__ void __launch_bounds__(256) k_mul48_batch(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, size_t n) {
  set_rz();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F48 x, y;
  for (int j = 0; j < NF; j++) { x.l[j] = a[i * NF + j]; y.l[j] = b[i * NF + j]; }
  F48 r = mul48(x, y);
  for (int j = 0; j < NF; j++) out[i * NF + j] = r.l[j];
}

// ---- part 1: issue rates (64 independent chains per lane would not fit: 8 independent accumulators, unrolled 8 x) ------------------
template <int OP, int WAVES> __global__ void __launch_bounds__(256, WAVES) k_rate(double* __restrict__ out, const double* __restrict__ in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (OP == 0 || OP == 1) {
    double x = in[tid & 255], y = in[(tid + 7) & 255];
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (OP == 0) { a0 = __builtin_fma(x, y, a0); a1 = __builtin_fma(x, y, a1); a2 = __builtin_fma(x, y, a2); a3 = __builtin_fma(x, y, a3);
                       a4 = __builtin_fma(x, y, a4); a5 = __builtin_fma(x, y, a5); a6 = __builtin_fma(x, y, a6); a7 = __builtin_fma(x, y, a7); }
        else { a0 += y; a1 += y; a2 += y; a3 += y; a4 += y; a5 += y; a6 += y; a7 += y; }
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
    out[tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else {
    u32 x = (u32)in[tid & 255] | 1u, y = (u32)in[(tid + 7) & 255] | 3u;
    u64 a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7, z = ((u64)y << 32) | x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (OP == 2) { a0 = (u64)x * y + a0; a1 = (u64)x * y + a1; a2 = (u64)x * y + a2; a3 = (u64)x * y + a3;
                       a4 = (u64)x * y + a4; a5 = (u64)x * y + a5; a6 = (u64)x * y + a6; a7 = (u64)x * y + a7; }
        else { a0 += z; a1 += z; a2 += z; a3 += z; a4 += z; a5 += z; a6 += z; a7 += z; }        // 64-bit integer add
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
    u64 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[tid] = (double)(u32)(s ^ (s >> 32));
  }
}

// ---- host restatement of fp.rs:487-609 (6 x 64-bit limbs, schoolbook + HAC 14.32 reduction, canonical result) ------------------
typedef unsigned __int128 u128;
static const u64 PM[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 INV64 = 0x89f3fffcfffcfffdull;
static void ref_mul(const u64 a[6], const u64 b[6], u64 out[6]) {
  u64 t[13] = {0};
  for (int i = 0; i < 6; i++) {
    u64 c = 0;
    for (int j = 0; j < 6; j++) { u128 v = (u128)a[i] * b[j] + t[i + j] + c; t[i + j] = (u64)v; c = (u64)(v >> 64); }
    t[i + 6] = c;
  }
  u64 top = 0;
  for (int i = 0; i < 6; i++) {
    u64 k = t[i] * INV64, c = 0;
    for (int j = 0; j < 6; j++) { u128 v = (u128)k * PM[j] + t[i + j] + c; t[i + j] = (u64)v; c = (u64)(v >> 64); }
    for (int j = i + 6; j < 12 && c; j++) { u128 v = (u128)t[j] + c; t[j] = (u64)v; c = (u64)(v >> 64); }
    top += c;
  }
  u64 r[7] = {t[6], t[7], t[8], t[9], t[10], t[11], top}, d[6];
  u64 bw = 0;
  for (int j = 0; j < 6; j++) { u128 v = (u128)r[j] - PM[j] - bw; d[j] = (u64)v; bw = (u64)(v >> 64) & 1; }
  const bool ge = r[6] || !bw;
  for (int j = 0; j < 6; j++) out[j] = ge ? d[j] : r[j];
}
// 8 x 48-bit doubles <-> 6 x 64-bit words
static void to48(const u64 w[6], double l[8]) {
  for (int j = 0; j < 8; j++) {
    u64 v = 0;
    for (int b = 0; b < 48; b++) { int bit = 48 * j + b; v |= ((w[bit >> 6] >> (bit & 63)) & 1ull) << b; }
    l[j] = (double)v;
  }
}
static bool from48(const double l[8], u64 w[7]) {        // false if a limb is not an integer in [0, 2^48)
  for (int j = 0; j < 7; j++) w[j] = 0;
  for (int j = 0; j < 8; j++) {
    if (!(l[j] >= 0.0 && l[j] < C48) || l[j] != (double)(u64)l[j]) return false;
    u64 v = (u64)l[j];
    for (int b = 0; b < 48; b++) { int bit = 48 * j + b; w[bit >> 6] |= ((v >> b) & 1ull) << (bit & 63); }
  }
  return true;
}

template <class K, class T, class U> static double run(const char* name, K kern, int waves, const T* in, U* out, double per_lane_iter, const char* unit, int iters) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * waves;       // `waves` blocks of 4 wavefronts per CU = `waves` wavefronts per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, 8); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double rate = (double)blocks * 256 * iters * per_lane_iter / (ms * 1e-3);
  printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"%s\": %.4e}\n", name, waves, ms, unit, rate);
  return rate;
}

int main() {
  // inputs: 256 operand pairs below p for the chains (48-bit form and 28-bit form of the SAME values are not needed: the chains only time)
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  std::vector<double> hd(256 * 16);
  std::vector<u32> hu(256 * 32);
  for (int i = 0; i < 256 * 16; i++) { u64 v = next() & ((1ull << 48) - 1); if ((i & 7) == 7) v &= (1ull << 44) - 1; hd[i] = (double)v; }     // < 2^380 < p
  for (auto& v : hu) v = (u32)(next() >> 11);
  double *din, *dout; u32 *uin, *uout;
  hipMalloc(&din, hd.size() * 8); hipMalloc(&dout, (size_t)4096 * 256 * 8 * 8); hipMalloc(&uin, hu.size() * 4); hipMalloc(&uout, (size_t)4096 * 256 * 16 * 4);
  hipMemcpy(din, hd.data(), hd.size() * 8, hipMemcpyHostToDevice); hipMemcpy(uin, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);

  // part 1: issue rates, lane-operations per second on the full chip
  run("v_fma_f64 (8 independent chains)", k_rate<0, 2>, 2, din, dout, 64, "lane_ops_per_s", 2000);
  run("v_fma_f64 (8 independent chains)", k_rate<0, 4>, 4, din, dout, 64, "lane_ops_per_s", 2000);
  run("v_fma_f64 (8 independent chains)", k_rate<0, 8>, 8, din, dout, 64, "lane_ops_per_s", 2000);
  run("v_add_f64", k_rate<1, 8>, 8, din, dout, 64, "lane_ops_per_s", 2000);
  run("v_mad_u64_u32", k_rate<2, 2>, 2, din, dout, 64, "lane_ops_per_s", 2000);
  run("v_mad_u64_u32", k_rate<2, 4>, 4, din, dout, 64, "lane_ops_per_s", 2000);
  run("v_mad_u64_u32", k_rate<2, 8>, 8, din, dout, 64, "lane_ops_per_s", 2000);
  run("64-bit integer add (v_lshl_add_u64)", k_rate<3, 8>, 8, din, dout, 64, "lane_ops_per_s", 2000);

  // part 2: bit-exactness of mul48 against the host restatement of fp.rs on 2^20 operand pairs (uniform below p, plus edge values)
  const size_t n = 1 << 20;
  std::vector<u64> A(n * 6), B(n * 6);
  std::vector<double> a48(n * 8), b48(n * 8), o48(n * 8);
  auto below_p = [&](u64* w) {
    for (;;) {
      for (int j = 0; j < 6; j++) w[j] = next();
      w[5] &= 0x1fffffffffffffffull;
      bool lt = false;
      for (int j = 5; j >= 0; j--) { if (w[j] < PM[j]) { lt = true; break; } if (w[j] > PM[j]) break; }
      if (lt) return;
    }
  };
  for (size_t i = 0; i < n; i++) { below_p(&A[i * 6]); below_p(&B[i * 6]); }
  // edge values: 0, 1, p - 1, 2^380, all-ones limbs below p
  for (int j = 0; j < 6; j++) { A[j] = 0; B[6 + j] = j ? 0 : 1; A[12 + j] = PM[j]; B[12 + j] = PM[j]; A[18 + j] = PM[j]; }
  A[12] -= 1; B[12] -= 1; A[18] -= 1;
  for (size_t i = 0; i < n; i++) { to48(&A[i * 6], &a48[i * 8]); to48(&B[i * 6], &b48[i * 8]); }
  double *da, *db, *dc;
  hipMalloc(&da, n * 64); hipMalloc(&db, n * 64); hipMalloc(&dc, n * 64);
  hipMemcpy(da, a48.data(), n * 64, hipMemcpyHostToDevice); hipMemcpy(db, b48.data(), n * 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mul48_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, dc, n);
  hipMemcpy(o48.data(), dc, n * 64, hipMemcpyDeviceToHost);
  size_t bad = 0, malformed = 0, above_p = 0;
  for (size_t i = 0; i < n; i++) {
    u64 want[6], got[7];
    ref_mul(&A[i * 6], &B[i * 6], want);
    if (!from48(&o48[i * 8], got)) { malformed++; continue; }
    // canonicalise: the kernel returns a value below 2p
    u64 d[6]; u64 bw = 0;
    for (int j = 0; j < 6; j++) { u128 v = (u128)got[j] - PM[j] - bw; d[j] = (u64)v; bw = (u64)(v >> 64) & 1; }
    const bool ge = got[6] || !bw;
    above_p += ge;
    bool same = true;
    for (int j = 0; j < 6; j++) same = same && (ge ? d[j] : got[j]) == want[j];
    bad += !same;
  }
  printf("{\"check\": \"mul48 vs host restatement of fp.rs Fp::mul (R = 2^384)\", \"pairs\": %zu, \"mismatches\": %zu, \"malformed_limbs\": %zu, \"results_in_[p,2p)\": %zu}\n", n, bad, malformed, above_p);

  // part 3: dependent chains
  double r48[3], r28[3]; int w[3] = {2, 4, 8};
  r48[0] = run("8x48 FP64 (this file)", k_chain48<2>, 2, din, dout, 1, "fp_mul_per_s", 2000); r28[0] = run("14x28 shipped", k_chain28<2>, 2, uin, uout, 1, "fp_mul_per_s", 2000);
  r48[1] = run("8x48 FP64 (this file)", k_chain48<4>, 4, din, dout, 1, "fp_mul_per_s", 2000); r28[1] = run("14x28 shipped", k_chain28<4>, 4, uin, uout, 1, "fp_mul_per_s", 2000);
  r48[2] = run("8x48 FP64 (this file)", k_chain48<8>, 8, din, dout, 1, "fp_mul_per_s", 2000); r28[2] = run("14x28 shipped", k_chain28<8>, 8, uin, uout, 1, "fp_mul_per_s", 2000);
  for (int i = 0; i < 3; i++) printf("{\"summary\": \"FP64 / integer\", \"waves_per_simd\": %d, \"ratio\": %.3f}\n", w[i], r48[i] / r28[i]);
  return bad || malformed ? 1 : 0;
}
