// tools/microbench_mul30.hip -- data-gated experiment: a 13 x 30-bit Montgomery product (R = 2^390,
// 169 + 169 multiply-adds + 13 quotient digits = 351 v_mad_u64_u32, against 406 for the shipped 14 x 28-bit layout) in the same
// dependent-chain harness as the shipped fe_mul_body, at 2, 3, 4 and 8 wavefronts per SIMD.
//
// With 30-bit limbs a column of 13 products reaches 13 * 2^60 = 0.81 * 2^64: the product half and the reduction half of a
// column cannot share one 64-bit accumulator (their sum reaches 2^64.7), so each column keeps TWO accumulators and merges them
// when it hands its carry on: (accP >> 30) + (accR >> 30) + carry of the two low parts.  And operands must arrive normalised
// (limbs < 2^30 exactly): there is no headroom for the lazy additions the 28-bit layout lives on.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bls12_381_amd/csrc tools/microbench_mul30.hip -o build/mb30 && build/mb30
// prints one JSON line per (variant, occupancy) and, first, the value of a 5-step chain for tools/check_mul30.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "fe.hip.h"
using namespace bls;

constexpr int N30 = 13, W30 = 30;
constexpr u32 M30 = (1u << W30) - 1;
struct P30 { u32 l[N30]; };
constexpr P30 make_p30() {
  P30 r{};
  for (int j = 0; j < N30; j++) {
    u64 v = 0;
    for (int b = 0; b < W30; b++) {
      int bit = j * W30 + b, i = bit / LW, s = bit % LW;
      if (i < NL) v |= (u64)((P_L.l[i] >> s) & 1u) << b;
    }
    r.l[j] = (u32)v;
  }
  return r;
}
constexpr P30 P_30 = make_p30();
constexpr u32 make_inv30() {            // -p^-1 mod 2^30 by Newton iteration
  u32 x = 1;
  for (int i = 0; i < 6; i++) x = x * (2u - P_30.l[0] * x);
  return (0u - x) & M30;
}
constexpr u32 INV30 = make_inv30();

struct F30 { u32 l[N30]; };
__device__ __forceinline__ F30 mul30(const F30& a, const F30& b) {
  constexpr P30 p = P_30;
  u32 m[N30];
  F30 r;
  u64 accP = 0, accR = 0;
#pragma unroll
  for (int k = 0; k < N30; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) accP += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) accR += (u64)m[i] * p.l[k - i];
    m[k] = ((((u32)accP + (u32)accR) & M30) * INV30) & M30;
    accR += (u64)m[k] * p.l[0];
    accP = (accP >> W30) + (accR >> W30) + ((((u32)accP & M30) + ((u32)accR & M30)) >> W30);
    accR = 0;
  }
#pragma unroll
  for (int k = N30; k < 2 * N30 - 1; k++) {
#pragma unroll
    for (int i = k - N30 + 1; i < N30; i++) accP += (u64)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - N30 + 1; i < N30; i++) accR += (u64)m[i] * p.l[k - i];
    const u32 lo = ((u32)accP & M30) + ((u32)accR & M30);
    r.l[k - N30] = lo & M30;
    accP = (accP >> W30) + (accR >> W30) + (lo >> W30);
    accR = 0;
  }
  r.l[N30 - 1] = (u32)accP;
  return r;
}
template <int WAVES> __global__ void __launch_bounds__(256, WAVES) k_chain30(u32* __restrict__ out, const u32* __restrict__ in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  F30 a, b;
  for (int j = 0; j < N30; j++) { a.l[j] = in[(tid & 255) * 32 + j] & M30; b.l[j] = in[(tid & 255) * 32 + 16 + j] & M30; }
  a.l[N30 - 1] &= 0xfffff; b.l[N30 - 1] &= 0xfffff;            // values < 2^380 < p
  for (int it = 0; it < iters; it++) { F30 r = mul30(a, b); a = b; b = r; }
  for (int j = 0; j < N30; j++) out[(size_t)tid * 16 + j] = b.l[j];
}
template <int WAVES> __global__ void __launch_bounds__(256, WAVES) k_chain28(u32* __restrict__ out, const u32* __restrict__ in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  v16 a, b;
  for (int j = 0; j < NL; j++) { a[j] = in[(tid & 255) * 32 + j] & LMASK; b[j] = in[(tid & 255) * 32 + 16 + j] & LMASK; }
  a[NL - 1] &= 0xffff; b[NL - 1] &= 0xffff; a[14] = a[15] = b[14] = b[15] = 0;
  for (int it = 0; it < iters; it++) { v16 r = fe_mul_body(a, b); a = b; b = r; }
  for (int j = 0; j < NL; j++) out[(size_t)tid * 16 + j] = b[j];
}
template <class K> static void run(const char* name, K kern, int waves, u32* in, u32* out) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * waves;       // `waves` blocks of 4 wavefronts per CU = `waves` wavefronts per SIMD
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, 8); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"fp_mul_per_s\": %.4e}\n", name, waves, ms, (double)blocks * 256 * iters / (ms * 1e-3));
}
int main() {
  u32 *in, *out; hipMalloc(&in, 256 * 32 * 4); hipMalloc(&out, (size_t)4096 * 256 * 16 * 4);
  u32 h[256 * 32]; unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < 256 * 32; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (u32)(s >> 11); }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  // correctness sample: 5 chain steps of lane 0, checked by tools/check_mul30.py against Python integers
  hipLaunchKernelGGL(k_chain30<2>, dim3(1), dim3(256), 0, 0, out, in, 5); hipDeviceSynchronize();
  u32 r[16]; hipMemcpy(r, out, 64, hipMemcpyDeviceToHost);
  printf("{\"check\": \"chain30\", \"steps\": 5, \"a\": ["); for (int j = 0; j < N30; j++) printf("%u%s", (h[j] & M30) & (j == N30 - 1 ? 0xfffffu : M30), j < N30 - 1 ? ", " : "");
  printf("], \"b\": ["); for (int j = 0; j < N30; j++) printf("%u%s", (h[16 + j] & M30) & (j == N30 - 1 ? 0xfffffu : M30), j < N30 - 1 ? ", " : "");
  printf("], \"out\": ["); for (int j = 0; j < N30; j++) printf("%u%s", r[j], j < N30 - 1 ? ", " : ""); printf("]}\n");
  run("13x30 two accumulators", k_chain30<2>, 2, in, out); run("14x28 shipped", k_chain28<2>, 2, in, out);
  run("13x30 two accumulators", k_chain30<3>, 3, in, out); run("14x28 shipped", k_chain28<3>, 3, in, out);
  run("13x30 two accumulators", k_chain30<4>, 4, in, out); run("14x28 shipped", k_chain28<4>, 4, in, out);
  run("13x30 two accumulators", k_chain30<8>, 8, in, out); run("14x28 shipped", k_chain28<8>, 8, in, out);
  return 0;
}
