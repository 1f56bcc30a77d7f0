// issue_rate.hip -- how fast does ONE wavefront issue?  (development microbenchmark behind the wide path's design)
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/issue_rate.hip -o build/issue_rate && build/issue_rate
// A block of W wavefronts (W = 4: one per SIMD, 8: two per SIMD, 16: four) runs N instructions of one kind, independent or in a
// dependent chain; cycles from s_memtime on wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32;

template <int KIND> __global__ void k(u64* out, u32 seed, int iters) {
  u64 a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  u32 x = seed * 3 + threadIdx.x, y = seed * 5 + 1;
  u32 b0 = x, b1 = x + 1, b2 = x + 2, b3 = x + 3, b4 = x + 4, b5 = x + 5, b6 = x + 6, b7 = x + 7;
  __shared__ __attribute__((aligned(16))) u32 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * seed;
  __syncthreads();
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {            // 8 independent 64-bit multiply-adds
      a0 += (u64)x * y; a1 += (u64)x * b1; a2 += (u64)x * b2; a3 += (u64)x * b3; a4 += (u64)y * b4; a5 += (u64)y * b5; a6 += (u64)y * b6; a7 += (u64)y * b7;
    } else if (KIND == 1) {     // dependent chain of multiply-adds
      a0 += (u64)x * (u32)a0; a0 += (u64)y * (u32)a0; a0 += (u64)x * (u32)a0; a0 += (u64)y * (u32)a0; a0 += (u64)x * (u32)a0; a0 += (u64)y * (u32)a0; a0 += (u64)x * (u32)a0; a0 += (u64)y * (u32)a0;
    } else if (KIND == 2) {     // 8 independent 32-bit adds / xors
      b0 = (b0 + x) ^ y; b1 = (b1 + x) ^ y; b2 = (b2 + x) ^ y; b3 = (b3 + x) ^ y;
    } else if (KIND == 3) {     // dependent 32-bit chain
      b0 = (b0 + x) ^ y; b0 = (b0 + y) ^ x; b0 = (b0 + x) ^ y; b0 = (b0 + y) ^ x;
    } else if (KIND == 4) {     // 8 LDS reads (independent addresses, accumulated)
      b0 += lds[(b1 + 0) & 4095]; b1 += lds[(b2 + 64) & 4095]; b2 += lds[(b3 + 128) & 4095]; b3 += lds[(b4 + 192) & 4095];
      b4 += lds[(b5 + 256) & 4095]; b5 += lds[(b6 + 320) & 4095]; b6 += lds[(b7 + 384) & 4095]; b7 += lds[(b0 + 448) & 4095];
    } else if (KIND == 6) {     // 64-bit LDS atomic adds, no return, every lane its own address (stride 8 B)
      unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
      for (int c = 0; c < 8; c++) atomicAdd(l64 + ((threadIdx.x + 64 * c) & 2047), a0 + c);
    } else if (KIND == 7) {     // the same with 4 lanes of a wavefront on one address
      unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
      for (int c = 0; c < 8; c++) atomicAdd(l64 + (((threadIdx.x >> 2) + 64 * c) & 2047), a0 + c);
    } else if (KIND == 8) {     // plain 64-bit LDS stores
      unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
      for (int c = 0; c < 8; c++) l64[(threadIdx.x + 64 * c) & 2047] = a0 + c;
    } else if (KIND == 9) {     // 128-bit LDS loads, lane-contiguous
      uint4* l128 = reinterpret_cast<uint4*>(lds);
      for (int c = 0; c < 8; c++) { uint4 v = l128[(threadIdx.x + 64 * c + b0) & 1023]; b1 += v.x ^ v.y ^ v.z ^ v.w; }
    } else if (KIND == 10) {    // 64-bit LDS atomic adds, accumulator-like: lanes 4 apart hit columns 4 apart of 32-column rows
      unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
      for (int c = 0; c < 8; c++) atomicAdd(l64 + ((((threadIdx.x >> 2) & 15) * 32 + (threadIdx.x & 3) * 4 + c) & 2047), a0 + c);
    } else if (KIND == 5) {     // 32-bit multiply low + 24-bit mad
      b0 = b0 * x + y; b1 = b1 * x + y; b2 = b2 * x + y; b3 = b3 * x + y; b4 = b4 * y + x; b5 = b5 * y + x; b6 = b6 * y + x; b7 = b7 * y + x;
    }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    asm volatile("" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
  }
  u64 t1 = __builtin_readcyclecounter();
  u64 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s; }
  else if (s == 0x1234567) out[2] = s;
}

template <int KIND> void run(const char* name, int per_iter, u64* d) {
  for (int waves : {1, 4, 8, 16}) {
    const int iters = 2000;
    k<KIND><<<1, 64 * waves>>>(d, 7, iters);
    k<KIND><<<1, 64 * waves>>>(d, 9, iters);
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-28s %2d waves/CU: %6.2f cycles per instruction of one wave\n", name, waves, (double)h[0] / ((double)iters * per_iter));
  }
}

// the same kernel on the whole chip: does the per-wavefront figure survive 256 busy CUs (clocks, power)?
template <int KIND> void run_chip(const char* name, int per_iter, u64* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4, 8, 16}) {
    for (int blocks_per_cu : {1, 2}) {
      const int iters = 20000, blocks = 256 * blocks_per_cu;
      k<KIND><<<blocks, 64 * waves>>>(d, 7, 100);
      hipEventRecord(e0);
      k<KIND><<<blocks, 64 * waves>>>(d, 9, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      const double instr = (double)iters * per_iter;
      printf("%-24s chip, %2d waves/block x %d blocks/CU: %6.2f ticks per instruction of one wave; wall %.3f ms -> %.2f T lane-ops/s, tick = %.3f ns\n", name, waves,
             blocks_per_cu, (double)h[0] / instr, ms, instr * blocks * waves * 64 / (ms * 1e-3) / 1e12, ms * 1e6 / (double)h[0]);
    }
  }
}

int main() {
  u64* d; hipMalloc(&d, 64);
  run_chip<0>("mad_u64_u32 independent", 8, d);
  run_chip<2>("add/xor independent", 8, d);
  run<0>("mad_u64_u32 independent", 8, d);
  run<1>("mad_u64_u32 dependent", 8, d);
  run<2>("add/xor independent", 8, d);
  run<3>("add/xor dependent", 8, d);
  run<4>("ds_read + add", 16, d);
  run<5>("mul_lo + add (mad_u32)", 8, d);
  run<6>("ds_add_u64 distinct", 8, d);
  run<7>("ds_add_u64 4 lanes/address", 8, d);
  run<10>("ds_add_u64 accumulator rows", 8, d);
  run<8>("ds_write_b64", 8, d);
  run<9>("ds_read_b128 + 4 xor", 8, d);
  return 0;
}
