// wide_bench.hip -- standalone timing / parity harness for k_pairing_wide (development tool; the product path is api_pairing.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibls12_381_amd/csrc tools/experiments/wide_bench.hip -o build/wide_bench
//   build/wide_bench bls12_381_amd/wide_prog.bin build/wide_case.bin      (case file: tools/experiments/wide_case.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define WIDE_STANDALONE
#include "wide.hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static std::vector<uint32_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(1); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint32_t> v(n / 4); if (fread(v.data(), 1, n, f) != (size_t)n) exit(1); fclose(f); return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  auto blob = slurp(argv[1]);
  auto cs = slurp(argv[2]);          // n, then n x (g1 24 words, g2 48 words, miller 144 words, gt 144 words)
  const size_t n = cs[0];
  std::vector<uint32_t> g1(n * 24), g2(n * 48), ml(n * 144), gt(n * 144);
  for (size_t i = 0; i < n; i++) {
    const uint32_t* p = cs.data() + 1 + i * 360;
    memcpy(&g1[i * 24], p, 96); memcpy(&g2[i * 48], p + 24, 192); memcpy(&ml[i * 144], p + 72, 576); memcpy(&gt[i * 144], p + 216, 576);
  }
  const size_t N = 1024;
  uint32_t *d_blob, *d_g1, *d_g2, *d_out, *d_f;
  CK(hipMalloc(&d_blob, blob.size() * 4)); CK(hipMemcpy(d_blob, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_g1, N * 96)); CK(hipMalloc(&d_g2, N * 192)); CK(hipMalloc(&d_out, N * 576)); CK(hipMalloc(&d_f, N * 576));
  for (size_t i = 0; i < N; i++) {
    CK(hipMemcpy(d_g1 + i * 24, &g1[(i % n) * 24], 96, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g2 + i * 48, &g2[(i % n) * 48], 192, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_f + i * 144, &ml[(i % n) * 144], 576, hipMemcpyHostToDevice));
  }
  // the blob holds the programs of every built-in configuration: pick the pair generated for this build's (lanes, limbs)
  int cfg = -1;
  for (uint32_t k = 0; k + 1 < blob[1]; k += 2)
    if (blob[blob[2 + 2 * k] + 10] == (uint32_t)bls::WIDE_LANES && blob[blob[2 + 2 * k] + 11] == (uint32_t)bls::WIDE_K) cfg = (int)k;
  if (cfg < 0) { fprintf(stderr, "no program for %d lanes x %d limbs in the blob\n", bls::WIDE_LANES, bls::WIDE_K); return 1; }
  const uint32_t* pm = d_blob + blob[2 + 2 * cfg]; const uint32_t* pf = d_blob + blob[4 + 2 * cfg];
  std::vector<uint32_t> out(N * 144);
  int bad = 0;
  for (int mode = 0; mode < 3; mode++) {
    CK(hipMemset(d_out, 0, N * 576));
    bls::k_pairing_wide<<<N, bls::WIDE_LANES>>>(mode, mode == 2 ? d_f : d_g1, nullptr, d_g2, nullptr, d_out, N, pm, pf);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), d_out, N * 576, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) if (memcmp(&out[i * 144], mode == 1 ? &ml[(i % n) * 144] : &gt[(i % n) * 144], 576)) bad++;
    printf("mode %d: %s\n", mode, bad ? "MISMATCH" : "ok");
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes[] = {1, 64, 256, 512, 768, 1024};
  for (int mode = 0; mode < 3; mode++)
    for (size_t m : sizes) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        bls::k_pairing_wide<<<m, bls::WIDE_LANES>>>(mode, mode == 2 ? d_f : d_g1, nullptr, d_g2, nullptr, d_out, m, pm, pf);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("mode %d n %4zu: %.3f ms\n", mode, m, best);
    }
#ifdef WIDE_PROFILE
  for (int mode = 1; mode < 3; mode++) {
    unsigned long long z[5] = {0, 0, 0, 0, 0}, t[5];
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bls::g_wide_ticks), z, sizeof z));
    bls::k_pairing_wide<<<1, bls::WIDE_LANES>>>(mode, mode == 2 ? d_f : d_g1, nullptr, d_g2, nullptr, d_out, 1, pm, pf);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(bls::g_wide_ticks), sizeof t));
    printf("mode %d ticks of wave 0: phase1 %llu  barrier1 %llu  phase2 %llu  barrier2 %llu  longest phase2 %llu\n", mode, t[0], t[1], t[2], t[3], t[4]);
  }
#endif
  return bad ? 1 : 0;
}
