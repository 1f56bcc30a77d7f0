// tests/simt/emu_pairing.cpp -- the pairing kernels of bls12_381_amd/csrc compiled for the HOST (test infrastructure only):
// every lane of a quad is a host thread, DPP moves are slot exchanges (tests/simt/hip/hip_runtime.h).  The entry points run
// the __global__ functions themselves, one quad per call, so that the CPU suite can compare the very code the GPU runs with
// the oracle (tests/test_simt_emulation.py).
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>

thread_local EmuDim3 threadIdx, blockIdx, blockDim, gridDim;
EmuGroup g_emu_group;
#if EMU_LANES > 8
EmuQuadBarriers g_emu_quads;
#endif

#include "pairing.hip.h"
#ifdef EMU_WITH_QUAD
#include "quad.hip.h"
#endif
#ifdef EMU_WITH_WIDE
#define WIDE_STANDALONE
#include "wide.hip.h"
#endif

using namespace bls;

template <class Fn> static void run_quad(Fn fn) {
  std::vector<std::thread> th;
  for (unsigned l = 0; l < EMU_LANES; l++)
    th.emplace_back([=] { threadIdx.x = l; blockDim.x = EMU_LANES; blockIdx.x = 0; gridDim.x = 1; fn(); });
  for (auto& t : th) t.join();
}

extern "C" {
#ifdef EMU_WITH_WIDE
// one workgroup (EMU_LANES = WIDE_LANES host threads) = one item; mode 0 pairing, 1 Miller value, 2 final exponentiation of g1 (72 u64)
void emu_wide(int mode, const u32* g1, const u32* g2, u32* out, const u32* prog_miller, const u32* prog_fe) {
  run_quad([=] { k_pairing_wide(mode, g1, nullptr, g2, nullptr, out, 1, prog_miller, prog_fe); });
}
#endif
#if EMU_LANES == 4
// pair layout (two pairings per quad): mode 0 pairing, 1 raw Miller value
void emu_pair_pairing(int mode, const u32* g1, const u32* g2, u32* out, size_t n) {
  run_quad([=] { k_pairing(mode, g1, nullptr, g2, nullptr, out, n); });
}
void emu_pair_final_exp(const u32* in, u32* out, size_t n) {
  run_quad([=] { k_final_exp(in, out, n); });
}
void emu_pair_fp12_op(int op, const u32* a, const u32* b, u32* out, size_t n) {
  run_quad([=] { k_fp12_op(op, a, b, out, n); });
}
#ifdef EMU_WITH_QUAD
void emu_quad_pairing(int mode, const u32* g1, const u32* g2, u32* out, size_t n) {
  run_quad([=] { k_pairing_quad(mode, g1, nullptr, g2, nullptr, out, n); });
}
void emu_quad_final_exp(const u32* in, u32* out, size_t n) {
  run_quad([=] { k_final_exp_quad(in, out, n); });
}
void emu_quad_fp12_op(int op, const u32* a, const u32* b, u32* out, size_t n) {
  run_quad([=] { k_fp12_op_quad(op, a, b, out, n); });
}
#endif
#endif
}
