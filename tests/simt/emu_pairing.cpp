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
#include "prep.hip.h"
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
// prep.hip.h: the table of ONE point (tab: PREP_POINT_WORDS words), its coefficients in wire form (68 x 3 x 24 words), and the
// shared-accumulator loop of ONE segment (off -> the segment's two offsets, or nullptr with kuni: the run [0, kuni) of one long product);
// work = kmax * 65 * 4 words
void emu_g2_prepare(const u32* g2, const uint8_t* g2inf, u32* tab, uint8_t* tab_inf) {
  run_quad([=] { k_g2_prepare_quad(g2, g2inf, 1, tab, tab_inf); });
}
void emu_g2_prepared_export(const u32* tab, u32* out) {
  for (unsigned blk = 0; blk < (2 * PREP_STEPS + 3) / 4; blk++) {
    std::vector<std::thread> th;
    for (unsigned l = 0; l < 4; l++)
      th.emplace_back([=] { threadIdx.x = l; blockDim.x = 4; blockIdx.x = blk; gridDim.x = (2 * PREP_STEPS + 3) / 4; k_g2_prepared_export(tab, 0, out); });
    for (auto& t : th) t.join();
  }
}
void emu_mml_prep(const u32* g1, const uint8_t* g1inf, const u32* g2, const uint8_t* g2inf, const u32* qidx, const u32* tab, const uint8_t* tab_inf, u32 tab_n,
                  const unsigned long long* off, size_t total, int kuni, int kmax, u32* work, u32* out, u32* status) {
  u32* wmeta = work;
  uint4* wpp = reinterpret_cast<uint4*>(work + (size_t)kmax * 4);
  uint4* wrr = wpp + (size_t)kmax * 4 * 4;
  run_quad([=] { k_mml_prep_quad(g1, g1inf, g2, g2inf, qidx, tab, tab_inf, tab_n, off, 1, total, kuni, kmax, wmeta, wpp, wrr, out, status); });
}
#endif
#endif
}
