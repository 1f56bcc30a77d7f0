// tests/simt/hip/hip_runtime.h -- host-side stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY.
//
// The lane-cooperative device code of bls12_381_amd/csrc (pairlane.hip.h, quad.hip.h, pairing code) is plain integer C++
// plus ONE cross-lane primitive (__builtin_amdgcn_update_dpp with a quad permutation).  Compiled for the host with this
// header first on the include path, every lane of a quad becomes a host thread and a DPP move becomes a slot exchange
// between those threads, so the CPU test-suite can run the SAME device functions bit for bit against the oracle
// (tests/test_simt_emulation.py).  Nothing in the product builds against or links this file.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstddef>
#include <cstring>

struct EmuDim3 { unsigned x = 0, y = 0, z = 0; };
extern thread_local EmuDim3 threadIdx, blockIdx, blockDim, gridDim;

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static

// ---- lane group: EMU_LANES host threads in lock step at every cross-lane operation -------------------------------------
#ifndef EMU_LANES
#define EMU_LANES 4
#endif
#include <pthread.h>
struct EmuGroup {
  std::atomic<unsigned> arrived{0};
  std::atomic<unsigned> phase{0};
  int slot[EMU_LANES];
#if EMU_LANES > 8
  // more lanes than host cores: a blocking barrier (spinning threads would starve the ones they wait for)
  pthread_barrier_t pb;
  EmuGroup() { pthread_barrier_init(&pb, nullptr, EMU_LANES); }
  void barrier() { pthread_barrier_wait(&pb); }
#else
  void barrier() {
    unsigned ph = phase.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == EMU_LANES) {
      arrived.store(0, std::memory_order_relaxed);
      phase.store(ph + 1, std::memory_order_release);
    } else {
      while (phase.load(std::memory_order_acquire) == ph) { __builtin_ia32_pause(); }
    }
  }
#endif
};
extern EmuGroup g_emu_group;

// v_mov_b32_dpp with quad_perm:[a,b,c,d] (dpp_ctrl 0..255), all rows / banks enabled: lane L reads lane (L & ~3) | perm[L & 3]
#if EMU_LANES > 8
// a workgroup of many lanes: only the four lanes of a quad meet at a DPP move (other lanes may be in other branches), so every
// quad has a barrier of its own
struct EmuQuadBarriers {
  pthread_barrier_t pb[EMU_LANES / 4];
  EmuQuadBarriers() { for (auto& b : pb) pthread_barrier_init(&b, nullptr, 4); }
};
extern EmuQuadBarriers g_emu_quads;
static inline void emu_dpp_sync(unsigned lane) { pthread_barrier_wait(&g_emu_quads.pb[lane >> 2]); }
#else
static inline void emu_dpp_sync(unsigned) { g_emu_group.barrier(); }
#endif
static inline int emu_update_dpp(int /*old*/, int src, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  const unsigned lane = threadIdx.x % EMU_LANES;
  g_emu_group.slot[lane] = src;
  emu_dpp_sync(lane);
  const unsigned from = (lane & ~3u) | ((unsigned)(ctrl >> (2 * (lane & 3))) & 3u);
  const int v = g_emu_group.slot[from];
  emu_dpp_sync(lane);
  return v;
}
// LDS atomics of the wide kernel (ds_add_u64 / ds_wrxchg_rtn_b64)
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
// status-word flags (global memory)
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
// wave vote: every emulated lane decides for itself (only used for an early loop exit whose extra iterations are identities)
static inline int __all(int p) { return p; }
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
#define __builtin_amdgcn_update_dpp emu_update_dpp
static inline void __syncthreads() { g_emu_group.barrier(); }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_readcyclecounter() 0ull
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
