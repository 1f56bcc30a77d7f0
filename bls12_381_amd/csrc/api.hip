// api.hip -- host side of libblsgpu.so: the C ABI declared in include/bls12_381_hip.h.
// One context = one device + one stream + grow-only scratch.  All heavy lifting is in the kernels of
// msm.hip.h / pairing.hip.h; this file only sequences launches and moves bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <cstdlib>
#include <vector>
#include <atomic>
#include <thread>
#include <array>
#include <mutex>
#include <condition_variable>
#include <functional>

#include "../../include/bls12_381_hip.h"
#include "msm.hip.h"
#include "pairing.hip.h"
#include "quad.hip.h"
#include "prep.hip.h"
#include "mulbatch.hip.h"
#include "wide.hip.h"
#include <dlfcn.h>
#include "fr.hip.h"
#include "h2c.hip.h"
#include "codec.hip.h"

using namespace bls;

static thread_local std::string g_err;
static int fail(const char* what, hipError_t e, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (api.hip:%d)", what, hipGetErrorString(e), line);
  g_err = buf;
  return BLSGPU_ERR_HIP;
}
static int bad(const char* what) { g_err = what; return BLSGPU_ERR_ARG; }
// A context is driven by ONE host thread at a time (include/bls12_381_hip.h).  Misuse is detected instead of corrupting the slot
// bookkeeping: every public entry point that takes a context claims it for its thread for the duration of the call (re-entrant
// for the same thread: entry points call one another) and fails with BLSGPU_ERR_ARG when another thread is inside.
struct CtxClaim {
  std::atomic<size_t>* owner; int* depth; bool clash = false;
  CtxClaim(std::atomic<size_t>* o, int* d) : owner(o), depth(d) {
    const size_t me = std::hash<std::thread::id>()(std::this_thread::get_id()) | 1;
    size_t cur = owner->load(std::memory_order_acquire);
    if (cur == me) { ++*depth; return; }
    size_t none = 0;
    if (owner->compare_exchange_strong(none, me, std::memory_order_acq_rel)) { *depth = 1; return; }
    clash = true;
  }
  ~CtxClaim() { if (!clash && --*depth == 0) owner->store(0, std::memory_order_release); }
};
#define CTX_CLAIM(c) CtxClaim claim_((c) ? &(c)->owner_thread : &g_no_ctx_owner, (c) ? &(c)->owner_depth : &g_no_ctx_depth); \
  if (claim_.clash) return bad("the context is in use by another host thread (one context per host thread: include/bls12_381_hip.h)"); \
  KtBind ktbind_((c) ? ktimer_of(c) : nullptr)
static thread_local std::atomic<size_t> g_no_ctx_owner{0};
static thread_local int g_no_ctx_depth = 0;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, e_, __LINE__); } while (0)
#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail("kernel launch", e_, __LINE__); } while (0)

// ---- per-kernel timing (diagnostics: blsgpu_kernel_timing / blsgpu_kernel_timing_report) ------------------------------------------
// Every kernel of this file is launched through KLAUNCH.  While a context has timing switched on, the entry points it is passed to
// bracket each of their launches with two HIP events ON THE STREAM THE KERNEL IS LAUNCHED ON (a torch / caller-side event sees only the
// caller's stream, and a call's kernels run on up to four library streams); the report aggregates the durations by kernel name.  Off
// (the default) the cost is one thread-local pointer test per launch.  The timer of the context an entry point was called with is
// bound to the calling thread for the duration of the call (CTX_CLAIM), so group workers time their own members.
struct KTimer {
  struct Rec { const char* name; hipEvent_t a, b; };
  bool on = false;
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
  }
  void clear() { for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); } recs.clear(); }
  void destroy() { clear(); for (auto e : pool) hipEventDestroy(e); pool.clear(); }
};
static thread_local KTimer* g_kt = nullptr;
static inline KTimer* ktimer_of(blsgpu_ctx* c);
struct KtBind {
  KTimer* saved;
  explicit KtBind(KTimer* t) : saved(g_kt) { if (t && t->on) g_kt = t; }
  ~KtBind() { g_kt = saved; }
};
struct KtScope {
  KTimer* t; hipStream_t st; const char* name; hipEvent_t a = nullptr, b = nullptr;
  KtScope(const char* n, hipStream_t s) : t(g_kt), st(s), name(n) {
    if (!t) return;
    a = t->get(); b = t->get();
    if (!a || !b) { if (a) t->pool.push_back(a); if (b) t->pool.push_back(b); t = nullptr; return; }
    hipEventRecord(a, st);
  }
  ~KtScope() { if (t) { hipEventRecord(b, st); t->recs.push_back({name, a, b}); } }
};
#define KLAUNCH(kern, grid, block, lds, stream, ...) \
  do { KtScope kt_(#kern, (stream)); hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__); } while (0)

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) {
      // growing: work queued on any stream may still reference the old block
      if (hipDeviceSynchronize() != hipSuccess) return -1;
      if (hipFree(p) != hipSuccess) return -1;
      p = nullptr; cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) return -1;
    cap = want; return 0;
  }
  void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

#ifndef BLS_NSLOT
#define BLS_NSLOT 4
#endif
constexpr int NSLOT = BLS_NSLOT;       // MSM calls whose tails may be in flight at once (A/B: 3 / 4 / 6 slots)
struct blsgpu_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  int msm_c = 0;
  bool profiling = false;
  bool pipelining = false;
  int acc_timing = 0;                   // blsgpu_msm_accumulate_stats: HIP-event duration of every acc_timing-th accumulation launch (0 = off)
  unsigned acc_tick = 0;
  double acc_ms_sum = 0.0; unsigned acc_count = 0;
  u32 item_cap = 0;                    // A/B hook (env BLSGPU_ITEM_CAP at create): entries per work item of the accumulation (0 = automatic)
  int pairing_layout = 0;              // lanes per pairing of pairing / Miller loop / final exponentiation batches: 0 = automatic (default: a workgroup per pairing --
                                       // wide.hip.h -- up to WIDE_AUTO_MAX items, the quad layout above), 4 = quad (quad.hip.h: next to no hot-loop scratch), 2 = lane pair
                                       // (pairing.hip.h, rounds 1-2), 256 = wide; env BLSGPU_PAIRING_LAYOUT=pair|quad|wide at create fixes one for A/B runs
  u32* d_wide = nullptr;               // the wide programs (bls12_381_amd/wide_prog.bin, generated at build time by tools/gen_wide_prog.py) in device memory
  size_t wide_off[4] = {0, 0, 0, 0};   // word offsets of the Miller-loop / final-exponentiation programs: [0..1] 1024 lanes x 4 limbs, [2..3] 512 lanes x 8 limbs
  int wide_state = 0;                  // 0 = not tried, 1 = loaded, -1 = unavailable (the quad kernels take every size then)
  std::string wide_why;                // ... and why (blsgpu_wide_status)
  int scalar_form = SCALAR_BYTES;      // blsgpu_set_scalar_form: what the scalar arguments of MSM / mul_batch / Gt * Scalar calls hold -- 32 canonical LE bytes (default) or
                                       // the four u64 Montgomery limbs of a `Scalar` (scalar.hip.h); the *_mont entry points switch it for one call
  bool assume_subgroup = false;        // blsgpu_set_assume_subgroup: skip the subgroup check of uploaded bases (the caller vouches for them)
  bool no_glv = false;                 // A/B hook (env BLSGPU_NO_GLV at create): plain 256-bit windows (no GLV for G1, no psi decomposition for G2)
  bool force_slow_sort = false;        // test hook (env BLSGPU_FORCE_SLOW_SORT at create): the global-atomic sort used beyond 2^24 points
  hipStream_t acc_stream = nullptr;     // bucket accumulation of pipelined calls (the caller's stream is never blocked)
  hipEvent_t ev[9] = {};
  u32* d_status = nullptr;              // [0]: sticky "a scalar was not canonical (>= r)" flag of the ASYNCHRONOUS (device-pointer) calls, reported and cleared by blsgpu_synchronize
                                        // (blsgpu_join is a stream-level wait without a host round trip and reports nothing);
                                        // [1]: scratch of the subgroup check; [2]: the flag of the synchronous call in progress (cleared before it, fetched with its result)
  std::atomic<size_t> owner_thread{0};  // CtxClaim: the host thread inside an entry point (0 = none)
  int owner_depth = 0;
  u32* status_word = nullptr;           // where the kernels of the calls being enqueued report: d_status (default) or d_status + 2 inside a synchronous entry point
  float phase_ms[8] = {0};
  // MSM: the chip-filling phases run on `stream`; the latency-bound tail (bucket reduction + window
  // combine, a few wavefronts) of call i runs on tail_stream[i & 1] and overlaps the next call's heavy
  // phases.  Everything the tail touches is double-buffered per slot.
  struct Slot {
    // front: digit sort + work items (LDS/atomic bound)  ->  main stream: bucket accumulation (VALU bound)  ->
    // tail: bucket reduction + window combine (latency bound).  Front and tail run on the slot's own streams so
    // that they overlap the accumulation kernels of neighbouring calls.
    hipStream_t front = nullptr, tail = nullptr, tail2 = nullptr;       // tail2: the T tree sums of the reduction levels (off the critical path)
    hipEvent_t ev_lvl[8] = {}, ev_tree = nullptr;
    hipEvent_t ev_in = nullptr, ev_front = nullptr, ev_acc = nullptr, ev_tail = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;     // around the accumulation kernel (timing enabled), see acc_stats
    bool k_pending = false;
    bool tail_pending = false;
    bool hist_dirty = false;
    unsigned long long seq = 0;
    DevBuf ent, sorted, hist, offs, cursor, bsum, items, heavy, ctrl, glv;
    DevBuf buckets, lvlR[2], lvlT, tsum[2], wacc[2], wsums, result;
  } slot[NSLOT];
  int next_slot = 0;
  unsigned long long msm_calls = 0;
  DevBuf result, io_a, io_b, io_c, io_d, io_e, io_f, io_out, flags_a, flags_b;
  int mmlp_k = 0;                       // A/B hook (env BLSGPU_MMLP_K): terms per accumulator of ONE long product (0 = automatic)
  int mml_impl = 0;                     // A/B hook (env BLSGPU_MML_IMPL): kernel behind blsgpu_multi_miller_loop_device with K > 1 -- 0 = automatic, 1 = k_multi_miller_shared
                                        // (rounds 2-4), 4 = k_mml_prep_quad with no prepared term
  struct BasesCacheEntry { int group; size_t n; uint64_t fp; blsgpu_bases* b; unsigned long long last; };
  std::vector<BasesCacheEntry> bcache;  // blsgpu_set_bases_cache: base arrays of repeated one-shot MSMs kept resident
  int bcache_cap = 0; unsigned long long bcache_tick = 0;
  bool bcache_verify = false;           // blsgpu_set_bases_cache_verify: recognise an array by a hash of ALL its words instead of the 65-point fingerprint
  DevBuf mmlp_work, mmlp_out;           // prepared Miller loops (prep.hip.h): per-quad work area, partial products of one long product
  DevBuf fold_c, fold_d, fold_result;   // scratch of the sums / Fp12 products an asynchronous group fold runs on fold_stream: NOT io_c / io_d / result, which calls on `stream` own
  bool on_fold_stream = false;          // set while partials_fold_device borrows the context: routes proj_sum_device / fp12_product_device to the fold scratch
  hipStream_t fold_stream = nullptr;    // the asynchronous group fold's copies and sums run here, NOT on `stream`: an MSM's front waits for whatever is queued on `stream`
  void* pin_stage = nullptr; hipEvent_t pin_ev[8] = {};      // pinned bounce buffers of staged_upload
  DevBuf gt_one; bool gt_one_ready = false; hipEvent_t ev_gt_one = nullptr;      // the wire form of Fp12::one() (blsgpu_gt_is_identity_device, bulk verification)
  DevBuf ver;                           // bulk verification (blsgpu_bls_verify_batch): every intermediate of the chain
  blsgpu_g2_prepared* ver_table = nullptr;   // ... and the resident `G2Prepared` of -G2 for mode 1
  bool ver_consts_ready = false; hipEvent_t ev_ver = nullptr;
  hipStream_t ver_stream[2] = {nullptr, nullptr}; hipEvent_t ev_ver_side[3] = {};     // the independent stages of the chain run side by side
  DevBuf fb_stage;                      // staging of the one-byte scalars the tables are built from
  DevBuf fb_table[2];                   // fixed-base comb tables of the generators (k_fixed_base): 32 x 256 affine records each, built at first use
  hipEvent_t ev_fb[2] = {};             // recorded where a table was built; awaited by every user (the caller may switch streams)
  bool fb_ready[2] = {false, false};
  DevBuf fr_tw[2], fr_tmp, fr_ninv;     // Fr transform: twiddle tables (forward / inverse), permutation target, n^-1
  int fr_tw_log[2] = {-1, -1};
  int fr_ninv_log = -1;
  KTimer ktimer;                        // blsgpu_kernel_timing
  int h2c_split = -1;                   // -1 by batch size / 0 never / 1 always: BLSGPU_H2C_SPLIT, read when the context is created
  int fr_cols_want = 1;                 // 0 never / 1 from 2^20 elements / 2 always: BLSGPU_NTT_IMPL=stage|cols, read when the context is created
  int fr_cols_ok = -1;                  // k_fr_cols usable on this device (144 KB of dynamic LDS granted); decided at the first transform
  hipEvent_t ev_fr[3] = {};             // twiddles forward / inverse, n^-1: recorded where the table was built, awaited by every user
                                        // (the caller may have switched streams with blsgpu_set_stream in between)
};

static inline KTimer* ktimer_of(blsgpu_ctx* c) { return &c->ktimer; }

struct blsgpu_bases {
  int group = 1; size_t n = 0; int device = 0; u32* rec = nullptr;   // AFF_WORDS per point
  mutable bool ready_seen = false;   // ev_ready has been observed complete: no further waits
  hipEvent_t ev_ready = nullptr;     // recorded behind the kernels that write rec / endo; every MSM waits for it on its own streams
  u32* endo = nullptr;               // G1: the images (BETA x, y) of the records under the GLV endomorphism, same order as rec;
                                     // G2: the four images psi^j(P), j = 0..3, interleaved (record 4 i + j) -- msm.hip.h
  // The images are used only for base sets that lie in the prime-order subgroup: phi(P) = -[z^2]P and psi(P) = [x]P hold
  // there and nowhere else on the curve, while the reference's `multiply` (g1.rs:754-774) is defined for every curve
  // point.  subgroup: 1 = every base passed is_torsion_free on the device (or was built as [k]G), 2 = the caller vouched
  // for the set (blsgpu_set_assume_subgroup), 0 = at least one base is outside the subgroup -> plain windows, no images,
  // 3 = not tested: a one-shot upload (the test would cost more than the split saves) or a set too large for the split anyway (plain windows).
  int subgroup = 0;
  // optional window-shifted tables: table[w * n + i] = [2^(table_c * w)] P_i   (blsgpu_bases_precompute)
  u32* table = nullptr; int table_c = 0, table_w = 0;
};

// Pageable host memory -> device.  hipMemcpyAsync from pageable memory is staged by the runtime on ONE host thread (measured here:
// ~4 GB/s, 7.6 ms for the 32 MB of scalars of a 2^20-point MSM -- twice the MSM itself); large uploads of the host-pointer entry points
// therefore go through pinned bounce buffers of the context, the host-side copy split over up to four threads, every chunk's DMA queued
// on the context's stream as soon as it is staged.  On return everything is queued on the stream (the source may be reused at once).
constexpr size_t STAGE_CHUNK = (size_t)2 << 20;
constexpr int STAGE_THREADS = 4;
static int staged_upload(blsgpu_ctx* c, void* dst, const void* src, size_t bytes) {
  bool direct = bytes < ((size_t)4 << 20);
  if (!direct) {
    // a source the runtime already knows as pinned (hipHostMalloc / hipHostRegister) is DMA-able as it is
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess) direct = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();
  }
  if (direct) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); return BLSGPU_OK; }
  if (!c->pin_stage) {
    if (hipHostMalloc(&c->pin_stage, STAGE_CHUNK * 2 * STAGE_THREADS, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError(); c->pin_stage = nullptr;
      HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); return BLSGPU_OK;        // no pinned memory: the plain path
    }
    for (auto& e : c->pin_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const size_t nchunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
  const int T = (int)(nchunks < (size_t)STAGE_THREADS ? nchunks : (size_t)STAGE_THREADS);
  std::atomic<int> failed{0};
  auto work = [&](int t) {
    if (hipSetDevice(c->device) != hipSuccess) { failed = 1; return; }
    const size_t lo = nchunks * (size_t)t / (size_t)T, hi = nchunks * (size_t)(t + 1) / (size_t)T;
    for (size_t j = lo; j < hi && !failed; j++) {
      const int b = (int)((j - lo) & 1);
      uint8_t* pin = (uint8_t*)c->pin_stage + ((size_t)t * 2 + (size_t)b) * STAGE_CHUNK;
      hipEvent_t ev = c->pin_ev[t * 2 + b];
      if (j - lo >= 2 && hipEventSynchronize(ev) != hipSuccess) { failed = 1; return; }        // the DMA that last read this bounce buffer
      const size_t off = j * STAGE_CHUNK, len = off + STAGE_CHUNK <= bytes ? STAGE_CHUNK : bytes - off;
      memcpy(pin, (const uint8_t*)src + off, len);
      if (hipMemcpyAsync((uint8_t*)dst + off, pin, len, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipEventRecord(ev, c->stream) != hipSuccess) { failed = 1; return; }
    }
  };
  std::vector<std::thread> th;
  try { for (int t = 1; t < T; t++) th.emplace_back(work, t); } catch (...) { failed = 1; }
  if (!failed) work(0);
  for (auto& x : th) x.join();
  // the bounce buffers are reused by the next upload: their last DMAs must have been issued -- and read -- before then
  for (int k = 0; k < 2 * T && !failed; k++) if (hipEventSynchronize(c->pin_ev[k]) != hipSuccess) failed = 1;
  if (failed) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(c->stream);          // DMAs already queued still read the bounce buffers: a retry must not overwrite them
    (void)hipGetLastError();
    g_err = "staged upload failed"; return BLSGPU_ERR_HIP;
  }
  return BLSGPU_OK;
}

template <class F> struct GroupTag;
template <> struct GroupTag<FpPolicy> { static constexpr int id = 1; };
template <> struct GroupTag<Fp2Policy> { static constexpr int id = 2; };

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

// fixed-base scalar multiplication of the generator: rec[i] = affine([k_i] G)     (synthetic inputs)
template <class F> DEV Aff<F> generator();
template <> DEV Aff<FpPolicy> generator<FpPolicy>() {
  constexpr PLimbs gx = {BLS_G1_GEN_X}, gy = {BLS_G1_GEN_Y};
  Aff<FpPolicy> g; g.x = fe1_const(gx); g.y = fe1_const(gy); return g;
}
template <> DEV Aff<Fp2Policy> generator<Fp2Policy>() {
  constexpr PLimbs x0 = {BLS_G2_GEN_X0}, x1 = {BLS_G2_GEN_X1}, y0 = {BLS_G2_GEN_Y0}, y1 = {BLS_G2_GEN_Y1};
  Aff<Fp2Policy> g; g.x.c0 = fe1_const(x0); g.x.c1 = fe1_const(x1); g.y.c0 = fe1_const(y0); g.y.c1 = fe1_const(y1); return g;
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

// ---- field / point self-test kernels ------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fp_op(int op, const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe1 x = fe_from_ref(a + i * 12);
  fe1 y = b ? fe_from_ref(b + i * 12) : x;
  switch (op) {
    case 0: fe_to_ref(mul(x, y), out + i * 12); break;
    case 1: fe_to_ref(add(x, y), out + i * 12); break;
    case 2: fe_to_ref(sub(x, y), out + i * 12); break;
    case 3: fe_to_ref(sqr(x), out + i * 12); break;
    case 4: fe_to_ref(inv(x), out + i * 12); break;
    case 6: fe_to_ref(from_v16<2>(fe_inv_fermat_raw(to_v16(x))), out + i * 12); break;     // x^(p-2): cross-check of op 4
    default: fe_to_ref(neg(x), out + i * 12); break;
  }
}
__global__ void __launch_bounds__(256) k_fp2_op(int op, const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe2_1 x = fe2_from_ref(a + i * 24);
  fe2_1 y = b ? fe2_from_ref(b + i * 24) : x;
  switch (op) {
    case 0: fe2_to_ref(mul(x, y), out + i * 24); break;
    case 1: fe2_to_ref(add(x, y), out + i * 24); break;
    case 2: fe2_to_ref(sub(x, y), out + i * 24); break;
    case 3: fe2_to_ref(sqr(x), out + i * 24); break;
    case 4: fe2_to_ref(inv(x), out + i * 24); break;
    case 5: fe2_to_ref(neg(x), out + i * 24); break;
    default: fe2_to_ref(mul_by_nonresidue(x), out + i * 24); break;
  }
}
template <class F>
__global__ void __launch_bounds__(256) k_point_op(int op, const u32* __restrict__ a, const u32* __restrict__ b, const uint8_t* __restrict__ binf,
                                                  u32* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int WW = Wire<F>::WORDS;
  Proj<F> p;
  p.x = F::st(Wire<F>::load(a + i * 3 * WW)); p.y = F::st(Wire<F>::load(a + i * 3 * WW + WW)); p.z = F::st(Wire<F>::load(a + i * 3 * WW + 2 * WW));
  Proj<F> r;
  if (op == 0) {
    Proj<F> q;
    q.x = F::st(Wire<F>::load(b + i * 3 * WW)); q.y = F::st(Wire<F>::load(b + i * 3 * WW + WW)); q.z = F::st(Wire<F>::load(b + i * 3 * WW + 2 * WW));
    r = pt_add<F>(p, q);
  } else if (op == 1) {
    r = pt_double<F>(p);
  } else {
    Aff<F> q; q.x = Wire<F>::load(b + i * 2 * WW); q.y = Wire<F>::load(b + i * 2 * WW + WW);
    r = pt_add_mixed<F>(p, q, binf ? binf[i] != 0 : false);
  }
  Wire<F>::save(r.x, out + i * 3 * WW); Wire<F>::save(r.y, out + i * 3 * WW + WW); Wire<F>::save(r.z, out + i * 3 * WW + 2 * WW);
}
__global__ void __launch_bounds__(256) k_fp_mul_chain(u32* __restrict__ out, const u32* __restrict__ in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  fe a, b;
  for (int j = 0; j < NL; j++) { a.l[j] = in[(tid & 255) * 28 + j] & LMASK; b.l[j] = in[(tid & 255) * 28 + 14 + j] & LMASK; }
  a.l[NL - 1] &= 0xffff; b.l[NL - 1] &= 0xffff;
  for (int it = 0; it < iters; it++) { fe r = (fe)mul(a, b); a = b; b = r; }
  for (int j = 0; j < NL; j++) out[(size_t)tid * NL + j] = b.l[j];
}
__global__ void __launch_bounds__(256) k_mad_chain(u32* __restrict__ out, const u32* __restrict__ in, int iters) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 x = in[tid & 1023], y = in[(tid + 7) & 1023] | 1;
  u64 a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      a0 = (u64)x * y + a0; a1 = (u64)x * y + a1; a2 = (u64)x * y + a2; a3 = (u64)x * y + a3;
      a4 = (u64)x * y + a4; a5 = (u64)x * y + a5; a6 = (u64)x * y + a6; a7 = (u64)x * y + a7;
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  u64 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[tid] = (u32)s ^ (u32)(s >> 32);
}

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
static inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" const char* blsgpu_last_error(void) { return g_err.c_str(); }
extern "C" int blsgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

static int ctx_init(blsgpu_ctx* c) {
  HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
  for (auto& e : c->ev_fr) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : c->ev_fb) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_gt_one, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_ver, hipEventDisableTiming));
  int prio_lo = 0, prio_hi = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  // A/B hooks for the stream priorities: BLSGPU_PRIO = three characters for accumulation / tail / front, each h, n or l
  // (default "nhl": accumulation normal, tail high, front low)
  int pr[3] = {(prio_lo + prio_hi) / 2, prio_hi, prio_lo};
  if (const char* v = getenv("BLSGPU_PRIO"))
    for (int i = 0; i < 3 && v[i]; i++) pr[i] = v[i] == 'h' ? prio_hi : v[i] == 'l' ? prio_lo : (prio_lo + prio_hi) / 2;
  HIPCHK(hipStreamCreateWithPriority(&c->acc_stream, hipStreamNonBlocking, pr[0]));
  HIPCHK(hipMalloc((void**)&c->d_status, 16));
  HIPCHK(hipMemset(c->d_status, 0, 16));
  c->status_word = c->d_status;
  for (auto& sl : c->slot) {
    // the tail is a handful of wavefronts racing a chip-filling kernel: give its queue the highest priority
    HIPCHK(hipStreamCreateWithPriority(&sl.tail, hipStreamNonBlocking, pr[1]));
    HIPCHK(hipStreamCreateWithPriority(&sl.tail2, hipStreamNonBlocking, pr[1]));
    for (auto& e : sl.ev_lvl) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&sl.ev_tree, hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithPriority(&sl.front, hipStreamNonBlocking, pr[2]));     // sort / items fill the gaps the accumulation leaves
    HIPCHK(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&sl.ev_front, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&sl.ev_acc, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&sl.ev_k0)); HIPCHK(hipEventCreate(&sl.ev_k1));
    HIPCHK(hipEventCreateWithFlags(&sl.ev_tail, hipEventDisableTiming));
  }
  return BLSGPU_OK;
}
extern "C" void blsgpu_destroy(blsgpu_ctx* c);
extern "C" void blsgpu_g2_prepared_free(blsgpu_g2_prepared* p);
extern "C" int blsgpu_create(int device, blsgpu_ctx** out) {
  if (!out) return bad("blsgpu_create: out is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_err = "no HIP device"; return BLSGPU_ERR_NODEV; }
  if (device < 0 || device >= n) return bad("blsgpu_create: device index out of range");
  HIPCHK(hipSetDevice(device));
  blsgpu_ctx* c = new blsgpu_ctx();
  c->device = device;
  c->force_slow_sort = getenv("BLSGPU_FORCE_SLOW_SORT") != nullptr;
  c->no_glv = getenv("BLSGPU_NO_GLV") != nullptr;
  if (const char* v = getenv("BLSGPU_PAIRING_LAYOUT")) {
    // exact names only: a typo must not silently select the slowest kernels
    const std::string s(v);
    if (s == "auto" || s == "0" || s.empty()) c->pairing_layout = 0;
    else if (s == "pair" || s == "2") c->pairing_layout = 2;
    else if (s == "quad" || s == "4") c->pairing_layout = 4;
    else if (s == "wide" || s == "256") c->pairing_layout = 256;
    else { delete c; return bad("blsgpu_create: BLSGPU_PAIRING_LAYOUT must be one of auto, pair, quad, wide"); }
  }
  if (const char* v = getenv("BLSGPU_MMLP_K")) { long k = atol(v); if (k >= 1 && k <= MMLP_MAX_K) c->mmlp_k = (int)k; }
  if (const char* v = getenv("BLSGPU_MML_IMPL")) { long k = atol(v); if (k == 1 || k == 4) c->mml_impl = (int)k; }
  if (const char* v = getenv("BLSGPU_H2C_SPLIT")) c->h2c_split = atoi(v) ? 1 : 0;
  if (const char* v = getenv("BLSGPU_NTT_IMPL")) c->fr_cols_want = !strcmp(v, "cols") ? 2 : !strcmp(v, "stage") ? 0 : 1;      // cols: at every size (tests)
  if (const char* v = getenv("BLSGPU_ITEM_CAP")) { long k = atol(v); if (k >= 8 && k <= ITEM_CAP_MAX) c->item_cap = (u32)k; }
  int rc = ctx_init(c);
  if (rc != BLSGPU_OK) { blsgpu_destroy(c); return rc; }       // destroy tolerates the half-built context (null handles are skipped)
  *out = c;
  return BLSGPU_OK;
}
extern "C" void blsgpu_destroy(blsgpu_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  if (c->d_status) hipFree(c->d_status);
  if (c->d_wide) hipFree(c->d_wide);
  DevBuf* bufs[] = {&c->result, &c->io_a, &c->io_b, &c->io_c, &c->io_d, &c->io_e, &c->io_f, &c->io_out, &c->flags_a, &c->flags_b, &c->fr_tw[0], &c->fr_tw[1], &c->fr_tmp, &c->fr_ninv, &c->fb_table[0], &c->fb_table[1], &c->fb_stage, &c->mmlp_work, &c->mmlp_out, &c->gt_one, &c->ver, &c->fold_c, &c->fold_d, &c->fold_result};
  for (auto b : bufs) b->release();
  for (auto& sl : c->slot) {
    DevBuf* sb[] = {&sl.ent, &sl.sorted, &sl.hist, &sl.offs, &sl.cursor, &sl.bsum, &sl.items, &sl.heavy, &sl.ctrl, &sl.glv,
                    &sl.buckets, &sl.lvlR[0], &sl.lvlR[1], &sl.lvlT, &sl.tsum[0], &sl.tsum[1], &sl.wacc[0], &sl.wacc[1], &sl.wsums, &sl.result};
    for (auto b : sb) b->release();
    hipEvent_t evs[] = {sl.ev_in, sl.ev_front, sl.ev_acc, sl.ev_tail, sl.ev_k0, sl.ev_k1, sl.ev_tree};
    for (auto e : evs) if (e) hipEventDestroy(e);
    for (auto& e : sl.ev_lvl) if (e) hipEventDestroy(e);
    hipStream_t sts[] = {sl.front, sl.tail, sl.tail2};
    for (auto q : sts) if (q) hipStreamDestroy(q);
  }
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  for (auto& e : c->ev_fr) if (e) hipEventDestroy(e);
  for (auto& e : c->ev_fb) if (e) hipEventDestroy(e);
  if (c->ev_gt_one) hipEventDestroy(c->ev_gt_one);
  if (c->fold_stream) hipStreamDestroy(c->fold_stream);
  if (c->pin_stage) hipHostFree(c->pin_stage);
  for (auto e : c->pin_ev) if (e) hipEventDestroy(e);
  if (c->ev_ver) hipEventDestroy(c->ev_ver);
  for (auto q : c->ver_stream) if (q) hipStreamDestroy(q);
  for (auto e : c->ev_ver_side) if (e) hipEventDestroy(e);
  if (c->ver_table) blsgpu_g2_prepared_free(c->ver_table);
  for (auto& e : c->bcache) if (e.b) blsgpu_bases_free(e.b);
  c->ktimer.destroy();
  if (c->acc_stream) hipStreamDestroy(c->acc_stream);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}
extern "C" int blsgpu_set_stream(blsgpu_ctx* c, void* s) { CTX_CLAIM(c);
  if (!c) return bad("ctx is NULL");
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return BLSGPU_OK;
}
// Read and clear the sticky input-validation flags (all queued work must have finished).  Bit 0: an MSM scalar was not
// canonical (>= r): the result of that call is unspecified, as the reference offers no such value (Scalar::from_bytes -> None).
static int take_status(blsgpu_ctx* c) {
  u32 st = 0;
  HIPCHK(hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
  if (st) {
    HIPCHK(hipMemset(c->d_status, 0, 4));
    if (st & 2u) return bad("multi_miller_loop_many_device: a segment is longer than the max_seg_terms the caller passed (its value is unspecified)");
    if (st & 4u) return bad("multi_miller_loop_prepared: a q_index lies outside the prepared table (the term was skipped)");
    return bad("msm: a scalar is not canonical (>= r); Scalar::to_bytes never produces such bytes (scalar.rs:284-296)");
  }
  return BLSGPU_OK;
}
// A synchronous entry point owns its verdict: its kernels report into d_status[2] (cleared on the caller's stream before anything
// of the call is enqueued), the word comes back with the call's result, and neither an earlier asynchronous call's flag is
// blamed on this call nor is it cleared by it.
struct SyncStatus {
  blsgpu_ctx* c; u32 host = 0;
  explicit SyncStatus(blsgpu_ctx* c_) : c(c_) { c->status_word = c->d_status + 2; }
  ~SyncStatus() { c->status_word = c->d_status; }
  int begin() { HIPCHK(hipMemsetAsync(c->d_status + 2, 0, 4, c->stream)); return BLSGPU_OK; }
  int fetch() { HIPCHK(hipMemcpyAsync(&host, c->d_status + 2, 4, hipMemcpyDeviceToHost, c->stream)); return BLSGPU_OK; }      // then synchronise the stream
  int verdict() const { return host ? bad("msm: a scalar is not canonical (>= r); Scalar::to_bytes never produces such bytes (scalar.rs:284-296)") : BLSGPU_OK; }
};
// one call with another scalar form than the context's setting (the *_mont entry points; the byte-format compositions pin SCALAR_BYTES)
struct ScalarFormScope {
  blsgpu_ctx* c; int saved;
  ScalarFormScope(blsgpu_ctx* c_, int form) : c(c_), saved(c_ ? c_->scalar_form : 0) { if (c) c->scalar_form = form; }
  ~ScalarFormScope() { if (c) c->scalar_form = saved; }
};
extern "C" int blsgpu_set_scalar_form(blsgpu_ctx* c, int form) { CTX_CLAIM(c);
  if (!c) return bad("ctx is NULL");
  if (form != SCALAR_BYTES && form != SCALAR_MONT) return bad("scalar form must be BLSGPU_SCALAR_BYTES (0) or BLSGPU_SCALAR_MONT (1)");
  c->scalar_form = form; return BLSGPU_OK;
}
extern "C" int blsgpu_synchronize(blsgpu_ctx* c) { CTX_CLAIM(c);
  if (!c) return bad("ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipStreamSynchronize(c->acc_stream));
  for (auto& sl : c->slot) { HIPCHK(hipStreamSynchronize(sl.front)); HIPCHK(hipStreamSynchronize(sl.tail)); HIPCHK(hipStreamSynchronize(sl.tail2)); sl.tail_pending = false; }
  if (c->fold_stream) HIPCHK(hipStreamSynchronize(c->fold_stream));
  return take_status(c);
}
// fold the finished accumulation timings into the running statistics (never blocks)
static void acc_harvest(blsgpu_ctx* c, bool wait) {
  for (auto& sl : c->slot) {
    if (!sl.k_pending) continue;
    if (wait) hipEventSynchronize(sl.ev_k1);
    else if (hipEventQuery(sl.ev_k1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sl.ev_k0, sl.ev_k1) == hipSuccess) { c->acc_ms_sum += ms; c->acc_count++; }
    sl.k_pending = false;
  }
}
extern "C" int blsgpu_msm_accumulate_stats(blsgpu_ctx* c, int enable, double* avg_ms, unsigned* launches) { CTX_CLAIM(c);
  if (!c) return bad("ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  acc_harvest(c, true);
  if (avg_ms) *avg_ms = c->acc_count ? c->acc_ms_sum / c->acc_count : 0.0;
  if (launches) *launches = c->acc_count;
  c->acc_ms_sum = 0.0; c->acc_count = 0;
  c->acc_timing = enable < 0 ? 0 : enable; c->acc_tick = 0;
  return BLSGPU_OK;
}
// diagnostics: HIP-event duration of every kernel the context's entry points launch (see KLAUNCH)
extern "C" int blsgpu_kernel_timing(blsgpu_ctx* c, int enable) {
  if (!c) return bad("ctx is NULL");
  CtxClaim claim_(&c->owner_thread, &c->owner_depth);
  if (claim_.clash) return bad("the context is in use by another host thread");
  HIPCHK(hipSetDevice(c->device));
  if (!c->ktimer.recs.empty()) HIPCHK(hipDeviceSynchronize());     // events still in flight go back to the pool
  c->ktimer.clear();
  c->ktimer.on = enable != 0;
  return BLSGPU_OK;
}
// One line per kernel name in first-launch order: "name<TAB>launches<TAB>total_ms<TAB>min_ms<TAB>max_ms\n".  Waits for the timed launches,
// writes at most cap - 1 characters + NUL, stores the full length in *needed (may be NULL), and clears the records.
extern "C" int blsgpu_kernel_timing_report(blsgpu_ctx* c, char* buf, size_t cap, size_t* needed) {
  if (!c || (cap && !buf)) return bad("kernel_timing_report: NULL argument");
  CtxClaim claim_(&c->owner_thread, &c->owner_depth);
  if (claim_.clash) return bad("the context is in use by another host thread");
  HIPCHK(hipSetDevice(c->device));
  struct Agg { const char* name; unsigned n; double tot, mn, mx; };
  std::vector<Agg> agg;
  for (auto& r : c->ktimer.recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); continue; }
    Agg* a = nullptr;
    for (auto& x : agg) if (!strcmp(x.name, r.name)) { a = &x; break; }
    if (!a) { agg.push_back({r.name, 0, 0.0, 1e300, 0.0}); a = &agg.back(); }
    a->n++; a->tot += ms; if (ms < a->mn) a->mn = ms; if (ms > a->mx) a->mx = ms;
  }
  c->ktimer.clear();
  std::string out;
  char line[512];
  for (auto& a : agg) { snprintf(line, sizeof line, "%s\t%u\t%.6f\t%.6f\t%.6f\n", a.name, a.n, a.tot, a.mn, a.mx); out += line; }
  if (needed) *needed = out.size();
  if (cap) { size_t k = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), k); buf[k] = 0; }
  return BLSGPU_OK;
}
extern "C" int blsgpu_set_pipelining(blsgpu_ctx* c, int on) { CTX_CLAIM(c); if (!c) return bad("ctx is NULL"); c->pipelining = on != 0; return BLSGPU_OK; }
extern "C" int blsgpu_join_lag(blsgpu_ctx* c, int lag) { CTX_CLAIM(c);
  if (!c || lag < 0) return bad("join: bad argument");
  HIPCHK(hipSetDevice(c->device));
  for (auto& sl : c->slot)
    if (sl.tail_pending && sl.seq + (unsigned long long)lag <= c->msm_calls) HIPCHK(hipStreamWaitEvent(c->stream, sl.ev_tail, 0));
  return BLSGPU_OK;
}
extern "C" int blsgpu_join(blsgpu_ctx* c) { CTX_CLAIM(c); return blsgpu_join_lag(c, 0); }
extern "C" int blsgpu_set_msm_window(blsgpu_ctx* c, int w) { CTX_CLAIM(c);
  if (!c) return bad("ctx is NULL");
  // 16 is the widest window of the LDS counting sort (8 coarse + 7 fine key bits); wider windows exist only with
  // resident tables (blsgpu_bases_precompute), which carry their own width
  if (w != 0 && (w < 4 || w > 16)) return bad("msm window must be 0 or in [4,16]");
  c->msm_c = w; return BLSGPU_OK;
}
extern "C" int blsgpu_set_profiling(blsgpu_ctx* c, int on) { CTX_CLAIM(c); if (!c) return bad("ctx is NULL"); c->profiling = on != 0; return BLSGPU_OK; }
extern "C" int blsgpu_last_msm_phase_ms(blsgpu_ctx* c, int phase, float* ms) { CTX_CLAIM(c);
  if (!c || !ms || phase < 0 || phase > 7) return bad("bad phase query");
  *ms = c->phase_ms[phase]; return BLSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// bases
// ---------------------------------------------------------------------------------------------------
// G1 bases also keep their images under the GLV endomorphism next to them (2x the resident memory; see k_glv_decompose);
// G2 bases keep P, psi(P), psi^2(P), psi^3(P) interleaved in a second array (5x the resident memory; see k_gls_decompose)
static void bases_drop(blsgpu_bases* b) {        // error paths of the constructors: nothing queued can still matter
  if (b->rec) hipFree(b->rec);
  if (b->endo) hipFree(b->endo);
  if (b->table) hipFree(b->table);
  if (b->ev_ready) hipEventDestroy(b->ev_ready);
  delete b;
}
static int bases_make_endo(blsgpu_ctx* c, blsgpu_bases* b, bool trusted) {
  if (!b->n) { b->subgroup = 1; return BLSGPU_OK; }
  // G2 sets with 4 n beyond the sort's 24-bit indices never take the split: no images, plain windows (exact for any curve
  // point), so there is nothing to test either
  if (b->group == 2 && b->n > ((size_t)1 << 22)) { b->subgroup = trusted ? 1 : (c->assume_subgroup ? 2 : 3); return BLSGPU_OK; }
  if (trusted) b->subgroup = 1;
  else if (c->assume_subgroup) b->subgroup = 2;
  else {
    // one pass of the reference's own subgroup test over the set (G1 ~2 k, G2 ~6 k field multiplications per point, once per
    // upload): the result decides on the host whether images are built, so this synchronises the context's stream
    u32 nbad = 0;
    HIPCHK(hipMemsetAsync(c->d_status + 1, 0, 4, c->stream));
    if (b->group == 1) KLAUNCH(k_bases_subgroup_check<FpPolicy>, dim3(nblk(b->n, 128)), dim3(128), 0, c->stream, b->rec, b->n, c->d_status + 1);
    else KLAUNCH(k_bases_subgroup_check<Fp2Policy>, dim3(nblk(b->n, 128)), dim3(128), 0, c->stream, b->rec, b->n, c->d_status + 1);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(&nbad, c->d_status + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nbad) { b->subgroup = 0; return BLSGPU_OK; }
    b->subgroup = 1;
  }
  // the images are an accelerator, not a requirement: without memory for them the MSM runs on plain 256-bit windows
  const size_t bytes = (b->group == 1 ? b->n * Store<FpPolicy>::AFF_WORDS : 4 * b->n * Store<Fp2Policy>::AFF_WORDS) * 4;
  if (hipMalloc((void**)&b->endo, bytes) != hipSuccess) { (void)hipGetLastError(); b->endo = nullptr; return BLSGPU_OK; }
  if (b->group == 1) KLAUNCH(k_bases_endo, dim3(nblk(b->n, 256)), dim3(256), 0, c->stream, b->rec, b->endo, b->n);
  else KLAUNCH(k_bases_endo_g2, dim3(nblk(b->n, 256)), dim3(256), 0, c->stream, b->rec, b->endo, b->n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(b->ev_ready, c->stream);     // MSMs on another stream (blsgpu_set_stream) wait for the records and images
  if (e != hipSuccess) { hipFree(b->endo); b->endo = nullptr; return fail("k_bases_endo", e, __LINE__); }
  return BLSGPU_OK;
}
// oneshot: the set serves exactly one MSM (blsgpu_g{1,2}_msm_host / *_msm_bytes).  The subgroup test costs ~2 k (G1) / ~6 k (G2)
// field multiplications per point -- several times the MSM it would speed up (2^20 G1 points: 23 ms of test for a 3 ms MSM) -- so
// unless the caller vouches for the set it is NOT tested and keeps no images: the call runs on plain 256-bit windows, which are the
// complete-formula bucket method and exact for every curve point (state 3).  Resident uploads amortise the test over their MSMs.
template <class F>
static int bases_import(blsgpu_ctx* c, const void* d_xy, const void* d_inf, size_t n, blsgpu_bases** out, bool oneshot = false) {
  blsgpu_bases* b = new blsgpu_bases();
  b->group = GroupTag<F>::id; b->n = n; b->device = c->device;
  size_t bytes = (n ? n : 1) * Store<F>::AFF_WORDS * 4;
  if (hipEventCreateWithFlags(&b->ev_ready, hipEventDisableTiming) != hipSuccess) { delete b; g_err = "hipEventCreate(bases) failed"; return BLSGPU_ERR_HIP; }
  if (hipMalloc((void**)&b->rec, bytes) != hipSuccess) { bases_drop(b); g_err = "hipMalloc(bases) failed"; return BLSGPU_ERR_HIP; }
  if (n) {
    KLAUNCH(k_bases_import<F>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (const u32*)d_xy, (const uint8_t*)d_inf, b->rec, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(b->ev_ready, c->stream);
    if (e != hipSuccess) { bases_drop(b); return fail("k_bases_import", e, __LINE__); }
  }
  if (oneshot && !c->assume_subgroup) { b->subgroup = 3; *out = b; return BLSGPU_OK; }
  if (int rc = bases_make_endo(c, b, false)) { bases_drop(b); return rc; }
  *out = b;
  return BLSGPU_OK;
}
template <class F>
static int bases_upload(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, blsgpu_bases** out, bool oneshot = false) {
  if (!c || !out || (n && !xy)) return bad("bases_upload: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  size_t xb = n * 2 * Wire<F>::WORDS * 4;
  if (c->io_a.reserve(xb ? xb : 16) || c->flags_a.reserve(n ? n : 16)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (n) { int ru = staged_upload(c, c->io_a.p, xy, xb); if (ru) return ru; }
  if (n && inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, inf, n, hipMemcpyHostToDevice, c->stream));
  int rc = bases_import<F>(c, c->io_a.p, inf ? c->flags_a.p : nullptr, n, out, oneshot);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_bases_upload(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, blsgpu_bases** out) { CTX_CLAIM(c); return bases_upload<FpPolicy>(c, xy, inf, n, out); }
extern "C" int blsgpu_g2_bases_upload(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, blsgpu_bases** out) { CTX_CLAIM(c); return bases_upload<Fp2Policy>(c, xy, inf, n, out); }
extern "C" int blsgpu_g1_bases_from_device(blsgpu_ctx* c, const void* xy, const void* inf, size_t n, blsgpu_bases** out) { CTX_CLAIM(c);
  if (!c || !out || (n && !xy)) return bad("bases_from_device: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  return bases_import<FpPolicy>(c, xy, inf, n, out);
}
extern "C" int blsgpu_g2_bases_from_device(blsgpu_ctx* c, const void* xy, const void* inf, size_t n, blsgpu_bases** out) { CTX_CLAIM(c);
  if (!c || !out || (n && !xy)) return bad("bases_from_device: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  return bases_import<Fp2Policy>(c, xy, inf, n, out);
}
extern "C" int blsgpu_bases_from_scalars(blsgpu_ctx* c, int group, const uint8_t* scalars, size_t n, blsgpu_bases** out) { CTX_CLAIM(c);
  if (!c || !out || (n && !scalars) || (group != 1 && group != 2)) return bad("bases_from_scalars: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n ? n * 32 : 16)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (n) HIPCHK(hipMemcpyAsync(c->io_a.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  blsgpu_bases* b = new blsgpu_bases();
  b->group = group; b->n = n; b->device = c->device;
  size_t words = group == 1 ? Store<FpPolicy>::AFF_WORDS : Store<Fp2Policy>::AFF_WORDS;
  if (hipEventCreateWithFlags(&b->ev_ready, hipEventDisableTiming) != hipSuccess) { delete b; g_err = "hipEventCreate(bases) failed"; return BLSGPU_ERR_HIP; }
  if (hipMalloc((void**)&b->rec, (n ? n : 1) * words * 4) != hipSuccess) { bases_drop(b); g_err = "hipMalloc(bases) failed"; return BLSGPU_ERR_HIP; }
  bool fb_built_now = false;
  if (n) {
    // below a few thousand multiples the double-and-add kernel is as fast as building the comb table would be
    const bool comb = n >= 4096 || c->fb_ready[group - 1];
    if (comb && !c->fb_ready[group - 1]) {
      // table[w * 256 + d] = [d * 2^(8 w)] G: 8 192 scalars with one non-zero byte each, through the double-and-add kernel, once per context
      DevBuf& tb = c->fb_table[group - 1];
      std::vector<uint8_t> one_byte((size_t)8192 * 32, 0);
      for (int w = 0; w < 32; w++) for (int d = 0; d < 256; d++) one_byte[((size_t)w * 256 + d) * 32 + w] = (uint8_t)d;
      // staged through a buffer of its own: io_c is the scratch of the asynchronous multi_miller_loop_many_device, whose partial products
      // may still be in flight on another stream (blsgpu_set_stream)
      if (tb.reserve((size_t)8192 * words * 4) || c->fb_stage.reserve((size_t)8192 * 32)) { bases_drop(b); g_err = "hipMalloc(fixed-base table) failed"; return BLSGPU_ERR_HIP; }
      hipError_t e = hipMemcpyAsync(c->fb_stage.p, one_byte.data(), one_byte.size(), hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);                 // `one_byte` lives on this frame
      if (e != hipSuccess) { bases_drop(b); return fail("fixed-base table upload", e, __LINE__); }
      if (group == 1) KLAUNCH(k_bases_from_scalars<FpPolicy>, dim3(nblk(8192, 256)), dim3(256), 0, c->stream, c->fb_stage.as<u32>(), tb.as<u32>(), (size_t)8192);
      else KLAUNCH(k_bases_from_scalars<Fp2Policy>, dim3(nblk(8192, 256)), dim3(256), 0, c->stream, c->fb_stage.as<u32>(), tb.as<u32>(), (size_t)8192);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipEventRecord(c->ev_fb[group - 1], c->stream);
      if (e != hipSuccess) { bases_drop(b); return fail("fixed-base table build", e, __LINE__); }
      fb_built_now = true;                 // marked ready only once the build is known to have run (the synchronisation at the end of this call)
    }
    if (comb) {
      hipError_t e = hipStreamWaitEvent(c->stream, c->ev_fb[group - 1], 0);
      if (e != hipSuccess) { bases_drop(b); return fail("fixed-base table wait", e, __LINE__); }
      if (group == 1) KLAUNCH(k_fixed_base<FpPolicy>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->io_a.as<u32>(), c->fb_table[0].as<u32>(), b->rec, n);
      else KLAUNCH(k_fixed_base<Fp2Policy>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->io_a.as<u32>(), c->fb_table[1].as<u32>(), b->rec, n);
    } else if (group == 1) KLAUNCH(k_bases_from_scalars<FpPolicy>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->io_a.as<u32>(), b->rec, n);
    else KLAUNCH(k_bases_from_scalars<Fp2Policy>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->io_a.as<u32>(), b->rec, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(b->ev_ready, c->stream);
    if (e != hipSuccess) { bases_drop(b); return fail("k_bases_from_scalars", e, __LINE__); }
  }
  if (int rc = bases_make_endo(c, b, true)) { bases_drop(b); return rc; }      // [k]G lies in the subgroup by construction
  {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { bases_drop(b); return fail("bases_from_scalars", e, __LINE__); }     // (a table built in this call stays unmarked: rebuilt next time)
  }
  if (fb_built_now) c->fb_ready[group - 1] = true;
  *out = b;
  return BLSGPU_OK;
}
extern "C" size_t blsgpu_bases_len(const blsgpu_bases* b) { return b ? b->n : 0; }
extern "C" int blsgpu_bases_subgroup_state(const blsgpu_bases* b) { return b ? b->subgroup : 0; }
extern "C" int blsgpu_set_assume_subgroup(blsgpu_ctx* c, int on) { CTX_CLAIM(c); if (!c) return bad("ctx is NULL"); c->assume_subgroup = on != 0; return BLSGPU_OK; }
extern "C" void blsgpu_bases_free(blsgpu_bases* b) {
  if (!b) return;
  hipSetDevice(b->device);
  hipDeviceSynchronize();                  // an asynchronous MSM may still be reading the records
  bases_drop(b);
}
template <class F>
static int bases_precompute(blsgpu_ctx* c, blsgpu_bases* b, int cw) {
  const int nwin = (256 + cw - 1) / cw;
  if ((size_t)nwin * b->n > ((size_t)1 << 24)) return bad("bases_precompute: n * windows must not exceed 2^24");
  if (b->table) { HIPCHK(hipFree(b->table)); b->table = nullptr; b->table_c = 0; }
  size_t bytes = (size_t)nwin * (b->n ? b->n : 1) * Store<F>::AFF_WORDS * 4;
  HIPCHK(hipMalloc((void**)&b->table, bytes));
  if (b->n) {
    KLAUNCH(k_bases_precompute<F>, dim3(nblk(b->n, 256)), dim3(256), 0, c->stream, b->rec, b->table, b->n, cw, nwin);
    LAUNCHCHK();
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  b->table_c = cw; b->table_w = nwin;
  return BLSGPU_OK;
}
extern "C" int blsgpu_bases_precompute(blsgpu_ctx* c, blsgpu_bases* b, int window_bits) { CTX_CLAIM(c);
  if (!c || !b) return bad("bases_precompute: NULL argument");
  if (window_bits == 0) window_bits = 20;
  if (window_bits < 9 || window_bits > 21) return bad("bases_precompute: window must be in [9, 21]");
  HIPCHK(hipSetDevice(c->device));
  return b->group == 1 ? bases_precompute<FpPolicy>(c, b, window_bits) : bases_precompute<Fp2Policy>(c, b, window_bits);
}

template <class F>
static int bases_download(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, size_t count, uint64_t* xy, uint8_t* inf) {
  size_t xb = count * 2 * Wire<F>::WORDS * 4;
  if (c->io_out.reserve(xb ? xb : 16) || c->flags_b.reserve(count ? count : 16)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (!count) return BLSGPU_OK;
  KLAUNCH(k_bases_export<F>, dim3(nblk(count, 256)), dim3(256), 0, c->stream, b->rec + first * Store<F>::AFF_WORDS, c->io_out.as<u32>(),
                     c->flags_b.as<uint8_t>(), count);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(xy, c->io_out.p, xb, hipMemcpyDeviceToHost, c->stream));
  if (inf) HIPCHK(hipMemcpyAsync(inf, c->flags_b.p, count, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_bases_download(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, size_t count, uint64_t* xy, uint8_t* inf) { CTX_CLAIM(c);
  if (!c || !b || (count && !xy) || first > b->n || count > b->n - first) return bad("bases_download: bad argument");
  if (b->device != c->device) return bad("bases_download: bases live on another device than the context");
  HIPCHK(hipSetDevice(c->device));
  return b->group == 1 ? bases_download<FpPolicy>(c, b, first, count, xy, inf) : bases_download<Fp2Policy>(c, b, first, count, xy, inf);
}

// ---------------------------------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------------------------------
static int pick_window(size_t n, int bits) {
  // minimise  n*W (mixed adds) + 2*W*2^(c-1)*1.2 (bucket reduction), W = ceil(bits/c); n = scalars of `bits` bits
  int best = 8; double bc = 1e300;
  for (int c = 6; c <= 16; c++) {
    int W = (bits + c - 1) / c;
    double cost = (double)n * W + 2.4 * W * (double)(1u << (c - 1));
    if (cost < bc) { bc = cost; best = c; }
  }
  return best;
}

// a reduction level runs one chain per TEAM of lanes when that still fits comfortably on the chip
constexpr size_t TEAM_LANES_MAX = 131072;
#define TEAM_LDS(threads) ((size_t)team_lds_words<F>(threads) * 4)

template <class F>
static int msm_device(blsgpu_ctx* c, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, void* d_out_wire) {
  if (!c || !bases || !d_out_wire || (n && !d_scalars)) return bad("msm: NULL argument");
  if (bases->group != GroupTag<F>::id) return bad("msm: bases belong to the other group");
  if (first > bases->n || n > bases->n - first) return bad("msm: range exceeds the resident bases");
  if (bases->device != c->device) return bad("msm: bases live on another device than the context");
  if (n > ((size_t)1 << 27)) return bad("msm: n too large for one call (shard the input)");
  HIPCHK(hipSetDevice(c->device));
  // Calls beyond the sort's index width with the endomorphism split (G1: 2 n > 2^24, G2: 4 n > 2^24) run on plain windows.
  // Cutting them into passes that each keep the split was measured (round 3, 2^24 G1 points on one MI355X: two GLV passes
  // 52.6 ms against 45.9 ms for one plain pass): the split halves the WINDOWS, not the bucket additions, and at this size
  // the additions are everything -- two tails and gathers over twice the memory only add to them.
  hipStream_t st = c->stream;
  constexpr int PW = Store<F>::PROJ_WORDS;
  if (c->result.reserve(PW * 4)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  if (n == 0) {
    if (blsgpu_join(c) != BLSGPU_OK) return BLSGPU_ERR_HIP;
    KLAUNCH(k_store_identity<F>, dim3(1), dim3(64), 0, st, c->result.as<u32>());
    LAUNCHCHK();
    KLAUNCH(k_proj_export<F>, dim3(1), dim3(256), 0, st, c->result.as<u32>(), (u32*)d_out_wire, (size_t)1);
    LAUNCHCHK();
    return BLSGPU_OK;
  }
  // resident window-shifted tables (blsgpu_bases_precompute): all windows share one bucket set
  const bool merged = bases->table != nullptr;
  // GLV (G1): 2n points (the bases and their images under the endomorphism) with balanced 127-bit scalars -> half the windows
  const bool glv = GroupTag<F>::id == 1 && !merged && bases->endo && !c->no_glv && !c->force_slow_sort && 2 * n <= ((size_t)1 << 24);
  // four-dimensional decomposition (G2): 4n points (every base with its images under psi, psi^2, psi^3) with 63-bit scalars -> a quarter of the windows
  const bool gls = GroupTag<F>::id == 2 && !merged && bases->endo && !c->no_glv && !c->force_slow_sort && 4 * n <= ((size_t)1 << 24);
  const size_t ns = glv ? 2 * n : gls ? 4 * n : n;      // scalars the sort sees
  const int sbits = glv ? 128 : gls ? 64 : 256;         // ... and their width (incl. the spare bit of the signed recoding)
  const int cw = merged ? bases->table_c : (c->msm_c ? c->msm_c : pick_window(ns, sbits));
  const int nwin = (sbits + cw - 1) / cw;               // digit windows per scalar
  const int nseg = merged ? 1 : nwin;                   // independent bucket sets
  const u32 nbw = 1u << (cw - 1);
  const size_t nb = (size_t)nseg * nbw;
  const size_t total = (size_t)nwin * ns;
  if (total > 0xfffffff0ull) return bad("msm: n * windows exceeds 2^32 entries");
  // the whole configuration is validated BEFORE a slot is taken or anything is enqueued
  const bool fast_sort = merged || ((ns <= ((size_t)1 << 24)) && cw <= 16 && cw >= 2 && !c->force_slow_sort);
  const int key_bits = cw - 1;
  const int coarse_bits = merged ? (key_bits > 7 ? key_bits - 7 : 0) : (key_bits < 8 ? key_bits : 8);
  const int fine_bits = key_bits - coarse_bits;           // <= 7
  const int ncoarse = 1 << coarse_bits;
  const int nc = nseg * ncoarse;
  if (fast_sort && nc > SORT_MAX_COUNTERS) return bad("msm: window configuration exceeds the sort's counter table");
  if (!fast_sort && nblk(nb, 1024) > 4096) return bad("msm: too many buckets for the fallback sort (use a window <= 16)");
  int bad_alloc = 0;
  blsgpu_ctx::Slot& sl = c->slot[c->next_slot];
  c->next_slot = (c->next_slot + 1) % NSLOT;
  // One call at a time (no pipelining): front, accumulation and tail run on the caller's stream -- every cross-stream
  // dependency costs a barrier packet and 20-90 us of idle time between the phases (kernel trace of a single call), and there is
  // nothing to overlap with.  Only the T tree sums keep their side stream.  Pipelined calls use the slot's own streams.
  const bool single = !c->pipelining;
  hipStream_t ft = single ? st : sl.front, tt = single ? st : sl.tail;
  // every buffer of this slot may still be in use by the call that used it last (NSLOT calls ago)
  if (sl.tail_pending) { HIPCHK(hipStreamWaitEvent(ft, sl.ev_tail, 0)); HIPCHK(hipStreamWaitEvent(st, sl.ev_tail, 0)); }
  bad_alloc |= sl.ent.reserve(total * 4);
  bad_alloc |= sl.sorted.reserve(total * 4);
  {
    size_t hb = (nb > 3 * (size_t)SORT_MAX_COUNTERS + 4 ? nb : 3 * (size_t)SORT_MAX_COUNTERS + 4) * 4;
    bool fresh = sl.hist.cap < hb;
    bad_alloc |= sl.hist.reserve(hb);
    if (fresh && !bad_alloc) HIPCHK(hipMemsetAsync(sl.hist.p, 0, sl.hist.cap, ft));      // the sort keeps its counters zeroed between calls
  }
  if (!fast_sort) bad_alloc |= sl.cursor.reserve(total * 4);      // per-entry rank inside its bucket (fallback sort only)
  const int sform = c->scalar_form;
  const bool plain_mont = sform == SCALAR_MONT && !glv && !gls;      // no decomposition kernel touches the scalars: reduce them first
  if (glv) bad_alloc |= sl.glv.reserve(ns * 16);
  if (gls) bad_alloc |= sl.glv.reserve(ns * 8);
  if (plain_mont) bad_alloc |= sl.glv.reserve(n * 32);
  bad_alloc |= sl.offs.reserve((nb + 1) * 4);
  bad_alloc |= sl.bsum.reserve(4096 * 4);
  // item cap: ~4x the mean bucket load, so that with uniform scalars (almost) no bucket is cut
  u32 cap = 128;
  while (cap < ITEM_CAP_MAX && (size_t)cap * nbw < 4 * ns) cap *= 2;
  if (c->item_cap) cap = c->item_cap;
  const size_t max_items = total / cap + nb + 1;                // every bucket has >= 1 item
  const size_t max_records = nb + max_items;                    // bucket sums + partial sums of heavy buckets
  {
    blsgpu_ctx::Slot* g = &sl;
    bad_alloc |= g->items.reserve(max_items * sizeof(ItemDesc));
    bad_alloc |= g->heavy.reserve(nb * sizeof(uint4));
    bad_alloc |= g->ctrl.reserve((4 + 2 * ITEM_BINS) * 4);
    // (re)allocation frees memory: make sure nothing of this slot is in flight
    size_t lvl = (nb / 2 + 1) * PW * 4;
    bool grow = g->buckets.cap < max_records * PW * 4 || g->lvlR[0].cap < lvl || g->lvlT.cap < 2 * lvl || g->wacc[0].cap < (size_t)nwin * 32 * PW * 4 || g->wsums.cap < (size_t)nwin * PW * 4;
    if (grow) { HIPCHK(hipStreamSynchronize(g->tail)); HIPCHK(hipStreamSynchronize(g->tail2)); HIPCHK(hipStreamSynchronize(st)); }
    bad_alloc |= g->buckets.reserve(max_records * PW * 4);
    bad_alloc |= g->lvlR[0].reserve(lvl); bad_alloc |= g->lvlR[1].reserve(lvl); bad_alloc |= g->lvlT.reserve(2 * lvl);     // every level's T records side by side
    bad_alloc |= g->tsum[0].reserve(lvl); bad_alloc |= g->tsum[1].reserve(lvl);
    bad_alloc |= g->wacc[0].reserve((size_t)nwin * 32 * PW * 4); bad_alloc |= g->wacc[1].reserve((size_t)nwin * 32 * PW * 4);
    bad_alloc |= g->wsums.reserve((size_t)nwin * PW * 4);
    bad_alloc |= g->result.reserve(PW * 4);
  }
  if (bad_alloc) { g_err = "hipMalloc(msm scratch) failed"; return BLSGPU_ERR_HIP; }
  const bool prof = c->profiling;
  auto mark = [&](int i) { if (prof) hipEventRecord(c->ev[i], ft); };
  // the front stream starts after whatever produced the scalars on the caller's stream
  if (ft != st) { HIPCHK(hipEventRecord(sl.ev_in, st)); HIPCHK(hipStreamWaitEvent(ft, sl.ev_in, 0)); }

  mark(0);
  if (plain_mont) {
    KLAUNCH(k_scalars_from_mont, dim3(nblk(n, 256)), dim3(256), 0, ft, (const u32*)d_scalars, sl.glv.as<u32>(), (int)n, c->status_word);
    LAUNCHCHK();
    d_scalars = sl.glv.p;
  }
  if (fast_sort) {
    // 1'-3'. two-level counting sort (LDS atomics; see msm.hip.h)
    // fixed layout: [MAX] counts (kept zero between calls) | [MAX+1] bases | [MAX] cursors
    u32* ghist = sl.hist.as<u32>();
    u32* gbase = ghist + SORT_MAX_COUNTERS;
    u32* gcur = gbase + SORT_MAX_COUNTERS + 1;
    if (sl.hist_dirty) { HIPCHK(hipMemsetAsync(ghist, 0, (size_t)SORT_MAX_COUNTERS * 4, ft)); sl.hist_dirty = false; }
    const unsigned tiles = nblk(ns, SORT_TILE);
    const u32* sort_in = (const u32*)d_scalars;
    if (glv) {
      KLAUNCH(k_glv_decompose, dim3(nblk(n, 256)), dim3(256), 0, ft, (const u32*)d_scalars, sl.glv.as<u32>(), (int)n, c->status_word, sform);
      sort_in = sl.glv.as<u32>();
      KLAUNCH(k_sort_hist<4>, dim3(tiles), dim3(SORT_THREADS), (size_t)nc * 4, ft, sort_in, ghist, (int)ns, cw, nwin, fine_bits, ncoarse, 0, c->status_word);
    } else if (gls) {
      KLAUNCH(k_gls_decompose, dim3(nblk(n, 256)), dim3(256), 0, ft, (const u32*)d_scalars, sl.glv.as<u32>(), (int)n, c->status_word, sform);
      sort_in = sl.glv.as<u32>();
      KLAUNCH(k_sort_hist<2>, dim3(tiles), dim3(SORT_THREADS), (size_t)nc * 4, ft, sort_in, ghist, (int)ns, cw, nwin, fine_bits, ncoarse, 0, c->status_word);
    } else {
      KLAUNCH(k_sort_hist<8>, dim3(tiles), dim3(SORT_THREADS), (size_t)nc * 4, ft, sort_in, ghist, (int)ns, cw, nwin, fine_bits, ncoarse, merged ? 1 : 0, c->status_word);
    }
    LAUNCHCHK();
    mark(1);
    KLAUNCH(k_sort_scan, dim3(1), dim3(1024), 0, ft, ghist, gbase, gcur, nc, sl.ctrl.as<u32>(), 4 + 2 * ITEM_BINS);
    LAUNCHCHK();
    mark(2);
    if (glv)
      KLAUNCH(k_sort_scatter<4>, dim3(tiles), dim3(SORT_THREADS), (size_t)nc * 8, ft, sort_in, gbase, gcur, sl.ent.as<u32>(), (int)ns, cw, nwin, fine_bits, ncoarse, 0, (u32)0);
    else if (gls)
      KLAUNCH(k_sort_scatter<2>, dim3(tiles), dim3(SORT_THREADS), (size_t)nc * 8, ft, sort_in, gbase, gcur, sl.ent.as<u32>(), (int)ns, cw, nwin, fine_bits, ncoarse, 0, (u32)0);
    else
      KLAUNCH(k_sort_scatter<8>, dim3(tiles), dim3(SORT_THREADS), (size_t)nc * 8, ft, sort_in, gbase, gcur, sl.ent.as<u32>(), (int)ns, cw, nwin,
                         fine_bits, ncoarse, merged ? 1 : 0, (u32)bases->n);
    KLAUNCH(k_sort_fine, dim3(nc), dim3(256), 0, ft, sl.ent.as<u32>(), gbase, sl.sorted.as<u32>(), sl.offs.as<u32>(), fine_bits, nc);
    LAUNCHCHK();
    mark(3);
  } else {
    // 1. digits + histogram
    sl.hist_dirty = true;
    HIPCHK(hipMemsetAsync(sl.hist.p, 0, nb * 4, ft));
    HIPCHK(hipMemsetAsync(sl.ctrl.p, 0, (4 + 2 * ITEM_BINS) * 4, ft));
    KLAUNCH(k_msm_digits, dim3(nblk(n, 256)), dim3(256), 0, ft, (const u32*)d_scalars, sl.ent.as<u32>(), sl.cursor.as<u32>(), sl.hist.as<u32>(), (int)n, cw, nwin, c->status_word);
    LAUNCHCHK();
    mark(1);
    // 2. scan
    unsigned sb = nblk(nb, 1024);
    KLAUNCH(k_scan_block_sums, dim3(sb), dim3(256), 0, ft, sl.hist.as<u32>(), sl.bsum.as<u32>(), (int)nb);
    KLAUNCH(k_scan_top, dim3(1), dim3(1024), 0, ft, sl.bsum.as<u32>(), (int)sb);
    KLAUNCH(k_scan_apply, dim3(sb), dim3(256), 0, ft, sl.hist.as<u32>(), sl.bsum.as<u32>(), sl.offs.as<u32>(), (int)nb);
    LAUNCHCHK();
    mark(2);
    // 3. scatter
    KLAUNCH(k_msm_scatter, dim3(nblk(total, 256)), dim3(256), 0, ft, sl.ent.as<u32>(), sl.cursor.as<u32>(), sl.offs.as<u32>(), sl.sorted.as<u32>(),
                       (int)n, total);
    LAUNCHCHK();
    mark(3);
  }
  // 4.-7. for the windows [g0, g0 + ng) (the whole call: cutting a single call into a high and a low window group whose tails overlap
  // was measured in round 4 and is slower, tools/experiments/msm_window_groups.patch): work items, accumulation, bucket reduction,
  // window combine.  gs = the slot whose buffers, side streams and events are used; gft / gas / gtt / gt2 = front, accumulation, tail
  // and tree streams
  auto run_group = [&](blsgpu_ctx::Slot& gs, int g0, int ng, hipStream_t gft, hipStream_t gas, hipStream_t gtt, hipStream_t gt2, bool last) -> int {
  const size_t gnb = (size_t)ng * nbw;
  // entries in THIS group's bucket sets / cap + one item per bucket: with resident tables (merged) the single bucket set holds the
  // entries of ALL nwin windows (the accumulation kernels run one lane per item without a grid stride, so the grid must cover them)
  const size_t gmax_items = (size_t)(merged ? nwin : ng) * ns / cap + gnb + 1;
  const u32* goffs = sl.offs.as<u32>() + (size_t)g0 * nbw;
  // 4. work items
  u32* ctrl = gs.ctrl.as<u32>();
  u32* bins = ctrl + 4;
  u32* bcur = ctrl + 4 + ITEM_BINS;
  {
    KLAUNCH(k_item_count, dim3(nblk(gnb, ITEM_BLOCK_BUCKETS)), dim3(256), 0, gft, goffs, bins, ctrl, (int)gnb, cap);
    KLAUNCH(k_item_scan, dim3(1), dim3(256), 0, gft, bins, ctrl, cap);
    KLAUNCH(k_item_fill, dim3(nblk(gnb, ITEM_BLOCK_BUCKETS)), dim3(256), 0, gft, goffs, bins, bcur, ctrl, gs.items.as<ItemDesc>(),
                       gs.heavy.as<uint4>(), (int)gnb, cap);
    LAUNCHCHK();
    if (last) mark(4);
    if (gas != gft) { HIPCHK(hipEventRecord(gs.ev_front, gft)); HIPCHK(hipStreamWaitEvent(gas, gs.ev_front, 0)); }
  }
  // the records (and images) may have been written on another stream than this call's (blsgpu_set_stream after the upload)
  if (!bases->ready_seen) {
    if (hipEventQuery(bases->ev_ready) == hipSuccess) bases->ready_seen = true;
    else HIPCHK(hipStreamWaitEvent(gas, bases->ev_ready, 0));
  }
  // 5. accumulate (grid covers the worst-case item count; surplus lanes exit on ctrl[2])
  // timing events are not free in a pipelined run (two records cost ~0.05-0.1 ms of queue time per MSM): sample every N-th launch
  const bool time_this = c->acc_timing && (c->acc_tick++ % (unsigned)c->acc_timing) == 0;
  if (time_this) { acc_harvest(c, false); if (gs.k_pending) { hipEventSynchronize(gs.ev_k1); acc_harvest(c, false); } hipEventRecord(gs.ev_k0, gas); }
  u32* records = gs.buckets.as<u32>();
  const u32* base_rec = (merged ? bases->table : bases->rec) + first * Store<F>::AFF_WORDS;
  if constexpr (GroupTag<F>::id == 2)
    KLAUNCH(k_msm_accumulate_g2pair, dim3(nblk(2 * gmax_items, BLS_G2ACC_BLOCK)), dim3(BLS_G2ACC_BLOCK), 0, gas, gls ? bases->endo + 4 * first * Store<F>::AFF_WORDS : base_rec,
                       sl.sorted.as<u32>(), gs.items.as<ItemDesc>(), ctrl, records);
  else
    KLAUNCH(k_msm_accumulate<F>, dim3(nblk(gmax_items, BLS_ACC_BLOCK)), dim3(BLS_ACC_BLOCK), 0, gas, base_rec, glv ? bases->endo + first * Store<F>::AFF_WORDS : (const u32*)nullptr,
                       glv ? (u32)n : 0xffffffffu, sl.sorted.as<u32>(), gs.items.as<ItemDesc>(), ctrl, records);
  if (time_this) { hipEventRecord(gs.ev_k1, gas); gs.k_pending = true; }
  // the fold of cut buckets (almost always a no-op) stays on the accumulation stream: as the first kernel of the tail it made
  // the next accumulation start ~90 us earlier, inside the previous call's bottom reduction level, and the pipelined rate FELL
  // by 2.6 % (A/B on one box, twice: 3.57 vs 3.66*10^8 scalar-muls/s)
  KLAUNCH(k_msm_heavy<F>, dim3(HEAVY_SMALL_BLOCKS + 512), dim3(256), 0, gas, gs.heavy.as<uint4>(), ctrl, records);
  LAUNCHCHK();
  if (prof) hipEventRecord(c->ev[5], gas);
  // ---- tail ------------------------------------------------------------------------------------------------------
  if (gtt != gas) { HIPCHK(hipEventRecord(gs.ev_acc, gas)); HIPCHK(hipStreamWaitEvent(gtt, gs.ev_acc, 0)); }
  // 6. per-window weighted sums:  wsum = sum_g T_g + M * wsum0(R)
  {
    const int nseg = ng;                                // (shadows the call's window count: everything below is per group)
    hipStream_t tt = gtt;
    blsgpu_ctx::Slot& sl = gs;
    std::vector<int> Ms;
    const u32* E = records;
    int nn = (int)nbw, off = 1, cur = 0, level = 0;
    // level T sums are stored consecutively in wacc[0]: level l at offset l * nseg
    u32* tstore = sl.wacc[0].template as<u32>();
    hipStream_t t2 = gt2;
    size_t toff = 0;                                    // offset (records) of this level's T block inside lvlT
    struct Tree { const u32* in; int n, pp, level; size_t off; };
    std::vector<Tree> trees;                            // T trees with passes left
    // one pass of every unfinished tree: a single multi-job launch when all of them fit the team form
    auto tree_step = [&]() -> int {
      TreeJobs J; J.njobs = 0; J.nseg = nseg; J.first_team[0] = 0; J.maxM = 0;
      for (auto& tr : trees) {
        if (tr.n <= 1) continue;
        int TM = tr.n >= 8 ? 8 : tr.n, TG = (tr.n + TM - 1) / TM;
        u32* o = TG == 1 ? tstore + (size_t)tr.level * nseg * PW : sl.tsum[tr.pp].template as<u32>() + tr.off * PW;   // the last pass lands in the Horner table
        if ((size_t)nseg * TG * TEAM <= TEAM_LANES_MAX && J.njobs < TREE_JOBS_MAX) {
          int j = J.njobs++;
          J.in[j] = tr.in; J.out[j] = o; J.n[j] = tr.n; J.M[j] = TM; J.G[j] = TG; J.first_team[j + 1] = J.first_team[j] + nseg * TG;
          if (TM > J.maxM) J.maxM = TM;
        } else {
          KLAUNCH(k_tree_sum<F>, dim3(nblk((size_t)nseg * TG, 256)), dim3(256), 0, t2, tr.in, o, nseg, tr.n, TM);
        }
        tr.in = o; tr.n = TG; tr.pp ^= 1;
      }
      if (J.njobs) KLAUNCH(k_tree_sum_team_multi<F>, dim3(nblk((size_t)J.first_team[J.njobs] * TEAM, 256)), dim3(256), TEAM_LDS(256), t2, J);
      LAUNCHCHK();
      return BLSGPU_OK;
    };
    while (nn > 1) {
      int M = nn >= 8 ? 8 : nn;
      int G = nn / M;
      u32* Rout = sl.lvlR[cur].template as<u32>();
      u32* Tout = sl.lvlT.template as<u32>() + toff * PW;
      // a level with a single group writes its T straight into the Horner table
      if (G == 1) Tout = tstore + (size_t)level * nseg * PW;
      // the two running sums of a chain (R and T) advance on two teams / two lanes, T one step behind R: M + 1 dependent
      // additions per level instead of 2 M
      if ((size_t)nseg * G * 2 * TEAM <= TEAM_LANES_MAX)
        KLAUNCH(k_wsum_level_team2<F>, dim3(nblk((size_t)nseg * G * 2 * TEAM, 256)), dim3(256),
                           TEAM_LDS(256) + (size_t)(256 / TEAM / 2) * 3 * TeamTraits<F>::WORDS * 4, tt, E, Rout, Tout, nseg, nn, M, off);
      else if ((size_t)nseg * G * TEAM <= TEAM_LANES_MAX)
        KLAUNCH(k_wsum_level_team<F>, dim3(nblk((size_t)nseg * G * TEAM, 256)), dim3(256), TEAM_LDS(256), tt, E, Rout, Tout, nseg, nn, M, off);
      else if constexpr (GroupTag<F>::id == 2)
        KLAUNCH(k_wsum_level_g2pair, dim3(nblk((size_t)nseg * G * 2, 256)), dim3(256), 0, tt, E, Rout, Tout, nseg, nn, M, off);
      else
        KLAUNCH(k_wsum_level_pair, dim3(nblk((size_t)nseg * G * 2, BLS_WSUM_BLOCK)), dim3(BLS_WSUM_BLOCK), 0, tt, E, Rout, Tout, nseg, nn, M, off);
      LAUNCHCHK();
      // sum the G T-records of each window down to one -- on the second tail stream: the next level needs only Rout.  Level l
      // needs log8(G_l) passes; after every level ONE launch carries the next pass of every tree that still has one (the T
      // trees of different levels are independent), so the trees finish while the R chain is still running.
      if (G > 1) {
        if (level < 8) {
          if (t2 != tt) { HIPCHK(hipEventRecord(sl.ev_lvl[level], tt)); HIPCHK(hipStreamWaitEvent(t2, sl.ev_lvl[level], 0)); }
          trees.push_back({Tout, G, 0, level, toff});
        } else {
          // (never reached with windows <= 16 bits: more than eight levels)  plain sequential tree on the tail stream
          const u32* Tin = Tout; int tn = G, tc = 0;
          while (tn > 1) {
            int TM = tn >= 8 ? 8 : tn; int TG = (tn + TM - 1) / TM;
            u32* o = TG == 1 ? tstore + (size_t)level * nseg * PW : sl.tsum[tc].template as<u32>() + toff * PW;
            KLAUNCH(k_tree_sum<F>, dim3(nblk((size_t)nseg * TG, 256)), dim3(256), 0, tt, Tin, o, nseg, tn, TM);
            LAUNCHCHK();
            Tin = o; tn = TG; tc ^= 1;
          }
        }
      }
      tree_step();
      toff += (size_t)nseg * G;
      Ms.push_back(M);
      E = Rout; nn = G; off = 0; cur ^= 1; level++;
      if (level >= 31) return bad("msm: reduction depth");
    }
    for (bool more = true; more;) {                    // passes that are left when the last level has run
      more = false;
      for (auto& tr : trees) more |= tr.n > 1;
      if (more) tree_step();
    }
    if (t2 != tt) { HIPCHK(hipEventRecord(sl.ev_tree, t2)); HIPCHK(hipStreamWaitEvent(tt, sl.ev_tree, 0)); }
    if (level == 0) {
      // a single bucket per window (c = 1): the bucket itself is the window sum
      HIPCHK(hipMemcpyAsync(sl.wsums.p, records, (size_t)nseg * PW * 4, hipMemcpyDeviceToDevice, tt));
    } else {
      // Horner over the levels: acc_L = T_L ; acc_l = T_l + M_l * acc_{l+1}
      u32* accbuf = sl.wacc[1].template as<u32>();
      HIPCHK(hipMemcpyAsync(accbuf, tstore + (size_t)(level - 1) * nseg * PW, (size_t)nseg * PW * 4, hipMemcpyDeviceToDevice, tt));
      for (int l = level - 2; l >= 0; l--) {
        int k = 0; while ((1 << k) < Ms[l]) k++;
        KLAUNCH(k_shift_add_team<F>, dim3(nblk((size_t)nseg * TEAM, 256)), dim3(256), TEAM_LDS(256), tt, accbuf, tstore + (size_t)l * nseg * PW, accbuf, nseg, k);
        LAUNCHCHK();
      }
      HIPCHK(hipMemcpyAsync(sl.wsums.p, accbuf, (size_t)nseg * PW * 4, hipMemcpyDeviceToDevice, tt));
    }
  }
  if (prof) hipEventRecord(c->ev[6], gtt);
  // 7. combine windows
  KLAUNCH(k_msm_combine_team<F>, dim3(1), dim3(TEAM), TEAM_LDS(TEAM), gtt, gs.wsums.as<u32>(), gs.result.as<u32>(), ng, cw);
  LAUNCHCHK();
  if (last) {
    KLAUNCH(k_proj_export<F>, dim3(1), dim3(256), 0, gtt, gs.result.as<u32>(), (u32*)d_out_wire, (size_t)1);
    LAUNCHCHK();
  }
  if (prof) hipEventRecord(c->ev[7], gtt);
  HIPCHK(hipEventRecord(gs.ev_tail, gtt));
  gs.tail_pending = true;
  gs.seq = c->msm_calls + 1;
  return BLSGPU_OK;
  };
  // pipelined calls accumulate on the library's own stream: front(i+1) must not queue behind accumulate(i)
  hipStream_t as = c->pipelining ? c->acc_stream : st;
  {
    int rc = run_group(sl, 0, nseg, ft, as, tt, sl.tail2, true);
    if (rc) return rc;
  }
  ++c->msm_calls;
  if (!c->pipelining && tt != st) HIPCHK(hipStreamWaitEvent(st, sl.ev_tail, 0));    // in-order semantics on the caller's stream
  if (prof) {
    HIPCHK(hipStreamSynchronize(tt));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 7; i++) hipEventElapsedTime(&c->phase_ms[i], c->ev[i], c->ev[i + 1]);
    hipEventElapsedTime(&c->phase_ms[7], c->ev[0], c->ev[7]);
  }
  return BLSGPU_OK;
}

template <class F>
static int msm_host(blsgpu_ctx* c, const blsgpu_bases* bases, size_t first, const uint8_t* scalars, size_t n, uint64_t* out) {
  if (!c || !out || (n && !scalars)) return bad("msm: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_b.reserve(n ? n * 32 : 16) || c->io_out.reserve(3 * Wire<F>::WORDS * 4)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  SyncStatus ss(c);
  int rc = ss.begin();
  if (rc) return rc;
  if (n) { int ru = staged_upload(c, c->io_b.p, scalars, n * 32); if (ru) return ru; }
  rc = msm_device<F>(c, bases, first, c->io_b.p, n, c->io_out.p);
  if (rc) return rc;
  rc = blsgpu_join(c);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, 3 * Wire<F>::WORDS * 4, hipMemcpyDeviceToHost, c->stream));
  rc = ss.fetch();
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return ss.verdict();
}
extern "C" int blsgpu_g1_msm(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const uint8_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c); return msm_host<FpPolicy>(c, b, first, s, n, out); }
extern "C" int blsgpu_g2_msm(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const uint8_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c); return msm_host<Fp2Policy>(c, b, first, s, n, out); }
extern "C" int blsgpu_g1_msm_device(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const void* s, size_t n, void* out) { CTX_CLAIM(c); return msm_device<FpPolicy>(c, b, first, s, n, out); }
extern "C" int blsgpu_g2_msm_device(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const void* s, size_t n, void* out) { CTX_CLAIM(c); return msm_device<Fp2Policy>(c, b, first, s, n, out); }
// the same four with the scalars as `&[Scalar]` memory holds them: four u64 Montgomery limbs each (scalar.rs:23-27); `Scalar::to_bytes`
// (:284-296) runs on the device, fused into the decomposition kernels
extern "C" int blsgpu_g1_msm_mont(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const uint64_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return msm_host<FpPolicy>(c, b, first, (const uint8_t*)s, n, out); }
extern "C" int blsgpu_g2_msm_mont(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const uint64_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return msm_host<Fp2Policy>(c, b, first, (const uint8_t*)s, n, out); }
extern "C" int blsgpu_g1_msm_mont_device(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const void* s, size_t n, void* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return msm_device<FpPolicy>(c, b, first, s, n, out); }
extern "C" int blsgpu_g2_msm_mont_device(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const void* s, size_t n, void* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return msm_device<Fp2Policy>(c, b, first, s, n, out); }
// k MSMs over the SAME resident bases (e.g. commitments to k polynomials under one SRS): scalars of call j at
// d_scalars + j * n * 32, result j at d_out + j * 3 * WORDS * 4.  The calls go through the pipeline slots, so the sort,
// accumulation and tail of consecutive MSMs overlap; results are ordered on the context's stream on return.
template <class F>
static int msm_many_device(blsgpu_ctx* c, const blsgpu_bases* bases, size_t first, const void* d_scalars, size_t n, size_t k, void* d_out) {
  if (!c || !bases || (k && (!d_out || (n && !d_scalars)))) return bad("msm_many: NULL argument");
  const bool was = c->pipelining;
  c->pipelining = true;
  int rc = BLSGPU_OK;
  for (size_t j = 0; j < k && rc == BLSGPU_OK; j++)
    rc = msm_device<F>(c, bases, first, (const uint8_t*)d_scalars + j * n * 32, n, (uint8_t*)d_out + j * 3 * Wire<F>::WORDS * 4);
  c->pipelining = was;
  int rj = blsgpu_join(c);
  return rc ? rc : rj;
}
extern "C" int blsgpu_g1_msm_many_device(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const void* s, size_t n, size_t k, void* out) { CTX_CLAIM(c); return msm_many_device<FpPolicy>(c, b, first, s, n, k, out); }
extern "C" int blsgpu_g2_msm_many_device(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const void* s, size_t n, size_t k, void* out) { CTX_CLAIM(c); return msm_many_device<Fp2Policy>(c, b, first, s, n, k, out); }
template <class F>
static int msm_many_host(blsgpu_ctx* c, const blsgpu_bases* bases, size_t first, const uint8_t* scalars, size_t n, size_t k, uint64_t* out) {
  if (!c || (k && (!out || (n && !scalars)))) return bad("msm_many: NULL argument");
  if (!k) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t ob = 3 * Wire<F>::WORDS * 4;
  if (c->io_b.reserve(n * k ? n * k * 32 : 16) || c->io_out.reserve(k * ob)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  SyncStatus ss(c);
  int rc = ss.begin();
  if (rc) return rc;
  if (n) { int ru = staged_upload(c, c->io_b.p, scalars, n * k * 32); if (ru) return ru; }
  rc = msm_many_device<F>(c, bases, first, c->io_b.p, n, k, c->io_out.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, k * ob, hipMemcpyDeviceToHost, c->stream));
  rc = ss.fetch();
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return ss.verdict();
}
extern "C" int blsgpu_g1_msm_many(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const uint8_t* s, size_t n, size_t k, uint64_t* out) { CTX_CLAIM(c); return msm_many_host<FpPolicy>(c, b, first, s, n, k, out); }
extern "C" int blsgpu_g2_msm_many(blsgpu_ctx* c, const blsgpu_bases* b, size_t first, const uint8_t* s, size_t n, size_t k, uint64_t* out) { CTX_CLAIM(c); return msm_many_host<Fp2Policy>(c, b, first, s, n, k, out); }
// Repeated one-shot MSMs over the SAME base array (a drop-in caller that passes its SRS slice on every call: the reference's surface
// has no place for a resident handle).  Opt-in (blsgpu_set_bases_cache): a base array is recognised by its length and a fingerprint of
// 64 evenly spaced points -- the caller promises not to change an array it passes again.  First sight: the one-shot path as always.
// Second sight: the set is uploaded as RESIDENT bases (subgroup test, endomorphism images) and kept; from then on a call only moves its
// scalars, i.e. it runs on the headline path (bases resident, 32 B per scalar over PCIe).
// every word of the array, four host threads (blsgpu_set_bases_cache_verify): ~100 MB at memory speed for 2^20 G1 points
static uint64_t bases_full_hash(const uint64_t* xy, const uint8_t* inf, size_t n, size_t words) {
  constexpr int T = 4;
  uint64_t part[T];
  auto run = [&](int t) {
    const size_t lo = n * (size_t)t / T, hi = n * (size_t)(t + 1) / T;
    uint64_t a = 0x9e3779b97f4a7c15ull ^ (uint64_t)t, b = 0xc2b2ae3d27d4eb4full;
    for (size_t i = lo * words; i < hi * words; i++) { a = (a ^ xy[i]) * 0xff51afd7ed558ccdull; a ^= a >> 29; b += a; }
    if (inf) for (size_t i = lo; i < hi; i++) { b = (b ^ inf[i]) * 0x100000001b3ull; }
    part[t] = a ^ (b * 0x9e3779b97f4a7c15ull);
  };
  std::vector<std::thread> th;
  try { for (int t = 1; t < T; t++) th.emplace_back(run, t); } catch (...) { for (auto& x : th) x.join(); th.clear(); for (int t = 1; t < T; t++) run(t); }
  run(0);
  for (auto& x : th) x.join();
  uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
  for (int t = 0; t < T; t++) { h ^= part[t]; h *= 1099511628211ull; }
  return h;
}
static uint64_t bases_fingerprint(const uint64_t* xy, const uint8_t* inf, size_t n, size_t words) {
  uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
  const size_t step = n > 64 ? n / 64 : 1;
  for (size_t i = 0; i < n; i += step) {
    for (size_t k = 0; k < words; k++) { h ^= xy[i * words + k]; h *= 1099511628211ull; }
    h ^= inf ? inf[i] : 0; h *= 1099511628211ull;
  }
  for (size_t k = 0; k < words && n; k++) { h ^= xy[(n - 1) * words + k]; h *= 1099511628211ull; }
  return h;
}
template <class F>
static int msm_oneshot(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint8_t* s, size_t n, uint64_t* out) {
  if (c && c->bcache_cap > 0 && n >= 1024 && xy) {
    constexpr size_t W = 2 * Wire<F>::WORDS / 2;            // u64 per affine point
    const uint64_t fp = c->bcache_verify ? bases_full_hash(xy, inf, n, W) : bases_fingerprint(xy, inf, n, W);
    blsgpu_ctx::BasesCacheEntry* hit = nullptr;
    for (auto& e : c->bcache) if (e.group == GroupTag<F>::id && e.n == n && e.fp == fp) hit = &e;
    if (hit) {
      hit->last = ++c->bcache_tick;
      if (!hit->b) {                                          // second sight: make it resident
        int rc = bases_upload<F>(c, xy, inf, n, &hit->b, false);
        if (rc) { hit->b = nullptr; return rc; }
      }
      return msm_host<F>(c, hit->b, 0, s, n, out);
    }
    if ((int)c->bcache.size() >= c->bcache_cap) {             // evict the least recently used entry
      size_t lru = 0;
      for (size_t k = 1; k < c->bcache.size(); k++) if (c->bcache[k].last < c->bcache[lru].last) lru = k;
      if (c->bcache[lru].b) blsgpu_bases_free(c->bcache[lru].b);
      c->bcache.erase(c->bcache.begin() + (long)lru);
    }
    c->bcache.push_back({GroupTag<F>::id, n, fp, nullptr, ++c->bcache_tick});
  }
  blsgpu_bases* b = nullptr;
  int rc = bases_upload<F>(c, xy, inf, n, &b, true);
  if (rc) return rc;
  rc = msm_host<F>(c, b, 0, s, n, out);
  blsgpu_bases_free(b);
  return rc;
}
extern "C" int blsgpu_set_bases_cache_verify(blsgpu_ctx* c, int on) { CTX_CLAIM(c);
  if (!c) return bad("ctx is NULL");
  if ((on != 0) != c->bcache_verify) {                 // fingerprints of the two kinds do not compare: start over
    for (auto& e : c->bcache) if (e.b) blsgpu_bases_free(e.b);
    c->bcache.clear();
  }
  c->bcache_verify = on != 0;
  return BLSGPU_OK;
}
extern "C" int blsgpu_set_bases_cache(blsgpu_ctx* c, int entries) { CTX_CLAIM(c);
  if (!c || entries < 0 || entries > 8) return bad("set_bases_cache: entries must be in [0, 8]");
  HIPCHK(hipSetDevice(c->device));
  c->bcache_cap = entries;
  while ((int)c->bcache.size() > entries) { if (c->bcache.back().b) blsgpu_bases_free(c->bcache.back().b); c->bcache.pop_back(); }
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_msm_host(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint8_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c); return msm_oneshot<FpPolicy>(c, xy, inf, s, n, out); }
extern "C" int blsgpu_g2_msm_host(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint8_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c); return msm_oneshot<Fp2Policy>(c, xy, inf, s, n, out); }

// ---------------------------------------------------------------------------------------------------
// batched variable-base scalar multiplication (mulbatch.hip.h): out[i] = [s_i] P_i, N in -> N out
// ---------------------------------------------------------------------------------------------------
template <class F>
static int mul_batch_device(blsgpu_ctx* c, const void* d_xy, const void* d_inf, const void* d_scalars, size_t n, void* d_out) {
  if (!c || (n && (!d_xy || !d_scalars || !d_out))) return bad("mul_batch: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if constexpr (MbIO<F>::LANES == 1) {
    // G1 points the caller vouches for (blsgpu_set_assume_subgroup): the endomorphism split halves the doublings, as in the MSM
    if (c->assume_subgroup && !c->no_glv) {
      KLAUNCH(k_mul_batch_glv, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (const u32*)d_xy, (const uint8_t*)d_inf, (const u32*)d_scalars, (u32*)d_out, n, c->status_word, c->scalar_form);
      LAUNCHCHK();
      return BLSGPU_OK;
    }
  }
  if constexpr (MbIO<F>::LANES == 2) {
    if (c->assume_subgroup && !c->no_glv) {         // G2 points the caller vouches for: the four-dimensional psi split
      KLAUNCH(k_mul_batch_gls, dim3(nblk(n * 2, 256)), dim3(256), 0, c->stream, (const u32*)d_xy, (const uint8_t*)d_inf, (const u32*)d_scalars, (u32*)d_out, n, c->status_word, c->scalar_form);
      LAUNCHCHK();
      return BLSGPU_OK;
    }
  }
  KLAUNCH(k_mul_batch<F>, dim3(nblk(n * MbIO<F>::LANES, 256)), dim3(256), 0, c->stream, (const u32*)d_xy, (const uint8_t*)d_inf, (const u32*)d_scalars,
                     (u32*)d_out, n, c->status_word, c->scalar_form);
  LAUNCHCHK();
  return BLSGPU_OK;
}
template <class F>
static int mul_batch_host(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint8_t* scalars, size_t n, uint64_t* out) {
  if (!c || (n && (!xy || !scalars || !out))) return bad("mul_batch: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  constexpr size_t WB = MbIO<F>::WW * 4;
  if (c->io_a.reserve(n * 2 * WB) || c->io_b.reserve(n * 32) || c->flags_a.reserve(n) || c->io_out.reserve(n * 3 * WB)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  SyncStatus ss(c);
  int rc = ss.begin();
  if (rc) return rc;
  { int ru = staged_upload(c, c->io_a.p, xy, n * 2 * WB); if (!ru) ru = staged_upload(c, c->io_b.p, scalars, n * 32); if (ru) return ru; }
  if (inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, inf, n, hipMemcpyHostToDevice, c->stream));
  rc = mul_batch_device<F>(c, c->io_a.p, inf ? c->flags_a.p : nullptr, c->io_b.p, n, c->io_out.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 3 * WB, hipMemcpyDeviceToHost, c->stream));
  rc = ss.fetch();
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return ss.verdict();
}
extern "C" int blsgpu_g1_mul_batch(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint8_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c); return mul_batch_host<FpPolicy>(c, xy, inf, s, n, out); }
extern "C" int blsgpu_g2_mul_batch(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint8_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c); return mul_batch_host<Fp2PairPolicy>(c, xy, inf, s, n, out); }
extern "C" int blsgpu_g1_mul_batch_device(blsgpu_ctx* c, const void* xy, const void* inf, const void* s, size_t n, void* out) { CTX_CLAIM(c); return mul_batch_device<FpPolicy>(c, xy, inf, s, n, out); }
extern "C" int blsgpu_g2_mul_batch_device(blsgpu_ctx* c, const void* xy, const void* inf, const void* s, size_t n, void* out) { CTX_CLAIM(c); return mul_batch_device<Fp2PairPolicy>(c, xy, inf, s, n, out); }
// `&G1Affine * &Scalar` over slices with the scalars as Montgomery limbs (g1.rs:556-594 calls `Scalar::to_bytes` per product)
extern "C" int blsgpu_g1_mul_batch_mont(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint64_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return mul_batch_host<FpPolicy>(c, xy, inf, (const uint8_t*)s, n, out); }
extern "C" int blsgpu_g2_mul_batch_mont(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, const uint64_t* s, size_t n, uint64_t* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return mul_batch_host<Fp2PairPolicy>(c, xy, inf, (const uint8_t*)s, n, out); }
extern "C" int blsgpu_g1_mul_batch_mont_device(blsgpu_ctx* c, const void* xy, const void* inf, const void* s, size_t n, void* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return mul_batch_device<FpPolicy>(c, xy, inf, s, n, out); }
extern "C" int blsgpu_g2_mul_batch_mont_device(blsgpu_ctx* c, const void* xy, const void* inf, const void* s, size_t n, void* out) { CTX_CLAIM(c);
  ScalarFormScope f(c, SCALAR_MONT); return mul_batch_device<Fp2PairPolicy>(c, xy, inf, s, n, out); }

// ---------------------------------------------------------------------------------------------------
// group helpers
// ---------------------------------------------------------------------------------------------------
template <class F>
static int proj_sum(blsgpu_ctx* c, const uint64_t* xyz, size_t n, uint64_t* out) {
  if (!c || !out || (n && !xyz)) return bad("sum: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  constexpr int WW = Wire<F>::WORDS, PW = Store<F>::PROJ_WORDS;
  if (c->io_a.reserve(n ? n * 3 * WW * 4 : 16) || c->io_c.reserve((n ? n : 1) * PW * 4) || c->result.reserve(PW * 4) || c->io_out.reserve(3 * WW * 4)) {
    g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP;
  }
  if (n) {
    HIPCHK(hipMemcpyAsync(c->io_a.p, xyz, n * 3 * WW * 4, hipMemcpyHostToDevice, c->stream));
    KLAUNCH(k_proj_import<F>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->io_a.as<u32>(), c->io_c.as<u32>(), n);
  }
  KLAUNCH(k_proj_sum_team<F>, dim3(1), dim3(TEAM), TEAM_LDS(TEAM), c->stream, c->io_c.as<u32>(), c->result.as<u32>(), n);
  KLAUNCH(k_proj_export<F>, dim3(1), dim3(256), 0, c->stream, c->result.as<u32>(), c->io_out.as<u32>(), (size_t)1);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, 3 * WW * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// device-pointer variant: wire-format partials in device memory -> wire-format sum in device memory, asynchronous
template <class F>
static int proj_sum_device(blsgpu_ctx* c, const void* d_xyz, size_t n, void* d_out) {
  if (!c || !d_out || (n && !d_xyz)) return bad("sum_device: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  constexpr int PW = Store<F>::PROJ_WORDS;
  // (a fold queued on the fold stream must not share scratch with calls on the main stream: nothing orders the two)
  DevBuf& recs = c->on_fold_stream ? c->fold_c : c->io_c;
  DevBuf& res = c->on_fold_stream ? c->fold_result : c->result;
  if (recs.reserve((n ? n : 1) * PW * 4) || res.reserve(PW * 4)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (n) KLAUNCH(k_proj_import<F>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (const u32*)d_xyz, recs.as<u32>(), n);
  KLAUNCH(k_proj_sum_team<F>, dim3(1), dim3(TEAM), TEAM_LDS(TEAM), c->stream, recs.as<u32>(), res.as<u32>(), n);
  KLAUNCH(k_proj_export<F>, dim3(1), dim3(256), 0, c->stream, res.as<u32>(), (u32*)d_out, (size_t)1);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_sum_device(blsgpu_ctx* c, const void* xyz, size_t n, void* out) { CTX_CLAIM(c); return proj_sum_device<FpPolicy>(c, xyz, n, out); }
extern "C" int blsgpu_g2_sum_device(blsgpu_ctx* c, const void* xyz, size_t n, void* out) { CTX_CLAIM(c); return proj_sum_device<Fp2Policy>(c, xyz, n, out); }
extern "C" int blsgpu_g1_sum(blsgpu_ctx* c, const uint64_t* xyz, size_t n, uint64_t* out) { CTX_CLAIM(c); return proj_sum<FpPolicy>(c, xyz, n, out); }
extern "C" int blsgpu_g2_sum(blsgpu_ctx* c, const uint64_t* xyz, size_t n, uint64_t* out) { CTX_CLAIM(c); return proj_sum<Fp2Policy>(c, xyz, n, out); }

// device core: projective wire records in device memory -> affine wire coordinates + infinity bytes in device memory (asynchronous)
template <class F>
static int batch_normalize_device(blsgpu_ctx* c, const void* d_xyz, size_t n, void* d_xy, void* d_inf) {
  if (!c || (n && (!d_xyz || !d_xy || !d_inf))) return bad("batch_normalize: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  constexpr int PW = Store<F>::PROJ_WORDS;
  if (c->io_c.reserve(n * PW * 4) || c->io_d.reserve(n * Store<F>::EL * 4)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  KLAUNCH(k_proj_import<F>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (const u32*)d_xyz, c->io_c.as<u32>(), n);
  if (n >= 4096) {
    const int K = normalize_k(n);
    size_t T = (n + K - 1) / K;
    KLAUNCH(k_batch_normalize<F>, dim3(nblk(T, 256)), dim3(256), 0, c->stream, c->io_c.as<u32>(), c->io_d.as<u32>(), (u32*)d_xy, (uint8_t*)d_inf, n, T, K);
  } else {
    KLAUNCH(k_proj_to_affine<F>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->io_c.as<u32>(), (u32*)d_xy, (uint8_t*)d_inf, n);
  }
  LAUNCHCHK();
  return BLSGPU_OK;
}
template <class F>
static int batch_normalize(blsgpu_ctx* c, const uint64_t* xyz, size_t n, uint64_t* xy, uint8_t* inf) {
  if (!c || (n && (!xyz || !xy))) return bad("batch_normalize: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  constexpr int WW = Wire<F>::WORDS;
  if (c->io_a.reserve(n * 3 * WW * 4) || c->io_out.reserve(n * 2 * WW * 4) || c->flags_b.reserve(n)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, xyz, n * 3 * WW * 4, hipMemcpyHostToDevice, c->stream));
  if (int rc = batch_normalize_device<F>(c, c->io_a.p, n, c->io_out.p, c->flags_b.p)) return rc;
  HIPCHK(hipMemcpyAsync(xy, c->io_out.p, n * 2 * WW * 4, hipMemcpyDeviceToHost, c->stream));
  if (inf) HIPCHK(hipMemcpyAsync(inf, c->flags_b.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_batch_normalize_device(blsgpu_ctx* c, const void* xyz, size_t n, void* xy, void* inf) { CTX_CLAIM(c); return batch_normalize_device<FpPolicy>(c, xyz, n, xy, inf); }
extern "C" int blsgpu_g2_batch_normalize_device(blsgpu_ctx* c, const void* xyz, size_t n, void* xy, void* inf) { CTX_CLAIM(c); return batch_normalize_device<Fp2Policy>(c, xyz, n, xy, inf); }
extern "C" int blsgpu_g1_batch_normalize(blsgpu_ctx* c, const uint64_t* xyz, size_t n, uint64_t* xy, uint8_t* inf) { CTX_CLAIM(c); return batch_normalize<FpPolicy>(c, xyz, n, xy, inf); }
extern "C" int blsgpu_g2_batch_normalize(blsgpu_ctx* c, const uint64_t* xyz, size_t n, uint64_t* xy, uint8_t* inf) { CTX_CLAIM(c); return batch_normalize<Fp2Policy>(c, xyz, n, xy, inf); }

// ---------------------------------------------------------------------------------------------------
// self-test hooks
// ---------------------------------------------------------------------------------------------------
static int elem_op(blsgpu_ctx* c, int words, int kind, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  if (!c || (n && (!a || !out))) return bad("op: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  size_t bytes = n * words * 4;
  if (c->io_a.reserve(bytes) || c->io_b.reserve(bytes) || c->io_out.reserve(bytes)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, a, bytes, hipMemcpyHostToDevice, c->stream));
  if (b) HIPCHK(hipMemcpyAsync(c->io_b.p, b, bytes, hipMemcpyHostToDevice, c->stream));
  const u32* bp = b ? c->io_b.as<u32>() : nullptr;
  if (kind == 1) KLAUNCH(k_fp_op, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, c->io_a.as<u32>(), bp, c->io_out.as<u32>(), n);
  else if (kind == 2) KLAUNCH(k_fp2_op, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, c->io_a.as<u32>(), bp, c->io_out.as<u32>(), n);
  else if (kind == 6) KLAUNCH(k_fp6_op, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, op, c->io_a.as<u32>(), bp, c->io_out.as<u32>(), n);
  else if (c->pairing_layout != 2 && op != 3) KLAUNCH(k_fp12_op_quad, dim3(nblk(n * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, op, c->io_a.as<u32>(), bp, c->io_out.as<u32>(), n);
  else KLAUNCH(k_fp12_op, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, op, c->io_a.as<u32>(), bp, c->io_out.as<u32>(), n);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_fp_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (op < 0 || op > 6) return bad("fp_op: unknown op");
  return elem_op(c, 12, 1, op, a, b, n, out);
}
extern "C" int blsgpu_fp2_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (op < 0 || op > 6) return bad("fp2_op: unknown op");
  return elem_op(c, 24, 2, op, a, b, n, out);
}
extern "C" int blsgpu_fp6_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!(op == 0 || op == 3 || op == 4 || op == 5 || op == 7 || op == 11 || op == 12)) return bad("fp6_op: unknown op");
  if ((op == 0 || op == 11 || op == 12) && n && !b) return bad("fp6_op: the second operand is missing");
  return elem_op(c, 72, 6, op, a, b, n, out);
}
extern "C" int blsgpu_fp12_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!(op == 0 || op == 3 || op == 4 || op == 7 || op == 8 || op == 9 || op == 10)) return bad("fp12_op: unknown op");
  if (op == 10 && c && c->pairing_layout == 2) return bad("fp12_op: op 10 (cyclotomic exponentiation) exists in the quad layout only");
  return elem_op(c, 144, 12, op, a, b, n, out);
}
template <class F>
static int point_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, const uint8_t* binf, size_t n, uint64_t* out) {
  constexpr int WW = Wire<F>::WORDS;
  size_t ab = n * 3 * WW * 4, bb = n * (op == 2 ? 2 : 3) * WW * 4;
  if (c->io_a.reserve(ab) || c->io_b.reserve(bb) || c->io_out.reserve(ab) || c->flags_a.reserve(n)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, a, ab, hipMemcpyHostToDevice, c->stream));
  if (op != 1) HIPCHK(hipMemcpyAsync(c->io_b.p, b, bb, hipMemcpyHostToDevice, c->stream));
  if (op == 2 && binf) HIPCHK(hipMemcpyAsync(c->flags_a.p, binf, n, hipMemcpyHostToDevice, c->stream));
  KLAUNCH(k_point_op<F>, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, c->io_a.as<u32>(), c->io_b.as<u32>(),
                     (op == 2 && binf) ? c->flags_a.as<uint8_t>() : nullptr, c->io_out.as<u32>(), n);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, ab, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_point_op(blsgpu_ctx* c, int group, int op, const uint64_t* a, const uint64_t* b, const uint8_t* binf, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!c || (n && (!a || !out || (op != 1 && !b))) || op < 0 || op > 2 || (group != 1 && group != 2)) return bad("point_op: bad argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  return group == 1 ? point_op<FpPolicy>(c, op, a, b, binf, n, out) : point_op<Fp2Policy>(c, op, a, b, binf, n, out);
}

static int chain_probe(blsgpu_ctx* c, int iters, double* rate, bool fp) {
  if (!c || !rate || iters <= 0) return bad("throughput: bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, c->device));
  int blocks = prop.multiProcessorCount * 8;   // 8 blocks x 4 waves = 8 waves per SIMD
  if (c->io_a.reserve(256 * 28 * 4 + 4096) || c->io_out.reserve((size_t)blocks * 256 * NL * 4)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemsetAsync(c->io_a.p, 0x11, 256 * 28 * 4 + 4096, c->stream));
  auto launch = [&](int it) {
    if (fp) KLAUNCH(k_fp_mul_chain, dim3(blocks), dim3(256), 0, c->stream, c->io_out.as<u32>(), c->io_a.as<u32>(), it);
    else KLAUNCH(k_mad_chain, dim3(blocks), dim3(256), 0, c->stream, c->io_out.as<u32>(), c->io_a.as<u32>(), it);
  };
  launch(4);
  HIPCHK(hipEventRecord(c->ev[0], c->stream));
  launch(iters);
  HIPCHK(hipEventRecord(c->ev[1], c->stream));
  HIPCHK(hipEventSynchronize(c->ev[1]));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  double ops = (double)blocks * 256.0 * iters * (fp ? 1.0 : 64.0);
  *rate = ops / (ms * 1e-3);
  return BLSGPU_OK;
}
extern "C" int blsgpu_fp_mul_throughput(blsgpu_ctx* c, int iters, double* r) { CTX_CLAIM(c); return chain_probe(c, iters, r, true); }
extern "C" int blsgpu_mad_throughput(blsgpu_ctx* c, int iters, double* r) { CTX_CLAIM(c); return chain_probe(c, iters, r, false); }

// ---------------------------------------------------------------------------------------------------
// pairings
// MSM on the reference's public encodings (for a wrapper crate that cannot see limbs): bases as uncompressed
// bytes (`to_uncompressed`, decoded like `from_uncompressed_unchecked`), scalars as `Scalar::to_bytes`, result as the
// uncompressed bytes of the affine sum.  A composition of the entry points above.
template <int G>
static int msm_bytes(blsgpu_ctx* c, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out) {
  constexpr int W = G == 1 ? 12 : 24, BYTES = G == 1 ? 96 : 192;
  if (!c || !out || (n && (!bases || !scalars))) return bad("msm_bytes: NULL argument");
  ScalarFormScope bytes_form(c, SCALAR_BYTES);        // this entry point's scalars ARE `Scalar::to_bytes()` output, whatever the context's setting
  std::vector<uint64_t> xy(n * W + 1), xyz(3 * W), axy(2 * W);
  std::vector<uint8_t> inf(n + 1), ok(n + 1);
  uint8_t ainf = 0;
  int rc = BLSGPU_OK;
  if (n) {
    rc = G == 1 ? blsgpu_g1_from_bytes_batch(c, bases, n, 0, 0, xy.data(), inf.data(), ok.data()) : blsgpu_g2_from_bytes_batch(c, bases, n, 0, 0, xy.data(), inf.data(), ok.data());
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) if (!ok[i]) return bad("msm_bytes: a base is not a valid uncompressed encoding");
  }
  rc = G == 1 ? blsgpu_g1_msm_host(c, xy.data(), inf.data(), scalars, n, xyz.data()) : blsgpu_g2_msm_host(c, xy.data(), inf.data(), scalars, n, xyz.data());
  if (rc) return rc;
  rc = G == 1 ? blsgpu_g1_batch_normalize(c, xyz.data(), 1, axy.data(), &ainf) : blsgpu_g2_batch_normalize(c, xyz.data(), 1, axy.data(), &ainf);
  if (rc) return rc;
  (void)BYTES;
  return G == 1 ? blsgpu_g1_to_bytes_batch(c, axy.data(), &ainf, 1, 0, out) : blsgpu_g2_to_bytes_batch(c, axy.data(), &ainf, 1, 0, out);
}
extern "C" int blsgpu_g1_msm_bytes(blsgpu_ctx* c, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[96]) { CTX_CLAIM(c); return msm_bytes<1>(c, bases, scalars, n, out); }
extern "C" int blsgpu_g2_msm_bytes(blsgpu_ctx* c, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[192]) { CTX_CLAIM(c); return msm_bytes<2>(c, bases, scalars, n, out); }

// ---------------------------------------------------------------------------------------------------
// hash-to-curve (h2c.hip.h)
// ---------------------------------------------------------------------------------------------------
// one launch of the batched hash: group 1 = one lane per message, group 2 = one lane pair; batches that leave the chip under-filled take
// the split form (two lane groups per message, h2c.hip.h) -- up to 2^15 messages to G1 (<= 1 024 wavefronts of 64 lanes at one per SIMD),
// up to 2^14 to G2 (4 lanes each: 1 024 wavefronts).  Measured on MI355X, 2^14 32-byte messages: see DESIGN.md 4.8.
static void h2c_launch(blsgpu_ctx* c, int group, const uint8_t* msgs, const unsigned long long* offs, size_t n, const uint8_t* dst, u32 dlen, int encode_only, u32* out) {
  const int forced = c->h2c_split;
  const bool split = !encode_only && (forced >= 0 ? forced == 1 : n <= (group == 1 ? (size_t)1 << 15 : (size_t)1 << 14));
  if (group == 1) {
    if (split) KLAUNCH(k_hash_to_curve_split<FpPolicy>, dim3(nblk(n * 2, 64)), dim3(64), 0, c->stream, msgs, offs, n, dst, dlen, out);
    else KLAUNCH(k_hash_to_curve<FpPolicy>, dim3(nblk(n, 64)), dim3(64), 0, c->stream, msgs, offs, n, dst, dlen, encode_only ? 1 : 0, out);
  } else {
    if (split) KLAUNCH(k_hash_to_curve_split<Fp2PairPolicy>, dim3(nblk(n * 4, 256)), dim3(256), 0, c->stream, msgs, offs, n, dst, dlen, out);
    else KLAUNCH(k_hash_to_curve<Fp2PairPolicy>, dim3(nblk(n * 2, 256)), dim3(256), 0, c->stream, msgs, offs, n, dst, dlen, encode_only ? 1 : 0, out);
  }
}
template <class F>
static int h2c_host(blsgpu_ctx* c, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len, int encode_only,
                    uint64_t* out) {
  if (!c || (n && (!offsets || !out)) || (dst_len && !dst)) return bad("hash_to_curve: NULL argument");
  if (!n) return BLSGPU_OK;
  const size_t total = (size_t)offsets[n];
  for (size_t i = 0; i < n; i++) if (offsets[i] > offsets[i + 1]) return bad("hash_to_curve: offsets must be non-decreasing");
  if (total && !msgs) return bad("hash_to_curve: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  // a DST longer than 255 bytes is replaced by H("H2C-OVERSIZE-DST-" || DST)  (expand_msg.rs:74-95)
  uint8_t d[255]; u32 dlen;
  if (dst_len > 255) {
    Sha256 s; sha_init(s);
    const char* salt = "H2C-OVERSIZE-DST-";
    for (int i = 0; salt[i]; i++) sha_put(s, (uint8_t)salt[i]);
    for (size_t i = 0; i < dst_len; i++) sha_put(s, dst[i]);
    u32 hw[8]; sha_finish(s, hw);
    for (int i = 0; i < 32; i++) d[i] = (uint8_t)(hw[i >> 2] >> (24 - 8 * (i & 3)));
    dlen = 32;
  } else {
    for (size_t i = 0; i < dst_len; i++) d[i] = dst[i];
    dlen = (u32)dst_len;
  }
  constexpr int WW = Wire<F>::WORDS;
  if (c->io_a.reserve(total + 16) || c->io_b.reserve((n + 1) * 8) || c->io_c.reserve(256) || c->io_out.reserve(n * 3 * WW * 4)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (total) HIPCHK(hipMemcpyAsync(c->io_a.p, msgs, total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->io_b.p, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (dlen) HIPCHK(hipMemcpyAsync(c->io_c.p, d, dlen, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));                 // `d` lives on this stack frame
  h2c_launch(c, GroupTag<F>::id, c->io_a.as<uint8_t>(), (const unsigned long long*)c->io_b.p, n, c->io_c.as<uint8_t>(), dlen, encode_only, c->io_out.as<u32>());
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 3 * WW * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_hash_to_curve_batch(blsgpu_ctx* c, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                             int encode_only, uint64_t* out_xyz) { CTX_CLAIM(c);
  return h2c_host<FpPolicy>(c, msgs, offsets, n, dst, dst_len, encode_only, out_xyz);
}
extern "C" int blsgpu_g2_hash_to_curve_batch(blsgpu_ctx* c, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                             int encode_only, uint64_t* out_xyz) { CTX_CLAIM(c);
  return h2c_host<Fp2Policy>(c, msgs, offsets, n, dst, dst_len, encode_only, out_xyz);
}
// device-resident variant: messages, offsets (n + 1 u64) and the DST (<= 255 bytes) already in device memory
extern "C" int blsgpu_hash_to_curve_device(blsgpu_ctx* c, int group, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len,
                                           int encode_only, void* d_out_xyz) { CTX_CLAIM(c);
  if (!c || (n && (!d_offsets || !d_out_xyz)) || (dst_len && !d_dst)) return bad("hash_to_curve: NULL argument");
  if (dst_len > 255) return bad("hash_to_curve_device: reduce a DST longer than 255 bytes on the host first");
  if (group != 1 && group != 2) return bad("hash_to_curve: group must be 1 or 2");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  h2c_launch(c, group, (const uint8_t*)d_msgs, (const unsigned long long*)d_offsets, n, (const uint8_t*)d_dst, (u32)dst_len, encode_only, (u32*)d_out_xyz);
  LAUNCHCHK();
  return BLSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// scalar field Fr: element-wise vector operations and the radix-2 transform (fr.hip.h)
// ---------------------------------------------------------------------------------------------------
extern "C" int blsgpu_fr_op_device(blsgpu_ctx* c, int op, const void* a, const void* b, size_t n, void* out, void* nonzero_flags) { CTX_CLAIM(c);
  if (!c || (n && (!a || !out))) return bad("fr_op: NULL argument");
  if (op < 0 || op > 6) return bad("fr_op: unknown op");
  if (op <= 2 && n && !b) return bad("fr_op: binary op needs b");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  KLAUNCH(k_fr_op, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, (const u32*)a, op <= 2 ? (const u32*)b : (const u32*)nullptr, (u32*)out,
                     (uint8_t*)nonzero_flags, n);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_fr_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, uint8_t* nonzero_flags) { CTX_CLAIM(c);
  if (!c || (n && (!a || !out))) return bad("fr_op: NULL argument");
  if (op < 0 || op > 6) return bad("fr_op: unknown op");
  if (op <= 2 && n && !b) return bad("fr_op: binary op needs b");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n * 32) || c->io_b.reserve(n * 32) || c->io_out.reserve(n * 32) || c->flags_a.reserve(n)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, a, n * 32, hipMemcpyHostToDevice, c->stream));
  if (op <= 2) HIPCHK(hipMemcpyAsync(c->io_b.p, b, n * 32, hipMemcpyHostToDevice, c->stream));
  int rc = blsgpu_fr_op_device(c, op, c->io_a.p, c->io_b.p, n, c->io_out.p, (op == 4 && nonzero_flags) ? c->flags_a.p : nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 32, hipMemcpyDeviceToHost, c->stream));
  if (op == 4 && nonzero_flags) HIPCHK(hipMemcpyAsync(nonzero_flags, c->flags_a.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// `Scalar::to_bytes` / `from_bytes` / `from_bytes_wide` over vectors (scalar.rs:284-296, :256-280, :300-331; k_fr_convert)
static int fr_convert_device(blsgpu_ctx* c, int op, const void* in, size_t n, void* out, void* ok) {
  if (!c || (n && (!in || !out))) return bad("fr conversion: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  KLAUNCH(k_fr_convert, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, (const u32*)in, (u32*)out, (uint8_t*)ok, n);
  LAUNCHCHK();
  return BLSGPU_OK;
}
static int fr_convert_host(blsgpu_ctx* c, int op, const void* in, size_t n, void* out, uint8_t* ok) {
  if (!c || (n && (!in || !out))) return bad("fr conversion: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t ib = n * (op == 2 ? 64 : 32);
  if (c->io_a.reserve(ib) || c->io_out.reserve(n * 32) || c->flags_a.reserve(n)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  { int ru = staged_upload(c, c->io_a.p, in, ib); if (ru) return ru; }
  int rc = fr_convert_device(c, op, c->io_a.p, n, c->io_out.p, ok ? c->flags_a.p : nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 32, hipMemcpyDeviceToHost, c->stream));
  if (ok) HIPCHK(hipMemcpyAsync(ok, c->flags_a.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_fr_to_bytes_device(blsgpu_ctx* c, const void* scalars, size_t n, void* bytes, void* ok) { CTX_CLAIM(c); return fr_convert_device(c, 0, scalars, n, bytes, ok); }
extern "C" int blsgpu_fr_from_bytes_device(blsgpu_ctx* c, const void* bytes, size_t n, void* scalars, void* ok) { CTX_CLAIM(c); return fr_convert_device(c, 1, bytes, n, scalars, ok); }
extern "C" int blsgpu_fr_from_bytes_wide_device(blsgpu_ctx* c, const void* bytes, size_t n, void* scalars) { CTX_CLAIM(c); return fr_convert_device(c, 2, bytes, n, scalars, nullptr); }
extern "C" int blsgpu_fr_to_bytes(blsgpu_ctx* c, const uint64_t* scalars, size_t n, uint8_t* bytes, uint8_t* ok) { CTX_CLAIM(c); return fr_convert_host(c, 0, scalars, n, bytes, ok); }
extern "C" int blsgpu_fr_from_bytes(blsgpu_ctx* c, const uint8_t* bytes, size_t n, uint64_t* scalars, uint8_t* ok) { CTX_CLAIM(c); return fr_convert_host(c, 1, bytes, n, scalars, ok); }
extern "C" int blsgpu_fr_from_bytes_wide(blsgpu_ctx* c, const uint8_t* bytes, size_t n, uint64_t* scalars) { CTX_CLAIM(c); return fr_convert_host(c, 2, bytes, n, scalars, nullptr); }
// in-place transform of 2^log_n scalars in device memory (natural order in and out)
extern "C" int blsgpu_fr_ntt_device(blsgpu_ctx* c, void* d_data, int log_n, int inverse) { CTX_CLAIM(c);
  if (!c || !d_data) return bad("fr_ntt: NULL argument");
  if (log_n < 0 || log_n > 28) return bad("fr_ntt: log_n must be in [0, 28]");
  if (log_n == 0) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const int dir = inverse ? 1 : 0;
  const size_t n = (size_t)1 << log_n, half = n >> 1;
  if (c->fr_tw[dir].reserve(n * 32) || c->fr_tmp.reserve(n * 32) || c->fr_ninv.reserve(64)) { g_err = "hipMalloc(fr scratch) failed"; return BLSGPU_ERR_HIP; }
  if (c->fr_tw_log[dir] != log_n) {
    KLAUNCH(k_fr_twiddles, dim3(nblk((half + FR_TW_RUN - 1) / FR_TW_RUN, 256)), dim3(256), 0, st, c->fr_tw[dir].as<u32>(), log_n, dir);
    if (log_n > 1) KLAUNCH(k_fr_tw_levels, dim3(nblk(half, 256)), dim3(256), 0, st, c->fr_tw[dir].as<u32>(), log_n);
    LAUNCHCHK();
    c->fr_tw_log[dir] = log_n;
    HIPCHK(hipEventRecord(c->ev_fr[dir], st));
  }
  HIPCHK(hipStreamWaitEvent(st, c->ev_fr[dir], 0));
  u32* data = (u32*)d_data;
  u32* tmp = c->fr_tmp.as<u32>();
  const u32* tw = c->fr_tw[dir].as<u32>();
  const int tl = log_n < FR_TILE_LOG ? log_n : FR_TILE_LOG;
  int lh = log_n - 1;                                   // log2 of the current half-span
  // The tile kernel permutes, so it cannot run in place.  With global passes the first one moves the data to the
  // scratch buffer (the rest run there in place) and the tile kernel brings the result home; a transform that fits
  // one tile goes through the scratch buffer and is copied back.
  const u32* src = data;
  u32* cur = lh >= tl ? tmp : data;
  // round 5: the top log_n - tl stages on column tiles in LDS (k_fr_cols), at most ten stages per pass over the data; needs 144 KB of
  // dynamic LDS per workgroup (gfx950 has 160 KB per CU) -- the stage-pair passes below remain for a device that refuses it and as the
  // A/B twin (BLSGPU_NTT_IMPL=stage)
  if (c->fr_cols_ok < 0) {
    int lds_max = 0;
    const size_t want = ((size_t)9 << FR_COLS_LOG) * 4;
    c->fr_cols_ok = 0;
    if (c->fr_cols_want && hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) == hipSuccess && (size_t)lds_max >= want &&
        hipFuncSetAttribute((const void*)k_fr_cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess)
      c->fr_cols_ok = 1;
    (void)hipGetLastError();
  }
  // Measured on MI355X (tools/ntt_time.py): tiles of 2^11 elements, at most seven stages per pass, 512 lanes per workgroup (two workgroups
  // per CU overlap their load / barrier / store phases): 2.52-2.55 ms at 2^24 and 11.2 ms at 2^26 against 2.85-2.91 / 12.2-12.4 ms for the
  // stage-pair passes (-11 % / -10 %); at 2^20 and 2^22 the two are equal within the run-to-run spread (0.155-0.17 / 0.60-0.66 ms): the
  // vector sits in the 256 MB Infinity Cache, a stage-pair pass is 19 us, and every variant costs 7-9 us per stage -- the butterflies'
  // ~375 instructions per multiplication, not the passes over the data, are what the transform pays for.  Below 2^20 the stage-pair
  // passes stay.  BLSGPU_NTT_COLS="tile log2,stages per pass,lanes" overrides the shape for experiments.
  if (c->fr_cols_ok && lh >= tl && (log_n >= 20 || c->fr_cols_want == 2)) {
    int tlog = 11, dmax = 7, block = 512;
    if (const char* v = getenv("BLSGPU_NTT_COLS")) { int a = 0, b = 0, cc = 0; if (sscanf(v, "%d,%d,%d", &a, &b, &cc) == 3 && a >= 6 && a <= FR_COLS_LOG && b >= 1 && b <= a && cc >= 64 && cc <= 1024) { tlog = a; dmax = b; block = cc; } }
    const int m = lh + 1 - tl, passes = (m + dmax - 1) / dmax;
    for (int ps = 0; ps < passes; ps++) {
      const int d = (lh + 1 - tl + (passes - ps) - 1) / (passes - ps);      // the remaining stages split evenly over the remaining passes
      const int ls = lh - d + 1;
      const int lk = tlog - d < ls ? tlog - d : ls;
      KLAUNCH(k_fr_cols, dim3((unsigned)(n >> (d + lk))), dim3(block), ((size_t)9 << (d + lk)) * 4, st, src, cur, tw, lh, d, lk);
      src = cur; lh -= d;
    }
  }
  while (lh - 1 >= tl) {                                // two stages per pass over the data
    KLAUNCH(k_fr_stage2, dim3(nblk(n / 4, 256)), dim3(256), 0, st, src, cur, tw, log_n, lh);
    src = cur; lh -= 2;
  }
  if (lh >= tl) { KLAUNCH(k_fr_stage1, dim3(nblk(n / 2, 256)), dim3(256), 0, st, src, cur, tw, log_n, lh); src = cur; lh--; }
  LAUNCHCHK();
  const u32* scale = nullptr;
  if (inverse) {
    if (c->fr_ninv_log != log_n) {
      KLAUNCH(k_fr_ninv, dim3(1), dim3(64), 0, st, c->fr_ninv.as<u32>(), log_n); c->fr_ninv_log = log_n;
      HIPCHK(hipEventRecord(c->ev_fr[2], st));
    }
    HIPCHK(hipStreamWaitEvent(st, c->ev_fr[2], 0));
    scale = c->fr_ninv.as<u32>();
  }
  u32* dst = src == data ? tmp : data;
  KLAUNCH(k_fr_tile, dim3((unsigned)(n >> tl)), dim3(256), ((size_t)9 << tl) * 4, st, src, dst, tw, log_n, tl, scale);
  LAUNCHCHK();
  if (dst != data) HIPCHK(hipMemcpyAsync(data, tmp, n * 32, hipMemcpyDeviceToDevice, st));
  return BLSGPU_OK;
}
extern "C" int blsgpu_fr_ntt(blsgpu_ctx* c, uint64_t* data, int log_n, int inverse) { CTX_CLAIM(c);
  if (!c || !data) return bad("fr_ntt: NULL argument");
  if (log_n < 0 || log_n > 28) return bad("fr_ntt: log_n must be in [0, 28]");
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)1 << log_n;
  if (c->io_a.reserve(n * 32)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, data, n * 32, hipMemcpyHostToDevice, c->stream));
  int rc = blsgpu_fr_ntt_device(c, c->io_a.p, log_n, inverse);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(data, c->io_a.p, n * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// ---- the wide (one workgroup per pairing) path -------------------------------------------------------------------
// Two built-in configurations (wide.hip.h): up to WIDE_ONE_PER_CU items one 1024-lane workgroup per CU (256 items at the latency of
// one, 1.1 ms); above that 512-lane workgroups, two per CU (512 items 1.6 ms, 1024 items 3.1 ms, 1536 items 4.7 ms), against the
// quad kernels' flat ~6.1 ms
constexpr size_t WIDE_ONE_PER_CU = 256;
constexpr size_t WIDE_AUTO_MAX = 1536;
static int wide_unavailable(blsgpu_ctx* c, const std::string& why) {
  c->wide_why = why;
  // with the default `auto` layout the quad kernels take small batches too (correct, but ~6 ms instead of ~1.1 ms for one pairing):
  // say so once per process instead of degrading in silence
  static std::atomic<bool> told{false};
  if (c->pairing_layout == 0 && !told.exchange(true))
    fprintf(stderr, "libblsgpu: the wide (small-batch) pairing path is unavailable -- %s; batches of <= %zu pairings run on the quad kernels\n", why.c_str(), WIDE_AUTO_MAX);
  return -1;
}
static int wide_load(blsgpu_ctx* c) {
  if (c->wide_state) return c->wide_state;
  c->wide_state = -1;
  std::string path;
  if (const char* e = getenv("BLSGPU_WIDE_PROG")) path = e;
  else {
    Dl_info info;
    if (!dladdr((const void*)&blsgpu_create, &info) || !info.dli_fname) return wide_unavailable(c, "the library's own path is unknown (static link?): set BLSGPU_WIDE_PROG to wide_prog.bin");
    path = info.dli_fname;
    const size_t slash = path.find_last_of('/');
    path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/wide_prog.bin";
  }
  FILE* fh = fopen(path.c_str(), "rb");
  if (!fh) return wide_unavailable(c, path + " cannot be opened (generate it: tools/gen_wide_prog.py, or __graft_entry__.build())");
  std::vector<u32> w;
  u32 buf[4096]; size_t got;
  while ((got = fread(buf, 4, 4096, fh)) > 0) w.insert(w.end(), buf, buf + got);
  fclose(fh);
  if (w.size() < 32 || w[0] != WIDE_BLOB_MAGIC || w[1] != 4) return wide_unavailable(c, path + " is not a wide program file (truncated or foreign)");
  if (w[15] != WIDE_FORMAT_VERSION) return wide_unavailable(c, path + " has another format version than this library (stale file: regenerate it)");
  static const u32 cfg[4][2] = {{1024, 4}, {1024, 4}, {512, 8}, {512, 8}};
  for (int k = 0; k < 4; k++) {
    const size_t off = w[2 + 2 * k], len = w[3 + 2 * k];
    // a program is only usable by the kernel it was generated for: same lanes per workgroup and limbs per product lane, slots and accumulators within the LDS arrays
    if (off + len > w.size() || len < 16 || (off & 3) || w[off] != WIDE_PROG_MAGIC || w[off + 2] > (u32)WIDE_MAX_SLOTS || w[off + 10] != cfg[k][0] ||
        w[off + 11] != cfg[k][1] || w[off + 12] >= (u32)WIDE_MAX_SLOTS || w[off + 13] > (u32)WIDE_MAX_ACC || (w[off + 7] & 3) || (w[off + 8] & 3) || (w[off + 9] & 1))
      return wide_unavailable(c, path + " was generated for another kernel configuration");
    c->wide_off[k] = off;
  }
  if (hipMalloc((void**)&c->d_wide, w.size() * 4) != hipSuccess) { (void)hipGetLastError(); c->d_wide = nullptr; return wide_unavailable(c, "hipMalloc for the wide programs failed"); }
  if (hipMemcpy(c->d_wide, w.data(), w.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); hipFree(c->d_wide); c->d_wide = nullptr; return wide_unavailable(c, "hipMemcpy of the wide programs failed"); }
  c->wide_state = 1;
  return 1;
}
static int wide_missing(blsgpu_ctx* c) {
  g_err = "pairing: BLSGPU_PAIRING_LAYOUT=wide but the wide programs are unavailable: " + c->wide_why;
  return BLSGPU_ERR_ARG;
}
// which kernels take a batch of n pairings / Miller loops / final exponentiations: 256 = wide, 4 = quad, 2 = lane pair
static int pairing_layout_for(blsgpu_ctx* c, size_t n) {
  if (c->pairing_layout == 2 || c->pairing_layout == 4) return c->pairing_layout;
  if (c->pairing_layout == 256) return wide_load(c) == 1 ? 256 : -1;          // asked for by name: no silent substitute
  return (n <= WIDE_AUTO_MAX && wide_load(c) == 1) ? 256 : 4;
}
extern "C" int blsgpu_pairing_layout(blsgpu_ctx* c, size_t n) { CTX_CLAIM(c);
  if (!c) return bad("pairing_layout: NULL context");
  if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); return bad("pairing_layout: hipSetDevice failed"); }
  const int l = pairing_layout_for(c, n);
  return l < 0 ? wide_missing(c) : l;
}
// "" when the wide programs are loaded, otherwise the reason they are not (also tried now if no pairing call has tried yet)
extern "C" const char* blsgpu_wide_status(blsgpu_ctx* c) {
  if (!c) return "NULL context";
  CtxClaim claim_(&c->owner_thread, &c->owner_depth);
  if (claim_.clash) return "the context is in use by another host thread";
  if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); return "hipSetDevice failed"; }
  return wide_load(c) == 1 ? "" : c->wide_why.c_str();
}
static void wide_launch(blsgpu_ctx* c, int mode, const void* g1, const void* g1inf, const void* g2, const void* g2inf, size_t n, void* out) {
  if (n <= WIDE_ONE_PER_CU)
    KLAUNCH((k_pairing_wide_t<1024, 4>), dim3((unsigned)n), dim3(1024), 0, c->stream, mode, (const u32*)g1, (const uint8_t*)g1inf, (const u32*)g2, (const uint8_t*)g2inf,
                       (u32*)out, n, c->d_wide + c->wide_off[0], c->d_wide + c->wide_off[1]);
  else
    KLAUNCH((k_pairing_wide_t<512, 8>), dim3((unsigned)n), dim3(512), 0, c->stream, mode, (const u32*)g1, (const uint8_t*)g1inf, (const u32*)g2, (const uint8_t*)g2inf,
                       (u32*)out, n, c->d_wide + c->wide_off[2], c->d_wide + c->wide_off[3]);
}
static int pairing_launch(blsgpu_ctx* c, int mode, const void* g1, const void* g1inf, const void* g2, const void* g2inf, size_t n, void* out) {
  // mode 0: full pairing, 1: Miller loop only
  const int layout = pairing_layout_for(c, n);
  if (layout < 0) return wide_missing(c);
  if (layout == 256) { wide_launch(c, mode, g1, g1inf, g2, g2inf, n, out); LAUNCHCHK(); return BLSGPU_OK; }
  if (layout == 4) {
    KLAUNCH(k_pairing_quad, dim3(nblk(n * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, mode, (const u32*)g1, (const uint8_t*)g1inf, (const u32*)g2,
                       (const uint8_t*)g2inf, (u32*)out, n);
    LAUNCHCHK();
    return BLSGPU_OK;
  }
  KLAUNCH(k_pairing, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, mode, (const u32*)g1, (const uint8_t*)g1inf, (const u32*)g2,
                     (const uint8_t*)g2inf, (u32*)out, n);
  LAUNCHCHK();
  return BLSGPU_OK;
}
static int pairing_host(blsgpu_ctx* c, int mode, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) {
  if (!c || (n && (!g1 || !g2 || !out))) return bad("pairing: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n * 96) || c->io_b.reserve(n * 192) || c->flags_a.reserve(n) || c->flags_b.reserve(n) || c->io_out.reserve(n * 576)) {
    g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP;
  }
  HIPCHK(hipMemcpyAsync(c->io_a.p, g1, n * 96, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->io_b.p, g2, n * 192, hipMemcpyHostToDevice, c->stream));
  if (g1inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, g1inf, n, hipMemcpyHostToDevice, c->stream));
  if (g2inf) HIPCHK(hipMemcpyAsync(c->flags_b.p, g2inf, n, hipMemcpyHostToDevice, c->stream));
  int rc = pairing_launch(c, mode, c->io_a.p, g1inf ? c->flags_a.p : nullptr, c->io_b.p, g2inf ? c->flags_b.p : nullptr, n, c->io_out.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_pairing_batch(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) { CTX_CLAIM(c);
  return pairing_host(c, 0, g1, g1inf, g2, g2inf, n, out);
}
extern "C" int blsgpu_miller_loop_batch(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) { CTX_CLAIM(c);
  return pairing_host(c, 1, g1, g1inf, g2, g2inf, n, out);
}
extern "C" int blsgpu_pairing_batch_device(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || (n && (!g1 || !g2 || !out))) return bad("pairing: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  return pairing_launch(c, 0, g1, g1inf, g2, g2inf, n, out);
}

extern "C" int blsgpu_miller_loop_batch_device(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || (n && (!g1 || !g2 || !out))) return bad("miller_loop: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  return pairing_launch(c, 1, g1, g1inf, g2, g2inf, n, out);
}
extern "C" int blsgpu_final_exponentiation_device(blsgpu_ctx* c, const void* in, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || (n && (!in || !out))) return bad("final_exponentiation: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  const int layout = pairing_layout_for(c, n);
  if (layout < 0) return wide_missing(c);
  if (layout == 256) wide_launch(c, 2, in, nullptr, nullptr, nullptr, n, out);
  else if (layout == 4) KLAUNCH(k_final_exp_quad, dim3(nblk(n * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, (const u32*)in, (u32*)out, n);
  else KLAUNCH(k_final_exp, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, (const u32*)in, (u32*)out, n);
  LAUNCHCHK();
  return BLSGPU_OK;
}

// product of n Fp12 wire values already in device memory (d_in) -> one wire value (d_out); tree of k_fp12_prod
static int fp12_product_device(blsgpu_ctx* c, const u32* d_in, size_t n, u32* d_out) {
  DevBuf& lvl_a = c->on_fold_stream ? c->fold_c : c->io_c;        // (see proj_sum_device)
  DevBuf& lvl_b = c->on_fold_stream ? c->fold_d : c->io_d;
  if (lvl_a.reserve((n / 2 + 1) * 576) || lvl_b.reserve((n / 4 + 1) * 576)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  if (n == 0) {
    KLAUNCH(k_fp12_one, dim3(1), dim3(64), 0, c->stream, d_out);
    LAUNCHCHK();
    return BLSGPU_OK;
  }
  const u32* in = d_in; int flip = 0;
  while (n > 1) {
    // a level that does not fill the chip (fewer than 2^16 products in flight) is pure latency: one Fp12 multiplication of a lone
    // lane pair is ~50 us, so such levels halve (fan 2: 50 us per level) instead of folding eight values in sequence (350 us);
    // 2^16 values: 2.27 -> ~0.9 ms, 8 values: 0.38 -> 0.15 ms (tools/experiments/prod_time.py)
    int fan = 2;
    while (fan < FP12_PROD_FAN && (n + fan - 1) / fan > 65536) fan *= 2;
    size_t m = (n + fan - 1) / fan;
    u32* o = (m == 1) ? d_out : (flip ? lvl_b.as<u32>() : lvl_a.as<u32>());
    if (m <= 32768 && c->pairing_layout != 2)            // latency-bound level: a quad per product
      KLAUNCH(k_fp12_prod_quad, dim3(nblk(m * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, in, o, n, m, fan);
    else
      KLAUNCH(k_fp12_prod, dim3(nblk(m * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, in, o, n, m, fan);
    LAUNCHCHK();
    in = o; n = m; flip ^= 1;
  }
  if (in != d_out) HIPCHK(hipMemcpyAsync(d_out, in, 576, hipMemcpyDeviceToDevice, c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_fp12_product_device(blsgpu_ctx* c, const void* in, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || !out || (n && !in)) return bad("fp12_product: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  return fp12_product_device(c, (const u32*)in, n, (u32*)out);
}
// the kernel behind the K > 1 case below: 1 = k_multi_miller_shared (measured fastest for unprepared terms: tools/mml_time.py), 4 = k_mml_prep_quad (prep.hip.h)
constexpr int MML_IMPL_DEFAULT = 1;
static int mmlp_launch(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, const void* qidx, const blsgpu_g2_prepared* p, const void* d_off,
                       size_t nseg, size_t total, int kuni, int kmax, void* out);
extern "C" int blsgpu_multi_miller_loop_device(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || !out || (n && (!g1 || !g2))) return bad("multi_miller_loop: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_out.reserve((n ? n : 1) * 576)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  // terms per accumulator: as many as still leave two wavefronts per SIMD (2^17 lanes) busy
  int K = 1;
  while (K < MML_MAX_K && n / (2 * (size_t)K) >= 65536) K *= 2;
  if (K == 1) {
    if (n) { int rc = pairing_launch(c, 1, g1, g1inf, g2, g2inf, n, c->io_out.p); if (rc) return rc; }
    return fp12_product_device(c, c->io_out.as<u32>(), n, (u32*)out);
  }
  if (c->mmlp_k > 0) K = c->mmlp_k;
  const size_t groups = (n + K - 1) / K;
  const int impl = c->mml_impl ? c->mml_impl : MML_IMPL_DEFAULT;
  if (impl == 1) {
    KLAUNCH(k_multi_miller_shared, dim3(nblk(groups * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, (const u32*)g1, (const uint8_t*)g1inf,
                       (const u32*)g2, (const uint8_t*)g2inf, c->io_out.as<u32>(), n, K);
    LAUNCHCHK();
  } else {
    int rc = mmlp_launch(c, g1, g1inf, g2, g2inf, nullptr, nullptr, nullptr, groups, n, K, K, c->io_out.p);
    if (rc) return rc;
  }
  return fp12_product_device(c, c->io_out.as<u32>(), groups, (u32*)out);
}
extern "C" int blsgpu_multi_miller_loop(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!c || !out || (n && (!g1 || !g2))) return bad("multi_miller_loop: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n ? n * 96 : 16) || c->io_b.reserve(n ? n * 192 : 16) || c->flags_a.reserve(n ? n : 16) || c->flags_b.reserve(n ? n : 16) || c->result.reserve(576)) {
    g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP;
  }
  if (n) {
    HIPCHK(hipMemcpyAsync(c->io_a.p, g1, n * 96, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->io_b.p, g2, n * 192, hipMemcpyHostToDevice, c->stream));
    if (g1inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, g1inf, n, hipMemcpyHostToDevice, c->stream));
    if (g2inf) HIPCHK(hipMemcpyAsync(c->flags_b.p, g2inf, n, hipMemcpyHostToDevice, c->stream));
  }
  int rc = blsgpu_multi_miller_loop_device(c, c->io_a.p, g1inf ? c->flags_a.p : nullptr, c->io_b.p, g2inf ? c->flags_b.p : nullptr, n, c->result.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->result.p, 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// ---- N independent multi_miller_loops in one call (bulk signature verification: N equations of k pairings each) -------------------
// Segment s = terms [off[s], off[s + 1]).  Miller values per term on the throughput kernels (or the wide path when there are few),
// one segmented Fp12 product, one batched final exponentiation.  The product of independently squared per-term values is the
// reference's shared-accumulator value exactly (Fp12 is a field: same element, canonical limbs).
constexpr size_t MML_SEG_SHARED_MIN = 49152;      // segments from which blsgpu_multi_miller_loop_many shares squarings inside a segment
static int final_exp_launch(blsgpu_ctx* c, const void* in, size_t n, void* out) {
  const int layout = pairing_layout_for(c, n);
  if (layout < 0) return wide_missing(c);
  if (layout == 256) wide_launch(c, 2, in, nullptr, nullptr, nullptr, n, out);
  else if (layout == 4) KLAUNCH(k_final_exp_quad, dim3(nblk(n * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, (const u32*)in, (u32*)out, n);
  else KLAUNCH(k_final_exp, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, (const u32*)in, (u32*)out, n);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_multi_miller_loop_many_device(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, const void* d_offsets, size_t nseg,
                                                    size_t total, size_t max_seg_terms, int final_exp, void* out) { CTX_CLAIM(c);
  if (!c || (nseg && (!d_offsets || !out)) || (total && (!g1 || !g2))) return bad("multi_miller_loop_many: NULL argument");
  if (!nseg) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  // runs per segment of the segmented product: 1 when the caller bounds the segments by 32 terms; otherwise (bound unknown or larger)
  // sized by the MEAN segment length -- ~8 values per run, at most 32 runs -- so that the partial products (576 B per run) stay
  // proportional to the input whatever the number of segments (2^20 three-term segments with an unknown bound: 1 run each, not 32);
  // a single long segment among many short ones is then walked by few quads: slower for that segment, never a failed allocation
  int parts = 1;
  if (max_seg_terms == 0 || max_seg_terms > 32) {
    const size_t mean = (total + nseg - 1) / nseg;
    parts = (int)((mean + 7) / 8);
    if (parts < 1) parts = 1;
    if (parts > 32) parts = 32;
  }
  // the shared-accumulator kernel is the lane-pair layout's: a context pinned to the quad (or wide) layout keeps the per-term path
  const bool seg_shared = total && max_seg_terms >= 2 && max_seg_terms <= (size_t)MML_MAX_K && nseg >= MML_SEG_SHARED_MIN && (c->pairing_layout == 0 || c->pairing_layout == 2);
  if ((!seg_shared && c->io_out.reserve((total ? total : 1) * 576)) || (!seg_shared && parts > 1 && c->io_c.reserve(nseg * parts * 576)) || (final_exp && c->io_d.reserve(nseg * 576))) {
    g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP;
  }
  u32* prod = final_exp ? c->io_d.as<u32>() : (u32*)out;
  // MANY short segments: one lane pair per segment with a shared accumulator (the reference's own schedule: (k - 1) / k of the 62
  // squarings per term disappear); it needs >= 2^16 lanes' worth of segments to beat the per-term quads (a quarter-filled chip runs
  // at the latency of one shared loop: ~13 ms for k = 3)
  if (seg_shared) {
    KLAUNCH(k_multi_miller_seg, dim3(nblk(nseg * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, (const u32*)g1, (const uint8_t*)g1inf, (const u32*)g2,
                       (const uint8_t*)g2inf, (const unsigned long long*)d_offsets, nseg, total, prod, c->d_status);
    LAUNCHCHK();
    return final_exp ? final_exp_launch(c, prod, nseg, out) : BLSGPU_OK;
  }
  if (total) { int rc = pairing_launch(c, 1, g1, g1inf, g2, g2inf, total, c->io_out.p); if (rc) return rc; }
  KLAUNCH(k_fp12_prod_seg_quad, dim3(nblk(nseg * parts * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, c->io_out.as<u32>(), (const unsigned long long*)d_offsets,
                     nseg, total, parts, parts > 1 ? c->io_c.as<u32>() : prod);
  LAUNCHCHK();
  if (parts > 1) {
    KLAUNCH(k_fp12_prod_quad, dim3(nblk(nseg * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, c->io_c.as<u32>(), prod, nseg * parts, nseg, parts);
    LAUNCHCHK();
  }
  return final_exp ? final_exp_launch(c, prod, nseg, out) : BLSGPU_OK;
}
extern "C" int blsgpu_multi_miller_loop_many(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint64_t* offsets, size_t nseg,
                                             int final_exp, uint64_t* out) { CTX_CLAIM(c);
  if (!c || (nseg && (!offsets || !out))) return bad("multi_miller_loop_many: NULL argument");
  if (!nseg) return BLSGPU_OK;
  if (offsets[0] != 0) return bad("multi_miller_loop_many: offsets[0] must be 0");
  size_t max_k = 0;
  for (size_t i = 0; i < nseg; i++) {
    if (offsets[i] > offsets[i + 1]) return bad("multi_miller_loop_many: offsets must be non-decreasing");
    if (offsets[i + 1] - offsets[i] > max_k) max_k = (size_t)(offsets[i + 1] - offsets[i]);
  }
  const size_t n = (size_t)offsets[nseg];
  if (n && (!g1 || !g2)) return bad("multi_miller_loop_many: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n ? n * 96 : 16) || c->io_b.reserve(n ? n * 192 : 16) || c->flags_a.reserve(n ? n : 16) || c->flags_b.reserve(n ? n : 16) || c->io_e.reserve((nseg + 1) * 8) ||
      c->io_f.reserve(nseg * 576)) {
    g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP;
  }
  if (n) {
    HIPCHK(hipMemcpyAsync(c->io_a.p, g1, n * 96, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->io_b.p, g2, n * 192, hipMemcpyHostToDevice, c->stream));
    if (g1inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, g1inf, n, hipMemcpyHostToDevice, c->stream));
    if (g2inf) HIPCHK(hipMemcpyAsync(c->flags_b.p, g2inf, n, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(hipMemcpyAsync(c->io_e.p, offsets, (nseg + 1) * 8, hipMemcpyHostToDevice, c->stream));
  int rc = blsgpu_multi_miller_loop_many_device(c, c->io_a.p, g1inf ? c->flags_a.p : nullptr, c->io_b.p, g2inf ? c->flags_b.p : nullptr, c->io_e.p, nseg, n, max_k ? max_k : 1, final_exp,
                                                c->io_f.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_f.p, nseg * 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// ---------------------------------------------------------------------------------------------------
// G2Prepared resident on the device (prep.hip.h): pairings.rs:487-546 (the table), :554-603 (its consumers)
// ---------------------------------------------------------------------------------------------------
struct blsgpu_g2_prepared { int device = 0; size_t n = 0; u32* tab = nullptr; uint8_t* inf = nullptr; hipEvent_t ev_ready = nullptr; };
static void prepared_drop(blsgpu_g2_prepared* p) {
  if (p->tab) hipFree(p->tab);
  if (p->inf) hipFree(p->inf);
  if (p->ev_ready) hipEventDestroy(p->ev_ready);
  delete p;
}
extern "C" int blsgpu_g2_prepare_device(blsgpu_ctx* c, const void* d_g2, const void* d_inf, size_t m, blsgpu_g2_prepared** out) { CTX_CLAIM(c);
  if (!c || !out || (m && !d_g2)) return bad("g2_prepare: NULL argument");
  if (m >= 0xfffffff0ull) return bad("g2_prepare: too many points for 32-bit table indices");
  HIPCHK(hipSetDevice(c->device));
  blsgpu_g2_prepared* p = new blsgpu_g2_prepared();
  p->device = c->device; p->n = m;
  if (hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming) != hipSuccess) { delete p; g_err = "hipEventCreate(g2_prepared) failed"; return BLSGPU_ERR_HIP; }
  if (hipMalloc((void**)&p->tab, (m ? m : 1) * PREP_POINT_WORDS * 4) != hipSuccess || hipMalloc((void**)&p->inf, m ? m : 1) != hipSuccess) {
    (void)hipGetLastError(); prepared_drop(p); g_err = "hipMalloc(g2_prepared) failed"; return BLSGPU_ERR_HIP;
  }
  if (m) KLAUNCH(k_g2_prepare_quad, dim3(nblk(m * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, (const u32*)d_g2, (const uint8_t*)d_inf, m, p->tab, p->inf);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(p->ev_ready, c->stream);          // consumers on another stream (blsgpu_set_stream) wait for the table
  if (e != hipSuccess) { prepared_drop(p); return fail("k_g2_prepare_quad", e, __LINE__); }
  *out = p;
  return BLSGPU_OK;
}
extern "C" int blsgpu_g2_prepare(blsgpu_ctx* c, const uint64_t* g2, const uint8_t* inf, size_t m, blsgpu_g2_prepared** out) { CTX_CLAIM(c);
  if (!c || !out || (m && !g2)) return bad("g2_prepare: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_b.reserve(m ? m * 192 : 16) || c->flags_b.reserve(m ? m : 16)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (m) HIPCHK(hipMemcpyAsync(c->io_b.p, g2, m * 192, hipMemcpyHostToDevice, c->stream));
  if (m && inf) HIPCHK(hipMemcpyAsync(c->flags_b.p, inf, m, hipMemcpyHostToDevice, c->stream));
  blsgpu_g2_prepared* p = nullptr;
  int rc = blsgpu_g2_prepare_device(c, c->io_b.p, inf ? c->flags_b.p : nullptr, m, &p);
  if (rc) return rc;
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { prepared_drop(p); return fail("g2_prepare", e, __LINE__); }
  *out = p;
  return BLSGPU_OK;
}
extern "C" size_t blsgpu_g2_prepared_len(const blsgpu_g2_prepared* p) { return p ? p->n : 0; }
extern "C" void blsgpu_g2_prepared_free(blsgpu_g2_prepared* p) {
  if (!p) return;
  hipSetDevice(p->device);
  hipDeviceSynchronize();                  // an asynchronous Miller loop may still be reading the table
  prepared_drop(p);
}
extern "C" int blsgpu_g2_prepared_coeffs(blsgpu_ctx* c, const blsgpu_g2_prepared* p, size_t index, uint64_t* out, uint8_t* out_inf) { CTX_CLAIM(c);
  if (!c || !p || !out || index >= p->n) return bad("g2_prepared_coeffs: bad argument");
  if (p->device != c->device) return bad("g2_prepared_coeffs: the table lives on another device than the context");
  HIPCHK(hipSetDevice(c->device));
  const size_t bytes = (size_t)PREP_STEPS * 3 * 24 * 4;
  if (c->io_out.reserve(bytes)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipStreamWaitEvent(c->stream, p->ev_ready, 0));
  KLAUNCH(k_g2_prepared_export, dim3(1), dim3(256), 0, c->stream, p->tab, index, c->io_out.as<u32>());
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, bytes, hipMemcpyDeviceToHost, c->stream));
  if (out_inf) HIPCHK(hipMemcpyAsync(out_inf, p->inf + index, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// one launch of k_mml_prep_quad: nseg quads, work area sized for kmax term slots per quad
static int mmlp_launch(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, const void* qidx, const blsgpu_g2_prepared* p, const void* d_off,
                       size_t nseg, size_t total, int kuni, int kmax, void* out) {
  if (p && p->device != c->device) return bad("multi_miller_loop_prepared: the table lives on another device than the context");
  if (kmax < 1) kmax = 1;
  if (kmax > MMLP_MAX_K) kmax = MMLP_MAX_K;
  // The work area is kmax x 260 B per lane.  Segments given by offsets are independent, so a call whose work area would pass
  // MMLP_WORK_MAX is cut into launches over consecutive runs of segments that share ONE bounded area (in stream order): 2^20
  // eight-term segments take nine launches over 1 GiB instead of reserving 8.7 GB.  (The offset-less form -- runs of ONE long product --
  // keeps a single launch: its callers size K themselves.)
  constexpr size_t MMLP_WORK_MAX = (size_t)1 << 30;
  const size_t per_seg = (size_t)kmax * QL * 260;
  size_t seg_cap = nseg;
  if (d_off && nseg * per_seg > MMLP_WORK_MAX) {
    seg_cap = MMLP_WORK_MAX / per_seg;
    seg_cap -= seg_cap % (QUAD_BLOCK / QL);                         // whole workgroups
    if (seg_cap < (size_t)(QUAD_BLOCK / QL)) seg_cap = QUAD_BLOCK / QL;
  }
  const unsigned blocks_max = nblk((seg_cap < nseg ? seg_cap : nseg) * QL, QUAD_BLOCK);
  const size_t threads = (size_t)blocks_max * QUAD_BLOCK;
  // [kmax][threads] u32 meta | [kmax][4][threads] uint4 P | [kmax][12][threads] uint4 running points
  const size_t meta_b = (size_t)kmax * threads * 4, pp_b = (size_t)kmax * 4 * threads * 16, rr_b = (size_t)kmax * 12 * threads * 16;
  if (c->mmlp_work.reserve(meta_b + pp_b + rr_b)) { g_err = "hipMalloc(prepared Miller work area) failed"; return BLSGPU_ERR_HIP; }
  uint8_t* w = c->mmlp_work.as<uint8_t>();
  if (p) HIPCHK(hipStreamWaitEvent(c->stream, p->ev_ready, 0));
  for (size_t s0 = 0; s0 < nseg; s0 += seg_cap) {
    const size_t ns = nseg - s0 < seg_cap ? nseg - s0 : seg_cap;
    // (the kernel strides its work area by ITS grid size: a shorter last launch uses a prefix of each plane)
    KLAUNCH(k_mml_prep_quad, dim3(nblk(ns * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, (const u32*)g1, (const uint8_t*)g1inf, (const u32*)g2, (const uint8_t*)g2inf,
                       (const u32*)(p ? qidx : nullptr), p ? p->tab : (const u32*)nullptr, p ? p->inf : (const uint8_t*)nullptr, (u32)(p ? p->n : 0),
                       d_off ? (const unsigned long long*)d_off + s0 : (const unsigned long long*)nullptr, ns, total, kuni, kmax, (u32*)w, (uint4*)(w + meta_b),
                       (uint4*)(w + meta_b + pp_b), (u32*)out + s0 * 144, c->d_status);
    LAUNCHCHK();
  }
  return BLSGPU_OK;
}
extern "C" int blsgpu_multi_miller_loop_prepared_device(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, const void* qidx,
                                                        const blsgpu_g2_prepared* p, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || !out || (n && !g1) || (n && qidx && !p) || (n && !qidx && !g2)) return bad("multi_miller_loop_prepared: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (!n) return fp12_product_device(c, nullptr, 0, (u32*)out);
  // terms per accumulator: as many as still leave two wavefronts per SIMD busy (2^15 quads)
  const size_t fill = 32768;                                          // quads that fill the chip at two wavefronts per SIMD
  int K = 1;
  while (K < MMLP_MAX_K && n / (2 * (size_t)K) >= fill) K *= 2;
  if (c->mmlp_k > 0) K = c->mmlp_k;
  const size_t groups = (n + K - 1) / K;
  if (c->mmlp_out.reserve(groups * 576)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  int rc = mmlp_launch(c, g1, g1inf, g2, g2inf, qidx, p, nullptr, groups, n, K, K, c->mmlp_out.p);
  if (rc) return rc;
  return fp12_product_device(c, c->mmlp_out.as<u32>(), groups, (u32*)out);
}
static int check_qidx(const uint32_t* qidx, size_t n, const blsgpu_g2_prepared* p, const uint64_t* g2) {
  if (!qidx) return (n && !g2) ? bad("multi_miller_loop_prepared: g2 is NULL and no term is prepared") : BLSGPU_OK;
  if (!p) return bad("multi_miller_loop_prepared: q_index without a prepared table");
  for (size_t i = 0; i < n; i++) {
    if (qidx[i] == BLSGPU_UNPREPARED) { if (!g2) return bad("multi_miller_loop_prepared: an unprepared term needs g2"); }
    else if (qidx[i] >= p->n) return bad("multi_miller_loop_prepared: q_index outside the prepared table");
  }
  return BLSGPU_OK;
}
// host-pointer staging shared by the two prepared entry points: g1 -> io_a, g2 -> io_b, flags -> flags_a/b, indices -> io_e
static int mmlp_stage(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint32_t* qidx, size_t n) {
  if (c->io_a.reserve(n ? n * 96 : 16) || c->io_b.reserve(n ? n * 192 : 16) || c->flags_a.reserve(n ? n : 16) || c->flags_b.reserve(n ? n : 16) || c->io_e.reserve(n ? n * 4 : 16)) {
    g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP;
  }
  if (!n) return BLSGPU_OK;
  HIPCHK(hipMemcpyAsync(c->io_a.p, g1, n * 96, hipMemcpyHostToDevice, c->stream));
  if (g2) HIPCHK(hipMemcpyAsync(c->io_b.p, g2, n * 192, hipMemcpyHostToDevice, c->stream));
  if (g1inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, g1inf, n, hipMemcpyHostToDevice, c->stream));
  if (g2 && g2inf) HIPCHK(hipMemcpyAsync(c->flags_b.p, g2inf, n, hipMemcpyHostToDevice, c->stream));
  if (qidx) HIPCHK(hipMemcpyAsync(c->io_e.p, qidx, n * 4, hipMemcpyHostToDevice, c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_multi_miller_loop_prepared(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint32_t* qidx,
                                                 const blsgpu_g2_prepared* p, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!c || !out || (n && !g1)) return bad("multi_miller_loop_prepared: NULL argument");
  if (int rc = check_qidx(qidx, n, p, g2)) return rc;
  HIPCHK(hipSetDevice(c->device));
  if (c->result.reserve(576)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (int rc = mmlp_stage(c, g1, g1inf, g2, g2inf, qidx, n)) return rc;
  int rc = blsgpu_multi_miller_loop_prepared_device(c, c->io_a.p, g1inf ? c->flags_a.p : nullptr, g2 ? c->io_b.p : nullptr, (g2 && g2inf) ? c->flags_b.p : nullptr,
                                                    qidx ? c->io_e.p : nullptr, p, n, c->result.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->result.p, 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_multi_miller_loop_prepared_many_device(blsgpu_ctx* c, const void* g1, const void* g1inf, const void* g2, const void* g2inf, const void* qidx,
                                                             const blsgpu_g2_prepared* p, const void* d_off, size_t nseg, size_t total, size_t max_seg_terms, int final_exp,
                                                             void* out) { CTX_CLAIM(c);
  if (!c || (nseg && (!d_off || !out)) || (total && !g1) || (total && qidx && !p) || (total && !qidx && !g2)) return bad("multi_miller_loop_prepared_many: NULL argument");
  if (!nseg) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (final_exp && c->io_d.reserve(nseg * 576)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  u32* prod = final_exp ? c->io_d.as<u32>() : (u32*)out;
  // terms per pass: the caller's bound, or -- bound unknown -- the mean segment length (a longer segment simply takes more passes, see
  // k_mml_prep_quad), so that the work area stays proportional to the input: 2^20 two-term segments no longer reserve eight slots each
  int kmax = (max_seg_terms == 0 || max_seg_terms > (size_t)MMLP_MAX_K) ? MMLP_MAX_K : (int)max_seg_terms;
  if (max_seg_terms == 0) {
    const size_t mean = (total + nseg - 1) / nseg;
    if (mean < (size_t)kmax) kmax = mean < 1 ? 1 : (int)mean;
  }
  int rc = mmlp_launch(c, g1, g1inf, g2, g2inf, qidx, p, d_off, nseg, total, 0, kmax, prod);
  if (rc) return rc;
  return final_exp ? final_exp_launch(c, prod, nseg, out) : BLSGPU_OK;
}
extern "C" int blsgpu_multi_miller_loop_prepared_many(blsgpu_ctx* c, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint32_t* qidx,
                                                      const blsgpu_g2_prepared* p, const uint64_t* offsets, size_t nseg, int final_exp, uint64_t* out) { CTX_CLAIM(c);
  if (!c || (nseg && (!offsets || !out))) return bad("multi_miller_loop_prepared_many: NULL argument");
  if (!nseg) return BLSGPU_OK;
  if (offsets[0] != 0) return bad("multi_miller_loop_prepared_many: offsets[0] must be 0");
  size_t max_k = 0;
  for (size_t i = 0; i < nseg; i++) {
    if (offsets[i] > offsets[i + 1]) return bad("multi_miller_loop_prepared_many: offsets must be non-decreasing");
    if (offsets[i + 1] - offsets[i] > max_k) max_k = (size_t)(offsets[i + 1] - offsets[i]);
  }
  const size_t n = (size_t)offsets[nseg];
  if (n && !g1) return bad("multi_miller_loop_prepared_many: NULL argument");
  if (int rc = check_qidx(qidx, n, p, g2)) return rc;
  HIPCHK(hipSetDevice(c->device));
  if (c->io_f.reserve(nseg * 576) || c->io_c.reserve((nseg + 1) * 8)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (int rc = mmlp_stage(c, g1, g1inf, g2, g2inf, qidx, n)) return rc;
  HIPCHK(hipMemcpyAsync(c->io_c.p, offsets, (nseg + 1) * 8, hipMemcpyHostToDevice, c->stream));
  int rc = blsgpu_multi_miller_loop_prepared_many_device(c, c->io_a.p, g1inf ? c->flags_a.p : nullptr, g2 ? c->io_b.p : nullptr, (g2 && g2inf) ? c->flags_b.p : nullptr,
                                                         qidx ? c->io_e.p : nullptr, p, c->io_c.p, nseg, n, max_k ? max_k : 1, final_exp, c->io_f.p);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_f.p, nseg * 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_final_exponentiation_batch(blsgpu_ctx* c, const uint64_t* in, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!c || (n && (!in || !out))) return bad("final_exponentiation: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n * 576) || c->io_out.reserve(n * 576)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, in, n * 576, hipMemcpyHostToDevice, c->stream));
  const int layout = pairing_layout_for(c, n);
  if (layout < 0) return wide_missing(c);
  if (layout == 256) wide_launch(c, 2, c->io_a.p, nullptr, nullptr, nullptr, n, c->io_out.p);
  else if (layout == 4) KLAUNCH(k_final_exp_quad, dim3(nblk(n * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, c->io_a.as<u32>(), c->io_out.as<u32>(), n);
  else KLAUNCH(k_final_exp, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, c->io_a.as<u32>(), c->io_out.as<u32>(), n);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_gt_mul_scalar_batch_device(blsgpu_ctx* c, const void* gt, const void* scalars, size_t n, void* out) { CTX_CLAIM(c);
  if (!c || (n && (!gt || !scalars || !out))) return bad("gt_mul_scalar: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  KLAUNCH(k_gt_mul_scalar, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, (const u32*)gt, (const u32*)scalars, (u32*)out, n, c->scalar_form);
  LAUNCHCHK();
  return BLSGPU_OK;
}
// flags[i] = (gt[i] == Fp12::one()): word-wise comparison with the canonical wire form of one (written once per context by k_fp12_one)
__global__ void __launch_bounds__(256) k_fp12_equals(const u32* __restrict__ gt, const u32* __restrict__ one, size_t n, uint8_t* __restrict__ flags) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* a = reinterpret_cast<const uint4*>(gt + i * 144);
  const uint4* b = reinterpret_cast<const uint4*>(one);
  u32 diff = 0;
  for (int k = 0; k < 36; k++) { const uint4 x = a[k], y = b[k]; diff |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w); }
  flags[i] = diff == 0 ? 1 : 0;
}
static int gt_one_ready(blsgpu_ctx* c) {
  if (c->gt_one_ready) return BLSGPU_OK;
  if (c->gt_one.reserve(576)) { g_err = "hipMalloc failed"; return BLSGPU_ERR_HIP; }
  KLAUNCH(k_fp12_one, dim3(1), dim3(64), 0, c->stream, c->gt_one.as<u32>());
  LAUNCHCHK();
  HIPCHK(hipEventRecord(c->ev_gt_one, c->stream));
  c->gt_one_ready = true;
  return BLSGPU_OK;
}
extern "C" int blsgpu_gt_is_identity_device(blsgpu_ctx* c, const void* gt, size_t n, void* flags) { CTX_CLAIM(c);
  if (!c || (n && (!gt || !flags))) return bad("gt_is_identity: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (int rc = gt_one_ready(c)) return rc;
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_gt_one, 0));
  KLAUNCH(k_fp12_equals, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (const u32*)gt, c->gt_one.as<u32>(), n, (uint8_t*)flags);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_gt_mul_scalar_batch(blsgpu_ctx* c, const uint64_t* gt, const uint8_t* scalars, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!c || (n && (!gt || !scalars || !out))) return bad("gt_mul_scalar: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n * 576) || c->io_b.reserve(n * 32) || c->io_out.reserve(n * 576)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, gt, n * 576, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->io_b.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  KLAUNCH(k_gt_mul_scalar, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, c->io_a.as<u32>(), c->io_b.as<u32>(), c->io_out.as<u32>(), n, c->scalar_form);
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_fp12_product(blsgpu_ctx* c, const uint64_t* in, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (!c || !out || (n && !in)) return bad("fp12_product: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->io_a.reserve(n ? n * 576 : 16) || c->result.reserve(576)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (n) HIPCHK(hipMemcpyAsync(c->io_a.p, in, n * 576, hipMemcpyHostToDevice, c->stream));
  int rc = fp12_product_device(c, c->io_a.as<u32>(), n, c->result.as<u32>());
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, c->result.p, 576, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// batched point (de)serialisation + validation  (codec.hip.h)
// ---------------------------------------------------------------------------------------------------
template <class F>
static int point_decode_device(blsgpu_ctx* c, const void* d_bytes, size_t n, int compressed, int checked, void* d_xy, void* d_inf, void* d_ok) {
  if (!c || (n && (!d_bytes || !d_xy || !d_inf || !d_ok))) return bad("decode: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  KLAUNCH(k_point_decode<F>, dim3(nblk(n, 128)), dim3(128), 0, c->stream, (const uint8_t*)d_bytes, n, (compressed ? 1 : 0) | (checked ? 2 : 0), (u32*)d_xy, (uint8_t*)d_inf,
                     (uint8_t*)d_ok);
  LAUNCHCHK();
  return BLSGPU_OK;
}
template <class F>
static int point_encode_device(blsgpu_ctx* c, const void* d_xy, const void* d_inf, size_t n, int compressed, void* d_out) {
  if (!c || (n && (!d_xy || !d_out))) return bad("encode: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  KLAUNCH(k_point_encode<F>, dim3(nblk(n, 128)), dim3(128), 0, c->stream, (const u32*)d_xy, (const uint8_t*)d_inf, n, compressed, (uint8_t*)d_out);
  LAUNCHCHK();
  return BLSGPU_OK;
}
template <class F>
static int point_decode(blsgpu_ctx* c, const uint8_t* bytes, size_t n, int compressed, int checked, uint64_t* xy, uint8_t* inf, uint8_t* ok) {
  if (!c || (n && (!bytes || !xy || !inf || !ok))) return bad("decode: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  constexpr int CB = Codec<F>::COORD_BYTES, WW = Wire<F>::WORDS;
  size_t ib = n * (compressed ? CB : 2 * CB), xb = n * 2 * WW * 4;
  if (c->io_a.reserve(ib) || c->io_out.reserve(xb) || c->flags_a.reserve(n) || c->flags_b.reserve(n)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, bytes, ib, hipMemcpyHostToDevice, c->stream));
  if (int rc = point_decode_device<F>(c, c->io_a.p, n, compressed, checked, c->io_out.p, c->flags_a.p, c->flags_b.p)) return rc;
  HIPCHK(hipMemcpyAsync(xy, c->io_out.p, xb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(inf, c->flags_a.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(ok, c->flags_b.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
template <class F>
static int point_encode(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, int compressed, uint8_t* out) {
  if (!c || (n && (!xy || !out))) return bad("encode: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  constexpr int CB = Codec<F>::COORD_BYTES, WW = Wire<F>::WORDS;
  size_t ob = n * (compressed ? CB : 2 * CB), xb = n * 2 * WW * 4;
  if (c->io_a.reserve(xb) || c->io_out.reserve(ob) || c->flags_a.reserve(n)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, xy, xb, hipMemcpyHostToDevice, c->stream));
  if (inf) HIPCHK(hipMemcpyAsync(c->flags_a.p, inf, n, hipMemcpyHostToDevice, c->stream));
  if (int rc = point_encode_device<F>(c, c->io_a.p, inf ? c->flags_a.p : nullptr, n, compressed, c->io_out.p)) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, ob, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_from_bytes_batch_device(blsgpu_ctx* c, const void* b, size_t n, int compressed, int checked, void* xy, void* inf, void* ok) { CTX_CLAIM(c);
  return point_decode_device<FpPolicy>(c, b, n, compressed, checked, xy, inf, ok);
}
extern "C" int blsgpu_g2_from_bytes_batch_device(blsgpu_ctx* c, const void* b, size_t n, int compressed, int checked, void* xy, void* inf, void* ok) { CTX_CLAIM(c);
  return point_decode_device<Fp2Policy>(c, b, n, compressed, checked, xy, inf, ok);
}
extern "C" int blsgpu_g1_to_bytes_batch_device(blsgpu_ctx* c, const void* xy, const void* inf, size_t n, int compressed, void* out) { CTX_CLAIM(c);
  return point_encode_device<FpPolicy>(c, xy, inf, n, compressed, out);
}
extern "C" int blsgpu_g2_to_bytes_batch_device(blsgpu_ctx* c, const void* xy, const void* inf, size_t n, int compressed, void* out) { CTX_CLAIM(c);
  return point_encode_device<Fp2Policy>(c, xy, inf, n, compressed, out);
}
extern "C" int blsgpu_g1_from_bytes_batch(blsgpu_ctx* c, const uint8_t* b, size_t n, int compressed, int checked, uint64_t* xy, uint8_t* inf, uint8_t* ok) { CTX_CLAIM(c);
  return point_decode<FpPolicy>(c, b, n, compressed, checked, xy, inf, ok);
}
extern "C" int blsgpu_g2_from_bytes_batch(blsgpu_ctx* c, const uint8_t* b, size_t n, int compressed, int checked, uint64_t* xy, uint8_t* inf, uint8_t* ok) { CTX_CLAIM(c);
  return point_decode<Fp2Policy>(c, b, n, compressed, checked, xy, inf, ok);
}
extern "C" int blsgpu_g1_to_bytes_batch(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, int compressed, uint8_t* out) { CTX_CLAIM(c);
  return point_encode<FpPolicy>(c, xy, inf, n, compressed, out);
}
extern "C" int blsgpu_g2_to_bytes_batch(blsgpu_ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, int compressed, uint8_t* out) { CTX_CLAIM(c);
  return point_encode<Fp2Policy>(c, xy, inf, n, compressed, out);
}

// ---------------------------------------------------------------------------------------------------
// bulk BLS signature verification: compressed bytes in -> verdict bytes out, every stage on the device
// ---------------------------------------------------------------------------------------------------
// consts[0..24): the affine wire coordinates of -G1 (g1.rs:86-104 negated, :126-134); consts[24..72): of -G2 (g2.rs:103-140)
__global__ void k_bls_consts(u32* __restrict__ consts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Aff<FpPolicy> g = generator<FpPolicy>();
  Wire<FpPolicy>::save(g.x, consts); Wire<FpPolicy>::save(neg(g.y), consts + 12);
  const Aff<Fp2Policy> h = generator<Fp2Policy>();
  Wire<Fp2Policy>::save(h.x, consts + 24); Wire<Fp2Policy>::save(neg(h.y), consts + 48);
}
// the two terms of equation i (segment i = terms 2 i, 2 i + 1).  A point whose decoding failed is flagged as the identity so that
// the Miller kernels never see unvalidated limbs; its verdict comes from the ok flags.
//   mode 0:  (pk_i, H_i), (-G1, sig_i)                       mode 1:  (sig_i, table[0] = -G2), (H_i, pk_i)
__global__ void __launch_bounds__(256) k_bls_assemble(int mode, const u32* __restrict__ pk, const uint8_t* __restrict__ pk_inf, const uint8_t* __restrict__ pk_ok,
                                                      const u32* __restrict__ sig, const uint8_t* __restrict__ sig_inf, const uint8_t* __restrict__ sig_ok,
                                                      const u32* __restrict__ h, const uint8_t* __restrict__ h_inf, const u32* __restrict__ consts, size_t n,
                                                      u32* __restrict__ g1t, uint8_t* __restrict__ g1f, u32* __restrict__ g2t, uint8_t* __restrict__ g2f,
                                                      u32* __restrict__ qidx, unsigned long long* __restrict__ off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  off[i] = 2ull * i;
  if (i == n) return;
  const bool pbad = !pk_ok[i], sbad = !sig_ok[i];
  u32* a0 = g1t + (2 * i) * 24; u32* a1 = a0 + 24;
  u32* b0 = g2t + (2 * i) * 48; u32* b1 = b0 + 48;
  if (mode == 0) {
    for (int k = 0; k < 24; k++) { a0[k] = pk[i * 24 + k]; a1[k] = consts[k]; }
    for (int k = 0; k < 48; k++) { b0[k] = h[i * 48 + k]; b1[k] = sig[i * 48 + k]; }
    g1f[2 * i] = (pk_inf[i] || pbad) ? 1 : 0; g1f[2 * i + 1] = 0;
    g2f[2 * i] = h_inf[i]; g2f[2 * i + 1] = (sig_inf[i] || sbad) ? 1 : 0;
    qidx[2 * i] = PREP_NONE; qidx[2 * i + 1] = PREP_NONE;
  } else {
    for (int k = 0; k < 24; k++) { a0[k] = sig[i * 24 + k]; a1[k] = h[i * 24 + k]; }
    for (int k = 0; k < 48; k++) { b0[k] = 0; b1[k] = pk[i * 48 + k]; }
    g1f[2 * i] = (sig_inf[i] || sbad) ? 1 : 0; g1f[2 * i + 1] = h_inf[i];
    g2f[2 * i] = 0; g2f[2 * i + 1] = (pk_inf[i] || pbad) ? 1 : 0;
    qidx[2 * i] = 0; qidx[2 * i + 1] = PREP_NONE;
  }
}
__global__ void __launch_bounds__(256) k_bls_verdict(const uint8_t* __restrict__ is_one, const uint8_t* __restrict__ pk_ok, const uint8_t* __restrict__ sig_ok, size_t n,
                                                     uint8_t* __restrict__ verdict) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  verdict[i] = !pk_ok[i] ? 2 : !sig_ok[i] ? 3 : is_one[i] ? 1 : 0;
}
extern "C" int blsgpu_bls_verify_batch_device(blsgpu_ctx* c, int mode, const void* d_pk, const void* d_sig, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst,
                                              size_t dst_len, void* d_verdict) { CTX_CLAIM(c);
  if (!c || (n && (!d_pk || !d_sig || !d_offsets || !d_verdict)) || (dst_len && !d_dst)) return bad("bls_verify_batch: NULL argument");
  if (mode != 0 && mode != 1) return bad("bls_verify_batch: mode must be 0 (public keys in G1) or 1 (public keys in G2)");
  if (dst_len > 255) return bad("bls_verify_batch_device: reduce a DST longer than 255 bytes on the host first");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  // G1-side and G2-side points of the equation: mode 0 = (pk, sig), mode 1 = (sig, pk); the hash goes to the signature's group
  const size_t a_xy = n * 96, b_xy = n * 192, h_xyz = n * (mode == 0 ? 288 : 144), h_xy = n * (mode == 0 ? 192 : 96);
  const size_t al = 256;
  auto up = [&](size_t x) { return (x + al - 1) / al * al; };
  size_t o = 0;
  const size_t o_consts = o; o += up(288);
  const size_t o_a = o; o += up(a_xy);
  const size_t o_b = o; o += up(b_xy);
  const size_t o_hp = o; o += up(h_xyz);
  const size_t o_h = o; o += up(h_xy);
  const size_t o_fl = o; o += up(6 * n);                // a_inf a_ok b_inf b_ok h_inf is_one
  const size_t o_g1t = o; o += up(2 * n * 96);
  const size_t o_g2t = o; o += up(2 * n * 192);
  const size_t o_tf = o; o += up(4 * n);                // g1f (2n) g2f (2n)
  const size_t o_qi = o; o += up(2 * n * 4);
  const size_t o_off = o; o += up((n + 1) * 8);
  const size_t o_gt = o; o += up(n * 576);
  const bool fresh = c->ver.cap < o;
  if (c->ver.reserve(o)) { g_err = "hipMalloc(bulk verification) failed"; return BLSGPU_ERR_HIP; }
  uint8_t* base = c->ver.as<uint8_t>();
  if (fresh || !c->ver_consts_ready) {
    KLAUNCH(k_bls_consts, dim3(1), dim3(64), 0, c->stream, (u32*)(base + o_consts));
    LAUNCHCHK();
    HIPCHK(hipEventRecord(c->ev_ver, c->stream));
    c->ver_consts_ready = true;
  }
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_ver, 0));
  if (mode == 1 && !c->ver_table) {
    // `G2Prepared::from(-G2Affine::generator())`, once per context (its own allocation: it outlives a regrown scratch block)
    int rc = blsgpu_g2_prepare_device(c, base + o_consts + 96, nullptr, 1, &c->ver_table);
    if (rc) return rc;
  }
  uint8_t* fl = base + o_fl;
  uint8_t *a_inf = fl, *a_ok = fl + n, *b_inf = fl + 2 * n, *b_ok = fl + 3 * n, *h_inf = fl + 4 * n, *is_one = fl + 5 * n;
  // 1.-3. independent, latency-shaped stages (one lane or lane pair per point, a few thousand field multiplications each): the two
  // checked decodings run on the context's stream, hash-to-curve + normalisation beside them on a side stream (ONE side stream: the
  // runtime multiplexes streams onto a few hardware queues, and two side streams created back to back shared one -- kernel trace of
  // round 5 -- which serialised exactly the two longest stages), and they meet again before the terms are assembled
  if (!c->ver_stream[0]) {
    for (auto& q : c->ver_stream) HIPCHK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    for (auto& e : c->ev_ver_side) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  hipStream_t main_stream = c->stream;
  HIPCHK(hipEventRecord(c->ev_ver_side[0], main_stream));
  int rc = BLSGPU_OK;
  {
    // hash the messages to the signature's group and normalise, on the side stream
    c->stream = c->ver_stream[0];
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_ver_side[0], 0);
    // (the plain form of the hash even for a small batch: the split form buys latency with 45 % more lane-time, which only pays while the chip
    // has nothing else to do -- here the decoders run beside it; measured 2^14 signatures: 15.5 ms plain, 18.7-18.9 ms split (16.6 / 21.7 ms with the
    // slower G2 decoder of before); BLSGPU_VERIFY_H2C_SPLIT=1 lets the batch-size rule apply here too, for re-measuring)
    const int keep_split = c->h2c_split;
    c->h2c_split = getenv("BLSGPU_VERIFY_H2C_SPLIT") ? keep_split : 0;
    if (e == hipSuccess) rc = blsgpu_hash_to_curve_device(c, mode == 0 ? 2 : 1, d_msgs, d_offsets, n, d_dst, dst_len, 0, base + o_hp);
    c->h2c_split = keep_split;
    if (e == hipSuccess && !rc)
      rc = mode == 0 ? batch_normalize_device<Fp2Policy>(c, base + o_hp, n, base + o_h, h_inf) : batch_normalize_device<FpPolicy>(c, base + o_hp, n, base + o_h, h_inf);
    if (e == hipSuccess && !rc) e = hipEventRecord(c->ev_ver_side[1], c->stream);
    c->stream = main_stream;
    if (e != hipSuccess) return fail("bls_verify_batch: side stream", e, __LINE__);
    if (rc) return rc;
  }
  // checked decoding (`from_compressed`: on the curve, in the subgroup) of both point arrays on the context's stream, which then waits for the side stream
  rc = point_decode_device<Fp2Policy>(c, mode == 0 ? d_sig : d_pk, n, 1, 1, base + o_b, b_inf, b_ok);
  if (rc) return rc;
  rc = point_decode_device<FpPolicy>(c, mode == 0 ? d_pk : d_sig, n, 1, 1, base + o_a, a_inf, a_ok);
  if (rc) return rc;
  HIPCHK(hipStreamWaitEvent(main_stream, c->ev_ver_side[1], 0));
  // 4. the two terms of every equation
  const uint8_t *pk_inf = mode == 0 ? a_inf : b_inf, *pk_ok = mode == 0 ? a_ok : b_ok, *sig_inf = mode == 0 ? b_inf : a_inf, *sig_ok = mode == 0 ? b_ok : a_ok;
  KLAUNCH(k_bls_assemble, dim3(nblk(n + 1, 256)), dim3(256), 0, c->stream, mode, (const u32*)(base + (mode == 0 ? o_a : o_b)), pk_inf, pk_ok,
                     (const u32*)(base + (mode == 0 ? o_b : o_a)), sig_inf, sig_ok, (const u32*)(base + o_h), h_inf, (const u32*)(base + o_consts), n, (u32*)(base + o_g1t),
                     base + o_tf, (u32*)(base + o_g2t), base + o_tf + 2 * n, (u32*)(base + o_qi), (unsigned long long*)(base + o_off));
  LAUNCHCHK();
  // 5. one multi_miller_loop + final exponentiation per equation
  if (mode == 0)
    rc = blsgpu_multi_miller_loop_many_device(c, base + o_g1t, base + o_tf, base + o_g2t, base + o_tf + 2 * n, base + o_off, n, 2 * n, 2, 1, base + o_gt);
  else
    rc = blsgpu_multi_miller_loop_prepared_many_device(c, base + o_g1t, base + o_tf, base + o_g2t, base + o_tf + 2 * n, base + o_qi, c->ver_table, base + o_off, n, 2 * n, 2, 1,
                                                       base + o_gt);
  if (rc) return rc;
  // 6. == Gt::identity()?
  rc = blsgpu_gt_is_identity_device(c, base + o_gt, n, is_one);
  if (rc) return rc;
  KLAUNCH(k_bls_verdict, dim3(nblk(n, 256)), dim3(256), 0, c->stream, is_one, pk_ok, sig_ok, n, (uint8_t*)d_verdict);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_bls_verify_batch(blsgpu_ctx* c, int mode, const uint8_t* pk, const uint8_t* sig, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst,
                                       size_t dst_len, uint8_t* verdict) { CTX_CLAIM(c);
  if (!c || (n && (!pk || !sig || !offsets || !verdict)) || (dst_len && !dst)) return bad("bls_verify_batch: NULL argument");
  if (mode != 0 && mode != 1) return bad("bls_verify_batch: mode must be 0 (public keys in G1) or 1 (public keys in G2)");
  if (!n) return BLSGPU_OK;
  for (size_t i = 0; i < n; i++) if (offsets[i] > offsets[i + 1]) return bad("bls_verify_batch: offsets must be non-decreasing");
  const size_t total = (size_t)offsets[n];
  if (total && !msgs) return bad("bls_verify_batch: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  uint8_t d[255]; u32 dlen;
  if (dst_len > 255) {                    // a DST longer than 255 bytes is replaced by H("H2C-OVERSIZE-DST-" || DST)  (expand_msg.rs:74-95)
    Sha256 sh; sha_init(sh);
    const char* salt = "H2C-OVERSIZE-DST-";
    for (int i = 0; salt[i]; i++) sha_put(sh, (uint8_t)salt[i]);
    for (size_t i = 0; i < dst_len; i++) sha_put(sh, dst[i]);
    u32 hw[8]; sha_finish(sh, hw);
    for (int i = 0; i < 32; i++) d[i] = (uint8_t)(hw[i >> 2] >> (24 - 8 * (i & 3)));
    dlen = 32;
  } else {
    for (size_t i = 0; i < dst_len; i++) d[i] = dst[i];
    dlen = (u32)dst_len;
  }
  const size_t pkb = n * (mode == 0 ? 48 : 96), sgb = n * (mode == 0 ? 96 : 48);
  // ONE staging block: pk | sig | msgs | offsets | dst | verdict
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t o_pk = 0, o_sg = up(pkb), o_ms = o_sg + up(sgb), o_of = o_ms + up(total + 16), o_ds = o_of + up((n + 1) * 8), o_vd = o_ds + 256, bytes = o_vd + up(n);
  if (c->io_a.reserve(bytes)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  uint8_t* b = c->io_a.as<uint8_t>();
  HIPCHK(hipMemcpyAsync(b + o_pk, pk, pkb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(b + o_sg, sig, sgb, hipMemcpyHostToDevice, c->stream));
  if (total) HIPCHK(hipMemcpyAsync(b + o_ms, msgs, total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(b + o_of, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (dlen) HIPCHK(hipMemcpyAsync(b + o_ds, d, dlen, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));                 // `d` lives on this stack frame
  int rc = blsgpu_bls_verify_batch_device(c, mode, b + o_pk, b + o_sg, b + o_ms, b + o_of, n, b + o_ds, dlen, b + o_vd);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(verdict, b + o_vd, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// device groups: the hot path sharded over the GPUs of ONE node from ONE process
// ---------------------------------------------------------------------------------------------------
// MSMs, batches of pairings and multi_miller_loops shard over their independent terms in contiguous slices; every
// member reduces its slice to ONE group element (144 B G1 / 288 B G2 / 576 B Fp12) and the members' partial results are folded with
// the reference's own operators -- `Sum for G1Projective` (g1.rs:161-171, g2.rs:162-172), `MillerLoopResult + MillerLoopResult`
// (pairings.rs:179-186) -- followed by ONE final exponentiation (:48-176).  Inside one process the exchange needs no collective
// library: each member hands its few hundred bytes back through host memory and member 0 folds them on its device.  One context and
// one host thread per member (a context is single-threaded by contract); a device may be listed more than once (logical members on
// one GPU: how the single-GPU tests exercise the 8-member code path).
// One PERSISTENT worker thread per member beyond the first (created by blsgpu_group_create; member 0 runs on the caller's thread): a
// sharded call posts one job per member and waits -- no thread is created or joined per call.  A context is driven by one host thread at
// a time, and a member's context is only ever driven by its worker (or, for member 0, by the thread inside the group call).
struct GroupWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has = false, done = true, quit = false;
  int rc = BLSGPU_OK;
  std::string msg;
  void loop() {
    for (;;) {
      std::function<int()> j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return has || quit; });
        if (quit) return;
        j = std::move(job); has = false;
      }
      int r; std::string e;
      try { r = j(); if (r) e = g_err; } catch (...) { r = BLSGPU_ERR_HIP; e = "exception in a group worker"; }
      {
        std::lock_guard<std::mutex> lk(m);
        rc = r; msg = e; done = true;
      }
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    { std::lock_guard<std::mutex> lk(m); job = std::move(j); has = true; done = false; }
    cv.notify_all();
  }
  int wait(std::string& e) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done; });
    e = msg;
    return rc;
  }
};
struct blsgpu_group {
  std::vector<blsgpu_ctx*> ctx;
  std::vector<GroupWorker*> worker;            // worker[i] drives member i, i >= 1 (worker[0] is null)
  std::vector<void*> pinned;                   // per member: 576 B of pinned host memory for its partial result
  // asynchronous fold (blsgpu_g{1,2}_partials_fold_device): four staging rows of w partial sums on member 0's device, one event per
  // member (its copy has been queued) and one per staging row (the sum that read it has been queued)
  void* fold_in = nullptr;
  // per member: the last eight folds' (partial-sum buffer, "its copy has run" event): an MSM that is about to overwrite a buffer a fold
  // still has to read waits for that fold's copy (a pipelined caller rotates >= 8 buffers, so the event it meets is long complete)
  struct FoldRead { const void* ptr = nullptr; hipEvent_t ev = nullptr; bool used = false; };
  std::vector<std::array<FoldRead, 8>> fold_reads;
  std::vector<hipEvent_t> ev_copy, ev_main;
  hipEvent_t ev_sum[4] = {};
  bool ev_sum_used[4] = {false, false, false, false};
  unsigned fold_seq = 0;
};
struct blsgpu_group_bases { int group = 1; size_t n = 0; std::vector<blsgpu_bases*> part; };

// contiguous slice [lo, hi) of n items owned by member k of w (sizes differ by at most one; the same rule as distributed.shard_range)
static void group_range(size_t n, size_t k, size_t w, size_t& lo, size_t& hi) {
  const size_t q = n / w, r = n % w;
  lo = k * q + (k < r ? k : r);
  hi = lo + q + (k < r ? 1 : 0);
}
// fn(member, context) on the member's persistent worker (member 0 on the caller's thread); the first failing member's code and message win
template <class Fn> static int group_run(blsgpu_group* g, Fn fn) {
  const size_t w = g->ctx.size();
  for (size_t i = 1; i < w; i++) g->worker[i]->post([&fn, g, i]() { return fn(i, g->ctx[i]); });
  int rc0 = fn(0, g->ctx[0]);
  std::string msg0 = rc0 ? g_err : std::string();
  int rc = BLSGPU_OK; std::string msg; size_t who = 0;
  if (rc0) { rc = rc0; msg = msg0; }
  for (size_t i = 1; i < w; i++) {             // every posted job is awaited, whatever failed (the jobs reference this frame)
    std::string e;
    int r = g->worker[i]->wait(e);
    if (r && !rc) { rc = r; msg = e; who = i; }
  }
  if (rc) { g_err = "group member " + std::to_string(who) + ": " + msg; return rc; }
  return BLSGPU_OK;
}
extern "C" void blsgpu_group_destroy(blsgpu_group* g) {
  if (!g) return;
  for (auto wk : g->worker) {
    if (!wk) continue;
    { std::lock_guard<std::mutex> lk(wk->m); wk->quit = true; }
    wk->cv.notify_all();
    if (wk->th.joinable()) wk->th.join();
    delete wk;
  }
  // peer copies into fold_in that members queued on their own fold streams may still be in flight: drain every member before any
  // event or buffer of the fold goes away
  for (auto c : g->ctx) if (c) { hipSetDevice(c->device); hipDeviceSynchronize(); }
  for (size_t i = 0; i < g->ev_copy.size(); i++) if (g->ev_copy[i]) { hipSetDevice(g->ctx[i]->device); hipEventDestroy(g->ev_copy[i]); }
  for (size_t i = 0; i < g->ev_main.size(); i++) if (g->ev_main[i]) { hipSetDevice(g->ctx[i]->device); hipEventDestroy(g->ev_main[i]); }
  for (size_t i = 0; i < g->fold_reads.size(); i++) for (auto& fr : g->fold_reads[i]) if (fr.ev) { hipSetDevice(g->ctx[i]->device); hipEventDestroy(fr.ev); }
  if (!g->ctx.empty() && g->ctx[0]) {
    hipSetDevice(g->ctx[0]->device);
    hipDeviceSynchronize();
    for (auto e : g->ev_sum) if (e) hipEventDestroy(e);
    if (g->fold_in) hipFree(g->fold_in);
  }
  for (size_t i = 0; i < g->pinned.size(); i++) if (g->pinned[i]) { if (i < g->ctx.size() && g->ctx[i]) hipSetDevice(g->ctx[i]->device); hipHostFree(g->pinned[i]); }
  for (auto c : g->ctx) blsgpu_destroy(c);
  delete g;
}
extern "C" int blsgpu_group_create(const int* devices, int ndev, blsgpu_group** out) {
  if (!out || !devices || ndev <= 0 || ndev > 64) return bad("group_create: bad argument (1..64 members)");
  blsgpu_group* g = new blsgpu_group();
  for (int i = 0; i < ndev; i++) {
    blsgpu_ctx* c = nullptr;
    int rc = blsgpu_create(devices[i], &c);
    if (rc) { const std::string keep = g_err; blsgpu_group_destroy(g); g_err = "group_create: member " + std::to_string(i) + ": " + keep; return rc; }
    g->ctx.push_back(c);
    void* pin = nullptr;
    if (hipHostMalloc(&pin, 576, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); blsgpu_group_destroy(g); g_err = "group_create: hipHostMalloc failed"; return BLSGPU_ERR_HIP; }
    g->pinned.push_back(pin);
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); blsgpu_group_destroy(g); g_err = "group_create: hipEventCreate failed"; return BLSGPU_ERR_HIP; }
    g->ev_copy.push_back(ev);
    hipEvent_t evm = nullptr;
    if (hipEventCreateWithFlags(&evm, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); blsgpu_group_destroy(g); g_err = "group_create: hipEventCreate failed"; return BLSGPU_ERR_HIP; }
    g->ev_main.push_back(evm);
    g->fold_reads.emplace_back();
    for (auto& fr : g->fold_reads.back())
      if (hipEventCreateWithFlags(&fr.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); blsgpu_group_destroy(g); g_err = "group_create: hipEventCreate failed"; return BLSGPU_ERR_HIP; }
  }
  {
    hipError_t e = hipSetDevice(g->ctx[0]->device);
    if (e == hipSuccess) e = hipMalloc(&g->fold_in, (size_t)4 * ndev * 576);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&g->ev_sum[i], hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipGetLastError(); blsgpu_group_destroy(g); g_err = "group_create: staging for the asynchronous fold could not be allocated"; return BLSGPU_ERR_HIP; }
  }
  g->worker.assign((size_t)ndev, nullptr);
  try {
    for (int i = 1; i < ndev; i++) {
      g->worker[(size_t)i] = new GroupWorker();
      g->worker[(size_t)i]->th = std::thread([wk = g->worker[(size_t)i]] { wk->loop(); });
    }
  } catch (...) {                                // e.g. the process's thread limit: nothing escapes the C ABI
    blsgpu_group_destroy(g);
    g_err = "group_create: a worker thread could not be started";
    return BLSGPU_ERR_HIP;
  }
  *out = g;
  return BLSGPU_OK;
}
extern "C" int blsgpu_group_size(const blsgpu_group* g) { return g ? (int)g->ctx.size() : 0; }
extern "C" blsgpu_ctx* blsgpu_group_ctx(blsgpu_group* g, int member) { return (g && member >= 0 && (size_t)member < g->ctx.size()) ? g->ctx[(size_t)member] : nullptr; }
extern "C" void blsgpu_group_bases_free(blsgpu_group_bases* b) {
  if (!b) return;
  for (auto p : b->part) blsgpu_bases_free(p);
  delete b;
}
extern "C" size_t blsgpu_group_bases_len(const blsgpu_group_bases* b) { return b ? b->n : 0; }
// member k keeps points [lo_k, hi_k) resident on its device
extern "C" int blsgpu_group_bases_upload(blsgpu_group* g, int group, const uint64_t* xy, const uint8_t* inf, size_t n, blsgpu_group_bases** out) {
  if (!g || !out || (n && !xy) || (group != 1 && group != 2)) return bad("group_bases_upload: bad argument");
  blsgpu_group_bases* b = new blsgpu_group_bases();
  b->group = group; b->n = n; b->part.assign(g->ctx.size(), nullptr);
  const size_t w = g->ctx.size(), words = group == 1 ? 12 : 24;
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(n, k, w, lo, hi);
    return group == 1 ? blsgpu_g1_bases_upload(c, xy + lo * words, inf ? inf + lo : nullptr, hi - lo, &b->part[k])
                      : blsgpu_g2_bases_upload(c, xy + lo * words, inf ? inf + lo : nullptr, hi - lo, &b->part[k]);
  });
  if (rc) { const std::string keep = g_err; blsgpu_group_bases_free(b); g_err = keep; return rc; }
  *out = b;
  return BLSGPU_OK;
}
extern "C" int blsgpu_group_bases_from_scalars(blsgpu_group* g, int group, const uint8_t* scalars, size_t n, blsgpu_group_bases** out) {
  if (!g || !out || (n && !scalars) || (group != 1 && group != 2)) return bad("group_bases_from_scalars: bad argument");
  blsgpu_group_bases* b = new blsgpu_group_bases();
  b->group = group; b->n = n; b->part.assign(g->ctx.size(), nullptr);
  const size_t w = g->ctx.size();
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(n, k, w, lo, hi);
    return blsgpu_bases_from_scalars(c, group, scalars + lo * 32, hi - lo, &b->part[k]);
  });
  if (rc) { const std::string keep = g_err; blsgpu_group_bases_free(b); g_err = keep; return rc; }
  *out = b;
  return BLSGPU_OK;
}
// sum_{i < n} scalars[i] * bases[i]: member k multiplies the part of [0, n) that lies in ITS resident slice, member 0 folds the w partial sums
template <int G>
static int msm_sharded(blsgpu_group* g, const blsgpu_group_bases* b, const uint8_t* scalars, size_t n, uint64_t* out) {
  constexpr size_t PW = G == 1 ? 18 : 36;
  if (!g || !b || !out || (n && !scalars)) return bad("msm_sharded: NULL argument");
  if (b->group != G) return bad("msm_sharded: bases belong to the other group");
  if (b->part.size() != g->ctx.size()) return bad("msm_sharded: the bases were sharded over another group");
  if (n > b->n) return bad("msm_sharded: more scalars than resident bases");
  const size_t w = g->ctx.size();
  std::vector<uint64_t> parts(w * PW);
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(b->n, k, w, lo, hi);
    const size_t end = hi < n ? hi : n, cnt = end > lo ? end - lo : 0;
    const uint8_t* s = scalars + (cnt ? lo * 32 : 0);
    return G == 1 ? blsgpu_g1_msm(c, b->part[k], 0, s, cnt, parts.data() + k * PW) : blsgpu_g2_msm(c, b->part[k], 0, s, cnt, parts.data() + k * PW);
  });
  if (rc) return rc;
  return G == 1 ? blsgpu_g1_sum(g->ctx[0], parts.data(), w, out) : blsgpu_g2_sum(g->ctx[0], parts.data(), w, out);
}
extern "C" int blsgpu_g1_msm_sharded(blsgpu_group* g, const blsgpu_group_bases* b, const uint8_t* scalars, size_t n, uint64_t out[18]) { return msm_sharded<1>(g, b, scalars, n, out); }
extern "C" int blsgpu_g2_msm_sharded(blsgpu_group* g, const blsgpu_group_bases* b, const uint8_t* scalars, size_t n, uint64_t out[36]) { return msm_sharded<2>(g, b, scalars, n, out); }
// Device-pointer form: member k multiplies ITS WHOLE resident slice by the scalars at d_scalars[k] (on its device) and writes its partial
// sum (projective wire form) to d_partials[k] (on its device); every member only ENQUEUES (with pipelining on -- blsgpu_group_set_pipelining --
// up to four calls per member are in flight), nothing is synchronised.  blsgpu_g{1,2}_partials_fold collects and adds the partial sums.
template <int G>
static int msm_sharded_device(blsgpu_group* g, const blsgpu_group_bases* b, const void* const* d_scalars, void* const* d_partials) {
  if (!g || !b || !d_scalars || !d_partials) return bad("msm_sharded_device: NULL argument");
  if (b->group != G) return bad("msm_sharded_device: bases belong to the other group");
  if (b->part.size() != g->ctx.size()) return bad("msm_sharded_device: the bases were sharded over another group");
  return group_run(g, [&](size_t k, blsgpu_ctx* c) {
    CTX_CLAIM(c);
    HIPCHK(hipSetDevice(c->device));
    for (auto& fr : g->fold_reads[k]) if (fr.used && fr.ptr == d_partials[k]) HIPCHK(hipStreamWaitEvent(c->stream, fr.ev, 0));     // a fold still owns this buffer
    const size_t cnt = blsgpu_bases_len(b->part[k]);
    return G == 1 ? blsgpu_g1_msm_device(c, b->part[k], 0, d_scalars[k], cnt, d_partials[k]) : blsgpu_g2_msm_device(c, b->part[k], 0, d_scalars[k], cnt, d_partials[k]);
  });
}
extern "C" int blsgpu_g1_msm_sharded_device(blsgpu_group* g, const blsgpu_group_bases* b, const void* const* d_scalars, void* const* d_partials) { return msm_sharded_device<1>(g, b, d_scalars, d_partials); }
extern "C" int blsgpu_g2_msm_sharded_device(blsgpu_group* g, const blsgpu_group_bases* b, const void* const* d_scalars, void* const* d_partials) { return msm_sharded_device<2>(g, b, d_scalars, d_partials); }
// out = sum_k partial_k: every member waits (on ITS stream) for its MSMs except the `lag` most recent ones, copies its partial sum into
// pinned host memory and synchronises that stream only -- the calls still in flight run on the member's internal streams and are not
// held up -- then member 0 adds the w points (`Sum for G1Projective`, g1.rs:161-171).  lag = 0 collects the most recent call.
template <int G>
static int partials_fold(blsgpu_group* g, const void* const* d_partials, int lag, uint64_t* out) {
  constexpr size_t PW = G == 1 ? 18 : 36;
  if (!g || !d_partials || !out || lag < 0) return bad("partials_fold: bad argument");
  const size_t w = g->ctx.size();
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    CTX_CLAIM(c);
    int r = blsgpu_join_lag(c, lag);
    if (r) return r;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(g->pinned[k], d_partials[k], PW * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return (int)BLSGPU_OK;
  });
  if (rc) return rc;
  std::vector<uint64_t> parts(w * PW);
  for (size_t k = 0; k < w; k++) memcpy(parts.data() + k * PW, g->pinned[k], PW * 8);
  return G == 1 ? blsgpu_g1_sum(g->ctx[0], parts.data(), w, out) : blsgpu_g2_sum(g->ctx[0], parts.data(), w, out);
}
extern "C" int blsgpu_g1_partials_fold(blsgpu_group* g, const void* const* d_partials, int lag, uint64_t out[18]) { return partials_fold<1>(g, d_partials, lag, out); }
extern "C" int blsgpu_g2_partials_fold(blsgpu_group* g, const void* const* d_partials, int lag, uint64_t out[36]) { return partials_fold<2>(g, d_partials, lag, out); }
// The same fold without a host round trip: every member queues (behind its MSMs except the `lag` most recent ones) a copy of its partial
// sum into a staging row on member 0's device; member 0's stream waits for the w copies and adds them into d_out (device memory of member
// 0, projective wire form).  Nothing is synchronised: the fold of MSM i - 2 runs under the accumulation of MSMs i - 1 and i.
template <int G>
static int partials_fold_device(blsgpu_group* g, const void* const* d_partials, int lag, void* d_out, int final_exp = 0) {
  constexpr size_t PB = G == 1 ? 144 : G == 2 ? 288 : 576;
  if (!g || !d_partials || !d_out || lag < 0) return bad("partials_fold_device: bad argument");
  const size_t w = g->ctx.size();
  const unsigned seq = g->fold_seq++;
  const unsigned row = seq & 3u;
  uint8_t* stage = (uint8_t*)g->fold_in + (size_t)row * w * 576;
  blsgpu_ctx* c0 = g->ctx[0];
  // everything below runs on the members' FOLD streams: the context's own stream stays empty, so the front of the next MSM (which waits
  // for whatever is queued on that stream when it is launched) never waits for a fold -- with the fold on the main stream an MSM's front
  // queued behind the copy / sum kernels, which in turn wait for a free CU slot under the running accumulation
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    CTX_CLAIM(c);
    HIPCHK(hipSetDevice(c->device));
    if (!c->fold_stream) HIPCHK(hipStreamCreateWithFlags(&c->fold_stream, hipStreamNonBlocking));
    hipStream_t keep = c->stream;
    c->stream = c->fold_stream;
    int r = blsgpu_join_lag(c, lag);
    c->stream = keep;
    if (r) return r;
    // ... and for whatever the caller queued on the context's own stream (the producers of Fp12 partials run there; for pipelined MSMs that
    // stream is empty and the event is complete at once)
    HIPCHK(hipEventRecord(g->ev_main[k], keep));
    HIPCHK(hipStreamWaitEvent(c->fold_stream, g->ev_main[k], 0));
    if (g->ev_sum_used[row]) HIPCHK(hipStreamWaitEvent(c->fold_stream, g->ev_sum[row], 0));       // the sum that last read this staging row
    // (a plain device-to-device copy for members on member 0's device: the peer form need not be asynchronous there)
    if (c->device == c0->device) HIPCHK(hipMemcpyAsync(w == 1 ? d_out : (void*)(stage + k * PB), d_partials[k], PB, hipMemcpyDeviceToDevice, c->fold_stream));
    else HIPCHK(hipMemcpyPeerAsync(stage + k * PB, c0->device, d_partials[k], c->device, PB, c->fold_stream));
    HIPCHK(hipEventRecord(g->ev_copy[k], c->fold_stream));
    blsgpu_group::FoldRead& fr = g->fold_reads[k][seq & 7u];
    HIPCHK(hipEventRecord(fr.ev, c->fold_stream));
    fr.ptr = d_partials[k]; fr.used = true;
    return (int)BLSGPU_OK;
  });
  if (rc) return rc;
  if (w == 1 && !(G == 12 && final_exp)) return BLSGPU_OK;      // one member: its partial result IS the result (copied straight to d_out above)
  CTX_CLAIM(c0);
  HIPCHK(hipSetDevice(c0->device));
  for (size_t k = 1; k < w; k++) HIPCHK(hipStreamWaitEvent(c0->fold_stream, g->ev_copy[k], 0));
  hipStream_t keep = c0->stream;
  c0->stream = c0->fold_stream;
  c0->on_fold_stream = true;
  if (G == 12) {
    // `MillerLoopResult + MillerLoopResult` over the members' partial products (pairings.rs:179-186), then -- if asked -- the ONE final exponentiation
    rc = w == 1 ? BLSGPU_OK : blsgpu_fp12_product_device(c0, stage, w, d_out);
    if (!rc && final_exp) rc = blsgpu_final_exponentiation_device(c0, d_out, 1, d_out);
  } else {
    rc = G == 1 ? blsgpu_g1_sum_device(c0, stage, w, d_out) : blsgpu_g2_sum_device(c0, stage, w, d_out);
  }
  c0->stream = keep;
  c0->on_fold_stream = false;
  if (rc) return rc;
  HIPCHK(hipEventRecord(g->ev_sum[row], c0->fold_stream));
  g->ev_sum_used[row] = true;
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_partials_fold_device(blsgpu_group* g, const void* const* d_partials, int lag, void* d_out) { return partials_fold_device<1>(g, d_partials, lag, d_out); }
extern "C" int blsgpu_g2_partials_fold_device(blsgpu_group* g, const void* const* d_partials, int lag, void* d_out) { return partials_fold_device<2>(g, d_partials, lag, d_out); }
extern "C" int blsgpu_fp12_partials_fold_device(blsgpu_group* g, const void* const* d_partials, int final_exp, void* d_out) { return partials_fold_device<12>(g, d_partials, 0, d_out, final_exp); }
// Device-pointer, asynchronous forms of the pairing entry points: member k works on ITS arrays (device pointers in its memory, counts[k] items) and
// only enqueues.  mode 0: out[k][i] = pairing, 1: raw Miller values (nothing to fold: the outputs stay sharded); mode 2: d_out[k] = the member-local
// product of its Miller values (one Fp12 value, 576 B), to be folded by blsgpu_fp12_partials_fold_device
extern "C" int blsgpu_pairings_sharded_device(blsgpu_group* g, int mode, const void* const* d_g1, const void* const* d_g1inf, const void* const* d_g2, const void* const* d_g2inf,
                                              const size_t* counts, void* const* d_out) {
  if (!g || !d_g1 || !d_g2 || !counts || !d_out || mode < 0 || mode > 2) return bad("pairings_sharded_device: bad argument");
  return group_run(g, [&](size_t k, blsgpu_ctx* c) {
    const void* f1 = d_g1inf ? d_g1inf[k] : nullptr; const void* f2 = d_g2inf ? d_g2inf[k] : nullptr;
    if (mode == 0) return blsgpu_pairing_batch_device(c, d_g1[k], f1, d_g2[k], f2, counts[k], d_out[k]);
    if (mode == 1) return blsgpu_miller_loop_batch_device(c, d_g1[k], f1, d_g2[k], f2, counts[k], d_out[k]);
    return blsgpu_multi_miller_loop_device(c, d_g1[k], f1, d_g2[k], f2, counts[k], d_out[k]);
  });
}
extern "C" int blsgpu_group_set_pipelining(blsgpu_group* g, int on) {
  if (!g) return bad("group_set_pipelining: NULL group");
  for (auto c : g->ctx) { int rc = blsgpu_set_pipelining(c, on); if (rc) return rc; }
  return BLSGPU_OK;
}
// wait for everything queued on every member; the first failing member's verdict (e.g. a non-canonical scalar of an asynchronous call) wins
extern "C" int blsgpu_group_synchronize(blsgpu_group* g) {
  if (!g) return bad("group_synchronize: NULL group");
  return group_run(g, [&](size_t, blsgpu_ctx* c) { return blsgpu_synchronize(c); });
}
// n independent pairings (mode 0) or raw Miller values (mode 1): index slices, every member writes its slice of `out`; no fold
static int pairings_sharded(blsgpu_group* g, int mode, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) {
  if (!g || (n && (!g1 || !g2 || !out))) return bad("pairing_batch_sharded: NULL argument");
  const size_t w = g->ctx.size();
  return group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(n, k, w, lo, hi);
    if (lo == hi) return (int)BLSGPU_OK;
    return mode == 0 ? blsgpu_pairing_batch(c, g1 + lo * 12, g1inf ? g1inf + lo : nullptr, g2 + lo * 24, g2inf ? g2inf + lo : nullptr, hi - lo, out + lo * 72)
                     : blsgpu_miller_loop_batch(c, g1 + lo * 12, g1inf ? g1inf + lo : nullptr, g2 + lo * 24, g2inf ? g2inf + lo : nullptr, hi - lo, out + lo * 72);
  });
}
extern "C" int blsgpu_pairing_batch_sharded(blsgpu_group* g, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) {
  return pairings_sharded(g, 0, g1, g1inf, g2, g2inf, n, out);
}
extern "C" int blsgpu_miller_loop_batch_sharded(blsgpu_group* g, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, uint64_t* out) {
  return pairings_sharded(g, 1, g1, g1inf, g2, g2inf, n, out);
}
// prod_i ML(g1[i], g2[i]): member-local products of index slices, folded by member 0; final_exp != 0: followed by ONE final exponentiation
extern "C" int blsgpu_multi_miller_loop_sharded(blsgpu_group* g, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, size_t n, int final_exp,
                                                uint64_t out[72]) {
  if (!g || !out || (n && (!g1 || !g2))) return bad("multi_miller_loop_sharded: NULL argument");
  const size_t w = g->ctx.size();
  std::vector<uint64_t> parts(w * 72);
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(n, k, w, lo, hi);
    return blsgpu_multi_miller_loop(c, n ? g1 + lo * 12 : g1, g1inf ? g1inf + lo : nullptr, n ? g2 + lo * 24 : g2, g2inf ? g2inf + lo : nullptr, hi - lo, parts.data() + k * 72);
  });
  if (rc) return rc;
  if (!final_exp) return blsgpu_fp12_product(g->ctx[0], parts.data(), w, out);
  uint64_t f[72];
  rc = blsgpu_fp12_product(g->ctx[0], parts.data(), w, f);
  if (rc) return rc;
  return blsgpu_final_exponentiation_batch(g->ctx[0], f, 1, out);
}
// `G2Prepared` tables for a group: the same m points prepared on EVERY member (a verification key is small: 26 KB per point), so that
// the prepared Miller loops shard exactly like the unprepared ones
struct blsgpu_group_g2_prepared { std::vector<blsgpu_g2_prepared*> part; size_t n = 0; };
extern "C" void blsgpu_group_g2_prepared_free(blsgpu_group_g2_prepared* p) {
  if (!p) return;
  for (auto t : p->part) blsgpu_g2_prepared_free(t);
  delete p;
}
extern "C" size_t blsgpu_group_g2_prepared_len(const blsgpu_group_g2_prepared* p) { return p ? p->n : 0; }
extern "C" int blsgpu_group_g2_prepare(blsgpu_group* g, const uint64_t* g2, const uint8_t* inf, size_t m, blsgpu_group_g2_prepared** out) {
  if (!g || !out || (m && !g2)) return bad("group_g2_prepare: NULL argument");
  blsgpu_group_g2_prepared* p = new blsgpu_group_g2_prepared();
  p->n = m; p->part.assign(g->ctx.size(), nullptr);
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) { return blsgpu_g2_prepare(c, g2, inf, m, &p->part[k]); });
  if (rc) { const std::string keep = g_err; blsgpu_group_g2_prepared_free(p); g_err = keep; return rc; }
  *out = p;
  return BLSGPU_OK;
}
// `multi_miller_loop` over n terms, prepared or not: member-local products of index slices, folded by member 0, ONE final exponentiation
extern "C" int blsgpu_multi_miller_loop_prepared_sharded(blsgpu_group* g, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint32_t* qidx,
                                                         const blsgpu_group_g2_prepared* p, size_t n, int final_exp, uint64_t out[72]) {
  if (!g || !out || (n && !g1)) return bad("multi_miller_loop_prepared_sharded: NULL argument");
  if (p && p->part.size() != g->ctx.size()) return bad("multi_miller_loop_prepared_sharded: the table was prepared for another group");
  const size_t w = g->ctx.size();
  std::vector<uint64_t> parts(w * 72);
  int rc = group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(n, k, w, lo, hi);
    return blsgpu_multi_miller_loop_prepared(c, n ? g1 + lo * 12 : g1, g1inf ? g1inf + lo : nullptr, g2 ? g2 + lo * 24 : g2, (g2 && g2inf) ? g2inf + lo : nullptr,
                                             qidx ? qidx + lo : nullptr, p ? p->part[k] : nullptr, hi - lo, parts.data() + k * 72);
  });
  if (rc) return rc;
  if (!final_exp) return blsgpu_fp12_product(g->ctx[0], parts.data(), w, out);
  uint64_t f[72];
  rc = blsgpu_fp12_product(g->ctx[0], parts.data(), w, f);
  if (rc) return rc;
  return blsgpu_final_exponentiation_batch(g->ctx[0], f, 1, out);
}
// blsgpu_multi_miller_loop_prepared_many with the SEGMENTS dealt to the members in contiguous slices
extern "C" int blsgpu_multi_miller_loop_prepared_many_sharded(blsgpu_group* g, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint32_t* qidx,
                                                              const blsgpu_group_g2_prepared* p, const uint64_t* offsets, size_t nseg, int final_exp, uint64_t* out) {
  if (!g || (nseg && (!offsets || !out))) return bad("multi_miller_loop_prepared_many_sharded: NULL argument");
  if (!nseg) return BLSGPU_OK;
  if (p && p->part.size() != g->ctx.size()) return bad("multi_miller_loop_prepared_many_sharded: the table was prepared for another group");
  if (offsets[0] != 0) return bad("multi_miller_loop_prepared_many: offsets[0] must be 0");
  for (size_t i = 0; i < nseg; i++) if (offsets[i] > offsets[i + 1]) return bad("multi_miller_loop_prepared_many: offsets must be non-decreasing");
  if (offsets[nseg] && !g1) return bad("multi_miller_loop_prepared_many_sharded: NULL argument");
  const size_t w = g->ctx.size();
  return group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(nseg, k, w, lo, hi);
    if (lo == hi) return (int)BLSGPU_OK;
    const size_t t0 = (size_t)offsets[lo];
    std::vector<uint64_t> off(hi - lo + 1);
    for (size_t i = lo; i <= hi; i++) off[i - lo] = offsets[i] - t0;
    return blsgpu_multi_miller_loop_prepared_many(c, g1 ? g1 + t0 * 12 : g1, g1inf ? g1inf + t0 : nullptr, g2 ? g2 + t0 * 24 : g2, (g2 && g2inf) ? g2inf + t0 : nullptr,
                                                  qidx ? qidx + t0 : nullptr, p ? p->part[k] : nullptr, off.data(), hi - lo, final_exp, out + lo * 72);
  });
}
// N independent multi_miller_loops (blsgpu_multi_miller_loop_many): the SEGMENTS are dealt in contiguous slices, nothing to fold
extern "C" int blsgpu_multi_miller_loop_many_sharded(blsgpu_group* g, const uint64_t* g1, const uint8_t* g1inf, const uint64_t* g2, const uint8_t* g2inf, const uint64_t* offsets,
                                                     size_t nseg, int final_exp, uint64_t* out) {
  if (!g || (nseg && (!offsets || !out))) return bad("multi_miller_loop_many_sharded: NULL argument");
  if (!nseg) return BLSGPU_OK;
  if (offsets[0] != 0) return bad("multi_miller_loop_many: offsets[0] must be 0");
  for (size_t i = 0; i < nseg; i++) if (offsets[i] > offsets[i + 1]) return bad("multi_miller_loop_many: offsets must be non-decreasing");
  if (offsets[nseg] && (!g1 || !g2)) return bad("multi_miller_loop_many_sharded: NULL argument");
  const size_t w = g->ctx.size();
  return group_run(g, [&](size_t k, blsgpu_ctx* c) {
    size_t lo, hi; group_range(nseg, k, w, lo, hi);
    if (lo == hi) return (int)BLSGPU_OK;
    const size_t t0 = (size_t)offsets[lo];
    std::vector<uint64_t> off(hi - lo + 1);
    for (size_t i = lo; i <= hi; i++) off[i - lo] = offsets[i] - t0;
    return blsgpu_multi_miller_loop_many(c, g1 ? g1 + t0 * 12 : g1, g1inf ? g1inf + t0 : nullptr, g2 ? g2 + t0 * 24 : g2, g2inf ? g2inf + t0 : nullptr, off.data(), hi - lo, final_exp,
                                         out + lo * 72);
  });
}
