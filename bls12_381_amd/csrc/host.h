// host.h -- what the host-side translation units of libblsgpu.so share: the context and handle structures, error reporting, the
// per-kernel timing wrapper, and the declarations of the few functions that cross translation units.  No kernels here.
//
//   api_ctx.hip      contexts, streams, status, diagnostics, staged uploads, field / point self-test hooks
//   api_msm.hip      resident bases, MSM, batched scalar multiplication, sums, batch_normalize        (msm.hip.h, mulbatch.hip.h)
//   api_pairing.hip  pairings, Miller loops, G2Prepared, final exponentiation, Gt, Fp6 / Fp12 hooks   (pairing / quad / prep / wide)
//   api_aux.hip      Fr vectors and transform, hash-to-curve, point codecs, bulk BLS verification      (fr / h2c / codec)
//   api_group.hip    device groups: one process driving several GPUs (host code only)
//
// Every kernel header is compiled into exactly ONE of them (a non-template __global__ function has one host-side stub per library).
#pragma once
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
#include <dlfcn.h>

#include "../../include/bls12_381_hip.h"
#include "convert.hip.h"
#include "scalar.hip.h"
#include "limits.h"
#include "diag.h"

using namespace bls;

inline thread_local std::string g_err;
static inline int fail_at(const char* what, hipError_t e, int line, const char* file) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_err = buf;
  return BLSGPU_ERR_HIP;
}
#define fail(what, e, line) fail_at(what, e, line, BLS_TU_NAME)
static inline int bad(const char* what) { g_err = what; return BLSGPU_ERR_ARG; }
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
inline thread_local std::atomic<size_t> g_no_ctx_owner{0};
inline thread_local int g_no_ctx_depth = 0;
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
inline thread_local KTimer* g_kt = nullptr;
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
  bool pin_busy[8] = {};                // ... whose last DMA may still be in flight (waited for where the buffer is needed again)
  DevBuf gt_one; bool gt_one_ready = false; hipEvent_t ev_gt_one = nullptr;      // the wire form of Fp12::one() (blsgpu_gt_is_identity_device, bulk verification)
  DevBuf ver;                           // bulk verification (blsgpu_bls_verify_batch): every intermediate of the chain
  blsgpu_g2_prepared* ver_table = nullptr;   // ... and the resident `G2Prepared` of -G2 for mode 1
  bool ver_consts_ready = false; hipEvent_t ev_ver = nullptr;
  hipStream_t ver_stream[2] = {nullptr, nullptr}; hipEvent_t ev_ver_side[3] = {};     // the independent stages of the chain run side by side
  DevBuf h2c_uniform;                   // uniform bytes between k_expand_message and the kernels that consume them (expand.hip.h)
  DevBuf fb_stage;                      // staging of the one-byte scalars the tables are built from
  DevBuf fb_table[2];                   // fixed-base comb tables of the generators (k_fixed_base): 32 x 256 affine records each, built at first use
  hipEvent_t ev_fb[2] = {};             // recorded where a table was built; awaited by every user (the caller may switch streams)
  bool fb_ready[2] = {false, false};
  DevBuf fr_tw[2], fr_tmp, fr_ninv;     // Fr transform: twiddle tables (forward / inverse), permutation target, n^-1
  int fr_tw_log[2] = {-1, -1};
  int fr_ninv_log = -1;
  BlsDiag diag;                         // the environment switches as they were when the context was created (diag.h)
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

template <class F> struct GroupTag;
template <> struct GroupTag<FpPolicy> { static constexpr int id = 1; };
template <> struct GroupTag<Fp2Policy> { static constexpr int id = 2; };

static inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

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

// G2Prepared resident on the device (api_pairing.hip; the bulk verification of api_aux.hip keeps one)
struct blsgpu_g2_prepared { int device = 0; size_t n = 0; u32* tab = nullptr; uint8_t* inf = nullptr; hipEvent_t ev_ready = nullptr; };

// ---- functions that cross translation units -------------------------------------------------------------------------------------
int staged_upload(blsgpu_ctx* c, void* dst, const void* src, size_t bytes);      // api_ctx.hip
void acc_harvest(blsgpu_ctx* c, bool wait);                                       // api_ctx.hip (MSM accumulation timings)

// host side of the element-wise self-test hooks (blsgpu_fp_op .. blsgpu_fp12_op): stage, launch, fetch
template <class Launch>
static int elem_op_run(blsgpu_ctx* c, int words, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, Launch launch) {
  if (!c || (n && (!a || !out))) return bad("op: NULL argument");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  size_t bytes = n * words * 4;
  if (c->io_a.reserve(bytes) || c->io_b.reserve(bytes) || c->io_out.reserve(bytes)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, a, bytes, hipMemcpyHostToDevice, c->stream));
  if (b) HIPCHK(hipMemcpyAsync(c->io_b.p, b, bytes, hipMemcpyHostToDevice, c->stream));
  launch(c->io_a.as<u32>(), b ? c->io_b.as<u32>() : (const u32*)nullptr, c->io_out.as<u32>());
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
