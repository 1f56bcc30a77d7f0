// api_group.hip -- device groups: the hot path sharded over the GPUs of ONE node from ONE process (host code only: every member is
// driven through the public entry points of its own context).
#define BLS_TU_NAME "api_group.hip"
#include "host.h"

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

