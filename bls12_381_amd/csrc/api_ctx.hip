// api_ctx.hip -- contexts, streams, status words, diagnostics, staged uploads, and the field / point self-test hooks.
#define BLS_TU_NAME "api_ctx.hip"
#include "host.h"
#include "generators.hip.h"

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


// Pageable host memory -> device.  hipMemcpyAsync from pageable memory is staged by the runtime on ONE host thread (measured here:
// ~4 GB/s, 7.6 ms for the 32 MB of scalars of a 2^20-point MSM -- twice the MSM itself); large uploads of the host-pointer entry points
// therefore go through pinned bounce buffers of the context, the host-side copy split over up to four threads, every chunk's DMA queued
// on the context's stream as soon as it is staged.  On return everything is queued on the stream (the source may be reused at once).
constexpr size_t STAGE_CHUNK = (size_t)2 << 20;
constexpr int STAGE_THREADS = 4;
int staged_upload(blsgpu_ctx* c, void* dst, const void* src, size_t bytes) {
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
  // the bounce buffers may still be read by the DMAs of the PREVIOUS staged upload: waited for here, where the buffers are needed again,
  // not at the end of that upload -- so the host is not held until everything queued before that upload has drained, and the upload of
  // call i + 1 overlaps the kernels of call i (the source is copied out by the time this function returns either way)
  for (int k = 0; k < 2 * STAGE_THREADS; k++)
    if (c->pin_busy[k]) { if (hipEventSynchronize(c->pin_ev[k]) != hipSuccess) failed = 1; c->pin_busy[k] = false; }
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
  for (int k = 0; k < 2 * T; k++) c->pin_busy[k] = true;            // (waited for by the next staged upload, blsgpu_synchronize or blsgpu_destroy)
  if (failed) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(c->stream);          // DMAs already queued still read the bounce buffers: a retry must not overwrite them
    (void)hipGetLastError();
    for (auto& b : c->pin_busy) b = false;
    g_err = "staged upload failed"; return BLSGPU_ERR_HIP;
  }
  return BLSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------

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
  int pr[3];
  for (int i = 0; i < 3; i++) pr[i] = c->diag.prio[i] == 'h' ? prio_hi : c->diag.prio[i] == 'l' ? prio_lo : (prio_lo + prio_hi) / 2;
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
  // the A/B and test switches of the environment, read here and nowhere else (diag.h)
  c->diag = diag_read(MMLP_MAX_K, ITEM_CAP_MAX, FR_COLS_LOG_MAX);
  if (c->diag.pairing_layout < 0) { delete c; return bad("blsgpu_create: BLSGPU_PAIRING_LAYOUT must be one of auto, pair, quad, wide"); }
  c->force_slow_sort = c->diag.force_slow_sort; c->no_glv = c->diag.no_glv; c->pairing_layout = c->diag.pairing_layout;
  c->mmlp_k = c->diag.mmlp_k; c->mml_impl = c->diag.mml_impl; c->h2c_split = c->diag.h2c_split; c->fr_cols_want = c->diag.fr_cols_want; c->item_cap = c->diag.item_cap;
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
  DevBuf* bufs[] = {&c->result, &c->io_a, &c->io_b, &c->io_c, &c->io_d, &c->io_e, &c->io_f, &c->io_out, &c->flags_a, &c->flags_b, &c->fr_tw[0], &c->fr_tw[1], &c->fr_tmp, &c->fr_ninv, &c->fb_table[0], &c->fb_table[1], &c->fb_stage, &c->mmlp_work, &c->mmlp_out, &c->gt_one, &c->ver, &c->fold_c, &c->fold_d, &c->fold_result, &c->h2c_uniform};
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
void acc_harvest(blsgpu_ctx* c, bool wait) {
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
// self-test hooks
// ---------------------------------------------------------------------------------------------------
static int elem_op(blsgpu_ctx* c, int words, int kind, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  return elem_op_run(c, words, a, b, n, out, [&](const u32* x, const u32* y, u32* o) {
    if (kind == 1) KLAUNCH(k_fp_op, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, x, y, o, n);
    else KLAUNCH(k_fp2_op, dim3(nblk(n, 256)), dim3(256), 0, c->stream, op, x, y, o, n);
  });
}
extern "C" int blsgpu_fp_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (op < 0 || op > 6) return bad("fp_op: unknown op");
  return elem_op(c, 12, 1, op, a, b, n, out);
}
extern "C" int blsgpu_fp2_op(blsgpu_ctx* c, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) { CTX_CLAIM(c);
  if (op < 0 || op > 6) return bad("fp2_op: unknown op");
  return elem_op(c, 24, 2, op, a, b, n, out);
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

