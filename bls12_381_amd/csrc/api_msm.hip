// api_msm.hip -- resident bases, multi-scalar multiplication, batched variable-base multiplication, sums and batch_normalize.
#define BLS_TU_NAME "api_msm.hip"
#include "host.h"
#include "abi_kernels.hip.h"
#include "mulbatch.hip.h"

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
    bad_alloc |= g->heavy.reserve(2 * nb * sizeof(uint4));            // the heavy list, then the indices of its block-folded entries
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
  const u32 heavy_small_blocks = (u32)nblk(gnb, 256);
  KLAUNCH(k_msm_heavy<F>, dim3(heavy_small_blocks + HEAVY_BIG_BLOCKS), dim3(256), 0, gas, gs.heavy.as<uint4>(), ctrl, records, (u32)gnb, heavy_small_blocks);
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

#ifdef BLS_ACC_TRACE
// diagnostic build only: where the accumulation kernel writes its per-wavefront trace (4 words per wavefront; nullptr = off)
extern "C" int blsgpu_diag_set_acc_trace(void* device_words) {
  unsigned long long* p = (unsigned long long*)device_words;
  return hipMemcpyToSymbol(HIP_SYMBOL(bls::g_acc_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
// sequence mode: 1 = every wavefront of every following launch appends its record (up to 2^18); 0 = back to one record per wavefront index
extern "C" int blsgpu_diag_set_acc_trace_seq(unsigned int on) {
  return hipMemcpyToSymbol(HIP_SYMBOL(bls::g_acc_seq), &on, sizeof(on)) == hipSuccess ? 0 : 1;
}
#endif
