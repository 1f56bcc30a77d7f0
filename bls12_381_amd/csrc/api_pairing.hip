// api_pairing.hip -- pairings, Miller loops, G2Prepared, final exponentiation, Gt, and the Fp6 / Fp12 self-test hooks.
#define BLS_TU_NAME "api_pairing.hip"
#include "host.h"
#include "pairing.hip.h"
#include "quad.hip.h"
#include "prep.hip.h"
#include "wide.hip.h"

using namespace bls;

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
  if (!c->diag.wide_prog.empty()) path = c->diag.wide_prog;
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

// ---- Fp6 / Fp12 self-test hooks (the kernels live with the pairing code) ------------------------------------------------------------
static int elem_op(blsgpu_ctx* c, int words, int kind, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  return elem_op_run(c, words, a, b, n, out, [&](const u32* x, const u32* y, u32* o) {
    if (kind == 6) KLAUNCH(k_fp6_op, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, op, x, y, o, n);
    else if (c->pairing_layout != 2 && op != 3) KLAUNCH(k_fp12_op_quad, dim3(nblk(n * QL, QUAD_BLOCK)), dim3(QUAD_BLOCK), 0, c->stream, op, x, y, o, n);
    else KLAUNCH(k_fp12_op, dim3(nblk(n * PL, PAIRING_BLOCK)), dim3(PAIRING_BLOCK), 0, c->stream, op, x, y, o, n);
  });
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
