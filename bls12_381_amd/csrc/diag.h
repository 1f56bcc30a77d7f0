// diag.h -- every environment switch of libblsgpu.so, in one place.  All of them are A/B or test hooks (INTEGRATION.md lists them for
// users); none is needed in production.  They are read ONCE, when a context is created (diag_read, called by blsgpu_create) -- nothing
// on a call path consults the environment.
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <string>

struct BlsDiag {
  bool force_slow_sort = false;     // BLSGPU_FORCE_SLOW_SORT        the global-atomic sort used beyond 2^24 points, at every size (tests)
  bool no_glv = false;              // BLSGPU_NO_GLV                 plain 256-bit windows: no endomorphism split for G1 / G2 MSMs and mul_batch
  int pairing_layout = 0;           // BLSGPU_PAIRING_LAYOUT         auto (0) | pair (2) | quad (4) | wide (256); -1 = unknown name (create fails)
  int mmlp_k = 0;                   // BLSGPU_MMLP_K                 terms per accumulator of ONE long prepared product (0 = automatic)
  int mml_impl = 0;                 // BLSGPU_MML_IMPL               kernel behind multi_miller_loop_device with K > 1: 1 shared lane pairs, 4 prepared-path quads
  int h2c_split = -1;               // BLSGPU_H2C_SPLIT              hash-to-curve on two lane groups per message: -1 by batch size, 0 never, 1 always
  bool verify_h2c_split = false;    // BLSGPU_VERIFY_H2C_SPLIT       let the batch-size rule apply inside bulk verification too (default: plain form there)
  int fr_cols_want = 1;             // BLSGPU_NTT_IMPL=stage|cols    column-tile passes of the transform: 0 never, 1 from 2^20 elements, 2 always
  int ntt_cols[3] = {0, 0, 0};      // BLSGPU_NTT_COLS=t,s,l         tile log2, stages per pass, lanes per workgroup of those passes (0 = built-in 11,7,512)
  unsigned item_cap = 0;            // BLSGPU_ITEM_CAP               entries per work item of the bucket accumulation (0 = automatic)
  char prio[4] = "nhl";             // BLSGPU_PRIO                   stream priorities of accumulation / tail / front: h, n or l each
  std::string wide_prog;            // BLSGPU_WIDE_PROG              path of wide_prog.bin (default: next to the library)
};

// limits come from the caller so that this header needs no kernel header
static inline BlsDiag diag_read(int mmlp_max_k, int item_cap_max, int cols_log_max) {
  BlsDiag d;
  d.force_slow_sort = getenv("BLSGPU_FORCE_SLOW_SORT") != nullptr;
  d.no_glv = getenv("BLSGPU_NO_GLV") != nullptr;
  if (const char* v = getenv("BLSGPU_PAIRING_LAYOUT")) {
    // exact names only: a typo must not silently select the slowest kernels
    const std::string s(v);
    if (s == "auto" || s == "0" || s.empty()) d.pairing_layout = 0;
    else if (s == "pair" || s == "2") d.pairing_layout = 2;
    else if (s == "quad" || s == "4") d.pairing_layout = 4;
    else if (s == "wide" || s == "256") d.pairing_layout = 256;
    else d.pairing_layout = -1;
  }
  if (const char* v = getenv("BLSGPU_MMLP_K")) { long k = atol(v); if (k >= 1 && k <= mmlp_max_k) d.mmlp_k = (int)k; }
  if (const char* v = getenv("BLSGPU_MML_IMPL")) { long k = atol(v); if (k == 1 || k == 4) d.mml_impl = (int)k; }
  if (const char* v = getenv("BLSGPU_H2C_SPLIT")) d.h2c_split = atoi(v) ? 1 : 0;
  d.verify_h2c_split = getenv("BLSGPU_VERIFY_H2C_SPLIT") != nullptr;
  if (const char* v = getenv("BLSGPU_NTT_IMPL")) d.fr_cols_want = !strcmp(v, "cols") ? 2 : !strcmp(v, "stage") ? 0 : 1;
  if (const char* v = getenv("BLSGPU_NTT_COLS")) {
    int a = 0, b = 0, cc = 0;
    if (sscanf(v, "%d,%d,%d", &a, &b, &cc) == 3 && a >= 6 && a <= cols_log_max && b >= 1 && b <= a && cc >= 64 && cc <= 1024) { d.ntt_cols[0] = a; d.ntt_cols[1] = b; d.ntt_cols[2] = cc; }
  }
  if (const char* v = getenv("BLSGPU_ITEM_CAP")) { long k = atol(v); if (k >= 8 && k <= item_cap_max) d.item_cap = (unsigned)k; }
  if (const char* v = getenv("BLSGPU_PRIO")) for (int i = 0; i < 3 && v[i]; i++) d.prio[i] = (v[i] == 'h' || v[i] == 'l') ? v[i] : 'n';
  if (const char* v = getenv("BLSGPU_WIDE_PROG")) d.wide_prog = v;
  return d;
}
