/* bls_oracle.c -- tier-1 oracle: C restatement of the reference's G1 hot path on 6x64-bit Montgomery limbs.
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (bls12_381_amd/) never does.
 *
 * The reference crate cannot be compiled here (no Rust toolchain), so this file restates its algorithms
 * limb for limb -- the same representation (six u64 limbs, R = 2^384, fully reduced after every
 * operation), the same schoolbook multiply + Montgomery reduction, the same RCB15 complete formulas and
 * the same 255-step double-and-add -- so that timing it is a fair stand-in for the Rust CPU path
 * ("kind": "port" in bench.py).  Cited lines refer to /root/reference/src.
 *
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_* against the tier-0 Python oracle (itself pinned
 * to the reference's KATs and golden files) and directly against the k*G golden vectors.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/bls_oracle.c -o oracle/_build/libblsoracle.so
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef struct { u64 l[6]; } fp;
typedef struct { fp x, y, z; } g1p;

/* fp.rs:70-80 */
static const u64 MODULUS[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                               0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 INV = 0x89f3fffcfffcfffdull;
/* fp.rs:83-90: R = 2^384 mod p */
static const fp FP_ONE = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                           0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};
static const fp FP_ZERO = {{0, 0, 0, 0, 0, 0}};

/* util.rs:1-20 */
static inline u64 adc(u64 a, u64 b, u64* carry) { u128 r = (u128)a + b + *carry; *carry = (u64)(r >> 64); return (u64)r; }
static inline u64 sbb(u64 a, u64 b, u64* borrow) { u128 r = (u128)a - ((u128)b + (*borrow >> 63)); *borrow = (u64)(r >> 64); return (u64)r; }
static inline u64 mac(u64 a, u64 b, u64 c, u64* carry) { u128 r = (u128)a + (u128)b * c + *carry; *carry = (u64)(r >> 64); return (u64)r; }

/* fp.rs:361-379 */
static inline fp subtract_p(const fp* a) {
  fp r; u64 bw = 0;
  for (int i = 0; i < 6; i++) r.l[i] = sbb(a->l[i], MODULUS[i], &bw);
  for (int i = 0; i < 6; i++) r.l[i] = (a->l[i] & bw) | (r.l[i] & ~bw);
  return r;
}
/* fp.rs:382-394 */
static inline fp fp_add(const fp* a, const fp* b) {
  fp d; u64 c = 0;
  for (int i = 0; i < 6; i++) d.l[i] = adc(a->l[i], b->l[i], &c);
  return subtract_p(&d);
}
/* fp.rs:397-418 */
static inline fp fp_neg(const fp* a) {
  fp d; u64 bw = 0, nz = 0;
  for (int i = 0; i < 6; i++) { d.l[i] = sbb(MODULUS[i], a->l[i], &bw); nz |= a->l[i]; }
  u64 mask = (u64)(nz == 0) - 1;
  for (int i = 0; i < 6; i++) d.l[i] &= mask;
  return d;
}
/* fp.rs:421-423 */
static inline fp fp_sub(const fp* a, const fp* b) { fp n = fp_neg(b); return fp_add(&n, a); }

/* fp.rs:487-562 */
static inline fp montgomery_reduce(u64 t[12]) {
  u64 carry2 = 0;
  for (int i = 0; i < 6; i++) {
    u64 k = t[i] * INV, carry = 0;
    (void)mac(t[i], k, MODULUS[0], &carry);
    for (int j = 1; j < 6; j++) t[i + j] = mac(t[i + j], k, MODULUS[j], &carry);
    t[i + 6] = adc(t[i + 6], carry2, &carry);
    carry2 = carry;
  }
  fp r; for (int i = 0; i < 6; i++) r.l[i] = t[6 + i];
  return subtract_p(&r);
}
/* fp.rs:565-609 */
static inline fp fp_mul(const fp* a, const fp* b) {
  u64 t[12] = {0};
  for (int i = 0; i < 6; i++) {
    u64 carry = 0;
    for (int j = 0; j < 6; j++) t[i + j] = mac(t[i + j], a->l[i], b->l[j], &carry);
    t[i + 6] = carry;
  }
  return montgomery_reduce(t);
}
/* fp.rs:613-660: the 15 cross products once, doubled by a one-bit shift, plus the six diagonal terms */
static inline fp fp_sqr(const fp* a) {
  u64 t[12] = {0};
  for (int i = 0; i < 5; i++) {
    u64 carry = 0;
    for (int j = i + 1; j < 6; j++) t[i + j] = mac(t[i + j], a->l[i], a->l[j], &carry);
    t[i + 6] = carry;
  }
  t[11] = t[10] >> 63;
  for (int k = 10; k >= 2; k--) t[k] = (t[k] << 1) | (t[k - 1] >> 63);
  t[1] <<= 1;
  u64 carry = 0;
  for (int i = 0; i < 6; i++) {
    t[2 * i] = mac(t[2 * i], a->l[i], a->l[i], &carry);
    t[2 * i + 1] = adc(t[2 * i + 1], 0, &carry);
  }
  return montgomery_reduce(t);
}
/* fp.rs:430-484 (Longa, ePrint 2022/367 Alg. 2): sum_i a_i b_i with the operand scanning of all T pairs interleaved and ONE
 * Montgomery reduction step per limb -- what Fp2::mul (T = 2) and Fp6::mul_interleaved (T = 6) are built on */
static inline fp fp_sum_of_products(int T, const fp* const* a, const fp* const* b) {
  u64 u[6] = {0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 6; j++) {
    u64 t[7] = {u[0], u[1], u[2], u[3], u[4], u[5], 0};
    for (int i = 0; i < T; i++) {
      u64 carry = 0;
      for (int k = 0; k < 6; k++) t[k] = mac(t[k], a[i]->l[j], b[i]->l[k], &carry);
      u64 c2 = 0; t[6] = adc(t[6], carry, &c2);
    }
    u64 k = t[0] * INV, carry = 0;
    (void)mac(t[0], k, MODULUS[0], &carry);
    for (int m = 1; m < 6; m++) u[m - 1] = mac(t[m], k, MODULUS[m], &carry);
    u64 c2 = 0; u[5] = adc(t[6], carry, &c2);
  }
  fp r; for (int i = 0; i < 6; i++) r.l[i] = u[i];
  return subtract_p(&r);
}

/* fp.rs:309-321,346-358 */
static fp fp_pow(const fp* a, const u64 e[6]) {
  fp res = FP_ONE;
  for (int w = 5; w >= 0; w--)
    for (int i = 63; i >= 0; i--) {
      res = fp_sqr(&res);
      if ((e[w] >> i) & 1) res = fp_mul(&res, a);
    }
  return res;
}
static fp fp_inv(const fp* a) {
  static const u64 e[6] = {0xb9feffffffffaaa9ull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                           0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
  return fp_pow(a, e);
}
static inline int fp_is_zero(const fp* a) { u64 t = 0; for (int i = 0; i < 6; i++) t |= a->l[i]; return t == 0; }

/* g1.rs:597-601 */
static inline fp mul_by_3b(fp a) { a = fp_add(&a, &a); a = fp_add(&a, &a); fp t = fp_add(&a, &a); return fp_add(&t, &a); }

static const g1p G1_IDENTITY = {{{0, 0, 0, 0, 0, 0}},
                                {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull,
                                  0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}},
                                {{0, 0, 0, 0, 0, 0}}};

/* g1.rs:638-667 */
static g1p g1_double(const g1p* p) {
  fp t0 = fp_sqr(&p->y);
  fp z3 = fp_add(&t0, &t0); z3 = fp_add(&z3, &z3); z3 = fp_add(&z3, &z3);
  fp t1 = fp_mul(&p->y, &p->z);
  fp t2 = fp_sqr(&p->z); t2 = mul_by_3b(t2);
  fp x3 = fp_mul(&t2, &z3);
  fp y3 = fp_add(&t0, &t2);
  z3 = fp_mul(&t1, &z3);
  t1 = fp_add(&t2, &t2); t2 = fp_add(&t1, &t2);
  t0 = fp_sub(&t0, &t2);
  y3 = fp_mul(&t0, &y3); y3 = fp_add(&x3, &y3);
  t1 = fp_mul(&p->x, &p->y);
  x3 = fp_mul(&t0, &t1); x3 = fp_add(&x3, &x3);
  g1p r = {x3, y3, z3};
  if (fp_is_zero(&p->z)) r = G1_IDENTITY;
  return r;
}
/* g1.rs:670-712 */
static g1p g1_add(const g1p* p, const g1p* q) {
  fp t0 = fp_mul(&p->x, &q->x), t1 = fp_mul(&p->y, &q->y), t2 = fp_mul(&p->z, &q->z);
  fp t3 = fp_add(&p->x, &p->y), t4 = fp_add(&q->x, &q->y);
  t3 = fp_mul(&t3, &t4); t4 = fp_add(&t0, &t1); t3 = fp_sub(&t3, &t4);
  t4 = fp_add(&p->y, &p->z);
  fp x3 = fp_add(&q->y, &q->z);
  t4 = fp_mul(&t4, &x3); x3 = fp_add(&t1, &t2); t4 = fp_sub(&t4, &x3);
  x3 = fp_add(&p->x, &p->z);
  fp y3 = fp_add(&q->x, &q->z);
  x3 = fp_mul(&x3, &y3); y3 = fp_add(&t0, &t2); y3 = fp_sub(&x3, &y3);
  x3 = fp_add(&t0, &t0); t0 = fp_add(&x3, &t0);
  t2 = mul_by_3b(t2);
  fp z3 = fp_add(&t1, &t2); t1 = fp_sub(&t1, &t2);
  y3 = mul_by_3b(y3);
  x3 = fp_mul(&t4, &y3); t2 = fp_mul(&t3, &t1); x3 = fp_sub(&t2, &x3);
  y3 = fp_mul(&y3, &t0); t1 = fp_mul(&t1, &z3); y3 = fp_add(&t1, &y3);
  t0 = fp_mul(&t0, &t3); z3 = fp_mul(&z3, &t4); z3 = fp_add(&z3, &t0);
  g1p r = {x3, y3, z3};
  return r;
}
/* g1.rs:754-774: 255 iterations, double then ALWAYS add, select on the bit */
static g1p g1_multiply(const g1p* p, const uint8_t by[32]) {
  g1p acc = G1_IDENTITY;
  int first = 1;
  for (int byte = 31; byte >= 0; byte--)
    for (int i = 7; i >= 0; i--) {
      if (first) { first = 0; continue; }
      acc = g1_double(&acc);
      g1p s = g1_add(&acc, p);
      if ((by[byte] >> i) & 1) acc = s;
    }
  return acc;
}

/* ---- exported API (plain pointers; limbs are the reference's canonical Montgomery limbs) ---------------- */
void ora_fp_mul(const u64* a, const u64* b, u64* out, long n) {
  for (long i = 0; i < n; i++) { fp r = fp_mul((const fp*)(a + 6 * i), (const fp*)(b + 6 * i)); memcpy(out + 6 * i, r.l, 48); }
}
void ora_fp_op(int op, const u64* a, const u64* b, u64* out, long n) {
  for (long i = 0; i < n; i++) {
    const fp* x = (const fp*)(a + 6 * i); const fp* y = (const fp*)(b + 6 * i); fp r;
    switch (op) { case 0: r = fp_mul(x, y); break; case 1: r = fp_add(x, y); break; case 2: r = fp_sub(x, y); break;
                  case 3: r = fp_sqr(x); break; case 4: r = fp_is_zero(x) ? FP_ZERO : fp_inv(x); break; default: r = fp_neg(x); }
    memcpy(out + 6 * i, r.l, 48);
  }
}
void ora_g1_double(const u64* p, u64* out) { g1p r = g1_double((const g1p*)p); memcpy(out, &r, 144); }
void ora_g1_add(const u64* p, const u64* q, u64* out) { g1p r = g1_add((const g1p*)p, (const g1p*)q); memcpy(out, &r, 144); }
/* `&G1Affine * &Scalar` (g1.rs:573-579): affine x|y (+ infinity flag) times 32-byte LE scalar -> projective */
void ora_g1_affine_mul(const u64* xy, int infinity, const uint8_t* scalar, u64* out) {
  g1p p; memcpy(&p.x, xy, 48); memcpy(&p.y, xy + 6, 48); p.z = infinity ? FP_ZERO : FP_ONE;
  g1p r = g1_multiply(&p, scalar); memcpy(out, &r, 144);
}
/* projective -> affine (g1.rs:49-63); returns infinity flag */
int ora_g1_to_affine(const u64* xyz, u64* xy) {
  const g1p* p = (const g1p*)xyz;
  if (fp_is_zero(&p->z)) { memcpy(xy, FP_ZERO.l, 48); memcpy(xy + 6, FP_ONE.l, 48); return 1; }
  fp zi = fp_inv(&p->z), x = fp_mul(&p->x, &zi), y = fp_mul(&p->y, &zi);
  memcpy(xy, x.l, 48); memcpy(xy + 6, y.l, 48); return 0;
}
/* The reference's only MSM:  points.zip(scalars).map(|(p,s)| p*s).sum()  (g1.rs:161-171, 573-579, 754-774).
 * `threads` <= 0 uses every OpenMP thread; the fold is done per thread then in thread order (the affine
 * result does not depend on the order).  Returns the number of threads used. */
int ora_g1_msm(const u64* xy, const uint8_t* inf, const uint8_t* scalars, long n, int threads, u64* out_xyz) {
  int used = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  if (threads > 1024) threads = 1024;
  used = threads;
  g1p part[1024];
  for (int t = 0; t < threads; t++) part[t] = G1_IDENTITY;
#pragma omp parallel num_threads(threads)
  {
    int t = omp_get_thread_num();
    g1p acc = G1_IDENTITY;
#pragma omp for schedule(dynamic, 16)
    for (long i = 0; i < n; i++) {
      g1p p; memcpy(&p.x, xy + 12 * i, 48); memcpy(&p.y, xy + 12 * i + 6, 48); p.z = (inf && inf[i]) ? FP_ZERO : FP_ONE;
      g1p m = g1_multiply(&p, scalars + 32 * i);
      acc = g1_add(&acc, &m);
    }
    part[t] = acc;
  }
  g1p acc = G1_IDENTITY;
  for (int t = 0; t < threads; t++) acc = g1_add(&acc, &part[t]);
#else
  (void)threads;
  g1p acc = G1_IDENTITY;
  for (long i = 0; i < n; i++) {
    g1p p; memcpy(&p.x, xy + 12 * i, 48); memcpy(&p.y, xy + 12 * i + 6, 48); p.z = (inf && inf[i]) ? FP_ZERO : FP_ONE;
    g1p m = g1_multiply(&p, scalars + 32 * i);
    acc = g1_add(&acc, &m);
  }
#endif
  memcpy(out_xyz, &acc, 144);
  return used;
}
/* n independent `&G1Affine * &Scalar` (g1.rs:573-579, 754-774), each converted to affine (g1.rs:49-63): the checker of the
 * batched variable-base kernel.  out_xy: n x 12 limbs, out_inf: n bytes. */
int ora_g1_mul_batch_affine(const u64* xy, const uint8_t* inf, const uint8_t* scalars, long n, int threads, u64* out_xy, uint8_t* out_inf) {
  int used = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  used = threads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
  for (long i = 0; i < n; i++) {
    g1p p; memcpy(&p.x, xy + 12 * i, 48); memcpy(&p.y, xy + 12 * i + 6, 48); p.z = (inf && inf[i]) ? FP_ZERO : FP_ONE;
    g1p m = g1_multiply(&p, scalars + 32 * i);
    out_inf[i] = (uint8_t)ora_g1_to_affine((const u64*)&m, out_xy + 12 * i);
  }
  return used;
}
int ora_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ================================================================================================================
 * Pairing path: Fp2 / Fp6 / Fp12 towers, the Miller loop and the final exponentiation, restating the reference
 * (fp2.rs, fp6.rs, fp12.rs, pairings.rs; line numbers at each function).  Same purpose as above: checker and CPU
 * baseline ("full pairing", "miller loop", "final exponentiation" of benches/groups.rs).  The three Frobenius
 * constants come from oracle/_build/ora_consts.h, written by oracle/c_oracle.py from the tier-0 oracle's values
 * (themselves compared with the reference's literals in tests/test_oracle_golden.py).
 * ================================================================================================================ */
#include "_build/ora_consts.h"      /* FROB6_C1, FROB6_C2, FROB12_C1: fp2 in Montgomery limbs */
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;
static const fp2 FP2_ZERO_C = {{{0, 0, 0, 0, 0, 0}}, {{0, 0, 0, 0, 0, 0}}};

static inline fp2 fp2_add(const fp2* a, const fp2* b) { fp2 r = {fp_add(&a->c0, &b->c0), fp_add(&a->c1, &b->c1)}; return r; }   /* fp2.rs:224-229 */
static inline fp2 fp2_sub(const fp2* a, const fp2* b) { fp2 r = {fp_sub(&a->c0, &b->c0), fp_sub(&a->c1, &b->c1)}; return r; }   /* :231-236 */
static inline fp2 fp2_neg(const fp2* a) { fp2 r = {fp_neg(&a->c0), fp_neg(&a->c1)}; return r; }                                 /* :238-243 */
static inline fp2 fp2_conj(const fp2* a) { fp2 r = {a->c0, fp_neg(&a->c1)}; return r; }                                         /* :148-153 */
static inline fp2 fp2_dbl(const fp2* a) { return fp2_add(a, a); }
static inline fp2 fp2_mul(const fp2* a, const fp2* b) {                                                                          /* :205-222 */
  fp na1 = fp_neg(&a->c1);
  const fp* x0[2] = {&a->c0, &na1};   const fp* y0[2] = {&b->c0, &b->c1};      /* c0 = a0 b0 - a1 b1 */
  const fp* x1[2] = {&a->c0, &a->c1}; const fp* y1[2] = {&b->c1, &b->c0};      /* c1 = a0 b1 + a1 b0 */
  fp2 r = {fp_sum_of_products(2, x0, y0), fp_sum_of_products(2, x1, y1)};
  return r;
}
static inline fp2 fp2_sqr(const fp2* a) {                                                                                       /* :182-203 */
  fp s = fp_add(&a->c0, &a->c1), d = fp_sub(&a->c0, &a->c1), t = fp_add(&a->c0, &a->c0);
  fp2 r = {fp_mul(&s, &d), fp_mul(&t, &a->c1)};
  return r;
}
static inline fp2 fp2_mul_by_nonresidue(const fp2* a) { fp2 r = {fp_sub(&a->c0, &a->c1), fp_add(&a->c0, &a->c1)}; return r; }   /* :156-166 */
static inline fp2 fp2_mul_fp(const fp2* a, const fp* k) { fp2 r = {fp_mul(&a->c0, k), fp_mul(&a->c1, k)}; return r; }
static fp2 fp2_inv(const fp2* a) {                                                                                             /* :300-319 */
  fp s0 = fp_sqr(&a->c0), s1 = fp_sqr(&a->c1), n = fp_add(&s0, &s1), t = fp_inv(&n), nt = fp_neg(&t);
  fp2 r = {fp_mul(&a->c0, &t), fp_mul(&a->c1, &nt)};
  return r;
}
static inline fp2 fp2_one(void) { fp2 r = {FP_ONE, FP_ZERO}; return r; }

static inline fp6 fp6_add(const fp6* a, const fp6* b) { fp6 r = {fp2_add(&a->c0, &b->c0), fp2_add(&a->c1, &b->c1), fp2_add(&a->c2, &b->c2)}; return r; }
static inline fp6 fp6_sub(const fp6* a, const fp6* b) { fp6 r = {fp2_sub(&a->c0, &b->c0), fp2_sub(&a->c1, &b->c1), fp2_sub(&a->c2, &b->c2)}; return r; }
static inline fp6 fp6_neg(const fp6* a) { fp6 r = {fp2_neg(&a->c0), fp2_neg(&a->c1), fp2_neg(&a->c2)}; return r; }
static inline fp6 fp6_mul_by_nonresidue(const fp6* a) { fp6 r = {fp2_mul_by_nonresidue(&a->c2), a->c0, a->c1}; return r; }      /* fp6.rs:139-150 */
static fp6 fp6_mul(const fp6* a, const fp6* b) {                                                                               /* :200-274 mul_interleaved */
  /* six sums of six products, one interleaved reduction each (fp6.rs:241-273) */
  fp b10p = fp_add(&b->c1.c0, &b->c1.c1), b10m = fp_sub(&b->c1.c0, &b->c1.c1);
  fp b20p = fp_add(&b->c2.c0, &b->c2.c1), b20m = fp_sub(&b->c2.c0, &b->c2.c1);
  fp n01 = fp_neg(&a->c0.c1), n11 = fp_neg(&a->c1.c1), n21 = fp_neg(&a->c2.c1);
  const fp* xe[6] = {&a->c0.c0, &n01, &a->c1.c0, &n11, &a->c2.c0, &n21};                 /* real parts: a_i0, -a_i1 */
  const fp* xo[6] = {&a->c0.c0, &a->c0.c1, &a->c1.c0, &a->c1.c1, &a->c2.c0, &a->c2.c1};  /* imaginary parts */
  const fp* y00[6] = {&b->c0.c0, &b->c0.c1, &b20m, &b20p, &b10m, &b10p};
  const fp* y01[6] = {&b->c0.c1, &b->c0.c0, &b20p, &b20m, &b10p, &b10m};
  const fp* y10[6] = {&b->c1.c0, &b->c1.c1, &b->c0.c0, &b->c0.c1, &b20m, &b20p};
  const fp* y11[6] = {&b->c1.c1, &b->c1.c0, &b->c0.c1, &b->c0.c0, &b20p, &b20m};
  const fp* y20[6] = {&b->c2.c0, &b->c2.c1, &b->c1.c0, &b->c1.c1, &b->c0.c0, &b->c0.c1};
  const fp* y21[6] = {&b->c2.c1, &b->c2.c0, &b->c1.c1, &b->c1.c0, &b->c0.c1, &b->c0.c0};
  fp6 r = {{fp_sum_of_products(6, xe, y00), fp_sum_of_products(6, xo, y01)},
           {fp_sum_of_products(6, xe, y10), fp_sum_of_products(6, xo, y11)},
           {fp_sum_of_products(6, xe, y20), fp_sum_of_products(6, xo, y21)}};
  return r;
}
static fp6 fp6_sqr(const fp6* a) {                                                                                             /* :277-291 */
  fp2 s0 = fp2_sqr(&a->c0), ab = fp2_mul(&a->c0, &a->c1), s1 = fp2_dbl(&ab);
  fp2 d = fp2_sub(&a->c0, &a->c1), e = fp2_add(&d, &a->c2), s2 = fp2_sqr(&e);
  fp2 bc = fp2_mul(&a->c1, &a->c2), s3 = fp2_dbl(&bc), s4 = fp2_sqr(&a->c2);
  fp2 n3 = fp2_mul_by_nonresidue(&s3), n4 = fp2_mul_by_nonresidue(&s4);
  fp2 u = fp2_add(&s1, &s2), v = fp2_add(&u, &s3), w = fp2_sub(&v, &s0);
  fp6 r = {fp2_add(&n3, &s0), fp2_add(&n4, &s1), fp2_sub(&w, &s4)};
  return r;
}
static fp6 fp6_mul_by_1(const fp6* a, const fp2* c1) {                                                                         /* :113-119 */
  fp2 t = fp2_mul(&a->c2, c1);
  fp6 r = {fp2_mul_by_nonresidue(&t), fp2_mul(&a->c0, c1), fp2_mul(&a->c1, c1)};
  return r;
}
static fp6 fp6_mul_by_01(const fp6* a, const fp2* c0, const fp2* c1) {                                                         /* :121-136 */
  fp2 a_a = fp2_mul(&a->c0, c0), b_b = fp2_mul(&a->c1, c1);
  fp2 t = fp2_mul(&a->c2, c1), nt = fp2_mul_by_nonresidue(&t), t1 = fp2_add(&nt, &a_a);
  fp2 cs = fp2_add(c0, c1), as = fp2_add(&a->c0, &a->c1), m = fp2_mul(&cs, &as), m1 = fp2_sub(&m, &a_a), t2 = fp2_sub(&m1, &b_b);
  fp2 u = fp2_mul(&a->c2, c0), t3 = fp2_add(&u, &b_b);
  fp6 r = {t1, t2, t3};
  return r;
}
static fp6 fp6_frobenius(const fp6* a) {                                                                                       /* :154-188 */
  fp2 c0 = fp2_conj(&a->c0), c1 = fp2_conj(&a->c1), c2 = fp2_conj(&a->c2);
  fp6 r = {c0, fp2_mul(&c1, &FROB6_C1), fp2_mul(&c2, &FROB6_C2)};
  return r;
}
static fp6 fp6_inv(const fp6* a) {                                                                                             /* :294-312 */
  fp2 s0 = fp2_sqr(&a->c0), m12 = fp2_mul(&a->c1, &a->c2), n12 = fp2_mul_by_nonresidue(&m12), c0 = fp2_sub(&s0, &n12);
  fp2 s2 = fp2_sqr(&a->c2), n2 = fp2_mul_by_nonresidue(&s2), m01 = fp2_mul(&a->c0, &a->c1), c1 = fp2_sub(&n2, &m01);
  fp2 s1 = fp2_sqr(&a->c1), m02 = fp2_mul(&a->c0, &a->c2), c2 = fp2_sub(&s1, &m02);
  fp2 x = fp2_mul(&a->c1, &c2), y = fp2_mul(&a->c2, &c1), xy = fp2_add(&x, &y), nx = fp2_mul_by_nonresidue(&xy);
  fp2 z = fp2_mul(&a->c0, &c0), tmp = fp2_add(&nx, &z), t = fp2_inv(&tmp);
  fp6 r = {fp2_mul(&t, &c0), fp2_mul(&t, &c1), fp2_mul(&t, &c2)};
  return r;
}
static inline fp12 fp12_one(void) { fp12 r; memset(&r, 0, sizeof r); r.c0.c0.c0 = FP_ONE; return r; }
static fp12 fp12_mul(const fp12* a, const fp12* b) {                                                                           /* fp12.rs:197-214 */
  fp6 aa = fp6_mul(&a->c0, &b->c0), bb = fp6_mul(&a->c1, &b->c1), o = fp6_add(&b->c0, &b->c1), s = fp6_add(&a->c1, &a->c0);
  fp6 c1 = fp6_mul(&s, &o); c1 = fp6_sub(&c1, &aa); c1 = fp6_sub(&c1, &bb);
  fp6 nb = fp6_mul_by_nonresidue(&bb);
  fp12 r = {fp6_add(&nb, &aa), c1};
  return r;
}
static fp12 fp12_sqr(const fp12* a) {                                                                                          /* :174-185 */
  fp6 ab = fp6_mul(&a->c0, &a->c1), c0c1 = fp6_add(&a->c0, &a->c1), n1 = fp6_mul_by_nonresidue(&a->c1), c0 = fp6_add(&n1, &a->c0);
  c0 = fp6_mul(&c0, &c0c1); c0 = fp6_sub(&c0, &ab);
  fp6 nab = fp6_mul_by_nonresidue(&ab);
  fp12 r = {fp6_sub(&c0, &nab), fp6_add(&ab, &ab)};
  return r;
}
static fp12 fp12_mul_by_014(const fp12* a, const fp2* c0, const fp2* c1, const fp2* c4) {                                      /* :116-128 */
  fp6 aa = fp6_mul_by_01(&a->c0, c0, c1), bb = fp6_mul_by_1(&a->c1, c4);
  fp2 o = fp2_add(c1, c4);
  fp6 s = fp6_add(&a->c1, &a->c0), r1 = fp6_mul_by_01(&s, c0, &o);
  r1 = fp6_sub(&r1, &aa); r1 = fp6_sub(&r1, &bb);
  fp6 nb = fp6_mul_by_nonresidue(&bb);
  fp12 r = {fp6_add(&nb, &aa), r1};
  return r;
}
static inline fp12 fp12_conj(const fp12* a) { fp12 r = {a->c0, fp6_neg(&a->c1)}; return r; }                                    /* :136-141 */
static fp12 fp12_frobenius(const fp12* a) {                                                                                    /* :145-171 */
  fp6 c0 = fp6_frobenius(&a->c0), c1 = fp6_frobenius(&a->c1);
  fp12 r = {c0, {fp2_mul(&c1.c0, &FROB12_C1), fp2_mul(&c1.c1, &FROB12_C1), fp2_mul(&c1.c2, &FROB12_C1)}};
  return r;
}
static fp12 fp12_inv(const fp12* a) {                                                                                          /* :187-194 */
  fp6 s0 = fp6_sqr(&a->c0), s1 = fp6_sqr(&a->c1), n1 = fp6_mul_by_nonresidue(&s1), d = fp6_sub(&s0, &n1), t = fp6_inv(&d), nt = fp6_neg(&t);
  fp12 r = {fp6_mul(&a->c0, &t), fp6_mul(&a->c1, &nt)};
  return r;
}

typedef struct { fp2 x, y, z; } g2r;          /* the running point of the Miller loop */
typedef struct { fp2 a, b, c; } line3;
static line3 doubling_step(g2r* r) {                                                                                           /* pairings.rs:709-738 */
  fp2 tmp0 = fp2_sqr(&r->x), tmp1 = fp2_sqr(&r->y), tmp2 = fp2_sqr(&tmp1);
  fp2 u = fp2_add(&tmp1, &r->x), u2 = fp2_sqr(&u), v = fp2_sub(&u2, &tmp0), tmp3 = fp2_sub(&v, &tmp2);
  tmp3 = fp2_dbl(&tmp3);
  fp2 d0 = fp2_dbl(&tmp0), tmp4 = fp2_add(&d0, &tmp0), tmp6 = fp2_add(&r->x, &tmp4), tmp5 = fp2_sqr(&tmp4), zsq = fp2_sqr(&r->z);
  fp2 x1 = fp2_sub(&tmp5, &tmp3); r->x = fp2_sub(&x1, &tmp3);
  fp2 zy = fp2_add(&r->z, &r->y), zy2 = fp2_sqr(&zy), z1 = fp2_sub(&zy2, &tmp1); r->z = fp2_sub(&z1, &zsq);
  fp2 w = fp2_sub(&tmp3, &r->x); r->y = fp2_mul(&w, &tmp4);
  tmp2 = fp2_dbl(&tmp2); tmp2 = fp2_dbl(&tmp2); tmp2 = fp2_dbl(&tmp2);
  r->y = fp2_sub(&r->y, &tmp2);
  tmp3 = fp2_mul(&tmp4, &zsq); tmp3 = fp2_dbl(&tmp3); tmp3 = fp2_neg(&tmp3);
  fp2 s6 = fp2_sqr(&tmp6), s7 = fp2_sub(&s6, &tmp0); tmp6 = fp2_sub(&s7, &tmp5);
  tmp1 = fp2_dbl(&tmp1); tmp1 = fp2_dbl(&tmp1);
  tmp6 = fp2_sub(&tmp6, &tmp1);
  tmp0 = fp2_mul(&r->z, &zsq); tmp0 = fp2_dbl(&tmp0);
  line3 l = {tmp0, tmp3, tmp6};
  return l;
}
static line3 addition_step(g2r* r, const fp2* qx, const fp2* qy) {                                                             /* :740-770 */
  fp2 zsq = fp2_sqr(&r->z), ysq = fp2_sqr(qy), t0 = fp2_mul(&zsq, qx);
  fp2 a = fp2_add(qy, &r->z), a2 = fp2_sqr(&a), b = fp2_sub(&a2, &ysq), c = fp2_sub(&b, &zsq), t1 = fp2_mul(&c, &zsq);
  fp2 t2 = fp2_sub(&t0, &r->x), t3 = fp2_sqr(&t2), t4 = fp2_dbl(&t3); t4 = fp2_dbl(&t4);
  fp2 t5 = fp2_mul(&t4, &t2), d = fp2_sub(&t1, &r->y), t6 = fp2_sub(&d, &r->y), t9 = fp2_mul(&t6, qx), t7 = fp2_mul(&t4, &r->x);
  fp2 e = fp2_sqr(&t6), f = fp2_sub(&e, &t5), g = fp2_sub(&f, &t7); r->x = fp2_sub(&g, &t7);
  fp2 h = fp2_add(&r->z, &t2), h2 = fp2_sqr(&h), i = fp2_sub(&h2, &zsq); r->z = fp2_sub(&i, &t3);
  fp2 t10 = fp2_add(qy, &r->z), j = fp2_sub(&t7, &r->x), t8 = fp2_mul(&j, &t6);
  t0 = fp2_mul(&r->y, &t5); t0 = fp2_dbl(&t0);
  r->y = fp2_sub(&t8, &t0);
  fp2 k = fp2_sqr(&t10); t10 = fp2_sub(&k, &ysq);
  fp2 ztsq = fp2_sqr(&r->z); t10 = fp2_sub(&t10, &ztsq);
  fp2 t9d = fp2_dbl(&t9); t9 = fp2_sub(&t9d, &t10);
  t10 = fp2_dbl(&r->z);
  t6 = fp2_neg(&t6); t1 = fp2_dbl(&t6);
  line3 l = {t10, t1, t9};
  return l;
}
static fp12 ell(const fp12* f, const line3* l, const fp* px, const fp* py) {                                                   /* :696-707 */
  fp2 c0 = fp2_mul_fp(&l->a, py), c1 = fp2_mul_fp(&l->b, px);
  return fp12_mul_by_014(f, &l->c, &c1, &c0);
}
static const uint64_t BLS_X_C = 0xd201000000010000ull;
static fp12 miller_loop_c(const fp* px, const fp* py, const fp2* qx, const fp2* qy) {                                          /* :607-653, :668-694 */
  g2r r = {*qx, *qy, fp2_one()};
  fp12 f = fp12_one();
  int found = 0;
  for (int b = 63; b >= 0; b--) {
    int i = (int)(((BLS_X_C >> 1) >> b) & 1);
    if (!found) { found = i; continue; }
    line3 l = doubling_step(&r); f = ell(&f, &l, px, py);
    if (i) { l = addition_step(&r, qx, qy); f = ell(&f, &l, px, py); }
    f = fp12_sqr(&f);
  }
  line3 l = doubling_step(&r); f = ell(&f, &l, px, py);
  return fp12_conj(&f);                                  /* BLS_X_IS_NEGATIVE */
}
static void fp4_square(const fp2* a, const fp2* b, fp2* c0, fp2* c1) {                                                         /* :50-62 */
  fp2 t0 = fp2_sqr(a), t1 = fp2_sqr(b), t2 = fp2_mul_by_nonresidue(&t1);
  *c0 = fp2_add(&t2, &t0);
  fp2 s = fp2_add(a, b); t2 = fp2_sqr(&s); t2 = fp2_sub(&t2, &t0);
  *c1 = fp2_sub(&t2, &t1);
}
static fp12 cyclotomic_square(const fp12* f) {                                                                                 /* :66-112 */
  fp2 z0 = f->c0.c0, z4 = f->c0.c1, z3 = f->c0.c2, z2 = f->c1.c0, z1 = f->c1.c1, z5 = f->c1.c2, t0, t1, t2, t3;
  fp4_square(&z0, &z1, &t0, &t1);
  z0 = fp2_sub(&t0, &z0); z0 = fp2_dbl(&z0); z0 = fp2_add(&z0, &t0);
  z1 = fp2_add(&t1, &z1); z1 = fp2_dbl(&z1); z1 = fp2_add(&z1, &t1);
  fp4_square(&z2, &z3, &t0, &t1);
  fp4_square(&z4, &z5, &t2, &t3);
  z4 = fp2_sub(&t0, &z4); z4 = fp2_dbl(&z4); z4 = fp2_add(&z4, &t0);
  z5 = fp2_add(&t1, &z5); z5 = fp2_dbl(&z5); z5 = fp2_add(&z5, &t1);
  t0 = fp2_mul_by_nonresidue(&t3);
  z2 = fp2_add(&t0, &z2); z2 = fp2_dbl(&z2); z2 = fp2_add(&z2, &t0);
  z3 = fp2_sub(&t2, &z3); z3 = fp2_dbl(&z3); z3 = fp2_add(&z3, &t2);
  fp12 r = {{z0, z4, z3}, {z2, z1, z5}};
  return r;
}
static fp12 cyclotomic_exp(const fp12* f) {                                                                                    /* :114-132 */
  fp12 tmp = fp12_one();
  int found = 0;
  for (int b = 63; b >= 0; b--) {
    int i = (int)((BLS_X_C >> b) & 1);
    if (found) tmp = cyclotomic_square(&tmp); else found = i;
    if (i) tmp = fp12_mul(&tmp, f);
  }
  return fp12_conj(&tmp);
}
static fp12 final_exponentiation_c(const fp12* fin) {                                                                          /* :134-173 */
  fp12 t0 = *fin;
  for (int i = 0; i < 6; i++) t0 = fp12_frobenius(&t0);
  fp12 t1 = fp12_inv(fin), t2 = fp12_mul(&t0, &t1);
  t1 = t2;
  t2 = fp12_frobenius(&t2); t2 = fp12_frobenius(&t2);
  t2 = fp12_mul(&t2, &t1);
  fp12 cs = cyclotomic_square(&t2); t1 = fp12_conj(&cs);
  fp12 t3 = cyclotomic_exp(&t2), t4 = cyclotomic_square(&t3), t5 = fp12_mul(&t1, &t3);
  t1 = cyclotomic_exp(&t5);
  t0 = cyclotomic_exp(&t1);
  fp12 t6 = cyclotomic_exp(&t0);
  t6 = fp12_mul(&t6, &t4);
  t4 = cyclotomic_exp(&t6);
  t5 = fp12_conj(&t5);
  fp12 t52 = fp12_mul(&t5, &t2); t4 = fp12_mul(&t4, &t52);
  t5 = fp12_conj(&t2);
  t1 = fp12_mul(&t1, &t2);
  t1 = fp12_frobenius(&t1); t1 = fp12_frobenius(&t1); t1 = fp12_frobenius(&t1);
  t6 = fp12_mul(&t6, &t5);
  t6 = fp12_frobenius(&t6);
  t3 = fp12_mul(&t3, &t0);
  t3 = fp12_frobenius(&t3); t3 = fp12_frobenius(&t3);
  t3 = fp12_mul(&t3, &t1);
  t3 = fp12_mul(&t3, &t6);
  return fp12_mul(&t3, &t4);
}
/* mode 0: pairing (pairings.rs:607-653), 1: Miller loop only, 2: final exponentiation of in[i] (g1 = the 72-limb inputs).
 * g1: n x 12 limbs, g2: n x 24 limbs (x.c0 x.c1 y.c0 y.c1), inf flags may be NULL; out: n x 72 limbs. */
int ora_pairing_batch(int mode, const u64* g1, const uint8_t* g1inf, const u64* g2, const uint8_t* g2inf, long n, int threads, u64* out) {
  int used = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  used = threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
  for (long i = 0; i < n; i++) {
    fp12 f;
    if (mode == 2) {
      fp12 in; memcpy(&in, g1 + 72 * i, 576);
      f = final_exponentiation_c(&in);
    } else if ((g1inf && g1inf[i]) || (g2inf && g2inf[i])) {
      f = fp12_one();
    } else {
      fp px, py; fp2 qx, qy;
      memcpy(&px, g1 + 12 * i, 48); memcpy(&py, g1 + 12 * i + 6, 48);
      memcpy(&qx, g2 + 24 * i, 96); memcpy(&qy, g2 + 24 * i + 12, 96);
      f = miller_loop_c(&px, &py, &qx, &qy);
      if (mode == 0) f = final_exponentiation_c(&f);
    }
    memcpy(out + 72 * i, &f, 576);
  }
  (void)FP2_ZERO_C;
  return used;
}

/* ---- `G2Prepared` and the reference's own `multi_miller_loop` schedule (pairings.rs:487-603) --------------------------------- */
typedef struct { line3 l[68]; } g2prep;
static void g2_prepare_c(const fp2* qx, const fp2* qy, g2prep* out) {                                                        /* :504-546 */
  g2r r = {*qx, *qy, fp2_one()};
  int n = 0, found = 0;
  for (int b = 63; b >= 0; b--) {
    int i = (int)(((BLS_X_C >> 1) >> b) & 1);
    if (!found) { found = i; continue; }
    out->l[n++] = doubling_step(&r);
    if (i) out->l[n++] = addition_step(&r, qx, qy);
  }
  out->l[n++] = doubling_step(&r);
}
/* coefficients of m points (m x 24 limbs in), m x 68 x 36 limbs out: `G2Prepared::from(G2Affine)` per point (identity: the generator's
 * coefficients are the caller's business -- pass the generator) */
int ora_g2_prepare(const u64* g2, long m, int threads, u64* out) {
  int used = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  used = threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
  for (long i = 0; i < m; i++) {
    fp2 qx, qy; memcpy(&qx, g2 + 24 * i, 96); memcpy(&qy, g2 + 24 * i + 12, 96);
    g2prep t; g2_prepare_c(&qx, &qy, &t);
    memcpy(out + (size_t)i * 68 * 36, &t, sizeof t);
  }
  return used;
}
/* N independent `multi_miller_loop(&[(&G1Affine, &G2Prepared)])` (pairings.rs:554-603: ONE accumulator per call, per step every
 * term's `ell`, one squaring for all), optionally followed by `.final_exponentiation()`.  Term t of segment s = terms off[s] ..
 * off[s+1]: qidx[t] names a prepared table (68 x 36 limbs each in `tabs`, flag in tabinf) or -- 0xffffffff -- the point g2[t], which
 * is PREPARED HERE first, as a caller of the reference must (`G2Prepared::from`, inside the timed region of the CPU baseline). */
int ora_multi_miller_prepared_many(const u64* g1, const uint8_t* g1inf, const u64* g2, const uint8_t* g2inf, const unsigned* qidx, const u64* tabs,
                                   const uint8_t* tabinf, const u64* off, long nseg, int final_exp, int threads, u64* out) {
  int used = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  used = threads;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
  for (long s = 0; s < nseg; s++) {
    const long lo = (long)off[s], k = (long)off[s + 1] - lo;
    const g2prep** tab = (const g2prep**)malloc((size_t)(k ? k : 1) * sizeof(*tab));
    g2prep* own = (g2prep*)malloc((size_t)(k ? k : 1) * sizeof(*own));
    fp* px = (fp*)malloc((size_t)(k ? k : 1) * sizeof(fp)); fp* py = (fp*)malloc((size_t)(k ? k : 1) * sizeof(fp));
    long live = 0;
    for (long j = 0; j < k; j++) {
      const long t = lo + j;
      const unsigned qi = qidx ? qidx[t] : 0xffffffffu;
      int skip = g1inf && g1inf[t];
      if (qi == 0xffffffffu) skip = skip || (g2inf && g2inf[t]); else skip = skip || (tabinf && tabinf[qi]);
      if (skip) continue;                                                                                                    /* :566-569 */
      memcpy(&px[live], g1 + 12 * t, 48); memcpy(&py[live], g1 + 12 * t + 6, 48);
      if (qi == 0xffffffffu) {
        fp2 qx, qy; memcpy(&qx, g2 + 24 * t, 96); memcpy(&qy, g2 + 24 * t + 12, 96);
        g2_prepare_c(&qx, &qy, &own[live]);
        tab[live] = &own[live];
      } else tab[live] = (const g2prep*)(tabs + (size_t)qi * 68 * 36);
      live++;
    }
    fp12 f = fp12_one();
    int idx = 0, found = 0;
    for (int b = 63; b >= 0; b--) {
      int i = (int)(((BLS_X_C >> 1) >> b) & 1);
      if (!found) { found = i; continue; }
      for (long j = 0; j < live; j++) f = ell(&f, &tab[j]->l[idx], &px[j], &py[j]);
      idx++;
      if (i) { for (long j = 0; j < live; j++) f = ell(&f, &tab[j]->l[idx], &px[j], &py[j]); idx++; }
      f = fp12_sqr(&f);
    }
    for (long j = 0; j < live; j++) f = ell(&f, &tab[j]->l[idx], &px[j], &py[j]);
    f = fp12_conj(&f);
    if (final_exp) f = final_exponentiation_c(&f);
    memcpy(out + 72 * s, &f, 576);
    free(tab); free(own); free(px); free(py);
  }
  return used;
}

/* ---- G2: the same complete formulas over Fp2 (g2.rs:709-738 double, :741-783 add, :825-845 multiply, :650-652 mul_by_3b) ---- */
typedef struct { fp2 x, y, z; } g2p;
static inline int fp2_is_zero(const fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline fp2 g2_mul_by_3b(fp2 a) {                       /* 3b' = 12 (1 + u) */
  fp2 t = fp2_mul_by_nonresidue(&a);
  fp2 d = fp2_dbl(&t); d = fp2_dbl(&d);                       /* 4t */
  fp2 e = fp2_dbl(&d);                                        /* 8t */
  return fp2_add(&e, &d);                                     /* 12t */
}
static g2p g2_identity(void) { g2p r; memset(&r, 0, sizeof r); r.y.c0 = FP_ONE; return r; }
static g2p g2_double(const g2p* p) {
  fp2 t0 = fp2_sqr(&p->y);
  fp2 z3 = fp2_dbl(&t0); z3 = fp2_dbl(&z3); z3 = fp2_dbl(&z3);
  fp2 t1 = fp2_mul(&p->y, &p->z);
  fp2 t2 = fp2_sqr(&p->z); t2 = g2_mul_by_3b(t2);
  fp2 x3 = fp2_mul(&t2, &z3);
  fp2 y3 = fp2_add(&t0, &t2);
  z3 = fp2_mul(&t1, &z3);
  t1 = fp2_dbl(&t2); t2 = fp2_add(&t1, &t2);
  t0 = fp2_sub(&t0, &t2);
  y3 = fp2_mul(&t0, &y3); y3 = fp2_add(&x3, &y3);
  t1 = fp2_mul(&p->x, &p->y);
  x3 = fp2_mul(&t0, &t1); x3 = fp2_dbl(&x3);
  g2p r = {x3, y3, z3};
  if (fp2_is_zero(&p->z)) r = g2_identity();
  return r;
}
static g2p g2_add(const g2p* p, const g2p* q) {
  fp2 t0 = fp2_mul(&p->x, &q->x), t1 = fp2_mul(&p->y, &q->y), t2 = fp2_mul(&p->z, &q->z);
  fp2 t3 = fp2_add(&p->x, &p->y), t4 = fp2_add(&q->x, &q->y);
  t3 = fp2_mul(&t3, &t4); t4 = fp2_add(&t0, &t1); t3 = fp2_sub(&t3, &t4);
  t4 = fp2_add(&p->y, &p->z);
  fp2 x3 = fp2_add(&q->y, &q->z);
  t4 = fp2_mul(&t4, &x3); x3 = fp2_add(&t1, &t2); t4 = fp2_sub(&t4, &x3);
  x3 = fp2_add(&p->x, &p->z);
  fp2 y3 = fp2_add(&q->x, &q->z);
  x3 = fp2_mul(&x3, &y3); y3 = fp2_add(&t0, &t2); y3 = fp2_sub(&x3, &y3);
  x3 = fp2_dbl(&t0); t0 = fp2_add(&x3, &t0);
  t2 = g2_mul_by_3b(t2);
  fp2 z3 = fp2_add(&t1, &t2); t1 = fp2_sub(&t1, &t2);
  y3 = g2_mul_by_3b(y3);
  x3 = fp2_mul(&t4, &y3); t2 = fp2_mul(&t3, &t1); x3 = fp2_sub(&t2, &x3);
  y3 = fp2_mul(&y3, &t0); t1 = fp2_mul(&t1, &z3); y3 = fp2_add(&t1, &y3);
  t0 = fp2_mul(&t0, &t3); z3 = fp2_mul(&z3, &t4); z3 = fp2_add(&z3, &t0);
  g2p r = {x3, y3, z3};
  return r;
}
static g2p g2_multiply(const g2p* p, const uint8_t by[32]) {
  g2p acc = g2_identity();
  int first = 1;
  for (int byte = 31; byte >= 0; byte--)
    for (int i = 7; i >= 0; i--) {
      if (first) { first = 0; continue; }
      acc = g2_double(&acc);
      g2p s = g2_add(&acc, p);
      if ((by[byte] >> i) & 1) acc = s;
    }
  return acc;
}
/* sum_i P_i * s_i over G2 (g2.rs:162-172, :626-632); xy: n x 24 limbs; out: 36 limbs projective */
int ora_g2_msm(const u64* xy, const uint8_t* inf, const uint8_t* scalars, long n, int threads, u64* out_xyz) {
  int used = 1;
  g2p total = g2_identity();
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  if (threads > 1024) threads = 1024;
  used = threads;
  static g2p part[1024];
#pragma omp parallel num_threads(threads)
  {
    g2p acc = g2_identity();
#pragma omp for schedule(dynamic, 16)
    for (long i = 0; i < n; i++) {
      g2p p; memcpy(&p.x, xy + 24 * i, 96); memcpy(&p.y, xy + 24 * i + 12, 96);
      p.z = (inf && inf[i]) ? FP2_ZERO_C : fp2_one();
      g2p m = g2_multiply(&p, scalars + 32 * i);
      acc = g2_add(&acc, &m);
    }
    part[omp_get_thread_num()] = acc;
  }
  for (int t = 0; t < threads; t++) total = g2_add(&total, &part[t]);
#else
  (void)threads;
  for (long i = 0; i < n; i++) {
    g2p p; memcpy(&p.x, xy + 24 * i, 96); memcpy(&p.y, xy + 24 * i + 12, 96);
    p.z = (inf && inf[i]) ? FP2_ZERO_C : fp2_one();
    g2p m = g2_multiply(&p, scalars + 32 * i);
    total = g2_add(&total, &m);
  }
#endif
  memcpy(out_xyz, &total, 288);
  return used;
}
int ora_g2_to_affine(const u64* xyz, u64* xy) {
  const g2p* p = (const g2p*)xyz;
  if (fp2_is_zero(&p->z)) { memset(xy, 0, 192); memcpy(xy + 12, FP_ONE.l, 48); return 1; }
  fp2 zi = fp2_inv(&p->z), x = fp2_mul(&p->x, &zi), y = fp2_mul(&p->y, &zi);
  memcpy(xy, &x, 96); memcpy(xy + 12, &y, 96); return 0;
}
/* n independent `&G2Affine * &Scalar` (g2.rs:626-632, 825-845), each converted to affine */
int ora_g2_mul_batch_affine(const u64* xy, const uint8_t* inf, const uint8_t* scalars, long n, int threads, u64* out_xy, uint8_t* out_inf) {
  int used = 1;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
  used = threads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
  for (long i = 0; i < n; i++) {
    g2p p; memcpy(&p.x, xy + 24 * i, 96); memcpy(&p.y, xy + 24 * i + 12, 96);
    p.z = (inf && inf[i]) ? FP2_ZERO_C : fp2_one();
    g2p m = g2_multiply(&p, scalars + 32 * i);
    out_inf[i] = (uint8_t)ora_g2_to_affine((const u64*)&m, out_xy + 24 * i);
  }
  return used;
}

/* ---- `Scalar::to_bytes` (scalar.rs:284-296): montgomery_reduce(l0..l3, 0, 0, 0, 0) (scalar.rs:506-550) then `sub(&MODULUS)` (:420-432).
 * What a host caller does per scalar before it can hand bytes to an MSM; bench.py times it on one thread next to the device-side
 * conversion. ------------------------------------------------------------------------------------------------- */
static const u64 FR_MODULUS[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};   /* scalar.rs:76-81 */
static const u64 FR_INV = 0xfffffffeffffffffull;                                                                                  /* scalar.rs:156 */
static void fr_montgomery_reduce(const u64* in8, u64* out4) {
  u64 r[8]; memcpy(r, in8, 64);
  u64 carry2 = 0;
  for (int i = 0; i < 4; i++) {
    u64 k = r[i] * FR_INV, carry = 0;
    (void)mac(r[i], k, FR_MODULUS[0], &carry);
    for (int j = 1; j < 4; j++) r[i + j] = mac(r[i + j], k, FR_MODULUS[j], &carry);
    r[i + 4] = adc(r[i + 4], carry2, &carry);
    carry2 = carry;
  }
  /* (&Scalar([r4, r5, r6, r7])).sub(&MODULUS) */
  u64 d[4], bw = 0, c = 0;
  for (int i = 0; i < 4; i++) d[i] = sbb(r[4 + i], FR_MODULUS[i], &bw);
  for (int i = 0; i < 4; i++) out4[i] = adc(d[i], FR_MODULUS[i] & bw, &c);
}
void ora_scalar_to_bytes_batch(const u64* limbs, long n, uint8_t* out) {
  for (long i = 0; i < n; i++) {
    u64 in8[8] = {limbs[4 * i], limbs[4 * i + 1], limbs[4 * i + 2], limbs[4 * i + 3], 0, 0, 0, 0}, t[4];
    fr_montgomery_reduce(in8, t);
    memcpy(out + 32 * i, t, 32);            /* to_le_bytes of each limb (little-endian host) */
  }
}
