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
/* fp.rs:613-660 (computed as a product here; same field element, same limbs) */
static inline fp fp_sqr(const fp* a) { return fp_mul(a, a); }

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
#pragma omp for schedule(static)
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
int ora_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
