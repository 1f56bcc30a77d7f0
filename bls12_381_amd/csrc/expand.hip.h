// expand.hip.h -- the message expanders of hash-to-curve other than XMD:SHA-256, and `HashToField for Scalar`.
//
// Reference: /root/reference/src/hash_to_curve/expand_msg.rs -- `ExpandMsgXmd<H>` is generic in the digest (:230-328; the reference's
// tests run it over Sha256 and Sha512), `ExpandMsgXof<H>` over an extendable-output function (:167-228; tests: Shake128, Shake256), DST
// reduction for tags longer than 255 bytes :47-95; map_scalar.rs:10-25 (`HashToField for Scalar`: 48 bytes -> from_bytes_wide).
// SHA-512 is FIPS 180-4, SHAKE128 / SHAKE256 are FIPS 202 (Keccak-f[1600], rates 168 / 136 bytes, domain suffix 0x1F).
//
// The fused SHA-256 kernels of h2c.hip.h stay as they are (the BLS-signature suites, the bulk-verification chain).  For the other
// expanders a message is expanded by k_expand_message into a buffer of uniform bytes, from which k_hash_to_curve_uniform (h2c.hip.h) maps
// to the curve and k_hash_to_scalar (here) to Fr -- one lane per message, byte-oriented streaming: these stages are a few microseconds
// of a hash whose square roots cost thousands of field multiplications.
#pragma once
#include "scalar.hip.h"

namespace bls {

#ifndef HD
#define HD __host__ __device__ inline
#endif

constexpr int EXPAND_XMD_SHA256 = 0, EXPAND_XMD_SHA512 = 1, EXPAND_XOF_SHAKE128 = 2, EXPAND_XOF_SHAKE256 = 3;

// ---- SHA-256 (FIPS 180-4), byte-oriented streaming; also used on the host to shorten an oversize DST ---------
struct Sha256 {
  u32 h[8];
  u32 w[16];
  u32 fill;            // bytes in w
  u64 total;           // bytes absorbed
};
HD u32 sha_rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
HD void sha256_compress(u32* h, const u32* blk) {
  constexpr u32 K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
      0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
      0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
      0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
      0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
      0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  u32 w[16];
  for (int i = 0; i < 16; i++) w[i] = blk[i];
  u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
      u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    u32 S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = hh + S1 + ch + K[i] + w[i & 15];
    u32 S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
    u32 mj = (a & b) ^ (a & c) ^ (b & c);
    u32 t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
HD void sha_init(Sha256& s) {
  const u32 iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  for (int i = 0; i < 8; i++) s.h[i] = iv[i];
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0; s.total = 0;
}
HD void sha_put(Sha256& s, uint8_t b) {
  s.w[s.fill >> 2] |= (u32)b << (24 - 8 * (s.fill & 3));
  s.fill++; s.total++;
  if (s.fill == 64) { sha256_compress(s.h, s.w); for (int i = 0; i < 16; i++) s.w[i] = 0; s.fill = 0; }
}
HD void sha_put_words(Sha256& s, const u32* words, int n) {          // n big-endian words
  for (int i = 0; i < n; i++) for (int k = 0; k < 4; k++) sha_put(s, (uint8_t)(words[i] >> (24 - 8 * k)));
}
HD void sha_finish(Sha256& s, u32* out) {                             // 8 big-endian words
  const u64 bits = s.total * 8;
  sha_put(s, 0x80);
  while (s.fill != 56) sha_put(s, 0);
  for (int k = 7; k >= 0; k--) sha_put(s, (uint8_t)(bits >> (8 * k)));
  for (int i = 0; i < 8; i++) out[i] = s.h[i];
}

// ---- SHA-512 (FIPS 180-4), byte-oriented streaming ----------------------------------------------------------------------------
struct Sha512 {
  u64 h[8];
  u64 w[16];
  u32 fill;            // bytes in w
  u64 total;           // bytes absorbed (messages here are far below 2^61 bytes: the high length word is zero)
};
HD u64 sha512_rotr(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
HD void sha512_compress(u64* h, const u64* blk) {
  constexpr u64 K[80] = {
      0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
      0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
      0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
      0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
      0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
      0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
      0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
      0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
      0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
      0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
      0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
      0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
      0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
      0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
  u64 w[16];
  for (int i = 0; i < 16; i++) w[i] = blk[i];
  u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 80; i++) {
    if (i >= 16) {
      const u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const u64 s0 = sha512_rotr(w15, 1) ^ sha512_rotr(w15, 8) ^ (w15 >> 7);
      const u64 s1 = sha512_rotr(w2, 19) ^ sha512_rotr(w2, 61) ^ (w2 >> 6);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    const u64 S1 = sha512_rotr(e, 14) ^ sha512_rotr(e, 18) ^ sha512_rotr(e, 41);
    const u64 ch = (e & f) ^ (~e & g);
    const u64 t1 = hh + S1 + ch + K[i] + w[i & 15];
    const u64 S0 = sha512_rotr(a, 28) ^ sha512_rotr(a, 34) ^ sha512_rotr(a, 39);
    const u64 mj = (a & b) ^ (a & c) ^ (b & c);
    const u64 t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
HD void sha512_init(Sha512& s) {
  const u64 iv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                     0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
  for (int i = 0; i < 8; i++) s.h[i] = iv[i];
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0; s.total = 0;
}
HD void sha512_put(Sha512& s, uint8_t b) {
  s.w[s.fill >> 3] |= (u64)b << (56 - 8 * (s.fill & 7));
  s.fill++; s.total++;
  if (s.fill == 128) { sha512_compress(s.h, s.w); for (int i = 0; i < 16; i++) s.w[i] = 0; s.fill = 0; }
}
HD void sha512_finish(Sha512& s, uint8_t* out64) {
  const u64 bits = s.total * 8;
  sha512_put(s, 0x80);
  while (s.fill != 112) sha512_put(s, 0);
  for (int k = 0; k < 8; k++) sha512_put(s, 0);                       // high 64 bits of the 128-bit length
  for (int k = 7; k >= 0; k--) sha512_put(s, (uint8_t)(bits >> (8 * k)));
  for (int i = 0; i < 64; i++) out64[i] = (uint8_t)(s.h[i >> 3] >> (56 - 8 * (i & 7)));
}

// ---- SHAKE128 / SHAKE256 (FIPS 202), byte-oriented streaming -------------------------------------------------------------------
struct Shake {
  u64 a[25];
  u32 rate;            // 168 (SHAKE128) or 136 (SHAKE256) bytes
  u32 pos;             // absorb / squeeze position inside the rate
};
HD u64 keccak_rotl(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
HD void keccak_f1600(u64* a) {
  constexpr u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                          0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                          0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                          0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};      // rotation of lane x + 5 y
  for (int round = 0; round < 24; round++) {
    u64 c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ keccak_rotl(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    // rho + pi: B[y, 2x + 3y] = rot(A[x, y])
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rotl(a[x + 5 * y], ROT[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[round];
  }
}
HD void shake_init(Shake& s, int bits) {
  for (int i = 0; i < 25; i++) s.a[i] = 0;
  s.rate = bits == 128 ? 168u : 136u;
  s.pos = 0;
}
HD void shake_put(Shake& s, uint8_t b) {
  s.a[s.pos >> 3] ^= (u64)b << (8 * (s.pos & 7));
  if (++s.pos == s.rate) { keccak_f1600(s.a); s.pos = 0; }
}
HD void shake_finish(Shake& s) {                                     // pad10*1 with the XOF suffix, switch to squeezing
  s.a[s.pos >> 3] ^= (u64)0x1F << (8 * (s.pos & 7));
  s.a[(s.rate - 1) >> 3] ^= (u64)0x80 << (8 * ((s.rate - 1) & 7));
  keccak_f1600(s.a);
  s.pos = 0;
}
HD uint8_t shake_get(Shake& s) {
  if (s.pos == s.rate) { keccak_f1600(s.a); s.pos = 0; }
  const uint8_t b = (uint8_t)(s.a[s.pos >> 3] >> (8 * (s.pos & 7)));
  s.pos++;
  return b;
}

// ---- the expanders (dst already <= 255 bytes: longer tags are reduced by the host, h2c_reduce_dst) ------------------------------------
// expand_msg.rs:247-328 with H = SHA-512: ell = ceil(len / 64) blocks, Z_pad = one 128-byte block of zeros
HD void expand_xmd_sha512(const uint8_t* msg, size_t mlen, const uint8_t* dst, u32 dlen, u32 len_in_bytes, uint8_t* out) {
  Sha512 s;
  uint8_t b0[64], bi[64];
  sha512_init(s);
  for (int i = 0; i < 128; i++) sha512_put(s, 0);
  for (size_t i = 0; i < mlen; i++) sha512_put(s, msg[i]);
  sha512_put(s, (uint8_t)(len_in_bytes >> 8)); sha512_put(s, (uint8_t)len_in_bytes); sha512_put(s, 0);
  for (u32 i = 0; i < dlen; i++) sha512_put(s, dst[i]);
  sha512_put(s, (uint8_t)dlen);
  sha512_finish(s, b0);
  const u32 ell = (len_in_bytes + 63) / 64;
  for (u32 k = 1; k <= ell; k++) {
    sha512_init(s);
    for (int j = 0; j < 64; j++) sha512_put(s, k == 1 ? b0[j] : (uint8_t)(b0[j] ^ bi[j]));
    sha512_put(s, (uint8_t)k);
    for (u32 i = 0; i < dlen; i++) sha512_put(s, dst[i]);
    sha512_put(s, (uint8_t)dlen);
    sha512_finish(s, bi);
    for (u32 j = 0; j < 64 && 64 * (k - 1) + j < len_in_bytes; j++) out[64 * (k - 1) + j] = bi[j];
  }
}
// the same with H = SHA-256 (what the fused kernels of h2c.hip.h compute in registers), as bytes: for blsgpu_expand_message* and hash_to_scalar
HD void expand_xmd_sha256(const uint8_t* msg, size_t mlen, const uint8_t* dst, u32 dlen, u32 len_in_bytes, uint8_t* out) {
  Sha256 s;
  u32 b0[8], bi[8];
  sha_init(s);
  for (int i = 0; i < 64; i++) sha_put(s, 0);
  for (size_t i = 0; i < mlen; i++) sha_put(s, msg[i]);
  sha_put(s, (uint8_t)(len_in_bytes >> 8)); sha_put(s, (uint8_t)len_in_bytes); sha_put(s, 0);
  for (u32 i = 0; i < dlen; i++) sha_put(s, dst[i]);
  sha_put(s, (uint8_t)dlen);
  sha_finish(s, b0);
  const u32 ell = (len_in_bytes + 31) / 32;
  for (u32 k = 1; k <= ell; k++) {
    u32 x[8];
    for (int j = 0; j < 8; j++) x[j] = k == 1 ? b0[j] : (b0[j] ^ bi[j]);
    sha_init(s);
    sha_put_words(s, x, 8);
    sha_put(s, (uint8_t)k);
    for (u32 i = 0; i < dlen; i++) sha_put(s, dst[i]);
    sha_put(s, (uint8_t)dlen);
    sha_finish(s, bi);
    for (u32 j = 0; j < 32 && 32 * (k - 1) + j < len_in_bytes; j++) out[32 * (k - 1) + j] = (uint8_t)(bi[j >> 2] >> (24 - 8 * (j & 3)));
  }
}
// expand_msg.rs:184-213: H(msg || I2OSP(len, 2) || DST || I2OSP(len(DST), 1)) read as an extendable output
HD void expand_xof(int bits, const uint8_t* msg, size_t mlen, const uint8_t* dst, u32 dlen, u32 len_in_bytes, uint8_t* out) {
  Shake s;
  shake_init(s, bits);
  for (size_t i = 0; i < mlen; i++) shake_put(s, msg[i]);
  shake_put(s, (uint8_t)(len_in_bytes >> 8)); shake_put(s, (uint8_t)len_in_bytes);
  for (u32 i = 0; i < dlen; i++) shake_put(s, dst[i]);
  shake_put(s, (uint8_t)dlen);
  shake_finish(s);
  for (u32 i = 0; i < len_in_bytes; i++) out[i] = shake_get(s);
}

// expand_msg.rs:47-95: a tag longer than 255 bytes is replaced by H("H2C-OVERSIZE-DST-" || DST) -- the digest itself for XMD (32 / 64 bytes), the
// first 32 bytes of the output for XOF (`L` = 32 at the 128-bit level of every BLS12-381 suite).  Host side; returns the new length.
inline u32 h2c_reduce_dst(int expander, const uint8_t* dst, size_t dst_len, uint8_t* out /* >= 255 bytes */) {
  if (dst_len <= 255) { for (size_t i = 0; i < dst_len; i++) out[i] = dst[i]; return (u32)dst_len; }
  const char* salt = "H2C-OVERSIZE-DST-";
  if (expander == EXPAND_XMD_SHA256) {
    Sha256 sh; sha_init(sh);
    for (int i = 0; salt[i]; i++) sha_put(sh, (uint8_t)salt[i]);
    for (size_t i = 0; i < dst_len; i++) sha_put(sh, dst[i]);
    u32 hw[8]; sha_finish(sh, hw);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(hw[i >> 2] >> (24 - 8 * (i & 3)));
    return 32;
  }
  if (expander == EXPAND_XMD_SHA512) {
    Sha512 sh; sha512_init(sh);
    for (int i = 0; salt[i]; i++) sha512_put(sh, (uint8_t)salt[i]);
    for (size_t i = 0; i < dst_len; i++) sha512_put(sh, dst[i]);
    sha512_finish(sh, out);
    return 64;
  }
  Shake s; shake_init(s, expander == EXPAND_XOF_SHAKE128 ? 128 : 256);
  for (int i = 0; salt[i]; i++) shake_put(s, (uint8_t)salt[i]);
  for (size_t i = 0; i < dst_len; i++) shake_put(s, dst[i]);
  shake_finish(s);
  for (int i = 0; i < 32; i++) out[i] = shake_get(s);
  return 32;
}

}  // namespace bls
