// api_aux.hip -- the rows either side of the hot path: Fr vectors and the transform, hash-to-curve, point codecs, bulk BLS verification.
#define BLS_TU_NAME "api_aux.hip"
#include "host.h"
#include "fr.hip.h"
#include "h2c.hip.h"
#include "expand_kernels.hip.h"
#include "codec.hip.h"
#include "generators.hip.h"

using namespace bls;

// ---------------------------------------------------------------------------------------------------
// hash-to-curve (h2c.hip.h)
// ---------------------------------------------------------------------------------------------------
// one launch of the batched hash: group 1 = one lane per message, group 2 = one lane pair; batches that leave the chip under-filled take
// the split form (two lane groups per message, h2c.hip.h) -- up to 2^15 messages to G1 (<= 1 024 wavefronts of 64 lanes at one per SIMD),
// up to 2^14 to G2 (4 lanes each: 1 024 wavefronts).  Measured on MI355X, 2^14 32-byte messages: see DESIGN.md 4.8.
// uniform bytes of n messages into c->h2c_uniform (any expander; k_expand_message)
static int expand_launch(blsgpu_ctx* c, int expander, const uint8_t* msgs, const unsigned long long* offs, size_t n, const uint8_t* dst, u32 dlen, u32 len_in_bytes, uint8_t* out) {
  KLAUNCH(k_expand_message, dim3(nblk(n, 64)), dim3(64), 0, c->stream, expander, msgs, offs, n, dst, dlen, len_in_bytes, out);
  LAUNCHCHK();
  return BLSGPU_OK;
}
static int h2c_launch(blsgpu_ctx* c, int group, int expander, const uint8_t* msgs, const unsigned long long* offs, size_t n, const uint8_t* dst, u32 dlen, int encode_only, u32* out) {
  if (expander != EXPAND_XMD_SHA256) {
    // the reference's other expanders (expand_msg.rs:167-328 over SHA-512 / SHAKE128 / SHAKE256): expand, then map from the uniform bytes
    const u32 len = (u32)((encode_only ? 1 : 2) * (group == 1 ? 1 : 2) * 64);
    if (c->h2c_uniform.reserve(n * (size_t)len)) { g_err = "hipMalloc(uniform bytes) failed"; return BLSGPU_ERR_HIP; }
    if (int rc = expand_launch(c, expander, msgs, offs, n, dst, dlen, len, c->h2c_uniform.as<uint8_t>())) return rc;
    if (group == 1) KLAUNCH(k_hash_to_curve_uniform<FpPolicy>, dim3(nblk(n, 64)), dim3(64), 0, c->stream, c->h2c_uniform.as<uint8_t>(), n, encode_only ? 1 : 0, out);
    else KLAUNCH(k_hash_to_curve_uniform<Fp2PairPolicy>, dim3(nblk(n * 2, 256)), dim3(256), 0, c->stream, c->h2c_uniform.as<uint8_t>(), n, encode_only ? 1 : 0, out);
    return BLSGPU_OK;
  }
  const int forced = c->h2c_split;
  const bool split = !encode_only && (forced >= 0 ? forced == 1 : n <= (group == 1 ? (size_t)1 << 15 : (size_t)1 << 14));
  if (group == 1) {
    if (split) KLAUNCH(k_hash_to_curve_split<FpPolicy>, dim3(nblk(n * 2, 64)), dim3(64), 0, c->stream, msgs, offs, n, dst, dlen, out);
    else KLAUNCH(k_hash_to_curve<FpPolicy>, dim3(nblk(n, 64)), dim3(64), 0, c->stream, msgs, offs, n, dst, dlen, encode_only ? 1 : 0, out);
  } else {
    if (split) KLAUNCH(k_hash_to_curve_split<Fp2PairPolicy>, dim3(nblk(n * 4, 256)), dim3(256), 0, c->stream, msgs, offs, n, dst, dlen, out);
    else KLAUNCH(k_hash_to_curve<Fp2PairPolicy>, dim3(nblk(n * 2, 256)), dim3(256), 0, c->stream, msgs, offs, n, dst, dlen, encode_only ? 1 : 0, out);
  }
  return BLSGPU_OK;
}
template <class F>
static int h2c_host(blsgpu_ctx* c, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len, int encode_only,
                    uint64_t* out) {
  if (expander < EXPAND_XMD_SHA256 || expander > EXPAND_XOF_SHAKE256) return bad("hash_to_curve: unknown expander");
  if (!c || (n && (!offsets || !out)) || (dst_len && !dst)) return bad("hash_to_curve: NULL argument");
  if (!n) return BLSGPU_OK;
  const size_t total = (size_t)offsets[n];
  for (size_t i = 0; i < n; i++) if (offsets[i] > offsets[i + 1]) return bad("hash_to_curve: offsets must be non-decreasing");
  if (total && !msgs) return bad("hash_to_curve: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  // a DST longer than 255 bytes is replaced by H("H2C-OVERSIZE-DST-" || DST)  (expand_msg.rs:47-95)
  uint8_t d[255];
  const u32 dlen = h2c_reduce_dst(expander, dst, dst_len, d);
  constexpr int WW = Wire<F>::WORDS;
  if (c->io_a.reserve(total + 16) || c->io_b.reserve((n + 1) * 8) || c->io_c.reserve(256) || c->io_out.reserve(n * 3 * WW * 4)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (total) HIPCHK(hipMemcpyAsync(c->io_a.p, msgs, total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->io_b.p, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (dlen) HIPCHK(hipMemcpyAsync(c->io_c.p, d, dlen, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));                 // `d` lives on this stack frame
  { int rl = h2c_launch(c, GroupTag<F>::id, expander, c->io_a.as<uint8_t>(), (const unsigned long long*)c->io_b.p, n, c->io_c.as<uint8_t>(), dlen, encode_only, c->io_out.as<u32>()); if (rl) return rl; }
  LAUNCHCHK();
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * 3 * WW * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
extern "C" int blsgpu_g1_hash_to_curve_batch(blsgpu_ctx* c, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                             int encode_only, uint64_t* out_xyz) { CTX_CLAIM(c);
  return h2c_host<FpPolicy>(c, EXPAND_XMD_SHA256, msgs, offsets, n, dst, dst_len, encode_only, out_xyz);
}
extern "C" int blsgpu_g2_hash_to_curve_batch(blsgpu_ctx* c, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                             int encode_only, uint64_t* out_xyz) { CTX_CLAIM(c);
  return h2c_host<Fp2Policy>(c, EXPAND_XMD_SHA256, msgs, offsets, n, dst, dst_len, encode_only, out_xyz);
}
// device-resident variant: messages, offsets (n + 1 u64) and the DST (<= 255 bytes) already in device memory
static int h2c_device(blsgpu_ctx* c, int group, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len, int encode_only, void* d_out_xyz) {
  if (!c || (n && (!d_offsets || !d_out_xyz)) || (dst_len && !d_dst)) return bad("hash_to_curve: NULL argument");
  if (dst_len > 255) return bad("hash_to_curve_device: reduce a DST longer than 255 bytes on the host first");
  if (group != 1 && group != 2) return bad("hash_to_curve: group must be 1 or 2");
  if (expander < EXPAND_XMD_SHA256 || expander > EXPAND_XOF_SHAKE256) return bad("hash_to_curve: unknown expander");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (int rc = h2c_launch(c, group, expander, (const uint8_t*)d_msgs, (const unsigned long long*)d_offsets, n, (const uint8_t*)d_dst, (u32)dst_len, encode_only, (u32*)d_out_xyz)) return rc;
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_hash_to_curve_device(blsgpu_ctx* c, int group, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len,
                                           int encode_only, void* d_out_xyz) { CTX_CLAIM(c);
  return h2c_device(c, group, EXPAND_XMD_SHA256, d_msgs, d_offsets, n, d_dst, dst_len, encode_only, d_out_xyz);
}
// ---- the reference's other expanders, uniform bytes, and `HashToField for Scalar` (expand.hip.h) ---------------------------------------
// `G1Projective::hash_to_curve::<X>` / `encode_to_curve::<X>` (hash_to_curve/mod.rs:86-108) for X = ExpandMsgXmd<Sha256 | Sha512> or
// ExpandMsgXof<Shake128 | Shake256> (expand_msg.rs:167-328)
extern "C" int blsgpu_hash_to_curve_expander_batch(blsgpu_ctx* c, int group, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len,
                                                   int encode_only, uint64_t* out_xyz) { CTX_CLAIM(c);
  if (group != 1 && group != 2) return bad("hash_to_curve: group must be 1 or 2");
  return group == 1 ? h2c_host<FpPolicy>(c, expander, msgs, offsets, n, dst, dst_len, encode_only, out_xyz) : h2c_host<Fp2Policy>(c, expander, msgs, offsets, n, dst, dst_len, encode_only, out_xyz);
}
extern "C" int blsgpu_hash_to_curve_expander_device(blsgpu_ctx* c, int group, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len,
                                                    int encode_only, void* d_out_xyz) { CTX_CLAIM(c);
  return h2c_device(c, group, expander, d_msgs, d_offsets, n, d_dst, dst_len, encode_only, d_out_xyz);
}
// the part behind the expander: caller-supplied uniform bytes -> from_okm -> map_to_curve -> (sum) -> clear_h
extern "C" int blsgpu_hash_to_curve_from_uniform_device(blsgpu_ctx* c, int group, const void* d_uniform, size_t n, int encode_only, void* d_out_xyz) { CTX_CLAIM(c);
  if (!c || (n && (!d_uniform || !d_out_xyz))) return bad("hash_to_curve_from_uniform: NULL argument");
  if (group != 1 && group != 2) return bad("hash_to_curve: group must be 1 or 2");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (group == 1) KLAUNCH(k_hash_to_curve_uniform<FpPolicy>, dim3(nblk(n, 64)), dim3(64), 0, c->stream, (const uint8_t*)d_uniform, n, encode_only ? 1 : 0, (u32*)d_out_xyz);
  else KLAUNCH(k_hash_to_curve_uniform<Fp2PairPolicy>, dim3(nblk(n * 2, 256)), dim3(256), 0, c->stream, (const uint8_t*)d_uniform, n, encode_only ? 1 : 0, (u32*)d_out_xyz);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_hash_to_curve_from_uniform_batch(blsgpu_ctx* c, int group, const uint8_t* uniform, size_t n, int encode_only, uint64_t* out_xyz) { CTX_CLAIM(c);
  if (!c || (n && (!uniform || !out_xyz))) return bad("hash_to_curve_from_uniform: NULL argument");
  if (group != 1 && group != 2) return bad("hash_to_curve: group must be 1 or 2");
  if (!n) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  const size_t per = (size_t)(encode_only ? 1 : 2) * (group == 1 ? 1 : 2) * 64, ob = n * 3 * (group == 1 ? 12 : 24) * 4;
  if (c->io_a.reserve(n * per) || c->io_out.reserve(ob)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(c->io_a.p, uniform, n * per, hipMemcpyHostToDevice, c->stream));
  if (int rc = blsgpu_hash_to_curve_from_uniform_device(c, group, c->io_a.p, n, encode_only, c->io_out.p)) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz, c->io_out.p, ob, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// `ExpandMessage::init_expand` + reading all `len_in_bytes` bytes, per message (out: n x len_in_bytes)
extern "C" int blsgpu_expand_message_device(blsgpu_ctx* c, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len, size_t len_in_bytes,
                                            void* d_out) { CTX_CLAIM(c);
  if (!c || (n && (!d_offsets || !d_out)) || (dst_len && !d_dst)) return bad("expand_message: NULL argument");
  if (expander < EXPAND_XMD_SHA256 || expander > EXPAND_XOF_SHAKE256) return bad("expand_message: unknown expander");
  if (dst_len > 255) return bad("expand_message_device: reduce a DST longer than 255 bytes on the host first");
  // expand_msg.rs:181-183, :263-268: the reference panics beyond these
  if (len_in_bytes > 65535) return bad("expand_message: len_in_bytes must not exceed 65535");
  if (expander <= EXPAND_XMD_SHA512 && (len_in_bytes + (expander == EXPAND_XMD_SHA256 ? 31 : 63)) / (expander == EXPAND_XMD_SHA256 ? 32 : 64) > 255) return bad("expand_message: more than 255 digest blocks");
  if (!n || !len_in_bytes) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  return expand_launch(c, expander, (const uint8_t*)d_msgs, (const unsigned long long*)d_offsets, n, (const uint8_t*)d_dst, (u32)dst_len, (u32)len_in_bytes, (uint8_t*)d_out);
}
// messages / offsets / DST from the host into io_a / io_b / io_c (the DST reduced if it is longer than 255 bytes); returns the DST length through *dlen
static int h2c_stage(blsgpu_ctx* c, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len, u32* dlen) {
  if (expander < EXPAND_XMD_SHA256 || expander > EXPAND_XOF_SHAKE256) return bad("unknown expander");
  const size_t total = (size_t)offsets[n];
  for (size_t i = 0; i < n; i++) if (offsets[i] > offsets[i + 1]) return bad("offsets must be non-decreasing");
  if (total && !msgs) return bad("NULL messages");
  HIPCHK(hipSetDevice(c->device));
  uint8_t d[255];
  *dlen = h2c_reduce_dst(expander, dst, dst_len, d);
  if (c->io_a.reserve(total + 16) || c->io_b.reserve((n + 1) * 8) || c->io_c.reserve(256)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (total) HIPCHK(hipMemcpyAsync(c->io_a.p, msgs, total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->io_b.p, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (*dlen) HIPCHK(hipMemcpyAsync(c->io_c.p, d, *dlen, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));                 // `d` lives on this stack frame
  return BLSGPU_OK;
}
extern "C" int blsgpu_expand_message_batch(blsgpu_ctx* c, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len, size_t len_in_bytes,
                                           uint8_t* out) { CTX_CLAIM(c);
  if (!c || (n && (!offsets || !out)) || (dst_len && !dst)) return bad("expand_message: NULL argument");
  if (!n || !len_in_bytes) return BLSGPU_OK;
  u32 dlen = 0;
  if (int rc = h2c_stage(c, expander, msgs, offsets, n, dst, dst_len, &dlen)) return rc;
  if (c->io_out.reserve(n * len_in_bytes)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (int rc = blsgpu_expand_message_device(c, expander, c->io_a.p, c->io_b.p, n, c->io_c.p, dlen, len_in_bytes, c->io_out.p)) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * len_in_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLSGPU_OK;
}
// `hash_to_field::<X, Scalar>` (mod.rs:32-49 with map_scalar.rs:10-25): `count` scalars per message as Montgomery limbs (out: n x count x 4 u64)
extern "C" int blsgpu_hash_to_scalar_device(blsgpu_ctx* c, int expander, const void* d_msgs, const void* d_offsets, size_t n, const void* d_dst, size_t dst_len, size_t count,
                                            void* d_out) { CTX_CLAIM(c);
  if (!c || (n && count && (!d_offsets || !d_out)) || (dst_len && !d_dst)) return bad("hash_to_scalar: NULL argument");
  if (count > 65535 / 48) return bad("hash_to_scalar: count * 48 must not exceed 65535 (expand_msg.rs:181-183)");
  if (!n || !count) return BLSGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->h2c_uniform.reserve(n * count * 48)) { g_err = "hipMalloc(uniform bytes) failed"; return BLSGPU_ERR_HIP; }
  if (int rc = blsgpu_expand_message_device(c, expander, d_msgs, d_offsets, n, d_dst, dst_len, count * 48, c->h2c_uniform.p)) return rc;
  KLAUNCH(k_hash_to_scalar, dim3(nblk(n * count, 256)), dim3(256), 0, c->stream, c->h2c_uniform.as<uint8_t>(), n * count, (u32*)d_out);
  LAUNCHCHK();
  return BLSGPU_OK;
}
extern "C" int blsgpu_hash_to_scalar_batch(blsgpu_ctx* c, int expander, const uint8_t* msgs, const uint64_t* offsets, size_t n, const uint8_t* dst, size_t dst_len, size_t count,
                                           uint64_t* out) { CTX_CLAIM(c);
  if (!c || (n && count && (!offsets || !out)) || (dst_len && !dst)) return bad("hash_to_scalar: NULL argument");
  if (!n || !count) return BLSGPU_OK;
  u32 dlen = 0;
  if (int rc = h2c_stage(c, expander, msgs, offsets, n, dst, dst_len, &dlen)) return rc;
  if (c->io_out.reserve(n * count * 32)) { g_err = "hipMalloc(io) failed"; return BLSGPU_ERR_HIP; }
  if (int rc = blsgpu_hash_to_scalar_device(c, expander, c->io_a.p, c->io_b.p, n, c->io_c.p, dlen, count, c->io_out.p)) return rc;
  HIPCHK(hipMemcpyAsync(out, c->io_out.p, n * count * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
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
    static_assert(FR_COLS_LOG == FR_COLS_LOG_MAX, "limits.h and fr.hip.h disagree on the largest column tile");
    if (c->diag.ntt_cols[0]) { tlog = c->diag.ntt_cols[0]; dmax = c->diag.ntt_cols[1]; block = c->diag.ntt_cols[2]; }
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
// n G2 encodings and n G1 encodings in ONE launch (bulk verification: its two decodings are independent, each is one lane per point and fills a
// quarter of the chip at 2^14 points -- one after the other on a stream they cost 3.3 + 1.9 ms, side by side 3.3 ms; a second side STREAM is not
// an answer: the runtime had put two side streams on one hardware queue).  Blocks [0, ceil(n / 128)) decode the G2 array, the rest the G1 array.
__global__ void __launch_bounds__(128) k_point_decode_both(const uint8_t* __restrict__ in2, u32* __restrict__ xy2, uint8_t* __restrict__ inf2, uint8_t* __restrict__ ok2,
                                                           const uint8_t* __restrict__ in1, u32* __restrict__ xy1, uint8_t* __restrict__ inf1, uint8_t* __restrict__ ok1,
                                                           size_t n, int mode) {
  const u32 nb = (u32)((n + blockDim.x - 1) / blockDim.x);
  if (blockIdx.x < nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) point_decode_one<Fp2Policy>(i, in2, mode, xy2, inf2, ok2);
  } else {
    const size_t i = (size_t)(blockIdx.x - nb) * blockDim.x + threadIdx.x;
    if (i < n) point_decode_one<FpPolicy>(i, in1, mode, xy1, inf1, ok1);
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
    c->h2c_split = c->diag.verify_h2c_split ? keep_split : 0;
    if (e == hipSuccess) rc = blsgpu_hash_to_curve_device(c, mode == 0 ? 2 : 1, d_msgs, d_offsets, n, d_dst, dst_len, 0, base + o_hp);
    c->h2c_split = keep_split;
    if (e == hipSuccess && !rc)
      rc = mode == 0 ? blsgpu_g2_batch_normalize_device(c, base + o_hp, n, base + o_h, h_inf) : blsgpu_g1_batch_normalize_device(c, base + o_hp, n, base + o_h, h_inf);
    if (e == hipSuccess && !rc) e = hipEventRecord(c->ev_ver_side[1], c->stream);
    c->stream = main_stream;
    if (e != hipSuccess) return fail("bls_verify_batch: side stream", e, __LINE__);
    if (rc) return rc;
  }
  // checked decoding (`from_compressed`: on the curve, in the subgroup) of both point arrays on the context's stream, which then waits for the side stream
  // (ONE launch for both arrays: k_point_decode_both)
  KLAUNCH(k_point_decode_both, dim3(2 * nblk(n, 128)), dim3(128), 0, c->stream, (const uint8_t*)(mode == 0 ? d_sig : d_pk), (u32*)(base + o_b), b_inf, b_ok,
                     (const uint8_t*)(mode == 0 ? d_pk : d_sig), (u32*)(base + o_a), a_inf, a_ok, n, 3);
  LAUNCHCHK();
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
  uint8_t d[255];
  const u32 dlen = h2c_reduce_dst(EXPAND_XMD_SHA256, dst, dst_len, d);        // expand_msg.rs:74-95
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

