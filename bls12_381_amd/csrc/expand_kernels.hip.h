// expand_kernels.hip.h -- the kernels over expand.hip.h's expanders (non-template kernels: included by api_aux.hip only).
#pragma once
#include "expand.hip.h"

namespace bls {

// out[i * len_in_bytes ..] = the uniform bytes of message i (one lane per message), any of the four expanders
__global__ void __launch_bounds__(64) k_expand_message(int expander, const uint8_t* __restrict__ msgs, const unsigned long long* __restrict__ offs, size_t n,
                                                       const uint8_t* __restrict__ dst, u32 dlen, u32 len_in_bytes, uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m = msgs + offs[i];
  const size_t mlen = (size_t)(offs[i + 1] - offs[i]);
  uint8_t* o = out + i * (size_t)len_in_bytes;
  if (expander == EXPAND_XMD_SHA256) expand_xmd_sha256(m, mlen, dst, dlen, len_in_bytes, o);
  else if (expander == EXPAND_XMD_SHA512) expand_xmd_sha512(m, mlen, dst, dlen, len_in_bytes, o);
  else expand_xof(expander == EXPAND_XOF_SHAKE128 ? 128 : 256, m, mlen, dst, dlen, len_in_bytes, o);
}

// `HashToField for Scalar` (map_scalar.rs:10-25; mod.rs:32-49 with InputLength = 48): element j of message i from the 48 big-endian bytes
// at uniform[(i * count + j) * 48 ..]: zero-extended to 64 bytes and reversed, i.e. the little-endian 512-bit integer with d0 = its
// low 256 bits and d1 = the 128 bits above -> `Scalar::from_bytes_wide` (fr_from_wide).  out: Montgomery limbs, the reference's `Scalar`.
__global__ void __launch_bounds__(256) k_hash_to_scalar(const uint8_t* __restrict__ uniform, size_t total, u32* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint8_t* okm = uniform + t * 48;
  Fr d0, d1;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    // little-endian word w of the integer = big-endian bytes okm[47 - 4w - 3 .. 47 - 4w]
    u32 lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) lo |= (u32)okm[47 - 4 * w - k] << (8 * k);
    if (w < 4) {
#pragma unroll
      for (int k = 0; k < 4; k++) hi |= (u32)okm[15 - 4 * w - k] << (8 * k);
    }
    d0.l[w] = lo; d1.l[w] = hi;
  }
  fr_store(out + t * 8, fr_from_wide(d0, d1));
}

}  // namespace bls
