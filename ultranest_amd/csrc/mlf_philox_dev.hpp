// mlf_philox_dev.hpp -- Philox-4x32-10 counter-based generator (Salmon et al., SC'11; Random123
// constants) and the word -> double mapping shared by the device-side samplers.  The tests pin the
// stream on the Random123 known-answer vectors.
#pragma once
#include <hip/hip_runtime.h>

namespace mlf {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// block `ctr` of stream `stream` under key `seed`
__device__ __forceinline__ void philox_block(unsigned long long seed, unsigned stream, unsigned long long ctr,
                                             unsigned out[4]) {
  philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), out);
}

// 53-bit uniform strictly inside (0, 1) from two 32-bit words
__device__ __forceinline__ double u01(unsigned hi, unsigned lo) {
  const unsigned long long m = ((unsigned long long)(hi >> 5) << 26) | (unsigned long long)(lo >> 6);
  return ((double)m + 0.5) * 0x1p-53;
}

// integer in [0, n) from one word (multiply-shift; bias < n / 2^32)
__device__ __forceinline__ unsigned below(unsigned w, unsigned n) {
  return (unsigned)(((unsigned long long)w * n) >> 32);
}

}  // namespace mlf
