// mlf_dpp_dev.hpp -- the DPP row broadcast (gfx90a+: `row_newbcast:k`, lane k of every row of 16 lanes to the whole row;
// the operand broadcast the instruction set has for binary64 matrix products) and a compile-time loop.
//
// Pattern shared by k_boot, k_boot_cov16, k_boot_solvemax and k_whiten_rows: a value that is the same for every lane
// (a coordinate of the live point a wave works on, a column of a small matrix) is kept 16 entries per register -- lane l
// holds entry 16 c + (l mod 16) of chunk c, so every row of 16 lanes holds the same 16 entries -- and reaches all lanes
// either through `v_mov_b64_dpp` (one extra vector instruction) or directly as the first operand of `v_fmac_f64`
// (inline asm: the compiler does not fold the DPP move into that instruction).  No scalar loads (they return out of
// order: every wait is lgkmcnt(0)), no LDS, no v_readlane pairs.
//
// Hazard: a DPP read of a register written by a VECTOR instruction needs two wait states.  The compiler inserts them
// for the builtins; in front of the inline-asm forms the caller ties an `s_nop 1` to the register (dpp_settle) unless the
// register comes straight from a memory instruction.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace mlf {

constexpr int kDppRowNewBcast = 0x150;   // DPP control row_newbcast:0

template <int K>
__device__ __forceinline__ double row_bcast(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, kDppRowNewBcast + K, 0xf, 0xf, true);
}
template <int K>
__device__ __forceinline__ unsigned row_bcast(unsigned x) {
  return __builtin_amdgcn_update_dpp(0u, x, kDppRowNewBcast + K, 0xf, 0xf, true);
}

// acc += entries16[lane K of the row] * y   (one fused multiply-add, as __builtin_fma)
template <int K>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double entries16, double y) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(entries16), "v"(y), "n"(K));
}

// The two halves of a wave (lanes l and l + 32) exchanged by ONE vector instruction (gfx950: v_permlane32_swap_b32 with both
// operands copies of x: the first result holds x of the LOW half in all 64 lanes, the second x of the HIGH half) -- instead
// of ds_bpermute, an LDS-crossbar round trip of 100+ cycles with dependent code behind it.
__device__ __forceinline__ void halves_of(unsigned x, unsigned &low, unsigned &high) {
  typedef unsigned uint2s __attribute__((ext_vector_type(2)));
  const uint2s r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  low = r[0];
  high = r[1];
}
// x(l) + x(l ^ 32): the bits of `x + __shfl_xor(x, 32)` in every lane (binary32 addition commutes)
__device__ __forceinline__ float half_sum32(float x) {
  unsigned lo, hi;
  halves_of(__float_as_uint(x), lo, hi);
  return __uint_as_float(lo) + __uint_as_float(hi);
}
// x of lane (l & 31) in every lane: `__shfl(x, l & 31)`
__device__ __forceinline__ float low_half32(float x) {
  unsigned lo, hi;
  halves_of(__float_as_uint(x), lo, hi);
  return __uint_as_float(lo);
}
// min over the two halves (signed compare): `min(x, __shfl_xor(x, 32))`
__device__ __forceinline__ int half_min32(int x) {
  unsigned lo, hi;
  halves_of((unsigned)x, lo, hi);
  return (int)lo < (int)hi ? (int)lo : (int)hi;
}

__device__ __forceinline__ void dpp_settle(double &x) { asm volatile("s_nop 1" : "+v"(x)); }

template <int K, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    static_for<K + 1, N>(f);
  }
}

}  // namespace mlf
