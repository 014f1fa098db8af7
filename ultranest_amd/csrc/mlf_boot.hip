// mlf_boot.hip -- K4: compute_maxradiussq (reference mlfriends.pyx:188-224) for up to 32
// bootstrap rounds at once (MLFriends.compute_enlargement :1044-1054).
//
// Every bootstrap round's (selected, unselected) distance block is a sub-block of ONE symmetric
// N x N pair-distance matrix, and (x-y)^2 == (y-x)^2 exactly in binary64, so the distances are
// computed once and reduced 32 times under selection masks -- bit-identical to the reference's
// 30 separate passes because min/max are exact and order independent.
//
// Mapping: one LANE owns one row j (a potential "unselected" point b_j), coordinates in
// registers, plus 32 running minima (one per bootstrap round).  The live points i are streamed
// through an LDS sub-tile and broadcast to the wave; whether i is selected in round b is a
// wave-uniform bit, so each masked min is a scalar select + one v_min_f64.  The live points are
// split over blockIdx.y chunks to fill the chip (N = 4000 rows are only 63 waves); the partial
// minima meet in M[b][j] through 64-bit atomicMin on the bit patterns (order preserving for
// non-negative doubles).
//
// Compiled with -ffp-contract=off: diff = a_i[k] - b_j[k]; acc += diff*diff, k ascending.
#include "mlf_common.hpp"

namespace mlf {

template <int DP>
__global__ __launch_bounds__(kWave) void k_boot(BootArgs a) {
  __shared__ __attribute__((aligned(16))) double tile[kBootTI * DP];
  __shared__ unsigned tsel[kBootTI];

  const int lane = threadIdx.x;
  const int j = blockIdx.x * kWave + lane;  // < npad by construction of the grid
  double b[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) b[k] = a.refT[(size_t)k * a.npad + j];
  const unsigned selj = a.sel[j];

  double mind[kBootGroup];
#pragma unroll
  for (int r = 0; r < kBootGroup; ++r) mind[r] = 1e300;  // reference :215

  const int i_begin = blockIdx.y * a.chunk;
  int i_end = i_begin + a.chunk;
  if (i_end > a.n) i_end = a.n;

  for (int i0 = i_begin; i0 < i_end; i0 += kBootTI) {
    __syncthreads();
    // rows i0 .. i0+TI-1 of refR are contiguous (npad >= i0+TI because chunk and npad are
    // multiples of kBootTI)
    const double *src = a.refR + (size_t)i0 * DP;
    for (int e = lane; e < kBootTI * DP; e += kWave) tile[e] = src[e];
    if (lane < kBootTI) tsel[lane] = a.sel[i0 + lane];
    __syncthreads();

    const int nt = (i_end - i0) < kBootTI ? (i_end - i0) : kBootTI;
    for (int ii = 0; ii < nt; ++ii) {
      const unsigned si = __builtin_amdgcn_readfirstlane(tsel[ii]);
      const double2 *row = reinterpret_cast<const double2 *>(tile + ii * DP);
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DP; k += 2) {
        const double2 v = row[k >> 1];
        const double d0 = v.x - b[k];
        acc += d0 * d0;
        const double d1 = v.y - b[k + 1];
        acc += d1 * d1;
      }
#pragma unroll
      for (int r = 0; r < kBootGroup; ++r) {
        const double cand = ((si >> r) & 1u) ? acc : 1e300;
        mind[r] = cand < mind[r] ? cand : mind[r];
      }
    }
  }

  if (j < a.n) {
#pragma unroll
    for (int r = 0; r < kBootGroup; ++r) {
      if (((selj >> r) & 1u) == 0u)
        atomicMin(&a.M[(size_t)r * a.npad + j], (unsigned long long)__double_as_longlong(mind[r]));
    }
  }
}

hipError_t launch_boot(int dp, const BootArgs &a, int nchunks, hipStream_t s) {
  const dim3 grid((unsigned)(a.npad / kWave), (unsigned)nchunks);
  switch (dp) {
#define X(D)                                                        \
  case D:                                                           \
    hipLaunchKernelGGL(k_boot<D>, grid, dim3(kWave), 0, s, a);      \
    break;
    MLF_FOR_EACH_DP(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mlf
