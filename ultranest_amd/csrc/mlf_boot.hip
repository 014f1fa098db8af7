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

// Round 3.  The live point i a wave is working on is the same for all its lanes, so its coordinates and its 32
// selection words travel through the SCALAR unit (s_load_dwordx16 from refR / selmask, issued one 8-coordinate block
// ahead of their use) and enter the vector instructions as scalar operands: no LDS tile, no barrier, no ds_read latency
// for the one or two waves a SIMD holds (round 2: 32-row LDS tiles, two barriers per tile, 40 % of the vector issue
// slots used).  The masked minimum of round r is  mind[r] = min(mind[r], acc')  with the high word of acc' = high word
// of acc OR selmask[i][r] (0 if i is selected in round r, 0xffffffff otherwise): an unselected i turns the candidate
// into a quiet NaN, which v_min_f64 ignores -- 2 vector instructions per round where select + compare + select took
// 4-5.  The non-NaN candidates are the same binary64 values as before and min is exact: results bit-identical.
typedef int sgpr16 __attribute__((ext_vector_type(16)));
typedef int sgpr8 __attribute__((ext_vector_type(8)));
typedef int sgpr4 __attribute__((ext_vector_type(4)));

// scalar loads are asynchronous: the value may be used only behind sload_wait on the same variable
#define MLF_SLOAD(SUFFIX, dst, ptr, byteoff)                                                                \
  do {                                                                                                      \
    asm volatile("s_load_dword" SUFFIX " %0, %1, %2" : "=s"(dst) : "s"(ptr), "n"(byteoff) : "memory");      \
    __builtin_amdgcn_sched_barrier(0); /* the request stays in front of the arithmetic it is meant to overlap */ \
  } while (0)

template <class V>
__device__ __forceinline__ void sload_wait(V &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }
template <class V, class W>
__device__ __forceinline__ void sload_wait(V &v, W &w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+s"(w)); }
template <class V, class W, class X>
__device__ __forceinline__ void sload_wait(V &v, W &w, X &x) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+s"(w), "+s"(x)); }

template <class V>
__device__ __forceinline__ double sgpr_double(const V &v, int k) { return __hiloint2double(v[2 * k + 1], v[2 * k]); }

template <int DP>
__global__ __launch_bounds__(kWave) void k_boot(BootArgs a) {
  static_assert(kBootGroup == 32, "two 16-word selection blocks per live point");
  constexpr int NB = DP / 8;        // full blocks of 8 coordinates
  constexpr int TAIL = DP - 8 * NB; // 0, 2, 4 or 6 coordinates (DP is even)
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
  if (i_begin >= i_end) return;

  const double *row = a.refR + (size_t)i_begin * DP;             // wave-uniform
  const unsigned *mk = a.selmask + (size_t)i_begin * kBootGroup;
  sgpr16 blk[2];   // two coordinate blocks in flight / in use
  sgpr8 t8;
  sgpr4 t4;
  sgpr16 mlo, mhi;
  auto issue = [&](int t, const double *r) __attribute__((always_inline)) {   // block t of the row at r
    if (t < NB) {
      if (t & 1) MLF_SLOAD("x16", blk[1], r, 0); else MLF_SLOAD("x16", blk[0], r, 0);
    }
  };
  (void)issue;
  // prologue: block 0 and both selection blocks of the first row
  MLF_SLOAD("x16", mlo, mk, 0);
  MLF_SLOAD("x16", mhi, mk, 64);
  if (NB > 0) MLF_SLOAD("x16", blk[0], row, 0);
  if (NB > 0) sload_wait(blk[0], mlo, mhi); else sload_wait(mlo, mhi);

  for (int i = i_begin; i < i_end; ++i) {
    const double *nrow = row + DP;   // the rows past i_end - 1 exist (npad rows, or the next chunk's): loaded, never used
    const unsigned *nmk = mk + kBootGroup;
    double acc = 0.0;
    if (NB == 0) {   // d < 8: the row is its own tail
      if (TAIL >= 4) MLF_SLOAD("x8", t8, row, 0);
      if (TAIL == 2) MLF_SLOAD("x4", t4, row, 0);
      if (TAIL == 6) MLF_SLOAD("x4", t4, row, 32);
      if (TAIL == 2) sload_wait(t4);
      if (TAIL == 4) sload_wait(t8);
      if (TAIL == 6) sload_wait(t8, t4);
    }
    // coordinate blocks: block t is in blk[t & 1] and ready; block t + 1 is requested before block t is consumed
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      if (t + 1 < NB) {
        if ((t + 1) & 1) MLF_SLOAD("x16", blk[1], row, 64 * (t + 1)); else MLF_SLOAD("x16", blk[0], row, 64 * (t + 1));
      } else {
        if (TAIL >= 4) MLF_SLOAD("x8", t8, row, 64 * NB);
        if (TAIL == 2) MLF_SLOAD("x4", t4, row, 64 * NB);
        if (TAIL == 6) MLF_SLOAD("x4", t4, row, 64 * NB + 32);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double d0 = sgpr_double(blk[t & 1], k) - b[8 * t + k];
        acc += d0 * d0;
      }
      if (t + 1 < NB) {
        sload_wait(blk[(t + 1) & 1]);
      } else {
        if (TAIL == 2) sload_wait(t4);
        if (TAIL == 4) sload_wait(t8);
        if (TAIL == 6) sload_wait(t8, t4);
      }
    }
    // next row's first block travels during the tail coordinates and the first half of the minima
    if (NB > 0) MLF_SLOAD("x16", blk[0], nrow, 0);
    if (TAIL >= 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double d0 = sgpr_double(t8, k) - b[8 * NB + k];
        acc += d0 * d0;
      }
    }
    if (TAIL == 2 || TAIL == 6) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double d0 = sgpr_double(t4, k) - b[8 * NB + (TAIL == 6 ? 4 : 0) + k];
        acc += d0 * d0;
      }
    }
    const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
    // four candidates are built before their minima are taken: a candidate's OR and its v_min_f64 are dependent, and with
    // one temporary register pair per wave every second instruction waited for the one before it
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
      double cand[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        cand[q] = __longlong_as_double((long long)(bits | ((unsigned long long)(unsigned)mlo[r0 + q] << 32)));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double m;
        asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(mind[r0 + q]), "v"(cand[q]));   // NaN candidate: the minimum stays (no canonicalisation)
        mind[r0 + q] = m;
      }
    }
    if (NB > 0) sload_wait(blk[0]);
    MLF_SLOAD("x16", mlo, nmk, 0);
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
      double cand[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        cand[q] = __longlong_as_double((long long)(bits | ((unsigned long long)(unsigned)mhi[r0 + q] << 32)));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double m;
        asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(mind[16 + r0 + q]), "v"(cand[q]));
        mind[16 + r0 + q] = m;
      }
    }
    sload_wait(mlo);
    MLF_SLOAD("x16", mhi, nmk, 64);
    sload_wait(mhi);
    row = nrow;
    mk = nmk;
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
