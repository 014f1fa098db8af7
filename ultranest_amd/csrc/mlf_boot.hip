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
// past the wave; whether i is selected in round b is a wave-uniform bit.  The live points are
// split over blockIdx.y chunks to fill the chip (N = 4000 rows are only 63 waves); the partial
// minima meet in M[b][j] through 64-bit atomicMin on the bit patterns (order preserving for
// non-negative doubles).
//
// Compiled with -ffp-contract=off: diff = a_i[k] - b_j[k]; acc += diff*diff, k ascending.
#include "mlf_common.hpp"
#include "mlf_dpp_dev.hpp"

namespace mlf {

// How the live point i reaches all 64 lanes (it is the same for the whole wave).
//   round 2: 32-row LDS tiles, two barriers per tile; 40 % of the vector issue slots used.
//   round 3a: scalar loads (s_load_dwordx16 from refR / selmask, one 8-coordinate block ahead of its use), the values
//     entering the vector instructions as scalar operands.  Counters (profiles/r03_rebuild_pmc_summary.json): 67 % of the
//     wave-cycles in s_waitcnt -- scalar loads return out of order, so every wait is lgkmcnt(0), i.e. a wait for the
//     request issued one block (96 cycles) earlier, and a request that misses the 16 KB scalar cache (the rows stream
//     through it) takes ~430 cycles; a SIMD holds two of these waves (178 VGPRs), nine waits per live point.
//   round 3b (this kernel): VECTOR loads, three live points ahead, and the DPP row broadcast.  Lane l loads coordinate
//     16 c + (l mod 16) of the live point into register pair c: each row of 16 lanes holds the same 16 coordinates, and
//     `v_mov_b64_dpp ... row_newbcast:k` hands coordinate 16 c + k to every lane (one extra vector instruction per
//     coordinate, 4 instead of 3; no scalar or LDS traffic, vector loads return in order so that vmcnt waits only for
//     what is needed).  The 32 selection words of the live point sit in two registers the same way and enter the
//     candidate's OR through the DPP operand of `v_or_b32` itself.
// The masked minimum of round r is  mind[r] = min(mind[r], acc')  with the high word of acc' = high word of acc OR
// selmask[i][r] (0 if i is selected in round r, 0xffffffff otherwise): an unselected i turns the candidate into a quiet
// NaN, which v_min_f64 ignores -- 2 vector instructions per round where select + compare + select took 4-5.  The
// non-NaN candidates are the same binary64 values as before and min is exact: results bit-identical.
template <int NCH>
struct BootRow {        // one live point as the wave holds it: 16 coordinates per register pair, 16 selection words per register
  double x[NCH];
  unsigned mlo, mhi;
};

template <int DP>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(DP <= 64 ? 2 : 1, DP <= 64 ? 2 : 1))) void k_boot(BootArgs a) {
  static_assert(kBootGroup == 32, "two registers of 16 selection words per live point");
  constexpr int NCH = (DP + 15) / 16;
  const int lane = threadIdx.x;
  const int sub = lane & 15;
  const int j = (blockIdx.x + a.blk0) * kWave + lane;  // < npad by construction of the grid
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

  int off[NCH];   // this lane's coordinate in chunk c (the last chunk of a row repeats its last coordinate: never broadcast)
#pragma unroll
  for (int c = 0; c < NCH; ++c) off[c] = 16 * c + sub < DP ? 16 * c + sub : DP - 1;
  const int last = a.n - 1;
  auto fetch = [&](BootRow<NCH> &R, int i) __attribute__((always_inline)) {
    const int ii = i < last ? i : last;   // requests past the chunk are harmless, past the array they are not
    const double *r = a.refR + (size_t)ii * DP;
#pragma unroll
    for (int c = 0; c < NCH; ++c) R.x[c] = r[off[c]];
    const unsigned *k = a.selmask + (size_t)ii * kBootGroup;
    R.mlo = k[sub];
    R.mhi = k[16 + sub];
    // the requests stay HERE, two live points ahead of their use: without the barrier the compiler sinks these loads of
    // read-only data down to their first use (and every live point then costs a full memory latency)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto process = [&](const BootRow<NCH> &R) __attribute__((always_inline)) {
    // the reference's loop (diff, diff * diff, acc +=, k ascending), written skewed: stage s broadcasts coordinate s,
    // subtracts for s - 1, squares for s - 2 and accumulates s - 3, so that every instruction's operand was produced four
    // instructions earlier (a dependent binary64 instruction issues 8 cycles after its producer, an independent one after
    // 4; left to itself the scheduler kept the four instructions of a coordinate back to back in two of three sections)
    double acc = 0.0, xb[4], dd[4], sq[4];
    static_for<0, DP + 3>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s >= 3) acc += sq[(s - 3) & 3];
      if constexpr (s >= 2 && s - 2 < DP) sq[(s - 2) & 3] = dd[(s - 2) & 3] * dd[(s - 2) & 3];
      if constexpr (s >= 1 && s - 1 < DP) dd[(s - 1) & 3] = xb[(s - 1) & 3] - b[s - 1];
      if constexpr (s < DP) xb[s & 3] = row_bcast<(s & 15)>(R.x[s >> 4]);
      __builtin_amdgcn_sched_barrier(0);
    });
    const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    // four candidates are built before their minima are taken: a candidate's OR and its v_min_f64 are dependent
    static_for<0, kBootGroup / 4>([&](auto gc) __attribute__((always_inline)) {
      constexpr int r0 = 4 * decltype(gc)::value;
      double cand[4];
      static_for<0, 4>([&](auto qc) __attribute__((always_inline)) {
        constexpr int r = r0 + decltype(qc)::value;
        const unsigned h = row_bcast<(r & 15)>(r < 16 ? R.mlo : R.mhi) | hi;
        cand[r - r0] = __hiloint2double((int)h, (int)lo);
      });
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double m;
        asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(mind[r0 + q]), "v"(cand[q]));   // NaN candidate: the minimum stays (no canonicalisation)
        mind[r0 + q] = m;
      }
    });
  };

  // three register sets: live points i + 1 and i + 2 on their way while i is consumed (a fourth set: no change)
  BootRow<NCH> R0, R1, R2;
  fetch(R0, i_begin);
  fetch(R1, i_begin + 1);
  for (int i = i_begin;;) {
    fetch(R2, i + 2);
    process(R0);
    if (++i >= i_end) break;
    fetch(R0, i + 2);
    process(R1);
    if (++i >= i_end) break;
    fetch(R1, i + 2);
    process(R2);
    if (++i >= i_end) break;
  }

  if (j < a.n) {
#pragma unroll
    for (int r = 0; r < kBootGroup; ++r) {
      if (((selj >> r) & 1u) == 0u)
        atomicMin(&a.M[(size_t)r * a.npad + j], (unsigned long long)__double_as_longlong(mind[r]));
    }
  }
}

hipError_t launch_boot(int dp, const BootArgs &a, int nchunks, hipStream_t s, int nblocks) {
  // nblocks 64-row blocks starting at a.blk0 (default: all of them)
  const dim3 grid((unsigned)(nblocks >= 0 ? nblocks : a.npad / kWave), (unsigned)nchunks);
  if (grid.x == 0) return hipSuccess;
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
