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

// ---- the same pass with every pair distance computed ONCE -----------------------------------------------------------------
// k_boot computes dist2(i, j) for both orders of a pair: once in the wave that owns row j (i streaming past) and once in the
// wave that owns row i.  k_boot_sym walks the lower triangle of 64 x 64 TILES of the pair matrix, half a tile (32 live points)
// at a time: for tile (J, I), I <= J, the lanes own the rows j of block J exactly as above (distance in the reference's
// arithmetic, 32 masked minima against the selection words of i: the "j side"), and for I < J every distance is also dropped
// into an LDS tile, [i][j] with rows of 65 doubles; after the 32 live points have passed, the lanes change roles -- lane l
// owns live point i = l mod 32 of the half, reads HALF of its row of the tile (lanes 0-31: j = 0 ... 31, lanes 32-63: j = 32
// ... 63; conflict-free: 65 is odd) and takes the same 32 masked minima against the selection words of those rows j (the
// "i side"; the DPP broadcast works per row of 16 lanes, so the two halves of the wave simply load different words); the
// two halves then meet through one cross-lane exchange per round.  (x - y)^2 == (y - x)^2 exactly, min is exact: M is
// bit-identical to k_boot's.  Per unordered pair: 200 distance + 64 + 64 minima instructions + an LDS write and read,
// where two ordered pairs cost 2 x 264.  Two waves per SIMD like k_boot (one wave leaves the dependent binary64 issue slots
// empty: 5.1 instead of 4 cycles per instruction, measured with a whole tile per step); 16.6 KB of LDS per wave; runs of
// consecutive half tiles per wave (J-major, so that the row block -- 50 register pairs, 1 us to load -- is kept while J stays).
constexpr int kBootSymUnit = 64;

template <int DP>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_boot_sym(BootArgs a) {
  static_assert(kBootGroup == 32, "two registers of 16 selection words per live point");
  constexpr int NCH = (DP + 15) / 16;
  constexpr int LD = kWave + 1;
  constexpr int HALF = kBootSymUnit;          // live points per step: a whole tile (64) or half of one (32)
  constexpr int UPT = kWave / HALF;           // steps per tile
  __shared__ double tile[HALF * LD];
  const int lane = threadIdx.x;
  const int sub = lane & 15;
  const int nblk = a.npad / kWave;
  const long long nunits = (long long)nblk * (nblk + 1) / 2 * UPT;
  const long long u0 = nunits * blockIdx.x / gridDim.x, u1 = nunits * (blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;
  // unit u = UPT (J (J + 1) / 2 + I) + half
  const long long t0 = u0 / UPT;
  int J = (int)((sqrt(8.0 * (double)t0 + 1.0) - 1.0) * 0.5);
  while ((long long)(J + 1) * (J + 2) / 2 <= t0) ++J;
  while ((long long)J * (J + 1) / 2 > t0) --J;
  int I = (int)(t0 - (long long)J * (J + 1) / 2);
  int half = (int)(u0 % UPT);

  int off[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) off[c] = 16 * c + sub < DP ? 16 * c + sub : DP - 1;
  const int last = a.n - 1;
  auto fetch = [&](BootRow<NCH> &R, int i) __attribute__((always_inline)) {
    const int ii = i < last ? i : last;
    const double *r = a.refR + (size_t)ii * DP;
#pragma unroll
    for (int c = 0; c < NCH; ++c) R.x[c] = r[off[c]];
    const unsigned *k = a.selmask + (size_t)ii * kBootGroup;
    R.mlo = k[sub];
    R.mhi = k[16 + sub];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // 32 masked minima of one candidate: the high word OR the selection word (0 / 0xffffffff -> a NaN that v_min_f64 ignores)
  auto masked_min = [&](double(&mind)[kBootGroup], double acc, unsigned mlo, unsigned mhi) __attribute__((always_inline)) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    static_for<0, kBootGroup / 4>([&](auto gc) __attribute__((always_inline)) {
      constexpr int r0 = 4 * decltype(gc)::value;
      double cand[4];
      static_for<0, 4>([&](auto qc) __attribute__((always_inline)) {
        constexpr int r = r0 + decltype(qc)::value;
        const unsigned h = row_bcast<(r & 15)>(r < 16 ? mlo : mhi) | hi;
        cand[r - r0] = __hiloint2double((int)h, (int)lo);
      });
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double m;
        asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(mind[r0 + q]), "v"(cand[q]));
        mind[r0 + q] = m;
      }
    });
  };
  // (tried: reading M first and sending only values below what was read -- most partial minima lose -- : the reads go to the
  // memory side like the atomics and cost a round trip per flush, 33 us per pass instead of the atomics' 17)
  auto flush = [&](const double(&mind)[kBootGroup], int row, bool active) __attribute__((always_inline)) {
    if (active && row < a.n) {
      const unsigned selrow = a.sel[row];
#pragma unroll
      for (int r = 0; r < kBootGroup; ++r)
        if (((selrow >> r) & 1u) == 0u && mind[r] < 1e300)
          atomicMin(&a.M[(size_t)r * a.npad + row], (unsigned long long)__double_as_longlong(mind[r]));
    }
  };

  double b[DP];
  double mind[kBootGroup];
  int curJ = -1;
  // four register sets: live points i + 1 ... i + 3 on their way while i is consumed.  A half tile is 32 = 8 x 4 live
  // points, so that after it the sets hold the first three live points of the NEXT half tile when that one continues the
  // stream (same row block: the usual case) -- their latency passes behind the role-swapped part instead of in front of
  // the next half
  BootRow<NCH> R0, R1, R2, R3;
  int primed = -1;
  for (long long u = u0; u < u1; ++u) {
    {
      const int ib = I * kWave + half * HALF;
      if (primed != ib && ib < a.n) {
        fetch(R0, ib);
        fetch(R1, ib + 1);
        fetch(R2, ib + 2);
      }
    }
    if (J != curJ) {
      if (curJ >= 0) flush(mind, curJ * kWave + lane, true);
      const int j = J * kWave + lane;
#pragma unroll
      for (int k = 0; k < DP; ++k) b[k] = a.refT[(size_t)k * a.npad + j];
#pragma unroll
      for (int r = 0; r < kBootGroup; ++r) mind[r] = 1e300;
      curJ = J;
    }
    const bool both = I < J;
    const int i_begin = I * kWave + half * HALF;
    int i_end = i_begin + HALF;
    if (i_end > a.n) i_end = a.n;
    if (i_begin < i_end) {
      double *col = tile + lane;   // [i][j]: this lane's j
      auto process = [&](const BootRow<NCH> &R, int i) __attribute__((always_inline)) {
        double acc = 0.0, xb[4], dd[4], sq[4];
        static_for<0, DP + 3>([&](auto sc) __attribute__((always_inline)) {
          constexpr int s = decltype(sc)::value;
          if constexpr (s >= 3) acc += sq[(s - 3) & 3];
          if constexpr (s >= 2 && s - 2 < DP) sq[(s - 2) & 3] = dd[(s - 2) & 3] * dd[(s - 2) & 3];
          if constexpr (s >= 1 && s - 1 < DP) dd[(s - 1) & 3] = xb[(s - 1) & 3] - b[s - 1];
          if constexpr (s < DP) xb[s & 3] = row_bcast<(s & 15)>(R.x[s >> 4]);
          __builtin_amdgcn_sched_barrier(0);
        });
        if (both) col[(i - i_begin) * LD] = acc;
        masked_min(mind, acc, R.mlo, R.mhi);
      };
      for (int i = i_begin;;) {
        fetch(R3, i + 3);
        process(R0, i);
        if (++i >= i_end) break;
        fetch(R0, i + 3);
        process(R1, i);
        if (++i >= i_end) break;
        fetch(R1, i + 3);
        process(R2, i);
        if (++i >= i_end) break;
        fetch(R2, i + 3);
        process(R3, i);
        if (++i >= i_end) break;
      }
      primed = i_end - i_begin == HALF ? i_end : -1;
      if (both) {
        // roles swapped (I < J: the half holds 32 live points, all below n): lane = live point l mod 32 of the half and the
        // rows j of its half of block J
        __syncthreads();   // one wave: orders the LDS writes above against the reads below for the compiler
        double mi[kBootGroup];
#pragma unroll
        for (int r = 0; r < kBootGroup; ++r) mi[r] = 1e300;
        const int jh = lane / HALF;   // 0 when a step is a whole tile
        const int j_begin = J * kWave + jh * HALF;            // per lane half
        const double *rowp = tile + (lane & (HALF - 1)) * LD + jh * HALF;
        const unsigned *km = a.selmask + sub;
        // the selection words of four rows j and the lane's four distances at a time, the next four on their way (the
        // words come from L2); a row past the array enters as a NaN (the minimum stays), decided where the value is USED: a
        // select next to the load waits for the load
        constexpr int CH = 4;
        unsigned la[CH], ha[CH], lb[CH], hb[CH];
        double va[CH], vb[CH];
        auto cfetch = [&](unsigned(&l)[CH], unsigned(&h)[CH], double(&v)[CH], int q0) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < CH; ++q) {
            const int j = j_begin + q0 + q;
            const int jj = j < last ? j : last;
            l[q] = km[(size_t)jj * kBootGroup];
            h[q] = km[(size_t)jj * kBootGroup + 16];
            v[q] = rowp[(q0 + q) & (kWave - 1)];
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        };
        const double nan = __longlong_as_double(-1LL);
        cfetch(la, ha, va, 0);
#pragma unroll 1
        for (int q0 = 0; q0 < kWave / UPT; q0 += 2 * CH) {   // the rows of this lane: all 64, or its half
          cfetch(lb, hb, vb, q0 + CH);
#pragma unroll
          for (int q = 0; q < CH; ++q) masked_min(mi, j_begin + q0 + q < a.n ? va[q] : nan, la[q], ha[q]);
          cfetch(la, ha, va, q0 + 2 * CH);
#pragma unroll
          for (int q = 0; q < CH; ++q) masked_min(mi, j_begin + q0 + CH + q < a.n ? vb[q] : nan, lb[q], hb[q]);
        }
        if constexpr (UPT == 2) {   // the two halves of the wave hold the minima of the same 32 live points over the two halves of the rows
#pragma unroll
          for (int r = 0; r < kBootGroup; ++r) {
            const double other = __shfl_xor(mi[r], HALF);
            mi[r] = other < mi[r] ? other : mi[r];
          }
        }
        flush(mi, i_begin + (lane & (HALF - 1)), lane < HALF);
        __syncthreads();   // the tile is free again
      }
    }
    if (++half == UPT) {
      half = 0;
      if (++I > J) {
        I = 0;
        ++J;
      }
    }
  }
  flush(mind, curJ * kWave + lane, true);
}

// the lower triangle needs enough tiles to fill the chip's 1024 one-wave-per-SIMD slots, and registers for one row block
bool boot_sym_usable(int dp, int npad) {
  const long long nblk = npad / kWave;
  return dp <= 64 && nblk * (nblk + 1) / 2 >= 1024;
}

hipError_t launch_boot_sym(int dp, const BootArgs &a, hipStream_t s) {
  const long long nblk = a.npad / kWave;
  const long long nunits = nblk * (nblk + 1) / 2 * (kWave / kBootSymUnit);
  // equal runs: ceil(nunits / 1024) half tiles per wave at most (one wave per SIMD), and as many waves as that takes
  const long long per = (nunits + 1023) / 1024;
  const dim3 grid((unsigned)((nunits + per - 1) / per));
  switch (dp) {
#define X(D)                                                            \
  case D:                                                               \
    if constexpr (D <= 64) hipLaunchKernelGGL(k_boot_sym<D>, grid, dim3(kWave), 0, s, a); \
    break;
    MLF_FOR_EACH_DP(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_boot(int dp, const BootArgs &a, int nchunks, hipStream_t s, int nblocks) {
  // nblocks 64-row blocks starting at a.blk0 (default: all of them)
  const dim3 grid((unsigned)(nblocks >= 0 ? nblocks : a.npad / kWave), (unsigned)nchunks);
  if (grid.x == 0) return hipSuccess;
  if (wide_dims(dp)) return launch_boot_wide(dp, a, dp, nchunks, s, nblocks);   // padded coordinates are zero: (0 - 0)^2 adds +0.0
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
