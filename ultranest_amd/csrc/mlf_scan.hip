// mlf_scan.hip -- the neighbour-scan kernel behind K1 (find_nearby, reference mlfriends.pyx:143),
// K2 (count_nearby :31), pass 1 of K3 (_subtract_nearby :73) and the scan stage of
// MLFriends.inside (:1186).
//
// Mapping (CDNA4, wave64): one LANE owns one LIVE POINT, held in registers for the whole pass
// over the workgroup's queries; the 64 queries of a workgroup sit in LDS and are broadcast to
// the wave (every lane reads the same 16 bytes -> conflict-free ds_read_b128).  A wave therefore
// evaluates 64 squared distances per query step and decides "any within r2" with one ballot;
// the lowest set bit of the first hitting tile is the reference's first-hit index.
// Early exit is per QUERY (not per lane): a query that has its answer is skipped by all four
// waves of the workgroup, so accepted proposals stop costing anything after their first hit.
//
// Arithmetic contract (bit-exact with the reference): acc = 0; for k ascending:
// diff = live[k] - query[k]; acc += diff*diff  with sub, mul, add individually rounded --
// this file is compiled with -ffp-contract=off so no v_fma_f64 is formed.
#include "mlf_common.hpp"

namespace mlf {

// QB = queries per workgroup: 64 for large batches; 16 when the batch is so small that 64-query workgroups would
// leave most CUs idle (the rebuild scans its 4000 live points against themselves: 63 workgroups for 256 CUs)
// EXTRA: the second-stage uses behind the MFMA pre-filter (routing bytes, in-place whitening of the staged rows, the
// finalise tail).  They are compiled into their own instance: with them in the one kernel, the plain scan needed 215
// instead of 128 VGPRs and ran with two instead of four waves per SIMD (exact scan 30 -> 23 TFLOP/s).
template <int DP, int QB, bool EXTRA>
__global__ __launch_bounds__(kScanThreads) void k_scan(ScanArgs a) {
  constexpr int kScanQB = QB;   // shadows the global default inside this kernel
  __shared__ __attribute__((aligned(16))) double qs[kScanQB * DP];
  __shared__ int state[kScanQB];  // SCAN_FIRST/MASK: first-hit index or kNone; -1 = inactive
  __shared__ int cnt[kScanQB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  __shared__ int any_active;
  if (EXTRA && a.fin_best && blockIdx.x >= a.fin_grid0) {   // tail: answers of the filtered queries, 4 per thread
    const long long p0 = (((long long)blockIdx.x - a.fin_grid0) * kScanThreads + tid) * 4;
    if (p0 == 0 && a.fin_reset) *a.fin_reset = 0u;
    if (p0 == 0 && a.fin_slots) {   // every filter launch of the batch is over: the compaction counter goes back to zero
      *a.fin_groups = (*a.fin_slots + 31u) / 32u;
      *a.fin_slots = 0u;
    }
    if (p0 == 0 && a.fin_slots2) {   // ... and the counter of the uncertain set (kept as a statistic)
      a.fin_groups[1] = *a.fin_slots2;
      *a.fin_slots2 = 0u;
    }
    if (p0 == 0 && a.fin_slots3) {
      a.fin_groups[5] = (*a.fin_slots3 + 31u) / 32u;
      *a.fin_slots3 = 0u;
    } else if (p0 == 0 && a.fin_slots2) {   // a two-range batch of the min-only path: no stale third-range statistic
      a.fin_groups[5] = 0u;
    }
    if (p0 >= a.nq) return;
    const bool overflowed = a.counters[1] != 0u;
    int rt[4], b[4];
    if (p0 + 3 < a.nq) {   // the arrays are allocation-aligned and p0 is a multiple of 4: one vector load each
      const uchar4 r4 = *reinterpret_cast<const uchar4 *>(a.route + p0);
      const int4 b4 = *reinterpret_cast<const int4 *>(a.fin_best + p0);
      rt[0] = r4.x, rt[1] = r4.y, rt[2] = r4.z, rt[3] = r4.w;
      b[0] = b4.x, b[1] = b4.y, b[2] = b4.z, b[3] = b4.w;
    } else {
      for (int j = 0; j < 4; ++j) {
        rt[j] = p0 + j < a.nq ? a.route[p0 + j] : 2;   // past the end: "not mine"
        b[j] = p0 + j < a.nq ? a.fin_best[p0 + j] : kNone;
      }
    }
    for (int j = 0; j < 4; ++j) {
      if (rt[j] == 2 || (rt[j] == 1 && overflowed)) continue;   // written by the scanning workgroups
      const bool found = rt[j] == 1 && b[j] != kNone;
      if (a.out_mask) a.out_mask[p0 + j] = found ? 1 : 0;
      if (a.out_idx) a.out_idx[p0 + j] = rt[j] == 0 ? -2ll : (found ? (long long)b[j] : -1ll);
    }
    return;
  }
  // no tail in this launch (the one-launch path of mlf_mid.hip writes its own answers): the flag word of the NEXT batch
  // is cleared here
  if (EXTRA && !a.fin_best && a.fin_reset && blockIdx.x == 0 && tid == 0) *a.fin_reset = 0u;
  const unsigned scan_blocks = (EXTRA && a.fin_best) ? a.fin_grid0 : gridDim.x;
  const bool overflow = EXTRA && a.route && a.counters[1] != 0u;
  if (EXTRA && a.any_flag && *a.any_flag == 0u && !overflow) return;   // nothing was routed to the exact scan
  auto gated = [&](long long j) {
    if (EXTRA && a.route) {
      const int rt = a.route[j];
      return rt == 2 || (rt == 1 && overflow);
    }
    return a.gate == nullptr || a.gate[j] != 0;
  };
  const long long nqblk = (a.nq + kScanQB - 1) / kScanQB;
  // one workgroup per query block; the gated second-stage launch is a bounded grid that strides over the blocks
  auto scan_block = [&](const long long qblk) {
  const long long q0 = qblk * kScanQB;
  const long long left = a.nq - q0;
  const int nqb = left < kScanQB ? (int)left : kScanQB;

  if (a.only_gated) {  // second-stage use behind the MFMA filter: usually nothing to do
    __syncthreads();
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < nqb && gated(q0 + tid)) any_active = 1;
    __syncthreads();
    if (!any_active) return;
  }

  // stage this workgroup's queries, zero padded to DP
  if (a.ldk <= 1) {
    for (int e = tid; e < kScanQB * DP; e += kScanThreads) {
      const int qq = e / DP;
      const int k = e - qq * DP;
      double v = 0.0;
      if (qq < nqb && k < a.d) v = a.q[(q0 + qq) * a.ldq + k];
      qs[e] = v;
    }
  } else {  // coordinate-major source: consecutive threads read consecutive queries
    for (int e = tid; e < kScanQB * DP; e += kScanThreads) {
      const int k = e / kScanQB;
      const int qq = e - k * kScanQB;
      double v = 0.0;
      if (qq < nqb && k < a.d) v = a.q[(q0 + qq) * a.ldq + (long long)k * a.ldk];
      qs[qq * DP + k] = v;
    }
  }
  if (EXTRA && a.raw_ctr) {   // rows are proposals as handed over: whiten them in place (reference arithmetic, see ScanArgs)
    constexpr int kPer = (kScanQB * DP + kScanThreads - 1) / kScanThreads;
    double tv[kPer];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kPer; ++m) {
      const int idx = tid + m * kScanThreads;
      const int qq = idx / a.d, c = idx - qq * a.d;
      double acc = 0.0;
      if (qq < kScanQB)
        for (int k = 0; k < a.d; ++k) acc = __builtin_fma(qs[qq * DP + k] - a.raw_ctr[k], a.raw_T8[(size_t)k * a.raw_ldt + c], acc);
      tv[m] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kPer; ++m) {
      const int idx = tid + m * kScanThreads;
      const int qq = idx / a.d, c = idx - qq * a.d;
      if (qq < kScanQB) qs[qq * DP + c] = tv[m];
    }
  }
  if (tid < kScanQB) {
    const bool active = tid < nqb && gated(q0 + tid);
    state[tid] = active ? kNone : -1;
    cnt[tid] = 0;
  }
  __syncthreads();

  const int mode = a.mode;
  // FLAGS launches may split the live-point tiles over gridDim.y (every (query, tile) ballot is its own output word:
  // no state crosses tiles): an all-pairs pass over a few thousand points is then 4 waves per SIMD instead of one
  const int t_lo = (int)((long long)a.ntiles * blockIdx.y / gridDim.y), t_hi = (int)((long long)a.ntiles * (blockIdx.y + 1) / gridDim.y);
  for (int t = t_lo + wave; t < t_hi; t += kScanThreads / kWave) {
    const int base = t * kWave;
    double r[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) r[k] = a.refT[(size_t)k * a.npad + base + lane];
    const bool valid = base + lane < a.n;

    for (int qq = 0; qq < nqb; ++qq) {
      const int st = __builtin_amdgcn_readfirstlane(*(volatile int *)&state[qq]);
      if (st < 0) continue;                                 // gated out
      if (mode == SCAN_FIRST && st < base) continue;        // an earlier tile already hit
      if (mode == SCAN_MASK && st != kNone) continue;       // any hit settles the mask

      const double2 *qrow = reinterpret_cast<const double2 *>(qs + qq * DP);
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DP; k += 2) {
        const double2 v = qrow[k >> 1];
        const double d0 = r[k] - v.x;
        acc += d0 * d0;
        const double d1 = r[k + 1] - v.y;
        acc += d1 * d1;
      }
      const bool hit = valid && (acc <= a.r2);
      const unsigned long long m = __ballot(hit);
      if (mode == SCAN_FLAGS) {
        if (lane == 0) a.out_flags[(size_t)(q0 + qq) * a.ntiles + t] = m;
      } else if (m != 0ull && lane == 0) {
        if (mode == SCAN_COUNT)
          atomicAdd(&cnt[qq], __popcll(m));
        else
          atomicMin(&state[qq], base + __ffsll((long long)m) - 1);
      }
    }
  }
  __syncthreads();

  if (tid < nqb && !(a.only_gated && !gated(q0 + tid))) {
    const int st = state[tid];
    const bool found = st >= 0 && st != kNone;
    if (mode == SCAN_FIRST)
      a.out_idx[q0 + tid] = found ? (long long)st : -1ll;
    else if (mode == SCAN_COUNT)
      a.out_idx[q0 + tid] = (long long)cnt[tid];
    else if (mode == SCAN_MASK)
      a.out_mask[q0 + tid] = found ? 1 : 0;
  }
  };
  if (EXTRA) {
    for (long long qblk = blockIdx.x; qblk < nqblk; qblk += scan_blocks) {
      scan_block(qblk);
      __syncthreads();   // the staged queries / states are reused by the next block of this workgroup
    }
  } else if ((long long)blockIdx.x < nqblk) {
    scan_block(blockIdx.x);   // plain scan: one workgroup per query block
  }
}

hipError_t launch_scan(int dp, const ScanArgs &a_in, hipStream_t s) {
  if (a_in.nq <= 0) return hipSuccess;
  if (wide_dims(dp)) return launch_scan_wide(dp, a_in, s);
  ScanArgs a = a_in;
  const bool small = a.nq <= 16384;
  const int qb = small ? 16 : kScanQB;
  unsigned grid = (unsigned)((a.nq + qb - 1) / qb);
  const bool extra = a.fin_best || a.route || a.any_flag || a.raw_ctr;
  // mostly idle second-stage launches: keep the dispatch short -- the workgroups of the EXTRA instance stride over the query
  // blocks.  The plain instance handles ONE block per workgroup: its grid is never capped (round 5: until then a gated launch
  // of the plain instance -- the scan behind the binary64 per-proposal stage, d = 65 ... 128, wrapped axes, prep_bounded = 0
  // -- covered only the first 512 blocks: a proposal past the first 32768 that the pre-filter had routed to the exact scan
  // kept its initial answer; tests/test_gpu_filter.py::test_exact_scan_tail_of_the_unbounded_path_covers_the_whole_batch)
  if (extra && a.only_gated && grid > 512u) grid = 512u;
  if (a.fin_best) {
    a.fin_grid0 = grid;
    grid += (unsigned)((a.nq + 4 * kScanThreads - 1) / (4 * kScanThreads));
  }
  unsigned ny = 1;
  if (a.mode == SCAN_FLAGS && !extra) {   // aim at ~4 waves per SIMD (4096 waves), at least 4 tiles per range
    ny = 4096u / (grid * 4u > 0u ? grid * 4u : 1u);
    if (ny > (unsigned)(a.ntiles / 4)) ny = (unsigned)(a.ntiles / 4);
    if (ny < 1u) ny = 1u;
    if (ny > 16u) ny = 16u;
  }

  switch (dp) {
#define X(D)                                                                          \
  case D:                                                                             \
    if (extra) {                                                                      \
      if (small)                                                                      \
        hipLaunchKernelGGL((k_scan<D, 16, true>), dim3(grid), dim3(kScanThreads), 0, s, a);      \
      else                                                                            \
        hipLaunchKernelGGL((k_scan<D, kScanQB, true>), dim3(grid), dim3(kScanThreads), 0, s, a); \
    } else if (small) {                                                               \
      hipLaunchKernelGGL((k_scan<D, 16, false>), dim3(grid, ny), dim3(kScanThreads), 0, s, a);       \
    } else {                                                                          \
      hipLaunchKernelGGL((k_scan<D, kScanQB, false>), dim3(grid, ny), dim3(kScanThreads), 0, s, a);  \
    }                                                                                 \
    break;
    MLF_FOR_EACH_DP(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mlf
