// mlf_sweepmin.hip -- the phased mask-mode sweep of the MFMA pre-filter, round 4 (R3: MLFriends.inside,
// mlfriends.pyx:1186-1211; the distance test it decides is find_nearby's, mlfriends.pyx:143-183).
//
// Same operands, thresholds and guarantees as k_sweep / k_filter (mlf_sweep.hip, mlf_filter.hip).  What
// changed is WHERE the thresholds are looked at.  k_sweep compares every lane's running minimum with (T_lo, T_hi] after
// every tile (two compares and two ballots per query group, a dozen scalar instructions and a branch per tile, and the band
// path for one tile in seven).  But a query's fate is a function of its MINIMUM Dt over all live points alone:
//     min Dt <= T_lo           certain hit (some pair is a certain hit),
//     min Dt >  T_hi           certain miss (no pair is within reach),
//     otherwise                no certain hit, at least one pair in the band: UNCERTAIN.
// So the two long launches carry nothing but the running minimum (16 matrix + 32 v_min3 instructions per tile and wave:
// the probe scripts/probes/sweep_src_probe.hip runs that loop with the matrix pipe 97-100 % occupied inside the loop,
// 88-93 % with the per-tile comparison), the minimum travels with the query from the first range to the second, and the
// uncertain queries -- the ones whose NEAREST live point sits in the band: 2.3 % at C5 -- are compacted a second time and
// handled by k_uncertain: one workgroup per set of 128 of them finds their band pairs (all tiles once more), whitens the
// queries in the reference's arithmetic and decides the pairs, all inside the launch.  The exact side no longer sees the
// band pairs of queries that have a certain hit elsewhere (3/4 of the listed pairs before).  Measured on the way: listing
// and re-checking per wave as k_sweep does for small batches (k_sweep_list with recheck_segment: 0.113 ms for the launch --
// 2048 waves of 230 registers walking ~12 pairs each through a chain of dependent loads); listing alone (26-29 us)
// followed by k_recheck_whiten (41 us with 4096 segments of ~6 pairs, 64 us with 2048 of ~12: one wave per segment, a
// chain of ~15 dependent memory round trips whatever the number of pairs) -- both no faster than what they replaced.
//
//   k_sweep_min<KS, QW, PF>   first range: slot = query; later range: compacted set + carried minima.
//                             not last: keeps the queries without a certain hit (with their minimum);
//                             last:     keeps the UNCERTAIN ones (minimum in (T_lo, T_hi]).
//   k_uncertain<KS, NCH>      resident grid, one workgroup of 8 waves per set of 128 uncertain queries: band pairs ->
//                             LDS list, queries whitened into LDS, pairs decided; the ellipsoid band of k_prep4 rides
//                             in trailing workgroups.
// Integer minima on the bit patterns: for non-negative values the order of the floats; a negative Dt (possible within
// the accumulation error) compares below every non-negative one, and T_lo is never negative and finite
// (filter_thresholds: a negative lower threshold becomes -inf), so "minimum <= T_lo" and "minimum <= T_hi" are decided
// correctly whichever negative value the integer minimum happens to keep.
#include "mlf_filter.hpp"
#include "mlf_dpp_dev.hpp"
#include "mlf_filter_dev.hpp"
#include "mlf_recheck_dev.hpp"

#include <math.h>

namespace mlf {

typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float float16v;
typedef __attribute__((ext_vector_type(4))) unsigned uint4v;

namespace {

__device__ __forceinline__ int min3i(int a, int b, int c) {
  const int m = a < b ? a : b;
  return m < c ? m : c;
}

constexpr int kPosInf = 0x7f800000;

template <int KS>
__device__ __forceinline__ void load_tile(half8 (&A)[KS], __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    union { uint4v u; half8 h; } c;
    c.u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + s * 1024, 0);
    A[s] = c.h;
  }
}

__device__ __forceinline__ int tree_min(const float16v &c, int run) {
  const int m0 = min3i(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]));
  const int m1 = min3i(__float_as_int(c[3]), __float_as_int(c[4]), __float_as_int(c[5]));
  const int m2 = min3i(__float_as_int(c[6]), __float_as_int(c[7]), __float_as_int(c[8]));
  const int m3 = min3i(__float_as_int(c[9]), __float_as_int(c[10]), __float_as_int(c[11]));
  const int m4 = min3i(__float_as_int(c[12]), __float_as_int(c[13]), __float_as_int(c[14]));
  return min3i(min3i(m0, m1, m2), min3i(m3, m4, __float_as_int(c[15])), run);
}

template <int I, int NM, int NV>
__device__ __forceinline__ void pin_step() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV * (I + 1)) / NM - (NV * I) / NM, 0);
    pin_step<I + 1, NM, NV>();
  }
}

// row stride (doubles) of the queries in k_uncertain's LDS: rows of an even d start 16-byte aligned (they arrive by LDS-DMA in
// 16-byte pieces), rows of an odd d keep the odd stride
__host__ __device__ constexpr int uncertain_row_stride(int d) { return d; }

// Arguments of a function that is NOT inlined arrive in vector registers: the compiler then treats a wave-uniform count or
// pointer as divergent (exec masks around every use, flat loads).  These put them back into scalar registers.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T *uni(T *p) {
  return reinterpret_cast<T *>(static_cast<uintptr_t>(uni((long long)reinterpret_cast<uintptr_t>(p))));
}

constexpr int min_waves(int ks, int qw, int pf) {
  const int need = 16 * qw + 4 * ks * qw + 4 * ks * (pf + 1) + 40;
  return (qw <= 2 && ks * qw <= 8) ? 4 : (need <= 128 ? 4 : (need <= 168 ? 3 : 2));
}

// workgroups of a k_sweep_min launch over a compacted set: two rounds of the 512 resident workgroups (two per CU)
constexpr long long kSweepMinSetGrid = 1024;

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int QW, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(min_waves(KS, QW, PF)))) void k_sweep_min(MinArgs a) {
  const int lane = threadIdx.x & 63;
  const long long nslots = a.nslots_dev ? (long long)*a.nslots_dev : -1;
  const long long ngroups = nslots >= 0 ? (nslots + 31) / 32 : a.ngroups;
  // slots of the compaction: ONE atomic per workgroup (the four waves share the tile order and finish together; one
  // atomic per wave on the same word -- 15 600 of them in 0.1 ms in the second range, where nearly every wave keeps a
  // query or two -- queues up at the L2: measured 0.111 against 0.092 ms for the launch)
  __shared__ unsigned wg_keep[4], wg_base;
  // diagnostics: shader-clock stamps of wave 0 of ONE workgroup (its first pass): 0 start, 1 operands and the first two tiles
  // asked for, 2 first tile done, 3 tile loop done, 4 fates known, 5 / 6 past the two barriers, 7 compaction stores issued
  bool stamp_on = a.stamps != nullptr && blockIdx.x == a.stamp_block && threadIdx.x < 64;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (stamp_on) {   // wave-uniform
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) a.stamps[k] = t;
    }
  };
  stamp(0);
  // ... and the 100 MHz clock all workgroups share: [8] / [9] start / end of workgroup 0's first pass, [10] / [11] of the selected one
  auto stamp_rt = [&](int k) __attribute__((always_inline)) {
    if (a.stamps != nullptr && threadIdx.x == 0) {
      if (blockIdx.x == 0) a.stamps[8 + k] = __builtin_amdgcn_s_memrealtime();
      if (blockIdx.x == a.stamp_block) a.stamps[10 + k] = __builtin_amdgcn_s_memrealtime();
    }
  };
  stamp_rt(0);
  // A compacted set's size is known on the device only, so its launch used to carry a workgroup for every 16 groups of the
  // BATCH: 1 953 at 10^6 proposals, of which a third range fills 510 -- the other 1 443 start, read the count and leave, two
  // or three rounds of them through the 512 resident slots behind the working ones.  For batches of that size the launcher
  // caps the grid of such a launch (kSweepMinSetGrid) and a workgroup walks the set with the grid's stride (C5: 0.082 -> 0.080
  // and 0.0571 -> 0.0567 ms for the two launches); every other launch passes this loop once.
  for (unsigned bidx = blockIdx.x, pass = 0;; bidx += gridDim.x, ++pass) {
  if ((long long)bidx * (4 * QW) >= ngroups) break;   // the whole workgroup is past the set
  if (pass) __syncthreads();                          // wg_keep / wg_base of the pass before are free
  const long long wave = (long long)bidx * 4 + (threadIdx.x >> 6);
  const long long g0 = wave * QW;
  if (g0 >= ngroups) {   // this wave is past the set: nothing to sweep, but the workgroup's two barriers are met
    if (lane == 0) wg_keep[threadIdx.x >> 6] = 0u;
    __syncthreads();
    __syncthreads();
    continue;
  }
  // (Asking for the operands BEFORE looking at the count -- the arrays are sized for every group of the batch -- saves the
  // count's round trip in the waves that have work and costs 16 KiB of reads in each of the others: the grid is sized for all
  // 31 250 groups, a third range holds 8 160: 0.062 against 0.056 ms for that launch.)

  const half8 *qF = reinterpret_cast<const half8 *>(a.qF);
  half8 bq[QW][KS];
  float tlo[QW], thi[QW];
  int run[QW], qnum[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const long long grp = (g0 + g < ngroups) ? g0 + g : ngroups - 1;   // clamp (results discarded)
#pragma unroll
    for (int s = 0; s < KS; ++s) bq[g][s] = qF[((size_t)grp * KS + s) * 64 + lane];
    const long long qi = grp * 32 + (lane & 31);
    // slots past the count of an unpadded last group: thresholds -1 AND a zero operand (Dt = |ah|^2 >= 0 > T: certain miss)
    const bool have = g0 + g < ngroups && (nslots < 0 || qi < nslots);
    tlo[g] = have ? a.tlo[qi] : -1.0f;
    thi[g] = have ? a.thi[qi] : -1.0f;
    if (!have) {
#pragma unroll
      for (int s = 0; s < KS; ++s) bq[g][s] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    // the minimum the query brought along from the ranges before (kept by the low half; +inf in the high half)
    run[g] = (a.qmin && have && lane < 32) ? a.qmin[qi] : kPosInf;
    // the query's number in the batch: asked for here, with the operands (in the epilogue it was a memory round trip at the
    // end of every wave's life)
    qnum[g] = have ? (a.qmap ? a.qmap[qi] : (int)qi) : -1;
  }

  const int ntl = a.tile1 - a.tile0;
  const int tstart = a.tile0 + (int)(((long long)bidx * 37) % ntl);
  constexpr int kTileBytes = KS * 1024;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.refF), 0, a.ntiles32 * kTileBytes, 0x00020000);
  const int voff = lane * 16;
  const int off_begin = a.tile0 * kTileBytes, off_end = a.tile1 * kTileBytes;

  float16v acc[QW];
  auto mm = [&](const half8(&A)[KS], int ga, int gb) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[ga] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[ga][s], s == 0 ? z : acc[ga], 0, 0, 0);
      if (gb >= 0) acc[gb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[gb][s], s == 0 ? z : acc[gb], 0, 0, 0);
    }
  };
  auto next_off = [&](int off) __attribute__((always_inline)) {
    const int n = off + kTileBytes;
    return n == off_end ? off_begin : n;
  };
  // one tile: matrix instructions in stages of two groups, the minima of a stage pinned between the next stage's
  // matrix instructions (8 v_min3 per group)
  auto tile = [&](const half8(&A)[KS]) __attribute__((always_inline)) {
    constexpr int NP = (QW + 1) / 2;
#pragma unroll
    for (int p = 0; p <= NP; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      if (p < NP) mm(A, 2 * p, (2 * p + 1 < QW) ? 2 * p + 1 : -1);
      if (p > 0) {
#pragma unroll
        for (int g = 2 * (p - 1); g < 2 * p && g < QW; ++g) run[g] = tree_min(acc[g], run[g]);
      }
      if (NP == 2 && p == 1) pin_step<0, (QW == 4 ? 2 : 1) * KS, 16>();
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  half8 A0[KS], A1[KS];
  if (PF == 2) {
    half8 A2[KS];
    int o0 = tstart * kTileBytes, o1 = next_off(o0), o2 = next_off(o1);
    // the two requests in THIS order, nothing moved across: the compiler interleaved them (three loads of the second tile in
    // front of the first tile's), and the wait counts of the loop -- which must also be right for its first pass -- then
    // demanded half of the NEXT tile's loads before the current tile's matrix instructions in every pass: vmcnt(6), (6), (5), (4)
    // where the steady state allows (10), (9), (9), (8)
    load_tile<KS>(A0, rsrc, voff, o0);
    __builtin_amdgcn_sched_barrier(0);
    load_tile<KS>(A1, rsrc, voff, o1);
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    for (int it = 0; it < ntl; it += 3) {
      load_tile<KS>(A2, rsrc, voff, o2);
      tile(A0);
      if (it == 0) stamp(2);
      if (it + 1 >= ntl) break;
      o0 = next_off(o2);
      load_tile<KS>(A0, rsrc, voff, o0);
      tile(A1);
      if (it + 2 >= ntl) break;
      o1 = next_off(o0);
      load_tile<KS>(A1, rsrc, voff, o1);
      tile(A2);
      o2 = next_off(o1);
    }
  } else {
    int off = tstart * kTileBytes;
    load_tile<KS>(A0, rsrc, voff, off);
    for (int it = 0; it < ntl; it += 2) {
      const int offn = next_off(off);
      load_tile<KS>(A1, rsrc, voff, offn);
      tile(A0);
      if (it + 1 >= ntl) break;
      off = next_off(offn);
      load_tile<KS>(A0, rsrc, voff, off);
      tile(A1);
    }
  }

  stamp(3);
  // ---- the queries' fates, and the compaction of those that go on
  unsigned keepm[QW];
  int qid[QW], qmn[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    keepm[g] = 0u;
    qid[g] = -1;
    qmn[g] = kPosInf;
    if (g0 + g >= ngroups) continue;
    const long long qi = qnum[g];
    const int m = half_min32(run[g]);                              // both halves hold the query's minimum now
    const float mf = __int_as_float(m);
    const bool valid = qi >= 0 && qi < a.nq && thi[g] > 0.0f;      // T_hi > 0 <=> the query has thresholds (route 1)
    const bool hit = valid && mf <= tlo[g];
    if (lane < 32 && hit) a.best[qi] = 0;
    const bool keep = valid && !hit && (a.last ? mf <= thi[g] : true);
    keepm[g] = (unsigned)__ballot(keep);   // low half; lanes l and l + 32 agree
    qid[g] = (int)qi;
    qmn[g] = m;
  }
  unsigned total = 0;
#pragma unroll
  for (int g = 0; g < QW; ++g) total += (unsigned)__popc(keepm[g]);
  if (lane == 0) wg_keep[threadIdx.x >> 6] = total;
  stamp(4);
  __syncthreads();
  stamp(5);
  if (threadIdx.x == 0) {
    const unsigned all = wg_keep[0] + wg_keep[1] + wg_keep[2] + wg_keep[3];
    wg_base = all ? atomicAdd(a.ccount, all) : 0u;
  }
  __syncthreads();
  stamp(6);
  if (total != 0u) {   // wave-uniform
    unsigned base = wg_base;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wg_keep[w];
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    uint4 *dst = reinterpret_cast<uint4 *>(a.cq);
    const unsigned row = (unsigned)(lane & 31);
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      if ((keepm[g] >> row) & 1u) {
        const unsigned rank = base + (unsigned)__popc(keepm[g] & ((1u << row) - 1u));
        const size_t gd = rank >> 5;
        const unsigned rd = (rank & 31u) + (unsigned)(lane & 32);
        if (rank < a.ccap) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            union { half8 h; uint4 u; } cv;
            cv.h = bq[g][s];
            dst[(gd * KS + s) * 64 + rd] = cv.u;
          }
        }
        if (lane < 32 && rank < a.ccap) {
          a.ctlo[rank] = tlo[g];
          a.cthi[rank] = thi[g];
          a.cmap[rank] = qid[g];
          if (a.cmin) a.cmin[rank] = qmn[g];
        }
      }
      base += (unsigned)__popc(keepm[g]);
    }
  }
  stamp(7);
  if (stamp_on || (pass == 0 && blockIdx.x == 0)) stamp_rt(1);
  stamp_on = false;
  }   // walk over the set
}

// ---------------------------------------------------------------------------------------------------------------------
// The stages of k_uncertain are functions of their own (not inlined): compiled into one body the sweep's 160 registers of
// operands and accumulators, the whitening's 80 and the pair stage's 140 ended in spills INSIDE the tile loop (6 500 cycles
// per tile instead of ~1 000).
// Their LDS arguments arrive as GENERIC pointers; used as such they become flat_* instructions, which count on vmcnt AND
// lgkmcnt and return in no fixed order relative to either: the compiler then waits with vmcnt(0) around them -- in
// uncertain_sweep that wait sat right behind the request for the tile two ahead, i.e. every third tile waited for a full
// memory round trip and the list's slow path for another (the stage: 38 800 cycles for 16 tiles).  Each function therefore
// casts them to LDS pointers first (lds_ptr).
template <class T>
using lds_t = __attribute__((address_space(3))) T;
template <class T>
__device__ __forceinline__ lds_t<T> *lds_ptr(T *p) {
  return (lds_t<T> *)p;
}
template <class T>
__device__ __forceinline__ const lds_t<T> *lds_ptr(const T *p) {
  return (const lds_t<T> *)p;
}

template <int KS>
__device__ __attribute__((noinline)) void uncertain_sweep(const void *refF_, int ntiles32_, const void *qF_, const float *thi_, long long set_,
                                                          long long ngroups_, long long nslots_, unsigned *plist_, unsigned *lcount_,
                                                          int wv_, int lane) {
  constexpr int QW = 4;
  lds_t<unsigned> *plist = lds_ptr(plist_);
  lds_t<unsigned> *lcount = lds_ptr(lcount_);
  const void *refF = uni(refF_);
  const int ntiles32 = uni(ntiles32_), wv = uni(wv_);
  const long long set = uni(set_), ngroups = uni(ngroups_), nslots = uni(nslots_);
  qF_ = uni(qF_);
  thi_ = uni(thi_);
  typedef __attribute__((address_space(1))) const half8 ghalf8;   // global, not generic: ordered with the tile loads on vmcnt
  typedef __attribute__((address_space(1))) const float gfloat;
  ghalf8 *qF = (ghalf8 *)qF_;
  gfloat *thig = (gfloat *)thi_;
  constexpr int kTileBytes = KS * 1024;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(refF), 0, ntiles32 * kTileBytes, 0x00020000);
  const int voff = lane * 16;
  const int rowbase = 4 * (lane >> 5);
  const long long g0 = set * QW;
  half8 bq[QW][KS];
  float thi[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const long long grp = (g0 + g < ngroups) ? g0 + g : ngroups - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) bq[g][s] = qF[((size_t)grp * KS + s) * 64 + lane];
    const long long qi = grp * 32 + (lane & 31);
    const bool have = g0 + g < ngroups && qi < nslots;
    thi[g] = have ? thig[qi] : -1.0f;
    if (!have) {
#pragma unroll
      for (int s = 0; s < KS; ++s) bq[g][s] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  const int tile0 = ntiles32 * wv / 8, tile1 = ntiles32 * (wv + 1) / 8;
  // the query operands are HERE before the first tile is asked for: left pending, their wait would sit at the head of the tile
  // loop -- as vmcnt(0), behind every request for the tile two ahead
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  float16v acc[QW];
  auto mm = [&](const half8(&A)[KS], int ga, int gb) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[ga] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[ga][s], s == 0 ? z : acc[ga], 0, 0, 0);
      acc[gb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[gb][s], s == 0 ? z : acc[gb], 0, 0, 0);
    }
  };
  auto list_group = [&](int g, int t) __attribute__((always_inline)) {
    const float16v &c = acc[g];
    const int tm = tree_min(c, kPosInf);
    const bool flagged = __int_as_float(tm) <= thi[g];
    if (__ballot(flagged) == 0ull) return;   // wave-uniform: nothing of this tile within reach of this group
    unsigned bits = 0u;
    if (flagged) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bits |= (c[r] <= thi[g]) ? (1u << r) : 0u;
    }
    if (bits != 0u) {
      unsigned at = __hip_atomic_fetch_add(lcount, (unsigned)__popc(bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned ql = (unsigned)(g * 32 + (lane & 31));
      while (bits != 0u) {
        const int r = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (at < kUncertainListCap) plist[at] = (ql << 24) | (unsigned)(t * 32 + rowbase + (r & 3) + 8 * (r >> 2));
        ++at;
      }
    }
  };
  auto tile = [&](const half8(&A)[KS], int t) __attribute__((always_inline)) {
    mm(A, 0, 1);
    mm(A, 2, 3);
#pragma unroll
    for (int g = 0; g < QW; ++g) list_group(g, t);
  };
  // tiles requested TWO ahead (three register sets): with one ahead every tile waited for its L2 round trip (2 500 cycles
  // per tile against ~1 000 of matrix instructions for the two waves of a SIMD); requests past the range re-read its last tile.
  // Still ~2 400 cycles per tile.  Tried: three and four ahead (four / five register sets): spills inside the loop at KS = 4,
  // 48 600 against 38 800 cycles for the stage; every workgroup starting at another tile of its range (so that the workgroups
  // of an XCD find each other's tiles in L2): no change.  The loop's own slow path (a group with a value in reach: 16 compares,
  // an LDS atomic, the list stores; one group in four here -- every proposal of the set has a band pair somewhere) costs about
  // as much as the matrix instructions
  half8 A0[KS], A1[KS], A2[KS];
  if (tile0 < tile1) {
    auto req = [&](half8(&A)[KS], int t) __attribute__((always_inline)) {
      load_tile<KS>(A, rsrc, voff, (t < tile1 ? t : tile1 - 1) * kTileBytes);
    };
    req(A0, tile0);
    __builtin_amdgcn_sched_barrier(0);   // in this order (see k_sweep_min)
    req(A1, tile0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    for (int t = tile0; t < tile1; t += 3) {
      req(A2, t + 2);
      tile(A0, t);
      if (t + 1 >= tile1) break;
      req(A0, t + 3);
      tile(A1, t + 1);
      if (t + 2 >= tile1) break;
      req(A1, t + 4);
      tile(A2, t + 2);
    }
  }
}

// one thread per listed pair: the live point's row (zero padded to dp, 16-byte aligned: dp is even) is requested as a whole --
// one round trip instead of one per unrolled batch of a 50-step chain -- then the reference's loop: sub, mul, add, each
// rounded, k ascending (this file is compiled with -ffp-contract=off)
template <int NCH>
__device__ __attribute__((noinline)) void uncertain_pairs(const unsigned *plist_, unsigned cnt_, const int *qid_, const double *tq_,
                                                          const double *refR_, int dp_, int d_, int n_, double r2, int *best_) {
  const lds_t<unsigned> *plist = lds_ptr(plist_);
  const lds_t<int> *qid = lds_ptr(qid_);
  const lds_t<double> *tq = lds_ptr(tq_);
  typedef __attribute__((address_space(1))) const double gdouble;
  gdouble *refR = (gdouble *)uni(refR_);
  int *best = uni(best_);
  const int dp = uni(dp_), d = uni(d_), n = uni(n_);
  const unsigned cnt = (unsigned)uni((int)cnt_);
  const int ds = uncertain_row_stride(d);
  for (unsigned e = threadIdx.x; e < cnt; e += 512) {
    const unsigned ent = plist[e];
    const int ql = (int)(ent >> 24), i = (int)(ent & 0xffffffu);
    const int q = qid[ql];
    if (q < 0 || i >= n) continue;
    // 16 NCH >= dp coordinates are requested whatever d is (past dp they belong to the next row: the array has a spare row
    // behind its last); the arithmetic stops at d
    typedef double double2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) const double2v gdouble2;
    gdouble2 *ar = (gdouble2 *)(refR + (size_t)i * dp);
    const lds_t<double> *br = tq + ql * ds;
    double2v row[8 * NCH];
#pragma unroll
    for (int k2 = 0; k2 < 8 * NCH; ++k2) row[k2] = ar[k2];
    double accd = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < 8 * NCH; ++k2) {   // terms past d add +0.0 (exact: the sum is non-negative); no branches
      const double d0 = row[k2].x - br[2 * k2];
      accd += 2 * k2 < d ? d0 * d0 : 0.0;
      const double d1 = row[k2].y - br[2 * k2 + 1];
      accd += 2 * k2 + 1 < d ? d1 * d1 : 0.0;
    }
    if (accd <= r2) atomicMin(&best[q], i);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16 queries of a set (rows 16 wv .. 16 wv + 15 of tq[.][stride]: the CENTRED proposals x - c) whitened in place:
// t_c = sum_k x_k T[k][c] with k ascending and one fused multiply-add per term -- the chain of k_prep / k_whiten_rows, the same
// operands, bit for bit (the padded terms k >= d are kept as there).  History: k_whiten_rows' own form (rows in registers, x_k
// through the DPP operand of v_fmac_f64, eight rows at a time) 30 000 cycles for the 16 rows; lane = output coordinate with
// T[k][.] one LDS read per k and x_k a broadcast LDS read per query: 20 800; this form: 11 700.
// On the FP64 matrix cores: v_mfma_f64_16x16x4_f64 accumulates k-ascending with one rounding per fused
// multiply-add (measured bit-identical to the scalar chain: scripts/probes/mfma64_probe.hip; k_prep3 whitens whole batches
// this way), so C[c][q] += sum over 4 k of T[k][c] x_q[k] IS four steps of the chain of query q, output c.  A = T^T fragment
// (lane l: T[4 ks + (l >> 4)][16 ct + (l & 15)], LDS), B = the centred rows (lane l: x_{q = l & 15}[4 ks + (l >> 4)], LDS), 13 x 4
// instructions and 65 LDS reads per lane for the wave's 16 queries where the vector form takes 800 fused multiply-adds and 850
// LDS reads.  Terms past d contribute 0 x T (kept, as in the chain); past dp both operands are zero.
typedef double double4m __attribute__((ext_vector_type(4)));

template <int NK>
__device__ __attribute__((noinline)) void whiten16_mfma(const double *tl_, int ldt8_, int d_, int dp_, double *tq_, int wv_, int lane) {
  const lds_t<double> *tl = lds_ptr(tl_);
  lds_t<double> *tq = lds_ptr(tq_);
  const int ldt8 = uni(ldt8_), d = uni(d_), dp = uni(dp_), wv = uni(wv_);
  constexpr int NC = (NK + 3) / 4;   // output tiles of 16 coordinates
  const int ds = uncertain_row_stride(d);
  const int q = lane & 15, kq = lane >> 4;
  const lds_t<double> *row = tq + (16 * wv + q) * ds;
  double bfr[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int k = 4 * ks + kq;
    bfr[ks] = k < d ? row[k] : 0.0;
  }
  double4m t[NC];
#pragma unroll
  for (int ct = 0; ct < NC; ++ct) t[ct] = (double4m){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int k = 4 * ks + kq;
    double afr[NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const int c = 16 * ct + q;
      afr[ct] = (k < dp && c < d) ? tl[k * ldt8 + c] : 0.0;
    }
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) t[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ct], bfr[ks], t[ct], 0, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();   // every lane has read the rows before they are overwritten
  lds_t<double> *out = tq + (16 * wv + q) * ds;
#pragma unroll
  for (int ct = 0; ct < NC; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * ct + kq + 4 * r;   // output row of the accumulator register (mlf_prep3.hip)
      if (c < d) out[c] = t[ct][r];
    }
}

// The uncertain queries: band pairs found, queries whitened, pairs decided -- in ONE launch, a workgroup of 8 waves per
// set of 128 uncertain queries (resident grid, sets taken in turn):
//   1. wave w sweeps its eighth of the live tiles against the set's four query groups; every value at or below T_hi goes
//      to the workgroup's pair list in LDS (slot from an LDS counter; values at or below T_lo too -- the earlier ranges
//      said there is none, and if this sweep should disagree the exact arithmetic decides),
//   2. the 128 queries are whitened in the reference's arithmetic -- k_whiten_rows' chain (delta_k = x_k - c_k, k-ascending
//      binary64 FMA per output coordinate; lane = output coordinate, x_k through the DPP operand of v_fmac_f64, T[k][.] one
//      coalesced load per k for 8 queries at a time) -- into LDS: every query of the set has a band pair, so all are needed,
//   3. one thread per listed pair: the reference's distance loop (sub, mul, add, each rounded, k ascending:
//      mlfriends.pyx:178-180) on the LDS copy, atomicMin on best[].
// Before: k_sweep_list (26-29 us) + k_recheck_whiten (41-64 us: one wave per list segment, a chain of ~15 dependent memory
// round trips whatever the number of pairs).  The ellipsoid band of k_prep4 rides in trailing workgroups.
template <int KS, int NCH>
__global__ __launch_bounds__(512, 1) void k_uncertain(UncertainArgs a) {
  constexpr int QW = 4;
  extern __shared__ __attribute__((aligned(16))) double lds_u[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int d = a.d, ds = uncertain_row_stride(d);
  double *tq = lds_u;                                                     // [128][ds] the queries' rows, then whitened in place
  double *tl = tq + 128 * ds;                                            // [dp][ldt8] the layer matrix
  unsigned *plist = reinterpret_cast<unsigned *>(tl + a.dp * a.ldt8);    // [kUncertainListCap] (query in set) << 24 | live index
  int *qid = reinterpret_cast<int *>(plist + kUncertainListCap);         // [128] original query of the slot, -1 = none
  unsigned *lcount = reinterpret_cast<unsigned *>(qid + 128);            // [1]
  // The ellipsoid band of the per-proposal stage rides in trailing workgroups: the uncertain sets occupy ~180 of the 256 CUs
  // (one workgroup per CU), the band's 512 waves (3-4 proposals each, ~30 us) run on the rest.  (In the middle of the first
  // k_sweep_min launch they cost that launch 10 us; in k_recheck_whiten they set the duration of the whole launch.)
  const unsigned nsweepblk = a.nsweepblk;
  if (blockIdx.x >= nsweepblk) {
    if (a.ell.count) ell_exact_wave(a.ell, lds_u + wv * 64, (blockIdx.x - nsweepblk) * 8 + wv, (gridDim.x - nsweepblk) * 8);
    return;
  }
  // Every sweeping workgroup writes its own statistics word, and workgroup 0's stamping thread clears the stamp words it will
  // write: no memset per batch (it was a 5 us fill kernel between the last sweep and this launch).
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a.seg_count[uncertain_stamp_base() + k] = 0u;
  }
  const long long nslots = (long long)*a.nslots_dev;
  if ((long long)blockIdx.x * 128 >= nslots) {   // no set for this workgroup
    if (threadIdx.x == 0) a.seg_count[blockIdx.x] = 0u;
    return;
  }
  // diagnostics: shader-clock stamps of workgroup 0 at the stage boundaries of its first set (mlf_region_debug_stats)
  unsigned nstamp = 0u;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && nstamp < 8u) a.seg_count[uncertain_stamp_base() + nstamp] = (unsigned)__builtin_readcyclecounter();
    ++nstamp;
  };
  stamp();
  {   // the layer matrix: at most 64 x 64 doubles, all requests first
    double v[8];
    const int nt = a.dp * a.ldt8;
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] = threadIdx.x + it * 512 < nt ? a.T8[threadIdx.x + it * 512] : 0.0;
#pragma unroll
    for (int it = 0; it < 8; ++it)
      if (threadIdx.x + it * 512 < nt) tl[threadIdx.x + it * 512] = v[it];
  }
  const long long ngroups = (nslots + 31) / 32;
  const long long nsets = (ngroups + QW - 1) / QW;
  unsigned listed_total = 0u;

  for (long long set = blockIdx.x; set < nsets; set += nsweepblk) {
    __syncthreads();   // the LDS of the set before is free
    if (threadIdx.x == 0) *lcount = 0u;
    if (threadIdx.x < 128) {
      const long long slot = set * 128 + threadIdx.x;
      qid[threadIdx.x] = slot < nslots ? a.qmap[slot] : -1;
    }
    __syncthreads();
    stamp();
    // ---- 1. this wave's tiles against the four groups (band pairs -> plist)
    uncertain_sweep<KS>(a.refF, a.ntiles32, a.qF, a.thi, set, ngroups, nslots, plist, lcount, wv, lane);
    stamp();
    // ---- 2. the set's queries: rows as handed over -> LDS (one round trip for all 128), then whitened in place (this
    // wave: queries 16 wv .. 16 wv + 15, eight at a time)
    {   // all requests first, then the stores (a plain loop waits for every element before it asks for the next: 13 HBM
        // round trips, 30 us).  Tried: the rows by LDS-DMA (global_load_lds) issued in front of the sweep -- the wave's first
        // in-order wait inside the sweep then includes them: same total.  Scattered rows of a 400 MB array: 16 us, mostly
        // address translation.
      double v[16];   // 128 x 64 / 512
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int e = threadIdx.x + it * 512;
        const int ql = e / d, k = e - ql * d;
        const int q = e < 128 * d ? qid[ql] : -1;
        v[it] = q >= 0 ? a.pts[(long long)q * d + k] - a.lay_ctr[k] : 0.0;   // delta_k = x_k - c_k: one rounding, as in k_prep
      }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int e = threadIdx.x + it * 512;
        const int ql = e / d, k = e - ql * d;
        if (e < 128 * d) tq[ql * ds + k] = v[it];
      }
    }
    __syncthreads();
    stamp();
    // d <= 64 here (launch_uncertain): 4 NK >= dp
    if (a.dp <= 16) whiten16_mfma<4>(tl, a.ldt8, d, a.dp, tq, wv, lane);
    else if (a.dp <= 32) whiten16_mfma<8>(tl, a.ldt8, d, a.dp, tq, wv, lane);
    else if (a.dp <= 52) whiten16_mfma<13>(tl, a.ldt8, d, a.dp, tq, wv, lane);
    else whiten16_mfma<16>(tl, a.ldt8, d, a.dp, tq, wv, lane);
    stamp();
    __syncthreads();
    stamp();
    // ---- 3. the listed pairs in the reference's arithmetic
    const unsigned cnt = *lcount;
    if (cnt > kUncertainListCap) {
      if (threadIdx.x == 0) a.counters[1] = 1u;   // overflow: the exact scan redoes the batch
    } else {
      uncertain_pairs<NCH>(plist, cnt, qid, tq, a.refR, a.dp, d, a.n, a.r2, a.best);
    }
    listed_total += cnt;
    stamp();
  }
  if (threadIdx.x == 0) a.seg_count[blockIdx.x] = listed_total;   // statistics (a workgroup without a set wrote 0 above)
}

// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int QW>
static hipError_t launch_sweep_min_t(const MinArgs &a, hipStream_t s) {
  const long long waves = (a.ngroups + QW - 1) / QW;
  long long wgs = (waves + 3) / 4;
  // a compacted set (its size is on the device) of a batch of up to 2 kSweepMinSetGrid x 16 groups (10^6 proposals: 1 953
  // workgroups): kSweepMinSetGrid workgroups, which walk the set in at most two passes (see the kernel).  Larger batches keep
  // one workgroup per 16 groups of the batch: there the empty workgroups are a small share of a long launch, and four passes
  // per workgroup measured 3-5 % slower than the hardware's own dispatch (4 * 10^6 proposals)
  if (a.nslots_dev && wgs > kSweepMinSetGrid && wgs <= 2 * kSweepMinSetGrid) wgs = kSweepMinSetGrid;
  const dim3 grid((unsigned)wgs);
  constexpr int PF = (KS * QW >= 12 && KS <= 4) ? 2 : 1;
  hipLaunchKernelGGL((k_sweep_min<KS, QW, PF>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_sweep_min(int ks, int qw, const MinArgs &a, hipStream_t s) {
  if (a.ngroups <= 0) return hipSuccess;
  switch (ks * 8 + qw) {
    case 1 * 8 + 4: return launch_sweep_min_t<1, 4>(a, s);
    case 2 * 8 + 4: return launch_sweep_min_t<2, 4>(a, s);
    case 3 * 8 + 4: return launch_sweep_min_t<3, 4>(a, s);
    case 4 * 8 + 4: return launch_sweep_min_t<4, 4>(a, s);
    case 1 * 8 + 2: return launch_sweep_min_t<1, 2>(a, s);
    case 2 * 8 + 2: return launch_sweep_min_t<2, 2>(a, s);
    case 3 * 8 + 2: return launch_sweep_min_t<3, 2>(a, s);
    case 4 * 8 + 2: return launch_sweep_min_t<4, 2>(a, s);
    default: return hipErrorInvalidValue;
  }
}

// workgroups (= statistics words) of the sweeping part of a k_uncertain launch: one per CU
long long uncertain_blocks() { return 256; }

hipError_t launch_uncertain(int ks, const UncertainArgs &a_in, hipStream_t s) {
  UncertainArgs a = a_in;
  a.nsweepblk = (unsigned)uncertain_blocks();
  const dim3 grid(a.nsweepblk + (a.ell.count ? kEllWaves / 8 : 0u));
  const int ds = a.d | 1;
  const size_t lds = (size_t)128 * ds * sizeof(double) + (size_t)a.dp * a.ldt8 * sizeof(double) +
                     kUncertainListCap * sizeof(unsigned) + 128 * sizeof(int) + 16;
  const int nch = (a.dp + 15) / 16;
  if (a.dp > 64 || nch < 1) return hipErrorInvalidValue;
#define X(KS, NCH)                                                                                                          \
  if (ks == KS && nch == NCH) {                                                                                             \
    static DeviceGrant grant;                                                                                               \
    if (hipError_t e = grant.ensure([] {                                                                                    \
          return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_uncertain<KS, NCH>),                                 \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                               \
        }))                                                                                                                 \
      return e;                                                                                                             \
    hipLaunchKernelGGL((k_uncertain<KS, NCH>), grid, dim3(512), lds, s, a);                                                 \
    return hipGetLastError();                                                                                               \
  }
  X(1, 1) X(2, 1) X(2, 2) X(3, 2) X(3, 3) X(4, 3) X(4, 4)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace mlf
