// mlf_recheck_dev.hpp -- device bodies of the exact side work behind the bounded per-proposal stage (mlf_prep4.hip):
// the re-check of ONE list segment of uncertain pairs, with the binary64 whitening of the queries that appear in it, and
// the ellipsoid band.  They run as their own launch (k_recheck_whiten: behind the first-index pre-filter) and, since round 3,
// inside the mask-mode sweep itself (k_sweep: every wave re-checks the segment it has just written -- no launch boundary
// between sweep and re-check, and the band proposals ride in the first sweep launch).
#pragma once
#include "mlf_prep4.hpp"

namespace mlf {

constexpr unsigned kEllWaves = 512;

// the ellipsoid band, one wave per proposal, as light as the re-check waves it shares a launch with: row k of L^T
// is read through the vector L1 (20 KB, resident), delta sits in 64 doubles of LDS
__device__ __forceinline__ void ell_exact_wave(const EllExactArgs &a, double *dls, unsigned wave, unsigned nwaves) {
  const unsigned count = *a.count < a.cap ? *a.count : a.cap;
  const int lane = threadIdx.x & 63;
  const int d = a.d;
  const bool own = lane < d;
  const double *lcol = a.ell_L + (own ? lane : 0);   // column `lane` of the lower factor: L[j][lane], coalesced over the lanes
  const double myctr = own ? a.ell_ctr[lane] : 0.0;
  for (unsigned e = wave; e < count; e += nwaves) {
    const long long p = a.list[e];
    const double *row = a.pts + p * (long long)d;
    const double dl = own ? row[lane] - myctr : 0.0;
    __builtin_amdgcn_wave_barrier();
    dls[lane] = dl;
    __builtin_amdgcn_wave_barrier();
    double y = 0.0;
#pragma unroll 10
    for (int j = 0; j < d; ++j) y = __builtin_fma(lcol[(size_t)j * a.dp], dls[j], y);   // (L^T delta)_lane; L is stored with its zeros
    if (!own) y = 0.0;
    double qt = y * y, nrm2 = dl * dl;
    for (int o = 32; o > 0; o >>= 1) {
      qt += __shfl_xor(qt, o, 64);
      nrm2 += __shfl_xor(nrm2, o, 64);
    }
    const double eps = a.eps_scale * nrm2;
    bool inside;
    if (a.chol_ok && qt + eps < a.enlarge) {
      inside = true;
    } else if (a.chol_ok && qt - eps > a.enlarge) {
      inside = false;
    } else {
      double acc = 0.0;
      if (lane == 0) {
        for (int j = 0; j < d; ++j) {
          const double dj = row[j] - a.ell_ctr[j];
          const double *arow = a.ell_A + (size_t)j * a.dp;
          for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * (row[k] - a.ell_ctr[k]);
        }
      }
      acc = __shfl(acc, 0, 64);
      inside = acc <= a.enlarge;
    }
    if (!inside && lane == 0) {
      a.gate[p] = 0;
      if (a.route) a.route[p] = 0;
    }
  }
  if (lane == 0) {
    const unsigned t = atomicAdd(a.done, 1u);
    if (t == nwaves - 1u) {
      if (a.last) *a.last = *a.count;   // kept for mlf_region_debug_stats
      *a.count = 0u;
      *a.done = 0u;
    }
  }
}

constexpr int kTQ = 8;     // whitened queries held in LDS per round
constexpr int kChunk = 32;  // entries handled together: at most 32 distinct queries, a 64-slot table

__host__ __device__ inline size_t recheck_w_lds(int d) { return ((size_t)kTQ * ((d + 1) | 1) + kTQ * 64) * sizeof(double) + (size_t)3 * 64 * sizeof(int); }

// One wave: the `count` entries of the list segment `seg` (query << 32 | live index).  lds_r: recheck_w_lds(d) bytes of
// LDS private to the wave.  Reference arithmetic throughout: whitening = k-ascending binary64 FMA chain (as k_prep), the
// distance = sub, mul, add, each rounded, k ascending (mlfriends.pyx:178-180).
__device__ __forceinline__ void recheck_segment(const RecheckWArgs &a, const unsigned long long *seg, unsigned count,
                                                double *lds_r, int lane) {
  if (count == 0) return;
  const int d = a.d;
  const int ds = (d + 1) | 1;                                  // row stride of the whitened queries in LDS
  double *tq = lds_r;                                          // [kTQ][ds]
  double *dlw = tq + kTQ * ds;                                 // [kTQ][64] the centred proposals being whitened
  int *hkey = reinterpret_cast<int *>(dlw + kTQ * 64);         // [64] query or -1
  int *hid = hkey + 64;                                        // [64] number of the slot's query
  int *qlist = hid + 64;                                       // [64] number -> query
  // A segment rarely holds more than a few dozen pairs; longer ones are taken in chunks of kChunk entries (a query
  // that appears in two chunks is whitened twice: same result).
  for (unsigned e0 = 0; e0 < count; e0 += kChunk) {
    __builtin_amdgcn_wave_barrier();
    hkey[lane] = -1;
    __builtin_amdgcn_wave_barrier();
    // 1. distinct queries of the live entries of this chunk
    const unsigned e = e0 + (unsigned)lane;
    long long qi = -1;
    int i = 0;
    bool livee = false;
    unsigned h = 0;
    if (lane < kChunk && e < count) {
      const unsigned long long ent = seg[e];
      qi = (long long)(ent >> 32);
      i = (int)(ent & 0xffffffffu);
      livee = i < a.n && qi < a.nq && a.best[qi] > i;   // a certain hit at or below i settles the pair
    }
    if (livee) {
      h = ((unsigned)qi * 2654435761u) & 63u;
      while (true) {
        const int old = atomicCAS(&hkey[h], -1, (int)qi);
        if (old == -1 || old == (int)qi) break;
        h = (h + 1u) & 63u;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // 2. number the occupied slots
    const bool occ = hkey[lane] != -1;
    const unsigned long long bm = __ballot(occ);
    const unsigned nqb = (unsigned)__popcll(bm);
    if (occ) {
      const unsigned id = (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
      hid[lane] = (int)id;
      qlist[id] = hkey[lane];
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned myid = livee ? (unsigned)hid[h] : 0xffffffffu;
    // 3. rounds of kTQ queries.  The rows of a round's queries are requested together and whitened together: one
    // coalesced load of the matrix row T[k][.] (64 x 64 doubles, zero padded: a.T64; resident in the vector L1) feeds
    // the chains of all of them.
    const bool inrow = lane < d;
    const double myctr = inrow ? a.lay_ctr[lane] : 0.0;
    const double *tp = a.T64 + lane;
    for (unsigned r0 = 0; r0 < nqb; r0 += kTQ) {
      const unsigned nr = nqb - r0 < (unsigned)kTQ ? nqb - r0 : (unsigned)kTQ;   // wave-uniform
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < kTQ; ++t)
        if ((unsigned)t < nr) dlw[t * 64 + lane] = inrow ? a.pts[(long long)qlist[r0 + t] * d + lane] - myctr : 0.0;
      __builtin_amdgcn_wave_barrier();
      double acc[kTQ];
#pragma unroll
      for (int t = 0; t < kTQ; ++t) acc[t] = 0.0;
      if (nr <= 2u) {   // the common case: keep the chains short; 25 rows of the matrix requested at a time (ten: five L2 round
                        // trips of ~0.6 us each in a chain that is all latency)
#pragma unroll 25
        for (int k = 0; k < d; ++k) {
          const double tk = tp[k * 64];
          acc[0] = __builtin_fma(dlw[k], tk, acc[0]);
          acc[1] = __builtin_fma(dlw[64 + k], tk, acc[1]);
        }
      } else {
#pragma unroll 5
        for (int k = 0; k < d; ++k) {
          const double tk = tp[k * 64];
#pragma unroll
          for (int t = 0; t < kTQ; ++t) acc[t] = __builtin_fma(dlw[t * 64 + k], tk, acc[t]);   // rows past nr: zeros or stale, unused
        }
      }
#pragma unroll
      for (int t = 0; t < kTQ; ++t)
        if ((unsigned)t < nr && inrow) tq[t * ds + lane] = acc[t];
      __builtin_amdgcn_wave_barrier();
      if (livee && myid >= r0 && myid < r0 + kTQ && a.best[qi] > i) {
        const double2 *ar = reinterpret_cast<const double2 *>(a.refR + (size_t)i * a.dp);   // rows are 16-byte aligned (dp even)
        const double *br = tq + (myid - r0) * ds;
        double accd = 0.0;
        const int d2 = d >> 1;
#pragma unroll 13
        for (int k2 = 0; k2 < d2; ++k2) {   // the reference's loop: sub, mul, add, each rounded, k ascending (26 coordinates requested at a time)
          const double2 av = ar[k2];
          const double d0 = av.x - br[2 * k2];
          accd += d0 * d0;
          const double d1 = av.y - br[2 * k2 + 1];
          accd += d1 * d1;
        }
        if (d & 1) {
          const double d0 = a.refR[(size_t)i * a.dp + d - 1] - br[d - 1];
          accd += d0 * d0;
        }
        if (accd <= a.r2) atomicMin(&a.best[qi], i);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace mlf
