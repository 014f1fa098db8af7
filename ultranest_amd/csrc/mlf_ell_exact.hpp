// mlf_ell_exact.hpp -- device body of the exact ellipsoid test of the proposals k_prep4 could not decide; it runs as
// its own kernel (k_ell_exact, mlf_prep4.hip) and as the tail of the second-stage scan launch (mlf_scan.hip).
#pragma once
#include "mlf_prep4.hpp"

namespace mlf {

constexpr unsigned kEllBlocks = 128;
inline size_t ell_exact_lds(int d) { return ((size_t)d * (d | 1) + 4 * 64) * sizeof(double); }

// One wave per listed proposal, workgroups of 256 threads.  Tier 1: qt = |L^T delta|^2 in binary64 with the band
// eps = 2^-34 |A|_F |delta|^2 of k_prep3 (the factor and the wave's delta sit in LDS); tier 2 (inside that band,
// practically never): the reference's arithmetic -- one accumulator, j outer, (d_j * A_jk) * d_k, no FMA.  The last
// workgroup to finish resets the list counter for the next batch (every workgroup has read it by then).  `blk` / `nblk`:
// this workgroup's index within the workgroups that run this body.  ltl: (d (d|1) + 256) doubles of LDS.
__device__ __forceinline__ void ell_exact_body(const EllExactArgs &a, double *ltl, unsigned blk, unsigned nblk) {
  const unsigned count = *a.count < a.cap ? *a.count : a.cap;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned wave = blk * 4 + wv, nwaves = nblk * 4;
  const int d = a.d;
  const int ls = d | 1;
  double *dls = ltl + (size_t)d * ls + wv * 64;   // this wave's delta
  if (count != 0u) {   // uniform over the workgroups
    for (int e = threadIdx.x; e < d * d; e += 256) {
      const int k = e / d, j = e - k * d;
      ltl[k * ls + j] = a.ell_Lt[(size_t)k * a.dp + j];
    }
    __syncthreads();
  }
  for (unsigned e = wave; e < count; e += nwaves) {
    const long long p = a.list[e];
    const double *row = a.pts + p * (long long)d;
    const bool own = lane < d;
    const double dl = own ? row[lane] - a.ell_ctr[lane] : 0.0;
    __builtin_amdgcn_wave_barrier();
    dls[lane] = dl;
    __builtin_amdgcn_wave_barrier();
    const double *lrow = ltl + (own ? lane : 0) * ls;
    double y = 0.0;
#pragma unroll 8
    for (int j = 0; j < d; ++j) y = __builtin_fma((own && j >= lane) ? lrow[j] : 0.0, dls[j], y);
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
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(a.done, 1u);
    if (t == nblk - 1u) {
      if (a.last) *a.last = *a.count;   // kept for mlf_region_debug_stats
      *a.count = 0u;
      *a.done = 0u;
    }
  }
}

}  // namespace mlf
