// mlf_filter_dev.hpp -- device helpers shared by the query quantisation kernels
// (k_quant_queries in mlf_filter.hip, the fused stages k_prep3 / k_prep4)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mlf {

typedef _Float16 half_t;

__device__ __forceinline__ size_t frag_index(int row, int k, int ks) {
  // element (row, k) of a [rows][16*ks] matrix in fragment-major order
  const int tile = row >> 5, rr = row & 31;
  const int kstep = k >> 4, kk = k & 15;
  const int lane = rr + ((kk >> 3) << 5);
  return ((((size_t)tile * ks + kstep) * 64 + lane) << 3) + (kk & 7);
}

// three binary16 pieces of a non-negative value < 60000 (covers ~33 bits)
__device__ __forceinline__ void split3(double v, half_t *p) {
  const half_t p1 = (half_t)(float)v;
  const double r1 = v - (double)(float)p1;
  const half_t p2 = (half_t)(float)r1;
  const double r2 = r1 - (double)(float)p2;
  p[0] = p1;
  p[1] = p2;
  p[2] = (half_t)(float)r2;
}

// Certain-hit / certain-miss thresholds of one query (derivation: header of mlf_filter.hip and
// DESIGN.md 4b).  nbn2 = |sigma (b - c)|^2 of the unrounded query.  Returns false if the query
// must take the exact scan instead (sentinel rows would not be certain misses any more).
__device__ __forceinline__ bool filter_thresholds(double sigma, double namax, double nbn2, double r2,
                                                  int K, float *lo_f, float *hi_f) {
  const double nbn = sqrt(nbn2);
  const double delta = 0x1p-11 * (1.0 + 0x1p-9) * (namax + nbn) + 2.0 * sqrt((double)K) * 0x1p-24 +
                       0x1p-40 * (namax + nbn);
  const double w = namax + nbn + 0x1p-8;
  const double eacc = 0x1p-15 * w * w + 0x1p-22;
  const double sr = sigma * sqrt(r2);
  const double lo = sr * (1.0 - 0x1p-30) - delta;
  const double hi = sr * (1.0 + 0x1p-30) + delta;
  // lo <= 0: no pair may count as a certain hit; -inf and not -1, because a computed Dt can be as low as -eacc
  // ... and never negative and finite: the kernels take minima on the bit patterns, which order negative values the wrong way
  // round; with T_lo >= 0 every negative Dt is a certain hit whichever of them the minimum keeps
  const double t_lo = (lo > 0.0 && lo * lo - eacc >= 0.0) ? lo * lo - eacc : -INFINITY;
  const double t_hi = hi * hi + eacc;
  float l = (float)t_lo;
  if ((double)l > t_lo) l = nextafterf(l, -INFINITY);
  float h = (float)t_hi;
  if ((double)h < t_hi) h = nextafterf(h, INFINITY);
  *lo_f = l;
  *hi_f = h;
  return t_hi < 30000.0;
}

// The same thresholds for k_prep4 (mlf_prep4.hip), where the query operand is the binary16 rounding of an APPROXIMATE
// whitened point bq (binary32 FMA chain) instead of the exact one: |bq - sigma b'| <= zeta.  Everything is evaluated in
// binary32; every intermediate that must be an upper (lower) bound is multiplied by up = 1 + 2^-18 (dn = 1 - 2^-18),
// which covers the < 16 roundings of 2^-24 each in front of it (all sums are sums of non-negative terms).
//   nb    computed |bh|^2 (relative error <= nu = 2^-18, added to Eacc)
//   zeta  >= |bq - sigma b'|
//   sr_lo <= sigma sqrt(r2) (1 - 2^-30),  sr_hi >= sigma sqrt(r2) (1 + 2^-30)
// |bh - sigma b'| <= |bh - bq| + zeta with |bh - bq| <= 2^-11 |bq| + sqrt(K) 2^-25 (binary16 rounding incl. subnormals) and
// |bq| <= |sigma b'| + zeta, so Delta = 2^-11 (1 + 2^-9)(namax + nbn) + 2 sqrt(K) 2^-24 + 2^-40 (namax + nbn) + (1 + 2^-10) zeta
// with nbn >= |sigma b'| obtained from |bh|: |sigma b'| <= |bq| + zeta, |bq| <= (|bh| + sqrt(K) 2^-25) / (1 - 2^-11).
__device__ __forceinline__ bool filter_thresholds4(float namax, float nb, float zeta, float sqrt_k, float sr_lo,
                                                   float sr_hi, float *lo_f, float *hi_f) {
  const float up = 1.0f + 0x1p-18f, dn = 1.0f - 0x1p-18f;
  const float nbh = __builtin_sqrtf(nb) * up;
  const float nbn = ((nbh + sqrt_k * 0x1p-25f) * (1.0f + 0x1p-10f) + zeta) * up;
  const float w0 = namax + nbn;
  const float delta = (0x1p-11f * (1.0f + 0x1p-9f) * w0 + 2.0f * sqrt_k * 0x1p-24f + 0x1p-40f * w0 +
                       (1.0f + 0x1p-10f) * zeta) * up;
  const float w = w0 + 0x1p-8f;
  const float eacc = (0x1p-15f * w * w + 0x1p-22f + 0x1p-18f * nb) * up;
  const float lo = (sr_lo * dn - delta) * dn;
  const float hi = (sr_hi * up + delta) * up;
  const float t_lo = (lo * lo) * dn - eacc * up;
  *lo_f = (lo > 0.0f && t_lo >= 0.0f) ? t_lo : -INFINITY;   // never negative and finite (see filter_thresholds)
  const float t_hi = ((hi * hi) * up + eacc) * up;
  *hi_f = t_hi;
  return t_hi < 30000.0f;   // false for NaN
}

}  // namespace mlf
