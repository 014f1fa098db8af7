// mlf_sample.hip -- device-side proposal generation and stream compaction for region.sample()
// (reference MLFriends.sample_from_boundingbox mlfriends.pyx:1096-1112 and
// sample_from_wrapping_ellipsoid :1135-1160).  Opt-in: the stream is Philox-4x32-10 (counter based,
// reproducible for a given (seed, offset) on any grid), not numpy's MT19937, so runs agree with
// the reference statistically, not draw by draw.
//
// With the host path a 10^6 x 50 batch costs 0.3-0.5 s of np.random.uniform plus a 400 MB upload
// in front of a ~1 ms membership kernel; here the proposals never leave HBM and only the accepted
// rows travel back.
#include "mlf_sample.hpp"
#include "mlf_philox_dev.hpp"

#include <math.h>

namespace mlf {

// raw Philox words for testing: out[4*i .. 4*i+3] = philox(counter = (i, 0, stream, 0), key = seed)
__global__ void k_philox_words(unsigned long long seed, unsigned stream, long long n, unsigned *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned w[4];
  philox4x32_10((unsigned)i, (unsigned)((unsigned long long)i >> 32), stream, 0u, (unsigned)seed,
                (unsigned)(seed >> 32), w);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[4 * i + j] = w[j];
}

// element e of the batch = uniform(0,1); counter = offset + e/2 (two doubles per Philox block)
__global__ void k_generate_cube(double *pts, long long nelem, unsigned long long seed,
                                unsigned long long offset) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // Philox block index
  if (2 * b >= nelem) return;
  const unsigned long long ctr = offset + (unsigned long long)b;
  unsigned w[4];
  philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
  pts[2 * b] = u01(w[0], w[1]);
  if (2 * b + 1 < nelem) pts[2 * b + 1] = u01(w[2], w[3]);
}

// one thread per proposal: z ~ N(0, I_d), scaled to a uniform draw in the ball of radius
// sqrt(enlarge):  z / |z| * sqrt(enlarge) * U^(1/d)   (reference :1145-1149)
__global__ void k_generate_ball(double *z, long long n, int d, double enlarge, unsigned long long seed,
                                unsigned long long offset) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double *row = z + p * d;
  double norm2 = 0.0;
  const int npairs = (d + 1) / 2;
  for (int j = 0; j < npairs; ++j) {
    const unsigned long long ctr = offset + (unsigned long long)p * (unsigned long long)(npairs + 1) + j;
    unsigned w[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
    const double r = sqrt(-2.0 * log(u01(w[0], w[1])));
    const double ang = 2.0 * M_PI * u01(w[2], w[3]);
    const double g0 = r * cos(ang), g1 = r * sin(ang);
    row[2 * j] = g0;
    norm2 += g0 * g0;
    if (2 * j + 1 < d) {
      row[2 * j + 1] = g1;
      norm2 += g1 * g1;
    }
  }
  const unsigned long long ctr = offset + (unsigned long long)p * (unsigned long long)(npairs + 1) + npairs;
  unsigned w[4];
  philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
  const double scale = sqrt(enlarge) * pow(u01(w[0], w[1]), 1.0 / (double)d) / sqrt(norm2);
  for (int k = 0; k < d; ++k) row[k] *= scale;
}

// Shared last stages of k_generate_ellipsoid / k_rows_affine: the workgroup's 64 rows lie in LDS (zs[r * zs_stride + j]);
// lane = row, WAVE = output chunk of CH columns: out_k = sum_j (z_j scale_r) A[j][k] + shift_k, FMA chain in ascending j.  The
// matrix elements are the same for all lanes of a wave and come through the SCALAR cache (wave number by readfirstlane), one SGPR
// pair per FMA -- read per lane (64 lanes x 8 bytes per FMA through the vector memory pipe) they were 0.6 of 0.88 ms.  A = the
// padded copy [d][4 CH].  Circular axes rotated back (AffineLayer.unwrap, mlfriends.pyx:538-545) where wrap_shift is given;
// unit-cube test; the finished rows (staged over the inputs, which nobody reads any more) leave as one contiguous piece.
template <int CH>
__device__ __forceinline__ void rows_times_matrix(double *zs, int zs_stride, const double *scale, int nrows, int d, const double *A,
                                                  const double *shift, const double *wrap_shift, unsigned *bad, double *dst,
                                                  uint8_t *in_cube, int tid) {
  const int r = tid & 63;
  const int c = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave's output chunk: a scalar, and with it every matrix address
  double acc[CH];
#pragma unroll
  for (int kk = 0; kk < CH; ++kk) acc[kk] = 0.0;
  if (c * CH < d) {   // wave-uniform
    const double sc = scale ? (r < nrows ? scale[r] : 0.0) : 1.0;
    const double *Ac = A + c * CH;
    const double *zr = zs + (r < nrows ? r : 0) * zs_stride;
    for (int j = 0; j < d; ++j) {
      const double zj = scale ? zr[j] * sc : zr[j];
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) acc[kk] = __builtin_fma(zj, Ac[(size_t)j * (4 * CH) + kk], acc[kk]);
    }
  }
  __syncthreads();   // every wave has read its inputs: the buffer takes the outputs
  if (c * CH < d && r < nrows) {
    bool out_of_cube = false;
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      const int k = c * CH + kk;
      if (k < d) {
        double v = acc[kk] + shift[k];
        if (wrap_shift) {
          const double sh = wrap_shift[k];
          if (sh == sh) v = fmod(v + (1.0 - sh), 1.0);   // cut = 1 - shift
        }
        zs[r * d + k] = v;
        out_of_cube = out_of_cube || !((v > 0.0) && (v < 1.0));
      }
    }
    if (out_of_cube) bad[r] = 1u;   // benign race: every writer stores 1
  }
  __syncthreads();
  for (int e = tid; e < nrows * d; e += 256) dst[e] = zs[e];
  if (tid < nrows) in_cube[tid] = bad[tid] ? 0 : 1;
}

// AffineLayer.untransform for whole batches (reference mlfriends.pyx:745-752): w = t . invT + ctr, circular axes rotated back,
// unit-cube test.  64 rows per workgroup arrive in LDS as one contiguous piece.  (k_untransform_rows, one thread per row and
// matrix elements per lane: 17.4 ms per 10^6 x 50 rows, profiles/r06_refill_tspace_before.csv.)
template <int CH>
__global__ __launch_bounds__(256) void k_rows_affine(const double *t, long long n, int d, const double *A, const double *ctr,
                                                     const double *wrap_shift, double *w, uint8_t *in_cube, unsigned dmagic) {
  extern __shared__ __attribute__((aligned(16))) double lds_ra[];
  const int zs_stride = d + 1;
  double *zs = lds_ra;
  unsigned *bad = reinterpret_cast<unsigned *>(zs + 64 * zs_stride);
  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * 64;
  const int nrows = n - row0 >= 64 ? 64 : (int)(n - row0);
  if (tid < 64) bad[tid] = 0u;
  const double *src = t + row0 * d;
  for (unsigned e = tid; e < (unsigned)(nrows * d); e += 256) {
    const unsigned r = d == 1 ? e : __umulhi(e, dmagic);
    zs[r * zs_stride + (e - r * (unsigned)d)] = src[e];
  }
  __syncthreads();
  rows_times_matrix<CH>(zs, zs_stride, nullptr, nrows, d, A, ctr, wrap_shift, bad, w + row0 * d, in_cube + row0, tid);
}

// sample_from_wrapping_ellipsoid in ONE launch (reference :1135-1160): the draws of k_generate_ball (same counters, same
// Box-Muller, same scaling), the product with the ellipsoid's axes, the centre and the unit-cube test -- the batch is written
// once, in whole contiguous pieces.  Before: k_generate_ball (one thread per row: stores with a stride of 8 d bytes between
// lanes, every value written twice, 1.17 ms per 2^20 x 50), k_prep's matrix product (0.67 ms) and k_center_and_cube (0.15 ms),
// 1.26 GB of traffic for a 0.42 GB batch.
//   workgroup = 256 threads = 64 rows.  1: thread per Box-Muller pair (flat over the rows' pairs), values to LDS.  2: 4 threads per
//   row add up |z|^2 (k = part, part + 4, ...; combined by xor-shuffles: a fixed order, not the sequential one of k_generate_ball --
//   the draws agree to the last bits only, 1e-16 relative) and one of them draws the radius.  3: lane = row, WAVE = output chunk
//   of CH columns: w_k = sum_j (z_j scale) A[j][k], FMA chain in ascending j as k_prep does -- the matrix elements are the same
//   for all lanes of a wave and come through the SCALAR cache (wave number by readfirstlane), one SGPR pair per FMA; the first
//   version read them per lane (64 lanes x 8 bytes per FMA through the vector memory pipe: 0.6 of its 0.88 ms).  A = the padded
//   copy [d][4 CH].  + centre; cube test.  4: the finished rows (staged over the z values, which nobody reads any more) leave as
//   one contiguous piece.
template <int CH>
__global__ __launch_bounds__(256) void k_generate_ellipsoid(double *w, long long n, int d, double enlarge, const double *A,
                                                            const double *center, uint8_t *in_cube, unsigned long long seed,
                                                            unsigned long long offset, unsigned pmagic) {
  extern __shared__ __attribute__((aligned(16))) double lds_ge[];
  constexpr int RB = 64;
  const int zs_stride = d + 1;
  double *zs = lds_ge;                       // [64][d + 1]; from stage 3's end on: the outputs, [64 d]
  double *scale = zs + RB * zs_stride;       // [64]
  unsigned *bad = reinterpret_cast<unsigned *>(scale + RB);   // [64]
  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * RB;
  const int nrows = n - row0 >= RB ? RB : (int)(n - row0);
  const int npairs = (d + 1) / 2;
  if (tid < RB) bad[tid] = 0u;
  for (unsigned e = tid; e < (unsigned)(nrows * npairs); e += 256) {
    const unsigned r = npairs == 1 ? e : __umulhi(e, pmagic), j = e - r * (unsigned)npairs;   // (2^32 / 1 does not fit the magic word)
    const unsigned long long ctr = offset + (unsigned long long)(row0 + r) * (unsigned long long)(npairs + 1) + j;
    unsigned wd[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32), wd);
    const double rad = sqrt(-2.0 * log(u01(wd[0], wd[1])));
    // cos / sin of 2 pi u as cospi / sinpi of 2 u: one exact argument reduction for both instead of two reductions by a
    // rounded 2 pi u (k_generate_ball's form; the values differ by the rounding of that product, ~1e-16)
    double sn, cs;
    sincospi(2.0 * u01(wd[2], wd[3]), &sn, &cs);
    zs[r * zs_stride + 2 * j] = rad * cs;
    if ((int)(2 * j + 1) < d) zs[r * zs_stride + 2 * j + 1] = rad * sn;
  }
  __syncthreads();
  {
    const int r = tid >> 2, part = tid & 3;
    double acc = 0.0;
    if (r < nrows)
      for (int k = part; k < d; k += 4) {
        const double v = zs[r * zs_stride + k];
        acc += v * v;
      }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && r < nrows) {
      const unsigned long long ctr = offset + (unsigned long long)(row0 + r) * (unsigned long long)(npairs + 1) + npairs;
      unsigned wd[4];
      philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32), wd);
      scale[r] = sqrt(enlarge) * pow(u01(wd[0], wd[1]), 1.0 / (double)d) / sqrt(acc);
    }
  }
  __syncthreads();
  rows_times_matrix<CH>(zs, zs_stride, scale, nrows, d, A, center, nullptr, bad, w + row0 * d, in_cube + row0, tid);
}

// w = center + (z @ axes_T) was produced without the centre by the whitening kernel; add it and
// record whether the point lies strictly inside the unit cube (reference :1154)
// One WAVE per 64 rows, lanes striding the rows' elements (coalesced 512-byte accesses); a row's verdict is the AND over
// its d elements, kept per lane for the rows a lane touches and combined through LDS.  (One thread per row read with a
// stride of 8 d bytes between lanes: 1.46 ms per 2^20 x 50, ten times the streaming time -- profiles/r06_refill_kernel_stats.csv.)
__global__ __launch_bounds__(256) void k_center_and_cube(double *w, long long n, int d, const double *center, uint8_t *in_cube) {
  __shared__ unsigned bad[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long row0 = ((long long)blockIdx.x * 4 + wv) * 64;
  if (row0 >= n) return;
  const int nrows = n - row0 >= 64 ? 64 : (int)(n - row0);
  bad[wv][lane] = 0u;
  __builtin_amdgcn_wave_barrier();
  double *base = w + row0 * d;
  const int total = nrows * d;
  for (int e = lane; e < total; e += 64) {
    const int r = e / d, k = e - r * d;
    const double v = base[e] + center[k];
    base[e] = v;
    if (!((v > 0.0) && (v < 1.0))) bad[wv][r] = 1u;   // benign race: every writer stores 1
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < nrows) in_cube[row0 + lane] = bad[wv][lane] ? 0 : 1;
}

// method 2 (reference mlfriends.pyx:1114-1133): uniform in the padded t-space bounding box,
// low + (high - low) * U per coordinate with low = bbox_lo - pad, high = bbox_hi + pad
__global__ void k_generate_tbox(double *t, long long nelem, int d, const double *lo, const double *hi, double pad,
                                unsigned long long seed, unsigned long long offset) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // Philox block index
  if (2 * b >= nelem) return;
  unsigned w[4];
  philox_block(seed, 5u, offset + (unsigned long long)b, w);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long e = 2 * b + j;
    if (e < nelem) {
      const int k = (int)(e % d);
      const double low = lo[k] - pad, high = hi[k] + pad;
      const double range = high - low;
      const double scaled = range * u01(w[2 * j], w[2 * j + 1]);
      t[e] = low + scaled;
    }
  }
}

// method 3 (reference :1072-1094): a random live point plus a uniform draw in its ball of radius sqrt(r2).
// Philox stream 6, (npairs + 2) blocks per proposal: block 0 = live index + radial uniform + thinning uniform
// (kept in thin_u), blocks 1.. = Box-Muller pairs.
// 64 rows per workgroup, one thread per Box-Muller pair (flat over the rows' pairs) into LDS, 4 threads per row for |z|^2, the
// finished rows written as one contiguous piece.  (One thread per row with stores 8 d bytes apart between lanes: 1.53 ms per
// 2^20 x 50.)  Same counters as that version; |z|^2 added up in a fixed order of its own (draws agree to ~1e-16).
__global__ __launch_bounds__(256) void k_generate_around_points(double *t, double *thin_u, long long n, int d, const double *refR,
                                                                int nlive, int dp, double r2, unsigned long long seed,
                                                                unsigned long long offset, unsigned pmagic, unsigned dmagic) {
  extern __shared__ __attribute__((aligned(16))) double lds_ap[];
  const int zs_stride = d + 1;
  double *zs = lds_ap;                     // [64][d + 1]
  double *fac = zs + 64 * zs_stride;       // [64]
  unsigned *which = reinterpret_cast<unsigned *>(fac + 64);   // [64]
  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * 64;
  const int nrows = n - row0 >= 64 ? 64 : (int)(n - row0);
  const int npairs = (d + 1) / 2;
  for (unsigned e = tid; e < (unsigned)(nrows * npairs); e += 256) {
    const unsigned r = npairs == 1 ? e : __umulhi(e, pmagic), j = e - r * (unsigned)npairs;
    const unsigned long long base = offset + (unsigned long long)(row0 + r) * (unsigned long long)(npairs + 2);
    unsigned r4[4];
    philox_block(seed, 6u, base + 2 + j, r4);
    const double rad = sqrt(-2.0 * log(u01(r4[0], r4[1])));
    double sn, cs;
    sincospi(2.0 * u01(r4[2], r4[3]), &sn, &cs);
    zs[r * zs_stride + 2 * j] = rad * cs;
    if ((int)(2 * j + 1) < d) zs[r * zs_stride + 2 * j + 1] = rad * sn;
  }
  __syncthreads();
  {
    const int r = tid >> 2, part = tid & 3;
    double acc = 0.0;
    if (r < nrows)
      for (int k = part; k < d; k += 4) {
        const double v = zs[r * zs_stride + k];
        acc += v * v;
      }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && r < nrows) {
      const unsigned long long base = offset + (unsigned long long)(row0 + r) * (unsigned long long)(npairs + 2);
      unsigned w0[4], w1[4];
      philox_block(seed, 6u, base, w0);
      philox_block(seed, 6u, base + 1, w1);
      which[r] = below(w0[0], (unsigned)nlive);
      const double radial = u01(w0[2], w0[3]);
      thin_u[row0 + r] = u01(w1[0], w1[1]);
      fac[r] = pow(radial, 1.0 / (double)d) / sqrt(acc) * sqrt(r2);
    }
  }
  __syncthreads();
  double *dst = t + row0 * d;
  for (unsigned e = tid; e < (unsigned)(nrows * d); e += 256) {
    const unsigned r = d == 1 ? e : __umulhi(e, dmagic), k = e - r * (unsigned)d;
    dst[e] = refR[(size_t)which[r] * dp + k] + zs[r * zs_stride + k] * fac[r];
  }
}

// keep a proposal with probability 1 / multiplicity (reference :1089: uniform(high=multiplicity) < 1)
__global__ void k_thin_by_multiplicity(const long long *count, const double *thin_u, long long n, uint8_t *mask) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double m = (double)count[p];
  mask[p] = (count[p] > 0 && thin_u[p] * m < 1.0) ? 1 : 0;
}

// AffineLayer.untransform (reference mlfriends.pyx:745-752, unwrap :538-545): w = t . invT + ctr with a
// k-ascending FMA chain, circular axes rotated back; also the unit-cube test.  One thread per row
// and coordinate block would be faster; the rows here are the (few) survivors of the scan.
__global__ void k_untransform_rows(const double *t, long long n, int d, const double *invT, const double *ctr,
                                   const double *wrap_shift, double *w, uint8_t *in_cube) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double *row = t + p * d;
  bool ok = true;
  for (int c = 0; c < d; ++c) {
    double acc = 0.0;
    for (int k = 0; k < d; ++k) acc = __builtin_fma(row[k], invT[(size_t)k * d + c], acc);
    double v = acc + ctr[c];
    if (wrap_shift) {
      const double sh = wrap_shift[c];
      if (sh == sh) v = fmod(v + (1.0 - sh), 1.0);   // cut = 1 - shift
    }
    w[p * d + c] = v;
    ok = ok && (v > 0.0) && (v < 1.0);
  }
  in_cube[p] = ok ? 1 : 0;
}

// prior transforms of the benchmark problems, elementwise: 0 identity, 1 x*a + b, 2 (x*a)*b
__global__ void k_elementwise_affine(const double *x, long long n, int tkind, double a, double b, double *out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const double v = x[e];
  double p = v;
  if (tkind == 1) {
    const double m = v * a;
    p = m + b;
  } else if (tkind == 2) {
    const double m = v * a;
    p = m * b;
  }
  out[e] = p;
}

// mask[e] = v[e] > threshold (and also[e], where a mask of the rows that count is given: the others hold no likelihood)
__global__ void k_mask_greater(const double *v, long long n, double threshold, const uint8_t *also, uint8_t *mask) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) mask[e] = (v[e] > threshold && (!also || also[e] != 0)) ? 1 : 0;
}

__global__ void k_mask_and(uint8_t *mask, const uint8_t *other, long long n) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) mask[p] = (mask[p] && other[p]) ? 1 : 0;
}

// proposals with pregate == 0 (outside the cube) leave the membership pipeline after the
// per-proposal stage: gate 0 (exact scan skips them), route 0 ("not scanned" for the MFMA
// pre-filter's finalise step) and thresholds -1 (no candidate pairs, no re-check entries)
__global__ void k_apply_pregate(const uint8_t *pregate, long long n, uint8_t *gate, uint8_t *route, float *tlo,
                                float *thi) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || pregate[p] != 0) return;
  gate[p] = 0;
  if (route) {
    route[p] = 0;
    tlo[p] = -1.0f;
    thi[p] = -1.0f;
  }
}

// ---- deterministic stream compaction of the accepted rows (256 proposals per workgroup) -------
__global__ __launch_bounds__(256) void k_count_accepted(const uint8_t *mask, long long n, unsigned *blk) {
  __shared__ unsigned wsum[4];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool acc = p < n && mask[p] != 0;
  const unsigned long long b = __ballot(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the per-workgroup counts (single workgroup); total -> blk[nblk]
__global__ __launch_bounds__(1024) void k_scan_counts(unsigned *blk, int nblk) {
  __shared__ unsigned part[1024];
  const int tid = threadIdx.x;
  const int per = (nblk + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < nblk ? lo + per : nblk;
  unsigned s = 0;
  for (int i = lo; i < hi; ++i) s += blk[i];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    const unsigned v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  unsigned run = part[tid] - s;
  for (int i = lo; i < hi; ++i) {
    const unsigned c = blk[i];
    blk[i] = run;
    run += c;
  }
  if (tid == 1023) blk[nblk] = part[1023];
}

// The 256 rows of a workgroup are one contiguous piece of the batch: its threads walk the piece element by element (coalesced
// reads; a row's d elements land next to each other at its rank: coalesced writes) and skip the elements of rejected rows.
// (One thread per row copied with a stride of 8 d bytes between lanes: 1.19 ms per 2^20 x 50 batch, 0.66 TB/s --
// profiles/r06_refill_kernel_stats_after.csv.)  dmagic = ceil(2^32 / d): e / d = umulhi(e, dmagic) for e < 2^32 / d.
__global__ __launch_bounds__(256) void k_scatter_accepted(const double *pts, const uint8_t *mask,
                                                          long long n, int d, unsigned dmagic, const unsigned *blk,
                                                          double *out, unsigned capacity) {
  __shared__ unsigned wsum[4];
  __shared__ unsigned rk[256];
  const long long row0 = (long long)blockIdx.x * 256;
  const long long p = row0 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool acc = p < n && mask[p] != 0;
  const unsigned long long b = __ballot(acc);
  if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned base = blk[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const unsigned rank = base + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
  rk[threadIdx.x] = (acc && rank < capacity) ? rank : 0xffffffffu;
  __syncthreads();
  if (wsum[0] + wsum[1] + wsum[2] + wsum[3] == 0u) return;
  const long long left = n - row0;
  const unsigned total = (unsigned)(left < 256 ? left : 256) * (unsigned)d;
  const double *src = pts + row0 * d;
  for (unsigned e = threadIdx.x; e < total; e += 256) {
    const unsigned r = __umulhi(e, dmagic);
    const unsigned rr = rk[r];
    if (rr != 0xffffffffu) out[(size_t)rr * d + (e - r * (unsigned)d)] = src[e];
  }
}

// d = 1 (a vector of likelihood values): one element per thread
__global__ __launch_bounds__(256) void k_scatter_scalars(const double *v, const uint8_t *mask, long long n, const unsigned *blk,
                                                         double *out, unsigned capacity) {
  __shared__ unsigned wsum[4];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool acc = p < n && mask[p] != 0;
  const unsigned long long b = __ballot(acc);
  if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned base = blk[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const unsigned rank = base + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
  if (acc && rank < capacity) out[rank] = v[p];
}

// ---------------------------------------------------------------- launchers -------------------
void launch_philox_words(unsigned long long seed, unsigned stream, long long n, unsigned *out, hipStream_t s) {
  hipLaunchKernelGGL(k_philox_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, stream, n, out);
}

void launch_generate_cube(double *pts, long long nelem, unsigned long long seed, unsigned long long offset,
                          hipStream_t s) {
  const long long nb = (nelem + 1) / 2;
  hipLaunchKernelGGL(k_generate_cube, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, pts, nelem, seed, offset);
}

void launch_generate_ball(double *z, long long n, int d, double enlarge, unsigned long long seed,
                          unsigned long long offset, hipStream_t s) {
  hipLaunchKernelGGL(k_generate_ball, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, n, d, enlarge, seed,
                     offset);
}

int generate_ellipsoid_chunk(int d) {   // outputs per wave: the instantiated size that covers ceil(d / 4)
  const int need = (d + 3) / 4;
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : need <= 13 ? 13 : need <= 16 ? 16 : 32;
}

hipError_t launch_generate_ellipsoid(double *w, long long n, int d, double enlarge, const double *A_padded, const double *center,
                                     uint8_t *in_cube, unsigned long long seed, unsigned long long offset, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (d < 1 || d > 128) return hipErrorInvalidValue;
  const int npairs = (d + 1) / 2;
  const unsigned pmagic = npairs > 1 ? (unsigned)((0x100000000ull + (unsigned)npairs - 1) / (unsigned)npairs) : 0u;
  const size_t lds = ((size_t)64 * (d + 1) + 64) * sizeof(double) + 64 * sizeof(unsigned);
  const dim3 grid((unsigned)((n + 63) / 64));
  static DeviceGrant grant;   // d = 128: 66 KiB
  if (hipError_t e = grant.ensure([] {
        hipError_t rc = hipSuccess;
#define MLF_GE(C)                                                                                                              \
  if (rc == hipSuccess)                                                                                                        \
    rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_generate_ellipsoid<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        MLF_GE(1) MLF_GE(2) MLF_GE(4) MLF_GE(8) MLF_GE(13) MLF_GE(16) MLF_GE(32)
#undef MLF_GE
        return rc;
      }))
    return e;
  switch (generate_ellipsoid_chunk(d)) {
#define MLF_GE(C)                                                                                                              \
  case C:                                                                                                                      \
    hipLaunchKernelGGL((k_generate_ellipsoid<C>), grid, dim3(256), lds, s, w, n, d, enlarge, A_padded, center, in_cube, seed,  \
                       offset, pmagic);                                                                                        \
    break;
    MLF_GE(1) MLF_GE(2) MLF_GE(4) MLF_GE(8) MLF_GE(13) MLF_GE(16) MLF_GE(32)
#undef MLF_GE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_rows_affine(const double *t, long long n, int d, const double *A_padded, const double *ctr, const double *wrap_shift,
                              double *w, uint8_t *in_cube, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (d < 1 || d > 128) return hipErrorInvalidValue;
  const unsigned dmagic = d > 1 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u;
  const size_t lds = (size_t)64 * (d + 1) * sizeof(double) + 64 * sizeof(unsigned);
  const dim3 grid((unsigned)((n + 63) / 64));
  static DeviceGrant grant;
  if (hipError_t e = grant.ensure([] {
        hipError_t rc = hipSuccess;
#define MLF_RA(C)                                                                                                              \
  if (rc == hipSuccess)                                                                                                        \
    rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows_affine<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        MLF_RA(1) MLF_RA(2) MLF_RA(4) MLF_RA(8) MLF_RA(13) MLF_RA(16) MLF_RA(32)
#undef MLF_RA
        return rc;
      }))
    return e;
  switch (generate_ellipsoid_chunk(d)) {
#define MLF_RA(C)                                                                                                              \
  case C:                                                                                                                      \
    hipLaunchKernelGGL((k_rows_affine<C>), grid, dim3(256), lds, s, t, n, d, A_padded, ctr, wrap_shift, w, in_cube, dmagic);   \
    break;
    MLF_RA(1) MLF_RA(2) MLF_RA(4) MLF_RA(8) MLF_RA(13) MLF_RA(16) MLF_RA(32)
#undef MLF_RA
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

void launch_center_and_cube(double *w, long long n, int d, const double *center, uint8_t *in_cube, hipStream_t s) {
  hipLaunchKernelGGL(k_center_and_cube, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n, d, center, in_cube);   // 4 waves x 64 rows
}

void launch_generate_tbox(double *t, long long n, int d, const double *lo, const double *hi, double pad,
                          unsigned long long seed, unsigned long long offset, hipStream_t s) {
  const long long nb = (n * d + 1) / 2;
  hipLaunchKernelGGL(k_generate_tbox, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, t, n * d, d, lo, hi, pad, seed,
                     offset);
}

void launch_generate_around_points(double *t, double *thin_u, long long n, int d, const double *refR, int nlive, int dp,
                                   double r2, unsigned long long seed, unsigned long long offset, hipStream_t s) {
  if (n <= 0) return;
  const int npairs = (d + 1) / 2;
  const unsigned pmagic = npairs > 1 ? (unsigned)((0x100000000ull + (unsigned)npairs - 1) / (unsigned)npairs) : 0u;
  const unsigned dmagic = d > 1 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u;
  const size_t lds = ((size_t)64 * (d + 1) + 64) * sizeof(double) + 64 * sizeof(unsigned);
  static DeviceGrant grant;
  (void)grant.ensure([] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_generate_around_points), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  });
  hipLaunchKernelGGL(k_generate_around_points, dim3((unsigned)((n + 63) / 64)), dim3(256), lds, s, t, thin_u, n, d, refR,
                     nlive, dp, r2, seed, offset, pmagic, dmagic);
}

void launch_thin_by_multiplicity(const long long *count, const double *thin_u, long long n, uint8_t *mask, hipStream_t s) {
  hipLaunchKernelGGL(k_thin_by_multiplicity, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, count, thin_u, n, mask);
}

void launch_untransform_rows(const double *t, long long n, int d, const double *invT, const double *ctr,
                             const double *wrap_shift, double *w, uint8_t *in_cube, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_untransform_rows, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, t, n, d, invT, ctr, wrap_shift, w,
                     in_cube);
}

void launch_elementwise_affine(const double *x, long long n, int tkind, double a, double b, double *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_elementwise_affine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, tkind, a, b, out);
}

void launch_mask_greater(const double *v, long long n, double threshold, uint8_t *mask, hipStream_t s, const uint8_t *also) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_mask_greater, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n, threshold, also, mask);
}

void launch_mask_and(uint8_t *mask, const uint8_t *other, long long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_mask_and, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mask, other, n);
}

void launch_apply_pregate(const uint8_t *pregate, long long n, uint8_t *gate, uint8_t *route, float *tlo,
                          float *thi, hipStream_t s) {
  hipLaunchKernelGGL(k_apply_pregate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pregate, n, gate, route,
                     tlo, thi);
}

// exclusive scan of nblk per-block counts in place, total -> blk[nblk]
void launch_scan_counts(unsigned *blk, int nblk, hipStream_t s) {
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, blk, nblk);
}

// blk[i] = number of set mask bytes before block i (256 entries per block), blk[nblk] = total
void launch_mask_offsets(const uint8_t *mask, long long n, unsigned *blk, hipStream_t s) {
  const int nblk = (int)((n + 255) / 256);
  hipLaunchKernelGGL(k_count_accepted, dim3(nblk), dim3(256), 0, s, mask, n, blk);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, blk, nblk);
}

void launch_compact(const double *pts, const uint8_t *mask, long long n, int d, unsigned *blk, double *out,
                    unsigned capacity, hipStream_t s) {
  launch_mask_offsets(mask, n, blk, s);
  launch_scatter(pts, mask, n, d, blk, out, capacity, s);
}

// the scatter alone: blk holds the offsets of THIS mask (launch_mask_offsets); several arrays compacted by one mask share them
void launch_scatter(const double *pts, const uint8_t *mask, long long n, int d, const unsigned *blk, double *out,
                    unsigned capacity, hipStream_t s) {
  const int nblk = (int)((n + 255) / 256);
  const unsigned dmagic = d > 1 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u;
  if (d == 1)
    hipLaunchKernelGGL(k_scatter_scalars, dim3(nblk), dim3(256), 0, s, pts, mask, n, blk, out, capacity);
  else
    hipLaunchKernelGGL(k_scatter_accepted, dim3(nblk), dim3(256), 0, s, pts, mask, n, d, dmagic, blk, out, capacity);
}

}  // namespace mlf
