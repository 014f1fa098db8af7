// mlf_misc.hip -- layout, reduction and likelihood kernels around the two distance kernels.
// Compiled with -ffp-contract=off; fused multiply-adds appear only where written as fma().
#include "mlf_misc.hpp"
#include "mlf_loglike_dev.hpp"

#include <math.h>

#include "mlf_dpp_dev.hpp"

namespace mlf {

// ---------------------------------------------------------------- layouts ----------------
// row-major (n, d)  ->  refT [dp][npad] (coordinate-major) and refR [npad][dp], zero padded
__global__ void k_build_layouts(const double *src, int n, int d, int dp, int npad, double *refT,
                                double *refR) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)npad * dp;
  if (e >= total) return;
  // thread e covers refR element (i, k) and refT element (k', i') of the same flat index
  {
    const int i = (int)(e / dp), k = (int)(e - (long long)i * dp);
    refR[e] = (i < n && k < d) ? src[(long long)i * d + k] : 0.0;
  }
  {
    const int k = (int)(e / npad), i = (int)(e - (long long)k * npad);
    refT[e] = (i < n && k < d) ? src[(long long)i * d + k] : 0.0;
  }
}

void launch_build_layouts(const double *src, int n, int d, int dp, int npad, double *refT,
                          double *refR, hipStream_t s) {
  const long long total = (long long)npad * dp;
  hipLaunchKernelGGL(k_build_layouts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n,
                     d, dp, npad, refT, refR);
}

__global__ void k_update_row(const double *rows, int d, int dp, int npad, const long long *index, double *refT,
                             double *refR) {
  const int k = threadIdx.x;
  if (k >= dp) return;
  const long long i = index[blockIdx.x];
  const double v = k < d ? rows[(long long)blockIdx.x * d + k] : 0.0;
  refR[i * dp + k] = v;
  refT[(long long)k * npad + i] = v;
}

// `count` whitened rows (count x d, packed) replace the live points index[0..count) in both layouts
void launch_update_rows(const double *rows, int count, int d, int dp, int npad, const long long *index, double *refT,
                        double *refR, hipStream_t s) {
  if (dp > 128) return launch_update_rows_wide(rows, count, d, dp, npad, index, refT, refR, s);
  hipLaunchKernelGGL(k_update_row, dim3((unsigned)count), dim3(128), 0, s, rows, d, dp, npad, index, refT, refR);
}

__global__ void k_fill_u64(unsigned long long *p, long long n, unsigned long long v) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = v;
}

void launch_fill_u64(unsigned long long *p, long long n, unsigned long long v, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

// (B, n) byte masks -> one 32-bit word per live point, bit r = selected in round b0 + r; and, for k_boot, the same bits
// expanded to one word per (live point, round): 0 = selected, 0xffffffff = not selected (OR-ed into the high word of a
// candidate distance, the latter turns it into a NaN that the running minimum ignores)
__global__ void k_pack_selection(const uint8_t *selected, int n, int npad, int b0, int nb,
                                 unsigned *sel, unsigned *selmask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  unsigned w = 0;
  if (i < n)
    for (int r = 0; r < nb; ++r)
      if (selected[(long long)(b0 + r) * n + i]) w |= 1u << r;
  sel[i] = w;
  if (selmask) {
    uint4 *dst = reinterpret_cast<uint4 *>(selmask + (size_t)i * kBootGroup);
#pragma unroll
    for (int q = 0; q < kBootGroup / 4; ++q) {
      const unsigned nib = ~(w >> (4 * q));
      dst[q] = make_uint4((nib & 1u) ? ~0u : 0u, (nib & 2u) ? ~0u : 0u, (nib & 4u) ? ~0u : 0u, (nib & 8u) ? ~0u : 0u);
    }
  }
}

void launch_pack_selection(const uint8_t *selected, int n, int npad, int b0, int nb, unsigned *sel,
                           hipStream_t s, unsigned *selmask) {
  hipLaunchKernelGGL(k_pack_selection, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s,
                     selected, n, npad, b0, nb, sel, selmask);
}

// ------------------------------------------------------- K4 epilogue ---------------------
// maxd[r] = (double)(float) max over unselected j of M[r][j]   (reference :222-224: the
// `cdef float` return narrows to binary32, round-to-nearest-even = v_cvt_f32_f64)
// Row-block sharding (one rank of several): the maximum runs over the rows [row_lo, row_hi) only; `skipped` still looks at
// all n rows (it is a property of the round).  Narrowing is monotone, so max over ranks of these = the one-rank value.
__global__ __launch_bounds__(256) void k_boot_final(const unsigned long long *M, const unsigned *sel,
                                                    int n, int npad, double *maxd,
                                                    uint8_t *skipped, int row_lo, int row_hi) {
  __shared__ unsigned long long smax[256];
  __shared__ int scnt[256];
  const int r = blockIdx.x;
  unsigned long long best = 0ull;  // +0.0, reference maxd = 0 (:209)
  int nsel = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    if ((sel[j] >> r) & 1u) {
      ++nsel;
    } else if (j >= row_lo && j < row_hi) {
      const unsigned long long v = M[(long long)r * npad + j];
      best = v > best ? v : best;
    }
  }
  smax[threadIdx.x] = best;
  scnt[threadIdx.x] = nsel;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      const unsigned long long o = smax[threadIdx.x + w];
      if (o > smax[threadIdx.x]) smax[threadIdx.x] = o;
      scnt[threadIdx.x] += scnt[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool skip = scnt[0] == 0 || scnt[0] == n;  // reference :1048
    const double m = __longlong_as_double((long long)smax[0]);
    maxd[r] = skip ? 0.0 : (double)(float)m;
    skipped[r] = skip ? 1 : 0;
  }
}

void launch_boot_final(const unsigned long long *M, const unsigned *sel, int n, int npad, int nb,
                       double *maxd, uint8_t *skipped, hipStream_t s, int row_lo, int row_hi) {
  if (row_hi < 0) row_hi = n;
  hipLaunchKernelGGL(k_boot_final, dim3(nb), dim3(256), 0, s, M, sel, n, npad, maxd, skipped, row_lo, row_hi);
}

// ------------------------------------------------------- constants of a region, one launch ---
// kScatterSplit workgroups per segment (grid.y), each with its share of the words: the source is pinned HOST memory read over
// the fabric, one workgroup per segment walked its 10-30 KB in dependent 2 KB steps (19 us for the longest segment of
// mlf_region_set, two such launches per call)
constexpr unsigned kScatterSplit = 8;
__global__ __launch_bounds__(256) void k_scatter_copy(ScatterArgs a) {
  const int seg = blockIdx.x;
  unsigned char *dst = static_cast<unsigned char *>(a.dst[seg]);
  const unsigned char *src = static_cast<const unsigned char *>(a.src[seg]);
  const unsigned bytes = a.bytes[seg];
  const unsigned words = bytes / 8u;   // both ends are 8-byte aligned (64-byte arena pieces, 256-byte device buffers)
  for (unsigned w = blockIdx.y * 256u + threadIdx.x; w < words; w += 256u * kScatterSplit)
    reinterpret_cast<unsigned long long *>(dst)[w] = reinterpret_cast<const unsigned long long *>(src)[w];
  if (blockIdx.y == 0)
    for (unsigned t = words * 8u + threadIdx.x; t < bytes; t += 256) dst[t] = src[t];
}

void launch_scatter_copy(const ScatterArgs &a, int count, hipStream_t s) {
  if (count > 0) hipLaunchKernelGGL(k_scatter_copy, dim3((unsigned)count, kScatterSplit), dim3(256), 0, s, a);
}

// ------------------------------------------------------- K3 pass 2 -----------------------
// One wave per point j, lane = coordinate (two per lane up to d = 128).  Neighbours are visited in ascending i
// from the hit ballots, so the sum has the reference's order (:100-109); then pts[j] - sum / (double)nn.
// A workgroup holds 16 points (8 waves x 2) and stages every 64-row tile of pts in LDS once for all of them: with the
// LocalAffineLayer radius quirk every point is a neighbour of every point, and one wave per workgroup re-read
// the whole array from L2 for each point (6.4 GB at N = 4000: 0.48 ms).
constexpr int kAccumWaves = 4;
constexpr int kAccumPer = 4;   // points per wave: one LDS read (4 cycles of the CU's LDS port per wave) feeds four sums (16 cycles of its SIMD)
template <int H>               // 64-column halves (d <= 64 H); lanes past d work on column 0, their sums are not stored
__global__ __launch_bounds__(64 * kAccumWaves) void k_subtract_accum(const double *__restrict__ pts, int n, int d,
                                                                    const unsigned long long *__restrict__ flags,
                                                                    int ntiles, double *__restrict__ out) {
  extern __shared__ double tile[];   // [64][d]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = (blockIdx.x * kAccumWaves + wave) * kAccumPer;
  int col[H];
#pragma unroll
  for (int h = 0; h < H; ++h) col[h] = lane + 64 * h < d ? lane + 64 * h : 0;
  double sum[kAccumPer][H];
  long long nn[kAccumPer];
#pragma unroll
  for (int p = 0; p < kAccumPer; ++p) {
#pragma unroll
    for (int h = 0; h < H; ++h) sum[p][h] = 0.0;
    nn[p] = 0;
  }
  // the next tile travels in registers while this one is consumed (two tiles ahead: no change)
  constexpr int kPer = (kWave * 64 * H + 64 * kAccumWaves - 1) / (64 * kAccumWaves);
  double nxt[kPer];
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int rows = (n - t * kWave) < kWave ? (n - t * kWave) : kWave;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int e = threadIdx.x + q * 64 * kAccumWaves;
      nxt[q] = pts[(long long)t * kWave * d + (e < rows * d ? e : 0)];
    }
  };
  fetch(0);
  unsigned long long fl[kAccumPer];
  auto one_tile = [&](int t) __attribute__((always_inline)) {
    const int rows = (n - t * kWave) < kWave ? (n - t * kWave) : kWave;
    if ((t & 63) == 0) {
#pragma unroll
      for (int p = 0; p < kAccumPer; ++p)
        fl[p] = (j0 + p < n && t + lane < ntiles) ? flags[(long long)(j0 + p) * ntiles + t + lane] : 0ull;
    }
    // the hit words of 64 tiles at a time, lane = tile, picked up HERE: behind the requests for the next tile any wait for
    // them is a wait for that tile as well (the counter is in order), and the prefetch hid nothing
    unsigned long long m[kAccumPer];
    unsigned long long any = 0ull, all = ~0ull;
#pragma unroll
    for (int p = 0; p < kAccumPer; ++p) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)fl[p], t & 63);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(fl[p] >> 32), t & 63);
      m[p] = ((unsigned long long)hi << 32) | lo;
      any |= m[p];
      all &= m[p];
    }
    __syncthreads();   // the previous tile is consumed
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int e = threadIdx.x + q * 64 * kAccumWaves;
      if (e < rows * d) tile[e] = nxt[q];
    }
    if (t + 1 < ntiles) fetch(t + 1);
    __syncthreads();
    if (all == ~0ull) {   // whole tile for every point of this wave: no bit scanning
      // 16 rows per batch, the next batch's LDS reads issued before this batch's adds (left to the scheduler, five reads
      // were in flight per wave and every row waited for its own)
      constexpr int kB = 16;
      double va[kB][H], vb[kB][H];
      auto rd = [&](double (&v)[kB][H], int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < kB; ++q)
#pragma unroll
          for (int h = 0; h < H; ++h) v[q][h] = tile[(q0 + q) * d + col[h]];
        __builtin_amdgcn_sched_barrier(0);
      };
      auto add = [&](const double (&v)[kB][H]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < kB; ++q)
#pragma unroll
          for (int h = 0; h < H; ++h)
#pragma unroll
            for (int p = 0; p < kAccumPer; ++p) sum[p][h] += v[q][h];
        __builtin_amdgcn_sched_barrier(0);
      };
      static_assert(kWave == 4 * kB, "four batches per tile");
      rd(va, 0);
      rd(vb, kB);
      add(va);
      rd(va, 2 * kB);
      add(vb);
      rd(vb, 3 * kB);
      add(va);
      add(vb);
#pragma unroll
      for (int p = 0; p < kAccumPer; ++p) nn[p] += kWave;
      return;
    }
    while (any) {   // ascending row order over the union; each point adds only its own neighbours
      const int q = __ffsll((long long)any) - 1;
      any &= any - 1;
      double v[H];
#pragma unroll
      for (int h = 0; h < H; ++h) v[h] = tile[q * d + col[h]];
#pragma unroll
      for (int p = 0; p < kAccumPer; ++p)
        if ((m[p] >> q) & 1ull) {   // wave-uniform
#pragma unroll
          for (int h = 0; h < H; ++h) sum[p][h] += v[h];
          ++nn[p];
        }
    }
  };
  for (int t = 0; t < ntiles; ++t) one_tile(t);
#pragma unroll
  for (int p = 0; p < kAccumPer; ++p) {
    const int j = j0 + p;
    if (j >= n) continue;
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (lane + 64 * h < d)
        out[(long long)j * d + lane + 64 * h] = pts[(long long)j * d + lane + 64 * h] - sum[p][h] / (double)nn[p];
  }
}

void launch_subtract_accum(const double *pts, int n, int d, const unsigned long long *flags,
                           int ntiles, double *out, hipStream_t s) {
  if (d > 128) return launch_subtract_wide(pts, n, d, flags, ntiles, out, s);
  const int per_block = kAccumWaves * kAccumPer;
  const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
  const size_t lds = (size_t)kWave * d * sizeof(double);
  if (d <= 64)
    hipLaunchKernelGGL(k_subtract_accum<1>, dim3(grid), dim3(64 * kAccumWaves), lds, s, pts, n, d, flags, ntiles, out);
  else
    hipLaunchKernelGGL(k_subtract_accum<2>, dim3(grid), dim3(64 * kAccumWaves), lds, s, pts, n, d, flags, ntiles, out);
}

// ------------------------------------------------------- K5 ------------------------------
// packed strict lower triangle of squared pair distances; diff = pts[i] - pts[j] (:265)
__global__ __launch_bounds__(256) void k_pair_dist2_lower(const double *pts, int n, int d,
                                                           double *out) {
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);
  const int j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (j >= n || i >= j) return;
  const double *pi = pts + (long long)i * d;
  const double *pj = pts + (long long)j * d;
  double acc = 0.0;
  for (int k = 0; k < d; ++k) {
    const double df = pi[k] - pj[k];
    acc += df * df;
  }
  out[(long long)j * (j - 1) / 2 + i] = acc;
}

// ------------------------------------------------------- column extents (device-resident rebuild) ---------
// lo[c] = min over rows of pts[row][c], hi[c] = max: the bounding box of the whitened live points (mlfriends.pyx:969-970),
// and -- on the unit-cube points -- the cube test of the driver (integrator.py:2059-2061) and the largest |u - ctr|.
// Stage 1: workgroup b scans rows b, b + G, ...; thread = (row within a step, column); partial extents per workgroup.
// Stage 2: one workgroup folds the G partials.  min / max are exact: any order gives the same bits.
constexpr int kExtentBlocks = 64;
__global__ __launch_bounds__(256) void k_col_extent(const double *__restrict__ pts, int n, int d, double *__restrict__ part) {
  __shared__ double slo[256], shi[256];
  const int rows_per_step = 256 / d > 0 ? 256 / d : 1;   // d <= 128: at least two rows per step
  const int r = threadIdx.x / d, c = threadIdx.x - r * d;
  double lo = INFINITY, hi = -INFINITY;
  if (r < rows_per_step)
    for (long long row = (long long)blockIdx.x * rows_per_step + r; row < n; row += (long long)gridDim.x * rows_per_step) {
      const double v = pts[row * d + c];
      lo = (v < lo || v != v) ? v : lo;      // a NaN stays (every later comparison is false): numpy's min / max propagate it too
      hi = (v > hi || v != v) ? v : hi;
    }
  slo[threadIdx.x] = lo;
  shi[threadIdx.x] = hi;
  __syncthreads();
  if (threadIdx.x < d) {
    for (int rr = 1; rr < rows_per_step; ++rr) {
      const double l2 = slo[rr * d + threadIdx.x], h2 = shi[rr * d + threadIdx.x];
      lo = (l2 < lo || l2 != l2) ? l2 : lo;
      hi = (h2 > hi || h2 != h2) ? h2 : hi;
    }
    part[(size_t)blockIdx.x * 2 * d + threadIdx.x] = lo;
    part[(size_t)blockIdx.x * 2 * d + d + threadIdx.x] = hi;
  }
}
__global__ __launch_bounds__(128) void k_col_extent_fold(const double *__restrict__ part, int nblk, int d, double *__restrict__ out) {
  const int c = threadIdx.x;
  if (c >= d) return;
  double lo = INFINITY, hi = -INFINITY;
  for (int b = 0; b < nblk; ++b) {
    const double l2 = part[(size_t)b * 2 * d + c], h2 = part[(size_t)b * 2 * d + d + c];
    lo = (l2 < lo || l2 != l2) ? l2 : lo;
    hi = (h2 > hi || h2 != h2) ? h2 : hi;
  }
  out[c] = lo;
  out[d + c] = hi;
}
// part: kExtentBlocks * 2 * d doubles of scratch; out: 2 * d doubles (lo then hi)
void launch_col_extent(const double *pts, int n, int d, double *part, double *out, hipStream_t s) {
  hipLaunchKernelGGL(k_col_extent, dim3(kExtentBlocks), dim3(256), 0, s, pts, n, d, part);
  hipLaunchKernelGGL(k_col_extent_fold, dim3(1), dim3(128), 0, s, part, kExtentBlocks, d, out);
}

void launch_pair_dist2_lower(const double *pts, int n, int d, double *out, hipStream_t s) {
  const unsigned g = (unsigned)((n + 15) / 16);
  hipLaunchKernelGGL(k_pair_dist2_lower, dim3(g, g), dim3(256), 0, s, pts, n, d, out);
}

// ------------------------------------------------------- ScalingLayer.transform ----------
// t = (wrap(u) - mean) / std, elementwise (reference :605-611); bit-exact (IEEE division)
__global__ void k_scaling_transform(const double *pts, long long np, int d, const double *mean,
                                    const double *std, const double *wrap_shift,
                                    const uint8_t *gate, double *out, long long ldt) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np * d) return;
  const long long p = e / d;
  const int k = (int)(e - p * d);
  if (gate && !gate[p]) return;
  double w = pts[e];
  if (wrap_shift) {
    const double sh = wrap_shift[k];
    if (sh == sh) w = fmod(w + sh, 1.0);
  }
  out[p * ldt + k] = (w - mean[k]) / std[k];
}

void launch_scaling_transform(const double *pts, long long np, int d, const double *mean,
                              const double *std, const double *wrap_shift, const uint8_t *gate,
                              double *out, long long ldt, hipStream_t s) {
  const long long total = np * d;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_scaling_transform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pts,
                     np, d, mean, std, wrap_shift, gate, out, ldt);
}

// ------------------------------------------------------- masked max ----------------------
// out[0] = max over i with selected[i] == 0 of q[i]  (bootstrap enlargement f, reference :1062)
__global__ __launch_bounds__(256) void k_masked_max(const double *q, const uint8_t *selected, int n,
                                                    double *out) {
  __shared__ double smax[256];
  double best = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256)
    if (!selected[i]) best = fmax(best, q[i]);
  smax[threadIdx.x] = best;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = smax[0];
}

void launch_masked_max(const double *q, const uint8_t *selected, int n, double *out, hipStream_t s) {
  hipLaunchKernelGGL(k_masked_max, dim3(1), dim3(256), 0, s, q, selected, n, out);
}

// ------------------------------------------------------- bootstrap moments ---------------
// The selection masks are first turned into ascending index lists (one workgroup per bootstrap round):
// looping over the mask itself made every row a load -> branch -> load chain (0.34 ms per kernel at
// N = 4000, B = 30, for 6 Mflop of work); over a list the row loads are independent and unrolled.
__global__ __launch_bounds__(256) void k_boot_index(const uint8_t *selected, int n, int *idx, int *count) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t *sel = selected + (long long)b * n;
  int *out = idx + (long long)b * n;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool on = i < n && sel[i] != 0;
    const unsigned long long m = __ballot(on);
    if (lane == 0) wsum[wave] = (int)__popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (on) out[off + (int)__popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[b] = base;
}

// mean[b][k] over the selected rows: workgroup = bootstrap b, 16 waves take the list entries j = g (mod 16)
// in ascending order, lane = coordinate (two per lane up to d = 128); the 16 partial sums are added in a
// fixed order
template <int H>   // 64-column halves (d <= 64 H); lanes past d work on column 0 and are not stored
__global__ __launch_bounds__(1024) void k_boot_mean(const double *__restrict__ u, int d, const int *__restrict__ idx, int n,
                                                    const int *__restrict__ count, double *__restrict__ mean) {
  __shared__ double part[16][64 * H];
  const int b = blockIdx.x;
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int *list = idx + (long long)b * n;
  const int cnt = count[b];
  int col[H];
  double sum[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    col[h] = lane + 64 * h < d ? lane + 64 * h : 0;
    sum[h] = 0.0;
  }
  constexpr int kFly = 8;   // rows in flight
  for (int t0 = 0; g + 16 * t0 < cnt; t0 += 64) {
    const int jt = g + 16 * (t0 + lane);
    const int mine = jt < cnt ? list[jt] : 0;   // this wave's next 64 list entries (j = g mod 16), one per lane
    int left = (cnt - g - 16 * t0 + 15) / 16;
    if (left > 64) left = 64;
    for (int t = 0; t < left; t += kFly) {   // entries past `left` repeat entry t and are left out of the sum
      double v[kFly][H];
#pragma unroll
      for (int q = 0; q < kFly; ++q) {
        const long long r = __builtin_amdgcn_readlane(mine, t + q < left ? t + q : t);
#pragma unroll
        for (int h = 0; h < H; ++h) v[q][h] = u[r * d + col[h]];
      }
#pragma unroll
      for (int q = 0; q < kFly; ++q)
        if (t + q < left) {   // wave-uniform
#pragma unroll
          for (int h = 0; h < H; ++h) sum[h] += v[q][h];
        }
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) part[g][lane + 64 * h] = sum[h];
  __syncthreads();
  if (threadIdx.x < d) {
    const int k = threadIdx.x;
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += part[w][k];
    mean[(long long)b * d + k] = tot / (double)cnt;
  }
}

// cov[b][k][l] = sum_sel (u_ik - m_k)(u_il - m_l) / (cnt - 1).  Workgroup = (bootstrap b, kCovRows matrix rows k),
// 8 waves over the list entries j = g (mod 8), lane = column l (two per lane up to d = 128).  A selected row is
// read once per workgroup and serves all kCovRows values of k (its u_ik come from the same registers by lane
// broadcast): one workgroup per single k re-read the selected rows from L2 d times per round (1.8 GB, 0.24 ms at
// N = 4000, d = 50, B = 30).  Per (k, l) the products are accumulated in ascending list order within a wave and the
// 8 partial sums are added in a fixed order.
constexpr int kCovRows = 8;
constexpr int kCovWaves = 8;   // 8 x 8 x 128 partial sums = 64 KB of LDS

__device__ __forceinline__ double readlane_f64(double v, int l) {   // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// H = number of 64-column halves (d <= 64 H).  Lanes past d work on column 0 (their sums are never stored): the
// row loads and the accumulation carry no per-lane conditions.
template <int H>
__global__ __launch_bounds__(64 * kCovWaves) void k_boot_cov(const double *__restrict__ u, int d, const int *__restrict__ idx,
                                                             int n, const double *__restrict__ mean,
                                                             const int *__restrict__ count, double *__restrict__ cov) {
  __shared__ double part[kCovWaves][kCovRows][64 * H];
  const int b = blockIdx.y, k0 = blockIdx.x * kCovRows;
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int *list = idx + (long long)b * n;
  const int cnt = count[b];
  int col[H];
  double ml[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    col[h] = lane + 64 * h < d ? lane + 64 * h : 0;
    ml[h] = mean[(long long)b * d + col[h]];
  }
  double acc[kCovRows][H];
#pragma unroll
  for (int q = 0; q < kCovRows; ++q)
#pragma unroll
    for (int h = 0; h < H; ++h) acc[q][h] = 0.0;
  for (int t0 = 0; g + kCovWaves * t0 < cnt; t0 += 64) {
    const int jt = g + kCovWaves * (t0 + lane);
    const int mine = jt < cnt ? list[jt] : 0;   // this wave's next 64 list entries, one per lane
    int left = (cnt - g - kCovWaves * t0 + kCovWaves - 1) / kCovWaves;
    if (left > 64) left = 64;
    int t = 0;
    for (; t + 1 < left; t += 2) {              // two rows in flight
      const long long r0 = __builtin_amdgcn_readlane(mine, t), r1 = __builtin_amdgcn_readlane(mine, t + 1);
      double x0[H], x1[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        x0[h] = u[r0 * d + col[h]];
        x1[h] = u[r1 * d + col[h]];
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        x0[h] -= ml[h];
        x1[h] -= ml[h];
      }
#pragma unroll
      for (int q = 0; q < kCovRows; ++q) {
        const int k = k0 + q < d ? k0 + q : 0;    // wave-uniform; the centred u_ik sits in lane k & 63 of half k >> 6
        const double dk0 = (H == 1 || k < 64) ? readlane_f64(x0[0], k & 63) : readlane_f64(x0[H - 1], k & 63);
        const double dk1 = (H == 1 || k < 64) ? readlane_f64(x1[0], k & 63) : readlane_f64(x1[H - 1], k & 63);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          acc[q][h] = __builtin_fma(dk0, x0[h], acc[q][h]);
          acc[q][h] = __builtin_fma(dk1, x1[h], acc[q][h]);
        }
      }
    }
    if (t < left) {
      const long long r0 = __builtin_amdgcn_readlane(mine, t);
      double x0[H];
#pragma unroll
      for (int h = 0; h < H; ++h) x0[h] = u[r0 * d + col[h]] - ml[h];
#pragma unroll
      for (int q = 0; q < kCovRows; ++q) {
        const int k = k0 + q < d ? k0 + q : 0;
        const double dk0 = (H == 1 || k < 64) ? readlane_f64(x0[0], k & 63) : readlane_f64(x0[H - 1], k & 63);
#pragma unroll
        for (int h = 0; h < H; ++h) acc[q][h] = __builtin_fma(dk0, x0[h], acc[q][h]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kCovRows; ++q)
#pragma unroll
    for (int h = 0; h < H; ++h) part[g][q][lane + 64 * h] = acc[q][h];
  __syncthreads();
  for (int e = threadIdx.x; e < kCovRows * 64 * H; e += 64 * kCovWaves) {
    const int q = e / (64 * H), l = e % (64 * H);
    if (k0 + q < d && l < d) {
      double tot = 0.0;
      for (int w = 0; w < kCovWaves; ++w) tot += part[w][q][l];
      cov[((long long)b * d + k0 + q) * d + l] = tot / (double)(cnt - 1);
    }
  }
}

// d <= 64: the same sums with the operand broadcast the instruction set has for binary64 products.  A wave takes FOUR
// selected rows per step, one per group of 16 lanes; lane l of a group holds the centred coordinates 16 c + (l mod 16),
// c = 0 ... NCH - 1, and acc[kk][c] += x[16 CK + kk] * x[16 c + l mod 16] is ONE v_fmac_f64 whose first operand comes from
// lane kk of the lane's own group through DPP (row_newbcast): no v_readlane pair per (row, k), every lane busy, and a
// workgroup (8 waves) owns a whole 16-row block CK of the matrix for one round, the upper chunks c >= CK only (the lower
// ones are the mirror image).  Per (k, l) the products of a wave's group are added in ascending list order; the four
// groups, then the eight waves, in a fixed order: deterministic, tolerance class of the moments (1e-9 relative).
// 30 rounds at N = 4000, d = 50: 0.076 -> 0.044 ms (the v_readlane version read every selected row in 7 workgroups; here
// only 120 workgroups exist and block CK = 0 has four times the work of CK = 3).
template <int NCH, int CK>
__device__ __forceinline__ void cov_block(const double *__restrict__ u, int d, const int *__restrict__ list, int cnt,
                                          const double *__restrict__ mean_b, double *__restrict__ cov_b, double (*part)[16][4][16]) {
  constexpr int NC = NCH - CK;   // chunks c = CK ... NCH - 1
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  int col[NCH];
  double ml[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    col[c] = 16 * c + sub < d ? 16 * c + sub : 0;
    ml[c] = mean_b[col[c]];
  }
  double acc[16][NC];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[kk][c] = 0.0;
  // list entries of this wave: 4 (g + kCovWaves t) + grp, t = 0, 1, ...  Sixteen steps at a time: lane `sub` of a group
  // loads the group's entry of step t0 + sub, each step picks its row index up by DPP broadcast inside the group, and the
  // rows of the next kAhead steps are on their way while a step's products are formed (the row loads depend on the
  // list loads: one step ahead left the wave waiting ~0.5 us per step)
  constexpr int kAhead = 7;
  const int steps = (cnt + 4 * kCovWaves - 1) / (4 * kCovWaves);   // wave-uniform
  for (int t0 = 0; t0 < steps; t0 += 16) {
    const int e = 4 * (g + kCovWaves * (t0 + sub)) + grp;
    const int mine = e < cnt ? list[e] : -1;
    double x[8][NCH];
    int row[8];
    auto fetch = [&](auto tc) __attribute__((always_inline)) {
      constexpr int tt = decltype(tc)::value;
      // no branch on "is there such a step": steps past the list load row 0 and add zeros (a branch per step makes every
      // wait a wait for ALL outstanding loads: the counter is not tracked across the blocks)
      row[tt & 7] = (int)row_bcast<tt>((unsigned)mine);
      const long long r = row[tt & 7] < 0 ? 0 : row[tt & 7];
#pragma unroll
      for (int c = 0; c < NCH; ++c) x[tt & 7][c] = u[r * d + col[c]];
    };
    static_for<0, kAhead>(fetch);
    static_for<0, 16>([&](auto tc) __attribute__((always_inline)) {
      constexpr int tt = decltype(tc)::value;
      if constexpr (tt + kAhead < 16) fetch(std::integral_constant<int, tt + kAhead>{});
      {
        double(&xs)[NCH] = x[tt & 7];
        const bool ok = row[tt & 7] >= 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) xs[c] = ok ? xs[c] - ml[c] : 0.0;
        // v_fmac_f64_dpp reads xs[CK] through DPP: a register just written by a vector instruction needs two wait states
        dpp_settle(xs[CK]);
        static_for<0, 16>([&](auto kc) __attribute__((always_inline)) {
          constexpr int kk = decltype(kc)::value;
#pragma unroll
          for (int c = 0; c < NC; ++c) fmac_row_bcast<kk>(acc[kk][c], xs[CK], xs[CK + c]);
        });
      }
    });
  }
  // the four groups of the wave, then the waves of the workgroup
#pragma unroll
  for (int kk = 0; kk < 16; ++kk)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double v = acc[kk][c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (grp == 0) part[g][kk][c][sub] = v;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * NC * 16; e += 64 * kCovWaves) {
    const int kk = e / (NC * 16), c = (e / 16) % NC, l = e % 16;
    const int k = 16 * CK + kk, col_l = 16 * (CK + c) + l;
    if (k < d && col_l < d) {
      double tot = 0.0;
      for (int w = 0; w < kCovWaves; ++w) tot += part[w][kk][c][l];
      tot /= (double)(cnt - 1);
      cov_b[(long long)k * d + col_l] = tot;
      if (c > 0) cov_b[(long long)col_l * d + k] = tot;   // the mirror image of an off-diagonal chunk
    }
  }
}

template <int NCH>   // d <= 16 NCH <= 64
__global__ __launch_bounds__(64 * kCovWaves) void k_boot_cov16(const double *__restrict__ u, int d, const int *__restrict__ idx,
                                                               int n, const double *__restrict__ mean,
                                                               const int *__restrict__ count, double *__restrict__ cov) {
  __shared__ double part[kCovWaves][16][4][16];   // 64 KB
  const int b = blockIdx.y, ck = blockIdx.x;
  const int *list = idx + (long long)b * n;
  const double *mean_b = mean + (long long)b * d;
  double *cov_b = cov + (long long)b * d * d;
  const int cnt = count[b];
  if (ck == 0) cov_block<NCH, 0>(u, d, list, cnt, mean_b, cov_b, part);
  if constexpr (NCH > 1) if (ck == 1) cov_block<NCH, 1>(u, d, list, cnt, mean_b, cov_b, part);
  if constexpr (NCH > 2) if (ck == 2) cov_block<NCH, 2>(u, d, list, cnt, mean_b, cov_b, part);
  if constexpr (NCH > 3) if (ck == 3) cov_block<NCH, 3>(u, d, list, cnt, mean_b, cov_b, part);
}

// Bootstrap enlargement without leaving the device (reference mlfriends.pyx:1056-1066 with minvol = 0):
// f_b = max over the left-out rows of (u_i - m_b)^T (scale cov_b)^-1 (u_i - m_b).  Two launches:
//   k_boot_chol      one WAVE per round: scale cov_b = L L^T, lane = matrix row held in registers (see the kernel)
//   k_boot_solvemax  one LANE per row: solves L y = u_i - m_b, f = |y|^2, per-round maximum over the left-out rows by
//                    atomicMax on the bit pattern (see the kernel)
// The host version inverted the B matrices with LAPACK (1.1 ms at B = 30, d = 50) between two device calls.  Result
// class: like the reference's inv + einsum to rounding (tolerance class of `f`); a round whose matrix is not positive
// definite or not finite returns NaN.  Same sequence of roundings in every version so far (right-looking, column by column).

// Lout: [B][dpc][dpc + 1] (lower triangle incl. diagonal; column dpc holds 1 / L_rr); bad[b] != 0: not positive definite.
// One wave per round, lane = matrix row, the row in registers SHIFTED left by one per column: at step j register c holds
// element (r, j + c) of the trailing matrix, so the pivot column is always register 0 and the loop over the columns is a
// run-time loop whose body (one column: pivot by v_readlane, then per remaining column the owner's L_cj by v_readlane and
// a_rc = fma(-L_rj, L_cj, a_rc), written one register to the left) is compiled once.  Element (r, c) receives its updates
// for j = 0, 1, ... in this order, L_rj = a_rj / sqrt(a_jj): the roundings of the first version of this kernel (both loops
// unrolled with the row in place: d^2 / 2 x 3 instructions per size class and 3.3 minutes of compile time for four
// classes); results bit-identical.
template <int DPC>   // registers per row: d <= dpc <= DPC
__global__ __launch_bounds__(64) void k_boot_chol(const double *__restrict__ cov, int d, int dpc, double scale,
                                                  double *__restrict__ Lout, int *__restrict__ bad_out) {
  const int r = threadIdx.x, b = blockIdx.x;
  double a[DPC];
#pragma unroll
  for (int c = 0; c < DPC; ++c)
    a[c] = (r < d && c < d) ? cov[((size_t)b * d + (r < d ? r : 0)) * d + (c < d ? c : 0)] * scale : (r == c ? 1.0 : 0.0);
  bool bad = false;
  double invd = 1.0;
  double *dst = Lout + ((size_t)b * dpc + (r < dpc ? r : 0)) * (dpc + 1);
  for (int j = 0; j < dpc; ++j) {
    const double p = readlane_f64(a[0], j);   // the pivot: element (j, j)
    if (!(p > 0.0) || !(p < 1e300)) bad = true;
    const double sp = sqrt(p > 0.0 ? p : 1.0);
    const double ip = 1.0 / sp;
    double lr = a[0];
    if (r == j) {
      lr = sp;
      invd = ip;
    } else if (r > j) {
      lr *= ip;
    }
    if (r < dpc) dst[j] = r >= j ? lr : 0.0;
#pragma unroll
    for (int c = 1; c < DPC; ++c) {
      if ((c & 7) == 1 && j + c >= dpc) break;                  // wave-uniform: nothing right of column dpc - 1
      const int owner = j + c < 64 ? j + c : 63;
      const double lc = readlane_f64(lr, owner);                  // L[j + c][j]
      a[c - 1] = __builtin_fma(-lr, lc, a[c]);   // element (r, j + c) of the trailing matrix (rows above the diagonal: never read)
    }
  }
  if (r < dpc) dst[dpc] = invd;
  if (r == 0) bad_out[b] = bad ? 1 : 0;
}

// Forward substitution for 64 rows of u at a time, one row per LANE: acc[0..d) = u_i - m_b in registers, and column by
// column (right-looking, the order of the factorisation above)  y_k = acc[k] / L_kk,  ss += y_k^2,  acc[r] -= L_rk y_k
// for r > k.  L_rk is the same for every lane: the column sits in registers 16 rows at a time (lane l holds row
// 16 c + (l mod 16), read from the LDS copy of the factor) and enters the update through the DPP operand of
// `v_fmac_f64` (row_newbcast: lane r mod 16 of every row of 16 lanes -- the operand broadcast the instruction set has
// for binary64 matrix products): one vector instruction per multiply-add, no scalar or cross-lane traffic in the chain,
// every lane busy.  Per row the roundings are those of the earlier kernel (one WAVE per row, lane = r, y_k handed round
// by v_readlane: a dependent chain of 64 broadcasts per row, 0.138 ms for 30 rounds at N = 4000, d = 50): the results are
// bit-identical to it.  All rows go through the substitution; selected ones are left out of the maximum.
constexpr int kSolveWaves = 4;

template <int DPC>   // d <= DPC, DPC a multiple of 8
__global__ __launch_bounds__(64 * kSolveWaves) void k_boot_solvemax(const double *__restrict__ u, int n, int d,
                                                                    const uint8_t *__restrict__ selected,
                                                                    const double *__restrict__ mean, const double *__restrict__ Lin,
                                                                    int lrows, const int *__restrict__ bad_in,
                                                                    unsigned long long *__restrict__ out_bits) {
  constexpr int NCH = (DPC + 15) / 16;
  __shared__ double Ls[DPC][DPC + 1];   // column DPC: 1 / L_rr
  __shared__ double ms[DPC];
  const int b = blockIdx.y, lane = threadIdx.x & 63, sub = lane & 15;
  const double *src = Lin + (size_t)b * lrows * (lrows + 1);
  for (int e = threadIdx.x; e < DPC * (DPC + 1); e += 64 * kSolveWaves) {
    const int rr = e / (DPC + 1), c = e - rr * (DPC + 1);
    Ls[rr][c] = src[(size_t)rr * (lrows + 1) + (c < DPC ? c : lrows)];
  }
  if (threadIdx.x < DPC) ms[threadIdx.x] = threadIdx.x < d ? mean[(size_t)b * d + threadIdx.x] : 0.0;
  __syncthreads();
  const int i = blockIdx.x * (64 * kSolveWaves) + threadIdx.x;
  const int ic = i < n ? i : n - 1;
  double acc[DPC];
#pragma unroll
  for (int r = 0; r < DPC; ++r) acc[r] = r < d ? u[(size_t)ic * d + r] - ms[r] : 0.0;
  double inv16[NCH];   // 1 / L_rr, 16 rows per register
#pragma unroll
  for (int c = 0; c < NCH; ++c) inv16[c] = Ls[16 * c + sub < DPC ? 16 * c + sub : 0][DPC];
  double ss = 0.0;
  static_for<0, DPC>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if (k < d) {   // wave-uniform
      const double yk = acc[k] * row_bcast<(k & 15)>(inv16[k >> 4]);
      ss = __builtin_fma(yk, yk, ss);
      const double ny = -yk;
      static_for<(k + 1) / 16, NCH>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        const double col = Ls[16 * c + sub < DPC ? 16 * c + sub : 0][k];   // L[16 c + sub][k]
        static_for<(16 * c > k + 1 ? 16 * c : k + 1), (16 * c + 16 < DPC ? 16 * c + 16 : DPC)>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value;
          fmac_row_bcast<(r & 15)>(acc[r], col, ny);
        });
      });
    }
  });
  const bool counts = i < n && selected[(size_t)b * n + ic] == 0;
  const bool nan_seen = __ballot(counts && ss != ss) != 0ull;
  double fbest = counts ? fmax(0.0, ss) : 0.0;
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) fbest = fmax(fbest, __shfl_xor(fbest, w));
  if (lane == 0) {
    // non-negative doubles order like their bit patterns; NaN / failed factorisation -> all ones
    atomicMax(&out_bits[b], (bad_in[b] || nan_seen) ? ~0ull : (unsigned long long)__double_as_longlong(fbest));
  }
}

size_t boot_cholmax_scratch_bytes(int d, int B) {
  const int dpc = (d + 7) / 8 * 8;
  return (size_t)B * dpc * (dpc + 1) * sizeof(double) + (size_t)B * sizeof(int);
}

// out_bits: B words, zeroed by the caller; afterwards the bit pattern of f_b, or all ones for a failed round.
// scratch: boot_cholmax_scratch_bytes(d, B)
hipError_t launch_boot_cholmax(const double *u, int n, int d, const uint8_t *selected, int B, const double *mean,
                               const double *cov, double scale, unsigned long long *out_bits, void *scratch,
                               hipStream_t s) {
  if (n <= 0 || B <= 0) return hipSuccess;
  if (d > 64) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((n + 64 * kSolveWaves - 1) / (64 * kSolveWaves)), (unsigned)B);
  const int dpc = (d + 7) / 8 * 8;   // size class of both kernels: multiples of 8 (padding rows / columns are identity)
  double *Ls = static_cast<double *>(scratch);
  int *bad = reinterpret_cast<int *>(Ls + (size_t)B * dpc * (dpc + 1));
  if (dpc <= 32)
    hipLaunchKernelGGL(k_boot_chol<32>, dim3((unsigned)B), dim3(64), 0, s, cov, d, dpc, scale, Ls, bad);
  else
    hipLaunchKernelGGL(k_boot_chol<64>, dim3((unsigned)B), dim3(64), 0, s, cov, d, dpc, scale, Ls, bad);
  switch (dpc) {
#define MLF_CHOLMAX(DPC)                                                                                                          \
  case DPC:                                                                                                                       \
    hipLaunchKernelGGL(k_boot_solvemax<DPC>, grid, dim3(64 * kSolveWaves), 0, s, u, n, d, selected, mean, Ls, DPC, bad, out_bits); \
    break;
    MLF_CHOLMAX(8) MLF_CHOLMAX(16) MLF_CHOLMAX(24) MLF_CHOLMAX(32) MLF_CHOLMAX(40) MLF_CHOLMAX(48) MLF_CHOLMAX(56) MLF_CHOLMAX(64)
#undef MLF_CHOLMAX
  }
  return hipGetLastError();
}

// idx: scratch of B * n ints
void launch_boot_moments(const double *u, int n, int d, const uint8_t *selected, int B, double *mean,
                         int *count, double *cov, int *idx, hipStream_t s) {
  hipLaunchKernelGGL(k_boot_index, dim3(B), dim3(256), 0, s, selected, n, idx, count);
  if (d > 128) return launch_boot_mean_cov_wide(u, n, d, idx, count, B, mean, cov, s);
  if (d <= 64)
    hipLaunchKernelGGL(k_boot_mean<1>, dim3(B), dim3(1024), 0, s, u, d, idx, n, count, mean);
  else
    hipLaunchKernelGGL(k_boot_mean<2>, dim3(B), dim3(1024), 0, s, u, d, idx, n, count, mean);
  const dim3 cgrid((unsigned)((d + kCovRows - 1) / kCovRows), (unsigned)B);
  const dim3 bgrid((unsigned)((d + 15) / 16), (unsigned)B);   // one workgroup per 16-row block of the matrix and round
  if (d <= 16)
    hipLaunchKernelGGL(k_boot_cov16<1>, bgrid, dim3(64 * kCovWaves), 0, s, u, d, idx, n, mean, count, cov);
  else if (d <= 32)
    hipLaunchKernelGGL(k_boot_cov16<2>, bgrid, dim3(64 * kCovWaves), 0, s, u, d, idx, n, mean, count, cov);
  else if (d <= 48)
    hipLaunchKernelGGL(k_boot_cov16<3>, bgrid, dim3(64 * kCovWaves), 0, s, u, d, idx, n, mean, count, cov);
  else if (d <= 64)
    hipLaunchKernelGGL(k_boot_cov16<4>, bgrid, dim3(64 * kCovWaves), 0, s, u, d, idx, n, mean, count, cov);
  else
    hipLaunchKernelGGL(k_boot_cov<2>, cgrid, dim3(64 * kCovWaves), 0, s, u, d, idx, n, mean, count, cov);
}

// ------------------------------------------------------- likelihoods (V1, L1-L3) ---------
// (params, d, n, like) convention of reference languages/c/mylib.c:33; one lane per parameter
// vector.  The (n, d) batch is row-major, so lane-strided reads would touch 64 cache lines per
// load instruction (measured 0.6 TB/s); instead each wave copies its 64 rows -- one contiguous
// 64*d*8-byte block -- into LDS with fully coalesced loads (row stride d+1 doubles: conflict-free
// lane = row reads) and evaluates from there.  Tolerance class (1e-12 relative): numpy's pairwise
// sum / libm cos are not bit-reproduced.
__global__ __launch_bounds__(128) void k_loglike(int kind, const double *params, int d, long long n,
                                                 const double *aux, double sigma, double *like) {
  extern __shared__ __attribute__((aligned(16))) double rows[];   // [2 waves][64][d + 1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ds = d + 1;
  double *mine = rows + (size_t)wave * 64 * ds;
  const long long j0 = ((long long)blockIdx.x * 2 + wave) * 64;
  const long long left = n - j0;
  const int nrows = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
  const double *src = params + j0 * d;
  const int total = nrows * d;
  for (int e0 = 0; e0 < total; e0 += 64 * 8) {   // eight coalesced 512-byte loads in flight
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = e0 + 64 * i + lane;
      v[i] = e < total ? src[e] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = e0 + 64 * i + lane;
      if (e < total) mine[(e / d) * ds + (e % d)] = v[i];
    }
  }
  // wave-private staging: LDS operations of one wave execute in order, no barrier needed
  if (lane < nrows) like[j0 + lane] = loglike_row(kind, mine + lane * ds, d, aux, sigma);
}

// Even d <= 128: no staging at all.  A row is read by HW = 2 ... 64 lanes (the smallest power of two with 2 HW >= d), two
// consecutive coordinates (one 16-byte non-temporal load) per lane, so a wave-wide load covers 64 / HW whole rows =
// one contiguous piece of the batch; four such loads are in flight per lane before the first is consumed.  The terms
// of a row are combined across its lanes (sum or product: tolerance class 1e-12, the reference's pairwise numpy sum
// is not bit-reproduced either way); lane 0 of the row stores the result.  No LDS: 16 waves per SIMD can be resident
// (the staged kernel above holds 52 KB of LDS per 2 waves: 6 waves per CU, 2.8 TB/s at 10^6 x 50).
template <int KIND, int HW>
__global__ __launch_bounds__(256) void k_loglike_rows(const double *__restrict__ params, int d, long long n,
                                                      const double *__restrict__ aux, double sigma,
                                                      double *__restrict__ like) {
  constexpr int RPW = 64 / HW;   // rows per wave-wide load
  constexpr int U = 4;           // loads in flight per lane
  const int lane = threadIdx.x & 63;
  const int sub = lane & (HW - 1), half = lane / HW;
  const int k0 = 2 * sub;
  const bool active = k0 < d;
  const long long nwaves = (long long)gridDim.x * 4;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  double c0 = 0.0, c1 = 0.0;
  if (KIND == 0 && active) {
    c0 = aux[k0];
    c1 = aux[k0 + 1];
  }
  const double gconst = KIND == 0 ? -0.5 * log(2.0 * M_PI * sigma * sigma) * (double)d : 0.0;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  for (long long base = gw * RPW; base < n; base += nwaves * RPW * U) {
    dbl2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = base + (long long)u * nwaves * RPW + half;
      v[u] = (dbl2){0.0, 0.0};
      if (active && r < n) v[u] = __builtin_nontemporal_load(reinterpret_cast<const dbl2 *>(params + r * d) + sub);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = base + (long long)u * nwaves * RPW + half;
      const double x0 = v[u].x, x1 = v[u].y;
      double acc;
      if (KIND == 0) {          // docs/gauss.py:25-27
        const double z0 = (x0 - c0) / sigma, z1 = (x1 - c1) / sigma;
        acc = active ? z0 * z0 + z1 * z1 : 0.0;
      } else if (KIND == 1) {   // examples/testeggbox.py:9-11
        acc = active ? cos(x0 / 2.0) * cos(x1 / 2.0) : 1.0;
      } else if (KIND == 2) {   // examples/test_PopSliceSampler.py:69-71
        acc = active ? cos(x0) * cos(x1) : 1.0;
      } else {                  // examples/testrosenbrock.py:10-13: terms k = 2 sub and 2 sub + 1 (the latter needs x[k0 + 2])
        const double nx = __shfl_down(x0, 1, HW);
        const double t0 = x1 - x0 * x0, w0 = 1.0 - x0;
        const double t1 = nx - x1 * x1, w1 = 1.0 - x1;
        acc = (k0 + 1 < d) ? 100.0 * (t0 * t0) + w0 * w0 : 0.0;
        if (k0 + 2 < d) acc += 100.0 * (t1 * t1) + w1 * w1;
      }
#pragma unroll
      for (int o = HW / 2; o > 0; o >>= 1) {
        const double other = __shfl_xor(acc, o, HW);
        acc = (KIND == 1 || KIND == 2) ? acc * other : acc + other;
      }
      if (sub == 0 && r < n) {
        double out;
        if (KIND == 0) {
          out = -0.5 * acc + gconst;
        } else if (KIND == 1) {
          const double b1 = 2.0 + acc, b2 = b1 * b1;
          out = b2 * b2 * b1;
        } else if (KIND == 2) {
          out = acc * acc;
        } else {
          out = -2.0 * acc;
        }
        like[r] = out;
      }
    }
  }
}

template <int KIND>
static void launch_loglike_rows(const double *params, int d, long long n, const double *aux, double sigma, double *like,
                                hipStream_t s) {
  int hw = 2;   // lanes per row: the smallest power of two with 2 hw >= d
  while (2 * hw < d) hw *= 2;
  const int rpw = 64 / hw;
  long long waves = (n + rpw * 4 - 1) / (rpw * 4);          // four loads per lane
  if (waves > 256 * 4 * 8 * 4) waves = 256 * 4 * 8 * 4;      // grid-stride beyond 32 waves per SIMD
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  switch (hw) {
#define MLF_LL(H)                                                                                              \
  case H:                                                                                                      \
    hipLaunchKernelGGL((k_loglike_rows<KIND, H>), grid, block, 0, s, params, d, n, aux, sigma, like);          \
    break;
    MLF_LL(2) MLF_LL(4) MLF_LL(8) MLF_LL(16) MLF_LL(32) MLF_LL(64)
#undef MLF_LL
    default: break;
  }
}

// above 128 parameters: one thread per row straight from global memory (a row no longer fits the staging of k_loglike)
__global__ __launch_bounds__(256) void k_loglike_wide(int kind, const double *params, int d, long long n, const double *aux, double sigma,
                                                     double *like) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) like[i] = loglike_row(kind, params + i * d, d, aux, sigma);
}

void launch_loglike(int kind, const double *params, int d, long long n, const double *aux,
                    double sigma, double *like, hipStream_t s) {
  if (n <= 0) return;
  if (d > 128) {
    hipLaunchKernelGGL(k_loglike_wide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kind, params, d, n, aux, sigma, like);
    return;
  }
  if (d % 2 == 0 && d <= 128 && (reinterpret_cast<uintptr_t>(params) & 15u) == 0) {   // rows are 16-byte aligned
    switch (kind) {
      case 0: launch_loglike_rows<0>(params, d, n, aux, sigma, like, s); return;
      case 1: launch_loglike_rows<1>(params, d, n, aux, sigma, like, s); return;
      case 2: launch_loglike_rows<2>(params, d, n, aux, sigma, like, s); return;
      default: launch_loglike_rows<3>(params, d, n, aux, sigma, like, s); return;
    }
  }
  const size_t lds = (size_t)2 * 64 * (d + 1) * sizeof(double);
  static DeviceGrant grant;
  (void)grant.ensure([] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_loglike), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL(k_loglike, dim3((unsigned)((n + 127) / 128)), dim3(128), lds, s, kind, params, d, n,
                     aux, sigma, like);
}

// idx[p] = -2 where the ellipsoid gate rejected proposal p (scan wrote -1 there)
__global__ void k_mark_gated(const uint8_t *gate, long long n, long long *idx) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n && !gate[e]) idx[e] = -2;
}

void launch_mark_gated(const uint8_t *gate, long long n, long long *idx, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_mark_gated, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gate, n, idx);
}

// ------------------------------------------------------- FP64 VALU rate probe ------------
// 8 independent add/mul chains per lane: measures the issue rate of non-fused v_add_f64 /
// v_mul_f64, the ceiling of the distance kernels (which may not use FMA).
__global__ __launch_bounds__(256) void k_fp64_probe(double *sink, int iters) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  double x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const double a = 1.0000001, b = 1e-9;
  for (int i = 0; i < iters; ++i) {
    x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b;
    x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b;
    x0 = x0 + b; x1 = x1 * a; x2 = x2 + b; x3 = x3 * a;
    x4 = x4 + b; x5 = x5 * a; x6 = x6 + b; x7 = x7 * a;
  }
  const double r = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
  if (r == 123.456) sink[blockIdx.x * 256 + threadIdx.x] = r;
}

double launch_fp64_probe(double *sink, int blocks, int iters, hipStream_t s) {
  hipLaunchKernelGGL(k_fp64_probe, dim3(blocks), dim3(256), 0, s, sink, iters);
  return (double)blocks * 256.0 * (double)iters * 16.0;
}

}  // namespace mlf
