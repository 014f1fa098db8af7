// mlf_wide.hip -- the kernels of the path for dimensionalities above 128 (round 5; up to MLF_MAX_DIM = 1024).
//
// The reference has no limit on ndim: every loop of mlfriends.pyx runs `for k in range(ndim)` (:63-66, :178-180, :217-219)
// and its own benchmark grid goes to d = 256 (tests/benchmark_maxradius.py:36-98).  The kernels of the other files keep a
// point in registers (DP is a template parameter, 2 DP registers per lane): fine up to 128 coordinates, impossible beyond.
// These are their k-CHUNKED forms with the dimensionality at run time: the same arithmetic, term by term, in the same
// order -- k ascending ACROSS the chunks, every sub / mul / add rounded on its own (this file is compiled with
// -ffp-contract=off), explicit fma() where the narrow kernels use one -- so every result is bit-identical to what the
// templated kernels would give (and to the CPU restatement the tests compare with).  They are the slow forms by construction (a point's coordinates come
// from L2 / LDS every time they are used); nothing below 129 dimensions is routed here.
//
//   k_scan_wide        K1 / K2 / K3 pass 1 / the scan of R3 (find_nearby :143-183, count_nearby :31-68): lane = live point,
//                      16 queries of the workgroup in LDS, 16 coordinates of the live tile in registers at a time
//   k_boot_wide        K4 (compute_maxradiussq :188-224) for up to 32 bootstrap rounds: lane = row j, four live points i
//                      at a time, masked minima, the same M as k_boot
//   k_prep_wide        H3 + T1 (_inside_ellipsoid :882-912 in numpy's einsum order, AffineLayer.transform :737-743 as a
//                      k-ascending fma chain, wraps :529-536): one thread per proposal, its centred row in LDS
//   k_quadmax_wide     the bootstrap's ellipsoid factor (:1060-1062), rows x rounds
//   k_subtract_wide    K3 pass 2 (_subtract_nearby :100-109): one wave per point, neighbours in ascending order
//   k_boot_mean_wide / k_boot_cov_wide   moments of the selected rows (bounding_ellipsoid :426-476; tolerance class)
#include "mlf_common.hpp"

#include <math.h>

namespace mlf {

namespace {

constexpr int kWideQB = 16;   // queries per scan workgroup
constexpr int kWideKC = 16;   // coordinates of a live tile held in registers at a time

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanThreads) void k_scan_wide(ScanArgs a, int dpw) {
  extern __shared__ __attribute__((aligned(16))) double qs[];   // [kWideQB][dpw]
  __shared__ int state[kWideQB];   // SCAN_FIRST / MASK: first-hit index or kNone; -1 = inactive
  __shared__ int cnt[kWideQB];
  __shared__ int any_active;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long q0 = (long long)blockIdx.x * kWideQB;
  const long long left = a.nq - q0;
  const int nqb = left < kWideQB ? (int)left : kWideQB;
  auto gated = [&](long long j) { return a.gate == nullptr || a.gate[j] != 0; };
  if (a.only_gated) {
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < nqb && gated(q0 + tid)) any_active = 1;
    __syncthreads();
    if (!any_active) return;
  }
  for (int e = tid; e < kWideQB * dpw; e += kScanThreads) {
    const int qq = e / dpw, k = e - qq * dpw;
    double v = 0.0;
    if (qq < nqb && k < a.d) v = a.q[(q0 + qq) * a.ldq + (long long)k * (a.ldk > 1 ? a.ldk : 1)];
    qs[e] = v;
  }
  if (tid < kWideQB) {
    const bool active = tid < nqb && gated(q0 + tid);
    state[tid] = active ? kNone : -1;
    cnt[tid] = 0;
  }
  __syncthreads();
  const int mode = a.mode;
  for (int t = wave; t < a.ntiles; t += kScanThreads / kWave) {
    const int base = t * kWave;
    // a tile none of the workgroup's queries still needs is skipped (wave-uniform)
    bool need = mode == SCAN_COUNT || mode == SCAN_FLAGS;
    if (!need)
      for (int qq = 0; qq < nqb; ++qq) {
        const int st = *(volatile int *)&state[qq];
        need |= st >= 0 && (mode == SCAN_FIRST ? st >= base : st == kNone);
      }
    if (!need) continue;
    double acc[kWideQB];
#pragma unroll
    for (int qq = 0; qq < kWideQB; ++qq) acc[qq] = 0.0;
    for (int c0 = 0; c0 < dpw; c0 += kWideKC) {   // k ascending across the chunks: each sum continues where it stopped
      double r[kWideKC];
#pragma unroll
      for (int k = 0; k < kWideKC; ++k) r[k] = a.refT[(size_t)(c0 + k) * a.npad + base + lane];   // rows past d are zero
#pragma unroll
      for (int qq = 0; qq < kWideQB; ++qq) {
        const double2 *qrow = reinterpret_cast<const double2 *>(qs + qq * dpw + c0);
#pragma unroll
        for (int k = 0; k < kWideKC; k += 2) {
          const double2 v = qrow[k >> 1];
          const double d0 = r[k] - v.x;
          acc[qq] += d0 * d0;
          const double d1 = r[k + 1] - v.y;
          acc[qq] += d1 * d1;
        }
      }
    }
    const bool valid = base + lane < a.n;
#pragma unroll
    for (int qq = 0; qq < kWideQB; ++qq) {
      const int st = qq < nqb ? __builtin_amdgcn_readfirstlane(*(volatile int *)&state[qq]) : -1;
      const bool skip = st < 0 || (mode == SCAN_FIRST && st < base) || (mode == SCAN_MASK && st != kNone);
      const unsigned long long m = __ballot(valid && acc[qq] <= a.r2);
      if (!skip) {   // wave-uniform
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
}

// ---------------------------------------------------------------------------------------------------------------------
// lane = row j of the pair matrix; the live points i of the chunk pass four at a time (one read of b_j[k] feeds four sums);
// dist2(i, j) = sum_k (a_i[k] - b_j[k])^2 as in k_boot; round r keeps the smallest distance to a SELECTED i; rows j that are
// themselves selected in r do not report (reference :209-221)
constexpr int kWideIB = 4;
__global__ __launch_bounds__(kWave) void k_boot_wide(BootArgs a, int d, int dp) {
  const int lane = threadIdx.x;
  const int j = (blockIdx.x + a.blk0) * kWave + lane;   // < npad by construction of the grid
  const int i_begin = blockIdx.y * a.chunk;
  int i_end = i_begin + a.chunk;
  if (i_end > a.n) i_end = a.n;
  if (i_begin >= i_end) return;
  double mind[kBootGroup];
#pragma unroll
  for (int r = 0; r < kBootGroup; ++r) mind[r] = 1e300;   // reference :215
  for (int i0 = i_begin; i0 < i_end; i0 += kWideIB) {
    const double *x[kWideIB];
    unsigned seli[kWideIB];
#pragma unroll
    for (int q = 0; q < kWideIB; ++q) {
      const int i = i0 + q < i_end ? i0 + q : i_end - 1;
      x[q] = a.refR + (size_t)i * dp;                    // wave-uniform: scalar loads
      seli[q] = i0 + q < i_end ? a.sel[i] : 0u;          // a repeated last point selects nothing
    }
    double acc[kWideIB];
#pragma unroll
    for (int q = 0; q < kWideIB; ++q) acc[q] = 0.0;
    for (int k = 0; k < d; ++k) {
      const double b = a.refT[(size_t)k * a.npad + j];
#pragma unroll
      for (int q = 0; q < kWideIB; ++q) {
        const double diff = x[q][k] - b;
        acc[q] += diff * diff;
      }
    }
#pragma unroll
    for (int q = 0; q < kWideIB; ++q) {
      const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)seli[q]);
#pragma unroll
      for (int r = 0; r < kBootGroup; ++r)
        if ((s >> r) & 1u) mind[r] = acc[q] < mind[r] ? acc[q] : mind[r];
    }
  }
  if (j < a.n) {
    const unsigned selj = a.sel[j];
#pragma unroll
    for (int r = 0; r < kBootGroup; ++r)
      if (((selj >> r) & 1u) == 0u) atomicMin(&a.M[(size_t)r * a.npad + j], (unsigned long long)__double_as_longlong(mind[r]));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// one thread per proposal; its row sits in LDS, coordinate-major ([k][threads]: conflict-free).  The matrix elements are
// wave-uniform (scalar loads).  H3: ONE accumulator, j outer / k inner, (delta_j A_jk) delta_k, no fma -- numpy's c_einsum order
// (k_prep); T1: t_c = sum_k delta_k T[k][c], k ascending, one fma per term (k_prep's chain)
__global__ void k_prep_wide(PrepArgs a, int dp) {
  extern __shared__ __attribute__((aligned(16))) double dl[];   // [d][PB]
  const int PB = blockDim.x, tid = threadIdx.x, d = a.d;
  const long long p = (long long)blockIdx.x * PB + tid;
  const bool live = p < a.np;
  const double *row = a.pts + (live ? p : 0) * (long long)d;
  bool inside = live;
  if (a.do_ell) {
    for (int k = 0; k < d; ++k) dl[k * PB + tid] = row[k] - a.ell_ctr[k];
    double acc = 0.0;
    for (int j = 0; j < d; ++j) {
      const double dj = dl[j * PB + tid];
      const double *arow = a.ell_A + (size_t)j * dp;
      for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * dl[k * PB + tid];
    }
    inside = live && (acc <= a.enlarge);
    if (live) {
      a.mask[p] = inside ? 1 : 0;
      if (a.q_out) a.q_out[p] = acc;
    }
  }
  if (a.do_tr) {
    if (!__any(inside)) return;
    for (int k = 0; k < d; ++k) {
      double w = row[k];
      if (a.wrap_shift) {
        const double sh = a.wrap_shift[k];
        if (sh == sh) w = fmod(w + sh, 1.0);   // NaN marks an unwrapped dimension
      }
      dl[k * PB + tid] = w - a.lay_ctr[k];
    }
    for (int c = 0; c < d; ++c) {
      const double *trow = a.lay_Tt + (size_t)c * dp;   // row c holds T[:, c]
      double acc = 0.0;
      for (int k = 0; k < d; ++k) acc = __builtin_fma(dl[k * PB + tid], trow[k], acc);
      if (inside) a.t_out[p * a.ldt + c] = acc;
    }
  }
}

// rows x rounds: workgroup = (RB rows, round b); q in the einsum order of k_prep; rows selected in round b do not take part;
// per-workgroup maxima (NaN propagates) -> part[b][blockIdx.x]
__global__ void k_quadmax_wide(QuadMaxArgs a, int dp) {
  extern __shared__ __attribute__((aligned(16))) double dl[];   // [d][RB], then [RB] for the reduction
  const int RB = blockDim.x, tid = threadIdx.x, b = blockIdx.y, d = a.d;
  const int i = blockIdx.x * RB + tid;
  const double *A = a.invcov + (size_t)b * d * dp;
  const double *ctr = a.ctr + (size_t)b * dp;
  const bool use = i < a.n && !a.selected[(size_t)b * a.n + i];
  const double *row = a.u + (size_t)(i < a.n ? i : 0) * d;
  double q = -INFINITY;
  for (int k = 0; k < d; ++k) dl[k * RB + tid] = row[k] - ctr[k];
  if (__any(use)) {
    double acc = 0.0;
    for (int j = 0; j < d; ++j) {
      const double dj = dl[j * RB + tid];
      const double *arow = A + (size_t)j * dp;
      for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * dl[k * RB + tid];
    }
    if (use) q = acc;
  }
  __syncthreads();
  dl[tid] = q;
  __syncthreads();
  if (tid == 0) {
    double m = dl[0];
    for (int t = 1; t < RB; ++t) {
      const double o = dl[t];
      m = (o > m || o != o) ? o : m;
    }
    a.part[(size_t)b * gridDim.x + blockIdx.x] = m;
  }
}

// one wave per point j; lane l owns the coordinates l, l + 64, ... (H of them); the neighbours i of j (hit ballots of pass 1)
// are added in ascending i (reference :100-109), then pts[j] - sum / nn
template <int H>
__global__ __launch_bounds__(kWave) void k_subtract_wide(const double *__restrict__ pts, int n, int d, const unsigned long long *__restrict__ flags,
                                                        int ntiles, double *__restrict__ out) {
  const int lane = threadIdx.x, j = blockIdx.x;
  double sum[H];
#pragma unroll
  for (int h = 0; h < H; ++h) sum[h] = 0.0;
  long long nn = 0;
  for (int t = 0; t < ntiles; ++t) {
    unsigned long long m = flags[(long long)j * ntiles + t];
    while (m) {
      const int q = __ffsll((long long)m) - 1;
      m &= m - 1;
      const double *r = pts + (long long)(t * kWave + q) * d;
#pragma unroll
      for (int h = 0; h < H; ++h)
        if (lane + 64 * h < d) sum[h] += r[lane + 64 * h];
      ++nn;
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
    if (lane + 64 * h < d) out[(long long)j * d + lane + 64 * h] = pts[(long long)j * d + lane + 64 * h] - sum[h] / (double)nn;
}

// mean of the selected rows of round b (index list of k_boot_index, ascending): thread = coordinate
__global__ __launch_bounds__(256) void k_boot_mean_wide(const double *__restrict__ u, int d, const int *__restrict__ idx, int n,
                                                       const int *__restrict__ count, double *__restrict__ mean) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const int cnt = count[b];
  const int *list = idx + (size_t)b * n;
  double s = 0.0;
  for (int e = 0; e < cnt; ++e) s += u[(size_t)list[e] * d + c];
  mean[(size_t)b * d + c] = cnt > 0 ? s / (double)cnt : 0.0;
}

// cov[a][c] = sum over the selected rows of (x_a - m_a)(x_c - m_c) / (count - 1)   (np.cov of the centred rows, tolerance class);
// workgroup = (row a of the matrix, round b), thread = column c (strided)
__global__ __launch_bounds__(256) void k_boot_cov_wide(const double *__restrict__ u, int d, const int *__restrict__ idx, int n,
                                                      const double *__restrict__ mean, const int *__restrict__ count, double *__restrict__ cov) {
  const int b = blockIdx.y, ar = blockIdx.x;
  const int cnt = count[b];
  const int *list = idx + (size_t)b * n;
  const double *m = mean + (size_t)b * d;
  const double ma = m[ar];
  for (int c = threadIdx.x; c < d; c += 256) {
    const double mc = m[c];
    double s = 0.0;
    for (int e = 0; e < cnt; ++e) {
      const double *r = u + (size_t)list[e] * d;
      s = __builtin_fma(r[ar] - ma, r[c] - mc, s);
    }
    cov[((size_t)b * d + ar) * d + c] = cnt > 1 ? s / (double)(cnt - 1) : NAN;
  }
}

// `count` rows (count x d, packed) replace the live points index[0 .. count) in both layouts
__global__ void k_update_row_wide(const double *rows, int d, int dp, int npad, const long long *index, double *refT, double *refR) {
  const long long i = index[blockIdx.x];
  for (int k = threadIdx.x; k < dp; k += blockDim.x) {
    const double v = k < d ? rows[(long long)blockIdx.x * d + k] : 0.0;
    refR[i * dp + k] = v;
    refT[(long long)k * npad + i] = v;
  }
}

// rows of LDS a one-thread-per-row kernel can hold: the largest of 64 / 32 / 16 / 8 rows that fits 144 KB
int rows_per_block(int d) {
  for (int rb = 64; rb >= 8; rb >>= 1)
    if ((size_t)rb * d * sizeof(double) <= 144 * 1024) return rb;
  return 0;
}

// dynamic LDS above the default 64 KB has to be granted per kernel; the grant leaves room for the kernel's static arrays
// (the attribute belongs to (function, device): one DeviceGrant per kernel, asked once per device -- ADVICE r5)
hipError_t allow_lds(DeviceGrant &grant, const void *fn, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  return grant.ensure([fn] { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); });
}

}  // namespace

bool wide_dims(int dp) { return dp > 128; }

hipError_t launch_scan_wide(int dp, const ScanArgs &a, hipStream_t s) {
  if (a.nq <= 0) return hipSuccess;
  // the second-stage uses behind the pre-filter (routing bytes, in-place whitening, finalise tail) do not exist up here:
  // above 128 dimensions nothing is filtered
  if (a.fin_best || a.route || a.any_flag || a.raw_ctr) return hipErrorInvalidValue;
  const size_t lds = (size_t)kWideQB * dp * sizeof(double);
  static DeviceGrant grant_k_scan_wide;
  if (hipError_t e = allow_lds(grant_k_scan_wide, reinterpret_cast<const void *>(&k_scan_wide), lds)) return e;
  const unsigned grid = (unsigned)((a.nq + kWideQB - 1) / kWideQB);
  hipLaunchKernelGGL(k_scan_wide, dim3(grid), dim3(kScanThreads), lds, s, a, dp);
  return hipGetLastError();
}

hipError_t launch_boot_wide(int dp, const BootArgs &a, int d, int nchunks, hipStream_t s, int nblocks) {
  const dim3 grid((unsigned)(nblocks >= 0 ? nblocks : a.npad / kWave), (unsigned)nchunks);
  if (grid.x == 0) return hipSuccess;
  hipLaunchKernelGGL(k_boot_wide, grid, dim3(kWave), 0, s, a, d, dp);
  return hipGetLastError();
}

hipError_t launch_prep_wide(int dp, const PrepArgs &a, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  const int pb = rows_per_block(a.d);
  if (pb == 0) return hipErrorInvalidValue;
  const size_t lds = (size_t)pb * a.d * sizeof(double);
  static DeviceGrant grant_k_prep_wide;
  if (hipError_t e = allow_lds(grant_k_prep_wide, reinterpret_cast<const void *>(&k_prep_wide), lds)) return e;
  hipLaunchKernelGGL(k_prep_wide, dim3((unsigned)((a.np + pb - 1) / pb)), dim3(pb), lds, s, a, dp);
  return hipGetLastError();
}

int quadmax_blocks_wide(int n, int d) {
  const int rb = rows_per_block(d);
  return rb ? (n + rb - 1) / rb : 0;
}

hipError_t launch_quadmax_wide(int dp, const QuadMaxArgs &a, int B, hipStream_t s) {
  if (a.n <= 0 || B <= 0) return hipSuccess;
  const int rb = rows_per_block(a.d);
  if (rb == 0) return hipErrorInvalidValue;
  const size_t lds = (size_t)rb * a.d * sizeof(double);
  static DeviceGrant grant_k_quadmax_wide;
  if (hipError_t e = allow_lds(grant_k_quadmax_wide, reinterpret_cast<const void *>(&k_quadmax_wide), lds)) return e;
  hipLaunchKernelGGL(k_quadmax_wide, dim3((unsigned)((a.n + rb - 1) / rb), (unsigned)B), dim3(rb), lds, s, a, dp);
  return hipGetLastError();
}

void launch_subtract_wide(const double *pts, int n, int d, const unsigned long long *flags, int ntiles, double *out, hipStream_t s) {
  if (d <= 256)
    hipLaunchKernelGGL(k_subtract_wide<4>, dim3((unsigned)n), dim3(kWave), 0, s, pts, n, d, flags, ntiles, out);
  else if (d <= 512)
    hipLaunchKernelGGL(k_subtract_wide<8>, dim3((unsigned)n), dim3(kWave), 0, s, pts, n, d, flags, ntiles, out);
  else
    hipLaunchKernelGGL(k_subtract_wide<16>, dim3((unsigned)n), dim3(kWave), 0, s, pts, n, d, flags, ntiles, out);
}

void launch_boot_mean_cov_wide(const double *u, int n, int d, const int *idx, const int *count, int B, double *mean, double *cov, hipStream_t s) {
  hipLaunchKernelGGL(k_boot_mean_wide, dim3((unsigned)((d + 255) / 256), (unsigned)B), dim3(256), 0, s, u, d, idx, n, count, mean);
  hipLaunchKernelGGL(k_boot_cov_wide, dim3((unsigned)d, (unsigned)B), dim3(256), 0, s, u, d, idx, n, mean, count, cov);
}

void launch_update_rows_wide(const double *rows, int count, int d, int dp, int npad, const long long *index, double *refT, double *refR, hipStream_t s) {
  hipLaunchKernelGGL(k_update_row_wide, dim3((unsigned)count), dim3(256), 0, s, rows, d, dp, npad, index, refT, refR);
}

}  // namespace mlf
