// mlf_prep.hip -- per-proposal stages in front of the neighbour scan:
//   H3  _inside_ellipsoid  (reference mlfriends.pyx:882-912)
//   T1  AffineLayer.transform (:737-743) incl. the wrap of circular dims (:529-536)
//
// Mapping: one LANE owns one proposal; its centred coordinates live in registers (DP doubles),
// the d x d matrix (inverse covariance, then T^T) is staged in LDS and broadcast.
//
// H3 is evaluated in exactly the order numpy's three-operand c_einsum uses for
// 'ij,jk,ik->i' (probed bit-for-bit against numpy 2.2.6): ONE accumulator per point,
// j outer / k inner, term = (delta_j * A_jk) * delta_k, no FMA (-ffp-contract=off).
// T1 is a k-ascending fused-multiply-add chain (explicit fma(): BLAS order is implementation
// defined, this is what OpenBLAS produced for most probed shapes; tolerance class).
#include "mlf_common.hpp"
#include "mlf_dpp_dev.hpp"

namespace mlf {

template <int DP>
__global__ __launch_bounds__(256) void k_prep(PrepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mat[];  // [d][DP]

  const int tid = threadIdx.x;
  const long long p = (long long)blockIdx.x * 256 + tid;
  const bool live = p < a.np;
  const double *row = a.pts + (live ? p : 0) * (long long)a.d;
  const int d = a.d;
  bool inside = live;

  if (a.do_ell) {
    for (int e = tid; e < d * DP; e += 256) mat[e] = a.ell_A[e];
    double dl[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) dl[k] = (k < d) ? row[k] - a.ell_ctr[k] : 0.0;
    __syncthreads();
    double acc = 0.0;
    for (int j = 0; j < d; ++j) {
      const double dj = row[j] - a.ell_ctr[j];  // same subtraction as dl[j], bit-identical
      const double2 *arow = reinterpret_cast<const double2 *>(mat + j * DP);
#pragma unroll
      for (int k = 0; k < DP; k += 2) {
        const double2 v = arow[k >> 1];
        acc += (dj * v.x) * dl[k];
        acc += (dj * v.y) * dl[k + 1];
      }
    }
    inside = live && (acc <= a.enlarge);
    if (live) {
      a.mask[p] = inside ? 1 : 0;
      if (a.q_out) a.q_out[p] = acc;
    }
    __syncthreads();  // mat is reused below
  }

  if (a.do_tr) {
    for (int e = tid; e < d * DP; e += 256) mat[e] = a.lay_Tt[e];
    __syncthreads();
    if (__any(inside)) {  // a wave with no surviving proposal skips the d^2 work
      double dl[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        double w = 0.0;
        if (k < d) {
          w = row[k];
          if (a.wrap_shift) {
            const double sh = a.wrap_shift[k];
            if (sh == sh) w = fmod(w + sh, 1.0);  // NaN marks an unwrapped dimension
          }
          w -= a.lay_ctr[k];
        }
        dl[k] = w;
      }
      for (int c = 0; c < d; ++c) {
        const double2 *trow = reinterpret_cast<const double2 *>(mat + c * DP);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DP; k += 2) {
          const double2 v = trow[k >> 1];
          acc = __builtin_fma(dl[k], v.x, acc);
          acc = __builtin_fma(dl[k + 1], v.y, acc);
        }
        if (inside) a.t_out[p * a.ldt + c] = acc;
      }
    }
  }
}

// Bootstrap enlargement (reference mlfriends.pyx:1060-1062) for ALL rounds in one launch:
// workgroup = (256 rows, round b); A_b in LDS; q in numpy-einsum order like k_prep; rows selected
// in round b do not take part; per-workgroup maxima go to part[b][blockIdx.x] (NaN propagates).
template <int DP>
__global__ __launch_bounds__(256) void k_boot_quadmax(QuadMaxArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mat[];
  __shared__ double red[256];
  const int tid = threadIdx.x, b = blockIdx.y, d = a.d;
  const int i = blockIdx.x * 256 + tid;
  const double *A = a.invcov + (size_t)b * d * DP;
  const double *ctr = a.ctr + (size_t)b * DP;
  for (int e = tid; e < d * DP; e += 256) mat[e] = A[e];
  __syncthreads();
  const bool use = i < a.n && !a.selected[(size_t)b * a.n + i];
  double q = -INFINITY;
  if (__any(use)) {
    const double *row = a.u + (size_t)(i < a.n ? i : 0) * d;
    double dl[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double v = row[k < d ? k : d - 1] - ctr[k];
      dl[k] = (k < d) ? v : 0.0;
    }
    double acc = 0.0;
    for (int j = 0; j < d; ++j) {
      const double dj = row[j] - ctr[j];
      const double2 *arow = reinterpret_cast<const double2 *>(mat + j * DP);
#pragma unroll
      for (int k = 0; k < DP; k += 2) {
        const double2 v = arow[k >> 1];
        acc += (dj * v.x) * dl[k];
        acc += (dj * v.y) * dl[k + 1];
      }
    }
    if (use) q = acc;
  }
  red[tid] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      const double o = red[tid + w], m = red[tid];
      red[tid] = (o > m || o != o) ? o : m;
    }
    __syncthreads();
  }
  if (tid == 0) a.part[(size_t)b * gridDim.x + blockIdx.x] = red[0];
}

hipError_t launch_boot_quadmax(int dp, const QuadMaxArgs &a, int B, hipStream_t s) {
  if (a.n <= 0 || B <= 0) return hipSuccess;
  if (wide_dims(dp)) return launch_quadmax_wide(dp, a, B, s);
  const dim3 grid((unsigned)((a.n + 255) / 256), (unsigned)B);
  const size_t lds = (size_t)a.d * dp * sizeof(double);
  switch (dp) {
#define X(D)                                                                                   \
  case D:                                                                                      \
    if (lds > 48 * 1024) {                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_boot_quadmax<D>),   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                           \
    }                                                                                          \
    hipLaunchKernelGGL(k_boot_quadmax<D>, grid, dim3(256), lds, s, a);                         \
    break;
    MLF_FOR_EACH_DP(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// T1 for a few rows (the live points of a region: 4000 rows are 63 waves of k_prep, each lane a chain of d x d dependent
// multiply-adds: 0.036 ms, latency bound).  Here a wave takes 8 rows and lane c computes their OUTPUT coordinate c:
// t_c = sum_k x_k T[k][c], k ascending, one fused multiply-add per term -- the chain of k_prep, bit for bit (the padded
// terms k >= d, zeros times zeros, are kept: they turn an accumulated -0 into +0 there, so they do here).  T[k][.] is one
// coalesced load per k for all 8 rows; x_k is the same for every lane: the centred row sits 16 coordinates per register
// and enters through the DPP operand of v_fmac_f64 (mlf_dpp_dev.hpp).
constexpr int kWhitenRows = 8;
template <int NCH>   // dp <= 16 NCH <= 64
__global__ __launch_bounds__(256) void k_whiten_rows(const double *__restrict__ pts, long long n, int d, int dp,
                                                     const double *__restrict__ lay_ctr, const double *__restrict__ T8, int ldt8,
                                                     const double *__restrict__ wrap_shift, double *__restrict__ t_out, long long ldt) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kWhitenRows;
  if (r0 >= n) return;
  double x[kWhitenRows][NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int k = 16 * ch + sub;
    const double ctr = k < d ? lay_ctr[k] : 0.0;
    double sh = NAN;
    if (wrap_shift && k < d) sh = wrap_shift[k];
#pragma unroll
    for (int i = 0; i < kWhitenRows; ++i) {
      const long long row = r0 + i < n ? r0 + i : n - 1;
      double w = 0.0;
      if (k < d) {
        w = pts[row * d + k];
        if (sh == sh) w = fmod(w + sh, 1.0);   // NaN marks an unwrapped dimension
        w -= ctr;
      }
      x[i][ch] = w;
    }
  }
#pragma unroll
  for (int i = 0; i < kWhitenRows; ++i)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) dpp_settle(x[i][ch]);
  double acc[kWhitenRows];
#pragma unroll
  for (int i = 0; i < kWhitenRows; ++i) acc[i] = 0.0;
  const int c = lane < d ? lane : 0;
  static_for<0, 16 * NCH>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if (k < dp) {   // wave-uniform
      const double tv = T8[(long long)k * ldt8 + c];
#pragma unroll
      for (int i = 0; i < kWhitenRows; ++i) fmac_row_bcast<(k & 15)>(acc[i], x[i][k >> 4], tv);
    }
  });
  if (lane < d) {
#pragma unroll
    for (int i = 0; i < kWhitenRows; ++i)
      if (r0 + i < n) t_out[(r0 + i) * ldt + lane] = acc[i];
  }
}

// rows -> whitened rows with an affine layer; T8: T row-major (row k = input coordinate), stride ldt8, zero padded to dp rows
hipError_t launch_whiten_rows(const double *pts, long long n, int d, int dp, const double *lay_ctr, const double *T8, int ldt8,
                              const double *wrap_shift, double *t_out, long long ldt, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (dp > 64) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)((n + 4 * kWhitenRows - 1) / (4 * kWhitenRows));
  const int nch = (dp + 15) / 16;
#define X(NCH)                                                                                                          \
  case NCH:                                                                                                             \
    hipLaunchKernelGGL(k_whiten_rows<NCH>, dim3(grid), dim3(256), 0, s, pts, n, d, dp, lay_ctr, T8, ldt8, wrap_shift, t_out, ldt); \
    break;
  switch (nch) { X(1) X(2) X(3) X(4) }
#undef X
  return hipGetLastError();
}

hipError_t launch_prep(int dp, const PrepArgs &a, hipStream_t s) {
  if (wide_dims(dp)) return launch_prep_wide(dp, a, s);
  if (a.np <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.np + 255) / 256);
  const size_t lds = (size_t)a.d * dp * sizeof(double);
  switch (dp) {
#define X(D)                                                                            \
  case D:                                                                               \
    if (lds > 48 * 1024) {                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep<D>),    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                         (int)lds);                                     \
      if (e != hipSuccess) return e;                                                    \
    }                                                                                   \
    hipLaunchKernelGGL(k_prep<D>, dim3(grid), dim3(256), lds, s, a);                    \
    break;
    MLF_FOR_EACH_DP(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mlf
