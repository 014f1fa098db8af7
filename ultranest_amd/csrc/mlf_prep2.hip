// mlf_prep2.hip -- fused per-proposal stage of MLFriends.inside for AffineLayer-family regions:
//   H3 ellipsoid test (reference mlfriends.pyx:882-912, numpy-einsum order, no FMA)
//   T1 whitening      (:737-743 incl. wraps :529-536, FMA chain)
//   + binary16 quantisation and thresholds for the MFMA pre-filter (mlf_filter.hip)
// in ONE pass over the proposals, with HBM traffic at the algorithmic minimum:
//   * a wave's 64 proposal rows (64*d contiguous doubles) are read with fully coalesced loads
//     into an LDS staging area (row stride DP+1 doubles -> conflict-free per-lane row reads);
//     the first version read rows lane-strided and fetched every row ~3.7x (rocprofv3 FETCH_SIZE)
//   * whitened coordinates are written COORDINATE-major (t[c*P + p]): coalesced
//   * the binary16 query fragments are written as 16-byte pieces that are contiguous across lanes
// Arithmetic of H3 / T1 is identical to k_prep (mlf_prep.hip); -ffp-contract=off.
#include "mlf_filter_dev.hpp"
#include "mlf_prep2.hpp"

namespace mlf {

template <int DP>
__global__ __launch_bounds__(256) void k_prep2(Prep2Args a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int DPS = DP + 1;  // padded staging row stride (doubles)
  const int d = a.d;
  double *mat = lds;                                   // [d][DP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;
  double *stage = lds + (size_t)d * DP + (size_t)wave * 64 * DPS;

  const long long p0 = ((long long)blockIdx.x * nw + wave) * 64;  // first proposal of this wave
  const long long p = p0 + lane;
  const bool live = p < a.np;

  if (blockIdx.x == 0 && tid == 0 && a.counters) {
    a.counters[0] = 0;
    a.counters[1] = 0;
  }

  // ---- phase 0: coalesced copy of this wave's rows into LDS, zero padded ------------------
  {
    const long long rows_left = a.np - p0;
    const int nrows = rows_left >= 64 ? 64 : (rows_left > 0 ? (int)rows_left : 0);
    const double *src = a.pts + p0 * d;
    const int total = nrows * d;
    for (int e = lane; e < 64 * d; e += 64) {
      const int row = e / d, k = e - row * d;
      stage[row * DPS + k] = e < total ? src[e] : 0.0;
    }
    for (int e = lane; e < 64 * (DPS - d); e += 64) {
      const int row = e / (DPS - d), k = d + e - row * (DPS - d);
      stage[row * DPS + k] = 0.0;
    }
  }
  for (int e = tid; e < d * DP; e += blockDim.x) mat[e] = a.ell_A[e];
  __syncthreads();

  const double *row = stage + lane * DPS;

  // ---- phase 1: ellipsoid quadratic form, numpy c_einsum order ---------------------------
  bool inside;
  {
    double dl[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) dl[k] = (k < d) ? row[k] - a.ell_ctr[k] : 0.0;
    double acc = 0.0;
    for (int j = 0; j < d; ++j) {
      const double dj = row[j] - a.ell_ctr[j];
      const double2 *arow = reinterpret_cast<const double2 *>(mat + j * DP);
#pragma unroll
      for (int k = 0; k < DP; k += 2) {
        const double2 v = arow[k >> 1];
        acc += (dj * v.x) * dl[k];
        acc += (dj * v.y) * dl[k + 1];
      }
    }
    inside = live && (acc <= a.enlarge);
    if (live) a.gate[p] = inside ? 1 : 0;
  }
  if (!a.do_tr) return;
  __syncthreads();
  for (int e = tid; e < d * DP; e += blockDim.x) mat[e] = a.lay_Tt[e];
  __syncthreads();

  // ---- phase 2: whitening + quantisation ------------------------------------------------
  const bool quant = a.qF != nullptr;
  const int K = a.ks * 16;
  half_t *hrow = reinterpret_cast<half_t *>(stage + lane * DPS);  // reuses the lane's own row
  double nb = 0.0, nbn2 = 0.0;
  bool fits = true;
  if (__any(inside)) {
    double dl[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      double w = 0.0;
      if (k < d) {
        w = row[k];
        if (a.wrap_shift) {
          const double sh = a.wrap_shift[k];
          if (sh == sh) w = fmod(w + sh, 1.0);
        }
        w -= a.lay_ctr[k];
      }
      dl[k] = w;
    }
    const double sigma = quant ? a.stats[0] : 1.0;
    for (int c = 0; c < d; ++c) {
      const double2 *trow = reinterpret_cast<const double2 *>(mat + c * DP);
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DP; k += 2) {
        const double2 v = trow[k >> 1];
        acc = __builtin_fma(dl[k], v.x, acc);
        acc = __builtin_fma(dl[k + 1], v.y, acc);
      }
      if (inside) a.t_out[p * a.t_ldq + (long long)c * a.t_ldk] = acc;
      if (quant) {
        const double x = sigma * (acc - a.stats[8 + c]);
        if (!(fabs(x) <= 16000.0)) fits = false;   // NaN lands here too
        nbn2 += x * x;
        const half_t h = (half_t)(float)x;
        const double hv = (double)(float)h;
        nb += hv * hv;
        hrow[c] = (half_t)(-2.0f * (float)h);
      }
    }
  }
  if (!quant) return;

  int rt = inside ? 1 : 0;
  if (rt == 1 && (!fits || !(nbn2 <= 30000.0))) rt = 2;
  half_t pc[3] = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
  float lo_f = -1.0f, hi_f = -1.0f;
  if (rt == 1) {
    split3(nb, pc);
    if (!filter_thresholds(a.stats[0], a.stats[1], nbn2, a.r2, K, &lo_f, &hi_f)) {
      rt = 2;
      lo_f = hi_f = -1.0f;
      pc[0] = pc[1] = pc[2] = (half_t)0.0f;
    }
  }
  if (rt != 1)
    for (int k = 0; k < d; ++k) hrow[k] = (half_t)0.0f;
  for (int j = 0; j < 3; ++j) hrow[d + j] = (half_t)(rt == 1 ? 1.0f : 0.0f);
  for (int j = 3; j < 6; ++j) hrow[d + j] = pc[j - 3];
  for (int k = d + 6; k < K; ++k) hrow[k] = (half_t)0.0f;

  __syncthreads();  // LDS half rows complete (also a compiler barrier for the re-typed reads below)
  if (p < a.nqpad) {
    a.tlo[p] = lo_f;
    a.thi[p] = hi_f;
    if (live) {
      a.route[p] = (uint8_t)rt;
      a.best[p] = kNone;
    }
    // 16-byte fragment pieces: (group, kstep, half) -> lane r + 32*half holds k = 16*kstep + 8*half + 0..7
    const long long grp = p >> 5;
    const int r = (int)(p & 31);
    uint4 *dst = reinterpret_cast<uint4 *>(a.qF);
    const uint2 *h2 = reinterpret_cast<const uint2 *>(hrow);   // rows are 8-byte aligned
    for (int s = 0; s < a.ks; ++s)
      for (int hf = 0; hf < 2; ++hf) {
        const uint2 lo2 = h2[(s * 16 + hf * 8) >> 2];
        const uint2 hi2 = h2[((s * 16 + hf * 8) >> 2) + 1];
        dst[((size_t)grp * a.ks + s) * 64 + r + 32 * hf] = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
      }
  }
}

size_t prep2_lds_bytes(int d, int dp, int nw) {
  return ((size_t)d * dp + (size_t)nw * 64 * (dp + 1)) * sizeof(double);
}

int prep2_waves(int d, int dp) {
  for (int nw = 4; nw >= 1; --nw)
    if (prep2_lds_bytes(d, dp, nw) <= 160 * 1024) return nw;
  return 0;
}

hipError_t launch_prep2(int dp, const Prep2Args &a, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  const int nw = prep2_waves(a.d, dp);
  if (nw == 0) return hipErrorInvalidValue;
  const long long rows = a.qF ? a.nqpad : a.np;
  const unsigned grid = (unsigned)((rows + 64 * nw - 1) / (64 * nw));
  const size_t lds = prep2_lds_bytes(a.d, dp, nw);
  switch (dp) {
#define X(D)                                                                                   \
  case D: {                                                                                    \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep2<D>),           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                             \
    hipLaunchKernelGGL(k_prep2<D>, dim3(grid), dim3(64 * nw), lds, s, a);                      \
    break;                                                                                     \
  }
    MLF_FOR_EACH_DP_PREP2(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mlf
