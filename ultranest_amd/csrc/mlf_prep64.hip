// mlf_prep64.hip -- per-proposal stage of MLFriends.inside for 65 ... 128 dimensions on the FP64 matrix cores (round 5):
//   H3 ellipsoid test (reference mlfriends.pyx:882-912) in the bounded |L^T delta|^2 form of k_prep3 (the reference's einsum
//   order only inside that form's own band), T1 whitening (:737-743 incl. wraps :529-536).
//
// Rounds 1-4 sent these dimensionalities through the vector kernel k_prep<DP> (lane = proposal, 2 DP registers per lane, the
// matrix broadcast from 90 KB of LDS, 3 d^2 non-fused operations for the quadratic form + 2 d^2 for the whitening: 4-6 ms per
// 10^6 x 100, one wave per SIMD), because the matrix-core stages stop at d = 64: k_prep3 keeps both fragment tables in LDS
// (200 KB at d = 128) and a proposal's k-steps in registers, k_prep4's binary16 tables and staging need 280 KB.  This kernel
// is the k-streamed form: a wave takes 16 proposals, their rows sit in the wave's LDS buffer, and for every k-step of four
// coordinates the A fragment comes straight from the row-major matrix in L2 (one 8-byte load per lane and matrix instruction,
// both matrices together 200 KB: L2-resident) -- v_mfma_f64_16x16x4_f64: A = M[4 ks + (l >> 4)][16 ct + (l & 15)], B = the
// proposals' coordinates 4 ks + (l >> 4), C = 16 outputs x 16 proposals.  The instruction accumulates k-ascending with one
// rounding per fused multiply-add (measured bit-identical to the scalar chain, scripts/probes/mfma64_probe.hip; k_prep3 and
// k_uncertain whiten this way), so the whitened coordinates are k_prep's bit for bit -- live points (whitened by k_prep) and
// proposals still meet at distance exactly 0 -- and the quadratic form keeps k_prep3's bound: |qt - q_ref| <= 2^-34 |A|_F
// |delta|^2, proposals inside that band take the einsum-order evaluation.
#include "mlf_prep64.hpp"

#include <math.h>

#include <atomic>

namespace mlf {

typedef double double4m __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((noinline)) double wrap_coordinate64(double w, double shift) { return fmod(w + shift, 1.0); }

__device__ __forceinline__ double quad_sum64(double v) {   // sum over lanes l, l^16, l^32, l^48
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int kP64Waves = 4;

// NK = k-steps of 4 coordinates (4 NK >= dp), NC = output tiles of 16
template <int NK>
__global__ __launch_bounds__(64 * kP64Waves, 2) void k_prep_mfma64(Prep64Args a) {
  constexpr int NC = (NK + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane & 15, kq = lane >> 4;
  const int d = a.d, dp = a.dp;
  const int ds = d | 1;                          // odd row stride: the 16 proposals of a k-step read 16 different banks
  double *xrow = lds64 + (size_t)wv * 16 * ds;   // [16][ds] this wave's proposals as handed over, later their whitened rows
  const long long ntiles = (a.np + 15) / 16;
  const long long total = a.np * (long long)d;
  for (long long tile = (long long)blockIdx.x * kP64Waves + wv; tile < ntiles; tile += (long long)gridDim.x * kP64Waves) {
    const long long p = tile * 16 + q;
    const bool live = p < a.np;
    // ---- the tile's 16 rows: 16 d contiguous doubles, all requests first
    {
      const long long base = tile * 16 * (long long)d;
      constexpr int kPer = (16 * 128 + 63) / 64;   // 32
      double v[kPer];
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int e = lane + 64 * i;
        const long long g = base + e;
        v[i] = (e < 16 * d && g < total) ? a.pts[g] : 0.0;
      }
      __builtin_amdgcn_wave_barrier();   // the rows of the tile before have been written out
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int e = lane + 64 * i;
        if (e < 16 * d) {
          const int rr = e / d;
          xrow[rr * ds + (e - rr * d)] = v[i];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const double *row = xrow + q * ds;
    // ---- H3 bound: y = L^T delta on the matrix cores, qt = |y|^2 (k_prep3's form)
    double qt = 0.0, nrm2 = 0.0;
    {
      double4m y[NC];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) y[ct] = (double4m){0.0, 0.0, 0.0, 0.0};
      // (two k-steps per pass: unrolled in full, the compiler kept every fragment address and half of the fragments live --
      // 256 + 224 registers, 710 scalar spills, one wave per SIMD)
#pragma unroll 2
      for (int ks = 0; ks < NK; ++ks) {
        const int k = 4 * ks + kq;
        const double dl = k < d ? row[k] - a.ell_ctr[k] : 0.0;
        nrm2 = __builtin_fma(dl, dl, nrm2);
        double afr[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
          const int c = 16 * ct + q;
          // y_c = sum_k delta_k L[k][c] (row-major L, stride dp: the 16 lanes of a k read 128 contiguous bytes); L[k][c] = 0
          // for c > k: the tiles above the diagonal are skipped
          afr[ct] = (16 * ct <= 4 * ks + 3 && k < d && c < d) ? a.ell_L[(size_t)k * dp + c] : 0.0;
        }
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
          if (16 * ct <= 4 * ks + 3) y[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ct], dl, y[ct], 0, 0, 0);
      }
#pragma unroll
      for (int ct = 0; ct < NC; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) qt = __builtin_fma(y[ct][r], y[ct][r], qt);
      qt = quad_sum64(qt);
      nrm2 = quad_sum64(nrm2);
    }
    const double eps = a.ell_eps_scale * nrm2;
    const bool sure_in = a.chol_ok && (qt + eps < a.enlarge);
    const bool sure_out = a.chol_ok && (qt - eps > a.enlarge);
    bool inside = sure_in;
    const bool need_exact = live && !sure_in && !sure_out;   // also every NaN
    if (__any(need_exact)) {
      // the reference's arithmetic: one accumulator, j outer, (d_j A_jk) d_k, no fma; by the proposal's first lane
      double acc = 0.0;
      if (need_exact && kq == 0) {
        const double *grow = a.pts + p * (long long)d;
        for (int j = 0; j < d; ++j) {
          const double dj = grow[j] - a.ell_ctr[j];
          const double *arow = a.ell_A + (size_t)j * a.lda;
          for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * (grow[k] - a.ell_ctr[k]);
        }
      }
      acc = __shfl(acc, q, 64);
      if (need_exact) inside = acc <= a.enlarge;
    }
    inside = inside && live;
    if (live && kq == 0) a.gate[p] = inside ? 1 : 0;
    if (!a.do_tr || !__any(inside)) continue;
    // ---- T1: t = delta_w T on the matrix cores, k ascending, one fma per term
    double4m t[NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) t[ct] = (double4m){0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int ks = 0; ks < NK; ++ks) {
      const int k = 4 * ks + kq;
      double dw = 0.0;
      if (k < d) {
        double w = row[k];
        if (a.wrap_shift) {
          const double sh = a.wrap_shift[k];
          if (sh == sh) {   // NaN marks an unwrapped dimension
            const double xs = w + sh;
            w = (xs >= 0.0 && xs < 2.0) ? (xs >= 1.0 ? xs - 1.0 : xs) : wrap_coordinate64(w, sh);
          }
        }
        dw = w - a.lay_ctr[k];
      }
      double afr[NC];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) {
        const int c = 16 * ct + q;
        afr[ct] = (k < dp && c < d) ? a.T8[(size_t)k * a.ldt8 + c] : 0.0;
      }
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) t[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ct], dw, t[ct], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();   // every lane has read the raw rows
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * ct + kq + 4 * r;   // output row of the accumulator register (mlf_prep3.hip)
        if (c < d) xrow[q * ds + c] = t[ct][r];
      }
    __builtin_amdgcn_wave_barrier();
    // ---- the whitened rows leave as they came: contiguous
    {
      const unsigned long long in_bits = __ballot(inside && kq == 0);   // bit q: proposal q of the tile is inside
      double *out = a.t_out + tile * 16 * a.ldt;
      for (int e = lane; e < 16 * d; e += 64) {
        const int rr = e / d, c = e - rr * d;
        if ((in_bits >> rr) & 1ull) out[(long long)rr * a.ldt + c] = xrow[rr * ds + c];
      }
    }
  }
}

}  // namespace

bool prep64_usable(int d) { return d > 64 && d <= 128; }

hipError_t launch_prep64(const Prep64Args &a, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  const int nk = (a.dp + 3) / 4;
  const size_t lds = (size_t)kP64Waves * 16 * (a.d | 1) * sizeof(double);
  const long long ntiles = (a.np + 15) / 16;
  long long wgs = (ntiles + kP64Waves - 1) / kP64Waves;
  if (wgs > 256 * 8) wgs = 256 * 8;   // grid-stride beyond 8 workgroups per CU
  const dim3 grid((unsigned)wgs), block(64 * kP64Waves);
  if (lds > 48 * 1024) {   // above the default grant (66 KB at d = 128): the attribute belongs to (function, device)
    static DeviceGrant grant;
    if (hipError_t e = grant.ensure([] {
          hipError_t g = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep_mfma64<20>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
          if (g == hipSuccess) g = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep_mfma64<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
          if (g == hipSuccess) g = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep_mfma64<28>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
          if (g == hipSuccess) g = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep_mfma64<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
          return g;
        }))
      return e;
  }
  // instances by k-steps: dp is a multiple of 16 above 64 (80, 96, 112, 128)
  switch (nk) {
    case 20: hipLaunchKernelGGL(k_prep_mfma64<20>, grid, block, lds, s, a); break;
    case 24: hipLaunchKernelGGL(k_prep_mfma64<24>, grid, block, lds, s, a); break;
    case 28: hipLaunchKernelGGL(k_prep_mfma64<28>, grid, block, lds, s, a); break;
    case 32: hipLaunchKernelGGL(k_prep_mfma64<32>, grid, block, lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mlf
