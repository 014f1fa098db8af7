// mlf_prep2.hip -- fused per-proposal stage of MLFriends.inside for AffineLayer-family regions:
//   H3 ellipsoid test (reference mlfriends.pyx:882-912)
//   T1 whitening      (:737-743 incl. wraps :529-536; k-ascending FMA chain as in k_prep)
//   + binary16 quantisation and thresholds for the MFMA pre-filter (mlf_filter.hip)
// One lane = one proposal, everything in registers (compile-time indices only), both d x d
// matrices resident in LDS and broadcast:
//
//   * H3 costs 3 d^2 non-fused FP64 ops in the reference's (numpy einsum) order.  Here a bound is
//     evaluated first: with the Cholesky factor A = L L^T (host),  qt = |L^T delta|^2  needs d^2
//     FMAs, and |qt - q_ref| <= 2^-34 |A|_F |delta|^2 (far above the rounding of both forms:
//     (d^2+2) 2^-52 for the einsum order, O(d^1.5) 2^-53 for the factorised form).  Only proposals
//     with |qt - enlarge| inside that band take the exact einsum-order evaluation, so the mask is
//     bit-identical to k_prep's at a third of the arithmetic.
//   * the whitened coordinates are produced eight at a time (eight independent FMA chains per
//     coordinate block), written coordinate-major (coalesced) and quantised on the fly into
//     16-byte binary16 fragment pieces that are contiguous across lanes.
//   * HBM traffic: the proposal row is read twice (the second read applies wraps / the layer
//     centre), 8d + 2K + 9 bytes are written per proposal.
// -ffp-contract=off; FMAs only where written.
#include "mlf_filter_dev.hpp"
#include "mlf_prep2.hpp"

namespace mlf {

// circular dimensions are rare: keep the (large) inline expansion of fmod out of the unrolled loop
__device__ __attribute__((noinline)) double wrap_coordinate(double w, double shift) {
  return fmod(w + shift, 1.0);
}

template <int DP, bool WRAP>
__global__ __launch_bounds__(256, 2) void k_prep2(Prep2Args a) {
  constexpr int DP8 = (DP + 7) / 8 * 8;
  constexpr int KS = (DP + 6 + 15) / 16;
  constexpr int K = KS * 16;
  constexpr int NFULL = DP / 8;         // complete blocks of 8 real coordinates
  constexpr int C0T = NFULL * 8;        // first column of the tail
  constexpr int NTAIL = K - C0T;        // tail columns: remaining coordinates, norm pieces, zeros
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *Lt = lds;                     // [DP][DP]   Lt[k][j] = L[j][k]
  double *Tm = lds + DP * DP;           // [DP][DP8]  Tm[k][c] = T[k][c]
  // wave-private staging for 16 proposal rows: rows are fetched with fully coalesced loads (a wave's
  // 64 rows are contiguous in HBM) and redistributed lane = row through LDS; reading them
  // lane-strided straight from global memory cost 0.5 ms per 10^6 x 50 batch (rocprofv3 ablation)
  double *stage = lds + DP * DP + DP * DP8 + (threadIdx.x >> 6) * (16 * DP);

  const int tid = threadIdx.x;
  const int d = a.d;
  const long long p = (long long)blockIdx.x * 256 + tid;
  const bool live = p < a.np;
  const double *row = a.pts + (live ? p : 0) * (long long)d;

  if (blockIdx.x == 0 && tid == 0 && a.counters) {
    a.counters[0] = 0;
    a.counters[1] = 0;
  }
  {  // stage both matrices: all loads in flight before the first LDS store (a plain copy loop waits
     // for every load in turn: ~20 serialized L2 round trips per workgroup)
    constexpr int NL = (DP * DP + 255) / 256, NT = (DP * DP8 + 255) / 256;
    double tl[NL], tt[NT];
#pragma unroll
    for (int i = 0; i < NL; ++i) tl[i] = (tid + 256 * i < DP * DP) ? a.ell_Lt[tid + 256 * i] : 0.0;
    if (a.do_tr) {
#pragma unroll
      for (int i = 0; i < NT; ++i) tt[i] = (tid + 256 * i < DP * DP8) ? a.lay_T8[tid + 256 * i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (tid + 256 * i < DP * DP) Lt[tid + 256 * i] = tl[i];
    if (a.do_tr) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (tid + 256 * i < DP * DP8) Tm[tid + 256 * i] = tt[i];
    }
  }
  __syncthreads();

  const int lane = tid & 63;
  const long long p0w = p - lane;                    // first proposal of this wave
  const long long rows_left = a.np - p0w;
  const int wave_rows = rows_left >= 64 ? 64 : (rows_left > 0 ? (int)rows_left : 0);
  const double *wsrc = a.pts + (wave_rows > 0 ? p0w : 0) * (long long)d;
  // copies rows [16c, 16c+16) of this wave into `stage` (row stride d), zero filled past the batch end
  auto stage_rows = [&](int c) {
    constexpr int NIT = (16 * DP + 63) / 64;
    const int base = c * 16 * d, valid = wave_rows * d, chunk = 16 * d;
    double tmp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {   // all loads first, then the LDS stores
      const int e = lane + 64 * it;
      tmp[it] = (e < chunk && base + e < valid) ? wsrc[base + e] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = lane + 64 * it;
      if (e < chunk) stage[e] = tmp[it];
    }
  };

  // Up to DP = 52 the proposal row stays in registers for both stages (the workgroup is LDS-limited
  // to two waves per SIMD anyway, so the extra 2*DP VGPRs are free): one pass over HBM instead of
  // two.  Wider rows are staged a second time instead (they would spill).
  constexpr bool KEEPX = DP <= 52;
  double x[KEEPX ? DP : 2];
  // stages this wave's rows chunk by chunk and hands lane = row its coordinates: sink(k, value)
  auto fetch_row = [&](auto &&sink) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      stage_rows(c);
      if ((lane >> 4) == c) {
        const double *srow = stage + (lane & 15) * d;
#pragma unroll
        for (int k = 0; k < DP; ++k) sink(k, srow[k < d ? k : d - 1]);   // clamped: rows hold d <= DP values
      }
    }
  };

  // ---- H3 -----------------------------------------------------------------------------------
  bool inside = false;
  {
    double dl[DP];
    double nrm2 = 0.0;
    if constexpr (KEEPX) {
      fetch_row([&](int k, double v) { x[k] = v; });
#pragma unroll
      for (int k = 0; k < DP; ++k) dl[k] = (k < d) ? x[k] - a.ell_ctr[k] : 0.0;   // centres are zero padded
    } else {
      fetch_row([&](int k, double v) { dl[k] = (k < d) ? v - a.ell_ctr[k] : 0.0; });
    }
#pragma unroll
    for (int k = 0; k < DP; ++k) nrm2 = __builtin_fma(dl[k], dl[k], nrm2);
    double qt = 0.0;
    {
#pragma unroll
      for (int kb = 0; kb < DP; kb += 4) {
        double y[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = kb & ~1; j < DP; j += 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (kb + i < DP) {
              const double2 v = reinterpret_cast<const double2 *>(Lt + (kb + i) * DP)[j >> 1];
              y[i] = __builtin_fma(dl[j], v.x, y[i]);
              y[i] = __builtin_fma(dl[j + 1], v.y, y[i]);
            }
          }
          if ((j & 6) == 6) asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kb + i < DP) qt = __builtin_fma(y[i], y[i], qt);
      }
    }
    const double eps = a.ell_eps_scale * nrm2;
    const bool sure_in = a.chol_ok && (qt + eps < a.enlarge);
    const bool sure_out = a.chol_ok && (qt - eps > a.enlarge);
    inside = sure_in;
    const bool need_exact = live && !sure_in && !sure_out;   // also every NaN
    if (__any(need_exact)) {
      if (need_exact) {   // the reference's arithmetic: one accumulator, j outer, (d_j*A_jk)*d_k
        double acc = 0.0;
        for (int j = 0; j < d; ++j) {
          const double dj = row[j] - a.ell_ctr[j];
          const double *arow = a.ell_A + (size_t)j * DP;
#pragma unroll
          for (int k = 0; k < DP; ++k) acc += (dj * arow[k]) * dl[k];
        }
        inside = acc <= a.enlarge;
      }
    }
    inside = inside && live;
    if (live) a.gate[p] = inside ? 1 : 0;
  }
  if (!a.do_tr) return;

  // ---- T1 + quantisation --------------------------------------------------------------------
  const bool quant = a.qF != nullptr;
  if (!quant && !__any(inside)) return;
  const long long grp = p >> 5;
  const int r = (int)(p & 31);
  uint4 *qdst = reinterpret_cast<uint4 *>(a.qF);
  const bool wr = quant && p < a.nqpad;
  const double sigma = quant ? a.stats[0] : 1.0;

  double dl[DP];
  auto whiten_input = [&](int k, double w) {
    if (WRAP) {
      const double sh = a.wrap_shift[k];
      if (sh == sh) {   // NaN marks an unwrapped dimension
        // fmod(w + sh, 1): for 0 <= x < 2 (cube coordinates) it is x or x - 1, both exact
        const double xs = w + sh;
        w = (xs >= 0.0 && xs < 2.0) ? (xs >= 1.0 ? xs - 1.0 : xs) : wrap_coordinate(w, sh);
      }
    }
    w -= a.lay_ctr[k];
    dl[k] = (k < d) ? w : 0.0;
  };
  if constexpr (KEEPX) {
#pragma unroll
    for (int k = 0; k < DP; ++k) whiten_input(k, x[k]);
  } else {
    fetch_row(whiten_input);
  }

  double nb = 0.0, nbn2 = 0.0;
  bool fits = true;

  // quantise one whitened coordinate; returns the operand value -2*f16(sigma (t - c))
  auto quantise = [&](double t, int c) -> half_t {
    const double x = sigma * (t - a.stats[8 + c]);
    if (!(fabs(x) <= 16000.0)) fits = false;   // NaN lands here too
    nbn2 = __builtin_fma(x, x, nbn2);
    const half_t h = (half_t)(float)x;
    const double hv = (double)(float)h;
    nb += hv * hv;                              // exact
    return (half_t)(-2.0f * (float)h);
  };
  auto piece_index = [&](int c0) -> size_t {   // 16-byte piece holding columns c0 .. c0+7
    return ((size_t)grp * KS + (c0 >> 4)) * 64 + r + 32 * ((c0 >> 3) & 1);
  };
  auto pack2 = [](half_t lo, half_t hi) -> unsigned {
    return (unsigned)__builtin_bit_cast(unsigned short, lo) |
           ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
  };

#pragma unroll 1
  for (int cc = 0; cc < NFULL; ++cc) {
    const double *tbase = Tm + cc * 8;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double2 *tk = reinterpret_cast<const double2 *>(tbase + k * DP8);
      const double2 v0 = tk[0], v1 = tk[1], v2 = tk[2], v3 = tk[3];
      acc[0] = __builtin_fma(dl[k], v0.x, acc[0]);
      acc[1] = __builtin_fma(dl[k], v0.y, acc[1]);
      acc[2] = __builtin_fma(dl[k], v1.x, acc[2]);
      acc[3] = __builtin_fma(dl[k], v1.y, acc[3]);
      acc[4] = __builtin_fma(dl[k], v2.x, acc[4]);
      acc[5] = __builtin_fma(dl[k], v2.y, acc[5]);
      acc[6] = __builtin_fma(dl[k], v3.x, acc[6]);
      acc[7] = __builtin_fma(dl[k], v3.y, acc[7]);
      // keep the evaluation k-major: without this pin hipcc (ROCm 7.2) evaluates one accumulator chain
      // at a time and spills the other seven operands of every LDS read to scratch (2.6 KB/lane)
      if ((k & 1) == 1)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                     "+v"(acc[6]), "+v"(acc[7]));
    }
    half_t hq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cc * 8 + i;
      if (inside && c < d) a.t_out[p * a.t_ldq + (long long)c * a.t_ldk] = acc[i];
      hq[i] = (quant && inside) ? quantise(acc[i], c) : (half_t)0.0f;
    }
    if (wr)
      qdst[piece_index(cc * 8)] = make_uint4(pack2(hq[0], hq[1]), pack2(hq[2], hq[3]), pack2(hq[4], hq[5]),
                                             pack2(hq[6], hq[7]));
  }

  // tail: remaining real coordinates, then the norm / ones columns, then zero padding
  half_t tail[NTAIL];
#pragma unroll
  for (int i = 0; i < NTAIL; ++i) tail[i] = (half_t)0.0f;
#pragma unroll
  for (int i = 0; i < DP - C0T; ++i) {
    const int c = C0T + i;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) acc = __builtin_fma(dl[k], Tm[k * DP8 + c], acc);
    if (inside && c < d) a.t_out[p * a.t_ldq + (long long)c * a.t_ldk] = acc;
    if (quant && inside) tail[i] = quantise(acc, c);
  }
  if (!quant) return;

  int rt = inside ? 1 : 0;
  if (rt == 1 && (!fits || !(nbn2 <= 30000.0))) rt = 2;
  float lo_f = -1.0f, hi_f = -1.0f;
  if (rt == 1) {
    half_t pc[3];
    split3(nb, pc);
    if (filter_thresholds(a.stats[0], a.stats[1], nbn2, a.r2, K, &lo_f, &hi_f)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        tail[DP - C0T + j] = (half_t)1.0f;     // x |ah|^2 pieces of the live point
        tail[DP - C0T + 3 + j] = pc[j];        // x ones column of the live point
      }
    } else {
      rt = 2;
      lo_f = hi_f = -1.0f;
    }
  }
  if (!wr) return;
  if (rt != 1) {   // not filtered after all: every operand column of this query must be zero
#pragma unroll
    for (int i = 0; i < NTAIL; ++i) tail[i] = (half_t)0.0f;
    if (rt == 2)
      for (int cc = 0; cc < NFULL; ++cc) qdst[piece_index(cc * 8)] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int i = 0; i < NTAIL; i += 8)
    qdst[piece_index(C0T + i)] = make_uint4(pack2(tail[i], tail[i + 1]), pack2(tail[i + 2], tail[i + 3]),
                                            pack2(tail[i + 4], tail[i + 5]), pack2(tail[i + 6], tail[i + 7]));
  a.tlo[p] = lo_f;
  a.thi[p] = hi_f;
  if (live) {
    a.route[p] = (uint8_t)rt;
    a.best[p] = kNone;
  }
}

static size_t prep2_lds_bytes(int dp) {
  const int dp8 = (dp + 7) / 8 * 8;
  return ((size_t)dp * dp + (size_t)dp * dp8 + (size_t)4 * 16 * dp) * sizeof(double);
}

bool prep2_usable(int dp) { return dp <= 64 && prep2_lds_bytes(dp) <= 100 * 1024; }

hipError_t launch_prep2(int dp, const Prep2Args &a, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  if (!prep2_usable(dp)) return hipErrorInvalidValue;
  const long long rows = a.qF ? a.nqpad : a.np;
  const unsigned grid = (unsigned)((rows + 255) / 256);
  const size_t lds = prep2_lds_bytes(dp);
  const bool wrap = a.wrap_shift != nullptr;
  switch (dp) {
#define X(D)                                                                                      \
  case D: {                                                                                       \
    static bool attr_set = false;                                                                 \
    if (!attr_set && lds > 48 * 1024) {                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep2<D, false>),     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e == hipSuccess)                                                                        \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep2<D, true>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
      if (e != hipSuccess) return e;                                                              \
      attr_set = true;                                                                            \
    }                                                                                             \
    if (wrap)                                                                                     \
      hipLaunchKernelGGL((k_prep2<D, true>), dim3(grid), dim3(256), lds, s, a);                   \
    else                                                                                          \
      hipLaunchKernelGGL((k_prep2<D, false>), dim3(grid), dim3(256), lds, s, a);                  \
    break;                                                                                        \
  }
    MLF_FOR_EACH_DP_PREP2(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mlf
