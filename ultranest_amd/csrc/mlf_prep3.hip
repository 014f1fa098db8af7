// mlf_prep3.hip -- fused per-proposal stage of MLFriends.inside for AffineLayer-family regions on
// the FP64 matrix cores:
//   H3 ellipsoid test (reference mlfriends.pyx:882-912), bounded form of mlf_prep2.hip
//   T1 whitening      (:737-743 incl. wraps :529-536)
//   + binary16 quantisation and thresholds for the MFMA pre-filter (mlf_filter.hip)
//
// Both d x d products are GEMMs over the batch:  Y = Lt . (X - c_e)^T  and  T = T^T . (X' - c_l)^T,
// evaluated with v_mfma_f64_16x16x4_f64: A = a 16 x 4 matrix fragment (LDS, pre-arranged by the
// host), B = 4 coordinates x 16 proposals straight from the proposal rows, C = 16 outputs x 16
// proposals.  The instruction accumulates k-ascending with one rounding per FMA (measured: bit
// identical to the scalar FMA chain, scripts/probes/mfma64_probe.hip), so T equals k_prep's /
// k_prep2's whitening bit for bit -- live points whitened by k_prep and proposals whitened here
// still meet at distance exactly 0.  The vector version (k_prep2) is bound by broadcasting every
// matrix element to the lanes through LDS once per wave; here an operand register feeds 16 x 16 x 4
// multiply-adds and the kernel runs at the FP64 issue rate (70+ TFLOP/s measured).
//
// Layout per 16-proposal tile: lane l holds proposal (l & 15); of every 16-row block of an output
// it holds rows (l >> 4) + 4 r, r = 0..3.  Per-proposal scalars are therefore reduced over the four
// lanes l, l^16, l^32, l^48.  binary16 fragment pieces (8 consecutive columns) are assembled
// through a wave-private LDS transpose.
// -ffp-contract=off; FMAs only where written.
#include "mlf_prep3.hpp"

#include <vector>

#include "mlf_filter_dev.hpp"

namespace mlf {

typedef double double4v __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((noinline)) double wrap_coordinate3(double w, double shift) { return fmod(w + shift, 1.0); }

__device__ __forceinline__ double quad_sum(double v) {   // sum over lanes l, l^16, l^32, l^48
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int kTileRowHalfs = 136;   // binary16 columns per proposal in the transpose buffer (128 + pad)

}  // namespace

// NC = ceil(d / 16) output row tiles
template <int NC, bool WRAP>
__global__ __launch_bounds__(256, 2) void k_prep3(Prep3Args a) {
  extern __shared__ __attribute__((aligned(16))) double lds3[];
  const int nk = a.nk;
  double *LtF = lds3;                         // [NC][nk][64]
  double *TtF = lds3 + (size_t)NC * nk * 64;  // [NC][nk][64]
  half_t *tbuf = reinterpret_cast<half_t *>(lds3 + (size_t)2 * NC * nk * 64) + (threadIdx.x >> 6) * (16 * kTileRowHalfs);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int pl = lane & 15;    // proposal within the tile
  const int kq = lane >> 4;    // coordinate within a k-step / row residue of the outputs
  const int d = a.d;
  const int KS = a.ks;         // binary16 k-steps of the filter operand (16 columns each)
  const bool quant = a.qF != nullptr;

  if (blockIdx.x == 0 && tid == 0 && a.counters) {
    a.counters[0] = 0;
    a.counters[1] = 0;
  }
  for (int e = tid; e < NC * nk * 64; e += 256) {
    LtF[e] = a.LtF[e];
    if (a.do_tr) TtF[e] = a.TtF[e];
  }
  __syncthreads();

  const long long rows_total = quant ? a.nqpad : a.np;
  const long long ntiles = (rows_total + 15) / 16;
  const long long wave_id = (long long)blockIdx.x * 4 + (tid >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const double sigma = quant ? a.stats[0] : 1.0;
  uint4 *qdst = reinterpret_cast<uint4 *>(a.qF);

  // raw coordinates of one tile in operand order: x[ks] = pts[tile*16 + pl][4 ks + kq] (0 outside)
  auto load_tile = [&](long long tile, double *x) {
    const long long p = tile * 16 + pl;
    const bool live = tile < ntiles && p < a.np;
    const double *row = a.pts + (live ? p : 0) * (long long)d;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + kq;
      x[ks] = (ks < nk && live && k < d) ? row[k] : 0.0;
    }
  };

  // four consecutive tiles (64 proposals) per wave and step: the proposals of a binary16 fragment
  // group (32 rows) stay within one wave.  The next tile's rows are requested before the current
  // tile's matrix products are issued (one wave has ~5000 cycles of MFMA work per tile to hide them).
  double xcur[16];
  if (wave_id * 4 < ntiles) load_tile(wave_id * 4, xcur);
  for (long long t4 = wave_id; t4 * 4 < ntiles; t4 += nwaves) {
    for (int sub = 0; sub < 4; ++sub) {
      const long long tile = t4 * 4 + sub;
      if (tile >= ntiles) break;
      const long long p = tile * 16 + pl;
      const bool live = p < a.np;
      const double *row = a.pts + (live ? p : 0) * (long long)d;

      // ---- operands: 4 coordinates x 16 proposals per k-step -------------------------------
      double dl[16], dw[16];
      double nrm2 = 0.0;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        dl[ks] = 0.0;
        dw[ks] = 0.0;
        if (ks < nk) {
          const int k = 4 * ks + kq;
          const bool ok = live && k < d;
          const double x = xcur[ks];
          double w = x;
          if (WRAP && ok && a.do_tr) {
            const double sh = a.wrap_shift[k];
            if (sh == sh) {   // NaN marks an unwrapped dimension
              const double xs = w + sh;
              w = (xs >= 0.0 && xs < 2.0) ? (xs >= 1.0 ? xs - 1.0 : xs) : wrap_coordinate3(w, sh);
            }
          }
          dl[ks] = ok ? x - a.ell_ctr[k] : 0.0;
          dw[ks] = (ok && a.do_tr) ? w - a.lay_ctr[k] : 0.0;
          nrm2 = __builtin_fma(dl[ks], dl[ks], nrm2);
        }
      }
      load_tile(sub < 3 ? tile + 1 : (t4 + nwaves) * 4, xcur);   // prefetch
      nrm2 = quad_sum(nrm2);

      // ---- H3 bound: Y = Lt . delta, qt = |Y|^2 ---------------------------------------------
      double qt = 0.0;
      {
        double4v y[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) y[ct] = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks < nk) {
#pragma unroll
            for (int ct = 0; ct < NC; ++ct)
              if (4 * ks + 3 >= 16 * ct)   // Lt[kb][j] = 0 for j < kb: tiles left of the diagonal are empty
                y[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(LtF[((size_t)ct * nk + ks) * 64 + lane], dl[ks], y[ct], 0, 0, 0);
          }
        }
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) qt = __builtin_fma(y[ct][r], y[ct][r], qt);
        qt = quad_sum(qt);
      }
      const double eps = a.ell_eps_scale * nrm2;
      const bool sure_in = a.chol_ok && (qt + eps < a.enlarge);
      const bool sure_out = a.chol_ok && (qt - eps > a.enlarge);
      bool inside = sure_in;
      const bool need_exact = live && !sure_in && !sure_out;   // also every NaN
      if (__any(need_exact)) {
        // the reference's arithmetic: one accumulator, j outer, (d_j*A_jk)*d_k; done by the
        // proposal's first lane and shared with the other three
        double acc = 0.0;
        if (need_exact && kq == 0) {
          for (int j = 0; j < d; ++j) {
            const double dj = row[j] - a.ell_ctr[j];
            const double *arow = a.ell_A + (size_t)j * a.lda;
            for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * (row[k] - a.ell_ctr[k]);
          }
        }
        acc = __shfl(acc, pl, 64);
        if (need_exact) inside = acc <= a.enlarge;
      }
      inside = inside && live;
      if (live && kq == 0) a.gate[p] = inside ? 1 : 0;
      if (!a.do_tr) continue;
      if (!quant && !__any(inside)) continue;

      // ---- T1: T = T^T . delta_w ------------------------------------------------------------
      double4v t[NC];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) t[ct] = (double4v){0.0, 0.0, 0.0, 0.0};
      if (__any(inside)) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks < nk) {
#pragma unroll
            for (int ct = 0; ct < NC; ++ct)
              t[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(TtF[((size_t)ct * nk + ks) * 64 + lane], dw[ks], t[ct], 0, 0, 0);
          }
        }
      }
      if (inside) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * ct + kq + 4 * r;
            if (c < d) a.t_out[p * a.t_ldq + (long long)c * a.t_ldk] = t[ct][r];
          }
      }
      if (!quant) continue;

      // ---- binary16 quantisation (columns 0 .. d-1), norms reduced over the proposal's lanes --
      half_t *trow = tbuf + pl * kTileRowHalfs;
      double nb = 0.0, nbn2 = 0.0;
      bool fits = true;
      const int K = KS * 16;
      // zero the whole row first (columns >= d, and everything for proposals that are not filtered)
      for (int c = kq; c < K; c += 4) trow[c] = (half_t)0.0f;
      __builtin_amdgcn_wave_barrier();
      if (inside) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * ct + kq + 4 * r;
            if (c < d) {
              const double x = sigma * (t[ct][r] - a.stats[8 + c]);
              if (!(fabs(x) <= 16000.0)) fits = false;   // NaN lands here too
              nbn2 = __builtin_fma(x, x, nbn2);
              const half_t h = (half_t)(float)x;
              const double hv = (double)(float)h;
              nb += hv * hv;
              trow[c] = (half_t)(-2.0f * (float)h);
            }
          }
      }
      nb = quad_sum(nb);
      nbn2 = quad_sum(nbn2);
      {
        int f = fits ? 1 : 0;
        f &= __shfl_xor(f, 16, 64);
        f &= __shfl_xor(f, 32, 64);
        fits = f != 0;
      }

      int rt = inside ? 1 : 0;
      if (rt == 1 && (!fits || !(nbn2 <= 30000.0))) rt = 2;
      float lo_f = -1.0f, hi_f = -1.0f;
      half_t pc[3] = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
      if (rt == 1) {
        split3(nb, pc);
        if (!filter_thresholds(a.stats[0], a.stats[1], nbn2, a.r2, K, &lo_f, &hi_f)) {
          rt = 2;
          lo_f = hi_f = -1.0f;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (rt == 1) {
        if (kq == 0) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            trow[a.dp + j] = (half_t)1.0f;     // x |ah|^2 pieces of the live point
            trow[a.dp + 3 + j] = pc[j];        // x ones column of the live point
          }
        }
      } else {   // not filtered: every operand column of this query must be zero
        for (int c = kq; c < K; c += 4) trow[c] = (half_t)0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      // 16 proposals x (K / 8) pieces of 16 bytes, contiguous across proposals in the fragment layout
      if (p - pl < a.nqpad) {
        const int npieces = 16 * (K >> 3);
        for (int q = lane; q < npieces; q += 64) {
          const int qp = q & 15, c0 = (q >> 4) << 3;
          const long long pp = tile * 16 + qp;
          if (pp < a.nqpad) {
            const uint4 v = *reinterpret_cast<const uint4 *>(tbuf + qp * kTileRowHalfs + c0);
            const long long grp = pp >> 5;
            const int r32 = (int)(pp & 31);
            qdst[((size_t)grp * KS + (c0 >> 4)) * 64 + r32 + 32 * ((c0 >> 3) & 1)] = v;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (kq == 0 && p < a.nqpad) {
        a.tlo[p] = lo_f;
        a.thi[p] = hi_f;
        if (live) {
          a.route[p] = (uint8_t)rt;
          a.best[p] = kNone;
        }
      }
    }
  }
}

size_t prep3_fragment_count(int d) { return (size_t)((d + 15) / 16) * ((d + 3) / 4) * 64; }

void prep3_fragments(const double *M, int d, bool transpose, double *out) {
  const int nc = (d + 15) / 16, nk = (d + 3) / 4;
  for (int ct = 0; ct < nc; ++ct)
    for (int ks = 0; ks < nk; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int row = 16 * ct + (l & 15), k = 4 * ks + (l >> 4);
        double v = 0.0;
        if (row < d && k < d) v = transpose ? M[(size_t)k * d + row] : M[(size_t)row * d + k];
        out[((size_t)ct * nk + ks) * 64 + l] = v;
      }
}

static size_t prep3_lds_bytes(int d) {
  return 2 * prep3_fragment_count(d) * sizeof(double) + (size_t)4 * 16 * kTileRowHalfs * sizeof(half_t);
}

bool prep3_usable(int d) { return d >= 1 && d <= 64; }

hipError_t launch_prep3(const Prep3Args &a, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  if (!prep3_usable(a.d)) return hipErrorInvalidValue;
  const long long rows = a.qF ? a.nqpad : a.np;
  const long long chunks = (rows + 63) / 64;            // 64 proposals per wave and step
  long long grid = (chunks + 3) / 4;
  if (grid > 512) grid = 512;                           // persistent: 2 workgroups per CU
  const size_t lds = prep3_lds_bytes(a.d);
  const int nc = (a.d + 15) / 16;
  const bool wrap = a.wrap_shift != nullptr;
#define LAUNCH3(NCV)                                                                                        \
  {                                                                                                         \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep3<NCV, false>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);            \
      if (e == hipSuccess)                                                                                  \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep3<NCV, true>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                     \
      if (e != hipSuccess) return e;                                                                        \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    if (wrap)                                                                                               \
      hipLaunchKernelGGL((k_prep3<NCV, true>), dim3((unsigned)grid), dim3(256), lds, s, a);                 \
    else                                                                                                    \
      hipLaunchKernelGGL((k_prep3<NCV, false>), dim3((unsigned)grid), dim3(256), lds, s, a);                \
  }
  switch (nc) {
    case 1: LAUNCH3(1) break;
    case 2: LAUNCH3(2) break;
    case 3: LAUNCH3(3) break;
    default: LAUNCH3(4) break;
  }
#undef LAUNCH3
  return hipGetLastError();
}

}  // namespace mlf
