// mlf_prep3.hip -- fused per-proposal stage of MLFriends.inside for AffineLayer-family regions on
// the FP64 matrix cores:
//   H3 ellipsoid test (reference mlfriends.pyx:882-912) in the bounded |L^T delta|^2 form (the reference's einsum order
//   only inside that form's own band)
//   T1 whitening      (:737-743 incl. wraps :529-536)
//   + binary16 quantisation and thresholds for the MFMA pre-filter (mlf_filter.hip)
//
// Both d x d products are GEMMs over the batch:  Y = Lt . (X - c_e)^T  and  T = T^T . (X' - c_l)^T,
// evaluated with v_mfma_f64_16x16x4_f64: A = a 16 x 4 matrix fragment (LDS, pre-arranged by the
// host), B = 4 coordinates x 16 proposals straight from the proposal rows, C = 16 outputs x 16
// proposals.  The instruction accumulates k-ascending with one rounding per FMA (measured: bit
// identical to the scalar FMA chain, scripts/probes/mfma64_probe.hip), so T equals k_prep's whitening bit for
// bit -- live points whitened by k_prep and proposals whitened here still meet at distance exactly 0.  A vector
// version of this stage (k_prep2, rounds 1-2: lane = proposal, both matrices broadcast from LDS; 0.57 against 0.31 ms,
// removed in round 3) was bound by broadcasting every matrix element to the lanes once per wave; here an operand
// register feeds 16 x 16 x 4 multiply-adds and the kernel runs at the FP64 issue rate (70+ TFLOP/s measured).
//
// Layout per 16-proposal tile: lane l holds proposal (l & 15); of every 16-row block of an output
// it holds rows (l >> 4) + 4 r, r = 0..3.  Per-proposal scalars are therefore reduced over the four
// lanes l, l^16, l^32, l^48.  Proposal rows are fetched with contiguous 8-byte-per-lane loads (a
// tile's 16 rows are one contiguous block) and redistributed into operand order through a
// wave-private LDS buffer (row stride = 4 mod 8 dwords: conflict-free ds_read_b64); fetching them
// in operand order straight from HBM (32-byte segments) cost 0.15 ms per 10^6 x 50 batch.  The same
// buffer is reused for the binary16 transpose (8 consecutive columns per 16-byte fragment piece).
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

// instantiated k-step counts are 1,2,3,4,5,6,8,10,13,16: an instance NK serves 4 prev(NK) < d <= 4 NK
constexpr int prev_ksteps(int nk) { return nk == 16 ? 13 : nk == 13 ? 10 : nk == 10 ? 8 : nk == 8 ? 6 : nk - 1; }

constexpr int kTileRowHalfs = 136;   // binary16 columns per proposal in the transpose buffer (128 + pad)
constexpr int kWaveBufBytes = 9216;  // per-wave staging: 16 rows x <= 66 doubles + 64 spare doubles, or 16 x 136 binary16

}  // namespace

// NK = k-steps of 4 coordinates (the host pads d up to 4 NK with zero matrix rows / columns)
template <int NK, bool WRAP>
__global__ __launch_bounds__(256, 2) void k_prep3(Prep3Args a) {
  extern __shared__ __attribute__((aligned(16))) double lds3[];
  constexpr int nk = NK;
  constexpr int NC = (NK + 3) / 4;   // output row tiles of 16
  // Lt fragments: only the tiles on and right of the diagonal are stored (ks >= 4 ct)
  double *LtF = lds3;
  const int nlt = a.nlt;                      // number of stored Lt tiles
  double *TtF = lds3 + (size_t)nlt * 64;      // [NC][nk][64]
  // per-coordinate constants (centres, wrap shifts, quantisation centres): read per tile, so they
  // live in LDS -- as global loads they were three exposed L2 round trips per tile
  double *c_ell = lds3 + (size_t)nlt * 64 + (size_t)NC * nk * 64;   // [64] each
  double *c_lay = c_ell + 64, *c_wrap = c_ell + 128, *c_stat = c_ell + 192;
  char *wbuf = reinterpret_cast<char *>(c_ell + 256) + (threadIdx.x >> 6) * kWaveBufBytes;
  double *xbuf = reinterpret_cast<double *>(wbuf);
  half_t *tbuf = reinterpret_cast<half_t *>(wbuf);

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
  for (int e = tid; e < nlt * 64; e += 256) LtF[e] = a.LtF[e];
  if (a.do_tr)
    for (int e = tid; e < NC * nk * 64; e += 256) TtF[e] = a.TtF[e];
  if (tid < 64) {
    const bool in = tid < a.d;
    c_ell[tid] = in ? a.ell_ctr[tid] : 0.0;
    c_lay[tid] = (in && a.do_tr) ? a.lay_ctr[tid] : 0.0;
    c_wrap[tid] = (in && a.do_tr && a.wrap_shift) ? a.wrap_shift[tid] : __longlong_as_double(0x7ff8000000000000ll);
    c_stat[tid] = (in && a.qF) ? a.stats[8 + tid] : 0.0;
  }
  __syncthreads();

  const long long rows_total = quant ? a.nqpad : a.np;
  const long long ntiles = (rows_total + 15) / 16;
  const long long wave_id = (long long)blockIdx.x * 4 + (tid >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const double sigma = quant ? a.stats[0] : 1.0;
  const double namax = quant ? a.stats[1] : 0.0;
  uint4 *qdst = reinterpret_cast<uint4 *>(a.qF);

  // A tile's 16 rows are 16 d contiguous doubles: element e = lane + 64 i goes to LDS row e / d,
  // column e % d (row stride xs2 doubles)
  const int xs2 = a.xstride;                 // doubles per staged row; 2 * xs2 = 4 mod 8 dwords
  const int nelem = 16 * d;
  // elements past the tile (16 d is not a multiple of 64) go to a spare slot behind the rows: the
  // stores need no predicate
  const int xdummy = 16 * xs2 + lane;
  int xoff[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int e = lane + 64 * i;
    const int rr = e / d;
    xoff[i] = e < nelem ? rr * xs2 + (e - rr * d) : xdummy;
  }
  const long long total = a.np * (long long)d;
  auto load_tile = [&](long long tile, double *x) {
    const long long base = tile * 16 * (long long)d;
    if (base + 64 * NK <= total) {   // wave-uniform: the whole 64 NK window is inside the batch
#pragma unroll
      for (int i = 0; i < NK; ++i) x[i] = a.pts[base + lane + 64 * i];
    } else {
#pragma unroll
      for (int i = 0; i < NK; ++i) {   // clamped address + select: no branch around the load
        const long long g = base + lane + 64 * i;
        const bool ok = g < total;
        const double v = a.pts[ok ? g : total - 1];
        x[i] = ok ? v : 0.0;
      }
    }
  };

  // four consecutive tiles (64 proposals) per wave and step: the proposals of a binary16 fragment
  // group (32 rows) stay within one wave.  The next tile's rows are requested before the current
  // tile's matrix products are issued (one wave has ~5000 cycles of MFMA work per tile to hide them).
  double xcur[NK];
  if (wave_id * 4 < ntiles) load_tile(wave_id * 4, xcur);
  for (long long t4 = wave_id; t4 * 4 < ntiles; t4 += nwaves) {
    for (int sub = 0; sub < 4; ++sub) {
      const long long tile = t4 * 4 + sub;
      if (tile >= ntiles) break;
      const long long p = tile * 16 + pl;
      const bool live = p < a.np;
      const double *row = a.pts + (live ? p : 0) * (long long)d;

      // redistribute: coalesced order -> operand order
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NK; ++i) xbuf[xoff[i]] = xcur[i];
      __builtin_amdgcn_wave_barrier();
      double xop[NK];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const int k = 4 * ks + kq;
        // columns below 4 prev(NK) + 1 exist for every d this instance serves
        xop[ks] = (4 * ks + 3 < 4 * prev_ksteps(NK) + 1 || k < d) ? xbuf[pl * xs2 + k] : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      load_tile(sub < 3 ? tile + 1 : (t4 + nwaves) * 4, xcur);   // prefetch

      // ---- operands: 4 coordinates x 16 proposals per k-step -------------------------------
      double dl[NK], dw[NK];
      double nrm2 = 0.0;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        dl[ks] = 0.0;
        dw[ks] = 0.0;
        {
          const int k = 4 * ks + kq;
          const bool ok = 4 * ks + 3 < 4 * prev_ksteps(NK) + 1 || k < d;   // rows past the batch carry zeros and are discarded
          const double x = xop[ks];
          double w = x;
          if (WRAP && ok && a.do_tr) {
            const double sh = c_wrap[k];
            if (sh == sh) {   // NaN marks an unwrapped dimension
              const double xs = w + sh;
              w = (xs >= 0.0 && xs < 2.0) ? (xs >= 1.0 ? xs - 1.0 : xs) : wrap_coordinate3(w, sh);
            }
          }
          dl[ks] = ok ? x - c_ell[k] : 0.0;
          dw[ks] = (ok && a.do_tr) ? w - c_lay[k] : 0.0;
          nrm2 = __builtin_fma(dl[ks], dl[ks], nrm2);
        }
      }
      nrm2 = quad_sum(nrm2);

      // ---- H3 bound: Y = Lt . delta, qt = |Y|^2 ---------------------------------------------
      double qt = 0.0;
      {
        double4v y[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) y[ct] = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          {
#pragma unroll
            for (int ct = 0; ct < NC; ++ct)
              if (ks >= 4 * ct)   // Lt[kb][j] = 0 for j < kb: tiles left of the diagonal are empty and not stored
                y[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(LtF[(size_t)(ct * nk - 2 * ct * (ct - 1) + ks - 4 * ct) * 64 + lane], dl[ks], y[ct], 0, 0, 0);
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
        for (int ks = 0; ks < NK; ++ks) {
          {
#pragma unroll
            for (int ct = 0; ct < NC; ++ct)
              t[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(TtF[((size_t)ct * nk + ks) * 64 + lane], dw[ks], t[ct], 0, 0, 0);
          }
        }
      }
      if (inside) {
        double *tp = a.t_out + p * a.t_ldq + (long long)kq * a.t_ldk;   // coordinate kq; +4 coordinates per step
        const long long step4 = 4 * a.t_ldk;
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            constexpr int kAlways = 4 * prev_ksteps(NK) + 1;   // coordinates below this exist for every d of this instance
            const int cbase = 16 * ct + 4 * r;                   // + kq (0..3)
            if (cbase + 3 < kAlways || cbase + kq < d) *tp = t[ct][r];
            tp += step4;
          }
      }
      if (!quant) continue;

      // ---- binary16 quantisation (columns 0 .. d-1), norms reduced over the proposal's lanes --
      half_t *trow = tbuf + pl * kTileRowHalfs;
      double nb = 0.0, nbn2 = 0.0;
      const int K = KS * 16;
      __builtin_amdgcn_wave_barrier();   // the staged rows have been consumed: the buffer becomes the transpose
      if (inside) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * ct + kq + 4 * r;
            constexpr int kAlwaysQ = 4 * prev_ksteps(NK) + 1;
            if (16 * ct + 4 * r + 3 < kAlwaysQ || c < d) {
              const double x = sigma * (t[ct][r] - c_stat[c]);
              nbn2 = __builtin_fma(x, x, nbn2);   // the binary16 range check is the test of this sum below
              const half_t h = (half_t)(float)x;
              const double hv = (double)(float)h;
              nb += hv * hv;
              trow[c] = (half_t)(-2.0f * (float)h);
            }
          }
      }
      nb = quad_sum(nb);
      nbn2 = quad_sum(nbn2);

      // |x|^2 <= nbn2 <= 30000 bounds every scaled coordinate by 174 (binary16 holds 65504); NaN / inf fail the test
      int rt = inside ? 1 : 0;
      if (rt == 1 && !(nbn2 <= 30000.0)) rt = 2;
      float lo_f = -1.0f, hi_f = -1.0f;
      half_t pc[3] = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
      if (rt == 1) {
        split3(nb, pc);
        if (!filter_thresholds(sigma, namax, nbn2, a.r2, K, &lo_f, &hi_f)) {
          rt = 2;
          lo_f = hi_f = -1.0f;
        }
      }
      // columns d .. K-1 of the row: zeros, except the six norm columns at dp (shared by the four lanes)
      for (int c = d + kq; c < K; c += 4) {
        half_t v = (half_t)0.0f;
        if (c >= a.dp && c < a.dp + 3) v = (half_t)1.0f;                 // x |ah|^2 pieces of the live point
        if (c >= a.dp + 3 && c < a.dp + 6) v = pc[c - a.dp - 3];         // x ones column of the live point
        trow[c] = v;
      }
      __builtin_amdgcn_wave_barrier();
      // 16 proposals x (K / 8) pieces of 16 bytes, contiguous across proposals in the fragment layout;
      // a proposal that is not filtered (gated out / exact scan) gets all-zero operand columns
      {
        const int npieces = 16 * (K >> 3);
        for (int q = lane; q < npieces; q += 64) {
          const int qp = q & 15, c0 = (q >> 4) << 3;
          const long long pp = tile * 16 + qp;
          const int rt_q = __shfl(rt, qp, 64);
          if (pp < a.nqpad) {
            uint4 v = *reinterpret_cast<const uint4 *>(tbuf + qp * kTileRowHalfs + c0);
            if (rt_q != 1) v = make_uint4(0u, 0u, 0u, 0u);
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

int prep3_ksteps(int d) {   // instantiated k-step counts; d is padded up with zero rows / columns
  static const int sizes[] = {1, 2, 3, 4, 5, 6, 8, 10, 13, 16};
  for (int v : sizes)
    if (4 * v >= d) return v;
  return 16;
}

size_t prep3_fragment_count(int d) {
  const int nk = prep3_ksteps(d);
  return (size_t)((nk + 3) / 4) * nk * 64;
}

void prep3_fragments(const double *M, int d, bool transpose, bool upper_only, double *out) {
  const int nk = prep3_ksteps(d), nc = (nk + 3) / 4;
  size_t tile = 0;
  for (int ct = 0; ct < nc; ++ct)
    for (int ks = 0; ks < nk; ++ks) {
      if (upper_only && ks < 4 * ct) continue;   // all k < first row of the tile: empty for an upper factor
      for (int l = 0; l < 64; ++l) {
        const int row = 16 * ct + (l & 15), k = 4 * ks + (l >> 4);
        double v = 0.0;
        if (row < d && k < d) v = transpose ? M[(size_t)k * d + row] : M[(size_t)row * d + k];
        out[tile * 64 + l] = v;
      }
      ++tile;
    }
}

int prep3_lt_tiles(int d) {
  const int nk = prep3_ksteps(d), nc = (nk + 3) / 4;
  int n = 0;
  for (int ct = 0; ct < nc; ++ct) n += nk > 4 * ct ? nk - 4 * ct : 0;
  return n;
}

int prep3_xstride(int d) {   // doubles per staged row: >= d, and 2 * stride = 4 (mod 8) dwords
  int s = d;
  while ((2 * s) % 8 != 4) ++s;
  return s;
}

static size_t prep3_lds_bytes(int d) {
  return ((size_t)prep3_lt_tiles(d) * 64 + prep3_fragment_count(d) + 256) * sizeof(double) + (size_t)4 * kWaveBufBytes;
}

bool prep3_usable(int d) { return d >= 1 && d <= 64; }

hipError_t launch_prep3(const Prep3Args &a_in, hipStream_t s) {
  Prep3Args a = a_in;
  a.nk = prep3_ksteps(a.d);
  a.nlt = prep3_lt_tiles(a.d);
  a.xstride = prep3_xstride(a.d);
  if (a.np <= 0) return hipSuccess;
  if (!prep3_usable(a.d)) return hipErrorInvalidValue;
  const long long rows = a.qF ? a.nqpad : a.np;
  const long long chunks = (rows + 63) / 64;            // 64 proposals per wave and step
  long long grid = (chunks + 3) / 4;
  if (grid > 512) grid = 512;                           // persistent: 2 workgroups per CU
  const size_t lds = prep3_lds_bytes(a.d);
  const bool wrap = a.wrap_shift != nullptr;
#define LAUNCH3(NKV)                                                                                        \
  case NKV: {                                                                                               \
    static DeviceGrant grant;                                                                               \
    if (hipError_t e = grant.ensure([] {                                                                    \
          hipError_t g = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep3<NKV, false>),          \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024);       \
          if (g == hipSuccess)                                                                              \
            g = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep3<NKV, true>),                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024);                \
          return g;                                                                                         \
        }))                                                                                                 \
      return e;                                                                                             \
    if (wrap)                                                                                               \
      hipLaunchKernelGGL((k_prep3<NKV, true>), dim3((unsigned)grid), dim3(256), lds, s, a);                 \
    else                                                                                                    \
      hipLaunchKernelGGL((k_prep3<NKV, false>), dim3((unsigned)grid), dim3(256), lds, s, a);                \
    break;                                                                                                  \
  }
  switch (a.nk) {
    LAUNCH3(1) LAUNCH3(2) LAUNCH3(3) LAUNCH3(4) LAUNCH3(5) LAUNCH3(6) LAUNCH3(8) LAUNCH3(10) LAUNCH3(13) LAUNCH3(16)
    default: return hipErrorInvalidValue;
  }
#undef LAUNCH3
  return hipGetLastError();
}

}  // namespace mlf
