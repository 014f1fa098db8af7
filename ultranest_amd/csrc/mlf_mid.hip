// mlf_mid.hip -- MLFriends.inside (reference mlfriends.pyx:1186-1211) for the batch sizes a real run draws
// (ndraw_min = 128 ... ndraw_max = 65536, integrator.py:1050-1051; the callers: :1776-1804, :1854-1855) in ONE launch:
//   H3 ellipsoid test + T1 whitening on the matrix cores (the bounded per-proposal stage of mlf_prep4.hip, same
//   arithmetic, same error model), the MFMA pre-filter sweep of mlf_sweep.hip with its exact re-check
//   (recheck_segment), the ellipsoid band in binary64, and the answers -- where the batched pipeline takes three
//   dependent launches (k_prep4 -> k_sweep -> k_scan tail: 31-37 us up to 16384 proposals, 62 us at 65536, each launch
//   with its own ramp, and the binary16 operand a round trip through HBM in between).
//
// Workgroup = 8 waves = TWO sets of 4 query groups (256 proposals) x FOUR tile ranges (grid.y workgroups cover the 4
// grid.y ranges of the live tiles):
//   A. wave w runs the per-proposal stage of group 8 bx + w (k_prep4's body for one group: rows -> LDS as they lie,
//      split binary16 operands, L^T delta and T^T delta on v_mfma_f32_32x32x16_f16, bounded ellipsoid decision,
//      thresholds).  Workgroups of the same two sets (blockIdx.y > 0) repeat this stage: 8 groups in ~6 us, cheaper than
//      a hand-off through memory.  The workgroup with blockIdx.y == 0 also decides its band proposals in binary64 on the spot.
//   B. the operand fragments a lane holds in registers after A ARE its 8-column pieces of the filter operand (the row
//      order of T^T, mlf_prep4.hip): each wave drops its group's fragments (KS KiB), thresholds and gate / route bits
//      into its LDS area; barrier; wave w = (set w >> 2, range 4 blockIdx.y + (w & 3)) collects the four groups of its set.
//   C. the sweep of k_sweep over the wave's tile range (per-tile band test, band pairs -> a wave-private list in LDS),
//      then recheck_segment on that list against a wave-private best[] in LDS.
//   D. hand-off (Guideline 16 of the CDNA guide): the wave stores its 128 hit bits (and, range 0, the gate / route bits)
//      with agent-scope atomic stores, drains, and bumps the set's arrival counter; the wave that completes the count
//      acquires, folds the R records and writes the mask bytes of its set -- nothing else writes them.  Proposals that
//      the pre-filter cannot take (route 2: NaN / out of binary16 range) or whose list overflowed get route[p] = 2 and
//      the batch's scan flag: the (otherwise idle) exact-scan launch behind this one takes them.
// All shared words are agent-scope atomics; the arrival counters return to zero by themselves.
#include "mlf_filter.hpp"
#include "mlf_filter_dev.hpp"
#include "mlf_recheck_dev.hpp"
#include "mlf_prep4.hpp"

#include <math.h>

#include <atomic>

namespace mlf {

typedef float float16v __attribute__((ext_vector_type(16)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

namespace {

template <int DP>
struct M4 {
  static constexpr int NS = (DP + 15) / 16;
  static constexpr int KS = (DP + 6 + 15) / 16;
  static constexpr int NT = (KS + 1) / 2;
  static constexpr int NE = (DP + 31) / 32;
  static constexpr int KMIN = DP <= 32 ? DP - 1 : (DP == 50 ? 49 : DP - 3);
  static constexpr int NLT = NS + (NE > 1 ? NS - 2 : 0);
  static constexpr int NPC = (DP + 3) / 4;
  static constexpr size_t r16(size_t x) { return (x + 15) / 16 * 16; }
  static constexpr size_t STAGE = (size_t)NPC * 1024 + 512;                 // the group's rows as they lie
  static constexpr size_t XCH = (size_t)KS * 1024 + 2 * 32 * 4 + 16;        // fragments, thresholds, gate / route bits
  static constexpr size_t kListCap = 128;                                   // band pairs of one (set, range)
  static constexpr size_t RCK = ((size_t)kTQ * ((DP + 1) | 1) + kTQ * 64) * 8 + 3 * 64 * 4;   // = recheck_w_lds(DP)
  static constexpr size_t TAIL = r16(RCK) + kListCap * 8 + 128 * 4;        // re-check scratch, list, best[]
  static constexpr size_t WAVE = r16(STAGE > XCH ? (STAGE > TAIL ? STAGE : TAIL) : (XCH > TAIL ? XCH : TAIL));
  static constexpr size_t FRAG = (size_t)(2 * NT * NS + 2 * NLT) * 1024 + (size_t)(32 * NE + 32 * NT) * 4 + (size_t)(16 * NS) * 8;
  // the layer matrix in binary64 for the re-check, [DP][DP], behind the wave areas -- where the 160 KB allow it (d <= 52)
  static constexpr size_t TLB = (size_t)DP * DP * 8;
  static constexpr bool TL_OK = r16(FRAG) + 8 * WAVE + r16(TLB) <= 160 * 1024;
  static constexpr size_t LDS = r16(FRAG) + 8 * WAVE + (TL_OK ? r16(TLB) : 0);
};

// recheck_segment (mlf_recheck_dev.hpp) for the one-launch path, where the re-check is the longest stage of a short kernel and
// all latency (round 4 stamps at 300 proposals: 12 of 26 us -- five dependent round trips for a handful of pairs: the
// queries' rows, two batches of matrix rows, two batches of live-point coordinates).  Same arithmetic, statement by
// statement; what differs is where the operands come from and when they are asked for:
//   * the layer matrix is read from LDS (tl: put there by the whole workgroup in the kernel's prologue) -- no round trip;
//   * a lane's live-point row is requested in full BEFORE the whitening (it depends on the list entry alone) and waits in
//     registers: one round trip, shared with the gather of the queries' rows.
template <int DP, bool TL>
__device__ __forceinline__ void recheck_mid(const RecheckWArgs &a, const double *tl, const unsigned long long *seg, unsigned count,
                                            double *lds_r, int lane) {
  if (count == 0) return;
  const int d = a.d;
  const int ds = (d + 1) | 1;
  double *tq = lds_r;                                          // [kTQ][ds]
  double *dlw = tq + kTQ * ds;                                 // [kTQ][64]
  int *hkey = reinterpret_cast<int *>(dlw + kTQ * 64);         // [64]
  int *hid = hkey + 64;
  int *qlist = hid + 64;
  const bool inrow = lane < d;
  const double myctr = inrow ? a.lay_ctr[lane] : 0.0;
  for (unsigned e0 = 0; e0 < count; e0 += kChunk) {
    __builtin_amdgcn_wave_barrier();
    hkey[lane] = -1;
    __builtin_amdgcn_wave_barrier();
    const unsigned e = e0 + (unsigned)lane;
    long long qi = -1;
    int i = 0;
    bool livee = false;
    unsigned h = 0;
    if (lane < kChunk && e < count) {
      const unsigned long long ent = seg[e];
      qi = (long long)(ent >> 32);
      i = (int)(ent & 0xffffffffu);
      livee = i < a.n && qi < a.nq && a.best[qi] > i;
    }
    // the live point of this lane's entry: all of it, now
    double2 av[DP / 2];
    {
      const double2 *ar = reinterpret_cast<const double2 *>(a.refR + (size_t)(livee ? i : 0) * a.dp);
#pragma unroll
      for (int k2 = 0; k2 < DP / 2; ++k2) av[k2] = ar[k2];
    }
    if (livee) {
      h = ((unsigned)qi * 2654435761u) & 63u;
      while (true) {
        const int old = atomicCAS(&hkey[h], -1, (int)qi);
        if (old == -1 || old == (int)qi) break;
        h = (h + 1u) & 63u;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const bool occ = hkey[lane] != -1;
    const unsigned long long bm = __ballot(occ);
    const unsigned nqb = (unsigned)__popcll(bm);
    if (occ) {
      const unsigned id = (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
      hid[lane] = (int)id;
      qlist[id] = hkey[lane];
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned myid = livee ? (unsigned)hid[h] : 0xffffffffu;
    const double *tg = a.T64 + lane;
    const double *tp = tl + (lane < DP ? lane : 0);
    for (unsigned r0 = 0; r0 < nqb; r0 += kTQ) {
      const unsigned nr = nqb - r0 < (unsigned)kTQ ? nqb - r0 : (unsigned)kTQ;   // wave-uniform
      // the centred rows stay in REGISTERS, lane l holding coordinate l of each (one round trip for all of them); term k of
      // the chain reaches the lanes through v_readlane (a scalar operand of the fused multiply-add) -- through LDS every one
      // of the 50 dependent steps waited for its reads (9.7 of the re-check's 21 thousand cycles at 300 proposals)
      double dq[kTQ];
#pragma unroll
      for (int t = 0; t < kTQ; ++t) dq[t] = ((unsigned)t < nr && inrow) ? a.pts[(long long)qlist[r0 + t] * d + lane] - myctr : 0.0;
      double acc[kTQ];
#pragma unroll
      for (int t = 0; t < kTQ; ++t) acc[t] = 0.0;
      auto bcast = [&](double v, int k) __attribute__((always_inline)) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      };
      auto chains = [&](auto nq_c) __attribute__((always_inline)) {
        constexpr int NQ = decltype(nq_c)::value;
        constexpr int KB = 10;   // matrix rows requested together
        for (int k0 = 0; k0 < d; k0 += KB) {
          double tk[KB];
#pragma unroll
          for (int q = 0; q < KB; ++q) {
            const int k = k0 + q < d ? k0 + q : d - 1;
            tk[q] = TL ? tp[k * DP] : tg[k * 64];
          }
#pragma unroll
          for (int q = 0; q < KB; ++q) {
            if (k0 + q < d) {
#pragma unroll
              for (int t = 0; t < NQ; ++t) acc[t] = __builtin_fma(bcast(dq[t], k0 + q), tk[q], acc[t]);
            }
          }
        }
      };
      if (nr <= 2u) {
        chains(std::integral_constant<int, 2>{});
      } else if (nr <= 4u) {
        chains(std::integral_constant<int, 4>{});
      } else {
        chains(std::integral_constant<int, kTQ>{});
      }
#pragma unroll
      for (int t = 0; t < kTQ; ++t)
        if ((unsigned)t < nr && inrow) tq[t * ds + lane] = acc[t];
      __builtin_amdgcn_wave_barrier();
      if (livee && myid >= r0 && myid < r0 + kTQ && a.best[qi] > i) {
        const double *br = tq + (myid - r0) * ds;
        double accd = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < DP / 2; ++k2) {   // the reference's loop: sub, mul, add, each rounded, k ascending
          if (2 * k2 < d) {
            const double d0 = av[k2].x - br[2 * k2];
            accd += d0 * d0;
          }
          if (2 * k2 + 1 < d) {
            const double d1 = av[k2].y - br[2 * k2 + 1];
            accd += d1 * d1;
          }
        }
        if (accd <= a.r2) atomicMin(&a.best[qi], i);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__host__ __device__ inline int m4_column(int t, int i) {
  return 32 * t + 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3);
}
__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ int min3i(int a, int b, int c) {
  const int m = a < b ? a : b;
  return m < c ? m : c;
}
constexpr int kPosInf = 0x7f800000;
__device__ __forceinline__ int tree_min(const float16v &c, int run) {
  const int m0 = min3i(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]));
  const int m1 = min3i(__float_as_int(c[3]), __float_as_int(c[4]), __float_as_int(c[5]));
  const int m2 = min3i(__float_as_int(c[6]), __float_as_int(c[7]), __float_as_int(c[8]));
  const int m3 = min3i(__float_as_int(c[9]), __float_as_int(c[10]), __float_as_int(c[11]));
  const int m4 = min3i(__float_as_int(c[12]), __float_as_int(c[13]), __float_as_int(c[14]));
  return min3i(min3i(m0, m1, m2), min3i(m3, m4, __float_as_int(c[15])), run);
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define MLF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// one band proposal decided in binary64 by the whole wave (ell_exact_wave's arithmetic for one proposal)
__device__ __forceinline__ bool ell_exact_one(const MidArgs &a, long long p, double *dls, int lane) {
  const int d = a.d;
  const bool own = lane < d;
  const double *row = a.pts + p * (long long)d;
  const double dl = own ? row[lane] - a.ell_ctr[lane] : 0.0;
  __builtin_amdgcn_wave_barrier();
  dls[lane] = dl;
  __builtin_amdgcn_wave_barrier();
  const double *lcol = a.ell_L + (own ? lane : 0);
  double y = 0.0;
#pragma unroll 10
  for (int j = 0; j < d; ++j) y = __builtin_fma(lcol[(size_t)j * a.dp], dls[j], y);
  if (!own) y = 0.0;
  double qt = y * y, nrm2 = dl * dl;
  for (int o = 32; o > 0; o >>= 1) {
    qt += __shfl_xor(qt, o, 64);
    nrm2 += __shfl_xor(nrm2, o, 64);
  }
  const double eps = a.ell_eps_scale * nrm2;
  if (a.chol_ok && qt + eps < a.enlarge) return true;
  if (a.chol_ok && qt - eps > a.enlarge) return false;
  double acc = 0.0;
  if (lane == 0) {   // the reference's own order: one accumulator, j outer, (d_j A_jk) d_k, no FMA (mlfriends.pyx:882-912)
    for (int j = 0; j < d; ++j) {
      const double dj = row[j] - a.ell_ctr[j];
      const double *arow = a.ell_A + (size_t)j * a.dp;
      for (int k = 0; k < d; ++k) acc += (dj * arow[k]) * (row[k] - a.ell_ctr[k]);
    }
  }
  acc = __shfl(acc, 0, 64);
  return acc <= a.enlarge;
}

}  // namespace

template <int DP>
__global__ __launch_bounds__(512, 1) void k_inside_mid(MidArgs a) {
  using C = M4<DP>;
  constexpr int NS = C::NS, NT = C::NT, NE = C::NE, KS = C::KS, QW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsm[];
  const uint4 *Th = reinterpret_cast<const uint4 *>(ldsm);
  const uint4 *Tl = Th + NT * NS * 64;
  const uint4 *Lh = Tl + NT * NS * 64;
  const uint4 *Ll = Lh + C::NLT * 64;
  float *y0l = reinterpret_cast<float *>(const_cast<uint4 *>(Ll + C::NLT * 64));
  float *csl = y0l + 32 * NE;
  double *ctrl = reinterpret_cast<double *>(csl + 32 * NT);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int p32 = lane & 31, h = lane >> 5;
  // diagnostics: shader-clock stamps of wave 0 of workgroup (0, 0) at the stage boundaries (mlf_region_debug_stats)
  unsigned nstamp = 0u;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nstamp < 8u) a.stamps[nstamp] = (unsigned)__builtin_readcyclecounter();
    ++nstamp;
  };
  stamp();
  unsigned char *area0 = ldsm + C::r16(C::FRAG);
  unsigned char *area = area0 + (size_t)wv * C::WAVE;
  double *xs = reinterpret_cast<double *>(area);
  const int d = a.d;
  constexpr float up = 1.0f + 0x1p-18f, dn = 1.0f - 0x1p-18f;

  const double sigma = a.stats[0];
  const bool sig_ok = sigma >= 0x1p-60 && sigma <= 0x1p60;
  const float sig_f = sig_ok ? (float)sigma : 1.0f;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {   // nobody else touches these in this path
    a.counters[0] = 0;
    a.counters[1] = 0;
  }
  {
    uint4 *dst = reinterpret_cast<uint4 *>(ldsm);
    const uint4 *srcT = reinterpret_cast<const uint4 *>(a.TtF);
    const uint4 *srcL = reinterpret_cast<const uint4 *>(a.LtF);
    for (int e = tid; e < 2 * NT * NS * 64; e += 512) dst[e] = srcT[e];
    for (int e = tid; e < 2 * C::NLT * 64; e += 512) dst[2 * NT * NS * 64 + e] = srcL[e];
  }
  double *tlds = reinterpret_cast<double *>(ldsm + C::r16(C::FRAG) + 8 * C::WAVE);   // [DP][DP] (TL_OK)
  if constexpr (C::TL_OK) {
    for (int e = tid; e < DP * DP; e += 512) tlds[e] = a.T64[(e / DP) * 64 + (e % DP)];
  }
  if (tid < 32 * NE) y0l[tid] = a.y0[tid];
  if (tid < 32 * NT) {
    const int col = m4_column(tid >> 5, tid & 31);
    csl[tid] = col < DP ? 2.0f * (float)(sigma * a.stats[8 + col]) : 0.0f;
  }
  if (tid < 16 * NS) ctrl[tid] = tid < d ? (double)a.c.s_x * a.lay_ctr[tid] : 0.0;

  const long long ngroups = (a.np + 31) / 32;
  const long long grp = (long long)blockIdx.x * 8 + wv;   // the group this wave prepares
  // ---- A0. this group's rows set out for LDS (as in k_prep4: 16-byte pieces, no staging registers)
  const long long total = a.np * (long long)d;
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  if (grp < ngroups) {
    const long long base = grp * 32 * (long long)d;
    if (base + 128 * C::NPC <= total) {
#pragma unroll
      for (int i = 0; i < C::NPC; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t *)(a.pts + base + 2 * (lane + 64 * i)), (lptr_t *)(area + i * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < C::NPC; ++i) {
        const int e = lane + 64 * i;
        const long long g = base + 2 * e;
        if (e < 16 * d && g + 1 < total)
          __builtin_amdgcn_global_load_lds((gptr_t *)(a.pts + g), (lptr_t *)(area + i * 1024), 16, 0, 0);
        else if (e < 16 * d && g + 1 == total)
          xs[2 * e] = a.pts[g];
      }
    }
  }
  __syncthreads();   // fragments and constants are in LDS
  stamp();

  // uniform quantities (binary32, rounded outward) -- k_prep4's
  const float sqrt_k = __builtin_sqrtf((float)(16 * KS));
  const float namax = (float)a.stats[1] * up;
  float cs2 = 0.0f;
  for (int e = 0; e < 32 * NT; ++e) cs2 = __builtin_fmaf(csl[e], csl[e], cs2);
  const float csn = 0.5f * __builtin_sqrtf(cs2) * up;
  const float zeta_scale = sig_f * a.c.zt * up;
  const float zeta0 = (a.c.g_chain * csn + sig_f * a.c.zt_abs) * up + 0x1p-100f;
  const float kappa = -2.0f * sig_f * a.c.inv_st_sx;
  const double sr = sigma * sqrt(a.r2);
  const float sr_lo = (float)(sr * (1.0 - 0x1p-30)) * dn;
  const float sr_hi = (float)(sr * (1.0 + 0x1p-30)) * up;
  const float inv_sx2 = a.c.inv_sx * a.c.inv_sx, inv_slsx2 = a.c.inv_sl_sx * a.c.inv_sl_sx;

  // ---- A. the per-proposal stage of this wave's group
  unsigned pk[KS * 4];
#pragma unroll
  for (int i = 0; i < KS * 4; ++i) pk[i] = 0u;
  float lo_f = -1.0f, hi_f = -1.0f;
  unsigned gate_bits = 0u, rt2_bits = 0u;   // valid in every lane after the ballots below
  if (grp < ngroups) {   // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    half8v hia[NS], loa[NS];
    float dn2a = 0.0f;
    {
      const double sxd = (double)a.c.s_x;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          float x32[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int jj = 2 * j2 + e;
            const int k = 16 * s + 8 * h + jj;
            const bool ok = 16 * s + 8 + jj < C::KMIN || k < d;
            const double xv = __builtin_fma(xs[p32 * d + k], sxd, -ctrl[k < 16 * NS ? k : 0]);
            x32[e] = ok ? (float)xv : 0.0f;
            dn2a = __builtin_fmaf(x32[e], x32[e], dn2a);
          }
          const half2v hp = __builtin_convertvector((float2v){x32[0], x32[1]}, half2v);
          const float2v res = {x32[0] - (float)hp[0], x32[1] - (float)hp[1]};
          const half2v lp = __builtin_convertvector(res, half2v);
          hia[s][2 * j2] = hp[0];
          hia[s][2 * j2 + 1] = hp[1];
          loa[s][2 * j2] = lp[0];
          loa[s][2 * j2 + 1] = lp[1];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    auto frag = [&](const uint4 *base, int idx) __attribute__((always_inline)) {
      union { uint4 u; half8v h8; } cv;
      cv.u = base[idx * 64 + lane];
      return cv.h8;
    };
    const long long p = grp * 32 + p32;
    const bool live = p < a.np;
    float qs = 0.0f;
    {
      float16v ye[NE];
#pragma unroll
      for (int t = 0; t < NE; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ye[t][r] = y0l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
      for (int t = 0; t < NE; ++t)
#pragma unroll
        for (int s = 2 * t; s < NS; ++s) {
          const int f = (t ? NS : 0) + s - 2 * t;
          const half8v lh = frag(Lh, f), ll = frag(Ll, f);
          ye[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh, hia[s], ye[t], 0, 0, 0);
          ye[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh, loa[s], ye[t], 0, 0, 0);
          ye[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ll, hia[s], ye[t], 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < NE; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) qs = __builtin_fmaf(ye[t][r], ye[t][r], qs);
    }
    float16v tt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tt[t] = (float16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const half8v th = frag(Th, t * NS + s), tl = frag(Tl, t * NS + s);
        tt[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, hia[s], tt[t], 0, 0, 0);
        tt[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, loa[s], tt[t], 0, 0, 0);
        tt[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, hia[s], tt[t], 0, 0, 0);
      }
    }
    qs = half_sum(qs) * inv_slsx2;
    const float dn2 = half_sum(dn2a) * inv_sx2;
    bool sure_in = false, sure_out = false;
    float dnorm = 0.0f;
    {
      const bool finite = qs < 3.0e38f && dn2 < 3.0e38f;
      const float sq = __builtin_sqrtf(qs);
      dnorm = __builtin_sqrtf(dn2) * up + 0x1p-100f;
      const float eta = (a.c.g_chain * (a.c.y0n + a.c.lf * dnorm) + a.c.el * dnorm + a.c.l_abs) * up;
      const float de = dnorm + a.c.s0n;
      const float eps = a.c.eps_scale * (de * de) * up;
      const float hi = sq * up + eta;
      const float qhi = ((hi * hi) * up + eps) * up;
      const float lo = (sq * dn - eta) * dn;
      const float qlo = ((lo * lo) * dn - eps * up) * dn;
      sure_in = finite && qhi < a.c.enl_lo;
      sure_out = finite && lo > 0.0f && qlo > a.c.enl_hi;
    }
    const bool band = live && !sure_in && !sure_out;
    const bool ins_any = live && !sure_out;   // band proposals: provisionally inside for the sweep
    float nbq = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float v0 = __builtin_fmaf(tt[t][2 * m], kappa, csl[32 * t + ((2 * m) & 3) + 8 * ((2 * m) >> 2) + 4 * h]);
        const float v1 = __builtin_fmaf(tt[t][2 * m + 1], kappa, csl[32 * t + ((2 * m + 1) & 3) + 8 * ((2 * m + 1) >> 2) + 4 * h]);
        union { half2v v; unsigned u; } cv;
        cv.v = __builtin_convertvector((float2v){v0, v1}, half2v);
        const float f0 = (float)cv.v[0], f1 = (float)cv.v[1];
        nbq = __builtin_fmaf(f0, f0, nbq);
        nbq = __builtin_fmaf(f1, f1, nbq);
        if (t * 8 + m < KS * 4) pk[t * 8 + m] = cv.u;
      }
    const float nb = 0.25f * half_sum(nbq);
    int rt = ins_any ? 1 : 0;
    if (rt == 1 && (!sig_ok || !(nb <= 29000.0f))) rt = 2;
    if (rt == 1) {
      const float zeta = (zeta_scale * dnorm + zeta0) * up;
      if (!filter_thresholds4(namax, nb, zeta, sqrt_k, sr_lo, sr_hi, &lo_f, &hi_f)) {
        rt = 2;
        lo_f = hi_f = -1.0f;
      }
    }
    const _Float16 p1 = (_Float16)nb;
    const float r1 = nb - (float)p1;
    const _Float16 p2 = (_Float16)r1;
    const float r2 = r1 - (float)p2;
    const _Float16 p3 = (_Float16)r2;
    const _Float16 one = (_Float16)1.0f;
    union { half2v v; unsigned u; } sp[3];
    sp[0].v = (half2v){one, one};
    sp[1].v = (half2v){one, p1};
    sp[2].v = (half2v){p2, p3};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = DP + 2 * q;
      if (h == ((c >> 3) & 1)) pk[4 * (c >> 4) + ((c & 7) >> 1)] = sp[q].u;
    }
    if (rt != 1) {
#pragma unroll
      for (int i = 0; i < KS * 4; ++i) pk[i] = 0u;
    }
    // the ellipsoid answer: exact for the band proposals (only the workgroup of range quad 0 publishes it)
    bool gate = ins_any;
    if (blockIdx.y == 0) {
      unsigned long long bm = __ballot(band && h == 0);
      while (bm != 0ull) {   // wave-uniform, rare
        const int l = __builtin_ctzll(bm);
        bm &= bm - 1ull;
        const bool in = ell_exact_one(a, grp * 32 + l, xs, lane);   // the rows of the group are no longer needed
        if (lane == l || lane == l + 32) gate = in;
      }
    }
    gate_bits = (unsigned)__ballot(gate && h == 0);
    rt2_bits = (unsigned)__ballot(gate && rt == 2 && h == 0);
  }
  stamp();
  // ---- B. exchange: fragments, thresholds, gate / route bits of this wave's group -> its LDS area
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  {
    uint4 *fx = reinterpret_cast<uint4 *>(area);
#pragma unroll
    for (int s = 0; s < KS; ++s) fx[s * 64 + lane] = make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]);
    float *tx = reinterpret_cast<float *>(area + KS * 1024);
    if (h == 0) {
      tx[p32] = lo_f;
      tx[32 + p32] = hi_f;
    }
    if (lane == 0) {
      unsigned *bx2 = reinterpret_cast<unsigned *>(area + KS * 1024 + 256);
      bx2[0] = gate_bits;
      bx2[1] = rt2_bits;
    }
  }
  __syncthreads();
  const int set_local = wv >> 2, rr = wv & 3;
  const long long set = (long long)blockIdx.x * 2 + set_local;
  const long long nsets = (ngroups + QW - 1) / QW;
  half8v bq[QW][KS];
  float tlo[QW], thi[QW];
  unsigned gbits[QW], rbits[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const unsigned char *src = area0 + (size_t)(4 * set_local + g) * C::WAVE;
    const uint4 *fx = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      union { uint4 u; half8v h8; } cv;
      cv.u = fx[s * 64 + lane];
      bq[g][s] = cv.h8;
    }
    const float *tx = reinterpret_cast<const float *>(src + KS * 1024);
    tlo[g] = tx[p32];
    thi[g] = tx[32 + p32];
    const unsigned *bx2 = reinterpret_cast<const unsigned *>(src + KS * 1024 + 256);
    gbits[g] = bx2[0];
    rbits[g] = bx2[1];
  }
  __syncthreads();   // everybody has collected its set: the areas are scratch from here on
  stamp();
  if (set >= nsets) return;

  // ---- C. sweep of this wave's tile range, band pairs into the wave's LDS list
  double *lds_r = reinterpret_cast<double *>(area);                                            // recheck_w_lds(d) bytes
  unsigned long long *plist = reinterpret_cast<unsigned long long *>(area + C::r16(C::RCK));   // [kListCap]
  int *lbest = reinterpret_cast<int *>(plist + C::kListCap);                                   // [128]
  lbest[lane] = kNone;
  lbest[64 + lane] = kNone;
  const int R = 4 * (int)gridDim.y;
  const int range = 4 * (int)blockIdx.y + rr;
  const int tile0 = (int)((long long)a.ntiles32 * range / R), tile1 = (int)((long long)a.ntiles32 * (range + 1) / R);
  const int ntl = tile1 - tile0;
  constexpr int kTileBytes = KS * 1024;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.refF), 0, a.ntiles32 * kTileBytes, 0x00020000);
  const int voff = lane * 16;
  const int rowbase = 4 * (lane >> 5);
  const long long g0 = set * QW;
  unsigned cursor = 0u;
  int run[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) run[g] = kPosInf;
  {
    float16v acc[QW];
    unsigned long long lo_m[QW], hi_m[QW];
    auto load_tile = [&](half8v(&A)[KS], int t) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        union { uint4v u; half8v h8; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, t * kTileBytes + s * 1024, 0);
        A[s] = c.h8;
      }
    };
    auto mm = [&](const half8v(&A)[KS], int ga, int gb) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[ga] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[ga][s], s == 0 ? z : acc[ga], 0, 0, 0);
        acc[gb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[gb][s], s == 0 ? z : acc[gb], 0, 0, 0);
      }
    };
    auto reduce = [&](int g) __attribute__((always_inline)) {
      run[g] = tree_min(acc[g], run[g]);
      const float rm = __int_as_float(run[g]);
      lo_m[g] = __ballot(rm <= tlo[g]);
      hi_m[g] = __ballot(rm <= thi[g]);
    };
    auto list_band = [&](int g, unsigned long long candm, int t) __attribute__((always_inline)) {
      const float16v &c = acc[g];
      const bool flagged = (candm >> lane) & 1ull;
      unsigned bits = 0u;
      if (flagged) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = c[r];
          bits |= (!(v <= tlo[g]) && (v <= thi[g])) ? (1u << r) : 0u;
        }
        run[g] = kPosInf;
      }
      const unsigned cnt = (unsigned)__popc(bits);
      unsigned mybase = 0u;
      for (unsigned long long m = candm; m != 0ull; m &= m - 1ull) {
        const int l = __builtin_ctzll(m);
        const unsigned cl = (unsigned)__builtin_amdgcn_readlane((int)cnt, l);
        if (lane == l) mybase = cursor;
        cursor += cl;
      }
      const long long qi = (g0 + g) * 32 + (lane & 31);
      while (bits != 0u) {
        const int r = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (mybase < C::kListCap) plist[mybase] = ((unsigned long long)qi << 32) | (unsigned)(t * 32 + rowbase + (r & 3) + 8 * (r >> 2));
        ++mybase;
      }
    };
    auto tile = [&](const half8v(&A)[KS], int t) __attribute__((always_inline)) {
      mm(A, 0, 1);
      mm(A, 2, 3);
#pragma unroll
      for (int g = 0; g < QW; ++g) reduce(g);
      unsigned long long need = 0ull;
#pragma unroll
      for (int g = 0; g < QW; ++g) need |= hi_m[g] & ~lo_m[g];
      if (need != 0ull) {
#pragma unroll
        for (int g = 0; g < QW; ++g) {
          const unsigned long long candm = hi_m[g] & ~lo_m[g];
          if (candm != 0ull) list_band(g, candm, t);
        }
      }
    };
    half8v A0[KS], A1[KS], A2[KS];
    if (ntl > 0) {
      auto req = [&](half8v(&A)[KS], int t) __attribute__((always_inline)) { load_tile(A, t < tile1 ? t : tile1 - 1); };
      req(A0, tile0);
      req(A1, tile0 + 1);
      for (int t = tile0; t < tile1; t += 3) {
        req(A2, t + 2);
        tile(A0, t);
        if (t + 1 >= tile1) break;
        req(A0, t + 3);
        tile(A1, t + 1);
        if (t + 2 >= tile1) break;
        req(A1, t + 4);
        tile(A2, t + 2);
      }
    }
  }
  stamp();
  // certain hits -> the wave's best[], then the exact re-check of the listed pairs
  bool ovf = cursor > (unsigned)C::kListCap;
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int mine = __int_as_float(run[g]) <= tlo[g] ? 1 : 0;
    const int any = mine | __shfl_xor(mine, 32);
    if (lane < 32 && any && thi[g] > 0.0f) lbest[g * 32 + lane] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (cursor != 0u && !ovf) {
    RecheckWArgs rw{};
    rw.refR = a.refR;
    rw.n = a.n;
    rw.d = d;
    rw.dp = a.dp;
    rw.pts = a.pts;
    rw.nq = a.np;
    rw.lay_ctr = a.lay_ctr;
    rw.T64 = a.T64;
    rw.r2 = a.r2;
    rw.best = lbest - set * 128;   // best[query] of the wave's own 128 queries
    recheck_mid<DP, C::TL_OK>(rw, tlds, plist, cursor, lds_r, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  unsigned hb[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) hb[g] = (unsigned)__ballot(lane < 32 && lbest[g * 32 + (lane & 31)] != kNone);

  stamp();
  // ---- D. hand-off: record -> arrival -> (last) answers of the set
  gu64 *rec = (gu64 *)a.rec + ((size_t)set * R + range) * 3;
  if (lane == 0) {
    __hip_atomic_store(rec + 0, ((unsigned long long)hb[1] << 32) | hb[0], MLF_RLX_AGENT);
    __hip_atomic_store(rec + 1, ((unsigned long long)hb[3] << 32) | hb[2], MLF_RLX_AGENT);
    __hip_atomic_store(rec + 2, ovf ? 1ull : 0ull, MLF_RLX_AGENT);
    if (range == 0) {
      gu64 *meta = (gu64 *)a.meta + (size_t)set * 4;
#pragma unroll
      for (int g = 0; g < QW; ++g) __hip_atomic_store(meta + g, ((unsigned long long)rbits[g] << 32) | gbits[g], MLF_RLX_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned arrived = 0u;
  if (lane == 0) arrived = __hip_atomic_fetch_add((gu32 *)a.arrive + set, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
  stamp();
  if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.stamps[7] = cursor;   // band pairs of this wave
  if (arrived != (unsigned)R - 1u) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  unsigned long long r0 = 0ull, r1 = 0ull, r2w = 0ull;
  if (lane < R) {
    gu64 *rc = (gu64 *)a.rec + ((size_t)set * R + lane) * 3;
    r0 = __hip_atomic_load(rc + 0, MLF_RLX_AGENT);
    r1 = __hip_atomic_load(rc + 1, MLF_RLX_AGENT);
    r2w = __hip_atomic_load(rc + 2, MLF_RLX_AGENT);
  }
  for (int o = 32; o > 0; o >>= 1) {
    r0 |= __shfl_xor(r0, o, 64);
    r1 |= __shfl_xor(r1, o, 64);
    r2w |= __shfl_xor(r2w, o, 64);
  }
  unsigned long long mg = 0ull;
  if (lane < QW) mg = __hip_atomic_load((gu64 *)a.meta + (size_t)set * 4 + lane, MLF_RLX_AGENT);
  const bool set_ovf = r2w != 0ull;
  bool any_scan = false;
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const unsigned long long m = __shfl(mg, g, 64);
    const unsigned gb = (unsigned)m, rb = (unsigned)(m >> 32);
    const unsigned hits = g == 0 ? (unsigned)r0 : (g == 1 ? (unsigned)(r0 >> 32) : (g == 2 ? (unsigned)r1 : (unsigned)(r1 >> 32)));
    const long long q = (g0 + g) * 32 + (lane & 31);
    if (lane < 32 && q < a.np) {
      const bool gate = (gb >> lane) & 1u;
      const bool scan = gate && (((rb >> lane) & 1u) || set_ovf);   // route 2, or the set's list overflowed: the exact scan's
      a.route[q] = scan ? 2 : 0;
      if (!scan) a.mask[q] = (gate && ((hits >> lane) & 1u)) ? 1 : 0;
      any_scan |= scan;
    }
  }
  if (__any(any_scan) && lane == 0) *a.scan_flag = 1u;
  if (lane == 0) __hip_atomic_store((gu32 *)a.arrive + set, 0u, MLF_RLX_AGENT);   // ready for the next batch
}

// tile-range quads (grid.y) of a batch of `ngroups` query groups: aim at one workgroup per CU, at least 4 tiles per range,
// at most 32 ranges (the fold is one lane per range)
int mid_range_quads(long long ngroups, int ntiles32) {
  const long long gx = (ngroups + 7) / 8;
  long long ny = 256 / (gx > 0 ? gx : 1);
  if (ny > 8) ny = 8;
  if (ny > ntiles32 / 16) ny = ntiles32 / 16;
  return ny < 1 ? 1 : (int)ny;
}

bool mid_usable(int dp) { return dp >= 2 && dp <= 56 && (dp & 1) == 0; }

size_t mid_lds_bytes(int dp) {
  switch (dp) {
#define X(D) \
  case D:    \
    return M4<D>::LDS;
    MLF_FOR_EACH_DP_MID(X)
#undef X
    default:
      return 0;
  }
}

template <int D>
static hipError_t launch_inside_mid_t(const MidArgs &a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = M4<D>::LDS;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static DeviceGrant grant;   // the attribute belongs to (function, device): set once per device
  if (hipError_t e = grant.ensure([] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_inside_mid<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }))
    return e;
  if (a.ks != M4<D>::KS) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_inside_mid<D>), grid, dim3(512), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_inside_mid(const MidArgs &a, int ny, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  const long long ngroups = (a.np + 31) / 32;
  const dim3 grid((unsigned)((ngroups + 7) / 8), (unsigned)ny);
  switch (a.dp) {
#define X(D) \
  case D:    \
    return launch_inside_mid_t<D>(a, grid, s);
    MLF_FOR_EACH_DP_MID(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace mlf
