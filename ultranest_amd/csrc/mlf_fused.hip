// mlf_fused.hip -- the per-proposal stage of MLFriends.inside (H3 ellipsoid test + T1 whitening, mlfriends.pyx:882-912,
// 737-743: k_prep4's bounded split-binary16 form, mlf_prep4.hip) and the FIRST live-point range of the min-only sweep
// (k_sweep_min, mlf_sweepmin.hip) in ONE launch for the large batches.
//
// k_prep4 is HBM bound (reads 8 d bytes per proposal, 0.115 ms per 10^6 x 50 at 4.7 TB/s) and leaves the matrix cores 87 %
// idle; k_sweep_min is matrix bound and leaves HBM idle; between them the binary16 operand makes a 128 MB write + 128 MB
// read round trip (per-step traffic 2.2x the algorithmic bytes).  After the per-proposal stage of a query group a lane
// HOLDS its 8-column pieces of the filter operand (the row order of T^T), so a wave that prepares its own four groups can
// sweep them straight from its registers: no operand array, no thresholds array, and the waves that are fetching rows run
// next to the ones that are multiplying.
//   workgroup = 8 waves (one per CU: 28 KiB of matrix fragments shared + a 13.5 KiB row buffer per wave), wave = one set of
//   4 query groups: group g's rows travel to LDS while group g - 1 is processed (k_prep4's pipeline), its fragments and
//   thresholds stay in registers; then k_sweep_min's tile loop over the first range and its epilogue (certain hits -> best,
//   the rest compacted with their minima, one atomic per workgroup).
// Everything k_prep4 wrote for the later stages is written here too: gate, route, best = none, the ellipsoid band list.
// (Round 2 measured this fusion with k_filter's loop: 395 against 112 + 273 us; with the min-only loop see DESIGN 4f.)
#include "mlf_filter.hpp"
#include "mlf_dpp_dev.hpp"
#include "mlf_filter_dev.hpp"
#include "mlf_prep4.hpp"

#include <math.h>

namespace mlf {

typedef float float16v __attribute__((ext_vector_type(16)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

#ifndef MLF_OPERAND_STEPS
#define MLF_OPERAND_STEPS 4
#endif
namespace {

template <int DP, int NW = 8>
struct F4 {
  static constexpr int NS = (DP + 15) / 16;
  static constexpr int KS = (DP + 6 + 15) / 16;
  static constexpr int NT = (KS + 1) / 2;
  static constexpr int NE = (DP + 31) / 32;
  static constexpr int KMIN = DP <= 32 ? DP - 1 : (DP == 50 ? 49 : DP - 3);
  static constexpr int NLT = NS + (NE > 1 ? NS - 2 : 0);
  static constexpr int NPC = (DP + 3) / 4;
  static constexpr size_t r16(size_t x) { return (x + 15) / 16 * 16; }
  // row buffer of a wave: 8 waves per workgroup -- whole 1 KiB pieces (the last one overshoots the 32 rows); 4 waves per
  // workgroup -- exactly the 32 rows (the lanes of a piece that lie behind them are masked out of the request), so that TWO
  // such workgroups fit one CU's 160 KiB (d = 50: 2 x 80 960 B)
  static constexpr size_t WAVE = NW == 8 ? r16((size_t)NPC * 1024 + 512) : r16((size_t)32 * DP * 8);
  static constexpr size_t FRAG = (size_t)(2 * NT * NS + 2 * NLT) * 1024 + (size_t)(32 * NE + 32 * NT) * 4 + (size_t)(16 * NS) * 8;
  // + the workgroup's compaction counts (all LDS in the dynamic region); the operand reads of the padded coordinates
  // (k >= d, values discarded) reach up to 16 NS - d doubles behind the last wave's rows: they stay inside the allocation
  static constexpr size_t LDS = r16(FRAG) + NW * WAVE + 64 + (NW == 8 ? 0 : 128);
};

// The matrix chains of a group as ONE list of steps (a step = one k-step of one output tile: two fragment reads, three matrix
// instructions): first the L^T tiles (ellipsoid form), then the T^T tiles, tiles taking turns within each part so that
// neighbouring steps are independent chains.  i -> (part, tile, k-step, fragment index in its table)
template <int NS, int NE, int NT>
struct F4Steps {
  static constexpr int NYE = NS + (NE > 1 ? NS - 2 : 0);
  static constexpr int N = NYE + NT * NS;
  static constexpr int ye_find(int i, bool want_t) {
    int c = 0;
    for (int s = 0; s < NS; ++s)
      for (int t = 0; t < NE; ++t)
        if (s >= 2 * t) {
          if (c == i) return want_t ? t : s;
          ++c;
        }
    return 0;
  }
  static constexpr bool is_t(int i) { return i >= NYE; }
  static constexpr int tile(int i) { return i < NYE ? ye_find(i, true) : (i - NYE) % NT; }
  static constexpr int kstep(int i) { return i < NYE ? ye_find(i, false) : (i - NYE) / NT; }
  static constexpr int fragment(int i) { return i < NYE ? (tile(i) ? NS : 0) + kstep(i) - 2 * tile(i) : tile(i) * NS + kstep(i); }
  static constexpr bool first_of_chain(int i) { return i >= NYE && kstep(i) == 0; }
};

__host__ __device__ constexpr int f4_column(int t, int i) {
  return 32 * t + 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3);
}
// accumulator registers 2 m, 2 m + 1 of tile t hold, in BOTH halves of the wave, rows of T^T that belong to padded columns
// (>= dp): their values are exact zeros (zero matrix rows, zero centre term)
__host__ __device__ constexpr bool f4_pair_is_padding(int t, int m, int dp) {
  for (int r = 2 * m; r < 2 * m + 2; ++r)
    for (int h = 0; h < 2; ++h)
      if (f4_column(t, (r & 3) + 8 * (r >> 2) + 4 * h) < dp) return false;
  return true;
}
__device__ __forceinline__ float half_sum(float v) { return half_sum32(v); }
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
template <int I, int NM, int NV>
__device__ __forceinline__ void pin_step() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV * (I + 1)) / NM - (NV * I) / NM, 0);
    pin_step<I + 1, NM, NV>();
  }
}

}  // namespace

// SQ ("same quadratic form"): the wrapping ellipsoid's matrix A equals T T^T of the layer up to a measured residue E and both
// share their centre (AffineLayer with one cluster: mlfriends.pyx:447-452 against :684-706 -- the same sample covariance times
// d + 2, inverted once by LAPACK and once through its eigen-decomposition).  Then delta^T A delta = |T^T delta|^2 + delta^T E delta:
// the ellipsoid form is read off the whitening chain's accumulators, the L^T chain (18 of 42 matrix instructions per group at
// d = 50), its fragments and its start values are gone.  The host hands over the constants of THAT form (region_prep4_setup:
// eta from the T chain's error model, eps enlarged by |E|_F); band proposals are decided by the exact einsum path as before.
template <int DP, int NW, bool SQ>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_prep_sweep(FusedArgs a) {
  using C = F4<DP, NW>;
  constexpr int NS = C::NS, NT = C::NT, NE = C::NE, KS = C::KS, QW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsf[];
  const uint4 *Th = reinterpret_cast<const uint4 *>(ldsf);
  const uint4 *Tl = Th + NT * NS * 64;
  const uint4 *Lh = Tl + NT * NS * 64;
  const uint4 *Ll = Lh + C::NLT * 64;
  float *y0l = reinterpret_cast<float *>(const_cast<uint4 *>(Ll + C::NLT * 64));
  float *csl = y0l + 32 * NE;
  double *ctrl = reinterpret_cast<double *>(csl + 32 * NT);
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave's number in a SCALAR register: what hangs on it (its LDS area, its groups, the conditions on them) is then computed and
  // branched on by the scalar unit -- the compiler does not see that tid >> 6 is the same in all lanes
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p32 = lane & 31, h = lane >> 5;
  unsigned char *area = ldsf + C::r16(C::FRAG) + (size_t)wv * C::WAVE;
  double *xs = reinterpret_cast<double *>(area);
  const int d = a.p.d;
  constexpr float up = 1.0f + 0x1p-18f, dn = 1.0f - 0x1p-18f;
  unsigned *wg_keep = reinterpret_cast<unsigned *>(ldsf + C::r16(C::FRAG) + NW * C::WAVE);   // [NW] + base
  unsigned &wg_base = wg_keep[8];
  // diagnostics (mlf_region_debug_fused_stamps): shader-clock stamps of ONE wave's stage boundaries
  const bool stamp_on = a.stamps != nullptr && blockIdx.x == a.stamp_block && wv == 0;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (stamp_on) {   // wave-uniform
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) a.stamps[k] = t;
    }
  };
  stamp(0);

  const double sigma = a.p.stats[0];
  const bool sig_ok = sigma >= 0x1p-60 && sigma <= 0x1p60;
  const float sig_f = sig_ok ? (float)sigma : 1.0f;
  if (blockIdx.x == 0 && tid == 0 && a.p.counters) {
    a.p.counters[0] = 0;
    a.p.counters[1] = 0;
  }
  if (!(a.variant & 1u)) {
    uint4 *dst = reinterpret_cast<uint4 *>(ldsf);
    const uint4 *srcT = reinterpret_cast<const uint4 *>(a.p.TtF);
    const uint4 *srcL = reinterpret_cast<const uint4 *>(a.p.LtF);
    for (int e = tid; e < 2 * NT * NS * 64; e += 64 * NW) dst[e] = srcT[e];
    if constexpr (!SQ)
      for (int e = tid; e < 2 * C::NLT * 64; e += 64 * NW) dst[2 * NT * NS * 64 + e] = srcL[e];
  } else {   // variant bit 0 (default): the matrix fragments as 1 KiB pieces straight into LDS (global_load_lds): every request of
             // the wave under way at once, no register round trip -- k_prep_sweep 0.203 -> 0.1997 ms in one process
             // (profiles/r06_fused_ab.jsonl); the load / store loop stays for A/B runs
    typedef __attribute__((address_space(1))) const void gp_t;
    typedef __attribute__((address_space(3))) void lp_t;
    constexpr int NPT = 2 * NT * NS, NPL = SQ ? 0 : 2 * C::NLT;   // pieces of the two tables
    const unsigned char *srcT = reinterpret_cast<const unsigned char *>(a.p.TtF);
    const unsigned char *srcL = reinterpret_cast<const unsigned char *>(a.p.LtF);
#pragma unroll
    for (int i = 0; i < (NPT + NPL + NW - 1) / NW; ++i) {
      const int pc = wv + NW * i;   // wave-uniform
      if (pc < NPT)
        __builtin_amdgcn_global_load_lds((gp_t *)(srcT + (size_t)pc * 1024 + lane * 16), (lp_t *)(ldsf + (size_t)pc * 1024), 16, 0, 0);
      else if (pc < NPT + NPL)
        __builtin_amdgcn_global_load_lds((gp_t *)(srcL + (size_t)(pc - NPT) * 1024 + lane * 16), (lp_t *)(ldsf + (size_t)pc * 1024), 16, 0, 0);
    }
  }
  static_assert(32 * NE <= 64 * NW && 32 * NT <= 64 * NW && 16 * NS <= 64 * NW, "one thread per constant");
  if constexpr (!SQ)
    if (tid < 32 * NE) y0l[tid] = a.p.y0[tid];
  if (tid < 32 * NT) {
    const int col = f4_column(tid >> 5, tid & 31);
    csl[tid] = col < DP ? 2.0f * (float)(sigma * a.p.stats[8 + col]) : 0.0f;
  }
  if (tid < 16 * NS) ctrl[tid] = tid < d ? (double)a.p.c.s_x * a.p.lay_ctr[tid] : 0.0;

  const long long np = a.p.np;
  const long long ngroups = (np + 31) / 32;
  const long long set = (long long)blockIdx.x * NW + wv;
  const long long g0 = set * QW;
  const long long total = np * (long long)d;
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  // The 32 rows of a group are 16 d contiguous 16-byte units; they travel as 1 KiB pieces (64 lanes x 16 bytes).  Four pieces
  // share ONE address register and ONE LDS base (M0): the instruction's immediate offset (0, 1024, 2048, 3072) moves both
  // sides -- 13 separate 64-bit address computations and M0 writes per group were 10 % of the stage's vector instructions.
  // Only the group's own units are requested (whole pieces by a scalar condition, the last one by a lane mask): nothing is read
  // behind the batch.  The one group that ends behind the batch takes the plain loop at the bottom.
  const int units = 16 * d, nfull = units >> 6, rem = units & 63;
  auto fetch_group = [&](long long grp) __attribute__((always_inline)) {
    if (grp >= ngroups) return;
    const long long base = grp * 32 * (long long)d;
    if (base + 32 * (long long)d <= total) {
      const unsigned char *src = reinterpret_cast<const unsigned char *>(a.p.pts + base) + lane * 16;
      // the piece count through an opaque copy: compared again at every use (one s_cmp) -- seen as a loop invariant of the
      // unrolled groups, the 13 comparison results were kept as 64-bit masks, spilled, and read back lane by lane
      int nf = nfull;
      asm volatile("" : "+s"(nf));
#pragma unroll
      for (int ib = 0; ib < C::NPC; ib += 4) {   // whole pieces: scalar conditions only (the immediate offset has to be a literal)
        gptr_t *gp = (gptr_t *)(src + ib * 1024);
        lptr_t *lp = (lptr_t *)(area + ib * 1024);
        if (ib < nf) __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        if (ib + 1 < C::NPC && ib + 1 < nf) __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        if (ib + 2 < C::NPC && ib + 2 < nf) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        if (ib + 3 < C::NPC && ib + 3 < nf) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
      }
      if (lane < rem)   // the last, partial piece (one lane mask per group)
        __builtin_amdgcn_global_load_lds((gptr_t *)(src + nf * 1024), (lptr_t *)(area + nf * 1024), 16, 0, 0);
    } else {
#pragma nounroll
      for (int e = lane; e < units; e += 64) {   // (at most one group per batch) plain loads, zero behind the batch
        const long long g = base + 2 * e;
        double v0 = 0.0, v1 = 0.0;
        if (g < total) v0 = a.p.pts[g];
        if (g + 1 < total) v1 = a.p.pts[g + 1];
        xs[2 * e] = v0;
        xs[2 * e + 1] = v1;
      }
    }
  };
  fetch_group(g0);
  if (a.variant & 1u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tables (and its first rows) have landed
  __syncthreads();   // fragments and constants are in LDS
  stamp(1);

  const float sqrt_k = __builtin_sqrtf((float)(16 * KS));
  const float namax = (float)a.p.stats[1] * up;
  float cs2 = 0.0f;
  for (int e = 0; e < 32 * NT; ++e) cs2 = __builtin_fmaf(csl[e], csl[e], cs2);
  const float csn = 0.5f * __builtin_sqrtf(cs2) * up;
  const float zeta_scale = sig_f * a.p.c.zt * up;
  const float zeta0 = (a.p.c.g_chain * csn + sig_f * a.p.c.zt_abs) * up + 0x1p-100f;
  const float kappa = -2.0f * sig_f * a.p.c.inv_st_sx;
  const double sr = sigma * sqrt(a.p.r2);
  const float sr_lo = (float)(sr * (1.0 - 0x1p-30)) * dn;
  const float sr_hi = (float)(sr * (1.0 + 0x1p-30)) * up;
  const float inv_sx2 = a.p.c.inv_sx * a.p.c.inv_sx, inv_slsx2 = a.p.c.inv_sl_sx * a.p.c.inv_sl_sx;
  const double sxd = (double)a.p.c.s_x;

  // ---- A. the per-proposal stage of this wave's four groups (k_prep4's body); fragments and thresholds stay in registers
  half8v bq[QW][KS];
  float tlo[QW], thi[QW];
  half8v hia[NS], loa[NS];
  float dn2a = 0.0f;
  // this lane's piece of its proposal's row and of the centre: ONE address register each, the coordinate in the instruction's
  // offset field (left to itself the compiler keeps 52 precomputed addresses in registers across the groups and reads four
  // values per LDS round trip for want of room)
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  lds_cdouble *xrow = (lds_cdouble *)(xs + p32 * d + 8 * h);
  lds_cdouble *crow = (lds_cdouble *)(ctrl + 8 * h);
  asm volatile("" : "+v"(xrow), "+v"(crow));
  auto operands = [&]() __attribute__((always_inline)) {
    float dn2 = 0.0f;
    // SB k-steps per LDS round trip: all their reads first, pinned there by an empty asm that takes the values (left alone the
    // compiler asks for four values, waits, converts, asks for the next four: 7 dependent round trips per group)
    constexpr int SB = MLF_OPERAND_STEPS;
#pragma unroll
    for (int sb = 0; sb < NS; sb += SB) {
      double xv[SB][8], cv[SB][8];
#pragma unroll
      for (int s = sb; s < sb + SB && s < NS; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          if (16 * s + (jj & ~1) >= DP) continue;   // (compile time) padding in both halves of the wave, see below
          xv[s - sb][jj] = xrow[16 * s + jj];
          cv[s - sb][jj] = crow[16 * s + jj];
        }
#pragma unroll
      for (int s = sb; s < sb + SB && s < NS; ++s)
#pragma unroll
        for (int j4 = 0; j4 < 8; j4 += 4) {
          if (16 * s + j4 >= DP) continue;
          if (16 * s + j4 + 2 >= DP)
            asm volatile("" : "+v"(xv[s - sb][j4]), "+v"(xv[s - sb][j4 + 1]), "+v"(cv[s - sb][j4]), "+v"(cv[s - sb][j4 + 1]) : : "memory");
          else
            asm volatile(""
                         : "+v"(xv[s - sb][j4]), "+v"(xv[s - sb][j4 + 1]), "+v"(xv[s - sb][j4 + 2]), "+v"(xv[s - sb][j4 + 3]),
                           "+v"(cv[s - sb][j4]), "+v"(cv[s - sb][j4 + 1]), "+v"(cv[s - sb][j4 + 2]), "+v"(cv[s - sb][j4 + 3])
                         :
                         : "memory");
        }
#pragma unroll
      for (int s = sb; s < sb + SB && s < NS; ++s) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          if (16 * s + 2 * j2 >= DP) {   // (compile time) both halves of the wave hold padding here: k >= DP >= d -- exact zeros,
            const _Float16 z = (_Float16)0.0f;   // nothing to read or convert (d = 50: 6 of the 32 pairs of a lane)
            hia[s][2 * j2] = z;
            hia[s][2 * j2 + 1] = z;
            loa[s][2 * j2] = z;
            loa[s][2 * j2 + 1] = z;
            continue;
          }
          float x32[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int jj = 2 * j2 + e;
            const int k = 16 * s + 8 * h + jj;
            const bool ok = 16 * s + 8 + jj < C::KMIN || k < d;
            const double x = __builtin_fma(xv[s - sb][jj], sxd, -cv[s - sb][jj]);
            x32[e] = ok ? (float)x : 0.0f;
            dn2 = __builtin_fmaf(x32[e], x32[e], dn2);
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
    return dn2;
  };
  auto frag = [&](const uint4 *base, int idx) __attribute__((always_inline)) {
    union { uint4 u; half8v h8; } cv;
    cv.u = base[idx * 64 + lane];
    return cv.h8;
  };
  if (g0 < ngroups) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    stamp(2);
    dn2a = operands();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    fetch_group(g0 + 1);
  }
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const long long grp = g0 + g;
    unsigned pk[KS * 4];
#pragma unroll
    for (int i = 0; i < KS * 4; ++i) pk[i] = 0u;
    float lo_f = -1.0f, hi_f = -1.0f;
    if (grp < ngroups) {   // wave-uniform
      const long long p = grp * 32 + p32;
      const bool live = p < np;
      // variant bit 2: gate[] arrives holding a pre-gate (device-side sampling: "inside the unit cube") -- a proposal that failed
      // it is outside for certain, whatever the ellipsoid says; asked for here, needed after the chains
      unsigned char pre = 1;
      if ((a.variant & 4u) && live) pre = a.p.gate[p];
      float qs = 0.0f;
      float16v tt[NT];
      float cs[NT][16];
      {
        // fragments TWO steps ahead of the matrix instructions that take them (a ring of three register pairs; the empty asm
        // keeps every read where it is written: left alone the compiler reads a step's pair, waits for it, and issues its three
        // instructions -- an LDS round trip per step in a chain that is serial anyway)
        using ST = F4Steps<NS, NE, NT>;
        constexpr int I0 = SQ ? ST::NYE : 0;   // SQ: the list starts at the whitening chain
        float16v ye[NE];
        if constexpr (!SQ) {
#pragma unroll
          for (int t = 0; t < NE; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ye[t][r] = y0l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
        }
        // the packing's column constants: asked for here, in front of the chains (their fences keep the reads here); the
        // packing loop used to wait for them piece by piece, six LDS round trips behind each other
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (!f4_pair_is_padding(t, r >> 1, DP)) cs[t][r] = csl[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
        half8v fh[3], fl[3];
        auto read_step = [&](auto I) __attribute__((always_inline)) {
          constexpr int i = decltype(I)::value;
          if constexpr (i < ST::N) {
            constexpr int f = ST::fragment(i);
            fh[i % 3] = frag(ST::is_t(i) ? Th : Lh, f);
            fl[i % 3] = frag(ST::is_t(i) ? Tl : Ll, f);
          }
        };
        read_step(std::integral_constant<int, I0>{});
        read_step(std::integral_constant<int, I0 + 1>{});
        static_for<I0, ST::N>([&](auto I) __attribute__((always_inline)) {
          constexpr int i = decltype(I)::value;
          read_step(std::integral_constant<int, i + 2>{});
          asm volatile("" ::: "memory");
          constexpr int t = ST::tile(i), ks = ST::kstep(i);
          const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if constexpr (ST::is_t(i)) {
            tt[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[i % 3], hia[ks], ST::first_of_chain(i) ? z : tt[t], 0, 0, 0);
            tt[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[i % 3], loa[ks], tt[t], 0, 0, 0);
            tt[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[i % 3], hia[ks], tt[t], 0, 0, 0);
          } else {
            ye[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[i % 3], hia[ks], ye[t], 0, 0, 0);
            ye[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[i % 3], loa[ks], ye[t], 0, 0, 0);
            ye[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[i % 3], hia[ks], ye[t], 0, 0, 0);
          }
          if constexpr (!SQ && i == ST::NYE - 1) {
#pragma unroll
            for (int t2 = 0; t2 < NE; ++t2)
#pragma unroll
              for (int r = 0; r < 16; ++r) qs = __builtin_fmaf(ye[t2][r], ye[t2][r], qs);
          }
        });
        if constexpr (SQ) {   // |s_T s_x T^T delta|^2: the rows of padded columns are exact zeros
#pragma unroll
          for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (!f4_pair_is_padding(t2, r >> 1, DP)) qs = __builtin_fmaf(tt[t2][r], tt[t2][r], qs);
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
        const float eta = (a.p.c.g_chain * (a.p.c.y0n + a.p.c.lf * dnorm) + a.p.c.el * dnorm + a.p.c.l_abs) * up;
        const float de = dnorm + a.p.c.s0n;
        const float eps = a.p.c.eps_scale * (de * de) * up;
        const float hi = sq * up + eta;
        const float qhi = ((hi * hi) * up + eps) * up;
        const float lo = (sq * dn - eta) * dn;
        const float qlo = ((lo * lo) * dn - eps * up) * dn;
        sure_in = finite && qhi < a.p.c.enl_lo && pre != 0;
        sure_out = (finite && lo > 0.0f && qlo > a.p.c.enl_hi) || pre == 0;
      }
      const bool band = live && !sure_in && !sure_out;
      const bool ins_any = live && !sure_out;
      {
        const unsigned long long bm = __ballot(band && h == 0);
        if (bm != 0ull) {   // wave-uniform, rare: the band list for the exact test (k_uncertain's trailing workgroups)
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(a.p.ell_count, (unsigned)__popcll(bm));
          base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
          if (band && h == 0) {
            const unsigned slot = base + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
            if (slot < a.p.ell_cap) a.p.ell_list[slot] = (int)p;
          }
        }
      }
      if (live && h == 0) a.p.gate[p] = ins_any ? 1 : 0;
      float nbq = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if (f4_pair_is_padding(t, m, DP)) continue;   // (compile time; pk was zeroed above; d = 50: 3 of the 16 pairs)
          const float v0 = __builtin_fmaf(tt[t][2 * m], kappa, cs[t][2 * m]);
          const float v1 = __builtin_fmaf(tt[t][2 * m + 1], kappa, cs[t][2 * m + 1]);
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
      if (__any(rt == 2) && lane == 0) *a.p.scan_flag = 1u;
      if (live && h == 0) {
        a.p.route[p] = (uint8_t)rt;
        a.p.best[p] = kNone;
      }
      // operands of the next group (its rows have been on their way since this one's were read); then the one after sets out
      stamp(3 + 2 * g);
      if (g + 1 < QW && grp + 1 < ngroups) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        stamp(4 + 2 * g);
        dn2a = operands();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (g + 2 < QW) fetch_group(grp + 2);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      union { uint4v u; half8v h8; } cv;
      cv.u = (uint4v){pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]};
      bq[g][s] = cv.h8;
    }
    // every lane of a query needs its thresholds (the per-proposal stage leaves them in the low half)
    tlo[g] = low_half32(lo_f);
    thi[g] = low_half32(hi_f);
  }

  // ---- B. the first live-point range, running minima only (k_sweep_min's loop)
  int run[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) run[g] = kPosInf;
  // a wave none of whose proposals is left for the filter (all gated out by the ellipsoid, or routed to the exact scan) has
  // nothing to sweep: thi <= 0 marks such a proposal, and stage C looks at no minimum of an invalid one (set U: the launch
  // 0.151 ms -> see profiles/r06_fused_ab.jsonl)
  bool any_valid = false;
#pragma unroll
  for (int g = 0; g < QW; ++g) any_valid = any_valid || thi[g] > 0.0f;
  if (g0 < ngroups && __any(any_valid)) {
    const int ntl = a.tile1 - a.tile0;
    const int tstart = a.tile0 + (int)((set * 37) % ntl);
    constexpr int kTileBytes = KS * 1024;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.refF), 0, a.ntiles32 * kTileBytes, 0x00020000);
    const int voff = lane * 16;
    const int off_begin = a.tile0 * kTileBytes, off_end = a.tile1 * kTileBytes;
    float16v acc[QW];
    auto load_tile = [&](half8v(&A)[KS], int soff) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        union { uint4v u; half8v h8; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + s * 1024, 0);
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
    auto next_off = [&](int off) __attribute__((always_inline)) {
      const int n = off + kTileBytes;
      return n == off_end ? off_begin : n;
    };
    auto tile = [&](const half8v(&A)[KS]) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      mm(A, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(A, 2, 3);
      run[0] = tree_min(acc[0], run[0]);
      run[1] = tree_min(acc[1], run[1]);
      pin_step<0, 2 * KS, 16>();
      __builtin_amdgcn_sched_barrier(0);
      run[2] = tree_min(acc[2], run[2]);
      run[3] = tree_min(acc[3], run[3]);
      __builtin_amdgcn_sched_barrier(0);
    };
    half8v A0[KS], A1[KS], A2[KS];
    int o0 = tstart * kTileBytes, o1 = next_off(o0), o2 = next_off(o1);
    load_tile(A0, o0);
    load_tile(A1, o1);
    stamp(10);
    for (int it = 0; it < ntl; it += 3) {
      load_tile(A2, o2);
      tile(A0);
      if (it + 1 >= ntl) break;
      o0 = next_off(o2);
      load_tile(A0, o0);
      tile(A1);
      if (it + 2 >= ntl) break;
      o1 = next_off(o0);
      load_tile(A1, o1);
      tile(A2);
      o2 = next_off(o1);
    }
  }

  stamp(11);
  // ---- C. certain hits -> best; the rest goes on with its minimum (one atomic per workgroup)
  unsigned keepm[QW];
  int qmn[QW];
  unsigned total_keep = 0u;
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    keepm[g] = 0u;
    qmn[g] = kPosInf;
    if (g0 + g >= ngroups) continue;
    const long long qi = (g0 + g) * 32 + p32;
    const int m = half_min32(run[g]);
    const float mf = __int_as_float(m);
    const bool valid = qi < np && thi[g] > 0.0f;
    const bool hit = valid && mf <= tlo[g];
    if (lane < 32 && hit) a.p.best[qi] = 0;
    keepm[g] = (unsigned)__ballot(valid && !hit);
    qmn[g] = m;
    total_keep += (unsigned)__popc(keepm[g]);
  }
  // (measured and dropped in round 6: every wave reserving its own slots -- no workgroup barrier, 7 800 atomics per launch on
  // one word -- 0.2015 against 0.2002 ms for the launch, profiles/r06_fused_ab.jsonl)
  if (lane == 0) wg_keep[wv] = total_keep;
  __syncthreads();
  if (tid == 0) {
    unsigned all = 0u;
    for (int w = 0; w < NW; ++w) all += wg_keep[w];
    wg_base = all ? atomicAdd(a.ccount, all) : 0u;
  }
  __syncthreads();
  if (total_keep != 0u) {
    unsigned base = wg_base;
    for (int w = 0; w < wv; ++w) base += wg_keep[w];
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    uint4 *dst = reinterpret_cast<uint4 *>(a.cq);
    const unsigned row = (unsigned)p32;
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      if ((keepm[g] >> row) & 1u) {
        const unsigned rank = base + (unsigned)__popc(keepm[g] & ((1u << row) - 1u));
        const size_t gd = rank >> 5;
        const unsigned rd = (rank & 31u) + (unsigned)(lane & 32);
        if (rank < a.ccap) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            union { half8v h8; uint4 u; } cv;
            cv.h8 = bq[g][s];
            dst[(gd * KS + s) * 64 + rd] = cv.u;
          }
        }
        if (lane < 32 && rank < a.ccap) {
          a.ctlo[rank] = tlo[g];
          a.cthi[rank] = thi[g];
          a.cmap[rank] = (int)((g0 + g) * 32 + p32);
          a.cmin[rank] = qmn[g];
        }
      }
      base += (unsigned)__popc(keepm[g]);
    }
  }
  stamp(12);
}

bool fused_usable(int dp) { return dp >= 2 && dp <= 56 && (dp & 1) == 0; }

template <int D, int NW, bool SQ>
static hipError_t launch_prep_sweep_t(const FusedArgs &a, long long nsets, hipStream_t s) {
  constexpr size_t lds = F4<D, NW>::LDS;
  static_assert(lds <= 160 * 1024 - 64, "LDS budget");
  static DeviceGrant grant;
  if (hipError_t e = grant.ensure([] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep_sweep<D, NW, SQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }))
    return e;
  if (a.p.ks != F4<D, NW>::KS) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((nsets + NW - 1) / NW));
  hipLaunchKernelGGL((k_prep_sweep<D, NW, SQ>), grid, dim3(64 * NW), lds, s, a);
  return hipGetLastError();
}

// waves = 4 (default): workgroups of 4 waves, TWO per CU where the rows of four groups + the tables fit half the LDS (d <= 50:
// 2 x 81 088 B) -- the CU's eight waves then start and end in two independent halves (0.1768 against 0.1805 ms at C5, one
// process, profiles/r06_fused_ab.jsonl); waves = 8, and every d above 50: one workgroup of 8 waves per CU
hipError_t launch_prep_sweep(const FusedArgs &a, hipStream_t s, int waves) {
  if (a.p.np <= 0) return hipSuccess;
  const long long ngroups = (a.p.np + 31) / 32;
  const long long nsets = (ngroups + 3) / 4;
  const bool sq = (a.variant & 2u) != 0u;   // the caller has put the constants of that form into a.p.c
  switch (a.p.dp) {
#define X(D)                                                                                  \
  case D:                                                                                     \
    if constexpr (2 * F4<D, 4>::LDS <= 160 * 1024) {                                          \
      if (waves == 4) return sq ? launch_prep_sweep_t<D, 4, true>(a, nsets, s) : launch_prep_sweep_t<D, 4, false>(a, nsets, s); \
    }                                                                                         \
    return sq ? launch_prep_sweep_t<D, 8, true>(a, nsets, s) : launch_prep_sweep_t<D, 8, false>(a, nsets, s);
    MLF_FOR_EACH_DP_MID(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace mlf
