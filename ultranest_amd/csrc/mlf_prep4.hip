// mlf_prep4.hip -- per-proposal stage of MLFriends.inside for AffineLayer-family regions, bounded form:
//   H3 ellipsoid test (reference mlfriends.pyx:882-912)
//   T1 whitening      (:737-743)              -- only as accurate as the binary16 pre-filter needs it
//   + binary16 quantisation and thresholds for the MFMA pre-filter (mlf_filter.hip)
//
// k_prep3 whitens every proposal in the reference's binary64 arithmetic and stores the result (400 MB per 10^6 x 50
// batch) although only the ~10^-4 of the pairs that the pre-filter cannot decide ever read it.  Here both d x d
// products run on the matrix cores with SPLIT binary16 operands (v_mfma_f32_32x32x16_f16, binary32 accumulate), the
// whitened point goes straight into the filter's binary16 operand fragments, and nothing else is written.  What
// keeps the results identical to the reference's:
//   * ellipsoid: q^ = |y^|^2 with y^ ~ L^T (x - c_e) carries an error bound eta (below); proposals with
//     |sqrt q^ - sqrt enlarge| inside that band go to a list and are decided in binary64 (ell_exact_*);
//   * whitening: |bq - sigma (b - c)| <= zeta enters the pre-filter's Delta (filter_thresholds4), so a pair is
//     "certain" only if it is certain for every point within zeta; the queries of the uncertain pairs are whitened in
//     the reference arithmetic by the re-check itself (k_recheck_whiten).
//
// Why binary16 and not the FP32 matrix instructions (the first version of this kernel used v_mfma_f32_32x32x2_f32, a
// bit-exact k-ascending fmaf chain): on gfx950 those execute on the vector ALU's own lanes -- SQ_VALU_MFMA_COEXEC_CYCLES
// was 0, 78 us of matrix time and ~45 us of vector time simply added up (0.155 ms per 10^6 x 50 batch) -- while the
// binary16 ones run beside the vector ALU and take a quarter of the time for the three partial products below.
//
// Operands.  x' = s_x (x - c) (binary64 FMA: one rounding), x32 = fl32(x'), hi = fl16(x32), lo = fl16(x32 - hi); a matrix
// entry m' = s_m m likewise Mh = fl16(m'), Ml = fl16(m' - Mh) (host, binary64; s_x, s_m powers of two that centre the
// values in the binary16 range).  The chain adds, per output, the exact products Mh hi + Mh lo + Ml hi over all k
// to the start value: 3 K / 16 matrix instructions.
// Error model.  (1) operands: |x' - hi - lo| <= 2^-21.6 |x'| + 2^-25 per coordinate (fl32, then two binary16 roundings;
// the absolute term covers binary16 subnormals); the dropped product |Ml lo| <= 2^-22 (1 + 2^-9) |m' x'|; the matrix's
// own representation error E = m' - Mh - Ml is known exactly on the host.  (2) accumulation: one instruction sums its
// two groups of 8 exact products and adds them to C one after the other, each step rounded to binary32
// (scripts/probes/mfma16_acc_probe.hip: with C = -2^20, one product 2^20 and fifteen products 2^-8 the result is
// 8 x 2^-8 or 0 depending on the group the large product sits in; random and cancelling inputs: error <= 4.3 u S for
// one instruction, <= 6.8 u S for a chain of 12, u = 2^-24, S = |C| + sum |a b|).  The bound used is
// nu = (4 n + 8) u for a chain of n instructions: twice the worst case of round-to-nearest steps (2 chain additions per
// instruction + a depth-3 tree inside a group), which also covers truncating adders.  Together, per output row,
//     |o^ - o| <= g (|start| + sum_k |m'_k x'_k|) + sum_k |E_k| |x'_k| + 2^-25 sum_k |m'_k|,   g = nu (1 + 2^-8) + 2^-21.6 + 2^-21.9
// and in the 2-norm over the rows (Cauchy-Schwarz per row)
//     eta  = g (|y0| + |L|_F |delta|) + |E_L|_F |delta| / s_L + |L|_F sqrt(K) 2^-25 / s_x
//     zeta = sigma [ g (|c_s| + |T|_F |delta|) + |E_T|_F |delta| / s_T + |T|_F sqrt(K) 2^-25 / s_x ]
// |delta| comes from the same x32 (binary32 sum of squares).  The reference's own rounding of the quadratic form and
// the factorisation A = L L^T are covered by eps = 2^-34 |A|_F |x - c_e|^2 exactly as in k_prep3.  The proposal is inside
// for certain if (sqrt q^ + eta)^2 + eps < enlarge, outside for certain if (sqrt q^ - eta)^2 - eps > enlarge,
// everything else (NaN / inf included) is decided in binary64.
//
// Layout.  One wave owns a group of 32 proposals = one 32-query group of the filter.  B operand (16 k x 32 proposals):
// lane (p = l & 31, h = l >> 5) holds coordinates 16 s + 8 h + j, j = 0..7, at k-step s.  A operand = 32 matrix rows x
// 16 k from LDS (one ds_read_b128 per instruction).  C (32 rows x 32 proposals): lane (p, h) holds rows
// i = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15.  The host orders the rows of T so that row i of tile t is filter column
// 32 t + 16 (r >> 3) + 8 h + (r & 7): a lane then holds, for filter k-step 2 t + (r >> 3), exactly the 8 consecutive
// columns of its own fragment piece -- the f16 operand is packed in registers and stored with one contiguous 1 KiB wave
// store per k-step, no transpose.  Rows arrive in LDS as they lie in HBM (global_load_lds, 16-byte pieces, no staging
// registers); the next group's rows are in flight while the current one is processed.
// -ffp-contract=off; FMAs only where written.
#include "mlf_prep4.hpp"

#include <math.h>

#include <vector>

#include "mlf_dpp_dev.hpp"
#include "mlf_filter_dev.hpp"
#include "mlf_prep3.hpp"
#include "mlf_ell_exact.hpp"
#include "mlf_recheck_dev.hpp"

namespace mlf {

typedef float float16v __attribute__((ext_vector_type(16)));
typedef double double4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

namespace {

template <int DP>
struct P4 {
  static constexpr int NS = (DP + 15) / 16;             // k-steps of 16 coordinates (K = 16 NS)
  static constexpr int KS = (DP + 6 + 15) / 16;         // filter k-steps of 16 binary16 columns
  static constexpr int NT = (KS + 1) / 2;               // 32-row output tiles of the whitening
  static constexpr int NE = (DP + 31) / 32;             // 32-row output tiles of L^T delta
  static constexpr int KMIN = DP <= 32 ? DP - 1 : (DP == 50 ? 49 : DP - 3);   // every d served by this instance exceeds KMIN - 1
  static constexpr int NLT = NS + (NE > 1 ? NS - 2 : 0);   // stored k-steps of the L^T fragments (tile 1: k >= 32)
  static constexpr int NPC = (DP + 3) / 4;              // 1 KiB pieces (64 lanes x 16 bytes) that cover a group of 32 rows
  static constexpr int WAVE_DOUBLES = NPC * 128 + 64;   // staging buffer per wave: the group's rows as they lie in HBM
  static constexpr size_t lds_for(int waves) {
    return (size_t)(2 * NT * NS + 2 * NLT) * 1024 + (size_t)(32 * NE + 32 * NT) * sizeof(float) +
           (size_t)(16 * NS) * sizeof(double) + (size_t)waves * WAVE_DOUBLES * sizeof(double);
  }
  // waves of a workgroup: one workgroup per CU with as many waves (up to two per SIMD, the register budget) as
  // the 160 KiB of LDS hold staging buffers for -- the fragments are shared by the workgroup
  static constexpr int NW = lds_for(8) <= 160 * 1024 ? 8 : (lds_for(6) <= 160 * 1024 ? 6 : 4);
  static constexpr size_t lds_bytes() { return lds_for(NW); }
};

// filter column of row i of tile t of the whitening product
__host__ __device__ constexpr int p4_column(int t, int i) {
  return 32 * t + 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3);
}
// accumulator registers 2 m, 2 m + 1 of tile t hold, in BOTH halves of the wave, rows of T^T of padded columns (>= dp): zeros
__host__ __device__ constexpr bool p4_pair_is_padding(int t, int m, int dp) {
  for (int r = 2 * m; r < 2 * m + 2; ++r)
    for (int h = 0; h < 2; ++h)
      if (p4_column(t, (r & 3) + 8 * (r >> 2) + 4 * h) < dp) return false;
  return true;
}

__device__ __forceinline__ float half_sum(float v) { return half_sum32(v); }

}  // namespace

template <int DP>
__global__ __launch_bounds__(64 * P4<DP>::NW, 1) void k_prep4(Prep4Args a) {
  using C = P4<DP>;
  constexpr int NS = C::NS, NT = C::NT, NE = C::NE, KS = C::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds4[];
  // A fragments, 1 KiB each (64 lanes x 8 binary16): T hi [NT][NS], T lo [NT][NS], L^T hi [NLT], L^T lo [NLT]
  const uint4 *Th = reinterpret_cast<const uint4 *>(lds4);
  const uint4 *Tl = Th + NT * NS * 64;
  const uint4 *Lh = Tl + NT * NS * 64;
  const uint4 *Ll = Lh + C::NLT * 64;
  float *y0l = reinterpret_cast<float *>(const_cast<uint4 *>(Ll + C::NLT * 64));
  float *csl = y0l + 32 * NE;
  double *ctrl = reinterpret_cast<double *>(csl + 32 * NT);   // [16 NS] s_x c_k, zero past d
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // in a scalar register: what hangs on it is scalar work (mlf_fused.hip)
  const int p32 = lane & 31, h = lane >> 5;
  double *xs = ctrl + 16 * NS + wv * C::WAVE_DOUBLES;
  const int d = a.d;
  const bool quant = a.do_tr != 0;
  constexpr float up = 1.0f + 0x1p-18f, dn = 1.0f - 0x1p-18f;

  const double sigma = quant ? a.stats[0] : 1.0;
  const bool sig_ok = sigma >= 0x1p-60 && sigma <= 0x1p60;
  const float sig_f = sig_ok ? (float)sigma : 1.0f;
  if (blockIdx.x == 0 && tid == 0 && a.counters) {
    a.counters[0] = 0;
    a.counters[1] = 0;
  }
  {   // the matrix fragments as 1 KiB pieces straight into LDS (global_load_lds; wave wv takes pieces wv, wv + NW, ...): every
      // request of the wave under way at once.  A load / store loop made two to four dependent memory round trips here --
      // 5-8 us in front of the first matrix instruction of a workgroup (stage stamps of k_prep_sweep, profiles/r06_fused_ab.jsonl)
    typedef __attribute__((address_space(1))) const void gp_t;
    typedef __attribute__((address_space(3))) void lp_t;
    constexpr int NPT = 2 * NT * NS, NPL = 2 * C::NLT;
    const unsigned char *srcT = reinterpret_cast<const unsigned char *>(a.TtF);
    const unsigned char *srcL = reinterpret_cast<const unsigned char *>(a.LtF);
#pragma unroll
    for (int i = 0; i < (NPT + NPL + C::NW - 1) / C::NW; ++i) {
      const int pc = wv + C::NW * i;   // wave-uniform
      if (pc < NPT) {
        if (quant)
          __builtin_amdgcn_global_load_lds((gp_t *)(srcT + (size_t)pc * 1024 + lane * 16), (lp_t *)(lds4 + (size_t)pc * 1024), 16, 0, 0);
      } else if (pc < NPT + NPL) {
        __builtin_amdgcn_global_load_lds((gp_t *)(srcL + (size_t)(pc - NPT) * 1024 + lane * 16), (lp_t *)(lds4 + (size_t)pc * 1024), 16, 0, 0);
      }
    }
  }
  if (tid < 32 * NE) y0l[tid] = a.y0[tid];   // already scaled by s_L s_x
  if (tid < 32 * NT) {
    const int col = p4_column(tid >> 5, tid & 31);
    csl[tid] = (quant && col < DP) ? 2.0f * (float)(sigma * a.stats[8 + col]) : 0.0f;   // + 2 sigma c_s: the operand is -2 (sigma t - sigma c_s)
  }
  if (tid < 16 * NS) ctrl[tid] = tid < d ? (double)a.c.s_x * a.lay_ctr[tid] : 0.0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces have landed
  __syncthreads();

  // uniform per-kernel quantities (binary32, rounded outward)
  float namax = 0.0f, zeta_scale = 0.0f, zeta0 = 0.0f, sr_lo = 0.0f, sr_hi = 0.0f, kappa = 0.0f;
  const float sqrt_k = __builtin_sqrtf((float)(16 * KS));
  if (quant) {
    namax = (float)a.stats[1] * up;
    float cs2 = 0.0f;
    for (int e = 0; e < 32 * NT; ++e) cs2 = __builtin_fmaf(csl[e], csl[e], cs2);
    const float csn = 0.5f * __builtin_sqrtf(cs2) * up;
    zeta_scale = sig_f * a.c.zt * up;                                   // per unit |delta|
    zeta0 = (a.c.g_chain * csn + sig_f * a.c.zt_abs) * up + 0x1p-100f;
    kappa = -2.0f * sig_f * a.c.inv_st_sx;                              // accumulator -> -2 sigma t (powers of two: exact)
    const double sr = sigma * sqrt(a.r2);
    sr_lo = (float)(sr * (1.0 - 0x1p-30)) * dn;
    sr_hi = (float)(sr * (1.0 + 0x1p-30)) * up;
  }
  const float inv_sx2 = a.c.inv_sx * a.c.inv_sx, inv_slsx2 = a.c.inv_sl_sx * a.c.inv_sl_sx;

  const long long ngroups = quant ? a.nqpad / 32 : (a.np + 31) / 32;
  const long long wave_id = (long long)blockIdx.x * C::NW + wv;
  const long long nwaves = (long long)gridDim.x * C::NW;
  const long long total = a.np * (long long)d;

  // A group of 32 rows is 32 d contiguous doubles; it is copied as it lies, in 16-byte pieces, straight into the
  // wave's LDS buffer (global_load_lds: no staging registers).  Lanes past the group / past the batch stay out; what
  // they would have written keeps its old contents (finite or not, those rows are never live).
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  auto fetch_group = [&](long long grp) {
    const long long base = grp * 32 * (long long)d;
    if (base + 128 * C::NPC <= total) {   // wave-uniform: the whole 1 KiB-granular window lies inside the batch
      // four pieces share one address register and one LDS base (M0): the instruction's immediate offset moves both sides
      const unsigned char *src = reinterpret_cast<const unsigned char *>(a.pts + base) + lane * 16;
#pragma unroll
      for (int ib = 0; ib < C::NPC; ib += 4) {
        gptr_t *gp = (gptr_t *)(src + ib * 1024);
        lptr_t *lp = (lptr_t *)(reinterpret_cast<char *>(xs) + ib * 1024);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        if (ib + 1 < C::NPC) __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        if (ib + 2 < C::NPC) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        if (ib + 3 < C::NPC) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
      }
    } else {
#pragma nounroll
      for (int e = lane; e < 16 * d; e += 64) {   // (the last groups of a batch) plain loads, zero behind the batch
        const long long g = base + 2 * e;
        double v0 = 0.0, v1 = 0.0;
        if (g < total) v0 = a.pts[g];
        if (g + 1 < total) v1 = a.pts[g + 1];
        xs[2 * e] = v0;
        xs[2 * e + 1] = v1;
      }
    }
  };
  // operands of the staged group: lane (p, h) takes coordinates 16 s + 8 h + j of row p, centred and scaled in binary64
  // (one FMA), rounded to binary32 and split into two binary16 pieces; returns the lane's part of |x32|^2
  const double sxd = (double)a.c.s_x;
  // this lane's piece of its proposal's row and of the centre: one address register each, the coordinate in the instruction's
  // offset field, and ALL reads of a group in one LDS round trip (pinned by an empty asm that takes the values; mlf_fused.hip)
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  lds_cdouble *xrow = (lds_cdouble *)(xs + p32 * d + 8 * h);
  lds_cdouble *crow = (lds_cdouble *)(ctrl + 8 * h);
  asm volatile("" : "+v"(xrow), "+v"(crow));
  auto operands = [&](half8v *hi, half8v *lo) {
    float dn2 = 0.0f;
    double xv[NS][8], cv[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (16 * s + (jj & ~1) >= DP) continue;   // (compile time) padding in both halves of the wave, see below
        xv[s][jj] = xrow[16 * s + jj];            // unconditional reads (k >= d: discarded)
        cv[s][jj] = crow[16 * s + jj];
      }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j4 = 0; j4 < 8; j4 += 4) {
        if (16 * s + j4 >= DP) continue;
        if (16 * s + j4 + 2 >= DP)
          asm volatile("" : "+v"(xv[s][j4]), "+v"(xv[s][j4 + 1]), "+v"(cv[s][j4]), "+v"(cv[s][j4 + 1]) : : "memory");
        else
          asm volatile(""
                       : "+v"(xv[s][j4]), "+v"(xv[s][j4 + 1]), "+v"(xv[s][j4 + 2]), "+v"(xv[s][j4 + 3]), "+v"(cv[s][j4]),
                         "+v"(cv[s][j4 + 1]), "+v"(cv[s][j4 + 2]), "+v"(cv[s][j4 + 3])
                       :
                       : "memory");
      }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        if (16 * s + 2 * j2 >= DP) {   // (compile time) both halves of the wave hold padding here: k >= DP >= d -- exact zeros
          const _Float16 z = (_Float16)0.0f;
          hi[s][2 * j2] = z;
          hi[s][2 * j2 + 1] = z;
          lo[s][2 * j2] = z;
          lo[s][2 * j2 + 1] = z;
          continue;
        }
        float x32[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int jj = 2 * j2 + e;
          const int k = 16 * s + 8 * h + jj;
          const bool ok = 16 * s + 8 + jj < C::KMIN || k < d;   // columns below KMIN exist for every d of this instance
          const double x = __builtin_fma(xv[s][jj], sxd, -cv[s][jj]);
          x32[e] = ok ? (float)x : 0.0f;
          dn2 = __builtin_fmaf(x32[e], x32[e], dn2);
        }
        const half2v hp = __builtin_convertvector((float2v){x32[0], x32[1]}, half2v);
        const float2v res = {x32[0] - (float)hp[0], x32[1] - (float)hp[1]};   // exact
        const half2v lp = __builtin_convertvector(res, half2v);
        hi[s][2 * j2] = hp[0];
        hi[s][2 * j2 + 1] = hp[1];
        lo[s][2 * j2] = lp[0];
        lo[s][2 * j2 + 1] = lp[1];
      }
    }
    return dn2;
  };
  auto frag = [&](const uint4 *base, int idx) {
    union { uint4 u; half8v h; } cv;
    cv.u = base[idx * 64 + lane];
    return cv.h;
  };

  // Pipeline over the wave's groups: the rows of group g + 1 travel to LDS while group g is processed; at the end of
  // the iteration they are turned into operands and the rows of group g + 2 set out.  Matrix and vector phases of the
  // two waves of a SIMD overlap each other (the binary16 matrix instructions run beside the vector ALU).
  half8v hia[NS], loa[NS];
  float dn2a = 0.0f;
  if (wave_id < ngroups) {
    fetch_group(wave_id);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    dn2a = operands(hia, loa);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next group overwrites the rows
    __builtin_amdgcn_wave_barrier();
    fetch_group(wave_id + nwaves);
  }
  uint4 *qdst = reinterpret_cast<uint4 *>(a.qF);
  for (long long grp = wave_id; grp < ngroups; grp += nwaves) {
    const long long p = grp * 32 + p32;
    const bool live = p < a.np;

    // ---- y^ = y0 + L^T delta
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
      for (int s = 2 * t; s < NS; ++s) {   // L^T is upper triangular: rows 32 t.. have no entries left of column 32 t
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

    // ---- -2 bq = -2 (sigma T^T delta - sigma c_s), straight into the filter operand
    float16v tt[NT];
    unsigned pk[NT * 8];
    float nbq = 0.0f;
    if (quant) {
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
    }

    qs = half_sum(qs) * inv_slsx2;          // powers of two: exact scalings
    const float dn2 = half_sum(dn2a) * inv_sx2;

    bool sure_in = false, sure_out = false;
    float dnorm = 0.0f;
    {
      const bool finite = qs < 3.0e38f && dn2 < 3.0e38f;   // false for NaN
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
    const bool ins_any = live && !sure_out;   // band proposals: provisionally inside, the binary64 test has the last word
    {
      const unsigned long long bm = __ballot(band && h == 0);
      if (bm != 0ull) {   // wave-uniform, rare
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(a.ell_count, (unsigned)__popcll(bm));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (band && h == 0) {
          const unsigned slot = base + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
          if (slot < a.ell_cap) a.ell_list[slot] = (int)p;
        }
      }
    }
    if (live && h == 0) a.gate[p] = ins_any ? 1 : 0;

    if (quant) {
      // binary16 operand: columns -2 bh, and 4 |bh|^2
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if (p4_pair_is_padding(t, m, DP)) {   // (compile time) exact zeros in both halves of the wave
            pk[t * 8 + m] = 0u;
            continue;
          }
          const float v0 = __builtin_fmaf(tt[t][2 * m], kappa, csl[32 * t + ((2 * m) & 3) + 8 * ((2 * m) >> 2) + 4 * h]);
          const float v1 = __builtin_fmaf(tt[t][2 * m + 1], kappa, csl[32 * t + ((2 * m + 1) & 3) + 8 * ((2 * m + 1) >> 2) + 4 * h]);
          union { half2v v; unsigned u; } cv;
          cv.v = __builtin_convertvector((float2v){v0, v1}, half2v);
          const float f0 = (float)cv.v[0], f1 = (float)cv.v[1];
          nbq = __builtin_fmaf(f0, f0, nbq);
          nbq = __builtin_fmaf(f1, f1, nbq);
          pk[t * 8 + m] = cv.u;
        }
      const float nb = 0.25f * half_sum(nbq);   // |bh|^2
      int rt = ins_any ? 1 : 0;
      if (rt == 1 && (!sig_ok || !(nb <= 29000.0f))) rt = 2;   // NaN / inf / does not fit binary16: exact scan
      float lo_f = -1.0f, hi_f = -1.0f;
      if (rt == 1) {
        const float zeta = (zeta_scale * dnorm + zeta0) * up;
        if (!filter_thresholds4(namax, nb, zeta, sqrt_k, sr_lo, sr_hi, &lo_f, &hi_f)) {
          rt = 2;
          lo_f = hi_f = -1.0f;
        }
      }
      // three pieces of |bh|^2 (x ones column of the live point) behind three ones (x its |ah|^2 pieces)
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
        const int c = DP + 2 * q;   // compile-time after unrolling
        if (h == ((c >> 3) & 1)) pk[4 * (c >> 4) + ((c & 7) >> 1)] = sp[q].u;
      }
      const bool keep = rt == 1;   // otherwise every operand column of this query is zero
#pragma unroll
      for (int s = 0; s < KS; ++s)
        qdst[((size_t)grp * KS + s) * 64 + lane] = make_uint4(keep ? pk[4 * s] : 0u, keep ? pk[4 * s + 1] : 0u,
                                                              keep ? pk[4 * s + 2] : 0u, keep ? pk[4 * s + 3] : 0u);
      if (__any(rt == 2) && lane == 0) *a.scan_flag = 1u;   // rare: wakes the exact-scan workgroups of this batch up
      if (h == 0) {
        a.tlo[p] = lo_f;
        a.thi[p] = hi_f;
        if (live) {
          a.route[p] = (uint8_t)rt;
          a.best[p] = kNone;
        }
      }
    }
    // ---- operands of group g + 1 (its rows have been in LDS for a while); then group g + 2 sets out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    dn2a = operands(hia, loa);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the rows are overwritten
    __builtin_amdgcn_wave_barrier();
    fetch_group(grp + 2 * nwaves);
  }
}

// ---------------------------------------------------------------- host helpers -----------------
bool prep4_usable(int d) { return d >= 1 && d <= 64; }

static int p4_ns(int dp) { return (dp + 15) / 16; }
static int p4_nt(int dp) { return (((dp + 6 + 15) / 16) + 1) / 2; }
static int p4_ne(int dp) { return (dp + 31) / 32; }
static int p4_nlt(int dp) { return p4_ns(dp) + (p4_ne(dp) > 1 ? p4_ns(dp) - 2 : 0); }

size_t prep4_ltf_count(int dp) { return (size_t)2 * p4_nlt(dp) * 512; }   // binary16 values: hi block then lo block
size_t prep4_ttf_count(int dp) { return (size_t)2 * p4_nt(dp) * p4_ns(dp) * 512; }

namespace {
// split s * v into two binary16 pieces; returns the representation error s v - hi - lo
double split16(double v, double s, _Float16 *hi, _Float16 *lo) {
  const double x = v * s;
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (double)h);
  *hi = h;
  *lo = l;
  return x - (double)h - (double)l;
}
}  // namespace

// fragments of (s_L L)^T: tile t (rows 32 t ..), k-steps s >= 2 t; lane (i, hh) holds (L^T)[32 t + i][16 s + 8 hh + j].
// Returns |E|_F (representation error of the scaled matrix).
double prep4_lt_fragments(const double *L, int d, int dp, double scale, uint16_t *out_bits) {
  _Float16 *out = reinterpret_cast<_Float16 *>(out_bits);
  const int ns = p4_ns(dp), ne = p4_ne(dp), nlt = p4_nlt(dp);
  double e2 = 0.0;
  size_t f = 0;
  for (int t = 0; t < ne; ++t)
    for (int s = 2 * t; s < ns; ++s, ++f)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int row = 32 * t + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
          const double v = (row < d && k < d) ? L[(size_t)k * d + row] : 0.0;   // (L^T)[row][k]
          _Float16 hi, lo;
          const double e = split16(v, scale, &hi, &lo);
          e2 += e * e;
          out[(f * 64 + l) * 8 + j] = hi;
          out[((size_t)nlt * 64 + f * 64 + l) * 8 + j] = lo;
        }
  return std::sqrt(e2);
}

// fragments of (s_T T)^T with the rows in filter-column order: tile t, k-step s; lane (i, hh) holds T[16 s + 8 hh + j][col(t, i)]
double prep4_t_fragments(const double *T, int d, int dp, double scale, uint16_t *out_bits) {
  _Float16 *out = reinterpret_cast<_Float16 *>(out_bits);
  const int ns = p4_ns(dp), nt = p4_nt(dp);
  double e2 = 0.0;
  for (int t = 0; t < nt; ++t)
    for (int s = 0; s < ns; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int col = p4_column(t, l & 31), k = 16 * s + 8 * (l >> 5) + j;
          const double v = (col < d && k < d) ? T[(size_t)k * d + col] : 0.0;
          _Float16 hi, lo;
          const double e = split16(v, scale, &hi, &lo);
          e2 += e * e;
          const size_t f = (size_t)t * ns + s;
          out[(f * 64 + l) * 8 + j] = hi;
          out[(((size_t)nt * ns + f) * 64 + l) * 8 + j] = lo;
        }
  return std::sqrt(e2);
}

hipError_t launch_prep4(const Prep4Args &a, hipStream_t s) {
  if (a.np <= 0) return hipSuccess;
  if (!prep4_usable(a.d) || a.dp < a.d || (a.dp & 1)) return hipErrorInvalidValue;
  const long long groups = a.do_tr ? a.nqpad / 32 : (a.np + 31) / 32;
  switch (a.dp) {
#define X(D)                                                                                                     \
  case D: {                                                                                                      \
    constexpr size_t lds = P4<D>::lds_bytes();                                                                   \
    static DeviceGrant grant;                                                                                    \
    if (hipError_t e = grant.ensure([] {                                                                         \
          return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prep4<D>),                                \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        }))                                                                                                      \
      return e;                                                                                                  \
    if (a.do_tr && a.ks != P4<D>::KS) return hipErrorInvalidValue;                                               \
    long long grid = (groups + P4<D>::NW - 1) / P4<D>::NW;                                                       \
    if (grid > 256) grid = 256;                                                                                  \
    hipLaunchKernelGGL((k_prep4<D>), dim3((unsigned)grid), dim3(64 * P4<D>::NW), lds, s, a);                     \
    break;                                                                                                       \
  }
    MLF_FOR_EACH_DP_PREP4(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------- exact ellipsoid test ----------
// (body: mlf_ell_exact.hpp -- it also runs as the tail of the second-stage scan launch)
__global__ __launch_bounds__(256) void k_ell_exact(EllExactArgs a) {
  extern __shared__ __attribute__((aligned(16))) double ltl[];
  ell_exact_body(a, ltl, blockIdx.x, gridDim.x);
}

void launch_ell_exact(const EllExactArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_ell_exact, dim3(kEllBlocks), dim3(256), ell_exact_lds(a.d), s, a);
}

// ---------------------------------------------------------------- exact re-check, whitening its own queries
// The pairs the pre-filter could not decide are re-evaluated in the reference's arithmetic.  k_prep4 stores no
// whitened coordinates, so the wave that owns a list segment whitens the few distinct queries of that segment itself,
// in exactly the arithmetic that whitened the live points (k_prep: delta_k = x_k - c_k, k-ascending binary64 FMA chain
// per output; also what k_prep3 / v_mfma_f64 compute), keeps them in LDS and then runs the reference's distance loop
// (acc = 0; k ascending: diff = a_k - t_k; acc += diff * diff, no FMA) per listed pair.
//   1. every live entry (query not yet settled by a certain hit at or below the pair's live index) inserts its query
//      into an LDS hash table (open addressing; a segment holds pairs of at most `unit_cap` distinct queries),
//   2. the occupied slots are numbered, 3. in rounds of kTQ queries: the wave whitens them one after the other (lane =
//      output coordinate, the matrix row T[k][.] is one coalesced load), then evaluates the entries of these queries.
// One wave = one workgroup, so that (almost) all segments of a batch are resident at once: the work per segment is a
// chain of ~8 dependent memory round trips, and what matters is how many of those chains run concurrently.  A first
// version numbered the queries of ALL segments globally (claim kernel, scan, whitening kernel, re-check kernel:
// 24 + 10 + 21 + 21 us per 10^6-proposal batch, almost all of it launch and dependent-load latency), a second one
// used 256-thread workgroups (74 us: a quarter of the segments resident).  The last waves of the launch decide the
// ellipsoid band (ell_exact_wave).
__global__ __launch_bounds__(64) void k_recheck_whiten(RecheckWArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_r[];
  const int lane = threadIdx.x;
  // the ellipsoid band goes FIRST in the grid: its waves (a few proposals each, 50-step chains) then run next to the
  // segment waves instead of after the last of them (44 -> 41 us)
  const unsigned nell = a.ell.count ? a.ell_waves : 0u;
  if (blockIdx.x < nell) {
    ell_exact_wave(a.ell, lds_r, blockIdx.x, nell);
    return;
  }
  const unsigned sidx = blockIdx.x - nell;
  recheck_segment(a, a.list + (size_t)sidx * a.seg_cap, a.seg_count[sidx], lds_r, lane);
}

void launch_recheck_whiten(const RecheckWArgs &a_in, hipStream_t s) {
  if (a_in.nsegs <= 0) return;
  RecheckWArgs a = a_in;
  const size_t lds = recheck_w_lds(a.d);
  if (lds > 48 * 1024) {   // (not reached for d <= 128: 12 KiB) the whole LDS, once per device
    static DeviceGrant grant;
    (void)grant.ensure([] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_recheck_whiten), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    });
  }
  if (a.ell_waves == 0u) a.ell_waves = kEllWaves;
  const unsigned grid = (unsigned)a.nsegs + (a.ell.count ? a.ell_waves : 0u);
  hipLaunchKernelGGL(k_recheck_whiten, dim3(grid), dim3(64), lds, s, a);
}

}  // namespace mlf
