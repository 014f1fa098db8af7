// mlf_filter.hip -- exactness-preserving MFMA pre-filter for the neighbour scan (K1 / R3).
//
// The exact scan (mlf_scan.hip) is bound by the non-fused FP64 vector rate: 3 v_*_f64 per
// (live point, query, coordinate).  Almost all of the 4e9 pair tests of a 10^6-proposal batch are
// nowhere near the threshold r2, so they can be DECIDED with a much cheaper bound and only the
// few pairs that fall inside a rigorous uncertainty band are re-evaluated with the reference's
// exact binary64 arithmetic.  The final answers (masks, first-hit indices) are bit-identical to
// the exact scan.
//
//   1. live points and queries are centred (c = mean live point), scaled by a power of two
//      sigma (|sigma*(a-c)| <= 1) and rounded to binary16:  ah = f16(sigma (a-c)),
//      bh = f16(sigma (b-c)).
//   2. one v_mfma_f32_32x32x16_f16 chain per 32x32 block of pairs gives
//         Dt = |ah|^2 + |bh|^2 - 2 ah.bh     (f32 accumulate)
//      The two squared norms ride along as spare K columns (each split into three f16 pieces
//      against a column of ones), and the factor -2 is folded into the query operand, so the
//      matrix core emits Dt directly and the epilogue is two compares per pair.
//   3. per query two thresholds (derivation: DESIGN.md section 4b)
//         Dt <= T_lo  =>  reference distance <= r2   (certain hit)
//         Dt >  T_hi  =>  reference distance >  r2   (certain miss)
//      with  lo = sigma*sqrt(r2)(1-2^-30) - Delta,  hi = sigma*sqrt(r2)(1+2^-30) + Delta,
//            Delta = 2^-11(1+2^-9)(max|sigma a'| + |sigma b'|) + 2 sqrt(K) 2^-24 + 2^-40(...)   [input rounding]
//            T_lo = lo^2 - Eacc,  T_hi = hi^2 + Eacc,  Eacc = 2^-15 (|ah|max+|bh|)^2 + 2^-22      [f32 accumulation]
//   4. pairs with T_lo < Dt <= T_hi (or NaN) are appended to a list and re-evaluated exactly
//      (k_recheck: the reference's sequential sub/mul/add in binary64, no FMA).
//   Queries whose scaled coordinates do not fit binary16 are routed to the exact scan kernel;
//   if the list overflows, the exact scan kernel redoes every filtered query (device-side flag,
//   no host round trip).
//
// Fragment layout for v_mfma_f32_32x32x16_f16 (A = live points, M x K; B = queries, K x N):
//   lane l holds 8 consecutive k (k = 8*(l>>5) + j) of row / column (l & 31); C[row][col] with
//   col = l & 31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Both operands are stored
//   FRAGMENT-MAJOR in HBM ([tile32][kstep][lane][8 halves]) so that every wave-wide load is one
//   contiguous, perfectly coalesced 1 KiB read.
#include "mlf_filter.hpp"
#include "mlf_filter_dev.hpp"
#include "mlf_sample.hpp"

#include <math.h>

namespace mlf {

typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float float16v;

// ---------------------------------------------------------------- live-point statistics -----
// centre c[k] = mean_i a_ik, amax = max |a_ik - c_k|, sigma = 2^-ceil(log2 amax), namax = max_i |sigma (a_i - c)|.
// stats: [0]=sigma [1]=namax [2]=amax [3]=finite flag; c follows at [8..].
// Three short launches over many workgroups (a single-workgroup version of the same passes took 0.2 ms: one CU,
// whatever its inner loops looked like).  Everything is deterministic: partial sums are added in a fixed order by
// every consumer, maxima are order independent.  scratch: kStatBlocks * 128 doubles + 2 u64.
constexpr int kStatBlocks = 64;

// rows i = block (mod kStatBlocks) x wave (mod 4): column sums of this workgroup -> scratch[block][128]
__global__ __launch_bounds__(256) void k_ref_colsum(const double *__restrict__ refR, int n, int d, int dp,
                                                    double *__restrict__ scratch) {
  __shared__ double part[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane < d ? lane : 0, c1 = lane + 64 < d ? lane + 64 : 0;
  double s0 = 0.0, s1 = 0.0;
  for (int i = blockIdx.x * 4 + wave; i < n; i += 4 * kStatBlocks) {
    s0 += refR[(size_t)i * dp + c0];
    s1 += refR[(size_t)i * dp + c1];
  }
  part[wave][lane] = s0;
  part[wave][lane + 64] = s1;
  __syncthreads();
  if (threadIdx.x < 128)
    scratch[blockIdx.x * 128 + threadIdx.x] =
        (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// centre from the partial sums (fixed order), then this workgroup's rows: largest |a_ik - c_k| and largest
// |a_i - c|^2 as bit patterns (non-negative doubles order like their bit patterns; NaN / inf -> all ones)
__global__ __launch_bounds__(256) void k_ref_extent(const double *__restrict__ refR, int n, int d, int dp,
                                                    const double *__restrict__ scratch,
                                                    unsigned long long *__restrict__ maxima, double *__restrict__ stats,
                                                    unsigned long long *__restrict__ keys) {
  __shared__ double cc[128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 128) {
    double tot = 0.0;
    for (int b = 0; b < kStatBlocks; ++b) tot += scratch[b * 128 + threadIdx.x];
    cc[threadIdx.x] = tot / (double)n;
    if (blockIdx.x == 0 && threadIdx.x < d) stats[8 + threadIdx.x] = cc[threadIdx.x];
  }
  __syncthreads();
  const bool h0 = lane < d, h1 = lane + 64 < d;
  const int c0 = h0 ? lane : 0, c1 = h1 ? lane + 64 : 0;
  const double m0 = cc[c0], m1 = cc[c1];
  double amax = 0.0, n2max = 0.0;
  bool finite = true;
  for (int i = blockIdx.x * 4 + wave; i < n; i += 4 * kStatBlocks) {
    const double v0 = h0 ? refR[(size_t)i * dp + c0] - m0 : 0.0;
    const double v1 = h1 ? refR[(size_t)i * dp + c1] - m1 : 0.0;
    if (!(fabs(v0) <= 1.7e308) || !(fabs(v1) <= 1.7e308)) finite = false;
    amax = fmax(amax, fmax(fabs(v0), fabs(v1)));
    double sq = v0 * v0 + v1 * v1;
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    n2max = fmax(n2max, sq);
    // |a_i - c|^2 is also the key of the mask-mode operand's order (k_ref_rank): non-negative doubles order like their bit
    // patterns, a NaN sorts last
    if (keys && lane == 0) keys[i] = (unsigned long long)__double_as_longlong(sq);
  }
  for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
  const bool all_finite = __all(finite) && amax <= 1.7e308 && n2max <= 1.7e308;
  if (lane == 0) {
    atomicMax(&maxima[0], all_finite ? (unsigned long long)__double_as_longlong(amax) : ~0ull);
    atomicMax(&maxima[1], all_finite ? (unsigned long long)__double_as_longlong(n2max) : ~0ull);
  }
}

__global__ void k_ref_finish(unsigned long long *maxima, double *stats) {
  const unsigned long long a = maxima[0], q = maxima[1];
  maxima[0] = maxima[1] = 0ull;   // ready for the next set of live points
  const bool finite = a != ~0ull && q != ~0ull;
  const double amax_all = finite ? __longlong_as_double((long long)a) : INFINITY;
  double sigma = 1.0;
  if (amax_all > 0.0 && amax_all < 1e300) {
    int e;
    frexp(amax_all, &e);  // amax = m * 2^e, m in [0.5, 1)  ->  sigma*amax in [0.5, 1)
    sigma = ldexp(1.0, -e);
  }
  // sigma is a power of two: |sigma (a_i - c)| = sigma |a_i - c| exactly; 1e-12 covers the rounding of the row sums
  const double nmax = finite ? sigma * sqrt(__longlong_as_double((long long)q)) : INFINITY;
  stats[0] = sigma;
  stats[1] = nmax * (1.0 + 1e-12);
  stats[2] = amax_all;
  stats[3] = (amax_all < 1e300) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------- live points -> f16 fragments
// one thread per live-point row (rows >= n are sentinels that can never be hit)
// Both operands in ONE launch: blockIdx.y = 0 the storage-order operand refF (first-index mode), blockIdx.y = 1 the mask-mode
// operand refFm -- slot i holds storage row perm[i] -- together with the rows in that order (rows_out: what the exact re-check
// of the mask-mode kernels reads; slots n .. nrows_out - 1 are zero rows).  SIXTEEN lanes per row, lane l the 16-byte pieces
// (8 columns) l and l + 16: eight loads issued together, the norm summed over the row's lanes, one store per piece.  (Rounds
// 1-4: one thread per row, one 2-byte store per element behind one load at a time: 16 us for 4000 rows.)
struct half8pack {
  half_t h[8];
};
__global__ __launch_bounds__(256) void k_quant_refs(const double *refR, int n, int npad32, int d, int dp, int ks,
                                                    const double *stats, half_t *refF, half_t *refFm, const int *perm, double *rows_out,
                                                    int nrows_out) {
  const bool ordered = blockIdx.y != 0;
  if (ordered && !refFm) return;
  half_t *dst = ordered ? refFm : refF;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (ordered && rows_out && i >= n && i < nrows_out)
    for (int k = sub; k < dp; k += 16) rows_out[(size_t)i * dp + k] = 0.0;
  if (i >= npad32) return;   // whole 16-lane groups leave together
  const int K = ks * 16;
  const double sigma = stats[0];
  const bool have = i < n;
  const int src = (ordered && have) ? perm[i] : (have ? i : 0);
  const double *row = refR + (size_t)src * dp;
  double v[2][8];
  double na = 0.0;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k0 = 8 * (sub + 16 * u);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[u][j] = (have && k0 + j < d) ? row[k0 + j] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k0 = 8 * (sub + 16 * u);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (have && k0 + j < d) {
        const half_t h = (half_t)(float)(sigma * (v[u][j] - stats[8 + k0 + j]));
        const double hv = (double)(float)h;
        na += hv * hv;  // exact: 22-bit products, <= 128 terms -- any order of the sum gives the same bits
      }
  }
  for (int o = 8; o > 0; o >>= 1) na += __shfl_xor(na, o, 16);
  half_t p[3];
  if (have) {
    split3(na, p);
  } else {
    p[0] = (half_t)60000.0f;  // sentinel row: Dt >= 60000 > every admissible T_hi
    p[1] = p[2] = (half_t)0.0f;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k0 = 8 * (sub + 16 * u);
    if (k0 >= K) continue;
    half8pack pk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      half_t h = (half_t)0.0f;
      if (k < d) {
        if (have) h = (half_t)(float)(sigma * (v[u][j] - stats[8 + k]));
      } else if (k < d + 3) {
        h = p[k - d];               // x 1 in the queries
      } else if (k < d + 6) {
        h = (half_t)1.0f;           // x |bh|^2 pieces
      }
      pk.h[j] = h;
    }
    *reinterpret_cast<half8pack *>(dst + frag_index(i, k0, ks)) = pk;   // 16-byte aligned: k0 is a multiple of 8
    if (ordered && rows_out && have)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < dp) rows_out[(size_t)i * dp + k0 + j] = k0 + j < d ? v[u][j] : 0.0;
  }
}

// ---------------------------------------------------------------- order of the mask-mode operand
// MLFriends.inside needs "is there ANY live point within the radius" (mlfriends.pyx:1186-1211); only find_nearby reports the
// FIRST index (:176-183).  The mask-mode kernels therefore sweep a PERMUTED copy of the live points -- nearest to the
// centre first: a proposal's nearest live points are, far more often than not, the central ones (|x - a|^2 = |x|^2 + |a|^2 -
// 2 x.a), so the first tile range of a two-range sweep decides more proposals (C5 set E, scripts/order_study.py: 75 % after
// half of the tiles against 62 % in storage order; 65 % against 46 % after 30 %).  The first-index operand keeps storage order.
// key = bit pattern of |a_i - c|^2, written by k_ref_extent on its way (always a permutation: bit patterns are totally ordered)
// rank by counting: slot of row i = #{j : key_j < key_i or (key_j == key_i and j < i)}; perm[slot] = i.  Workgroup = 64 rows
// x 16 slices of the keys (wave w walks slice w: every lane reads the same key -- an LDS broadcast)
constexpr int kRankSlices = 16;
__global__ __launch_bounds__(64 * kRankSlices) void k_ref_rank(const unsigned long long *__restrict__ keys, int n, int *__restrict__ perm) {
  __shared__ unsigned long long kb[2048];
  __shared__ unsigned part[kRankSlices][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const unsigned long long mine = i < n ? keys[i] : ~0ull;
  unsigned count = 0;
  for (int base = 0; base < n; base += 2048) {
    __syncthreads();
    for (int e = threadIdx.x; e < 2048; e += 64 * kRankSlices) kb[e] = base + e < n ? keys[base + e] : ~0ull;
    __syncthreads();
    // slots past n hold ~0 with an index past every row: they never count.  Eight keys per step, their LDS reads issued
    // together (one read per step waited out its own latency: 66 us for 4000 keys with 4 slices)
    for (int e = wave; e < 2048; e += 8 * kRankSlices) {
      unsigned long long kj[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) kj[q] = kb[e + kRankSlices * q];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = base + e + kRankSlices * q;
        count += (kj[q] < mine || (kj[q] == mine && j < i)) ? 1u : 0u;
      }
    }
  }
  part[wave][lane] = count;
  __syncthreads();
  if (wave == 0 && i < n) {
    unsigned slot = 0;
#pragma unroll
    for (int w = 0; w < kRankSlices; ++w) slot += part[w][lane];
    perm[slot] = i;
  }
}

// ---------------------------------------------------------------- queries -> f16 fragments ---
// one thread per query.  route: 0 = not scanned (gated out), 1 = filtered, 2 = exact scan only.
// SIXTEEN lanes per query, lane l the 16-byte pieces (8 columns) l and l + 16: the loads of a row are issued together, the
// sums meet through lane exchanges, one store per piece (rounds 1-4: one thread per query, one 2-byte store per element behind
// one load at a time: 1.3-2.2 ms per 10^6 x 100 in front of a 0.5 ms sweep).  |bh|^2 is a sum of exact products (any order gives
// the same bits); |sigma (b - c)|^2 enters the thresholds as a bound with explicit slack (2^-40 relative and more: filter_thresholds),
// far above what the order of a binary64 sum of <= 144 terms can move
__global__ __launch_bounds__(256) void k_quant_queries(const double *q, long long ldq, long long nq, long long nqpad,
                                                       int d_src, int d, int ks, const double *stats, double r2,
                                                       const uint8_t *gate, half_t *qF, float *tlo, float *thi,
                                                       uint8_t *route, int *best, unsigned *counters) {
  // d_src: coordinates present in q; d (>= d_src): filter dimensionality (zero padded), the norm /
  // ones columns sit at d .. d+5 in both operands
  const long long p = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counters[0] = 0;
    counters[1] = 0;  // overflow flag
  }
  if (p >= nqpad) return;   // whole 16-lane groups leave together
  const int K = ks * 16;
  const double sigma = stats[0], namax = stats[1];
  int rt = 0;
  if (p < nq && (gate == nullptr || gate[p])) rt = 1;
  double x[2][8];
  double nb = 0.0, nbn2 = 0.0;
  bool fits = true;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k0 = 8 * (sub + 16 * u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      x[u][j] = (rt == 1 && k < d) ? sigma * ((k < d_src ? q[p * ldq + k] : 0.0) - stats[8 + k]) : 0.0;
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!(fabs(x[u][j]) <= 16000.0)) fits = false;  // -2x must stay well inside binary16; NaN lands here too
      nbn2 += x[u][j] * x[u][j];
      const double hv = (double)(float)(half_t)(float)x[u][j];
      nb += hv * hv;
    }
  for (int o = 8; o > 0; o >>= 1) {
    nbn2 += __shfl_xor(nbn2, o, 16);
    nb += __shfl_xor(nb, o, 16);
    fits = fits && (__shfl_xor(fits ? 1 : 0, o, 16) != 0);
  }
  if (rt == 1 && (!fits || !(nbn2 <= 30000.0))) rt = 2;
  half_t pc[3] = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
  float lo_f = -1.0f, hi_f = -1.0f;
  if (rt == 1) {
    split3(nb, pc);
    if (!filter_thresholds(sigma, namax, nbn2, r2, K, &lo_f, &hi_f)) {
      rt = 2;
      lo_f = hi_f = -1.0f;
    }
  }
  const size_t gbase = (size_t)(p >> 5) * ((size_t)ks * 512);
  const int pr = (int)(p & 31);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k0 = 8 * (sub + 16 * u);
    if (k0 >= K) continue;
    half8pack pk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      half_t h = (half_t)0.0f;
      if (rt == 1) {
        if (k < d)
          h = (half_t)(-2.0f * (float)(half_t)(float)x[u][j]);  // exact
        else if (k < d + 3)
          h = (half_t)1.0f;
        else if (k < d + 6)
          h = pc[k - d - 3];
      }
      pk.h[j] = h;
    }
    *reinterpret_cast<half8pack *>(qF + gbase + frag_index(pr, k0, ks)) = pk;
  }
  if (sub == 0) {
    tlo[p] = lo_f;
    thi[p] = hi_f;
    if (p < nq) {
      route[p] = (uint8_t)rt;
      best[p] = kNone;
    }
  }
}

// Minimum of three MFMA results through their bit patterns (v_min3_i32): fminf() makes hipcc
// canonicalise every operand first (one extra v_max per value), and an inline-asm v_min3_f32 would
// read the MFMA result without the wait states the compiler only inserts for its own instructions.
// Signed-integer order equals float order for values >= 0; a (tiny, rounding-induced) negative value
// has a negative pattern and so still wins the minimum, which is all the threshold tests need.
__device__ __forceinline__ int min3i(int a, int b, int c) {
  const int m = a < b ? a : b;
  return m < c ? m : c;
}

// ---------------------------------------------------------------- the MFMA filter ------------
// FIRST-INDEX mode (find_nearby: the lowest live index within r2, mlfriends.pyx:176-183).  One wave owns QW groups of
// 32 queries (B fragments resident in registers) and sweeps ALL live-point tiles; a workgroup is 4 independent waves.
// The mask mode of the same sweep (MLFriends.inside, where any hit decides) is k_sweep in mlf_sweep.hip.
template <int KS, int QW, bool COMPACT>
__global__ __launch_bounds__(256) void k_filter(FilterArgs a) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long g0 = wave * QW;  // first query group of this wave
  const long long nslots = a.nslots_dev ? (long long)*a.nslots_dev : -1;
  const long long ngroups = nslots >= 0 ? (nslots + 31) / 32 : (a.ngroups_dev ? (long long)*a.ngroups_dev : a.ngroups);
  // gridDim.y > 1 (batches too small to fill the chip with one wave per 4 query groups; never a compacting launch): the
  // live-point tiles are split into gridDim.y ranges, one wave per (query groups, range), each with its own list segment
  const int sub = COMPACT ? 0 : (int)blockIdx.y, nsub = COMPACT ? 1 : (int)gridDim.y;
  const long long seg = wave + (long long)sub * gridDim.x * 4;
  if (!a.append && a.seg_extra > 0 && lane == 0 && sub == 0)   // segments only a later, narrower launch of this batch uses start empty
    for (long long i = wave; i < a.seg_extra; i += (long long)gridDim.x * 4) a.seg_count[a.seg_first_extra + i] = 0u;
  if (g0 >= ngroups) {
    if (lane == 0 && !a.append) a.seg_count[seg] = 0;
    return;
  }

  const half8 *qF = reinterpret_cast<const half8 *>(a.qF);
  const half8 *refF = reinterpret_cast<const half8 *>(a.refF);

  half8 bq[QW][KS];
  float tlo[QW], thi[QW];
  int first[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const long long grp = (g0 + g < ngroups) ? g0 + g : ngroups - 1;  // clamp (results discarded)
#pragma unroll
    for (int s = 0; s < KS; ++s) bq[g][s] = qF[((size_t)grp * KS + s) * 64 + lane];
    const long long qi = grp * 32 + (lane & 31);
    // an unpadded last group: the slots past the count hold whatever an earlier batch left there.  Thresholds -1 alone do
    // not silence them (stale operands can give Dt <= -1, a "certain hit" reported to a stale query number): their
    // operand is zeroed as well, so Dt = 0 exactly as in a padded group
    const bool have = g0 + g < ngroups && (nslots < 0 || qi < nslots);
    tlo[g] = have ? a.tlo[qi] : -1.0f;
    thi[g] = have ? a.thi[qi] : -1.0f;
    if (nslots >= 0 && !have) {
#pragma unroll
      for (int s = 0; s < KS; ++s) bq[g][s] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    first[g] = kNone;
  }
  const int rowbase = 4 * (lane >> 5);
  unsigned cursor = a.append ? a.seg_count[seg] : 0u;            // wave-uniform
  unsigned long long *seglist = a.list + (size_t)seg * a.seg_cap;   // this wave's list segment

  half8 af[KS];
  // Waves start at different live-point tiles (results are order independent): all waves
  // sweeping the same 4 KB tile at the same moment would queue on one L2 channel.
  const int ntl_all = a.tile1 - a.tile0;   // tiles of this phase
  const int tile0 = a.tile0 + (int)((long long)ntl_all * sub / nsub);        // ... and of this wave's range of it
  const int tile1 = a.tile0 + (int)((long long)ntl_all * (sub + 1) / nsub);
  const int ntl = tile1 - tile0;
  const int tstart = tile0 + (int)(((long long)blockIdx.x * 37) % ntl);   // one sweep order per workgroup: its 4 waves share L1 lines
#pragma unroll
  for (int s = 0; s < KS; ++s) af[s] = refF[((size_t)tstart * KS + s) * 64 + lane];

  for (int it = 0; it < ntl; ++it) {
    int t = tstart + it;
    if (t >= tile1) t -= ntl;
    int tn = t + 1;
    if (tn >= tile1) tn = tile0;
    half8 an[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) an[s] = refF[((size_t)tn * KS + s) * 64 + lane];  // prefetch

    // k-step-major issue order: QW independent accumulator chains are in flight together
    float16v acc[QW];
#pragma unroll
    for (int g = 0; g < QW; ++g)
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
          af[0], bq[g][0], (float16v){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
          0, 0, 0);
#pragma unroll
    for (int s = 1; s < KS; ++s)
#pragma unroll
      for (int g = 0; g < QW; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bq[g][s], acc[g], 0, 0, 0);

    // All 16 values of a lane belong to ONE query (column = lane & 31) and 16 live points, so the lane-wise minimum
    // decides the common case with 8 v_min3 + 1 compare per group: vmin > T_hi -- nothing within reach in this block
    // (certain misses).  Operands are finite binary16 values of bounded size, so acc holds no NaN / inf.
    float vmin[QW];
    unsigned long long candm[QW];
    unsigned long long need = 0ull;
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      const float16v &c = acc[g];
      const int m0 = min3i(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]));
      const int m1 = min3i(__float_as_int(c[3]), __float_as_int(c[4]), __float_as_int(c[5]));
      const int m2 = min3i(__float_as_int(c[6]), __float_as_int(c[7]), __float_as_int(c[8]));
      const int m3 = min3i(__float_as_int(c[9]), __float_as_int(c[10]), __float_as_int(c[11]));
      const int m4 = min3i(__float_as_int(c[12]), __float_as_int(c[13]), __float_as_int(c[14]));
      vmin[g] = __int_as_float(min3i(min3i(m0, m1, m2), min3i(m3, m4, __float_as_int(c[15])), m0));
      candm[g] = __ballot(vmin[g] <= thi[g]);
      need |= candm[g];
    }
    if (need != 0ull) {   // wave-uniform: an uncertain pair (first-index mode: or a certain hit)
#pragma unroll
      for (int g = 0; g < QW; ++g) {
        if (candm[g] == 0ull) continue;
        {
          // a query with a certain hit below this tile cannot get a lower first index here, and its uncertain
          // pairs in this tile cannot matter either (without this every later tile of an accepted proposal went
          // through the detail path: first-index batches with many hits ran slower than the exact scan)
          const int other = __shfl_xor(first[g], 32);
          const int fb = first[g] < other ? first[g] : other;
          candm[g] &= ~__ballot(fb < t * 32);
        }
        if (candm[g] == 0ull) continue;
        const float16v &c = acc[g];
        const bool detail = (candm[g] >> lane) & 1ull;
        if (detail) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = c[r];
            const int idx = t * 32 + rowbase + (r & 3) + 8 * (r >> 2);
            if (v <= tlo[g]) first[g] = idx < first[g] ? idx : first[g];
          }
        }
        // uncertainty band: append (query, live point) to this wave's PRIVATE list segment for the
        // exact re-check.  No atomics: one global counter saturates near 90 M increments/s and cost
        // 2.5 ms per 10^6-proposal batch in the first version.
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = c[r];
          const bool band = detail && !(v <= tlo[g]) && (v <= thi[g]);
          const unsigned long long bm = __ballot(band);
          if (bm != 0ull) {
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
            const unsigned slot = cursor + below;
            if (band && slot < a.seg_cap) {
              const int idx = t * 32 + rowbase + (r & 3) + 8 * (r >> 2);
              const long long slot_q = (g0 + g) * 32 + (lane & 31);
              const long long qi = a.qmap ? (long long)a.qmap[slot_q] : slot_q;
              seglist[slot] = qi >= 0 ? (((unsigned long long)qi << 32) | (unsigned)idx) : ~0ull;
            }
            cursor += (unsigned)__popcll(bm);
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) af[s] = an[s];
  }

  unsigned keepm[QW];   // COMPACT: queries of each group that stay in the sweep (bit = query row)
  int qid[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    keepm[g] = 0u;
    qid[g] = -1;
    if (g0 + g >= ngroups) continue;
    const long long slot_q = (g0 + g) * 32 + (lane & 31);
    const long long qi = a.qmap ? (long long)a.qmap[slot_q] : slot_q;
    const int other = __shfl_xor(first[g], 32);
    const int res = first[g] < other ? first[g] : other;
    if (lane < 32 && qi >= 0 && qi < a.nq && res != kNone) {
      if (nsub > 1)
        atomicMin(a.best + qi, res);   // the waves of the other tile ranges report too
      else
        a.best[qi] = res;
    }
    if (COMPACT) {
      // route == 1 <=> the query has thresholds (T_hi > 0; the stages in front write -1 for every other route, and a
      // certain hit lowered it to -inf): no dependent load of the route byte at the end of the wave's life
      const bool keep = qi >= 0 && qi < a.nq && res == kNone && thi[g] > 0.0f;
      keepm[g] = (unsigned)__ballot(keep);   // low half; lanes l and l + 32 agree
      qid[g] = (int)qi;
    }
  }
  if (COMPACT) {
    unsigned total = 0;
#pragma unroll
    for (int g = 0; g < QW; ++g) total += (unsigned)__popc(keepm[g]);
    unsigned base = 0;
    if (total != 0u) {   // wave-uniform
      if (lane == 0) base = atomicAdd(a.ccount, total);
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
      uint4 *dst = reinterpret_cast<uint4 *>(a.cq);
      const unsigned row = (unsigned)(lane & 31);
#pragma unroll
      for (int g = 0; g < QW; ++g) {
        if ((keepm[g] >> row) & 1u) {
          const unsigned rank = base + (unsigned)__popc(keepm[g] & ((1u << row) - 1u));
          const size_t gd = rank >> 5;
          const unsigned rd = (rank & 31u) + (unsigned)(lane & 32);
          if (rank < a.ccap) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              union { half8 h; uint4 u; } cv;
              cv.h = bq[g][s];
              dst[(gd * KS + s) * 64 + rd] = cv.u;
            }
          }
          if (lane < 32 && rank < a.ccap) {
            a.ctlo[rank] = tlo[g];
            a.cthi[rank] = thi[g];
            a.cmap[rank] = qid[g];
          }
        }
        base += (unsigned)__popc(keepm[g]);
      }
    }
  }
  if (lane == 0) {
    a.seg_count[seg] = cursor < a.seg_cap ? cursor : a.seg_cap;
    if (cursor > a.seg_cap) a.counters[1] = 1u;   // overflow: the exact scan redoes the batch
  }
}

// ---------------------------------------------------------------- phase compaction -------------
// In mask mode a query with a certain hit is decided, in first-index mode every later tile can only
// give a larger index: either way it leaves the sweep.  With 1-3 neighbours per accepted proposal the
// first hit sits anywhere in the live set, so splitting the sweep into phases and compacting the
// undecided queries in between removes ~40 % of the matrix work at N = 4000.  The compaction itself rides in the
// epilogue of the matrix kernels (COMPACT instances of k_filter / k_sweep).

// One wave after a compacting k_filter launch: group count, padding of the last group, counter reset.
__global__ __launch_bounds__(64) void k_phase_finish(uint4 *cq, float *ctlo, float *cthi, int *cmap, unsigned *ccount,
                                                     unsigned *ngroups_dst, int ks) {
  const unsigned total = *ccount;   // never above the capacity: one slot per undecided query of the source set
  const unsigned ngroups = (total + 31u) / 32u;
  const unsigned slot = total + threadIdx.x;
  if (threadIdx.x < 32 && slot < 32u * ngroups) {
    const size_t gd = slot >> 5;
    const unsigned rd = slot & 31u;
    for (int s = 0; s < ks; ++s) {
      cq[(gd * ks + s) * 64 + rd] = make_uint4(0u, 0u, 0u, 0u);
      cq[(gd * ks + s) * 64 + rd + 32] = make_uint4(0u, 0u, 0u, 0u);
    }
    ctlo[slot] = -1.0f;
    cthi[slot] = -1.0f;
    cmap[slot] = -1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *ngroups_dst = ngroups;
    *ccount = 0u;
  }
}

void launch_phase_finish(void *cq, float *ctlo, float *cthi, int *cmap, unsigned *ccount, unsigned *ngroups_dst,
                         int ks, hipStream_t s) {
  hipLaunchKernelGGL(k_phase_finish, dim3(1), dim3(64), 0, s, reinterpret_cast<uint4 *>(cq), ctlo, cthi, cmap, ccount,
                     ngroups_dst, ks);
}

// ---------------------------------------------------------------- exact re-check --------------
// list entry = (query << 32) | live index.  The reference's arithmetic: acc = 0; k ascending:
// diff = a[k] - b[k]; acc += diff*diff (this file is compiled with -ffp-contract=off).
__global__ __launch_bounds__(256) void k_recheck(RecheckArgs a) {
  const unsigned count = a.seg_count[blockIdx.x];
  const unsigned long long *seg = a.list + (size_t)blockIdx.x * a.seg_cap;
  for (unsigned e = threadIdx.x; e < count; e += 256) {
    const unsigned long long ent = seg[e];
    const long long qi = (long long)(ent >> 32);
    const int i = (int)(ent & 0xffffffffu);
    if (i >= a.n || qi >= a.nq) continue;
    if (a.best[qi] <= i) continue;   // a certain hit at a lower (mask mode: any) index settles this query
    const double *ar = a.refR + (size_t)i * a.dp;
    const double *br = a.q + qi * a.ldq;
    const long long ldk = a.ldk > 1 ? a.ldk : 1;
    double acc = 0.0;
#pragma unroll 10
    for (int k = 0; k < a.d; ++k) {   // loads are independent of the sum: keep ten in flight
      const double df = ar[k] - br[k * ldk];
      acc += df * df;
    }
    if (acc <= a.r2) atomicMin(&a.best[qi], i);
  }
}

// route 1 queries take their answer from best[]; route 0 = gated out; route 2 were written by the
// exact scan kernel.  If the uncertain-pair list overflowed, route 1 answers also come from the
// exact scan (second launch, gate = overflow), so nothing is written here.
__global__ void k_filter_finalize(const uint8_t *route, const int *best, const unsigned *counters,
                                  long long nq, uint8_t *out_mask, long long *out_idx, uint8_t *exact_gate,
                                  unsigned *reset_word) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0 && reset_word) *reset_word = 0u;   // slot counter of the batch: its readers ran before this launch
  if (p >= nq) return;
  const int rt = route[p];
  const bool to_scan = rt == 2 || (rt == 1 && counters[1] != 0u);
  if (exact_gate) exact_gate[p] = to_scan ? 1 : 0;   // gate of the exact scan launch that follows
  if (to_scan) return;
  const int b = best[p];
  const bool found = rt == 1 && b != kNone;
  if (out_mask) out_mask[p] = found ? 1 : 0;
  if (out_idx) out_idx[p] = rt == 0 ? -2ll : (found ? (long long)b : -1ll);
}

// gate for the exact scan kernel: which == 2 -> queries routed to the exact path;
// which == 1 -> filtered queries, but only if the list overflowed
__global__ void k_route_gate(const uint8_t *route, const unsigned *counters, long long nq, int which,
                             uint8_t *gate) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nq) return;
  const bool exact_only = route[p] == 2, redo = route[p] == 1 && counters[1] != 0u;
  gate[p] = (which == 2) ? exact_only : (which == 1 ? redo : (exact_only || redo));
}

// ---------------------------------------------------------------- launchers -------------------
void launch_ref_stats(const double *refR, int n, int d, int dp, double *stats, double *scratch, hipStream_t s, unsigned long long *keys) {
  unsigned long long *maxima = reinterpret_cast<unsigned long long *>(scratch + (size_t)kStatBlocks * 128);
  hipLaunchKernelGGL(k_ref_colsum, dim3(kStatBlocks), dim3(256), 0, s, refR, n, d, dp, scratch);
  hipLaunchKernelGGL(k_ref_extent, dim3(kStatBlocks), dim3(256), 0, s, refR, n, d, dp, scratch, maxima, stats, keys);
  hipLaunchKernelGGL(k_ref_finish, dim3(1), dim3(1), 0, s, maxima, stats);
}

void launch_quant_refs(const double *refR, int n, int npad32, int d, int dp, int ks,
                       const double *stats, void *refF, hipStream_t s, void *refFm, const int *perm, double *rows_out, int nrows_out) {
  const int rows = refFm && rows_out && nrows_out > npad32 ? nrows_out : npad32;
  hipLaunchKernelGGL(k_quant_refs, dim3((unsigned)((rows + 15) / 16), refFm ? 2u : 1u), dim3(256), 0, s, refR, n, npad32,
                     d, dp, ks, stats, reinterpret_cast<half_t *>(refF), reinterpret_cast<half_t *>(refFm), perm, rows_out, nrows_out);
}

void launch_ref_rank(const unsigned long long *keys, int n, int *perm, hipStream_t s) {
  hipLaunchKernelGGL(k_ref_rank, dim3((unsigned)((n + 63) / 64)), dim3(64 * kRankSlices), 0, s, keys, n, perm);
}

void launch_quant_queries(const double *q, long long ldq, long long nq, long long nqpad, int d_src, int d,
                          int ks, const double *stats, double r2, const uint8_t *gate, void *qF, float *tlo,
                          float *thi, uint8_t *route, int *best, unsigned *counters, hipStream_t s) {
  hipLaunchKernelGGL(k_quant_queries, dim3((unsigned)((nqpad + 15) / 16)), dim3(256), 0, s, q, ldq, nq,
                     nqpad, d_src, d, ks, stats, r2, gate, reinterpret_cast<half_t *>(qF), tlo, thi, route, best,
                     counters);
}

template <int KS, int QW>
static hipError_t launch_filter_t(const FilterArgs &a, hipStream_t s) {
  const long long waves = (a.ngroups + QW - 1) / QW;
  const dim3 grid((unsigned)((waves + 3) / 4), (unsigned)((a.cq || a.split < 1) ? 1 : a.split));
  if (a.cq)
    hipLaunchKernelGGL((k_filter<KS, QW, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_filter<KS, QW, false>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

int filter_tile_split(int ks, long long ngroups, int ntiles, int target_waves) {
  const long long waves = filter_wave_count(ks, ngroups) > 0 ? filter_wave_count(ks, ngroups) : 1;
  long long kmax = kFilterMaxSplit;
  if (kmax > ntiles / 4) kmax = ntiles / 4;   // a wave of a few thousand proposals is latency bound: 4 tiles are 64 dependent matrix instructions
  if (kmax < 1) kmax = 1;
  // up to half a round of waves (target_waves resident slots, default 2048: two waves per SIMD): as many ranges as fit ONE round
  if (target_waves / waves >= 2) return (int)(target_waves / waves > kmax ? kmax : target_waves / waves);
  // More waves than that used to sweep all tiles each, in one partly filled round (163 840 proposals at N = 4000: 1280 waves
  // of 125 tiles, 113 us against 76 us for 131 072).  Cost of a split: rounds x (a task's start and end -- 16 KiB of operand,
  // thresholds, list segment, re-check: about 10 tiles' worth -- plus its tiles), a partly filled last round counted as a
  // whole one; a larger split has to win 3 % (its extra list segments).  140 000-165 000 proposals: -8 %
  // (profiles/r05_tile_split_ab.jsonl); from 1408 waves on the choice makes no measurable difference.
  double best = 0.0;
  int best_k = 1;
  for (long long k = 1; k <= kmax; ++k) {
    const double rounds = (double)((waves * k + target_waves - 1) / target_waves);
    const double cost = rounds * (10.0 + (double)((ntiles + k - 1) / k));
    if (k == 1 || cost < best * 0.97) {
      best = cost;
      best_k = (int)k;
    }
  }
  return best_k;
}

// first = first-index mode (k_filter); otherwise the mask-mode sweep (k_sweep, mlf_sweep.hip).  narrow: two query
// groups per wave instead of four (later ranges of a phased sweep: twice the waves of half the length).
hipError_t launch_filter(int ks, const FilterArgs &a, bool first, hipStream_t s, int narrow) {
  if (a.ngroups <= 0) return hipSuccess;
  const int qw = filter_groups_per_wave(ks, narrow);
  if (!first) return launch_sweep(ks, qw, a, s);
  switch (ks * 8 + qw) {
    case 1 * 8 + 4: return launch_filter_t<1, 4>(a, s);
    case 2 * 8 + 4: return launch_filter_t<2, 4>(a, s);
    case 3 * 8 + 4: return launch_filter_t<3, 4>(a, s);
    case 4 * 8 + 4: return launch_filter_t<4, 4>(a, s);
    case 1 * 8 + 2: return launch_filter_t<1, 2>(a, s);
    case 2 * 8 + 2: return launch_filter_t<2, 2>(a, s);
    case 3 * 8 + 2: return launch_filter_t<3, 2>(a, s);
    case 4 * 8 + 2: return launch_filter_t<4, 2>(a, s);
    case 5 * 8 + 2: return launch_filter_t<5, 2>(a, s);
    case 6 * 8 + 2: return launch_filter_t<6, 2>(a, s);
    case 7 * 8 + 2: return launch_filter_t<7, 2>(a, s);
    case 8 * 8 + 2: return launch_filter_t<8, 2>(a, s);
    case 9 * 8 + 1: return launch_filter_t<9, 1>(a, s);
    default: return hipErrorInvalidValue;
  }
}

void launch_recheck(const RecheckArgs &a, long long nwaves, hipStream_t s) {
  if (nwaves <= 0) return;
  hipLaunchKernelGGL(k_recheck, dim3((unsigned)nwaves), dim3(256), 0, s, a);
}

int filter_groups_per_wave(int ks, int narrow) { return ks <= 4 ? (narrow ? 2 : 4) : (ks <= 8 ? 2 : 1); }

long long filter_wave_count(int ks, long long ngroups, int narrow) {
  const int qw = filter_groups_per_wave(ks, narrow);
  return (ngroups + qw - 1) / qw;
}

void launch_filter_finalize(const uint8_t *route, const int *best, const unsigned *counters, long long nq,
                            uint8_t *out_mask, long long *out_idx, uint8_t *exact_gate, hipStream_t s,
                            unsigned *reset_word) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(k_filter_finalize, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, route, best,
                     counters, nq, out_mask, out_idx, exact_gate, reset_word);
}

void launch_route_gate(const uint8_t *route, const unsigned *counters, long long nq, int which,
                       uint8_t *gate, hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(k_route_gate, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, route, counters, nq,
                     which, gate);
}

}  // namespace mlf
