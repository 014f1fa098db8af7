// mlf_sweep.hip -- the mask-mode sweep of the MFMA pre-filter (R3: MLFriends.inside, mlfriends.pyx:1186-1211;
// the distance test it decides is find_nearby's, mlfriends.pyx:143-183).
//
// Same mathematics as k_filter (mlf_filter.hip: operands, thresholds, list of uncertain pairs), restructured around
// what the round-2 disassembly and counters showed about the vector issue port, which the matrix instructions share
// with everything else a wave does:
//   * the lane keeps a RUNNING minimum of Dt over the whole sweep: it rides in the min3 tree of every tile for free
//     (17 values -> 8 v_min3), so a certain hit needs no per-tile bookkeeping at all (round 2: compare + select per
//     group and tile); a query is decided if its running minimum ends at or below T_lo.
//   * candidates of a tile = lanes whose running minimum sits inside (T_lo, T_hi]: two compares per group, combined on
//     the scalar unit.  The lane lists the tile's uncertain pairs and RESETS its running minimum to +inf, so it does not
//     trigger again on the same value.
//   * the uncertain pairs of a flagged lane are found per LANE (a 16-bit mask of its 16 values, slots from a prefix over
//     the flagged lanes): round 2 looped over the 16 accumulator registers with a wave-wide ballot and three branches
//     each -- 64 vector + 160 scalar instructions + 48 branches per flagged group, a third of all vector instructions
//     of the launch although only one tile in seven has a flagged lane.
//   * live-point tiles arrive by buffer loads (scalar base + scalar tile offset + constant lane offset): no 64-bit
//     vector address arithmetic per tile; the tile loop is unrolled twice over two register sets (no fragment copies);
//     the wrap-around of the staggered sweep order is one scalar compare + select.
//   * matrix instructions are issued in two stages of two query groups; the min3 tree of one stage is pinned between
//     the matrix instructions of the next (sched_group_barrier), so the wave itself keeps the matrix pipe fed while it
//     reduces.
// First-index mode (find_nearby proper) stays on k_filter.
#include "mlf_filter.hpp"
#include "mlf_filter_dev.hpp"
#include "mlf_recheck_dev.hpp"

#include <math.h>

namespace mlf {

typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float float16v;
typedef __attribute__((ext_vector_type(4))) unsigned uint4v;

namespace {

__device__ __forceinline__ int min3i(int a, int b, int c) {
  const int m = a < b ? a : b;
  return m < c ? m : c;
}

constexpr int kPosInf = 0x7f800000;

template <int KS>
__device__ __forceinline__ void load_tile(half8 (&A)[KS], __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    union { uint4v u; half8 h; } c;
    c.u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + s * 1024, 0);
    A[s] = c.h;
  }
}

// running minimum of a lane: its 16 values of this tile and what it had
__device__ __forceinline__ int tree_min(const float16v &c, int run) {
  const int m0 = min3i(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]));
  const int m1 = min3i(__float_as_int(c[3]), __float_as_int(c[4]), __float_as_int(c[5]));
  const int m2 = min3i(__float_as_int(c[6]), __float_as_int(c[7]), __float_as_int(c[8]));
  const int m3 = min3i(__float_as_int(c[9]), __float_as_int(c[10]), __float_as_int(c[11]));
  const int m4 = min3i(__float_as_int(c[12]), __float_as_int(c[13]), __float_as_int(c[14]));
  return min3i(min3i(m0, m1, m2), min3i(m3, m4, __float_as_int(c[15])), run);
}

// NM matrix instructions with NV vector instructions of the neighbouring reduction pinned between them
template <int I, int NM, int NV>
__device__ __forceinline__ void pin_step() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV * (I + 1)) / NM - (NV * I) / NM, 0);
    pin_step<I + 1, NM, NV>();
  }
}

}  // namespace

// waves per SIMD the register budget is set for: accumulators 16 QW, query operands 4 KS QW, PF + 1 tile sets of 4 KS
constexpr int sweep_waves(int ks, int qw, int pf) {
  const int need = 16 * qw + 4 * ks * qw + 4 * ks * (pf + 1) + 44;
  return (qw <= 2 && ks * qw <= 8) ? 4 : (need <= 128 ? 4 : (need <= 168 ? 3 : 2));
}

template <int KS, int QW, bool COMPACT, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sweep_waves(KS, QW, PF)))) void k_sweep(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_sw[];   // re-check scratch, one slice per wave (own re-check only)
  const int lane = threadIdx.x & 63;
  // the first launch of a batch carries the ellipsoid band of the per-proposal stage in kEllWaves leading waves
  // (the leading workgroups of every tile range of the grid: only those of range 0 have work)
  const unsigned nellblk = (a.rw.pts && a.rw.ell.count) ? kEllWaves / 4 : 0u;
  double *lds_wave = lds_sw + (size_t)(threadIdx.x >> 6) * (recheck_w_lds(a.rw.d) / sizeof(double));
  if (blockIdx.x < nellblk) {
    if (blockIdx.y == 0) ell_exact_wave(a.rw.ell, lds_wave, blockIdx.x * 4 + (threadIdx.x >> 6), kEllWaves);
    return;
  }
  const unsigned bx = blockIdx.x - nellblk, gx = gridDim.x - nellblk;   // this workgroup among the sweeping ones
  const long long wave = (long long)bx * 4 + (threadIdx.x >> 6);
  const long long g0 = wave * QW;  // first query group of this wave
  const long long nslots = a.nslots_dev ? (long long)*a.nslots_dev : -1;
  const long long ngroups = nslots >= 0 ? (nslots + 31) / 32 : (a.ngroups_dev ? (long long)*a.ngroups_dev : a.ngroups);
  const int sub = COMPACT ? 0 : (int)blockIdx.y, nsub = COMPACT ? 1 : (int)gridDim.y;
  const long long seg = wave + (long long)sub * gx * 4;
  if (!a.append && a.seg_extra > 0 && lane == 0 && sub == 0)
    for (long long i = wave; i < a.seg_extra; i += (long long)gx * 4) a.seg_count[a.seg_first_extra + i] = 0u;
  if (g0 >= ngroups) {
    if (lane == 0 && !a.append) a.seg_count[seg] = 0;
    return;
  }

  const half8 *qF = reinterpret_cast<const half8 *>(a.qF);
  half8 bq[QW][KS];
  float tlo[QW], thi[QW];
  int run[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const long long grp = (g0 + g < ngroups) ? g0 + g : ngroups - 1;  // clamp (results discarded)
#pragma unroll
    for (int s = 0; s < KS; ++s) bq[g][s] = qF[((size_t)grp * KS + s) * 64 + lane];
    const long long qi = grp * 32 + (lane & 31);
    // slots past the count of an unpadded last group: thresholds -1 AND a zero operand (Dt = 0 exactly, as in a padded group)
    const bool have = g0 + g < ngroups && (nslots < 0 || qi < nslots);
    tlo[g] = have ? a.tlo[qi] : -1.0f;
    thi[g] = have ? a.thi[qi] : -1.0f;
    if (!have) {
#pragma unroll
      for (int s = 0; s < KS; ++s) bq[g][s] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    run[g] = kPosInf;
  }
  const int rowbase = 4 * (lane >> 5);
  unsigned cursor = a.append ? a.seg_count[seg] : 0u;               // wave-uniform
  unsigned long long *seglist = a.list + (size_t)seg * a.seg_cap;   // this wave's list segment

  // tiles of this wave's range, visited from a staggered start (all waves on the same 4 KB tile at the same moment would
  // queue on one L2 channel); the 4 waves of a workgroup share the order and with it their L1 lines
  const int ntl_all = a.tile1 - a.tile0;
  const int tile0 = a.tile0 + (int)((long long)ntl_all * sub / nsub);
  const int tile1 = a.tile0 + (int)((long long)ntl_all * (sub + 1) / nsub);
  const int ntl = tile1 - tile0;
  const int tstart = tile0 + (int)(((long long)bx * 37) % ntl);
  constexpr int kTileBytes = KS * 1024;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.refF), 0, a.ntiles32 * kTileBytes, 0x00020000);
  const int voff = lane * 16;
  const int off_begin = tile0 * kTileBytes, off_end = tile1 * kTileBytes;

  float16v acc[QW];
  unsigned long long lo_m[QW], hi_m[QW];
  // matrix instructions of one stage: query groups ga (and gb) against the tile in A, k-step major
  auto mm = [&](const half8(&A)[KS], int ga, int gb) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[ga] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[ga][s], s == 0 ? z : acc[ga], 0, 0, 0);
      if (gb >= 0) acc[gb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bq[gb][s], s == 0 ? z : acc[gb], 0, 0, 0);
    }
  };
  // running minimum of group g through this tile, and where it stands against the thresholds (10 vector instructions)
  auto reduce = [&](int g) __attribute__((always_inline)) {
    run[g] = tree_min(acc[g], run[g]);
    const float rm = __int_as_float(run[g]);
    lo_m[g] = __ballot(rm <= tlo[g]);
    hi_m[g] = __ballot(rm <= thi[g]);
  };
  // lanes of group g whose running minimum fell into the band in the tile at byte offset `off`: their uncertain pairs
  // of that tile go to the list (acc[g] must still hold the tile)
  auto list_band = [&](int g, unsigned long long candm, int off) __attribute__((always_inline)) {
    const int t = off / kTileBytes;
    const float16v &c = acc[g];
    const bool flagged = (candm >> lane) & 1ull;
    unsigned bits = 0u;
    if (flagged) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = c[r];
        bits |= (!(v <= tlo[g]) && (v <= thi[g])) ? (1u << r) : 0u;
      }
      run[g] = kPosInf;   // its pairs of this tile are listed below; later tiles start afresh
    }
    const unsigned cnt = (unsigned)__popc(bits);
    // slots: exclusive prefix of cnt over the flagged lanes (usually one or two of them)
    unsigned mybase = 0u;
    for (unsigned long long m = candm; m != 0ull; m &= m - 1ull) {
      const int l = __builtin_ctzll(m);
      const unsigned cl = (unsigned)__builtin_amdgcn_readlane((int)cnt, l);
      if (lane == l) mybase = cursor;
      cursor += cl;
    }
    const long long slot_q = (g0 + g) * 32 + (lane & 31);
    long long qi = slot_q;
    if (flagged && a.qmap) qi = (long long)a.qmap[slot_q];
    while (bits != 0u) {
      const int r = __builtin_ctz(bits);
      bits &= bits - 1u;
      if (mybase < a.seg_cap) {
        const int idx = t * 32 + rowbase + (r & 3) + 8 * (r >> 2);
        seglist[mybase] = qi >= 0 ? (((unsigned long long)qi << 32) | (unsigned)idx) : ~0ull;
      }
      ++mybase;
    }
  };
  // groups [ga, gb): anything in the band?  (wave-uniform branch, taken by about one tile in seven)
  auto check = [&](int ga, int gb, int off) __attribute__((always_inline)) {
    unsigned long long need = 0ull;
#pragma unroll
    for (int g = ga; g < gb; ++g) need |= hi_m[g] & ~lo_m[g];
    if (need != 0ull) {
#pragma unroll
      for (int g = ga; g < gb; ++g) {
        const unsigned long long candm = hi_m[g] & ~lo_m[g];
        if (candm != 0ull) list_band(g, candm, off);
      }
    }
  };
  auto next_off = [&](int off) __attribute__((always_inline)) {
    const int n = off + kTileBytes;
    return n == off_end ? off_begin : n;
  };

  half8 A0[KS], A1[KS];
  {
    // one tile: matrix instructions in stages of two groups, the reduction of a stage behind the next stage's instructions
    auto tile = [&](const half8(&A)[KS], int off) __attribute__((always_inline)) {
      constexpr int NP = (QW + 1) / 2;
#pragma unroll
      for (int p = 0; p <= NP; ++p) {
        __builtin_amdgcn_sched_barrier(0);
        if (p < NP) mm(A, 2 * p, (2 * p + 1 < QW) ? 2 * p + 1 : -1);
        if (p > 0) {
#pragma unroll
          for (int g = 2 * (p - 1); g < 2 * p && g < QW; ++g) reduce(g);
        }
        if (NP == 2 && p == 1) pin_step<0, (QW == 4 ? 2 : 1) * KS, 20>();   // stage 0's two reductions between stage 1's matrix instructions
      }
      __builtin_amdgcn_sched_barrier(0);
      check(0, QW, off);
    };
    if (PF == 2) {
      // tiles requested TWO tiles ahead (three register sets): 0.151 -> 0.139 ms on a zero operand, 0.227 -> 0.221 on the real one
      half8 A2[KS];
      int o0 = tstart * kTileBytes, o1 = next_off(o0), o2 = next_off(o1);
      load_tile<KS>(A0, rsrc, voff, o0);
      load_tile<KS>(A1, rsrc, voff, o1);
      for (int it = 0; it < ntl; it += 3) {
        load_tile<KS>(A2, rsrc, voff, o2);
        tile(A0, o0);
        if (it + 1 >= ntl) break;
        o0 = next_off(o2);
        load_tile<KS>(A0, rsrc, voff, o0);
        tile(A1, o1);
        if (it + 2 >= ntl) break;
        o1 = next_off(o0);
        load_tile<KS>(A1, rsrc, voff, o1);
        tile(A2, o2);
        o2 = next_off(o1);
      }
    } else {
      int off = tstart * kTileBytes;
      load_tile<KS>(A0, rsrc, voff, off);
      for (int it = 0; it < ntl; it += 2) {
        const int offn = next_off(off);
        load_tile<KS>(A1, rsrc, voff, offn);   // in flight while this tile is multiplied
        tile(A0, off);
        if (it + 1 >= ntl) break;
        off = next_off(offn);
        load_tile<KS>(A0, rsrc, voff, off);
        tile(A1, offn);
      }
    }
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
    const int mine = __int_as_float(run[g]) <= tlo[g] ? 1 : 0;   // certain hit somewhere in the sweep
    const int res = (mine | __shfl_xor(mine, 32)) ? 0 : kNone;
    if (lane < 32 && qi >= 0 && qi < a.nq && res != kNone) a.best[qi] = res;
    if (COMPACT) {
      // T_hi > 0 <=> the query has thresholds (the stages in front write -1 for every other route)
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
  if (a.rw.pts && cursor != 0u && cursor <= a.seg_cap) {   // wave-uniform: re-check the pairs this wave has just listed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the entries were written by other lanes of this wave
    recheck_segment(a.rw, seglist, cursor, lds_wave, lane);
  }
}

template <int KS, int QW>
static hipError_t launch_sweep_t(const FilterArgs &a, hipStream_t s) {
  const long long waves = (a.ngroups + QW - 1) / QW;
  const unsigned nellblk = (a.rw.pts && a.rw.ell.count) ? kEllWaves / 4 : 0u;
  const dim3 grid((unsigned)((waves + 3) / 4) + nellblk, (unsigned)((a.cq || a.split < 1) ? 1 : a.split));
  const size_t lds = a.rw.pts ? 4 * recheck_w_lds(a.rw.d) : 0;   // <= 36 KiB (d <= 64)
  constexpr int PF = (KS * QW >= 12 && KS <= 4) ? 2 : 1;   // room for a third set of tile registers next to 2 waves per SIMD
  if (a.cq)
    hipLaunchKernelGGL((k_sweep<KS, QW, true, PF>), grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((k_sweep<KS, QW, false, PF>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

// query groups per wave as filter_groups_per_wave(ks, narrow) has them
hipError_t launch_sweep(int ks, int qw, const FilterArgs &a, hipStream_t s) {
  if (a.ngroups <= 0) return hipSuccess;
  switch (ks * 8 + qw) {
    case 1 * 8 + 4: return launch_sweep_t<1, 4>(a, s);
    case 2 * 8 + 4: return launch_sweep_t<2, 4>(a, s);
    case 3 * 8 + 4: return launch_sweep_t<3, 4>(a, s);
    case 4 * 8 + 4: return launch_sweep_t<4, 4>(a, s);
    case 1 * 8 + 2: return launch_sweep_t<1, 2>(a, s);
    case 2 * 8 + 2: return launch_sweep_t<2, 2>(a, s);
    case 3 * 8 + 2: return launch_sweep_t<3, 2>(a, s);
    case 4 * 8 + 2: return launch_sweep_t<4, 2>(a, s);
    case 5 * 8 + 2: return launch_sweep_t<5, 2>(a, s);
    case 6 * 8 + 2: return launch_sweep_t<6, 2>(a, s);
    case 7 * 8 + 2: return launch_sweep_t<7, 2>(a, s);
    case 8 * 8 + 2: return launch_sweep_t<8, 2>(a, s);
    case 9 * 8 + 1: return launch_sweep_t<9, 1>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mlf
