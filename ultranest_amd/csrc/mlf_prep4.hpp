// mlf_prep4.hpp -- bounded per-proposal stage of MLFriends.inside on the FP32 matrix cores (mlf_prep4.hip) and the
// exact side kernels that go with it (ellipsoid band, exact whitening of the few queries that need it)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

// Host-side constants of one region for k_prep4, all in binary32 and rounded in the safe direction.
struct Prep4Consts {
  float g_chain;     // relative error of one binary32 FMA chain incl. operand rounding: (DP + 4) 2^-24 (1 + 2^-10) + 2^-40
  float y0n;         // >= | L^T (c_lay - c_ell) |
  float lf;          // >= | L |_F                (A = L L^T, the ellipsoid matrix)
  float s0n;         // >= | c_lay - c_ell |
  float eps_scale;   // >= 2^-34 | A |_F          (reference rounding + factorisation, as in k_prep3)
  float enl_lo, enl_hi;   // enlarge rounded down / up
  float tf;          // >= | T |_F                (layer matrix, unscaled)
};

struct Prep4Args {
  const double *pts;      // (np, d) row-major proposals, 16-byte aligned
  long long np;
  int d, dp;
  const float *LtF;       // A fragments of L^T: tile t (32 rows), k-steps s >= 16 t; lane (i, h) holds L[2s+h][32t+i]
  const float *y0;        // [32 NE] start values of the ellipsoid chain
  const float *TtF;       // A fragments of T (unscaled): [NT][DP/2][64], rows in filter-column order (see .hip)
  const double *lay_ctr;  // [>= d]
  Prep4Consts c;
  uint8_t *gate;          // out: inside the wrapping ellipsoid (band proposals: provisionally 1)
  int do_tr;              // 0: ellipsoid only (regions without a neighbour scan)
  // band proposals for k_ell_exact
  unsigned *ell_count;
  int *ell_list;
  unsigned ell_cap;
  // filter side (do_tr = 1)
  const double *stats;    // device: [0] sigma, [1] namax, [8 + c] centre of the whitened live points
  double r2;
  void *qF;
  float *tlo, *thi;
  uint8_t *route;
  int *best;
  int *slot;              // out: -1 per proposal (exact-whitening slots are claimed later)
  unsigned *counters;
  unsigned *scan_flag;
  int ks;
  long long nqpad;
};

// instantiated padded dimensionalities (the even DPs of pick_dp up to 64)
#define MLF_FOR_EACH_DP_PREP4(X)                                                              \
  X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
  X(36) X(40) X(44) X(48) X(50) X(52) X(56) X(60) X(64)

bool prep4_usable(int d);                       // d <= 64
size_t prep4_ltf_count(int dp);                 // floats
size_t prep4_ttf_count(int dp);
// host helpers: L = lower Cholesky factor (d x d row-major), T = layer matrix (d x d row-major)
void prep4_lt_fragments(const double *L, int d, int dp, float *out);
void prep4_t_fragments(const double *T, int d, int dp, float *out);
hipError_t launch_prep4(const Prep4Args &a, hipStream_t s);

// exact ellipsoid test of the proposals k_prep4 could not decide (bounded binary64 form first, the reference's
// summation order only inside that form's own band)
struct EllExactArgs {
  unsigned *count;         // reset to 0 by the last workgroup
  unsigned *done;          // workgroups finished (returns to 0)
  unsigned *last;          // optional: receives the count before it is reset
  const int *list;
  unsigned cap;
  const double *pts;
  int d, dp;
  const double *ell_ctr;   // [dp]
  const double *ell_Lt;    // [dp][dp]  Lt[k][j] = L[j][k]
  const double *ell_A;     // [d][dp]
  double eps_scale, enlarge;
  int chol_ok;
  uint8_t *gate;
  uint8_t *route;          // may be null
};
void launch_ell_exact(const EllExactArgs &a, hipStream_t s);

// after the filter sweeps: which queries need their whitened coordinates in the reference arithmetic
struct MarkArgs {
  const unsigned long long *list;
  unsigned seg_cap;
  const unsigned *seg_count;
  long long nsegs;
  long long nq;
  int nlive;
  const uint8_t *route;
  const int *best;
  const unsigned *counters;   // [1] = list overflow
  int *slot;                  // [nq] -1 / claimed (-2) / dense slot (written by k_whiten_slots)
  unsigned unit_cap;          // distinct queries a segment can hold pairs of
  int *uq;                    // [nsegs][unit_cap] claimed queries per segment
  unsigned *ucount;           // [nsegs + 1] claimed queries per segment (scanned in place afterwards)
  int *xq;                    // [nq] queries routed to the exact scan
  unsigned *nx;               // their number (shared counter: rare)
  unsigned *scan_flag;        // set to 1 if any query takes the exact scan
  EllExactArgs ell;           // ell.count != nullptr: the band proposals are decided by the tail of this launch
};
void launch_mark_exact(const MarkArgs &a, hipStream_t s);

// exact whitening (k-ascending binary64 FMA chain on v_mfma_f64_16x16x4_f64, identical to k_prep / k_prep3) of the
// claimed queries into a compact buffer
struct WhitenSlotsArgs {
  const double *pts;
  int d;
  const int *xq;
  const unsigned *nx;
  const int *uq;
  const unsigned *ubase;   // exclusive scan of the per-segment counts, total at [nsegs]
  long long nsegs;
  unsigned unit_cap;
  int *slot;               // out: dense slot of every claimed query
  const double *lay_ctr;
  const double *TtF;       // k_prep3's fragments of T
  double *out;             // [slot][d]
  unsigned *stats_out;     // [0] = number of slots of this batch
};
hipError_t launch_whiten_slots(const WhitenSlotsArgs &a, long long max_slots, hipStream_t s);

}  // namespace mlf
