// mlf_prep4.hpp -- bounded per-proposal stage of MLFriends.inside on the matrix cores (split binary16 operands) (mlf_prep4.hip) and the
// exact side kernels that go with it (ellipsoid band, exact whitening of the few queries that need it)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

// Host-side constants of one region for k_prep4 (error model: header of mlf_prep4.hip), all in binary32 and rounded
// in the safe direction; the scales are powers of two.
struct Prep4Consts {
  float g_chain;     // g: relative error of one split-binary16 chain (accumulation + operand splitting + dropped product)
  float y0n;         // >= | L^T (c_lay - c_ell) |
  float lf;          // >= | L |_F                (A = L L^T, the ellipsoid matrix)
  float el;          // >= | E_L |_F / s_L        (representation error of the two binary16 pieces of s_L L)
  float l_abs;       // >= | L |_F sqrt(K) 2^-25 / s_x     (binary16 subnormals of the proposal operand)
  float s0n;         // >= | c_lay - c_ell |
  float eps_scale;   // >= 2^-34 | A |_F          (reference rounding + factorisation, as in k_prep3)
  float enl_lo, enl_hi;   // enlarge rounded down / up
  float zt;          // >= g | T |_F + | E_T |_F / s_T     (zeta per unit sigma |delta|)
  float zt_abs;      // >= | T |_F sqrt(K) 2^-25 / s_x     (zeta's absolute term per unit sigma)
  float s_x;         // scale of the centred proposal
  float inv_sx;      // 1 / s_x
  float inv_sl_sx;   // 1 / (s_L s_x): accumulator of the ellipsoid chain -> y
  float inv_st_sx;   // 1 / (s_T s_x): accumulator of the whitening chain -> T^T delta
};

struct Prep4Args {
  const double *pts;      // (np, d) row-major proposals, 16-byte aligned
  long long np;
  int d, dp;
  const void *LtF;        // binary16 A fragments of s_L L^T, hi block then lo block (prep4_lt_fragments)
  const float *y0;        // [32 NE] start values of the ellipsoid chain: s_L s_x L^T (c_lay - c_ell)
  const void *TtF;        // binary16 A fragments of s_T T^T, rows in filter-column order, hi block then lo block
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
  unsigned *counters;
  unsigned *scan_flag;    // set to 1 if a proposal of this batch is routed to the exact scan (reset by the scan launch of the NEXT batch's predecessor: two words alternate)
  int ks;
  long long nqpad;
};

// instantiated padded dimensionalities (the even DPs of pick_dp up to 64)
#define MLF_FOR_EACH_DP_PREP4(X)                                                              \
  X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
  X(36) X(40) X(44) X(48) X(50) X(52) X(56) X(60) X(64)

bool prep4_usable(int d);                       // d <= 64
size_t prep4_ltf_count(int dp);                 // binary16 values
size_t prep4_ttf_count(int dp);
// host helpers: L = lower Cholesky factor (d x d row-major), T = layer matrix (d x d row-major); `scale` = power of
// two applied before the split; return |E|_F, the Frobenius norm of (scale M - hi - lo)
double prep4_lt_fragments(const double *L, int d, int dp, double scale, uint16_t *out_bits);
double prep4_t_fragments(const double *T, int d, int dp, double scale, uint16_t *out_bits);
hipError_t launch_prep4(const Prep4Args &a, hipStream_t s);

void launch_ell_exact(const EllExactArgs &a, hipStream_t s);

// exact re-check of the uncertain pairs with the whitening of their queries inside the launch (mlf_prep4.hip)
struct RecheckWArgs {
  const unsigned long long *list;
  unsigned seg_cap;
  const unsigned *seg_count;
  long long nsegs;
  unsigned unit_cap;       // distinct queries a segment can hold pairs of
  const double *refR;      // [npad][dp] whitened live points
  int n, d, dp;
  const double *pts;       // (nq, d) proposals as handed over
  long long nq;
  const double *lay_ctr;   // [>= d]
  const double *T64;       // layer matrix as 64 x 64 row-major, zero padded: element (k, c) at T64[k * 64 + c]
  double r2;
  int *best;
  EllExactArgs ell;        // ell.count != nullptr: the band proposals are decided by the last waves of this launch
  unsigned ell_waves;      // waves of k_recheck_whiten that take the band proposals (0 = kEllWaves)
};
void launch_recheck_whiten(const RecheckWArgs &a, hipStream_t s);

}  // namespace mlf
