// mlf_prep3.hpp -- per-proposal stage of MLFriends.inside on the FP64 matrix cores (mlf_prep3.hip)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

struct Prep3Args {
  const double *pts;      // (np, d) row-major proposals
  long long np;
  int d;
  int dp;                 // padded dimensionality of the live-point layouts (norm columns of the filter operand start here)
  int nk;                 // k-steps of 4 coordinates (filled in by launch_prep3: prep3_ksteps(d))
  const double *ell_ctr;  // [>= d]
  const double *ell_A;    // [d][lda]  exact path (rare)
  int lda;
  const double *LtF;      // A-fragments Lt[kb = 16 ct + (lane & 15)][j = 4 ks + (lane >> 4)], tiles with ks >= 4 ct only
  double ell_eps_scale;   // 2^-34 |A|_F
  int chol_ok;
  double enlarge;
  uint8_t *gate;          // out: inside the wrapping ellipsoid
  int do_tr;
  const double *lay_ctr;  // [>= d]
  const double *TtF;      // [NC][nk][64] A-fragments: T[k = 4 ks + (lane >> 4)][c = 16 ct + (lane & 15)]
  const double *wrap_shift;   // [>= d], NaN = unwrapped; nullptr = no wraps
  double *t_out;          // whitened coordinates, element (p, c) at p*t_ldq + c*t_ldk
  long long t_ldq, t_ldk;
  // quantisation for the MFMA filter (qF == nullptr: skip)
  void *qF;
  float *tlo, *thi;
  uint8_t *route;
  int *best;
  unsigned *counters;
  const double *stats;
  double r2;
  int ks;                 // filter k-steps of 16 binary16 columns
  long long nqpad;
  int nlt, xstride;       // filled in by launch_prep3
};

bool prep3_usable(int d);
int prep3_ksteps(int d);
// host helper: d x d row-major M -> fragment order; transpose = false: element (row, k) = M[row][k]
// (rows = output index), transpose = true: element (row, k) = M[k][row]
// upper_only: keep only the tiles with ks >= 4 ct (a factor with M-element (row, k) = 0 for k < row)
void prep3_fragments(const double *M, int d, bool transpose, bool upper_only, double *out);
size_t prep3_fragment_count(int d);
hipError_t launch_prep3(const Prep3Args &a, hipStream_t s);

}  // namespace mlf
