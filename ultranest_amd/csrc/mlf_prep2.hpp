// mlf_prep2.hpp -- fused ellipsoid + whitening + binary16 quantisation kernel (mlf_prep2.hip)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

struct Prep2Args {
  const double *pts;      // (np, d) row-major proposals
  long long np;
  int d;
  const double *ell_ctr;  // [DP]
  const double *ell_A;    // [d][DP]   exact path (rare)
  const double *ell_Lt;   // [DP][DP]  Lt[k][j] = L[j][k], A = L L^T
  double ell_eps_scale;   // 2^-34 |A|_F
  int chol_ok;            // 0: A is not positive definite, always take the exact path
  double enlarge;
  uint8_t *gate;          // out: inside the wrapping ellipsoid
  int do_tr;
  const double *lay_ctr;  // [DP]
  const double *lay_T8;   // [DP][DP8] row-major T, zero padded (DP8 = DP rounded up to 8)
  const double *wrap_shift;
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
  int ks;
  long long nqpad;
};

#define MLF_FOR_EACH_DP_PREP2(X)                                                              \
  X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
  X(36) X(40) X(44) X(48) X(50) X(52) X(56) X(60) X(64)

bool prep2_usable(int dp);
hipError_t launch_prep2(int dp, const Prep2Args &a, hipStream_t s);

}  // namespace mlf
