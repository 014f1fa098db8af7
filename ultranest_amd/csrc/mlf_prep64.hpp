// mlf_prep64.hpp -- per-proposal stage of MLFriends.inside for 65 ... 128 dimensions on the FP64 matrix cores (mlf_prep64.hip)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

struct Prep64Args {
  const double *pts;      // (np, d) row-major proposals
  long long np;
  int d, dp;              // dp: padded dimensionality of the matrices below (a multiple of 16)
  const double *ell_ctr;  // [>= d]
  const double *ell_L;    // [dp][dp]  L[k][c] row-major: the lower Cholesky factor of the ellipsoid matrix (A = L L^T), zero padded
  const double *ell_A;    // [d][lda]  exact path (proposals inside the bounded form's band)
  int lda;
  double ell_eps_scale;   // 2^-34 |A|_F
  int chol_ok;
  double enlarge;
  uint8_t *gate;          // out: inside the wrapping ellipsoid
  int do_tr;
  const double *lay_ctr;  // [>= d]
  const double *T8;       // row-major layer matrix T[k][c], row stride ldt8, zero padded to dp rows
  int ldt8;
  const double *wrap_shift;   // [>= d], NaN = unwrapped; nullptr = no wraps
  double *t_out;          // whitened coordinates of the proposals inside the ellipsoid, row-major, row stride ldt
  long long ldt;
};

bool prep64_usable(int d);
hipError_t launch_prep64(const Prep64Args &a, hipStream_t s);

}  // namespace mlf
