// mlf_small.hpp -- MLFriends.inside for a handful of proposals in ONE launch (mlf_small.hip)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

constexpr int kSmallMaxPoints = 256;   // proposals per call the single-launch path takes
constexpr int kSmallMaxDim = 128;      // the templated kernels' range (MLF_MAX_DIM = 1024: above 128 the batched pipeline)

struct SmallArgs {
  const double *pts;      // (np, d) row-major; may be pinned host memory mapped into the device (read once per workgroup)
  int np, d, dp;
  // wrapping ellipsoid (H3)
  const double *ell_ctr;  // [dp]
  const double *ell_A;    // [d][dp]
  const double *ell_Lt;   // [dp][dp] rows of L^T (A = L L^T), valid if chol_ok
  double eps_scale, enlarge;
  int chol_ok;
  // layer (T1 / T2)
  int use_scan, layer_kind;   // 0 affine, 1 scaling
  const double *lay_ctr;  // [>= d]
  const double *lay_T8;   // affine: row-major T, row stride ldt8
  int ldt8;
  const double *lay_std;  // scaling: [d]
  const double *wrap;     // [>= d] or null; NaN marks an unwrapped dimension
  // live points, coordinate-major [k][npad]
  const double *refT;
  int n, npad;
  double r2;
  int wpp;                // workgroups per proposal
  unsigned *state;        // [kSmallMaxPoints], zero between launches: workgroups reported (low half), hits (high half); fetched and zeroed by the publisher
  unsigned *finished;     // proposals completed in this launch; returns to zero
  uint8_t *mask;          // out (np); may be pinned host memory
  unsigned *flag;         // host-visible word that receives `seq` when every mask byte has been written
  unsigned seq;
};
void launch_inside_small(const SmallArgs &a, hipStream_t s);

}  // namespace mlf
