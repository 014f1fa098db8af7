// mlf_loglike_dev.hpp -- the benchmark likelihoods as device functions shared by the batch
// kernels (mlf_misc.hip: k_loglike, k_loglike_rows) and the resident walkers' multi-round kernel (mlf_walk.hip), which
// evaluates a walker's proposal inside the wave that owns the walker.
//   Gaussian    docs/gauss.py:25-27          eggbox      examples/testeggbox.py:9-11
//   eggbox (2)  examples/test_PopSliceSampler.py:69-71      Rosenbrock  examples/testrosenbrock.py:10-13
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace mlf {

// one row, one thread, terms in ascending order (odd dimensionalities: k_loglike; above 128: k_loglike_wide)
__device__ __forceinline__ double loglike_row(int kind, const double *x, int d, const double *aux,
                                              double sigma) {
  if (kind == 0) {  // docs/gauss.py:25-27
    double s = 0.0;
    for (int k = 0; k < d; ++k) {
      const double z = (x[k] - aux[k]) / sigma;
      s += z * z;
    }
    return -0.5 * s - 0.5 * log(2.0 * M_PI * sigma * sigma) * (double)d;
  }
  if (kind == 1) {  // examples/testeggbox.py:9-11
    double chi = 1.0;
    for (int k = 0; k < d; ++k) chi *= cos(x[k] / 2.0);
    const double base = 2.0 + chi;
    const double b2 = base * base;
    return b2 * b2 * base;
  }
  if (kind == 2) {  // examples/test_PopSliceSampler.py:69-71
    double chi = 1.0;
    for (int k = 0; k < d; ++k) chi *= cos(x[k]);
    return chi * chi;
  }
  double s = 0.0;  // examples/testrosenbrock.py:10-13
  for (int k = 0; k + 1 < d; ++k) {
    const double av = x[k], bv = x[k + 1];
    const double t = bv - av * av;
    const double w = 1.0 - av;
    s += 100.0 * (t * t) + w * w;
  }
  return -2.0 * s;
}

// One row evaluated by one WAVE (every lane gets the value), BIT-IDENTICAL to what launch_loglike writes for the same row:
// even d <= 128 follows k_loglike_rows (lane l < HW holds coordinates 2 l, 2 l + 1; HW = the smallest power of two with
// 2 HW >= d; terms combined by the same xor tree), otherwise lane 0 runs loglike_row.  tests/test_popstepsampler.py compares
// the multi-round walker kernel, which uses this, with the call-by-call path, which uses launch_loglike.
// the even-d form on values already in the pair layout: lane l < hw holds x0 = x[2 l], x1 = x[2 l + 1]
__device__ inline double loglike_pairs(int kind, double x0, double x1, int d, int hw, const double *aux, double sigma, int lane) {
  const int k0 = 2 * lane;
  const bool active = lane < hw && k0 < d;
  if (!active) x0 = x1 = 0.0;
  double acc;
  if (kind == 0) {
    double c0 = 0.0, c1 = 0.0;
    if (active) {
      c0 = aux[k0];
      c1 = aux[k0 + 1];
    }
    const double z0 = (x0 - c0) / sigma, z1 = (x1 - c1) / sigma;
    acc = active ? z0 * z0 + z1 * z1 : 0.0;
  } else if (kind == 1) {
    acc = active ? cos(x0 / 2.0) * cos(x1 / 2.0) : 1.0;
  } else if (kind == 2) {
    acc = active ? cos(x0) * cos(x1) : 1.0;
  } else {
    const double nx = __shfl_down(x0, 1, 64);
    const double t0 = x1 - x0 * x0, w0 = 1.0 - x0;
    const double t1 = nx - x1 * x1, w1 = 1.0 - x1;
    acc = (lane < hw && k0 + 1 < d) ? 100.0 * (t0 * t0) + w0 * w0 : 0.0;
    if (lane < hw && k0 + 2 < d) acc += 100.0 * (t1 * t1) + w1 * w1;
  }
  for (int o = hw / 2; o > 0; o >>= 1) {
    const double other = __shfl_xor(acc, o, 64);
    acc = (kind == 1 || kind == 2) ? acc * other : acc + other;
  }
  double out;
  if (kind == 0) {
    out = -0.5 * acc + (-0.5 * log(2.0 * M_PI * sigma * sigma) * (double)d);
  } else if (kind == 1) {
    const double b1 = 2.0 + acc, b2 = b1 * b1;
    out = b2 * b2 * b1;
  } else if (kind == 2) {
    out = acc * acc;
  } else {
    out = -2.0 * acc;
  }
  return __shfl(out, 0, 64);
}

__device__ inline double loglike_wave(int kind, const double *x, int d, const double *aux, double sigma, int lane) {
  if ((d & 1) || d > 128) {
    double out = 0.0;
    if (lane == 0) out = loglike_row(kind, x, d, aux, sigma);
    return __shfl(out, 0, 64);
  }
  int hw = 2;
  while (2 * hw < d) hw *= 2;
  const int k0 = 2 * lane;
  const bool active = lane < hw && k0 < d;
  double x0 = 0.0, x1 = 0.0;
  if (active) {
    x0 = x[k0];
    x1 = x[k0 + 1];
  }
  return loglike_pairs(kind, x0, x1, d, hw, aux, sigma, lane);
}

}  // namespace mlf
