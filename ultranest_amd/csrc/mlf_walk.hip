// mlf_walk.hip -- population step-sampler state machine, resident in HBM.
//
// One lane per walker.  A walker's state is a few scalars plus two d-vectors (position, slice
// direction), so every kernel here is a short HBM-bound pass over P walkers; what the device
// buys is that (u, v, t, brackets, chain history) never cross PCIe between likelihood batches.
// Arithmetic that the reference performs in a fixed order (u + v*t, bracket doubling, the
// bisection draw low + (high-low)*U) is done in the same order without FMA (the file is compiled
// with -ffp-contract=off), so the state machine is bit-identical to ultranest/stepfuncs.pyx for
// the same random numbers; tests/test_stepfuncs_golden.py checks that against recorded vectors.
#include "mlf_walk.hpp"

#include <math.h>

#include "mlf_loglike_dev.hpp"
#include "mlf_philox_dev.hpp"

namespace mlf {

namespace {

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000ll); }

__device__ __forceinline__ bool inside_open_unit(double x) { return 0.0 < x && x < 1.0; }

// evolve_update for one walker (stepfuncs.pyx:160-183); returns the final success flag
__device__ __forceinline__ bool update_walker(bool hit, double &t, double &left, double &right, uint8_t &sl,
                                              uint8_t &sr) {
  const bool l = sl != 0, r = sr != 0;
  const bool search_right = !l && r, bisecting = !(l || r);
  bool success = hit;
  if (success) {
    if (l)
      left *= 2;
    else if (search_right)
      right *= 2;
  } else {
    if (l)
      sl = 0;
    else if (search_right)
      sr = 0;
  }
  if (bisecting) {
    if (t < 0)
      left = t;
    else
      right = t;
    if (success) t = qnan();
  } else {
    success = false;
  }
  return success;
}

// second half of step_back: drop chain slots from the end until no below-threshold entry is left
__device__ __forceinline__ void step_back_unwind(double Lmin, double *L, long long width, int nbelow, long long &gen,
                                                 double &t) {
  while (nbelow > 0) {
    const long long g = gen;
    if (g < 0 || g >= width) break;
    gen = g - 1;
    t = qnan();
    if (L[g] < Lmin) --nbelow;
    L[g] = qnan();
  }
}

// step_back for one walker (stepfuncs.pyx:285-334).  `width` = max generation + 1 over the
// population.  Exact for generation >= 0; a walker that unwinds below generation 0 with
// below-threshold entries left (a state the sampler never produces) stops at -1.
__device__ __forceinline__ void step_back_walker(double Lmin, double *L, int G, long long width, long long &gen,
                                                 double &t) {
  if (width > G) width = G;
  int nbelow = 0;
  for (long long k = 0; k < width; ++k) nbelow += (L[k] < Lmin) ? 1 : 0;
  step_back_unwind(Lmin, L, width, nbelow, gen, t);
}

// whitened coordinates of one point (T1: fmod wrap, centre, k-ascending FMA chain like BLAS)
__device__ __forceinline__ void whiten_point(const WalkLayer &ly, const double *x, int d, int c, double &out) {
  if (ly.kind == 1) {
    double v = x[c];
    if (ly.wrap && !isnan(ly.wrap[c])) v = fmod(v + ly.wrap[c], 1.0);
    out = (v - ly.ctr[c]) / ly.mat[c];
    return;
  }
  double acc = 0.0;
  for (int k = 0; k < d; ++k) {
    double v = x[k];
    if (ly.wrap && !isnan(ly.wrap[k])) v = fmod(v + ly.wrap[k], 1.0);
    acc = __builtin_fma(v - ly.ctr[k], ly.mat[(size_t)k * d + c], acc);
  }
  out = acc;
}

}  // namespace

// ------------------------------------------------------------------ resident state ---------------
__global__ void k_walk_reset(WalkState w) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nu = (long long)w.P * w.G * w.d;
  if (e < nu) w.allu[e] = qnan();
  if (e < (long long)w.P * w.G) w.allL[e] = qnan();
  if (e < (long long)w.P * w.d) w.currentv[e] = qnan();
  if (e < w.P) {
    w.generation[e] = -1;
    w.currentt[e] = qnan();
    w.left[e] = 0.0;
    w.right[e] = 0.0;
    w.sl[e] = 0;
    w.sr[e] = 0;
  }
}

__global__ __launch_bounds__(256) void k_max_generation(const long long *generation, int n, long long *gmax) {
  __shared__ long long part[256];
  long long m = -(1ll << 62);
  for (int i = threadIdx.x; i < n; i += 256) m = generation[i] > m ? generation[i] : m;
  part[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off && part[threadIdx.x + off] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *gmax = part[0];
}

__global__ void k_step_back(double Lmin, double *allL, int n, int G, long long *generation, double *currentt,
                            const long long *gmax, const uint8_t *sl, const uint8_t *sr, uint8_t *flags,
                            const StepParams *sp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (sp) Lmin = sp->Lmin;
  long long gen = generation[i];
  double t = currentt[i];
  step_back_walker(Lmin, allL + (size_t)i * G, G, *gmax + 1, gen, t);
  generation[i] = gen;
  currentt[i] = t;
  if (flags) flags[i] = (uint8_t)((isfinite(t) ? 0 : 1) | (sl[i] ? 2 : 0) | (sr[i] ? 4 : 0));
}

__global__ void k_walk_start(WalkState w, const long long *idx, int n, const double *rows, const double *L) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w.d) return;
  const int j = e / w.d, k = e % w.d;
  const long long i = idx[j];
  w.allu[((size_t)i * w.G) * w.d + k] = rows[(size_t)j * w.d + k];
  if (k == 0) {
    w.allL[(size_t)i * w.G] = L[j];
    w.generation[i] = 0;
  }
}

__global__ void k_walk_points(WalkState w, const long long *idx, int n, double *out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w.d) return;
  const int j = e / w.d, k = e % w.d;
  const long long i = idx[j];
  long long g = w.generation[i];
  if (g < 0) g += w.G;
  out[(size_t)j * w.d + k] = w.allu[((size_t)i * w.G + g) * w.d + k];
}

// setup_brackets with host-provided directions (popstepsampler.py:483-505)
__global__ void k_walk_brackets(WalkState w, const long long *idx, int n, double scale, const double *v_rows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w.d) return;
  const int j = e / w.d, k = e % w.d;
  const long long i = idx[j];
  w.currentv[(size_t)i * w.d + k] = v_rows[(size_t)j * w.d + k];
  if (k == 0) {
    w.left[i] = -scale;
    w.right[i] = scale;
    w.sl[i] = 1;
    w.sr[i] = 1;
    w.currentt[i] = 0.0;
  }
}

// ---- wave-per-walker kernels: lane = coordinate (d <= 128: two coordinates per lane) ----------------
// One thread per walker read its rows with a 8 d-byte stride between lanes (every load a different cache
// line) and walked them serially: 0.93 ms per step at 10^5 walkers x 50, ~25 us of pure latency per kernel at
// 100 walkers.  Here a wave owns one walker: rows are read coalesced, per-walker scalars are computed by all
// lanes alike (same inputs, same result) and written by lane 0.

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Direction of a new slice for walker i (stepfuncs.pyx:348-535), drawn on the device: every lane gets ITS coordinates
// vr[h] = v[lane + 64 h] (0 beyond d).  Philox stream 2, (npairs + 2) blocks per walker: block 0 = integer picks + mixture
// coin, blocks 1.. = Box-Muller pairs (coordinate k takes the cosine / sine branch of pair k / 2).
__device__ void dw_direction(const WalkState &w, int i, int lane, int kind, double dirscale, const WalkDirData &dd,
                             unsigned long long seed, unsigned long long offset, double (&vr)[2]) {
  const int d = w.d;
  const int npairs = (d + 1) / 2;
  const unsigned long long base = offset + (unsigned long long)i * (unsigned long long)(npairs + 2);
  unsigned pick[4];
  philox_block(seed, 2u, base, pick);
  int k = kind;
  if (k == DIR_MIXTURE) k = (u01(pick[2], pick[3]) < 0.5) ? DIR_DIFFERENTIAL : DIR_REGION_ORIENTED;
  vr[0] = vr[1] = 0.0;
  if (k == DIR_CUBE_ORIENTED || k == DIR_CUBE_ORIENTED_SCALED) {
    const int j = (int)below(pick[0], (unsigned)d);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h;
      if (c < d) vr[h] = c == j ? ((k == DIR_CUBE_ORIENTED) ? dirscale : dirscale * dd.std[j]) : 0.0;
    }
  } else if (k == DIR_REGION_ORIENTED) {
    const int j = (int)below(pick[0], (unsigned)d);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h;
      if (c < d) vr[h] = dd.axes[(size_t)j * d + c] * dirscale;
    }
  } else if (k == DIR_DIFFERENTIAL) {
    const unsigned a = below(pick[0], (unsigned)dd.nlive);
    unsigned b = below(pick[1], (unsigned)(dd.nlive - 1));
    if (b >= a) ++b;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h;
      if (c < d) vr[h] = (dd.live[(size_t)a * d + c] - dd.live[(size_t)b * d + c]) * dirscale;
    }
  } else {   // DIR_RANDOM, DIR_REGION_RANDOM: isotropic unit vector of length dirscale
    double g[2] = {0.0, 0.0};
    double part = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h;
      if (c < d) {
        unsigned r4[4];
        philox_block(seed, 2u, base + 1 + (c >> 1), r4);
        const double rad = sqrt(-2.0 * log(u01(r4[0], r4[1])));
        const double ang = 2.0 * M_PI * u01(r4[2], r4[3]);
        g[h] = (c & 1) ? rad * sin(ang) : rad * cos(ang);
        part += g[h] * g[h];
      }
    }
    const double f = dirscale / sqrt(wave_sum(part));
    g[0] *= f;
    g[1] *= f;
    if (k == DIR_RANDOM) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < d) vr[h] = g[h];
    } else {   // v[r] = sum_c axes[r][c] * v1[c]   (einsum 'ij,kj->ki', stepfuncs.pyx:476)
      double acc[2] = {0.0, 0.0};
      for (int c = 0; c < d; ++c) {
        const double v1c = __shfl(c < 64 ? g[0] : g[1], c & 63, 64);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = lane + 64 * h;
          if (r < d) acc[h] += dd.axes[(size_t)r * d + c] * v1c;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < d) vr[h] = acc[h];
    }
  }
}

// setup_brackets with a device-side direction draw for a walker whose bracket is undefined (popstepsampler.py:483-505).
// `t`, `left`, `right`, `sl`, `sr` are the wave's copies of the walker state (updated here); vr (optional) receives the
// lane's coordinates of the new direction.
__device__ void dw_brackets_philox(const WalkState &w, int i, int lane, double scale, int kind, double dirscale,
                                   const WalkDirData &dd, unsigned long long seed, unsigned long long offset, double &t,
                                   double &left, double &right, bool &sl, bool &sr, double *vr_out = nullptr) {
  if (isfinite(t)) return;   // wave-uniform
  const int d = w.d;
  double *v = w.currentv + (size_t)i * d;
  double vr[2];
  dw_direction(w, i, lane, kind, dirscale, dd, seed, offset, vr);
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (lane + 64 * h < d) v[lane + 64 * h] = vr[h];
  if (vr_out) {
    vr_out[0] = vr[0];
    vr_out[1] = vr[1];
  }
  left = -scale;
  right = scale;
  sl = true;
  sr = true;
  t = 0.0;
  if (lane == 0) {
    w.left[i] = left;
    w.right[i] = right;
    w.sl[i] = 1;
    w.sr[i] = 1;
    w.currentt[i] = 0.0;
  }
}

__global__ __launch_bounds__(64) void k_walk_brackets_philox(WalkState w, double scale, int kind, double dirscale,
                                                             WalkDirData dd, unsigned long long seed,
                                                             unsigned long long offset, const StepParams *sp) {
  const int lane = threadIdx.x;
  if (sp) {
    scale = sp->scale;
    dirscale = sp->dirscale;
    seed = sp->seed;
    offset = sp->offset;
  }
  for (int i = blockIdx.x; i < w.P; i += gridDim.x) {   // several walkers per wave: 10^5 one-wave workgroups are dispatch bound
    double t = w.currentt[i], left = w.left[i], right = w.right[i];
    bool sl = w.sl[i] != 0, sr = w.sr[i] != 0;
    dw_brackets_philox(w, i, lane, scale, kind, dirscale, dd, seed, offset, t, left, right, sl, sr);
  }
}

// evolve, first half (stepfuncs.pyx:249-261): slice coordinate, proposed point, cube test; optionally the
// prior transform of the proposal (tkind < 0: none).  gen, t, left, right, sl, sr = the wave's copies.
__device__ void dw_propose(const WalkState &w, int i, int lane, const double *unif, unsigned long long seed,
                           unsigned long long offset, long long gen, double t, double left, double right, bool sl, bool sr,
                           int tkind, double ta, double tb) {
  const bool movable = gen >= 0 && gen < w.G - 1;
  if (lane == 0) w.movable[i] = movable ? 1 : 0;
  if (!movable) {
    if (lane == 0) w.acceptable[i] = 0;
    return;
  }
  if (sl) {
    t = left;
  } else if (sr) {
    t = right;
  } else {
    double u;
    if (unif) {
      u = unif[i];
    } else {
      unsigned r4[4];
      philox_block(seed, 3u, offset + (unsigned long long)i, r4);
      u = u01(r4[0], r4[1]);
    }
    const double range = right - left;
    const double scaled = range * u;
    t = left + scaled;
    if (lane == 0) w.currentt[i] = t;
  }
  const double *u0 = w.allu + ((size_t)i * w.G + gen) * w.d;
  const double *v = w.currentv + (size_t)i * w.d;
  double *un = w.unew + (size_t)i * w.d;
  bool ok = true;
  for (int k = lane; k < w.d; k += 64) {
    const double step = v[k] * t;
    const double x = u0[k] + step;
    un[k] = x;
    ok = ok && inside_open_unit(x);
    if (tkind >= 0) {
      double p = x;
      if (tkind == 1) {
        const double m = x * ta;
        p = m + tb;
      } else if (tkind == 2) {
        const double m = x * ta;
        p = m * tb;
      }
      w.pnew[(size_t)i * w.nparams + k] = p;
    }
  }
  const bool all_ok = __all(ok);
  if (lane == 0) w.acceptable[i] = all_ok ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_walk_propose(WalkState w, const double *unif, unsigned long long seed,
                                                     unsigned long long offset, const StepParams *sp) {
  const int lane = threadIdx.x;
  if (sp) {
    seed = sp->seed;
    offset = sp->offset;
  }
  for (int i = blockIdx.x; i < w.P; i += gridDim.x)
    dw_propose(w, i, lane, unif, seed, offset, w.generation[i], w.currentt[i], w.left[i], w.right[i], w.sl[i] != 0,
               w.sr[i] != 0, -1, 0.0, 0.0);
}

__global__ void k_walk_transform(WalkState w, int tkind, double a, double b) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)w.P * w.d) return;
  const double x = w.unew[e];
  double p = x;
  if (tkind == 1) {
    const double m = x * a;
    p = m + b;
  } else if (tkind == 2) {
    const double m = x * a;
    p = m * b;
  }
  w.pnew[e] = p;
}

// blk = exclusive per-256 offsets of the acceptable walkers (launch_compact); walker i takes row
// rank(i) of the compacted host results
__global__ __launch_bounds__(256) void k_walk_expand(WalkState w, const unsigned *blk, const double *pc,
                                                     const double *Lc) {
  __shared__ unsigned wsum[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool acc = i < w.P && w.acceptable[i] != 0;
  const unsigned long long b = __ballot(acc);
  if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned base = blk[blockIdx.x];
  for (int k = 0; k < wave; ++k) base += wsum[k];
  const unsigned rank = base + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
  if (!acc) return;
  w.Lnew[i] = Lc[rank];
  for (int k = 0; k < w.nparams; ++k) w.pnew[(size_t)i * w.nparams + k] = pc[(size_t)rank * w.nparams + k];
}

// diagnose_move_distances for one walker that moved (wave-wide, lane = whitened coordinate; T rows are read
// coalesced, squared differences summed by a fixed shuffle tree): uo = the point the step started from, un = the
// accepted point
// d <= 64, affine layer: lane k holds coordinate k of the two points (vo: where the step started, vn: the accepted point; 0
// beyond d); returns the squared whitened distance on lane 0.  Both points share every matrix element; the chains read the
// centred coordinates by lane broadcast (same values and order as whiten_point: results are identical)
__device__ __forceinline__ double move_distance_regs(const WalkLayer &ly, int d, int lane, double vo, double vn) {
  if (lane < d) {
    if (ly.wrap && !isnan(ly.wrap[lane])) {
      vo = fmod(vo + ly.wrap[lane], 1.0);
      vn = fmod(vn + ly.wrap[lane], 1.0);
    }
    vo -= ly.ctr[lane];
    vn -= ly.ctr[lane];
  } else {
    vo = vn = 0.0;
  }
  double ta = 0.0, tb = 0.0;
  const int c = lane < d ? lane : 0;
  for (int k = 0; k < d; ++k) {
    const double m = ly.mat[(size_t)k * d + c];
    ta = __builtin_fma(__shfl(vo, k, 64), m, ta);
    tb = __builtin_fma(__shfl(vn, k, 64), m, tb);
  }
  double acc = 0.0;
  if (lane < d) {
    const double diff = ta - tb;
    acc = diff * diff;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  return acc;
}

__device__ __forceinline__ void dw_move_distance(const WalkState &w, const WalkLayer &ly, int i, int lane, const double *uo,
                                                 const double *un) {
  double acc = 0.0;
  if (ly.kind == 0 && w.d <= 64) {
    double vo = 0.0, vn = 0.0;
    if (lane < w.d) {
      vo = uo[lane];
      vn = un[lane];
    }
    acc = move_distance_regs(ly, w.d, lane, vo, vn);
  } else {
    for (int c = lane; c < w.d; c += 64) {
      double ta, tb;
      whiten_point(ly, uo, w.d, c, ta);
      whiten_point(ly, un, w.d, c, tb);
      const double diff = ta - tb;
      acc += diff * diff;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  }
  if (lane == 0) w.dist2[i] = acc;
}

// evolve, second half (evolve_update) + PopulationSliceSampler.advance bookkeeping
// (popstepsampler.py:585-603) + move diagnostics (diagnose_move_distances :64-94) for ONE walker, by the wave that owns it
__device__ void dw_update(const WalkState &w, int i, int lane, double Lmin, const WalkLayer &ly) {
  // every lane reads the walker's scalars and derives the same decision; lane 0 writes them back
  const bool movable = w.movable[i] != 0;
  const bool hit = w.acceptable[i] != 0 && w.Lnew[i] > Lmin;
  double t = w.currentt[i], left = w.left[i], right = w.right[i];
  uint8_t sl = w.sl[i], sr = w.sr[i];
  const long long g0 = w.generation[i];
  const double Lnew = w.Lnew[i];
  bool success = false;
  if (movable) success = update_walker(hit, t, left, right, sl, sr);
  if (lane == 0) {
    w.dist2[i] = qnan();
    w.success[i] = success ? 1 : 0;
    if (movable) {
      w.currentt[i] = t;
      w.left[i] = left;
      w.right[i] = right;
      w.sl[i] = sl;
      w.sr[i] = sr;
    }
    if (success) {
      w.generation[i] = g0 + 1;
      w.allL[(size_t)i * w.G + g0 + 1] = Lnew;
    }
  }
  if (!success) return;
  const double *un = w.unew + (size_t)i * w.d;
  double *dst = w.allu + ((size_t)i * w.G + g0 + 1) * w.d;
  for (int k = lane; k < w.d; k += 64) dst[k] = un[k];
  for (int k = lane; k < w.nparams; k += 64) w.currentp[(size_t)i * w.nparams + k] = w.pnew[(size_t)i * w.nparams + k];
  // move diagnostics of this step while the wave still owns the walker (a second launch re-read all of this)
  if (ly.kind >= 0) dw_move_distance(w, ly, i, lane, w.allu + ((size_t)i * w.G + g0) * w.d, un);
}

__global__ __launch_bounds__(64) void k_walk_update(WalkState w, double Lmin, const StepParams *sp, WalkLayer ly) {
  const int lane = threadIdx.x;
  if (sp) Lmin = sp->Lmin;
  for (int i = blockIdx.x; i < w.P; i += gridDim.x) dw_update(w, i, lane, Lmin, ly);
}

// step statistics, stage 1: every workgroup reduces 1024 walkers to one row of partial sums
// (likelihood evaluations, walkers moved on, successes, far-enough moves, sum of log relative distances)
constexpr int kStatsChunk = 1024;
constexpr int kStatsCols = 6;   // + number of walkers (re)started in this call (setup_start's ring shift needs it)
__global__ __launch_bounds__(256) void k_walk_stats(WalkState w, double r2, const StepParams *sp, double *partials,
                                                    const uint8_t *was_starting) {
  __shared__ double part[256][kStatsCols];
  if (sp) r2 = sp->r2;
  const double ref = sqrt(r2);
  double nc = 0, nmov = 0, nsucc = 0, nfar = 0, slog = 0, nstart = 0;
  const int i0 = blockIdx.x * kStatsChunk;
  for (int j = threadIdx.x; j < kStatsChunk; j += 256) {
    const int i = i0 + j;
    if (i >= w.P) continue;
    if (was_starting && was_starting[i]) nstart += 1;
    if (!w.movable[i]) continue;
    nmov += 1;
    nc += w.acceptable[i] ? 1 : 0;
    if (w.success[i]) {
      nsucc += 1;
      const double d2 = w.dist2[i];
      if (!isnan(d2)) {
        nfar += (d2 > r2) ? 1 : 0;
        slog += log(sqrt(d2) / ref + 1e-10);
      }
    }
  }
  part[threadIdx.x][0] = nc;
  part[threadIdx.x][1] = nmov;
  part[threadIdx.x][2] = nsucc;
  part[threadIdx.x][3] = nfar;
  part[threadIdx.x][4] = slog;
  part[threadIdx.x][5] = nstart;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int c = 0; c < kStatsCols; ++c) part[threadIdx.x][c] += part[threadIdx.x + off][c];
    __syncthreads();
  }
  if (threadIdx.x < kStatsCols) partials[blockIdx.x * kStatsCols + threadIdx.x] = part[0][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_walk_harvest(WalkState w, long long ring_host, long long *ring_dev, double r2,
                                                      double *rec, const StepParams *sp, const uint8_t *was_starting,
                                                      const double *partials) {
  __shared__ double part[256][kStatsCols];
  __shared__ long long s_ring;
  if (sp) r2 = sp->r2;
  double nc = 0, nmov = 0, nsucc = 0, nfar = 0, slog = 0, nstart = 0;
  const int nrows = (w.P + kStatsChunk - 1) / kStatsChunk;   // stage 2 of the statistics: rows of k_walk_stats
  for (int b = threadIdx.x; b < nrows; b += 256) {
    nc += partials[b * kStatsCols + 0];
    nmov += partials[b * kStatsCols + 1];
    nsucc += partials[b * kStatsCols + 2];
    nfar += partials[b * kStatsCols + 3];
    slog += partials[b * kStatsCols + 4];
    nstart += partials[b * kStatsCols + 5];
  }
  part[threadIdx.x][0] = nc;
  part[threadIdx.x][1] = nmov;
  part[threadIdx.x][2] = nsucc;
  part[threadIdx.x][3] = nfar;
  part[threadIdx.x][4] = slog;
  part[threadIdx.x][5] = nstart;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int c = 0; c < kStatsCols; ++c) part[threadIdx.x][c] += part[threadIdx.x + off][c];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long r = ring_dev ? *ring_dev : ring_host;
    if (was_starting && ring_dev) {   // setup_start's ring shift (:456-462), deferred from the prologue kernel
      const double ns = part[0][5];
      if (ns > 0 && ns < (double)w.P)
        for (int guard = 0; guard < w.P && was_starting[r]; ++guard) r = (r + 1) % w.P;
      *ring_dev = r;
    }
    s_ring = r;
  }
  __syncthreads();
  const long long ring = s_ring;
  const bool found = w.generation[ring] == (long long)(w.G - 1);
  const size_t row = ((size_t)ring * w.G + (w.G - 1)) * w.d;
  if (threadIdx.x == 0) {
    rec[0] = found ? 1.0 : 0.0;
    rec[1] = found ? w.allL[(size_t)ring * w.G + (w.G - 1)] : qnan();
    rec[2] = w.left[ring];
    rec[3] = w.right[ring];
    for (int c = 0; c < 5; ++c) rec[4 + c] = part[0][c];
  }
  if (found) {
    for (int k = threadIdx.x; k < w.d; k += 256) rec[9 + k] = w.allu[row + k];
    for (int k = threadIdx.x; k < w.nparams; k += 256) rec[9 + w.d + k] = w.currentp[(size_t)ring * w.nparams + k];
  }
  __syncthreads();
  if (found) {   // popstepsampler.py:678-681
    for (int e = threadIdx.x; e < w.G * w.d; e += 256) w.allu[(size_t)ring * w.G * w.d + e] = qnan();
    for (int e = threadIdx.x; e < w.G; e += 256) w.allL[(size_t)ring * w.G + e] = qnan();
    if (threadIdx.x == 0) {
      w.generation[ring] = -1;
      w.currentt[ring] = qnan();
    }
  }
  if (threadIdx.x == 0 && ring_dev) {   // device-side ring index: advance past the harvested walker (shift(), :605-609)
    const long long next = found ? (ring + 1) % w.P : ring;
    *ring_dev = next;
    rec[9 + w.d + w.nparams] = (double)next;
  }
}

// setup_start on the device (popstepsampler.py:443-470): the ring index skips walkers that are being
// restarted (unless all of them are), then every walker with generation < 0 starts from a random
// live point above the threshold.  Philox stream 4, up to 64 blocks per walker (rejection of the
// few live points at or below Lmin; a linear search is the fallback).
__global__ __launch_bounds__(256) void k_walk_ring_shift(WalkState w, long long *ring) {
  __shared__ int nstart[256];
  int n = 0;
  for (int i = threadIdx.x; i < w.P; i += 256) n += w.generation[i] < 0 ? 1 : 0;
  nstart[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) nstart[threadIdx.x] += nstart[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && nstart[0] > 0 && nstart[0] < w.P) {
    long long r = *ring;
    for (int guard = 0; guard < w.P && w.generation[r] < 0; ++guard) r = (r + 1) % w.P;
    *ring = r;
  }
}

__device__ void dw_restart_philox(const WalkState &w, int i, int lane, const double *live, const double *Ls, int nlive,
                                  double Lmin, unsigned long long seed, unsigned long long offset, long long &gen) {
  if (gen >= 0) return;   // wave-uniform
  int pick = -1;          // every lane runs the same draws
  for (int attempt = 0; attempt < 64 && pick < 0; ++attempt) {
    unsigned r4[4];
    philox_block(seed, 4u, offset + (unsigned long long)i * 64ull + attempt, r4);
#pragma unroll
    for (int j = 0; j < 4 && pick < 0; ++j) {
      const int cand = (int)below(r4[j], (unsigned)nlive);
      if (Ls[cand] > Lmin) pick = cand;
    }
  }
  for (int j = 0; j < nlive && pick < 0; ++j)
    if (Ls[j] > Lmin) pick = j;
  if (pick < 0) return;   // no live point above the threshold: the walker stays unstarted
  for (int k = lane; k < w.d; k += 64) w.allu[((size_t)i * w.G) * w.d + k] = live[(size_t)pick * w.d + k];
  gen = 0;
  if (lane == 0) {
    w.allL[(size_t)i * w.G] = Ls[pick];
    w.generation[i] = 0;
  }
}

__global__ __launch_bounds__(64) void k_walk_restart_philox(WalkState w, const double *live, const double *Ls, int nlive,
                                                            double Lmin, unsigned long long seed,
                                                            unsigned long long offset, const StepParams *sp) {
  if (sp) {
    Lmin = sp->Lmin;
    seed = sp->seed;
    offset = sp->offset;
  }
  for (int i = blockIdx.x; i < w.P; i += gridDim.x) {
    long long gen = w.generation[i];
    dw_restart_philox(w, i, threadIdx.x, live, Ls, nlive, Lmin, seed, offset, gen);
  }
}

// The per-walker front half of a whole step in ONE kernel (each dependent launch costs ~5 us of dispatch
// latency, and a 100-walker population has nothing else to hide it): step_back, restart, new slice,
// proposal, prior transform.  was_starting feeds the ring-index shift in the harvest kernel.  step_back
// looks at all G chain slots: slots past a walker's generation hold NaN, so this equals the reference's
// window of max(generation) + 1 slots.
__device__ void dw_prologue(const WalkState &w, int i, int lane, const double *live, const double *Ls, int nlive, int dirkind,
                            const WalkDirData &dd, int tkind, double ta, double tb, uint8_t *was_starting, const StepParams &p) {
  long long gen = w.generation[i];
  double t = w.currentt[i];
  // all per-walker scalars are requested up front: one memory round trip instead of one per stage
  double left = w.left[i], right = w.right[i];
  bool sl = w.sl[i] != 0, sr = w.sr[i] != 0;
  // step_back: the likelihood history is read by the whole wave at once (one thread walking the G slots paid
  // G dependent load latencies per walker: 0.15 ms of this kernel at 10^5 walkers); the rare unwinding stays scalar
  double *Lrow = w.allL + (size_t)i * w.G;
  int nbelow = 0;
  for (int k0 = 0; k0 < w.G; k0 += 64) {
    const int k = k0 + lane;
    const bool below_thr = k < w.G && Lrow[k] < p.Lmin;
    nbelow += (int)__popcll(__ballot(below_thr));
  }
  if (lane == 0) {
    if (nbelow > 0) step_back_unwind(p.Lmin, Lrow, w.G, nbelow, gen, t);
    w.generation[i] = gen;
    w.currentt[i] = t;
    was_starting[i] = gen < 0 ? 1 : 0;
  }
  gen = __shfl(gen, 0, 64);
  t = __shfl(t, 0, 64);
  dw_restart_philox(w, i, lane, live, Ls, nlive, p.Lmin, p.seed, p.offset, gen);
  dw_brackets_philox(w, i, lane, p.scale, dirkind, p.dirscale, dd, p.seed, p.offset, t, left, right, sl, sr);
  // the rows written above (restart point, direction) are read back by other lanes of this wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  dw_propose(w, i, lane, nullptr, p.seed, p.offset, gen, t, left, right, sl, sr, tkind, ta, tb);
}

__global__ __launch_bounds__(64) void k_walk_prologue(WalkState w, const double *live, const double *Ls, int nlive,
                                                      int dirkind, WalkDirData dd, int tkind, double ta, double tb,
                                                      uint8_t *was_starting, StepParams p, const StepParams *sp) {
  const int lane = threadIdx.x;
  if (sp) p = *sp;
  for (int i = blockIdx.x; i < w.P; i += gridDim.x)
    dw_prologue(w, i, lane, live, Ls, nlive, dirkind, dd, tkind, ta, tb, was_starting, p);
}

// ------------------------------------------------------------------ several rounds per launch ----
// PopulationSliceSampler.__next__ advances every walker by ONE likelihood evaluation and then looks at the walker the ring
// index points to (popstepsampler.py:610-697); the driver calls it again and again until that walker has finished its
// nsteps (integrator.py:1839-1950) -- with the threshold, the live points and the scale unchanged in between.  On a GPU each
// such call is a host round trip (72 us at popsize 1024: 96 % of a d = 10 eggbox run).  Between two harvests the walkers
// do not interact, so the rounds of one walker can run back to back inside the wave that owns it:
//   k_walk_round0   every walker: round 0 (step_back, restart, new slice, proposal, prior transform, likelihood, update)
//   k_walk_ring     one workgroup: the ring index skips restarting walkers exactly as k_walk_harvest does; the ring walker
//                   then runs rounds 1, 2, ... ALONE until it has finished (or max_rounds): that fixes R, the number of
//                   rounds the reference's loop would have made; harvest of the record, ring index advanced
//   k_walk_rest     every other walker: rounds 1 ... R - 1
//   k_walk_round_stats  per round: the step statistics of that round (one row of the sampler's logstat each)
// Round r draws from the Philox counters of call r of the call-by-call path (offset + r * per_call), and every stage is
// the same device function, so the state after the launch sequence is bit for bit the state after R calls of
// mlf_walkers_step_dev (tests/test_popstepsampler.py::test_rounds_equal_single_steps).
__device__ void dw_round(const RoundsArgs &a, int i, int lane, int r, const StepParams &p0) {
  StepParams p = p0;
  p.offset = p0.offset + (unsigned long long)r * a.per_call;
  dw_prologue(a.w, i, lane, a.live, a.Ls, a.nlive, a.dirkind, a.dd, a.tkind, a.ta, a.tb, a.was_starting, p);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // (the call-by-call path evaluates every walker's row, acceptable or not; the value of a row that is not acceptable is
  // never looked at: update_walker's `hit` tests acceptable first)
  const double L = loglike_wave(a.lkind, a.w.pnew + (size_t)i * a.w.nparams, a.w.d, a.aux, a.sigma, lane);
  if (lane == 0) a.w.Lnew[i] = L;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  dw_update(a.w, i, lane, p.Lmin, a.ly);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) {
    const uint8_t succ = a.w.success[i];
    a.rflags[(size_t)r * a.w.P + i] = (uint8_t)((a.w.movable[i] ? 1 : 0) | (a.w.acceptable[i] ? 2 : 0) | (succ ? 4 : 0) |
                                                (a.was_starting[i] ? 8 : 0));
    a.rdist2[(size_t)r * a.w.P + i] = a.w.dist2[i];
    a.rlast[i] = r + 1;
  }
}

// Rounds r0, r0 + 1, ... of ONE walker with its state in REGISTERS (the form above passes every intermediate through global
// memory between the lanes of the wave: ~6 us per round, four fences and half a dozen dependent L2 round trips; here a round
// is a few hundred instructions).  Same arithmetic, stage by stage, as dw_prologue / loglike_wave / dw_update -- the shared
// cores (dw_direction, loglike_wave's pair layout, update_walker, move_distance_regs) are the SAME functions -- so the
// resident state afterwards is bit for bit that of the memory form.  What rounds after the first of a call never do is
// dropped: step_back (every chain entry appended under this threshold lies above it; round 0 removed the others) and
// restarts (a walker that could not restart in round 0 -- no live point above the threshold -- cannot later).
// Preconditions (rounds_in_registers): even d <= 64 (loglike_wave's pair layout, move_distance_regs), affine layer or none.
// `until_finished`: stop when the walker has its nsteps (the ring walker); returns the round index after the last one made.
// Hand-shake between the ring walker's wave and all the others inside ONE launch (k_walk_rounds): ONE word, ctl[3] = the last
// round the ring walker has committed to make (it writes r BEFORE making round r) + kRingThrough once it has made its last;
// ctl[5] = a follower gave up waiting.  A follower makes round r when the word says >= r: it trails the ring walker by less than a
// round instead of repeating its rounds after it (k_walk_ring, then k_walk_rest: the same chain of round latencies twice).
// Relaxed atomics at agent scope (the word lives in L2; nobody reads DATA the other side wrote, so no cache is flushed or
// invalidated: with release / acquire every poll of 1 000 waves invalidated its CU's caches -- 22 us per round instead of 5).
constexpr int kRingThrough = 1 << 30;
__device__ __forceinline__ void ring_commits(int *ctl, int r, int lane) {
  if (lane == 0) __hip_atomic_store(ctl + 3, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ring_is_through(int *ctl, int last_round, int lane) {
  if (lane == 0) __hip_atomic_store(ctl + 3, last_round | kRingThrough, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool follower_may_make(int *ctl, int r, int lane) {
  int ok = 0;
  if (lane == 0) {
    for (int spins = 0;; ++spins) {
      const int word = __hip_atomic_load(ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((word & (kRingThrough - 1)) >= r) {
        ok = 1;
        break;
      }
      if (word & kRingThrough) break;   // the ring walker is through and never made round r
      if (spins > (1 << 21)) {   // ~1 s: never seen; the call then fails (mlf_walkers_rounds_dev) instead of hanging the device
        __hip_atomic_store(ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// role: 0 = no hand-shake, 1 = the ring walker (commits to every round before it makes it), 2 = a follower (waits for the commit)
__device__ int dw_rounds_regs(const RoundsArgs &a, int i, int lane, int r0, int r1, bool until_finished, const StepParams &p0, int role = 0) {
  const WalkState &w = a.w;
  const int d = w.d;
  long long gen = w.generation[i];
  double tcur = w.currentt[i], left = w.left[i], right = w.right[i];
  uint8_t sl = w.sl[i], sr = w.sr[i];
  const bool have = lane < d;
  double u0 = 0.0, v = 0.0;   // the lane's coordinate of the chain's current point and of the slice direction
  if (gen >= 0 && have) u0 = w.allu[((size_t)i * w.G + gen) * d + lane];
  if (have) v = w.currentv[(size_t)i * d + lane];
  int hw = 2;
  while (2 * hw < d) hw *= 2;
  int r = r0;
  for (; r < r1; ++r) {
    if (until_finished && gen == (long long)(w.G - 1)) break;
    if (role == 1) ring_commits(a.ctl, r, lane);
    if (role == 2 && !follower_may_make(a.ctl, r, lane)) break;
    const unsigned long long offset = p0.offset + (unsigned long long)r * a.per_call;
    const bool movable = gen >= 0 && gen < w.G - 1;
    uint8_t flags = gen < 0 ? 8 : 0;
    double dist2 = qnan();
    if (gen >= 0 && !isfinite(tcur)) {   // new slice (dw_brackets_philox; a walker that never started has no point to slice from
                                         // -- the memory form draws a direction for it all the same, which nothing ever reads)
      bool bsl = sl != 0, bsr = sr != 0;
      double vr[2];
      dw_brackets_philox(w, i, lane, p0.scale, a.dirkind, p0.dirscale, a.dd, p0.seed, offset, tcur, left, right, bsl, bsr, vr);
      sl = bsl ? 1 : 0;
      sr = bsr ? 1 : 0;
      v = vr[0];
    } else if (gen < 0 && !isfinite(tcur)) {
      bool bsl = sl != 0, bsr = sr != 0;
      dw_brackets_philox(w, i, lane, p0.scale, a.dirkind, p0.dirscale, a.dd, p0.seed, offset, tcur, left, right, bsl, bsr, nullptr);
      sl = bsl ? 1 : 0;
      sr = bsr ? 1 : 0;
    }
    if (movable) {
      flags |= 1;
      // evolve, first half (dw_propose)
      double tu;
      if (sl) {
        tu = left;
      } else if (sr) {
        tu = right;
      } else {
        unsigned r4[4];
        philox_block(p0.seed, 3u, offset + (unsigned long long)i, r4);
        const double un = u01(r4[0], r4[1]);
        const double range = right - left;
        const double scaled = range * un;
        tcur = left + scaled;
        tu = tcur;
      }
      const double step = v * tu;
      const double x = u0 + step;
      const bool acceptable = __all(!have || inside_open_unit(x));
      double pcoord = x;
      if (a.tkind == 1) {
        const double m = x * a.ta;
        pcoord = m + a.tb;
      } else if (a.tkind == 2) {
        const double m = x * a.ta;
        pcoord = m * a.tb;
      }
      bool hit = false;
      double L = 0.0;
      if (acceptable) {   // wave-uniform
        flags |= 2;
        // loglike_wave's pair layout: lane l < hw holds parameters 2 l, 2 l + 1
        const double x0 = __shfl(pcoord, (2 * lane) & 63, 64), x1 = __shfl(pcoord, (2 * lane + 1) & 63, 64);
        L = loglike_pairs(a.lkind, x0, x1, d, hw, a.aux, a.sigma, lane);
        hit = L > p0.Lmin;
      }
      // evolve_update + advance (dw_update)
      const bool success = update_walker(hit, tcur, left, right, sl, sr);
      if (success) {
        flags |= 4;
        const long long g1 = gen + 1;
        if (have) {
          w.allu[((size_t)i * w.G + g1) * d + lane] = x;
          w.currentp[(size_t)i * w.nparams + lane] = pcoord;
        }
        if (lane == 0) w.allL[(size_t)i * w.G + g1] = L;
        if (a.ly.kind >= 0) dist2 = move_distance_regs(a.ly, d, lane, u0, x);
        gen = g1;
        u0 = x;
      }
    }
    if (lane == 0) {
      a.rflags[(size_t)r * w.P + i] = flags;
      a.rdist2[(size_t)r * w.P + i] = dist2;
    }
  }
  if (lane == 0) {
    w.generation[i] = gen;
    w.currentt[i] = tcur;
    w.left[i] = left;
    w.right[i] = right;
    w.sl[i] = sl;
    w.sr[i] = sr;
    a.rlast[i] = r;   // rounds [r, ...) of this call carry no flags of this walker (k_walk_round_stats)
  }
  return r;
}

// the layer of the move diagnostics staged in LDS for the register form (d <= 64: 32 KiB + centre + wrap shifts): the chain
// of move_distance_regs reads one matrix element per step, and from global memory every step waited for its own L2 round trip
struct LayerLds {
  double mat[64 * 64];
  double ctr[64];
  double wrap[64];
};
__device__ WalkLayer stage_layer(const WalkLayer &ly, int d, LayerLds &lds, int tid, int nthreads) {
  WalkLayer out = ly;
  if (ly.kind != 0) return out;
  for (int e = tid; e < d * d; e += nthreads) lds.mat[e] = ly.mat[e];
  for (int e = tid; e < d; e += nthreads) {
    lds.ctr[e] = ly.ctr[e];
    if (ly.wrap) lds.wrap[e] = ly.wrap[e];
  }
  out.mat = lds.mat;
  out.ctr = lds.ctr;
  if (ly.wrap) out.wrap = lds.wrap;
  return out;
}

__device__ __forceinline__ bool rounds_in_registers(const RoundsArgs &a) {
  return !(a.w.d & 1) && a.w.d <= 64 && a.w.nparams == a.w.d && (a.ly.kind == 0 || a.ly.kind < 0) && !a.force_memory_form;
}

__global__ __launch_bounds__(64) void k_walk_round0(RoundsArgs a) {
  const StepParams p = *a.sp;
  for (int i = blockIdx.x; i < a.w.P; i += gridDim.x) dw_round(a, i, threadIdx.x, 0, p);
}

// setup_start's ring shift (:456-462), as k_walk_harvest makes it, and the hand-shake words of the launch that follows
__global__ __launch_bounds__(256) void k_walk_pick(RoundsArgs a) {
  __shared__ int s_n[256];
  const WalkState &w = a.w;
  int n = 0;
  for (int i = threadIdx.x; i < w.P; i += 256) n += a.was_starting[i] ? 1 : 0;
  s_n[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_n[threadIdx.x] += s_n[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long r = *a.ring;
    if (s_n[0] > 0 && s_n[0] < w.P)
      for (int guard = 0; guard < w.P && a.was_starting[r]; ++guard) r = (r + 1) % w.P;
    a.ctl[0] = 0;
    a.ctl[1] = (int)r;
    a.ctl[2] = 0;
    a.ctl[3] = 0;
    a.ctl[4] = 0;
    a.ctl[5] = 0;
  }
}

// Rounds 1 ... of everybody in one launch.  Workgroup 0 (dispatched first, so always resident): the ring walker, until it has
// finished -- in those rounds nothing restarts that did not restart in round 0 (same threshold, same live points), so the ring
// index stays where round 0 left it.  Workgroups 1 ...: the other walkers, each making round r as soon as the ring walker has
// committed to it (dw_rounds_regs, role 2).  Before: k_walk_ring (the ring walker alone, R rounds), THEN k_walk_rest (everybody else,
// the same R rounds): 84 us median / 148 mean per call at C3's shape.
__global__ __launch_bounds__(64) void k_walk_rounds(RoundsArgs a) {
  const WalkState &w = a.w;
  const StepParams p = *a.sp;
  const int lane = threadIdx.x;
  const int ring = a.ctl[1];
  const bool regs = rounds_in_registers(a);
  __shared__ LayerLds lds;
  RoundsArgs al = a;
  if (blockIdx.x != 0 || a.phase == 2) {
    if (a.phase == 1) return;
    // no round 1 (the ring walker finished in round 0): nothing to do -- in particular no 20 KB layer matrix into LDS per walker
    if (a.phase == 2 ? a.ctl[0] <= 1 : !follower_may_make(a.ctl, 1, lane)) return;
    if (regs) {
      al.ly = stage_layer(a.ly, w.d, lds, lane, 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int first = a.phase == 2 ? (int)blockIdx.x : (int)blockIdx.x - 1, stride = a.phase == 2 ? (int)gridDim.x : (int)gridDim.x - 1;
    for (int i = first; i < w.P; i += stride) {
      if (i == ring) continue;
      if (a.phase == 2) {   // the ring walker is through (the launch before): R is known, nobody polls
        const int R = a.ctl[0];
        if (regs) {
          dw_rounds_regs(al, i, lane, 1, R, false, p, 0);
        } else {
          for (int r = 1; r < R; ++r) dw_round(a, i, lane, r, p);
        }
      } else if (regs) {
        dw_rounds_regs(al, i, lane, 1, a.max_rounds, false, p, 2);
      } else {
        for (int r = 1; r < a.max_rounds; ++r) {
          if (!follower_may_make(a.ctl, r, lane)) break;
          dw_round(a, i, lane, r, p);
        }
      }
    }
    return;
  }
  int R = 1;
  if (regs) {
    if (w.generation[ring] != (long long)(w.G - 1)) {   // (wave-uniform) there will be rounds
      al.ly = stage_layer(a.ly, w.d, lds, lane, 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    R = dw_rounds_regs(al, ring, lane, 1, a.max_rounds, true, p, 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    while (w.generation[ring] != (long long)(w.G - 1) && R < a.max_rounds) {
      ring_commits(a.ctl, R, lane);
      dw_round(a, ring, lane, R, p);
      ++R;
    }
  }
  ring_is_through(a.ctl, R - 1, lane);   // no further commits
  const bool found = w.generation[ring] == (long long)(w.G - 1);
  const size_t row = ((size_t)ring * w.G + (w.G - 1)) * w.d;
  double *rec = a.rec;
  if (lane == 0) {
    a.ctl[0] = R;
    a.ctl[2] = found ? 1 : 0;
    rec[0] = found ? 1.0 : 0.0;
    rec[1] = found ? w.allL[(size_t)ring * w.G + (w.G - 1)] : qnan();
    rec[2] = w.left[ring];
    rec[3] = w.right[ring];
    rec[4] = (double)R;
  }
  if (found) {
    for (int k = lane; k < w.d; k += 64) rec[9 + k] = w.allu[row + k];
    for (int k = lane; k < w.nparams; k += 64) rec[9 + w.d + k] = w.currentp[(size_t)ring * w.nparams + k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (found) {   // popstepsampler.py:678-681
    for (int e = lane; e < w.G * w.d; e += 64) w.allu[(size_t)ring * w.G * w.d + e] = qnan();
    for (int e = lane; e < w.G; e += 64) w.allL[(size_t)ring * w.G + e] = qnan();
    if (lane == 0) {
      w.generation[ring] = -1;
      w.currentt[ring] = qnan();
    }
  }
  if (lane == 0) {   // shift(), :605-609
    const long long next = found ? ((long long)ring + 1) % w.P : (long long)ring;
    *a.ring = next;
    rec[9 + w.d + w.nparams] = (double)next;
  }
}

// rows[r] = (nc, nmovable, nsuccess, nfar, sum log(dist / ref + 1e-10)) of round r, summed in k_walk_stats' order per
// 1024-walker chunk and k_walk_harvest's order over the chunks (P <= 1024: one chunk; the sums are then the same doubles)
// one 1024-walker chunk of round r: (nc, nmovable, nsuccess, nfar, sum log(dist / ref + 1e-10)) in k_walk_stats' order; thread 0 returns it
__device__ __forceinline__ void round_stats_chunk(const RoundsArgs &a, int r, int i0, double (*part)[5], double (&out)[5]) {
  const double r2 = a.sp->r2;
  const double ref = sqrt(r2);
  const int P = a.w.P;
  double nc = 0, nmov = 0, nsucc = 0, nfar = 0, slog = 0;
  for (int j = threadIdx.x; j < kStatsChunk; j += 256) {
    const int i = i0 + j;
    if (i >= P) continue;
    if (r >= a.rlast[i]) continue;      // the walker made no such round (the ring walker's R is everybody's bound, see rlast)
    const uint8_t f = a.rflags[(size_t)r * P + i];
    if (!(f & 1)) continue;
    nmov += 1;
    nc += (f & 2) ? 1 : 0;
    if (f & 4) {
      nsucc += 1;
      const double d2 = a.rdist2[(size_t)r * P + i];
      if (!isnan(d2)) {
        nfar += (d2 > r2) ? 1 : 0;
        slog += log(sqrt(d2) / ref + 1e-10);
      }
    }
  }
  part[threadIdx.x][0] = nc;
  part[threadIdx.x][1] = nmov;
  part[threadIdx.x][2] = nsucc;
  part[threadIdx.x][3] = nfar;
  part[threadIdx.x][4] = slog;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int c = 0; c < 5; ++c) part[threadIdx.x][c] += part[threadIdx.x + off][c];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int c = 0; c < 5; ++c) out[c] = part[0][c];
  __syncthreads();
}

// rows[r] of round r, summed in k_walk_stats' order per 1024-walker chunk and k_walk_harvest's order over the chunks.  One chunk
// (populations up to 1024 walkers): one workgroup per round ...
__global__ __launch_bounds__(256) void k_walk_round_stats(RoundsArgs a) {
  __shared__ double part[256][5];
  const int r = blockIdx.x;
  if (r >= a.ctl[0]) return;
  double tot[5] = {0, 0, 0, 0, 0};
  for (int i0 = 0; i0 < a.w.P; i0 += kStatsChunk) {
    double c5[5];
    round_stats_chunk(a, r, i0, part, c5);
    if (threadIdx.x == 0)
      for (int c = 0; c < 5; ++c) tot[c] += c5[c];
  }
  if (threadIdx.x == 0) {
    for (int c = 0; c < 5; ++c) a.rows[(size_t)r * 5 + c] = tot[c];
    if (r == 0) a.rec[5] = (double)a.ctl[5];   // a follower gave up waiting (k_walk_rounds): the call fails
  }
}
// ... more: one workgroup per (round, chunk) and a second launch that adds the chunks up as k_walk_harvest does (one workgroup
// walking the 98 chunks of a 10^5-walker population took 0.44 ms of a 0.75 ms call -- and added them up one after the other, which
// is not the harvest's tree: the sum of logarithms differed in its last bits from three chunks on; tests now cover 4 500 walkers)
__global__ __launch_bounds__(256) void k_walk_round_stats_chunks(RoundsArgs a, int nchunks) {
  __shared__ double part[256][5];
  const int r = blockIdx.x, c0 = blockIdx.y;
  if (r >= a.ctl[0]) return;
  double c5[5];
  round_stats_chunk(a, r, c0 * kStatsChunk, part, c5);
  if (threadIdx.x == 0)
    for (int c = 0; c < 5; ++c) a.rparts[((size_t)r * nchunks + c0) * 5 + c] = c5[c];
}
__global__ __launch_bounds__(256) void k_walk_round_stats_sum(RoundsArgs a, int nchunks) {
  __shared__ double part[256][5];
  const int r = blockIdx.x;
  if (r >= a.ctl[0]) return;
  double v[5] = {0, 0, 0, 0, 0};   // k_walk_harvest's order over the chunks: thread t takes chunks t, t + 256, ..., then the tree
  for (int b = threadIdx.x; b < nchunks; b += 256)
    for (int c = 0; c < 5; ++c) v[c] += a.rparts[((size_t)r * nchunks + b) * 5 + c];
  for (int c = 0; c < 5; ++c) part[threadIdx.x][c] = v[c];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int c = 0; c < 5; ++c) part[threadIdx.x][c] += part[threadIdx.x + off][c];
    __syncthreads();
  }
  if (threadIdx.x < 5) a.rows[(size_t)r * 5 + threadIdx.x] = part[0][threadIdx.x];
  if (r == 0 && threadIdx.x == 0) a.rec[5] = (double)a.ctl[5];
}

__global__ void k_within_unit_cube(const double *u, int n, int d, uint8_t *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  for (int k = 0; k < d; ++k) ok = ok && inside_open_unit(u[(size_t)i * d + k]);
  out[i] = ok ? 1 : 0;
}

__global__ void k_bisect_draw(const double *left, const double *right, const uint8_t *sl, const uint8_t *sr,
                              const double *unif, int n, double *currentt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || sl[i] || sr[i]) return;
  const double range = right[i] - left[i];
  const double scaled = range * unif[i];
  currentt[i] = left[i] + scaled;
}

__global__ void k_evolve_propose(const double *currentu, const double *currentv, const double *left,
                                 const double *right, const uint8_t *sl, const uint8_t *sr, const double *currentt,
                                 int n, int d, double *unew) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * d) return;
  const int i = e / d;
  const double t = sl[i] ? left[i] : (sr[i] ? right[i] : currentt[i]);
  const double step = currentv[e] * t;
  unew[e] = currentu[e] + step;
}

__global__ void k_evolve_update(const uint8_t *acceptable, const double *Lnew, double Lmin, double *currentt,
                                double *left, double *right, uint8_t *sl, uint8_t *sr, uint8_t *success, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool hit = acceptable[i] != 0 && Lnew[i] > Lmin;
  success[i] = update_walker(hit, currentt[i], left[i], right[i], sl[i], sr[i]) ? 1 : 0;
}

// popstepsampler.py:26-61; nanmax / nanmin semantics
__global__ void k_line_intersection(const double *origin, const double *direction, int n, int d, double *tleft,
                                    double *tright) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double lo = qnan(), hi = qnan();
  for (int k = 0; k < d; ++k) {
    const double m = 1.0 / direction[(size_t)i * d + k];
    const double nn = m * (origin[(size_t)i * d + k] - 0.5);
    const double kk = fabs(m) * 0.5;
    const double t1 = -nn - kk;
    const double t2 = -nn + kk;
    if (!isnan(t1) && (isnan(lo) || t1 > lo)) lo = t1;
    if (!isnan(t2) && (isnan(hi) || t2 < hi)) hi = t2;
  }
  tleft[i] = lo;
  tright[i] = hi;
}

// stepfuncs.pyx:537-630 in one workgroup.  Workers are read in order l = 0..popsize-1 and only
// interact through the point they serve, so the points are processed in parallel (one lane per
// point walks the worker list in order); the worker list is staged through LDS in tiles.
__global__ __launch_bounds__(1024) void k_slice_update(const double *t, double *tleft, double *tright,
                                                       const double *pL, const double *pu, const double *pp,
                                                       long long *worker_running, long long *status,
                                                       double threshold, double shrink, double *allu, double *allL,
                                                       double *allp, int popsize, int d, int nparams,
                                                       long long *discarded, long long *zlist) {
  __shared__ double s_t[1024], s_L[1024];
  __shared__ int s_w[1024];
  __shared__ unsigned s_scan[1024];
  __shared__ long long s_disc[1024];
  __shared__ unsigned s_total;
  const int tid = threadIdx.x;
  long long ndisc = 0;
  for (int w0 = 0; w0 < popsize; w0 += 1024) {   // points w0 .. w0+1023
    const int w = w0 + tid;
    double lo = 0, hi = 0;
    long long st = 1;
    int taken = -1;
    if (w < popsize) {
      lo = tleft[w];
      hi = tright[w];
      st = status[w];
    }
    for (int l0 = 0; l0 < popsize; l0 += 1024) {
      __syncthreads();
      if (l0 + tid < popsize) {
        s_t[tid] = t[l0 + tid];
        s_L[tid] = pL[l0 + tid];
        s_w[tid] = (int)worker_running[l0 + tid];
      }
      __syncthreads();
      const int m = popsize - l0 < 1024 ? popsize - l0 : 1024;
      if (w < popsize)
        for (int j = 0; j < m; ++j) {
          if (s_w[j] != w) continue;
          const double tl = s_t[j];
          if (tl > hi || tl < lo) {
            if (s_L[j] > threshold) ++ndisc;
            continue;
          }
          if (0 < tl && tl < hi) hi = tl / shrink;
          if (0 > tl && tl > lo) lo = tl / shrink;
          if (s_L[j] > threshold && st == 0) {
            st = 1;
            taken = l0 + j;
          }
        }
    }
    if (w < popsize) {
      tleft[w] = lo;
      tright[w] = hi;
      status[w] = st;
      if (taken >= 0) {
        for (int k = 0; k < d; ++k) allu[(size_t)w * d + k] = pu[(size_t)taken * d + k];
        allL[w] = pL[taken];
        for (int k = 0; k < nparams; ++k) allp[(size_t)w * nparams + k] = pp[(size_t)taken * nparams + k];
      }
    }
  }
  s_disc[tid] = ndisc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) s_disc[tid] += s_disc[tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    *discarded = s_disc[0];
    s_total = 0;
  }
  __syncthreads();
  // unfinished points in ascending order, then dealt round-robin to the workers
  for (int k0 = 0; k0 < popsize; k0 += 1024) {
    const int k = k0 + tid;
    const unsigned flag = (k < popsize && status[k] == 0) ? 1u : 0u;
    s_scan[tid] = flag;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const unsigned v = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const unsigned base = s_total;
    if (flag) zlist[base + s_scan[tid] - 1] = k;
    __syncthreads();
    if (tid == 1023) s_total = base + s_scan[1023];
    __syncthreads();
  }
  const unsigned nz = s_total;
  if (nz > 0)
    for (int j = tid; j < popsize; j += 1024) worker_running[j] = zlist[j % nz];
}

__global__ void k_row_dist2(const double *a, const double *b, int n, int d, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int k = 0; k < d; ++k) {
    const double diff = a[(size_t)i * d + k] - b[(size_t)i * d + k];
    acc += diff * diff;
  }
  out[i] = acc;
}

// ------------------------------------------------------------------ launchers -------------------
static inline dim3 grid_for(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }
// wave-per-walker kernels: one one-wave workgroup per walker (fewer, looping waves measured slower: 0.88 vs 0.71 ms
// per step at 10^5 walkers); the loop in the kernels only matters beyond 4 M walkers
static inline dim3 walker_grid(int P) { return dim3((unsigned)(P < (1 << 22) ? P : (1 << 22))); }

void launch_walk_reset(const WalkState &w, hipStream_t s) {
  const long long n = (long long)w.P * w.G * w.d;
  hipLaunchKernelGGL(k_walk_reset, grid_for(n), dim3(256), 0, s, w);
}

void launch_walk_step_back(const WalkState &w, double Lmin, long long *gmax_scratch, uint8_t *flags, hipStream_t s,
                           const StepParams *sp) {
  hipLaunchKernelGGL(k_max_generation, dim3(1), dim3(256), 0, s, w.generation, w.P, gmax_scratch);
  hipLaunchKernelGGL(k_step_back, grid_for(w.P), dim3(256), 0, s, Lmin, w.allL, w.P, w.G, w.generation, w.currentt,
                     gmax_scratch, w.sl, w.sr, flags, sp);
}

void launch_walk_restart_philox(const WalkState &w, const double *live, const double *Ls, int nlive, double Lmin,
                                unsigned long long seed, unsigned long long offset, long long *ring, hipStream_t s,
                                const StepParams *sp) {
  hipLaunchKernelGGL(k_walk_ring_shift, dim3(1), dim3(256), 0, s, w, ring);
  hipLaunchKernelGGL(k_walk_restart_philox, walker_grid(w.P), dim3(64), 0, s, w, live, Ls, nlive, Lmin, seed, offset, sp);
}

void launch_walk_prologue(const WalkState &w, const double *live, const double *Ls, int nlive, int dirkind, WalkDirData dd,
                          int tkind, double ta, double tb, uint8_t *was_starting, const StepParams &p, const StepParams *sp,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_walk_prologue, walker_grid(w.P), dim3(64), 0, s, w, live, Ls, nlive, dirkind, dd, tkind, ta, tb,
                     was_starting, p, sp);
}

void launch_walk_start(const WalkState &w, const long long *idx, int n, const double *rows, const double *L,
                       hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_walk_start, grid_for((long long)n * w.d), dim3(256), 0, s, w, idx, n, rows, L);
}

void launch_walk_points(const WalkState &w, const long long *idx, int n, double *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_walk_points, grid_for((long long)n * w.d), dim3(256), 0, s, w, idx, n, out);
}

void launch_walk_brackets(const WalkState &w, const long long *idx, int n, double scale, const double *v_rows,
                          hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_walk_brackets, grid_for((long long)n * w.d), dim3(256), 0, s, w, idx, n, scale, v_rows);
}

void launch_walk_brackets_philox(const WalkState &w, double scale, int kind, double dirscale, WalkDirData dd,
                                 unsigned long long seed, unsigned long long offset, hipStream_t s, const StepParams *sp) {
  hipLaunchKernelGGL(k_walk_brackets_philox, walker_grid(w.P), dim3(64), 0, s, w, scale, kind, dirscale, dd, seed,
                     offset, sp);
}

void launch_walk_propose(const WalkState &w, const double *unif, unsigned long long seed, unsigned long long offset,
                         hipStream_t s, const StepParams *sp) {
  hipLaunchKernelGGL(k_walk_propose, walker_grid(w.P), dim3(64), 0, s, w, unif, seed, offset, sp);
}

void launch_walk_transform(const WalkState &w, int tkind, double a, double b, hipStream_t s) {
  hipLaunchKernelGGL(k_walk_transform, grid_for((long long)w.P * w.d), dim3(256), 0, s, w, tkind, a, b);
}

void launch_walk_expand(const WalkState &w, const unsigned *blk, const double *pc, const double *Lc, hipStream_t s) {
  hipLaunchKernelGGL(k_walk_expand, grid_for(w.P), dim3(256), 0, s, w, blk, pc, Lc);
}

void launch_walk_update(const WalkState &w, double Lmin, WalkLayer layer, hipStream_t s, const StepParams *sp) {
  hipLaunchKernelGGL(k_walk_update, walker_grid(w.P), dim3(64), 0, s, w, Lmin, sp, layer);
}

__global__ void k_walk_scatter_live(const double *rows, const double *Ls, const long long *idx, int n, int d, double *live, double *liveL) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * d) return;
  const int j = e / d, k = e - j * d;
  const long long i = idx[j];
  live[(size_t)i * d + k] = rows[(size_t)j * d + k];
  if (k == 0) liveL[i] = Ls[j];
}

void launch_walk_scatter_live(const double *rows, const double *Ls, const long long *idx, int n, int d, double *live, double *liveL,
                              hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_walk_scatter_live, grid_for((long long)n * d), dim3(256), 0, s, rows, Ls, idx, n, d, live, liveL);
}

void launch_walk_rounds(const RoundsArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_walk_round0, walker_grid(a.w.P), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_walk_pick, dim3(1), dim3(256), 0, s, a);
  // populations the chip holds at once: one launch, the followers trail the ring walker.  Larger ones are throughput-bound: the
  // ring walker first, alone (its rounds are quick with nobody next to it), then everybody else (popsize 100 000, d = 50:
  // 0.77 against 1.39 ms per call)
  RoundsArgs b = a;
  if (a.w.P <= 4096) {
    b.phase = 0;
    hipLaunchKernelGGL(k_walk_rounds, dim3(walker_grid(a.w.P).x + 1), dim3(64), 0, s, b);
  } else {
    b.phase = 1;
    hipLaunchKernelGGL(k_walk_rounds, dim3(1), dim3(64), 0, s, b);
    b.phase = 2;   // 2048 workgroups walk the population: the layer matrix is staged 2048 times, not once per walker, and a call
                   // without further rounds does not pay for 10^5 workgroups that start (33 KB of LDS each) only to leave
    hipLaunchKernelGGL(k_walk_rounds, dim3(2048), dim3(64), 0, s, b);
  }
  const int nchunks = (a.w.P + kStatsChunk - 1) / kStatsChunk;
  if (nchunks <= 1 || !a.rparts) {
    hipLaunchKernelGGL(k_walk_round_stats, dim3((unsigned)a.max_rounds), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_walk_round_stats_chunks, dim3((unsigned)a.max_rounds, (unsigned)nchunks), dim3(256), 0, s, a, nchunks);
    hipLaunchKernelGGL(k_walk_round_stats_sum, dim3((unsigned)a.max_rounds), dim3(256), 0, s, a, nchunks);
  }
}

void launch_walk_harvest(const WalkState &w, long long ring, long long *ring_dev, double r2, double *rec, double *partials,
                         hipStream_t s, const StepParams *sp, const uint8_t *was_starting) {
  const int nrows = (w.P + kStatsChunk - 1) / kStatsChunk;
  hipLaunchKernelGGL(k_walk_stats, dim3(nrows), dim3(256), 0, s, w, r2, sp, partials, was_starting);
  hipLaunchKernelGGL(k_walk_harvest, dim3(1), dim3(256), 0, s, w, ring, ring_dev, r2, rec, sp, was_starting, partials);
}

void launch_within_unit_cube(const double *u, int n, int d, uint8_t *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_within_unit_cube, grid_for(n), dim3(256), 0, s, u, n, d, out);
}

void launch_evolve_propose(const double *currentu, const double *currentv, const double *left, const double *right,
                           const uint8_t *sl, const uint8_t *sr, const double *currentt, int n, int d, double *unew,
                           hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_evolve_propose, grid_for((long long)n * d), dim3(256), 0, s, currentu, currentv, left, right,
                     sl, sr, currentt, n, d, unew);
}

void launch_bisect_draw(const double *left, const double *right, const uint8_t *sl, const uint8_t *sr,
                        const double *unif_full, int n, double *currentt, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_bisect_draw, grid_for(n), dim3(256), 0, s, left, right, sl, sr, unif_full, n, currentt);
}

void launch_evolve_update(const uint8_t *acceptable, const double *Lnew_full, double Lmin, double *currentt,
                          double *left, double *right, uint8_t *sl, uint8_t *sr, uint8_t *success, int n,
                          hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_evolve_update, grid_for(n), dim3(256), 0, s, acceptable, Lnew_full, Lmin, currentt, left,
                     right, sl, sr, success, n);
}

void launch_step_back(double Lmin, double *allL, int n, int G, long long *generation, double *currentt,
                      long long *gmax_scratch, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_max_generation, dim3(1), dim3(256), 0, s, generation, n, gmax_scratch);
  hipLaunchKernelGGL(k_step_back, grid_for(n), dim3(256), 0, s, Lmin, allL, n, G, generation, currentt, gmax_scratch,
                     (const uint8_t *)nullptr, (const uint8_t *)nullptr, (uint8_t *)nullptr, (const StepParams *)nullptr);
}

void launch_line_intersection(const double *origin, const double *direction, int n, int d, double *tleft,
                              double *tright, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_line_intersection, grid_for(n), dim3(256), 0, s, origin, direction, n, d, tleft, tright);
}

void launch_slice_update(const double *t, double *tleft, double *tright, const double *pL, const double *pu,
                         const double *pp, long long *worker_running, long long *status, double threshold,
                         double shrink, double *allu, double *allL, double *allp, int popsize, int d, int nparams,
                         long long *discarded, hipStream_t s) {
  if (popsize <= 0) return;
  // zlist: the caller provides popsize int64 after `discarded`
  hipLaunchKernelGGL(k_slice_update, dim3(1), dim3(1024), 0, s, t, tleft, tright, pL, pu, pp, worker_running, status,
                     threshold, shrink, allu, allL, allp, popsize, d, nparams, discarded, discarded + 1);
}

void launch_row_dist2(const double *a, const double *b, int n, int d, double *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_row_dist2, grid_for(n), dim3(256), 0, s, a, b, n, d, out);
}

}  // namespace mlf
