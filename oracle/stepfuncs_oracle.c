/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of the population step-sampler state machine of the reference
 * (SURVEY.md 8f row f1): ultranest/stepfuncs.pyx and the two geometry helpers of
 * ultranest/popstepsampler.py.  Plain C, one walker at a time, same operation order as the
 * reference (no FMA: built with -ffp-contract=off like the rest of the oracle).
 *
 * Parity status: PINNED -- tests/test_stepfuncs_golden.py compares every function with vectors
 * recorded from the real (cythonized) reference by tests/golden/make_golden.py (group g9).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* stepfuncs.pyx:20-33: acceptable[i] = all(0 < u[i,:] < 1) */
void orc_within_unit_cube(const double *u, size_t n, size_t d, uint8_t *acceptable) {
  for (size_t i = 0; i < n; ++i) {
    uint8_t ok = 1;
    for (size_t j = 0; j < d; ++j) {
      const double x = u[i * d + j];
      if (!(0.0 < x && x < 1.0)) {
        ok = 0;
        break;
      }
    }
    acceptable[i] = ok;
  }
}

/* stepfuncs.pyx:56-70: the three mutually exclusive walker states */
void orc_evolve_prepare(const uint8_t *searching_left, const uint8_t *searching_right, size_t n,
                        uint8_t *search_right, uint8_t *bisecting) {
  for (size_t i = 0; i < n; ++i) {
    search_right[i] = (!searching_left[i] && searching_right[i]) ? 1 : 0;
    bisecting[i] = !(searching_left[i] || searching_right[i]) ? 1 : 0;
  }
}

/* stepfuncs.pyx:253-259 (inside evolve): the proposed point of every walker.  Walkers stepping
 * out propose their current bracket end, bisecting walkers the coordinate currentt (already
 * drawn: currentt = left + (right - left) * U, numpy's legacy uniform).  unew = u + v * t with
 * a separate multiply and add per coordinate. */
void orc_evolve_propose(const double *currentu, const double *currentv, const double *current_left,
                        const double *current_right, const uint8_t *searching_left,
                        const uint8_t *searching_right, const double *currentt, size_t n, size_t d,
                        double *unew) {
  for (size_t i = 0; i < n; ++i) {
    double t;
    if (searching_left[i])
      t = current_left[i];
    else if (searching_right[i])
      t = current_right[i];
    else
      t = currentt[i];
    for (size_t k = 0; k < d; ++k) {
      const double step = currentv[i * d + k] * t;
      unew[i * d + k] = currentu[i * d + k] + step;
    }
  }
}

/* numpy legacy uniform(low, high): low + (high - low) * U (numpy/random/src/distributions,
 * random_uniform; third-party arithmetic, pinned by the g9 evolve vectors) */
void orc_bisect_draw(const double *current_left, const double *current_right, const uint8_t *bisecting,
                     const double *unif_compact, size_t n, double *currentt) {
  size_t j = 0;
  for (size_t i = 0; i < n; ++i)
    if (bisecting[i]) {
      const double range = current_right[i] - current_left[i];
      const double scaled = range * unif_compact[j++];
      currentt[i] = current_left[i] + scaled;
    }
}

/* stepfuncs.pyx:99-183.  Lnew is compacted over acceptable walkers.  success must be zeroed by
 * the caller (the reference passes np.zeros_like). */
void orc_evolve_update(const uint8_t *acceptable, const double *Lnew, double Lmin,
                       const uint8_t *search_right, const uint8_t *bisecting, double *currentt,
                       double *current_left, double *current_right, uint8_t *searching_left,
                       uint8_t *searching_right, uint8_t *success, size_t n) {
  size_t j = 0;
  for (size_t k = 0; k < n; ++k)
    if (acceptable[k]) {
      if (Lnew[j] > Lmin) success[k] = 1;
      ++j;
    }
  const double my_nan = (double)NAN; /* reference: cdef float my_nan, widened on store */
  for (size_t i = 0; i < n; ++i) {
    if (success[i]) {
      if (searching_left[i])
        current_left[i] *= 2;
      else if (search_right[i])
        current_right[i] *= 2;
    } else {
      if (searching_left[i])
        searching_left[i] = 0;
      else if (search_right[i])
        searching_right[i] = 0;
    }
    if (bisecting[i]) {
      if (currentt[i] < 0)
        current_left[i] = currentt[i];
      else
        current_right[i] = currentt[i];
      if (success[i]) currentt[i] = my_nan;
    } else {
      success[i] = 0;
    }
  }
}

/* stepfuncs.pyx:285-334.  allL is (n, ngen) row-major.  The reference peels one generation off
 * every problematic walker per round; rounds of different walkers do not interact, so each
 * walker is unwound on its own.  Negative generations index from the end like numpy does.
 * Returns the number of walkers that were reverted. */
size_t orc_step_back(double Lmin, double *allL, size_t n, size_t ngen, int64_t *generation,
                     double *currentt) {
  int64_t gmax = generation[0];
  for (size_t i = 1; i < n; ++i)
    if (generation[i] > gmax) gmax = generation[i];
  const int64_t width = gmax + 1;
  if (width <= 0) return 0;
  size_t nrev = 0;
  for (size_t i = 0; i < n; ++i) {
    /* below[k] = allL[i,k] < Lmin, frozen at entry (the reference computes it once) */
    uint8_t below[4096];
    int64_t w = width < 4096 ? width : 4096;
    int64_t nbelow = 0;
    for (int64_t k = 0; k < w; ++k) {
      below[k] = allL[i * ngen + (size_t)k] < Lmin;
      nbelow += below[k];
    }
    if (!nbelow) continue;
    ++nrev;
    while (nbelow > 0) {
      const int64_t g = generation[i];
      generation[i] -= 1;
      currentt[i] = (double)NAN;
      const int64_t gi = g < 0 ? g + (int64_t)ngen : g;
      if (gi < 0 || gi >= (int64_t)ngen) break; /* numpy would raise IndexError */
      allL[i * ngen + (size_t)gi] = (double)NAN;
      const int64_t bi = g < 0 ? g + w : g;
      if (bi < 0 || bi >= w) break;
      if (below[bi]) {
        below[bi] = 0;
        --nbelow;
      }
    }
  }
  return nrev;
}

/* popstepsampler.py:26-61: intersection of the line origin + t*direction with the unit cube.
 * nanmax / nanmin skip NaN (0 * inf); a row of only NaN gives NaN. */
void orc_unitcube_line_intersection(const double *origin, const double *direction, size_t n, size_t d,
                                    double *tleft, double *tright) {
  for (size_t i = 0; i < n; ++i) {
    double lo = (double)NAN, hi = (double)NAN;
    for (size_t k = 0; k < d; ++k) {
      const double m = 1.0 / direction[i * d + k];
      const double nn = m * (origin[i * d + k] - 0.5);
      const double kk = fabs(m) * 0.5;
      const double t1 = -nn - kk;
      const double t2 = -nn + kk;
      if (!isnan(t1) && (isnan(lo) || t1 > lo)) lo = t1;
      if (!isnan(t2) && (isnan(hi) || t2 < hi)) hi = t2;
    }
    tleft[i] = lo;
    tright[i] = hi;
  }
}

/* stepfuncs.pyx:537-630.  Sequential by construction: workers l are read in order and several
 * workers may serve the same point.  Returns the number of discarded evaluations. */
int64_t orc_update_vectorised_slice_sampler(const double *t, double *tleft, double *tright,
                                            const double *proposed_L, const double *proposed_u,
                                            const double *proposed_p, int64_t *worker_running,
                                            int64_t *status, double Lthreshold, double shrink_factor,
                                            double *allu, double *allL, double *allp, size_t popsize,
                                            size_t d, size_t nparams) {
  int64_t discarded = 0;
  for (size_t l = 0; l < popsize; ++l) {
    const int64_t w = worker_running[l];
    if (t[l] > tright[w] || t[l] < tleft[w]) {
      if (proposed_L[l] > Lthreshold) ++discarded;
      continue;
    }
    if (0 < t[l] && t[l] < tright[w]) tright[w] = t[l] / shrink_factor;
    if (0 > t[l] && t[l] > tleft[w]) tleft[w] = t[l] / shrink_factor;
    if (proposed_L[l] > Lthreshold && status[w] == 0) {
      status[w] = 1;
      for (size_t k = 0; k < d; ++k) allu[(size_t)w * d + k] = proposed_u[l * d + k];
      allL[w] = proposed_L[l];
      for (size_t k = 0; k < nparams; ++k) allp[(size_t)w * nparams + k] = proposed_p[l * nparams + k];
    }
  }
  size_t j = 0;
  for (;;) {
    int any = 0;
    for (size_t k = 0; k < popsize; ++k)
      if (status[k] == 0) any = 1;
    if (!(j < popsize && any)) break;
    for (size_t k = 0; k < popsize; ++k)
      if (status[k] == 0 && j < popsize) worker_running[j++] = (int64_t)k;
  }
  return discarded;
}

/* popstepsampler.py:64-94 after the layer transform: squared t-space distance per row,
 * sequential over coordinates (numpy sums rows pairwise: compared with a tolerance) */
void orc_row_dist2(const double *a, const double *b, size_t n, size_t d, double *out) {
  for (size_t i = 0; i < n; ++i) {
    double acc = 0.0;
    for (size_t k = 0; k < d; ++k) {
      const double diff = a[i * d + k] - b[i * d + k];
      acc += diff * diff;
    }
    out[i] = acc;
  }
}
