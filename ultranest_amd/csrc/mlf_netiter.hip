// mlf_netiter.hip -- host C++ restatement of the per-iteration integrator bookkeeping of the
// reference: ultranest/netiter.py MultiCounter.passing_node (:721-855) with the
// insertion-order U test (ultranest/ordertest.py:49-104).  The reference does this with ~40 small
// numpy calls per nested-sampling iteration over (nbootstraps + 1) counters (1.4 s of an 8 s run);
// here it is one call into compiled code.  No GPU involved: plain host arithmetic in the
// reference's order; exp / log / log1p come from libm, numpy's own vector routines may differ in the
// last bit (tolerance class, tests/test_netiter.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/mlfriends_hip.h"
#include "mlf_ctx.hpp"

struct mlf_counter {
  size_t nroots = 0, ncounters = 0;
  std::vector<uint8_t> member;   // [ncounters][nroots]: root belongs to bootstrap instance
  std::vector<uint8_t> member_t; // [nroots][ncpad]: the same, one contiguous row per root
  size_t ncpad = 0;
  std::vector<uint32_t> count;
  bool random = false, check_order = false;
  double order_threshold = 4.0;
  std::vector<double> all_H, all_logZ, all_logVol, all_logZremain;
  double logZ, logZerr, logVol, logZremainMax, logZremain, remainder_ratio, remainder_fraction;
  // U-test accumulator (ordertest.py:49-104)
  long long acc_N = 0;
  double acc_U = 0.0;
  std::vector<long long> runs;
  long long niter = 0;
  std::vector<long long> nlive;
  std::vector<double> logleft, logright, sorted;
};

namespace {

const double kInf = std::numeric_limits<double>::infinity();
const double kNaN = std::numeric_limits<double>::quiet_NaN();

// numpy's logaddexp (npy_math): max + log1p(exp(-|x - y|)), x + log 2 for equal arguments
double logaddexp(double x, double y) {
  if (x == y) return x + 0.69314718055994530942;
  const double tmp = x - y;
  if (tmp > 0) return x + std::log1p(std::exp(-tmp));
  if (tmp <= 0) return y + std::log1p(std::exp(tmp));
  return tmp;   // NaN
}

void reset(mlf_counter *c) {
  const size_t n = c->ncounters;
  c->all_H.assign(n, kNaN);
  c->all_logZ.assign(n, -kInf);
  c->all_logVol.assign(n, 0.0);
  c->all_logZremain.assign(n, kInf);
  c->logZ = -kInf;
  c->logZerr = kInf;
  c->logVol = 0.0;
  c->logZremainMax = kInf;
  c->logZremain = kInf;
  c->remainder_ratio = 1.0;
  c->remainder_fraction = 1.0;
  c->acc_N = 0;
  c->acc_U = 0.0;
  c->runs.clear();
  c->niter = 0;
}

}  // namespace

extern "C" {

int mlf_counter_create(mlf_counter **out, size_t nroots, size_t ncounters, const uint8_t *member, int random,
                       int check_insertion_order) {
  if (!out || !member) return mlf::ctx_fail_arg(MLF_E_BADARG, "null pointer");
  *out = nullptr;
  if (nroots == 0 || ncounters == 0) return mlf::ctx_fail_arg(MLF_E_BADARG, "nroots and ncounters must be positive");
  mlf_counter *c = new mlf_counter();
  c->nroots = nroots;
  c->ncounters = ncounters;
  c->member.assign(member, member + nroots * ncounters);
  c->ncpad = (ncounters + 31) / 32 * 32;
  c->member_t.assign(nroots * c->ncpad, 0);
  for (size_t b = 0; b < ncounters; ++b)
    for (size_t r = 0; r < nroots; ++r) c->member_t[r * c->ncpad + b] = member[b * nroots + r] ? 1 : 0;
  c->count.resize(c->ncpad);
  c->random = random != 0;
  c->check_order = check_insertion_order != 0;
  c->nlive.resize(ncounters);
  c->logleft.resize(ncounters);
  c->logright.resize(ncounters);
  reset(c);
  *out = c;
  return 0;
}

int mlf_counter_destroy(mlf_counter *c) {
  delete c;
  return 0;
}

int mlf_counter_reset(mlf_counter *c) {
  if (!c) return mlf::ctx_fail_arg(MLF_E_BADARG, "null pointer");
  reset(c);
  return 0;
}

int mlf_counter_passing_node(mlf_counter *c, int64_t rootid, double Li, size_t nchildren, const double *child_values,
                             const int64_t *rootids, const double *parallel_values, size_t nparallel,
                             const double *random_beta, double *logwidth) {
  if (!c || !logwidth || (nparallel && (!rootids || !parallel_values)) || (nchildren && !child_values))
    return mlf::ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (rootid < 0 || (size_t)rootid >= c->nroots) return mlf::ctx_fail_arg(MLF_E_BADARG, "rootid out of range");
  if (nparallel == 0) return mlf::ctx_fail_arg(MLF_E_BADARG, "no parallel nodes");
  if (c->random && nchildren >= 1 && !random_beta)
    return mlf::ctx_fail_arg(MLF_E_BADARG, "random volume shrinkage needs the beta draws");
  for (size_t j = 0; j < nparallel; ++j)
    if (rootids[j] < 0 || (size_t)rootids[j] >= c->nroots) return mlf::ctx_fail_arg(MLF_E_BADARG, "root id out of range");
  const size_t nb = c->ncounters, nr = c->nroots;
  // in which bootstraps is rootid, and how many live points does each bootstrap have (:743-747)
  {   // one contiguous membership row per parallel node: the inner loop vectorises
    const size_t np_ = c->ncpad;
    uint32_t *cnt = c->count.data();
    for (size_t b = 0; b < np_; ++b) cnt[b] = 0;
    for (size_t j = 0; j < nparallel; ++j) {
      const uint8_t *m = c->member_t.data() + (size_t)rootids[j] * np_;
      for (size_t b = 0; b < np_; ++b) cnt[b] += m[b];
    }
    for (size_t b = 0; b < nb; ++b) c->nlive[b] = cnt[b];
  }
  const long long nlive0 = c->nlive[0];
  auto active = [&](size_t b) { return c->member[b * nr + (size_t)rootid] != 0; };
  if (!active(0)) return mlf::ctx_fail_arg(MLF_E_STATE, "the main counter must contain every root");

  if (nchildren >= 1) {   // one arc terminates, another is spawned (:749-820)
    for (size_t b = 0; b < nb; ++b) {
      if (c->random) {
        c->logleft[b] = std::log(random_beta[b]);
        c->logright[b] = std::log1p(-random_beta[b]);
      } else {
        c->logleft[b] = std::log1p(-std::exp(-1.0 / (double)c->nlive[b]));
        c->logright[b] = -1.0 / (double)c->nlive[b];
      }
    }
    if (c->random) {
      c->logleft[0] = std::log1p(-std::exp(-1.0 / (double)nlive0));
      c->logright[0] = -1.0 / (double)nlive0;
    }
    for (size_t b = 0; b < nb; ++b) {
      if (!active(b)) {
        logwidth[b] = -kInf;
        continue;
      }
      const double lw = c->logleft[b] + c->all_logVol[b];
      logwidth[b] = lw;
      const double wi = lw + Li;
      const double logZ = c->all_logZ[b];
      const double logZnew = logaddexp(logZ, wi);
      const double H = std::exp(wi - logZnew) * Li + std::exp(logZ - logZnew) * (c->all_H[b] + logZ) - logZnew;
      const bool first_setting = std::isnan(H);
      c->all_logZ[b] = first_setting ? wi : logZnew;
      c->all_H[b] = first_setting ? -lw : H;
    }
    c->logZ = c->all_logZ[0];
    if (c->all_H[0] > 0) c->logZerr = std::sqrt(c->all_H[0] / (double)nlive0);
    for (size_t b = 0; b < nb; ++b)
      if (active(b)) c->all_logVol[b] += c->logright[b];
    c->logVol = c->all_logVol[0];

    if (c->check_order) {   // insertion-order U test, only while all parallel values are distinct (:801-812)
      c->sorted.assign(parallel_values, parallel_values + nparallel);
      std::sort(c->sorted.begin(), c->sorted.end());
      const bool distinct = std::adjacent_find(c->sorted.begin(), c->sorted.end()) == c->sorted.end();
      if (distinct) {
        for (size_t k = 0; k < nchildren; ++k) {
          // rank of the child among the parallel values of the main counter
          const long long order = std::lower_bound(c->sorted.begin(), c->sorted.end(), child_values[k]) - c->sorted.begin();
          if (order < 0 || order > nlive0) return mlf::ctx_fail_arg(MLF_E_BADARG, "insertion order out of range");
          c->acc_U += ((double)order + 0.5) / (double)nlive0;
          c->acc_N += 1;
          const double m_U = (double)c->acc_N * 0.5;
          const double sigma = std::sqrt((double)c->acc_N / 12.0);
          const double z = (c->acc_U - m_U) / sigma;
          if (std::fabs(z) > c->order_threshold) {
            c->runs.push_back(c->acc_N);
            c->acc_N = 0;
            c->acc_U = 0.0;
          }
        }
      }
    }
  } else {   // contracting: weight is volume / nlive (:822-842)
    for (size_t b = 0; b < nb; ++b) {
      if (!active(b)) {
        logwidth[b] = -kInf;
        continue;
      }
      const double lw = c->all_logVol[b] - std::log((double)c->nlive[b]);
      logwidth[b] = lw;
      c->all_logZ[b] = logaddexp(c->all_logZ[b], lw + Li);
    }
    c->logZ = c->all_logZ[0];
    for (size_t b = 0; b < nb; ++b)
      if (active(b)) c->all_logVol[b] += std::log1p(-1.0 / (double)c->nlive[b]);
    c->logVol = c->all_logVol[0];
  }

  // weight of the unexplored remainder (:844-855)
  double Lmax = parallel_values[0];
  for (size_t j = 1; j < nparallel; ++j) Lmax = parallel_values[j] > Lmax ? parallel_values[j] : Lmax;
  double s = 0.0;
  for (size_t j = 0; j < nparallel; ++j) s += std::exp(parallel_values[j] - Lmax);
  const double tail = std::log(s) + Lmax;
  const double lognlive0 = std::log((double)nlive0);
  double zmax = -kInf;
  for (size_t b = 0; b < nb; ++b) {
    const double v = (c->all_logVol[b] - lognlive0) + tail;
    c->all_logZremain[b] = v;
    zmax = (v > zmax || std::isnan(v)) ? v : zmax;
  }
  c->logZremainMax = zmax;
  c->logZremain = c->all_logZremain[0];
  c->remainder_ratio = std::exp(c->logZremain - c->logZ);
  c->remainder_fraction = 1.0 / (1.0 + std::exp(c->logZ - c->logZremain));
  c->niter += 1;
  return 0;
}

// Host helper of the lazy device mirror (regions._DeviceState): indices of the rows in which two (n, d) float64
// matrices differ bytewise -- the driver replaces one live point per iteration in place, and finding that
// row with numpy costs ~0.4 ms at 4000 x 50.  count = number of differing rows (may exceed capacity).
int mlf_host_changed_rows(const double *a, const double *b, size_t n, size_t d, int64_t *rows, size_t capacity,
                          size_t *count) {
  if (!a || !b || !count || (capacity && !rows)) return mlf::ctx_fail_arg(MLF_E_BADARG, "null pointer");
  size_t c = 0;
  const size_t bytes = d * sizeof(double);
  for (size_t i = 0; i < n; ++i)
    if (memcmp(a + i * d, b + i * d, bytes) != 0) {
      if (c < capacity) rows[c] = (int64_t)i;
      ++c;
    }
  *count = c;
  return 0;
}

// Host helper of the bootstrap (regions._draw_selection): the selection masks of `nrounds` rounds, drawn from
// numpy's legacy MT19937 stream exactly as `nrounds` calls of np.random.randint(npoints, size=npoints) draw them
// (reference mlfriends.pyx:1044-1047): 32-bit words, masked with the smallest 2^k - 1 >= npoints - 1, values above
// npoints - 1 rejected.  `key`/`pos` are the generator state (np.random.get_state()[1:3]) and are advanced in place.
// 120000 draws cost numpy 1.2 ms of a 6 ms region rebuild.
namespace {
inline void mt19937_refill(uint32_t *key) {
  constexpr int kN = 624, kM = 397;
  auto twist = [](uint32_t hi, uint32_t lo) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  };
  int i = 0;
  for (; i < kN - kM; ++i) key[i] = key[i + kM] ^ twist(key[i], key[i + 1]);
  for (; i < kN - 1; ++i) key[i] = key[i + (kM - kN)] ^ twist(key[i], key[i + 1]);
  key[kN - 1] = key[kM - 1] ^ twist(key[kN - 1], key[0]);
}
}  // namespace

int mlf_host_draw_selection(uint32_t *key, int32_t *pos, size_t npoints, size_t nrounds, uint8_t *masks) {
  if (!key || !pos || (npoints && nrounds && !masks)) return mlf::ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (npoints == 0 || npoints > 0x80000000ull || *pos < 0 || *pos > 624)
    return mlf::ctx_fail_arg(MLF_E_BADARG, "npoints or generator position out of range");
  memset(masks, 0, npoints * nrounds);
  if (npoints == 1) {   // numpy fills the zeros without drawing
    memset(masks, 1, nrounds);
    return 0;
  }
  const uint32_t top = (uint32_t)(npoints - 1);
  uint32_t mask = top;
  mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
  int p = *pos;
  for (size_t b = 0; b < nrounds; ++b) {
    uint8_t *row = masks + b * npoints;
    for (size_t i = 0; i < npoints; ++i) {
      uint32_t v;
      do {
        if (p == 624) mt19937_refill(key), p = 0;
        uint32_t y = key[p++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        v = y & mask;
      } while (v > top);
      row[v] = 1;
    }
  }
  *pos = p;
  return 0;
}

int mlf_counter_state(const mlf_counter *c, double *scalars, double *all_H, double *all_logZ, double *all_logVolremaining,
                      double *all_logZremain, int64_t *runs, size_t runs_capacity, size_t *nruns) {
  if (!c || !scalars) return mlf::ctx_fail_arg(MLF_E_BADARG, "null pointer");
  scalars[0] = c->logZ;
  scalars[1] = c->logZerr;
  scalars[2] = c->logVol;
  scalars[3] = c->logZremainMax;
  scalars[4] = c->logZremain;
  scalars[5] = c->remainder_ratio;
  scalars[6] = c->remainder_fraction;
  scalars[7] = (double)c->acc_N;
  scalars[8] = c->acc_U;
  scalars[9] = (double)c->niter;
  const size_t n = c->ncounters * sizeof(double);
  if (all_H) memcpy(all_H, c->all_H.data(), n);
  if (all_logZ) memcpy(all_logZ, c->all_logZ.data(), n);
  if (all_logVolremaining) memcpy(all_logVolremaining, c->all_logVol.data(), n);
  if (all_logZremain) memcpy(all_logZremain, c->all_logZremain.data(), n);
  if (nruns) *nruns = c->runs.size();
  if (runs)
    for (size_t i = 0; i < c->runs.size() && i < runs_capacity; ++i) runs[i] = c->runs[i];
  return 0;
}

}  // extern "C"
