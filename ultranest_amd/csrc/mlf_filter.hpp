// mlf_filter.hpp -- MFMA pre-filter for the neighbour scan (see mlf_filter.hip)
#pragma once
#include "mlf_common.hpp"

#define MLF_FILTER_MAXD 128

namespace mlf {

struct FilterArgs {
  const void *refF;      // f16 fragments of the live points  [ntiles32][KS][64][8]
  const void *qF;        // f16 fragments of the queries      [ngroups][KS][64][8]
  const float *tlo, *thi;
  int ntiles32;
  long long ngroups;     // query groups of 32
  long long nq;
  int *best;             // per query: first certain/confirmed live index, kNone if none
  unsigned long long *list;   // [nwaves][seg_cap] wave-private segments of uncertain pairs
  unsigned seg_cap;
  unsigned *seg_count;        // [nwaves]
  unsigned *counters;         // [1] overflow flag
};

struct RecheckArgs {
  const unsigned long long *list;
  unsigned seg_cap;
  const unsigned *seg_count;
  const double *refR;    // [npad][dp]
  int n, d, dp;
  const double *q;
  long long ldq, ldk, nq;   // query element (j, k) at q[j*ldq + k*ldk]
  double r2;
  int *best;
};

void launch_ref_stats(const double *refR, int n, int d, int dp, double *stats, hipStream_t s);
void launch_quant_refs(const double *refR, int n, int npad32, int d, int dp, int ks,
                       const double *stats, void *refF, hipStream_t s);
void launch_quant_queries(const double *q, long long ldq, long long nq, long long nqpad, int d_src, int d, int ks,
                          const double *stats, double r2, const uint8_t *gate, void *qF, float *tlo,
                          float *thi, uint8_t *route, int *best, unsigned *counters, hipStream_t s);
hipError_t launch_filter(int ks, const FilterArgs &a, bool first, hipStream_t s);
void launch_recheck(const RecheckArgs &a, long long nwaves, hipStream_t s);
long long filter_wave_count(int ks, long long ngroups);  // waves (= list segments) of a k_filter launch
void launch_filter_finalize(const uint8_t *route, const int *best, const unsigned *counters, long long nq,
                            uint8_t *out_mask, long long *out_idx, hipStream_t s);
void launch_route_gate(const uint8_t *route, const unsigned *counters, long long nq, int which,
                       uint8_t *gate, hipStream_t s);

}  // namespace mlf
