// mlf_misc.hpp -- launchers of the layout / reduction / likelihood kernels (mlf_misc.hip)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

void launch_build_layouts(const double *src, int n, int d, int dp, int npad, double *refT,
                          double *refR, hipStream_t s);
void launch_update_rows(const double *rows, int count, int d, int dp, int npad, const long long *index, double *refT,
                        double *refR, hipStream_t s);
// up to kScatterMax (destination, source, bytes) copies in ONE launch: the small constants of mlf_region_set, staged in a
// pinned arena the device reads directly (ten separate copies on the stream cost ~4 us each, and as much on the host)
constexpr int kScatterMax = 24;
struct ScatterArgs {
  void *dst[kScatterMax];
  const void *src[kScatterMax];
  unsigned bytes[kScatterMax];
};
void launch_scatter_copy(const ScatterArgs &a, int count, hipStream_t s);
void launch_fill_u64(unsigned long long *p, long long n, unsigned long long v, hipStream_t s);
void launch_pack_selection(const uint8_t *selected, int n, int npad, int b0, int nb, unsigned *sel,
                           hipStream_t s, unsigned *selmask = nullptr);
void launch_boot_final(const unsigned long long *M, const unsigned *sel, int n, int npad, int nb,
                       double *maxd, uint8_t *skipped, hipStream_t s, int row_lo = 0, int row_hi = -1);
void launch_subtract_accum(const double *pts, int n, int d, const unsigned long long *flags,
                           int ntiles, double *out, hipStream_t s);
void launch_pair_dist2_lower(const double *pts, int n, int d, double *out, hipStream_t s);
constexpr int kExtentScratchBlocks = 64;
void launch_col_extent(const double *pts, int n, int d, double *part, double *out, hipStream_t s);
void launch_scaling_transform(const double *pts, long long np, int d, const double *mean,
                              const double *std, const double *wrap_shift, const uint8_t *gate,
                              double *out, long long ldt, hipStream_t s);
void launch_masked_max(const double *q, const uint8_t *selected, int n, double *out, hipStream_t s);
// f_b = max over left-out rows of the quadratic form with (scale cov_b)^-1, by Cholesky on the device (d <= 64)
size_t boot_cholmax_scratch_bytes(int d, int B);
hipError_t launch_boot_cholmax(const double *u, int n, int d, const uint8_t *selected, int B, const double *mean,
                               const double *cov, double scale, unsigned long long *out_bits, void *scratch,
                               hipStream_t s);
void launch_boot_moments(const double *u, int n, int d, const uint8_t *selected, int B, double *mean,
                         int *count, double *cov, int *idx_scratch, hipStream_t s);
void launch_loglike(int kind, const double *params, int d, long long n, const double *aux,
                    double sigma, double *like, hipStream_t s);

void launch_mark_gated(const uint8_t *gate, long long n, long long *idx, hipStream_t s);
// returns the number of FP64 lane-operations the probe executes
double launch_fp64_probe(double *sink, int blocks, int iters, hipStream_t s);

}  // namespace mlf
