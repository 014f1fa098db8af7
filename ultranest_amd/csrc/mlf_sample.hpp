// mlf_sample.hpp -- device-side proposal generation + compaction (mlf_sample.hip)
#pragma once
#include "mlf_common.hpp"

namespace mlf {

void launch_philox_words(unsigned long long seed, unsigned stream, long long n, unsigned *out, hipStream_t s);
void launch_generate_cube(double *pts, long long nelem, unsigned long long seed, unsigned long long offset,
                          hipStream_t s);
void launch_generate_ball(double *z, long long n, int d, double enlarge, unsigned long long seed,
                          unsigned long long offset, hipStream_t s);
// sample_from_wrapping_ellipsoid in one launch: w = centre + (ball draw) . A, in_cube[p] = all coordinates strictly inside (0, 1).
// A_padded: the axes matrix (element (j, k) = axes_T[j][k]) as [d][4 generate_ellipsoid_chunk(d)] doubles, zero padded.  d <= 128.
int generate_ellipsoid_chunk(int d);
hipError_t launch_generate_ellipsoid(double *w, long long n, int d, double enlarge, const double *A_padded, const double *center,
                                     uint8_t *in_cube, unsigned long long seed, unsigned long long offset, hipStream_t s);
// AffineLayer.untransform of whole batches: w = t . M + ctr (M padded like A_padded above), circular axes rotated back where
// wrap_shift (NaN = not circular) is given, in_cube[p] = all coordinates strictly inside (0, 1).  d <= 128.
hipError_t launch_rows_affine(const double *t, long long n, int d, const double *M_padded, const double *ctr, const double *wrap_shift,
                              double *w, uint8_t *in_cube, hipStream_t s);
void launch_center_and_cube(double *w, long long n, int d, const double *center, uint8_t *in_cube, hipStream_t s);
void launch_generate_tbox(double *t, long long n, int d, const double *lo, const double *hi, double pad,
                          unsigned long long seed, unsigned long long offset, hipStream_t s);
void launch_generate_around_points(double *t, double *thin_u, long long n, int d, const double *refR, int nlive, int dp,
                                   double r2, unsigned long long seed, unsigned long long offset, hipStream_t s);
void launch_thin_by_multiplicity(const long long *count, const double *thin_u, long long n, uint8_t *mask, hipStream_t s);
void launch_untransform_rows(const double *t, long long n, int d, const double *invT, const double *ctr,
                             const double *wrap_shift, double *w, uint8_t *in_cube, hipStream_t s);
void launch_elementwise_affine(const double *x, long long n, int tkind, double a, double b, double *out, hipStream_t s);
// mask[e] = v[e] > threshold (&& also[e] where `also` is given)
void launch_mask_greater(const double *v, long long n, double threshold, uint8_t *mask, hipStream_t s, const uint8_t *also = nullptr);
void launch_mask_and(uint8_t *mask, const uint8_t *other, long long n, hipStream_t s);
void launch_apply_pregate(const uint8_t *pregate, long long n, uint8_t *gate, uint8_t *route, float *tlo,
                          float *thi, hipStream_t s);
void launch_scan_counts(unsigned *blk, int nblk, hipStream_t s);
void launch_mask_offsets(const uint8_t *mask, long long n, unsigned *blk, hipStream_t s);
// blk: (ceil(n/256) + 1) counters; after the call blk[ceil(n/256)] holds the number of accepted rows
void launch_compact(const double *pts, const uint8_t *mask, long long n, int d, unsigned *blk, double *out,
                    unsigned capacity, hipStream_t s);
// the scatter of launch_compact alone, on offsets that launch_mask_offsets computed for the same mask
void launch_scatter(const double *pts, const uint8_t *mask, long long n, int d, const unsigned *blk, double *out,
                    unsigned capacity, hipStream_t s);

}  // namespace mlf
