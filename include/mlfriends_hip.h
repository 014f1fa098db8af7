/*
 * mlfriends_hip.h -- C ABI of libmlfriends_hip.so: the MI355X (gfx950) implementation of the
 * UltraNest MLFriends hot path.  This is the drop-in boundary: plain pointers and sizes, no
 * torch / numpy types.  The Python host layer (ultranest_amd/_lib.py) binds exactly these
 * symbols with ctypes, in the convention of the reference's own compiled-callback examples
 * (reference languages/c/mylib.c:33 + languages/c/runc.py:8-28: C-contiguous float64 (n, d)
 * arrays, sizes as size_t, caller-allocated outputs).
 *
 * Every entry point names the reference routine it replaces (paths relative to the UltraNest
 * 4.5.0 tree).  All arrays are row-major float64 unless stated; index outputs are int64
 * (reference mlfriends.pyx:25-26), masks are one byte per element (numpy bool).
 *
 * Return value: 0 = ok; < 0 = -(hipError_t) (message via mlf_last_error()); > 0 = domain
 * condition (MLF_E_*).  Host-pointer entry points are synchronous: inputs are copied to the
 * device, outputs are complete on return, no host pointer is retained.  *_dev entry points
 * take DEVICE pointers plus a hipStream_t (passed as void*) and only enqueue work.
 */
#ifndef MLFRIENDS_HIP_H
#define MLFRIENDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLF_ABI_VERSION 1

#define MLF_E_BADARG 1      /* null pointer, zero dimension, inconsistent sizes            */
#define MLF_E_DIM 2         /* dimensionality above MLF_MAX_DIM                            */
#define MLF_E_NODEVICE 3    /* no usable gfx950 device                                      */
#define MLF_E_STATE 4       /* region handle used before mlf_region_set                     */

/* Largest dimensionality of the K1-K5 / H3 / T1 / R3 entry points.  Up to 128 the templated kernels; 129 ... 1024 the
 * run-time-dimensionality kernels (csrc/mlf_wide.hip: exact binary64 scan only, no matrix-core pre-filter).  Entry points
 * that stop earlier and say so with MLF_E_DIM: mlf_walkers_create (d <= 128: one wave per walker, lane = coordinate pair),
 * mlf_col_extent (d <= 128), mlf_bootstrap_factor (d <= 64; above: mlf_bootstrap_moments + mlf_bootstrap_quadmax);
 * the single-launch small-batch path of mlf_region_inside (up to 256 host proposals) is used for d <= 128 and the batched
 * pipeline above. */
#define MLF_MAX_DIM 1024

/* ---- library / device ---------------------------------------------------------------- */
int mlf_abi_version(void);
const char *mlf_last_error(void);
int mlf_device_count(int *count);
int mlf_set_device(int device);            /* device used by this process (default 0)      */
int mlf_device_name(char *buf, size_t buflen);
int mlf_synchronize(void);

/* ---- device memory for callers that keep arrays resident between calls.  The reference has no counterpart (its arrays
 * are numpy arrays); ultranest_amd.device_rebuild keeps the live points of a region rebuild (integrator.py:2055-2122) in
 * HBM across the calls below instead of re-uploading them per call.  Every array argument of the stateless entry points
 * documented as "host or device" may be such a pointer.  mlf_dev_copy: either side host or device, ordered on the
 * library's stream, sync != 0 waits for completion. */
int mlf_dev_alloc(size_t bytes, void **out);
int mlf_dev_free(void *p);
int mlf_dev_copy(void *dst, const void *src, size_t bytes, int sync);

/* Column extents of an (n, d) row-major array (host or device): lo[c] = min, hi[c] = max over the rows (host outputs).
 * The bounding box of the whitened live points (mlfriends.pyx:969-970: unormed.min(axis=0) / .max(axis=0)) and the
 * driver's cube test (integrator.py:2059-2061) on the resident copy.  min / max are exact: same bits as numpy's. */
int mlf_col_extent(const double *pts, size_t n, size_t d, double *lo, double *hi);
/* Tuning options.  Every option has a PROCESS default (mlf_set_option) and can be overridden per region handle
 * (mlf_region_set_option; inherit != 0 returns one option -- or, with name == NULL, all of them -- to the process
 * default), so that two regions of one process can run different routings.  Results never depend on them.
 *   "filter"                   1/0: matrix-core pre-filter in front of the exact neighbour scan
 *   "filter_min_queries"       batches below this size use the exact scan only
 *   "filter_phases"            0: one sweep; 1 (default): two live-point ranges, decided proposals dropped in between
 *                              (the compaction rides in the first range's epilogue); n >= 2: n ranges
 *   "filter_phase_min_queries" smaller batches sweep all tiles in one launch
 *   "filter_first_range_pct"   10 ... 90, default 30 (50 until the operand was ordered, "filter_order"): share of the live-point tiles the
 *                              first of two ranges takes (with three ranges: its share of the tiles in front of the last range)
 *   "filter_second_range_pct"  default 50; 0 ... 90.  Batches of at least "filter_third_range_min_work" run the min-only sweep in THREE
 *                              ranges: the last starts at this share of the tiles, and the tiles before it are split at
 *                              "filter_first_range_pct" of them (30 % of 50 %: ranges of 15 %, 35 % and 50 %); the proposals
 *                              without a certain hit are compacted after each.  0: always two ranges.  Ignored where a range
 *                              would have fewer than 4 tiles
 *   "filter_third_range_min_work"  default 100000000: (proposals x 32-row live-point tiles) from which three ranges are used (at
 *                              N = 4000: 800 000 proposals); below, the third launch and its compaction cost more than the pairs it saves
 *   "filter_order"             1 (default): the mask-mode kernels (MLFriends.inside: any hit decides, mlfriends.pyx:1186-1211) sweep a
 *                              copy of the live points ordered nearest-to-the-centre first, so that the first of two tile
 *                              ranges decides more proposals; 0: storage order.  find_nearby's first-index operand always
 *                              keeps storage order (mlfriends.pyx:176-183).  Set per handle it takes effect with the next batch
 *   "filter_narrow_tail"       0: every range with 4 query groups per wave; 1 (default): later ranges with 2
 *   "filter_split_waves"       waves a single-sweep launch over a small batch aims at when it splits the tiles (1-16 ranges)
 *   "fused_prep"               1/0: fused per-proposal stage, or k_prep followed by a separate quantisation
 *   "prep_bounded"             1/0: the bounded matrix-core per-proposal stage (split binary16) or the binary64 one
 *   "fused_first_range"        1 (default): on the min-only path the per-proposal stage runs inside the first sweep launch (k_prep_sweep:
 *                              the binary16 operand never leaves the registers, 256 MB less HBM traffic per 10^6 x 50 batch);
 *                              0: k_prep4, then k_sweep_min (the default until the first range became short: with a first range
 *                              of 30 % and more of the tiles the separate launches were 3 % faster)
 *   "fused_waves"              4 (default) / 8: waves per workgroup of k_prep_sweep.  4: 79 KiB of LDS per workgroup at d = 50, two per CU --
 *                              the CU's eight waves start and end in two independent halves (0.177 against 0.181 ms at C5); used up
 *                              to d = 50 (above, two such workgroups do not fit a CU).  8: one workgroup of 8 waves per CU
 *   "fused_variant"            default 3: bit 0 = k_prep_sweep fetches its matrix fragments by LDS-DMA (every request of a wave
 *                              under way at once); 0 = by a load / store loop (round 5's form), kept for A/B runs inside ONE process
 *                              (scripts/fused_ab.py: boxes of the pool differ by 8 % in this kernel).  bit 1 = where the wrapping
 *                              ellipsoid's matrix equals T T^T of the layer up to a residue measured at mlf_region_set (|E|_F <= 2^-34 |A|_F)
 *                              and both share their centre bit for bit (AffineLayer with one cluster), the ellipsoid test reads
 *                              delta^T A delta = |T^T delta|^2 + delta^T E delta off the whitening chain: 24 instead of 42 matrix
 *                              instructions per 32 proposals at d = 50, no L^T fragments; band proposals take the exact path as before
 *   "mid_max_queries"          default 2048: batches up to this size (and at least "filter_min_queries") run the per-proposal
 *                              stage, the pre-filter sweep, the re-check and the answers in ONE launch (k_inside_mid); 0 = never
 *   "sweep_min"                1 (default): phased batches through the min-only sweep (k_sweep_min: running minima only
 *                              in the long launches, the proposals whose minimum ends in the band handled by
 *                              k_uncertain); 0: k_sweep (per-tile band test) followed by the re-check launch
 *   "boot_symmetric"           1 (default): a whole-range mlf_maxradiussq_bootstrap pass over enough live points (>= 1024 tiles
 *                              of 64 x 64 pairs, i.e. n > 2816; d <= 64) computes every pair distance once and uses it for both
 *                              points of the pair (k_boot_sym); 0: k_boot (both orders); 2: k_boot_sym whatever the number of
 *                              tiles (tests).  Same minima, bit for bit
 *   "small_path"               1/0: mlf_region_inside with up to 256 proposals as ONE launch over pinned staging -- the calls
 *                              of the scalar step samplers -- or through the batched pipeline
 *   "time_filter_launches"     1/0: event pairs around every matrix-kernel launch (mlf_region_timing_filter_launch_ms)
 * mlf_option_name enumerates the names (index 0, 1, ... until it fails); mlf_get_option returns the process default in force
 * (so that a caller that reports a routing -- bench.py -- reads it instead of assuming it). */
int mlf_set_option(const char *name, long long value);
int mlf_get_option(const char *name, long long *value);
/* test hook: forget which (kernel, device) pairs hold their dynamic-LDS grant (hipFuncSetAttribute belongs to the pair), so
 * that the next launch of every kernel grants again -- what a second device would need; *grants_so_far (optional) receives
 * the number of grants issued by the process up to now. */
int mlf_debug_forget_grants(unsigned long long *grants_so_far);
int mlf_option_name(int index, char *buf, size_t buflen);

/* ---- K1: find_nearby -- ultranest/mlfriends.pyx:143-183 -----------------------------------
 * out[j] = lowest i with sum_k (apts[i,k]-bpts[j,k])^2 <= radiussq (k ascending, no FMA),
 * else -1. */
int mlf_find_nearby(const double *apts, size_t na, const double *bpts, size_t nb, size_t d,
                    double radiussq, int64_t *out);

/* ---- K2: count_nearby -- ultranest/mlfriends.pyx:31-68 (cdef; used by sample_from_points :1088) */
int mlf_count_nearby(const double *apts, size_t na, const double *bpts, size_t nb, size_t d,
                     double radiussq, int64_t *out);

/* ---- K3: subtract_nearby -- ultranest/mlfriends.pyx:73-138 ----------------------------------
 * out[j,:] = pts[j,:] - mean of all pts[i,:] with dist2(i,j) <= radiussq (i ascending sum).
 * Array arguments of the stateless entry points (here: pts, out; mlf_find_nearby: apts, bpts, out; the point arrays of
 * the bootstrap calls, mlf_affine_transform, the live points of mlf_region_set) may be HOST or DEVICE pointers: a
 * device-resident caller (ultranest_amd.device_rebuild) keeps them in HBM between calls. */
int mlf_subtract_nearby(const double *pts, size_t n, size_t d, double radiussq, double *out);

/* ---- H1: the labels of update_clusters -- ultranest/mlfriends.pyx:275-343 ----------------------
 * Friends-of-friends clusters of tpts with linking length sqrt(radiussq): labels[i] in 1 ... (start at 1; `previous`,
 * optional, seeds cluster c at the first point that carried label c before, exactly as the reference re-uses old ids
 * :289-291, :317-320), *nclusters = number of distinct labels.  One all-pairs pass on the device (hit ballots), the
 * reference's growth rounds replayed on the bit rows on the host.  tpts: host or device; previous / labels: host. */
int mlf_cluster_labels(const double *tpts, size_t n, size_t d, double radiussq, const int64_t *previous,
                       int64_t *labels, int64_t *nclusters);
/* the two halves of mlf_cluster_labels: the adjacency bits (n rows of ceil(n / 64) 64-bit words; bit i of row j: points
 * i and j are within the radius; *adj_out points into a pinned buffer of the library, valid until the next call of
 * either function) and the host-only replay of the growth rounds, which touches no device and no library state -- a
 * caller may run it on another thread next to its device calls (ultranest_amd.device_rebuild does). */
int mlf_adjacency_bits(const double *pts, size_t n, size_t d, double radiussq, const unsigned long long **adj_out);
int mlf_host_cluster_replay(const unsigned long long *adj, size_t n, const int64_t *previous, int64_t *labels,
                            int64_t *nclusters);

/* ---- K4: compute_maxradiussq over bootstrap rounds -- mlfriends.pyx:188-224 called from
 * MLFriends.compute_enlargement :1044-1054 and MLFriends.compute_maxradiussq :1004-1012.
 * selected is (B, n) bytes (non-zero = selected); here and in mlf_bootstrap_factor it may be a HOST or a DEVICE
 * pointer (the multi-GPU rebuild broadcasts the masks between devices).  maxd_out[b] = max over unselected j of the
 * min over selected i of dist2(i,j), narrowed to binary32 and widened back exactly as the
 * reference's `cdef float` return does.  skipped_out[b] = 1 when bootstrap b selects all or
 * no points (reference :1048 `continue`); maxd_out[b] = 0 then.  skipped_out may be NULL. */
int mlf_maxradiussq_bootstrap(const double *pts, size_t n, size_t d, const uint8_t *selected,
                              size_t B, double *maxd_out, uint8_t *skipped_out);
/* The same over the rows [row_lo, row_hi) as left-out points only (every live point i, all B rounds): one rank's share
 * when the bootstrap of integrator.py:375-415 is sharded over W GPUs by ROW BLOCKS -- each rank computes 1/W of the pair
 * distances, and the max over ranks of maxd_out[b] (an all-reduce MAX; narrowing to binary32 is monotone) is bit for bit
 * what mlf_maxradiussq_bootstrap returns.  skipped_out[b] is the property of the round (all n rows), identical on all ranks. */
int mlf_maxradiussq_bootstrap_rows(const double *pts, size_t n, size_t d, const uint8_t *selected, size_t B,
                                   size_t row_lo, size_t row_hi, double *maxd_out, uint8_t *skipped_out);

/* ---- K5: pair distances for compute_mean_pair_distance -- mlfriends.pyx:229-270 -------------
 * dist2_out is the packed strict lower triangle: entry j*(j-1)/2 + i for i < j.  The host
 * layer applies sqrt, the cluster mask and the reference's (j outer, i inner) running sum. */
int mlf_pair_dist2_lower(const double *pts, size_t n, size_t d, double *dist2_out);

/* ---- H3: _inside_ellipsoid -- mlfriends.pyx:882-912 -----------------------------------------
 * mask[p] = einsum('ij,jk,ik->i', pts-ctr, invcov, pts-ctr)[p] <= sqradius, evaluated in
 * numpy's c_einsum order (one accumulator, j outer / k inner, (d_j*A_jk)*d_k, no FMA).
 * q_out (may be NULL) receives the quadratic form. */
int mlf_inside_ellipsoid(const double *pts, size_t np, size_t d, const double *ctr,
                         const double *invcov, double sqradius, uint8_t *mask, double *q_out);

/* ---- T1: AffineLayer.transform -- mlfriends.pyx:737-743 (+ wrap :529-536) -------------------
 * out = (wrap(pts) - ctr) . T, k-ascending FMA chain.  wrap_shift is NULL or d doubles:
 * for a wrapped dimension the value (1 - cut), NaN for an unwrapped one. */
int mlf_affine_transform(const double *pts, size_t np, size_t d, const double *ctr,
                         const double *T, const double *wrap_shift, double *out);

/* ---- bootstrap ellipsoid statistics -- mlfriends.pyx:1056-1066, :1426-1436, :1586-1594 ------
 * For each bootstrap b: mean (d) and sample covariance (d,d; ddof=1, NOT yet scaled by d+2) of
 * the selected rows of u.  Two-pass (centred) accumulation. */
int mlf_bootstrap_moments(const double *u, size_t n, size_t d, const uint8_t *selected, size_t B,
                          double *mean_out /* B*d */, double *cov_out /* B*d*d */);
/* Both steps above and the matrix inversion between them in one call, without leaving the device (d <= 64, the
 * reference's minvol = 0 case): f_out[b] = max over the rows NOT selected in round b of
 * (u_i - m_b)^T (scale cov_b)^-1 (u_i - m_b), with m_b / cov_b the mean and sample covariance of the selected rows
 * (reference mlfriends.pyx:1056-1066 passes scale = d + 2).  The form is evaluated through a Cholesky factor, so the
 * values agree with inv + einsum to rounding (tolerance class).  A round whose matrix is not positive definite or
 * not finite gives NaN (the reference raises LinAlgError there). */
int mlf_bootstrap_factor(const double *u, size_t n, size_t d, const uint8_t *selected, size_t B, double scale,
                         double *f_out);

/* f_out[b] = max over UNselected rows of (u-ctr_b)^T invcov_b (u-ctr_b). */
int mlf_bootstrap_quadform_max(const double *u, size_t n, size_t d, const uint8_t *selected,
                               size_t B, const double *ctr /* B*d */,
                               const double *invcov /* B*d*d */, double *f_out);

/* ---- device-resident region: MLFriends.inside -- mlfriends.pyx:1186-1211 --------------------
 * Holds the whitened live points (both layouts), the layer (ctr, T, wraps), the wrapping
 * ellipsoid and the two thresholds.  mlf_region_inside = H3 -> T1 -> K1 on the device with the
 * neighbour scan gated on the ellipsoid result (reference :1202-1209). */
typedef struct mlf_region mlf_region;
int mlf_region_create(mlf_region **out);
int mlf_region_destroy(mlf_region *r);
/* layer_kind: 0 = AffineLayer family (ctr[d], T[d,d]); 1 = ScalingLayer (ctr = mean[d],
 * T = std[d]).  use_scan = 0 gives RobustEllipsoidRegion.inside (:1374-1390, H3 only).
 * live_space: 0 = `live` rows are already whitened (region.unormed); 1 = `live` rows are
 * cube-space points (region.u) and are whitened on the device with the SAME kernel that whitens
 * proposals, so that a live point tested against its own region is at distance exactly 0
 * (the `d <= r2` pin of reference tests/test_regionsampling.py:46-48). */
int mlf_region_set(mlf_region *r, const double *live, size_t n, size_t d, int live_space,
                   int layer_kind, const double *layer_ctr, const double *layer_T,
                   const double *wrap_shift, const double *ell_center, const double *ell_invcov,
                   double enlarge, double radiussq, int use_scan);
/* optional, before mlf_region_set with `live` on the DEVICE: the largest |live[i][k] - layer_ctr[k]| (it fixes the scale
 * of the binary16 proposal operand; without the hint the rows are fetched back once to find it).  Consumed by the next
 * mlf_region_set of the handle. */
int mlf_region_hint_live_extent(mlf_region *r, double amax);
/* per-handle tuning option (see mlf_set_option); mlf_region_get_option: the value IN FORCE for this handle (its override, else
 * the process default) -- what a caller that reports a routing should read.  A changed "filter_order" (process default or
 * per handle) takes effect with the next mlf_region_set / row replacement / batch of the handle, when the ordered operand is
 * rebuilt or dropped. */
int mlf_region_set_option(mlf_region *r, const char *name, long long value, int inherit);
int mlf_region_get_option(mlf_region *r, const char *name, long long *value);
/* in-place live point replacement, integrator.py:2753-2754; the row is in the space given by
 * live_space at mlf_region_set */
int mlf_region_update_point(mlf_region *r, size_t row, const double *live_row);
/* the same for `count` DISTINCT rows in one call (the driver replaces one live point per iteration and tests
 * membership once per refill, integrator.py:1776-1804: every replaced row since the last test goes in one upload
 * and two launches); live_rows = (count, d) row-major, the new contents of the live points rows[0..count) */
int mlf_region_update_points(mlf_region *r, size_t count, const int64_t *rows, const double *live_rows);
int mlf_region_set_thresholds(mlf_region *r, double enlarge, double radiussq);
int mlf_region_set_ellipsoid_center(mlf_region *r, const double *ell_center);
int mlf_region_inside(mlf_region *r, const double *pts, size_t np, uint8_t *mask);
/* device pointers; enqueues on `stream`; d_mask is np bytes */
int mlf_region_inside_dev(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                          void *stream);
/* scan only (K1 against the resident live points), device pointers */
int mlf_region_find_nearby_dev(mlf_region *r, const double *d_tpts, size_t np, int64_t *d_idx,
                               void *stream);

/* ---- vectorized likelihood batch (V1; signature of languages/c/mylib.c:33) -------------------
 * params is (n, d) row-major; like receives n values. */
int mlf_loglike_gauss(const double *params, size_t d, size_t n, const double *centers,
                      double sigma, double *like);          /* docs/gauss.py:25-27          */
int mlf_loglike_eggbox(const double *params, size_t d, size_t n, double *like);
                                                            /* examples/testeggbox.py:9-11  */
int mlf_loglike_eggbox2(const double *params, size_t d, size_t n, double *like);
                                                            /* test_PopSliceSampler.py:69-71 */
int mlf_loglike_rosenbrock(const double *params, size_t d, size_t n, double *like);
                                                            /* examples/testrosenbrock.py:10-13 */
/* kind: 0 gauss (aux = centers[d] then sigma), 1 eggbox, 2 eggbox2, 3 rosenbrock; device ptrs */
int mlf_loglike_dev(int kind, const double *d_params, size_t d, size_t n, const double *d_aux,
                    double sigma, double *d_like, void *stream);

/* ---- device-side proposal generation (opt-in) ---------------------------
 * Replaces the host draws of MLFriends.sample_from_boundingbox (mlfriends.pyx:1096-1112, method 0:
 * uniform in the unit cube) and sample_from_wrapping_ellipsoid (:1135-1160, method 1: uniform in the
 * wrapping ellipsoid, cube test) by a Philox-4x32-10 stream keyed by `seed`, starting at counter
 * `offset`; the nsamples proposals go through the region's membership pipeline on the device and
 * only the accepted rows (at most `capacity`, in draw order) are copied to `out`.  The random
 * stream differs from numpy's, so agreement with the reference is statistical.
 * mlf_region_set_axes provides ellipsoid_axes_T (d x d, reference :1232-1233) for method 1.
 * Methods 2 and 3 draw in whitened space: 2 = uniform in the padded bounding box of the live points
 * (sample_from_transformed_boundingbox :1114-1133), 3 = around random live points, thinned by the
 * number of balls containing the draw (sample_from_points :1072-1094); both are untransformed with
 * invT (AffineLayer.untransform :745-752), cube- and ellipsoid-tested.  mlf_region_set_sampling_data
 * provides invT (d x d) and bbox_lo / bbox_hi (reference :983-984).  Method 3 applies the cube and ellipsoid tests BEFORE it
 * counts the balls (each test is a function of the draw alone, the thinning uniform included: the accepted set is the
 * reference order's), so that the exact count runs on the survivors only. */
int mlf_region_set_axes(mlf_region *r, const double *axes_T);
int mlf_region_set_sampling_data(mlf_region *r, const double *invT, const double *bbox_lo, const double *bbox_hi);
int mlf_region_sample(mlf_region *r, int method, size_t nsamples, uint64_t seed, uint64_t offset,
                      double *out, size_t capacity, size_t *naccepted, uint64_t *next_offset);
/* One proposal batch of the driver's _refill_samples (integrator.py:1773-1837) without leaving the
 * device: mlf_region_sample's accepted points -> prior transform (tkind 0 identity, 1 x*a+b,
 * 2 (x*a)*b) -> likelihood (kind as in mlf_loglike_dev) -> only the points with L > Lmin are copied
 * back (u, p, L; at most `capacity`).  nevaluated = region-accepted proposals = the likelihood evaluations the reference would
 * make (methods 0 and 1 evaluate the batch where it was drawn, under the membership mask, when at least a quarter of it was
 * accepted: a rejected row then costs a discarded evaluation instead of a copy of the accepted ones). */
int mlf_region_refill(mlf_region *r, int method, size_t nsamples, uint64_t seed, uint64_t offset,
                      double Lmin, int tkind, double ta, double tb, int lkind, const double *aux,
                      double sigma, double *out_u, double *out_p, double *out_L, size_t capacity,
                      size_t *nevaluated, size_t *nkept, uint64_t *next_offset);
/* raw Philox blocks (counter = (i, 0, stream, 0), key = seed) for known-answer tests */
int mlf_debug_philox(uint64_t seed, unsigned stream, size_t nblocks, uint32_t *out);

/* ---- population step sampler -------------------------------------------
 * Stateless forms: one call = the reference function of the same name on host arrays
 * (ultranest/stepfuncs.pyx within_unit_cube :36-51, evolve :249-261 [mlf_evolve_propose],
 * evolve_update :99-183, step_back :285-334, update_vectorised_slice_sampler :537-630;
 * ultranest/popstepsampler.py unitcube_line_intersection :26-61, the distance part of
 * diagnose_move_distances :64-94).  Flags are uint8 (numpy bool), indices int64.
 *   mlf_evolve_propose: unif_full (one U[0,1) per walker, used by bisecting walkers:
 *     currentt = left + (right-left)*U, numpy's legacy uniform) may be NULL when currentt already
 *     holds the drawn coordinate; writes currentt, unew = u + v*t and the cube test.
 *   mlf_evolve_update: Lnew_full has one entry per walker (ignored where acceptable == 0).     */
int mlf_within_unit_cube(const double *u, size_t n, size_t d, uint8_t *out);
int mlf_evolve_propose(const double *currentu, const double *currentv, const double *left,
                       const double *right, const uint8_t *searching_left,
                       const uint8_t *searching_right, const double *unif_full, double *currentt,
                       size_t n, size_t d, double *unew, uint8_t *acceptable);
int mlf_evolve_update(const uint8_t *acceptable, const double *Lnew_full, double Lmin,
                      double *currentt, double *left, double *right, uint8_t *searching_left,
                      uint8_t *searching_right, uint8_t *success, size_t n);
int mlf_step_back(double Lmin, double *allL, size_t n, size_t ngen, int64_t *generation,
                  double *currentt);
int mlf_unitcube_line_intersection(const double *origin, const double *direction, size_t n,
                                   size_t d, double *tleft, double *tright);
int mlf_update_vectorised_slice_sampler(const double *t, double *tleft, double *tright,
                                        const double *proposed_L, const double *proposed_u,
                                        const double *proposed_p, int64_t *worker_running,
                                        int64_t *status, double threshold, double shrink_factor,
                                        double *allu, double *allL, double *allp, size_t popsize,
                                        size_t d, size_t nparams, int64_t *discarded);
int mlf_row_dist2(const double *a, const double *b, size_t n, size_t d, double *out);

/* Resident walker population = the state of PopulationSliceSampler (popstepsampler.py:347-697:
 * allu, allL, generation, currentt, currentv, current_left/right, searching_left/right,
 * currentp) kept in HBM.  One sampler step (= one __next__ of the reference, :610-697) is
 *   begin   : step_back(Lmin); returns generation[P] and flags[P]
 *             (bit0 bracket undefined, bit1 searching_left, bit2 searching_right)
 *   start   : setup_start for the listed walkers (:443-470)
 *   brackets: setup_brackets (:483-505) with host directions, or _philox with a device draw
 *             (kind 0-6 = the seven generate_*direction functions of stepfuncs.pyx:348-535)
 *   propose : evolve's proposal for all walkers with generation < nsteps; with unew_out the
 *             acceptable rows come back compacted in walker order for a host likelihood
 *   finish  : evolve_update + advance() bookkeeping + harvest of walker `ringindex`;
 *             finish_dev evaluates transform (tkind 0 identity, 1 x*a+b, 2 (x*a)*b) and
 *             likelihood (kind as in mlf_loglike_dev) on the device instead.
 * rec (9 + d + nparams doubles): [0] harvested, [1] L, [2] current_left, [3] current_right of the
 * ring walker, [4] nc (likelihood evaluations), [5] walkers moved on, [6] successes, [7] moves
 * farther than the MLFriends radius, [8] sum log(distance/radius + 1e-10), then u and p.        */
typedef struct mlf_walkers mlf_walkers;
int mlf_walkers_create(mlf_walkers **out, size_t popsize, size_t nsteps, size_t d);
int mlf_walkers_destroy(mlf_walkers *w);
int mlf_walkers_reset(mlf_walkers *w);
int mlf_walkers_begin(mlf_walkers *w, double Lmin, int64_t *generation, uint8_t *flags);
int mlf_walkers_start(mlf_walkers *w, const int64_t *idx, size_t n, const double *u_rows,
                      const double *L);
int mlf_walkers_points(mlf_walkers *w, const int64_t *idx, size_t n, double *out_rows);
int mlf_walkers_brackets(mlf_walkers *w, const int64_t *idx, size_t n, double scale,
                         const double *v_rows);
int mlf_walkers_set_direction_data(mlf_walkers *w, const double *axes, const double *live,
                                   size_t nlive, const double *std);
int mlf_walkers_brackets_philox(mlf_walkers *w, double scale, int kind, double dirscale,
                                uint64_t seed, uint64_t offset, uint64_t *next_offset);
int mlf_walkers_set_layer(mlf_walkers *w, int kind, const double *ctr, const double *mat,
                          const double *wrap, double maxradiussq);
int mlf_walkers_propose(mlf_walkers *w, const double *unif, uint64_t seed, uint64_t offset,
                        double *unew_out, size_t *nacc);
int mlf_walkers_finish(mlf_walkers *w, double Lmin, const double *pnew, const double *Lnew,
                       size_t nacc, size_t nparams, int64_t ringindex, double *rec);
int mlf_walkers_finish_dev(mlf_walkers *w, double Lmin, int tkind, double ta, double tb, int lkind,
                           const double *aux, double sigma, int64_t ringindex, double *rec);
/* Whole step on the device (Philox stream, resident likelihood): step_back, restarts from the live
 * points uploaded by mlf_walkers_set_live (uniform among those with L > Lmin), ring index kept on the
 * device, new slices, proposal, transform + likelihood, update, harvest -- one record back, one
 * synchronisation.  rec has 10 + 2 d doubles: as above, then the ring index after the step. */
int mlf_walkers_set_live(mlf_walkers *w, const double *us, const double *Ls, size_t nlive);
/* the rows `rows[0 ... count)` of that device copy replaced (the driver replaces ONE live point per iteration,
 * integrator.py:2753-2754: uploading the whole set again costs two pageable copies per sampler call); queued on the
 * library's stream, no synchronisation */
int mlf_walkers_update_live(mlf_walkers *w, const int64_t *rows, size_t count, const double *us_rows, const double *Ls_rows);
int mlf_walkers_step_dev(mlf_walkers *w, double Lmin, double scale, int dirkind, double dirscale,
                         uint64_t seed, uint64_t offset, int tkind, double ta, double tb, int lkind,
                         const double *aux, double sigma, double *rec, uint64_t *next_offset);
/* The same step replayed as one hipGraph launch: the kernel sequence and the record copy are captured once
 * (and again whenever a captured argument changes: kinds, buffer addresses); Lmin, scale, seed / offset and
 * the MLFriends radius reach the kernels through a pinned parameter block. */
int mlf_walkers_step_graph(mlf_walkers *w, double Lmin, double scale, int dirkind, double dirscale,
                           uint64_t seed, uint64_t offset, int tkind, double ta, double tb, int lkind,
                           const double *aux, double sigma, double *rec, uint64_t *next_offset);
/* SEVERAL such steps ("rounds") without a host round trip in between.  PopulationSliceSampler.__next__ advances every walker
 * by one likelihood evaluation and returns a point only when the walker under the ring index has finished its nsteps
 * (popstepsampler.py:610-697); the driver calls it again and again with the SAME threshold, live points and scale until
 * it does (integrator.py:1839-1950).  This call runs that loop on the device: round 0 for every walker, then the ring
 * walker alone until it has finished (at most max_rounds rounds; that fixes the number R of rounds the driver's loop
 * would have made), then the other walkers' rounds 1 ... R - 1, each walker's rounds back to back inside the wave that owns
 * it.  Round r draws from the Philox counters call r of mlf_walkers_step_dev would use, so the resident state, the
 * record and *next_offset are those of R consecutive mlf_walkers_step_dev calls, bit for bit.
 *   rec         10 + 2 d doubles: [0] harvested, [1] L, [2] left, [3] right, [4] R, [9 ...] u (d), p (d), ring index after
 *   round_rows  max_rounds x 5 doubles, rows 0 ... R - 1 filled: (likelihood evaluations, movable walkers, successes,
 *               far-enough moves, sum log(distance / radius + 1e-10)) of each round -- one row of the sampler's logstat each
 *   *rounds     R.  rec[0] == 0 with R == max_rounds: call again (the driver's loop does exactly that).
 * max_rounds is clamped to 4096 and to 64 MiB of per-round bookkeeping (9 bytes per walker and round).  Rounds after the first
 * run with the walker's state in registers where the shapes allow it (even d <= 64, affine layer or none); a NEGATIVE max_rounds
 * (test hook) forces |max_rounds| rounds through the general form, which passes every stage through global memory. */
int mlf_walkers_rounds_dev(mlf_walkers *w, double Lmin, double scale, int dirkind, double dirscale,
                           uint64_t seed, uint64_t offset, int tkind, double ta, double tb, int lkind,
                           const double *aux, double sigma, int max_rounds, double *rec, double *round_rows,
                           int *rounds, uint64_t *next_offset);
int mlf_walkers_export(mlf_walkers *w, double *allu, double *allL, int64_t *generation,
                       double *currentt, double *currentv, double *left, double *right,
                       uint8_t *searching_left, uint8_t *searching_right);

/* ---- integrator bookkeeping on the host (no GPU involved) -----------------
 * MultiCounter.passing_node of ultranest/netiter.py:721-855 for (nbootstraps + 1) counters, with the
 * insertion-order U test of ultranest/ordertest.py.  member: [ncounters][nroots] uint8, row 0 all
 * ones (reference MultiCounter.__init__ :611-621 draws it).  passing_node: the node (root id, value
 * Li, values of its children) leaves the active set whose root ids / values are given; logwidth
 * (ncounters) is what the reference appends to `logweights`; random_beta = the
 * np.random.beta(1, nlive) draws when the counter was created with random != 0 (else NULL).
 * state: scalars[10] = logZ, logZerr, logVolremaining, logZremainMax, logZremain, remainder_ratio,
 * remainder_fraction, accumulator N, accumulator U, iterations; arrays of ncounters (any NULL). */
typedef struct mlf_counter mlf_counter;
int mlf_counter_create(mlf_counter **out, size_t nroots, size_t ncounters, const uint8_t *member,
                       int random, int check_insertion_order);
int mlf_counter_destroy(mlf_counter *c);
int mlf_counter_reset(mlf_counter *c);
int mlf_counter_passing_node(mlf_counter *c, int64_t rootid, double Li, size_t nchildren,
                             const double *child_values, const int64_t *rootids,
                             const double *parallel_values, size_t nparallel,
                             const double *random_beta, double *logwidth);
int mlf_counter_state(const mlf_counter *c, double *scalars, double *all_H, double *all_logZ,
                      double *all_logVolremaining, double *all_logZremain, int64_t *runs,
                      size_t runs_capacity, size_t *nruns);

/* host helper (no GPU): rows in which two (n, d) float64 matrices differ bytewise; used by the lazy device
 * mirror of the live points (the driver replaces one row per iteration in place, integrator.py:2753) */
int mlf_host_changed_rows(const double *a, const double *b, size_t n, size_t d, int64_t *rows,
                          size_t capacity, size_t *count);

/* host helper (no GPU): (nrounds, npoints) bootstrap selection masks from numpy's legacy MT19937 stream, the
 * same draws as nrounds calls of np.random.randint(npoints, size=npoints) (reference mlfriends.pyx:1044-1047,
 * :565-566); key[624] / pos = np.random.get_state()[1:3], advanced in place */
int mlf_host_draw_selection(uint32_t *key, int32_t *pos, size_t npoints, size_t nrounds, uint8_t *masks);

/* H3 -> T1 -> K1 with the index kept: d_idx[p] = first live point within radiussq (>= 0),
 * -1 = inside the ellipsoid but no neighbour, -2 = outside the wrapping ellipsoid. */
int mlf_region_first_index_dev(mlf_region *r, const double *d_pts, size_t np, int64_t *d_idx,
                               void *stream);

/* ---- bench / profiling helpers ---------------------------------------------------------- */
/* Runs the neighbour-scan kernel `reps` times on device data and returns the mean kernel time
 * in milliseconds measured with hipEvents on `stream`. */
int mlf_region_time_inside_dev(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                               void *stream, int reps, float *ms_total, float *ms_scan);

/* mlf_region_inside_dev with hipEvents recorded around the per-proposal stage and the
 * neighbour scan, on `stream`, without synchronising.  mlf_region_timing_collect waits for the
 * recorded events and returns the number of timed calls and the summed milliseconds since the last
 * collect of: the per-proposal stage (k_prep*), the scan kernel (k_filter, or k_scan when the filter
 * is off), and the rest of the scan stage (exact re-check, routing, finalisation). */
int mlf_region_inside_dev_timed(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                                void *stream);
int mlf_region_timing_collect(mlf_region *r, int *ncalls, double *ms_prep, double *ms_scan,
                              double *ms_rest);
/* k_filter launches of the timed calls since the last collect: their number and summed duration
 * (events bracket each launch on `stream`; the compaction kernels between phases are excluded) */
int mlf_region_timing_filter_launches(mlf_region *r, int *nlaunches, double *ms_total);
/* the same launches one by one, in launch order (a timed call with p phases contributes p consecutive
 * entries); up to cap durations are written, *nlaunches is their full number.  Does not reset. */
int mlf_region_timing_filter_launch_ms(mlf_region *r, double *ms, int cap, int *nlaunches);
/* whether a batch of np proposals takes the MFMA pre-filter, and its GEMM shape (K columns per pair,
 * number of 32-row live-point tiles) */
int mlf_region_filter_info(mlf_region *r, size_t np, int *active, int *kdim, int *ntiles32);
/* ---- multi-GPU exchange step (reference integrator.py:395-404: gather + bcast + np.max) --------
 * MAX all-reduce of a few doubles over RCCL for callers without torch.distributed.  librccl is dlopen()ed at the
 * first call.  One process per GPU: rank 0 obtains 128 bytes with mlf_comm_unique_id and distributes them (MPI, file,
 * socket ...), every rank calls mlf_comm_init_rank after mlf_set_device; mlf_allreduce_max(v, count) then leaves the
 * element-wise maximum over the ranks in v.  mlf_comm_init(ndev) is the EXCHANGE STEP ONLY for a process that holds ndev
 * rows of `count` doubles (one per device 0 ... ndev - 1) obtained elsewhere: mlf_allreduce_max then leaves the column maxima in
 * every row.  It does NOT make the compute entry points multi-device: the library has ONE compute context per process (the
 * device of mlf_set_device, one stream, shared scratch; the C ABI is not re-entrant) -- compute on several GPUs
 * means one process per GPU and mlf_comm_init_rank.  (Tested with ndev = the devices present, 1 on this pool:
 * tests/test_distributed.py::test_abi_allreduce_max_one_rank_and_one_process.)  NaN must travel as an explicit flag element,
 * not through MAX (the Python layer does that: ultranest_amd.distributed). */
int mlf_comm_unique_id(char *id_out, size_t len /* >= 128 */);
int mlf_comm_init_rank(const char *id, size_t len, int nranks, int rank);
int mlf_comm_init(int ndev);
int mlf_allreduce_max(double *values, size_t count);
int mlf_comm_destroy(void);

/* Diagnostic counters of the LAST filtered batch of this region (synchronises the device): out[0] proposals the
 * bounded ellipsoid form could not decide (decided in binary64 by the tail of the re-check launch), out[1] reserved
 * (0), out[2] uncertain pairs listed by the pre-filter, out[3] largest list segment, out[4]
 * list segments, out[5] 32-query groups left for the second live-point range, out[6] (cap > 6) queries whose minimum
 * over all live points ended in the band (the set the min-only sweep hands to its listing pass), out[7] (cap > 7)
 * 32-query groups left for the third range ("filter_second_range_pct"), out[8 ... 15] (cap >= 16) shader-clock stamps of the
 * stage boundaries of one workgroup of the last launch that records them, out[16], out[17] (cap >= 18) the tile cuts of the last
 * min-only batch (the second is 0 with two ranges), out[18] (cap >= 19) 1 if the last fused first launch read the ellipsoid's
 * quadratic form off the whitening chain ("fused_variant" bit 1).  cap >= 6. */
int mlf_region_debug_stats(mlf_region *r, unsigned long long *out, int cap);
/* diagnostics of the fused first launch (k_prep_sweep): returns in out[0 ... 12] the shader-clock stamps wave 0 of the workgroup
 * chosen by the PREVIOUS call wrote during the last phased batch -- [0] entry, [1] matrix fragments in LDS, [2 + 2 g] rows of
 * query group g landed, [3 + 2 g] group g's per-proposal stage done (g = 0 ... 3), [10] first-range sweep starts, [11] sweep
 * done, [12] compaction done -- and selects workgroup `block` for the following batches (block < 0: off).
 * block = 1 000 000 + b / 2 000 000 + b: workgroup b of the k_sweep_min launch that follows the first range / of the one after
 * it instead -- [0] entry, [1] operands and the first two tiles asked for, [2] first tile done, [3] tile loop done, [4] fates
 * known, [5] / [6] past the two barriers, [7] compaction stores issued; [8] / [9] the 100 MHz clock all workgroups share at the
 * start / end of workgroup 0, [10] / [11] of workgroup b (scripts/sweep_stamps.py). */
int mlf_region_debug_fused_stamps(mlf_region *r, int block, unsigned long long *out, int cap);
/* Measured issue rate of independent v_add_f64/v_mul_f64 (the non-fused FP64 vector rate that
 * bounds the distance kernels), in Tera-instructions*lanes per second (= TFLOP/s, 1 flop each). */
int mlf_bench_fp64_valu(double *tflops);

#ifdef __cplusplus
}
#endif
#endif /* MLFRIENDS_HIP_H */
