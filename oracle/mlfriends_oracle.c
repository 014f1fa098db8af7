/*
 * oracle/mlfriends_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the numerical contract of the
 * UltraNest MLFriends hot path (reference = /root/reference, UltraNest 4.5.0).
 * It exists to CHECK the HIP kernels (tests/, __graft_entry__.smoke()) and to
 * serve as the timed "port" CPU baseline in bench.py.  Nothing under
 * ultranest_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py compares every function
 * here against golden vectors produced by the real reference (Cython build of
 * ultranest/mlfriends.pyx, imported in the dev container by
 * tests/golden/make_golden.py) -- bit-exact for indices, counts, masks, radii.
 *
 * Arithmetic contract restated (all must be compiled with -ffp-contract=off,
 * no -ffast-math, so that no FMA is formed; the x86-64 reference build has none):
 *   dist2(a,b) = (((0 + (a0-b0)^2) + (a1-b1)^2) + ...)  k ascending,
 *                 sub, mul, add individually rounded in binary64
 *                 [mlfriends.pyx:63-66, 102-104, 178-180, 217-219, 263-265]
 *   membership  = dist2 <= radiussq (inclusive)   [mlfriends.pyx:67,105,181]
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double dist2_seq(const double *a, const double *b, size_t d)
{
    double acc = 0.0;
    for (size_t k = 0; k < d; k++) {
        double diff = a[k] - b[k];
        acc += diff * diff;
    }
    return acc;
}

/* K1 -- find_nearby, mlfriends.pyx:143-183.
 * out[j] = lowest i with dist2(apts[i], bpts[j]) <= radiussq, else -1. */
void orc_find_nearby(const double *apts, size_t na, const double *bpts, size_t nb,
                     size_t d, double radiussq, int64_t *out)
{
    for (size_t j = 0; j < nb; j++) {
        out[j] = -1;
        for (size_t i = 0; i < na; i++) {
            if (dist2_seq(apts + i * d, bpts + j * d, d) <= radiussq) {
                out[j] = (int64_t)i;
                break;
            }
        }
    }
}

/* K2 -- count_nearby, mlfriends.pyx:31-68 (no early exit). */
void orc_count_nearby(const double *apts, size_t na, const double *bpts, size_t nb,
                      size_t d, double radiussq, int64_t *out)
{
    for (size_t j = 0; j < nb; j++) {
        int64_t c = 0;
        for (size_t i = 0; i < na; i++)
            if (dist2_seq(apts + i * d, bpts + j * d, d) <= radiussq)
                c++;
        out[j] = c;
    }
}

/* K3 -- _subtract_nearby, mlfriends.pyx:73-113.
 * out[j,:] = pts[j,:] - mean_{i: dist2(i,j) <= r2} pts[i,:]; the neighbour sum
 * is accumulated in ascending i (the point itself is always a neighbour) and
 * divided by (double)count. */
void orc_subtract_nearby(const double *pts, size_t n, size_t d, double radiussq, double *out)
{
    for (size_t j = 0; j < n; j++) {
        double *o = out + j * d;
        size_t nn = 0;
        for (size_t k = 0; k < d; k++)
            o[k] = 0.0;
        for (size_t i = 0; i < n; i++) {
            if (dist2_seq(pts + i * d, pts + j * d, d) <= radiussq) {
                nn++;
                for (size_t k = 0; k < d; k++)
                    o[k] += pts[i * d + k];
            }
        }
        for (size_t k = 0; k < d; k++)
            o[k] = pts[j * d + k] - o[k] / (double)nn;
    }
}

/* K4 -- compute_maxradiussq, mlfriends.pyx:188-224.  The reference function is
 * declared `cdef float`, so the binary64 maximum is narrowed to binary32 on
 * return (round-to-nearest-even); the C `float` return type reproduces that. */
float orc_maxradiussq(const double *apts, size_t na, const double *bpts, size_t nb, size_t d)
{
    double maxd = 0.0;
    for (size_t j = 0; j < nb; j++) {
        double mind = 1e300;
        for (size_t i = 0; i < na; i++) {
            double dd = dist2_seq(apts + i * d, bpts + j * d, d);
            mind = dd < mind ? dd : mind; /* Cython min(mind, d) */
        }
        maxd = mind > maxd ? mind : maxd; /* Cython max(maxd, mind) */
    }
    return (float)maxd;
}

/* R2 inner loop -- MLFriends.compute_enlargement, mlfriends.pyx:1044-1054:
 * for bootstrap b, a = pts[selected[b]], b = pts[~selected[b]].
 * selected is (B, n) uint8.  maxd_out[b] = (double)(float) result; bootstraps
 * that select all or no points are skipped by the reference (:1048) -> 0.0 here
 * with skipped_out[b] = 1. */
void orc_maxradiussq_bootstrap(const double *pts, size_t n, size_t d,
                               const uint8_t *selected, size_t B,
                               double *maxd_out, uint8_t *skipped_out)
{
    double *a = (double *)malloc(sizeof(double) * n * d);
    double *b = (double *)malloc(sizeof(double) * n * d);
    for (size_t r = 0; r < B; r++) {
        const uint8_t *sel = selected + r * n;
        size_t na = 0, nb = 0;
        for (size_t i = 0; i < n; i++) {
            if (sel[i])
                memcpy(a + (na++) * d, pts + i * d, sizeof(double) * d);
            else
                memcpy(b + (nb++) * d, pts + i * d, sizeof(double) * d);
        }
        if (na == 0 || nb == 0) {
            maxd_out[r] = 0.0;
            if (skipped_out) skipped_out[r] = 1;
            continue;
        }
        if (skipped_out) skipped_out[r] = 0;
        maxd_out[r] = (double)orc_maxradiussq(a, na, b, nb, d);
    }
    free(a);
    free(b);
}

/* K5 -- compute_mean_pair_distance, mlfriends.pyx:229-270.  One running sum in
 * (j outer, i<j inner) order; pairs need equal, non-zero cluster ids.
 * Npairs is a C int in the reference (:253). */
double orc_mean_pair_distance(const double *pts, size_t n, size_t d, const int64_t *ids)
{
    double total = 0.0;
    int npairs = 0;
    for (size_t j = 0; j < n; j++) {
        if (ids[j] == 0)
            continue;
        for (size_t i = 0; i < j; i++) {
            if (ids[j] == ids[i]) {
                total += sqrt(dist2_seq(pts + i * d, pts + j * d, d));
                npairs++;
            }
        }
    }
    return total / npairs;
}

/* H3 -- _inside_ellipsoid, mlfriends.pyx:882-912:
 *   delta = points - center;  r = einsum('ij,jk,ik->i', delta, invcov, delta);
 *   mask = r <= square_radius.
 * numpy's un-optimised three-operand einsum (numpy 2.2.6, probed bit-for-bit in
 * the dev container) evaluates one accumulator per point in (j outer, k inner)
 * order with the product associated as (delta_j * A_jk) * delta_k and no FMA.
 * q_out (optional) receives r. */
void orc_inside_ellipsoid(const double *pts, size_t np_, size_t d, const double *ctr,
                          const double *invcov, double sqradius, uint8_t *mask, double *q_out)
{
    double *delta = (double *)malloc(sizeof(double) * d);
    for (size_t p = 0; p < np_; p++) {
        for (size_t k = 0; k < d; k++)
            delta[k] = pts[p * d + k] - ctr[k];
        double acc = 0.0;
        for (size_t j = 0; j < d; j++)
            for (size_t k = 0; k < d; k++)
                acc += (delta[j] * invcov[j * d + k]) * delta[k];
        if (q_out) q_out[p] = acc;
        mask[p] = acc <= sqradius;
    }
    free(delta);
}

/* T1 -- AffineLayer.transform without wraps, mlfriends.pyx:737-743:
 * out = (pts - ctr) . T.  The reference calls BLAS dgemm whose summation order
 * is implementation defined ("parity unpinned" at this stage boundary; see
 * SURVEY.md 8c).  Restated here as a k-ascending fused-multiply-add chain, which
 * is what OpenBLAS 0.3.29 produced for most shapes probed; tolerance-class. */
void orc_affine_transform(const double *pts, size_t np_, size_t d, const double *ctr,
                          const double *T, double *out)
{
    double *delta = (double *)malloc(sizeof(double) * d);
    for (size_t p = 0; p < np_; p++) {
        for (size_t k = 0; k < d; k++)
            delta[k] = pts[p * d + k] - ctr[k];
        for (size_t c = 0; c < d; c++) {
            double acc = 0.0;
            for (size_t k = 0; k < d; k++)
                acc = fma(delta[k], T[k * d + c], acc);
            out[p * d + c] = acc;
        }
    }
    free(delta);
}

/* R3 -- MLFriends.inside, mlfriends.pyx:1186-1211, composed from H3, T1, K1 for
 * a layer without wrapped dims.  mask[p] = inside ellipsoid AND has a live
 * point within radiussq in t-space. */
void orc_region_inside(const double *pts, size_t np_, size_t d,
                       const double *unormed, size_t n,
                       const double *layer_ctr, const double *layer_T,
                       const double *ell_ctr, const double *ell_invcov, double enlarge,
                       double radiussq, uint8_t *mask)
{
    double *t = (double *)malloc(sizeof(double) * d);
    orc_inside_ellipsoid(pts, np_, d, ell_ctr, ell_invcov, enlarge, mask, NULL);
    for (size_t p = 0; p < np_; p++) {
        if (!mask[p])
            continue;
        int64_t idx;
        orc_affine_transform(pts + p * d, 1, d, layer_ctr, layer_T, t);
        orc_find_nearby(unormed, n, t, 1, d, radiussq, &idx);
        mask[p] = idx >= 0;
    }
    free(t);
}

/* ---- V1/L1-L3: benchmark likelihoods, (params, d, n, like) convention of
 * languages/c/mylib.c:33.  Tolerance-class (1e-12 relative): numpy's pairwise
 * sum(axis=1) and libm cos are not bit-reproduced. ---- */

/* L1 -- docs/gauss.py:25-27 with scalar centre/sigma per dimension array. */
void orc_loglike_gauss(const double *params, size_t d, size_t n, const double *centers,
                       double sigma, double *like)
{
    const double norm = -0.5 * log(2.0 * M_PI * sigma * sigma) * (double)d;
    for (size_t j = 0; j < n; j++) {
        double s = 0.0;
        for (size_t k = 0; k < d; k++) {
            double z = (params[j * d + k] - centers[k]) / sigma;
            s += z * z;
        }
        like[j] = -0.5 * s + norm;
    }
}

/* L2 -- examples/testeggbox.py:9-11: (2 + prod_k cos(z_k/2))**5. */
void orc_loglike_eggbox(const double *params, size_t d, size_t n, double *like)
{
    for (size_t j = 0; j < n; j++) {
        double chi = 1.0;
        for (size_t k = 0; k < d; k++)
            chi *= cos(params[j * d + k] / 2.0);
        like[j] = pow(2.0 + chi, 5.0);
    }
}

/* L2' -- examples/test_PopSliceSampler.py:69-71: (prod_k cos(theta_k))**2. */
void orc_loglike_eggbox2(const double *params, size_t d, size_t n, double *like)
{
    for (size_t j = 0; j < n; j++) {
        double chi = 1.0;
        for (size_t k = 0; k < d; k++)
            chi *= cos(params[j * d + k]);
        like[j] = chi * chi;
    }
}

/* L3 -- examples/testrosenbrock.py:10-13. */
void orc_loglike_rosenbrock(const double *params, size_t d, size_t n, double *like)
{
    for (size_t j = 0; j < n; j++) {
        double s = 0.0;
        for (size_t k = 0; k + 1 < d; k++) {
            double a = params[j * d + k], b = params[j * d + k + 1];
            double t = b - a * a;
            double u = 1.0 - a;
            s += 100.0 * (t * t) + u * u;
        }
        like[j] = -2.0 * s;
    }
}
