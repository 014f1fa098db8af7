"""Opt-in DEVICE-RESIDENT region rebuild (counterpart of the reference driver's `_update_region` body,
integrator.py:2055-2122: `transformLayer.create_new` -> `region_class(active_u, layer)` -> bootstrap ->
`create_ellipsoid` -> `inside(active_u).all()`; the layer mathematics of mlfriends.pyx:666-710, 827-850, the region's
of :927-986, 1017-1070, 1213-1237).

The default rebuild keeps every d x d and N x d numpy call of the reference on the host, so that `T`, `unormed` and
with them a seeded run's trajectory are the reference's bit for bit -- and pays for it: 3.1 of its 5.7 ms at N = 4000,
d = 50 are host numpy (np.cov twice, np.dot twice, reductions) and 0.7 ms eight pageable uploads of the same two
arrays.  Here the live points are uploaded ONCE into memory of the library's own allocator (mlf_dev_alloc: no torch in
this module since round 4) and everything that touches all N points is one of the library's kernels:

  cube test, |u - ctr| extent, bounding box    mlf_col_extent                  column minima / maxima (exact)
  whitening with the old and the new layer     mlf_affine_transform            k_whiten_rows: the SAME k-ascending FMA chain
                                                                               that whitens the region's live points and the
                                                                               re-checked proposals (device in, device out)
  friends-of-friends labels                    mlf_adjacency_bits + replay     one all-pairs pass + bit-row replay (worker thread)
  neighbour-mean subtraction (LocalAffineLayer) mlf_subtract_nearby            device in, device out
  means / covariances                          mlf_bootstrap_moments           the bootstrap's own mean / covariance kernels with
                                                                               an all-ones selection (fixed summation order)
  bootstrapped radius and enlargement          mlf_maxradiussq_bootstrap / mlf_bootstrap_factor on device pointers
  membership of the live points                mlf_region_set + mlf_region_inside_dev on device pointers

and the host keeps what is O(d^3) on one d x d matrix: `eigh` of the layer covariance (T = V L^-1/2, invT = L^1/2 V^T,
logvolscale = 1/2 sum log L: no explicit inverse), `inv` + two `eigh` for the wrapping ellipsoid.

TOLERANCE CLASS, not bit parity with the default path: T, cov agree to ~1e-12 relative (different summation orders in the
moments; eigh instead of inv), `unormed` with them; GIVEN T, the whitening is the region's own chain, so the radius is the
kernel's exact answer for the `unormed` it is handed (tests/test_device_rebuild.py: against the golden bootstrap vectors
g3 on their own inputs bit for bit, against the default path equal-or-adjacent binary32).  Cluster labels exactly (as long
as no pair sits within rounding of the linking length); masks are exact FOR THE REGION AS BUILT.  `np.random` is consumed
exactly as by the default path (one `randint(N, size=N)` per bootstrap round).  Supported: AffineLayer / LocalAffineLayer
without wrapped axes, MLFriends, minvol = 0, d <= 64, one process; anything else -> `supported()` is False and the caller
uses the default path.
"""
import ctypes

import numpy as np

from . import _lib, kernels, regions
from .layers import AffineLayer, LocalAffineLayer, MaxPrincipleGapAffineLayer, int_dtype


def supported(layer, region_class, ndim, minvol, world_size=1):
    if world_size != 1 or minvol != 0 or ndim > 64 or region_class is not regions.MLFriends:
        return False
    if type(layer) not in (AffineLayer, LocalAffineLayer) or isinstance(layer, MaxPrincipleGapAffineLayer):
        return False
    if getattr(layer, "has_wraps", False) or np.ndim(getattr(layer, "T", 1)) != 2:
        return False
    try:
        return _lib.device_count() >= 1
    except Exception:
        return False


def _p(a):
    return a.ptr


def _ellipsoid_axes(cov):
    """create_ellipsoid's LAPACK calls (mlfriends.pyx:1224-1235) in the reference's order; errstate is per thread."""
    with np.errstate(all='raise'):
        precision = np.linalg.inv(cov)
        lam, vec = np.linalg.eigh(precision)
        axlens = 1. / np.sqrt(lam)
        axes = np.dot(vec, np.diag(axlens))
        lam2, vec2 = np.linalg.eigh(cov)
        inv_axlens = 1. / np.sqrt(lam2)
        inv_axes = np.dot(vec2, np.diag(inv_axlens))
    return precision, (axlens, axes, axes.transpose(), inv_axlens, inv_axes)


class DeviceRebuild(object):
    """One instance per RegionUpdater; keeps nothing between rebuilds."""

    def __init__(self):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=2)
        self.trace = None     # a list: (label, seconds) checkpoints of the next call, each behind a device synchronisation

    def _mark(self, label):
        if self.trace is not None:
            import time
            _lib.check(_lib.lib().mlf_synchronize())
            self.trace.append((label, time.perf_counter()))

    def next_region(self, active_u, layer, maxradiussq, nbootstraps):
        """(next layer, next region with radius / enlargement / ellipsoid set, all live points inside?) -- what the
        default path obtains from create_new, the region constructor, the bootstrap, create_ellipsoid and inside()."""
        self._draw_job = None
        self._draws_due = False
        state = np.random.get_state()
        try:
            return self._next_region(active_u, layer, maxradiussq, nbootstraps)
        except BaseException:
            # the masks are drawn ahead of time by a worker; a failure in front of the bootstrap leaves np.random where
            # the default path would have left it: untouched
            if self._draw_job is not None and not self._draws_due:
                try:
                    self._draw_job.result()
                finally:
                    np.random.set_state(state)
            raise

    def _next_region(self, active_u, layer, maxradiussq, nbootstraps):
        L = _lib.lib()
        DevArray = kernels.DevArray
        active_u = np.ascontiguousarray(active_u, dtype=np.float64)
        n, d = active_u.shape
        nbytes = n * d * 8
        self._mark('start')
        # the bootstrap's selection masks (0.17 ms of host work) are drawn by a worker while the device starts: nothing
        # else touches np.random before they are used, so the stream is consumed exactly as by the default path
        draw_job = self._draw_job = self.pool.submit(regions._draw_selection, np.random, n, nbootstraps)
        U = DevArray.from_host(active_u)                              # the ONE upload of the live points
        ulo, uhi = kernels.col_extent(U, n, d)                        # NaN propagates: a non-finite row fails the test
        in_cube = bool((ulo > 0).all() and (uhi < 1).all())
        self._mark('upload + cube test')
        if not in_cube:
            ok = np.logical_and(active_u > 0, active_u < 1)
            raise ValueError("not all u values are between 0 and 1: %s" % active_u[~ok.all(axis=1)])
        # ---- create_new (mlfriends.pyx:712-735, 827-850): clusters in the OLD layer's space ------------------------
        ctr_old = np.ascontiguousarray(np.broadcast_to(layer.ctr, (d,)), dtype=np.float64)
        T_old = np.ascontiguousarray(layer.T, dtype=np.float64)
        t_old = DevArray(nbytes)
        _lib.check(L.mlf_affine_transform(_p(U), n, d, _lib.ptr(ctr_old), _lib.ptr(T_old), None, _p(t_old)))
        prev = None if layer.clusterids is None else np.ascontiguousarray(np.asarray(layer.clusterids)[:n], dtype=int_dtype)
        labels = np.empty(n, dtype=int_dtype)
        ncl = ctypes.c_int64(0)
        adj = ctypes.c_void_p()
        _lib.check(L.mlf_adjacency_bits(_p(t_old), n, d, float(maxradiussq), ctypes.byref(adj)))
        # the growth rounds are replayed on the bit rows by the worker thread (host only: 0.5 ms at n = 4000) while the
        # device goes on; only AffineLayer needs the labels before its covariance
        label_job = self.pool.submit(L.mlf_host_cluster_replay, adj, n, _lib.ptr(prev), _lib.ptr(labels), ctypes.byref(ncl))
        local = type(layer) is LocalAffineLayer
        if not local:
            _lib.check(label_job.result())
        nclusters = int(ncl.value)
        self._mark('old-layer whitening + cluster labels')
        if local:
            centred = DevArray(nbytes)
            _lib.check(L.mlf_subtract_nearby(_p(U), n, d, float(maxradiussq), _p(centred)))   # u-space points, t-space radius (:844-849)
        elif nclusters == 1:
            centred = U
        else:
            # cluster means removed; a singleton is centred on the global mean (:333-341).  A handful of group means over
            # an index set the host has just computed: the reference's own numpy statements on the host copy, one upload
            chost = np.empty_like(active_u)
            everyone = None
            for cid in np.unique(labels):
                members = labels == cid
                group = active_u[members]
                if group.shape[0] > 1:
                    centre = group.mean(axis=0)
                else:
                    if everyone is None:
                        everyone = active_u.mean(axis=0)
                    centre = everyone
                chost[members] = group - centre
            centred = DevArray.from_host(chost)
        self._mark('neighbour-mean subtraction')
        # ---- optimize (mlfriends.pyx:666-710): mean and covariance of the centred points (layer) and of the points
        # themselves (wrapping ellipsoid, :447-476) -- the bootstrap's moment kernels with every row selected
        ones = np.ones((1, n), dtype=np.uint8)
        cmean, ccov = np.empty((1, d)), np.empty((1, d, d))
        _lib.check(L.mlf_bootstrap_moments(_p(centred), n, d, _lib.ptr(ones), 1, _lib.ptr(cmean), _lib.ptr(ccov)))
        if centred is U:
            umean, ucov = cmean, ccov
        else:
            umean, ucov = np.empty((1, d)), np.empty((1, d, d))
            _lib.check(L.mlf_bootstrap_moments(_p(U), n, d, _lib.ptr(ones), 1, _lib.ptr(umean), _lib.ptr(ucov)))
        ctr = umean[0].copy()
        cov = ccov[0] * (d + 2.0)
        ecov = ucov[0] * (d + 2.0)
        cov = 0.5 * (cov + cov.T)
        ecov = 0.5 * (ecov + ecov.T)
        amax = float(max((uhi - ctr).max(), (ctr - ulo).max()))       # = max |u - ctr| (subtraction is monotone)
        # the wrapping ellipsoid's LAPACK calls (inv, two eigh: 0.35 ms) only need ecov: they run on a worker thread
        # next to the device calls below (ctypes and LAPACK both release the interpreter lock)
        assert np.isfinite(ecov).all(), ecov
        ell_job = self.pool.submit(_ellipsoid_axes, ecov)
        self._mark('means, covariances, download')
        eigval, eigvec = np.linalg.eigh(cov)
        if not (eigval > 0).all():
            raise np.linalg.LinAlgError("Singular matrix")           # what inv(cov) raises in the reference (:697)
        logvolscale = 0.5 * np.sum(np.log(eigval))                   # = -0.5 slogdet(inv(cov))
        floor = eigval.max() * 1e-40
        eigval = np.where(eigval < floor, floor, eigval)
        T = np.ascontiguousarray(eigvec * eigval**-0.5)
        invT = (eigvec * eigval**0.5).T
        _lib.check(label_job.result())
        nclusters = int(ncl.value)
        nxt = type(layer)(ctr=ctr, T=T, invT=invT, nclusters=nclusters, wrapped_dims=layer.wrapped_dims, clusterids=labels)
        nxt.cov = cov
        nxt.logvolscale = logvolscale
        nxt.axes = invT
        self._mark('eigh, layer object')
        # ---- region constructor (mlfriends.pyx:927-986): the whitening chain of the region's own live points ----------
        unormed_d = t_old                                              # its buffer is free again
        _lib.check(L.mlf_affine_transform(_p(U), n, d, _lib.ptr(ctr), _lib.ptr(T), None, _p(unormed_d)))
        bbox_lo, bbox_hi = kernels.col_extent(unormed_d, n, d)
        # ---- bootstrap (mlfriends.pyx:1017-1070): the draws are the default path's ----------------------------------------
        self._mark('new-layer whitening, box')
        self._draws_due = True
        masks = draw_job.result()
        masks_d = DevArray.from_host(masks.view(np.uint8))
        self._mark('draw + upload of the selection masks')
        r2s = np.empty(nbootstraps)
        skipped = np.empty(nbootstraps, dtype=np.uint8)
        _lib.check(L.mlf_maxradiussq_bootstrap(_p(unormed_d), n, d, _p(masks_d), nbootstraps, _lib.ptr(r2s), _lib.ptr(skipped)))
        self._mark('bootstrapped radius')
        use = skipped == 0
        maxd = maxf = 0.0
        if use.any():
            maxd = float(r2s[use].max())
            sel_d = masks_d if use.all() else DevArray.from_host(np.ascontiguousarray(masks[use]).view(np.uint8))
            f = np.empty(int(use.sum()))
            _lib.check(L.mlf_bootstrap_factor(_p(U), n, d, _p(sel_d), len(f), float(d + 2), _lib.ptr(f)))
            if not np.isfinite(f).all():
                raise np.linalg.LinAlgError("Singular matrix")
            if not (f > 0).all():
                raise np.linalg.LinAlgError("Distances are not positive")
            maxf = float(f.max())
        if not (maxd > 0 and maxf > 0 and np.isfinite(maxd) and np.isfinite(maxf)):
            raise np.linalg.LinAlgError("compute_enlargement failed")
        self._mark('bootstrapped enlargement')
        # ---- the region object, attributes as the constructor + create_ellipsoid leave them -----------------------------
        region = regions.MLFriends.__new__(regions.MLFriends)
        region.u = active_u                                            # private write-counted copy, as always
        region.device_rng = None
        region._dev = regions._DeviceState()
        region.transformLayer = nxt
        region.unormed = unormed_d.to_host(np.float64, (n, d))
        region.bbox_lo = bbox_lo
        region.bbox_hi = bbox_hi
        region.maxradiussq = maxd
        region.enlarge = maxf
        region.sampling_methods = [region.sample_from_transformed_boundingbox, region.sample_from_boundingbox,
                                   region.sample_from_points, region.sample_from_wrapping_ellipsoid]
        region.current_sampling_method = region.sample_from_boundingbox
        region.vol_prefactor = regions.vol_prefactor(d)
        precision, axes = ell_job.result()                           # raises what the LAPACK calls raised
        region.ellipsoid_center = ctr.copy()                          # mean of the live points: the layer's centre (no wraps)
        region.ellipsoid_invcov = precision
        region.ellipsoid_cov = ecov
        (region.ellipsoid_axlens, region.ellipsoid_axes, region.ellipsoid_axes_T, region.ellipsoid_inv_axlens,
         region.ellipsoid_inv_axes) = axes
        self._mark('region object, downloads, ellipsoid LAPACK')
        # ---- membership of the live points: region state set from the device copy, test on the device copy -----------------
        state = region._dev
        handle = state.handle = kernels.DeviceRegion()
        handle.set_from_device(U, n, d, 0, nxt.ctr, nxt.T, None, region.ellipsoid_center, precision, maxf, maxd,
                               use_scan=True, live_space=1, live_amax=amax)
        state.adopt(region, True)
        self._mark('region state on the device')
        mask_d = DevArray(n)
        handle.inside_dev(U.data_ptr(), n, mask_d.data_ptr(), 0)
        contains_live = bool(mask_d.to_host(np.uint8, (n,)).all())
        self._mark('membership of the live points')
        return nxt, region, contains_live
