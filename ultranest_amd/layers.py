"""Whitening ("transform") layers with the API of the reference's
``ScalingLayer`` / ``AffineLayer`` / ``MaxPrincipleGapAffineLayer`` / ``LocalAffineLayer``
(reference ultranest/mlfriends.pyx:479-850) and the friends-of-friends clustering
``update_clusters`` (:275-384).

Division of labour: everything that is O(d^2)/O(d^3) on a d x d matrix (covariance, eigh, inv,
slogdet) is host numpy/LAPACK, exactly as in the reference; everything
that touches all point PAIRS (the neighbour scans inside clustering, ``subtract_nearby``) runs on
the MI355X through ``ultranest_amd.kernels``.  ``transform`` of the N live points stays a numpy
``dot`` so that ``region.unormed`` is bit-identical to the reference's; proposal batches are
transformed on the device inside the region's ``inside`` pipeline.
"""
import functools
import threading

import numpy as np

from . import kernels

int_dtype = kernels.int_dtype

# ---- host linear algebra of a rebuild runs on ONE BLAS thread ------------------------------------------
# The d x d problems of a region rebuild (covariance of 4000 x 50 points, eigh / inv of 50 x 50) are a millisecond of
# work.  OpenBLAS wakes its whole pool for the products among them; on a many-core host that pool, spinning, pushed
# the calling thread off its core: every few rebuilds one took 60-70 ms instead of 7 (scripts/rebuild_breakdown.py).
_blas = {"controller": None, "tried": False}


def _blas_controller():
    if not _blas["tried"]:
        _blas["tried"] = True
        try:
            from threadpoolctl import ThreadpoolController
            _blas["controller"] = ThreadpoolController()
        except Exception:      # threadpoolctl missing or a BLAS it cannot drive: leave the pool alone
            _blas["controller"] = None
    return _blas["controller"]


# The controller is made HERE, while the module is imported: threadpoolctl finds the BLAS libraries with dl_iterate_phdr
# and a Python callback, i.e. it needs the interpreter lock while it holds the loader's lock -- made lazily on the host worker
# thread, it met a dlopen (ctypes.CDLL) on the main thread, which holds the interpreter lock while it waits for the
# loader's: a deadlock, seen as a two-rank CPU test that never returned once in a few runs.
_blas_controller()

_blas_lock = threading.Lock()
_blas_scope = {"depth": 0, "limit": None}


def single_blas_thread(fn):
    """Decorator: run `fn` with the BLAS pool limited to one thread (a no-op without threadpoolctl).

    threadpoolctl's limit is a process-global save / restore.  The scopes of this module overlap across TWO threads (the
    calling thread and the host worker that takes `ellipsoid_parts` during the bootstrap), and not nested: the limit is
    therefore taken by whoever enters first and given back by whoever leaves last (a counter under a lock) -- an inner
    scope that restored "its" saved state while another thread is still inside, or in the wrong order, would leave the
    pool at one thread for the rest of the process, the user's likelihood included."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        ctl = _blas_controller()
        if ctl is None:
            return fn(*args, **kwargs)
        with _blas_lock:
            if _blas_scope["depth"] == 0:
                limit = ctl.limit(limits=1, user_api="blas")
                limit.__enter__()
                _blas_scope["limit"] = limit
            _blas_scope["depth"] += 1
        try:
            return fn(*args, **kwargs)
        finally:
            with _blas_lock:
                _blas_scope["depth"] -= 1
                if _blas_scope["depth"] == 0:
                    limit, _blas_scope["limit"] = _blas_scope["limit"], None
                    limit.__exit__(None, None, None)
    return wrapper


_HOST_WORKER = (None, None)      # (executor, pid of the process that created it)


def host_worker():
    """The one worker thread that takes host numpy / LAPACK work while the calling thread waits for the GPU inside a
    C-ABI call (ctypes releases the interpreter lock there).  A forked child starts its own: the parent's thread does not
    exist there, and an executor that still counts it would never run a job."""
    global _HOST_WORKER
    import os
    if _HOST_WORKER[0] is None or _HOST_WORKER[1] != os.getpid():
        from concurrent.futures import ThreadPoolExecutor
        _HOST_WORKER = (ThreadPoolExecutor(max_workers=1), os.getpid())
    return _HOST_WORKER[0]


def _call_with_errstate(errstate, fn, *args):
    with np.errstate(**errstate):      # numpy keeps the error state per thread: the worker takes the caller's
        return fn(*args)


# the all-pairs pass keeps one hit bit per pair (n^2 / 8 bytes on the device and in pinned host memory: 2 MB at n = 4000,
# 128 MB at 32768); larger live sets take one neighbour scan per growth round instead
_ADJACENCY_MAX_POINTS = 32768


def _labels_by_growth_rounds(tpoints, maxradiussq, clusterids):
    """The reference's loop as it stands (mlfriends.pyx:275-331): each growth round is one GPU neighbour scan (K1) of the
    current cluster's members against all still unlabelled points."""
    npts = len(tpoints)
    previous = np.zeros(npts, dtype=int_dtype) if clusterids is None else np.asarray(clusterids)[:npts]
    labels = np.zeros(npts, dtype=int_dtype)

    def seed_for(cid, default):
        carried = np.flatnonzero(previous == cid)
        return carried[0] if len(carried) else default

    current = 1
    labels[seed_for(current, 0)] = current
    while True:
        unlabelled = np.flatnonzero(labels == 0)
        if len(unlabelled) == 0:
            break
        hits = np.empty(len(unlabelled), dtype=int_dtype)
        kernels.find_nearby(tpoints[labels == current], tpoints[unlabelled], maxradiussq, hits)
        joined = unlabelled[hits >= 0]
        if len(joined):
            labels[joined] = current
        else:
            current += 1
            labels[seed_for(current, unlabelled[0])] = current
    return labels


def update_clusters(upoints, tpoints, maxradiussq, clusterids=None):
    """Friends-of-friends clustering of `tpoints` with linking length sqrt(`maxradiussq`).

    Returns ``(nclusters, clusterids, overlapped_upoints)`` like the reference
    (mlfriends.pyx:348-384): labels start at 1; previous `clusterids` seed the labels so that
    clusters keep their identity; ``overlapped_upoints`` are `upoints` with their cluster mean
    removed (a singleton is centred on the global mean instead).

    The labels come from ONE all-pairs pass of exact distances on the device (hit bits, K1's arithmetic) and the
    reference's growth rounds replayed on the bit rows (kernels.cluster_labels -> mlf_cluster_labels): the same partition
    and the same numbering as one neighbour scan per growth round, without the rounds' host copies and launches.
    """
    upoints = np.asarray(upoints)
    tpoints = np.asarray(tpoints)
    if upoints.shape != tpoints.shape:
        raise AssertionError(('different shapes of points', upoints.shape, tpoints.shape))
    if len(tpoints) == 0:
        raise IndexError('update_clusters: no points')   # the reference seeds clusterids[0] (:287)
    if len(tpoints) <= _ADJACENCY_MAX_POINTS:
        _, labels = kernels.cluster_labels(tpoints, maxradiussq, clusterids)
    else:
        labels = _labels_by_growth_rounds(tpoints, maxradiussq, clusterids)

    assert (labels > 0).all()
    present = np.unique(labels)
    if len(present) == 1:
        overlapped = upoints
    else:
        overlapped = np.empty_like(upoints)
        everyone = None
        for cid in present:
            members = labels == cid
            group = upoints[members]
            if len(group) > 1:
                centre = group.mean(axis=0)
            else:
                if everyone is None:
                    everyone = upoints.mean(axis=0)
                centre = everyone
            overlapped[members] = group - centre.reshape((1, -1))
    return len(present), labels, overlapped


class ScalingLayer(object):
    """Per-axis shift and scale (reference mlfriends.pyx:479-620)."""

    def __init__(self, mean=0, std=1, nclusters=1, wrapped_dims=[], clusterids=None):
        self.mean = mean
        self.std = std
        self._init_common(nclusters, wrapped_dims, clusterids)

    def _init_common(self, nclusters, wrapped_dims, clusterids):
        self.nclusters = nclusters
        self.wrapped_dims = wrapped_dims
        self.has_wraps = len(wrapped_dims) > 0
        self.clusterids = clusterids

    # ---- circular parameters ------------------------------------------------------------
    def optimize_wrap(self, points):
        """For each circular axis find the widest empty interval (including the cube borders
        as sentinels) and remember its midpoint as the place to cut the circle."""
        if not self.has_wraps:
            return
        self.wrap_cuts = []
        for axis in self.wrapped_dims:
            vals = np.sort(np.concatenate(([0.0], points[:, axis], [1.0])))
            gaps = np.diff(vals)
            widest = gaps.argmax()
            self.wrap_cuts.append((vals[widest] + vals[widest + 1]) / 2.)

    def wrap(self, points):
        """Rotate circular axes so that the cut sits at the cube border."""
        if not self.has_wraps:
            return points
        wpoints = points.copy().reshape((-1, points.shape[-1]))
        for axis, cut in zip(self.wrapped_dims, self.wrap_cuts):
            wpoints[:, axis] = np.fmod(wpoints[:, axis] + (1 - cut), 1)
        return wpoints

    def unwrap(self, wpoints):
        """Inverse of :meth:`wrap`."""
        if not self.has_wraps:
            return wpoints
        points = wpoints.copy().reshape((-1, wpoints.shape[-1]))
        for axis, cut in zip(self.wrapped_dims, self.wrap_cuts):
            points[:, axis] = np.fmod(points[:, axis] + cut, 1)
        return points

    def wrap_shift_vector(self, ndim):
        """Device form of the wraps: (1 - cut) per wrapped axis, NaN elsewhere; None if no wraps."""
        if not self.has_wraps:
            return None
        shift = np.full(ndim, np.nan)
        for axis, cut in zip(self.wrapped_dims, self.wrap_cuts):
            shift[axis] = 1 - cut
        return shift

    # ---- learning ------------------------------------------------------------------------
    def set_clusterids(self, clusterids=None, npoints=None):
        if clusterids is None and self.clusterids is None and npoints is not None:
            clusterids = np.ones(npoints, dtype=int_dtype)
        if clusterids is not None:
            self.clusterids = clusterids

    def optimize(self, points, centered_points, clusterids=None, minvol=0.):
        """mean from the (wrapped) points, std from the cluster-centred points."""
        self.optimize_wrap(points)
        self.mean = self.wrap(points).mean(axis=0).reshape((1, -1))
        self.std = centered_points.std(axis=0).reshape((1, -1))
        self.axes = np.diag(self.std[0])
        self.logvolscale = np.sum(np.log(self.std))
        self.set_clusterids(clusterids=clusterids, npoints=len(points))

    def _recluster(self, upoints, maxradiussq):
        uwpoints = self.wrap(upoints)
        nclusters, ids, centred = update_clusters(uwpoints, self.transform(upoints), maxradiussq, self.clusterids)
        return uwpoints, nclusters, ids, centred

    def create_new(self, upoints, maxradiussq, minvol=0.):
        """Next-generation layer learned from this layer's clustering."""
        _, nclusters, ids, centred = self._recluster(upoints, maxradiussq)
        nxt = self.__class__(nclusters=nclusters, wrapped_dims=self.wrapped_dims, clusterids=ids)
        nxt.optimize(upoints, centred)
        return nxt

    # ---- mapping ------------------------------------------------------------------------
    def transform(self, u):
        w = self.wrap(u) if self.has_wraps else u
        return ((w - self.mean) / self.std).reshape(u.shape)

    def untransform(self, ww):
        w = (ww * self.std) + self.mean
        if self.has_wraps:
            return self.unwrap(w).reshape(ww.shape)
        return w.reshape(ww.shape)

    # ---- device description (consumed by regions._DeviceState) -----------------------------
    def device_params(self, ndim):
        """(layer_kind, ctr, T) for mlf_region_set: kind 1 = per-axis (mean, std)."""
        mean = np.broadcast_to(np.asarray(self.mean, dtype=float).reshape(-1), (ndim,)) \
            if np.ndim(self.mean) else np.full(ndim, float(self.mean))
        std = np.broadcast_to(np.asarray(self.std, dtype=float).reshape(-1), (ndim,)) \
            if np.ndim(self.std) else np.full(ndim, float(self.std))
        return 1, mean, std


class AffineLayer(ScalingLayer):
    """Full-covariance whitening (reference mlfriends.pyx:623-752)."""

    def __init__(self, ctr=0, T=1, invT=1, nclusters=1, wrapped_dims=[], clusterids=None):
        self.ctr = ctr
        self.T = T
        self.invT = invT
        self._init_common(nclusters, wrapped_dims, clusterids)

    @single_blas_thread
    def optimize(self, points, centered_points, clusterids=None, minvol=0.):
        """ctr = mean of the wrapped points; cov = sample covariance of `centered_points`
        inflated by (d+2); T = eigvec * eigval**-0.5 with eigenvalues floored at 1e-40 of the
        largest.  A singular covariance escalates as LinAlgError from the explicit inverse."""
        self.optimize_wrap(points)
        self.ctr = np.mean(self.wrap(points), axis=0)
        ndim = len(self.ctr)
        cov = np.cov(centered_points, rowvar=0)
        cov *= (ndim + 2)
        self.cov = cov
        eigval, eigvec = np.linalg.eigh(cov)
        floor = eigval.max() * 1e-40
        eigval[eigval < floor] = floor
        precision = np.linalg.inv(cov)
        self.logvolscale = np.linalg.slogdet(precision)[1] * -0.5
        self.T = eigvec * eigval**-0.5
        self.invT = np.linalg.inv(self.T)
        self.axes = self.invT
        self.set_clusterids(clusterids=clusterids, npoints=len(points))

    def create_new(self, upoints, maxradiussq, minvol=0.):
        _, nclusters, ids, centred = self._recluster(upoints, maxradiussq)
        nxt = self.__class__(nclusters=nclusters, wrapped_dims=self.wrapped_dims, clusterids=ids)
        nxt.optimize(upoints, centred, minvol=minvol)
        return nxt

    def transform(self, u):
        w = self.wrap(u) if self.has_wraps else u
        return np.dot(w - self.ctr, self.T)

    def untransform(self, ww):
        w = np.dot(ww, self.invT) + self.ctr
        if self.has_wraps:
            return self.unwrap(w).reshape(ww.shape)
        return w.reshape(ww.shape)

    def device_params(self, ndim):
        ctr = np.broadcast_to(np.asarray(self.ctr, dtype=float), (ndim,))
        T = np.asarray(self.T, dtype=float)
        if T.ndim != 2:
            T = np.eye(ndim) * float(T)
        return 0, ctr, T


class MaxPrincipleGapAffineLayer(AffineLayer):
    """Affine layer that additionally splits the co-centred points at the widest gap along the
    principal axis before taking the covariance (reference mlfriends.pyx:754-816)."""

    def create_new(self, upoints, maxradiussq, minvol=0.):
        _, nclusters, ids, centred = self._recluster(upoints, maxradiussq)
        cov = np.cov(centred, rowvar=0)
        cov *= (len(self.ctr) + 2)
        _, eigvec = np.linalg.eigh(cov)
        principal = eigvec[:, -1]
        pos = np.dot(centred - centred.mean(axis=0).reshape((1, -1)), principal)
        ordered = np.sort(pos)
        widest = np.argmax(np.diff(ordered))
        split = (ordered[widest] + ordered[widest + 1]) / 2
        left = pos < split
        halved = centred.copy()
        halved[left, :] -= centred[left, :].mean(axis=0)
        halved[~left, :] -= centred[~left, :].mean(axis=0)
        nxt = MaxPrincipleGapAffineLayer(nclusters=nclusters, wrapped_dims=self.wrapped_dims, clusterids=ids)
        nxt.optimize(upoints, halved, minvol=minvol)
        return nxt


class LocalAffineLayer(AffineLayer):
    """Affine layer whose covariance is taken after subtracting from each point the mean of its
    neighbours within the MLFriends radius (reference mlfriends.pyx:819-850).  As in the
    reference the neighbourhood is evaluated on the wrapped u-space points with the t-space
    radius (:844-849) -- this is the default layer of ReactiveNestedSampler (integrator.py:1137).
    The O(N^2 d) neighbourhood pass is the GPU kernel pair behind ``subtract_nearby``."""

    def create_new(self, upoints, maxradiussq, minvol=0.):
        # The reference clusters first and subtracts the neighbour means second (:843-848); the two do not depend on each
        # other and draw no random numbers.  Here the host product of `transform` (0.3 ms at N = 4000, d = 50) runs on
        # the worker thread while this one waits for the GPU's neighbour pass; same calls, same inputs, same results.
        # Round 5: the new layer's own host work -- `optimize`: np.cov, eigh, two inverses, slogdet: 0.8 ms -- needs the
        # neighbour means but not the cluster labels, so it follows `transform` on the worker thread while this one waits for
        # the GPU's all-pairs clustering pass (0.5 ms): the labels are attached afterwards (an error of either side surfaces
        # here, in the reference's order: clustering first).
        uwpoints = self.wrap(upoints)
        errstate = np.geterr()
        job = host_worker().submit(_call_with_errstate, errstate, self.transform, upoints)
        local = kernels.subtract_nearby(uwpoints, maxradiussq)
        nxt = self.__class__(nclusters=1, wrapped_dims=self.wrapped_dims)
        opt_job = host_worker().submit(_call_with_errstate, errstate, nxt.optimize, upoints, local, None, minvol)
        try:
            nclusters, ids, _ = update_clusters(uwpoints, job.result(), maxradiussq, self.clusterids)
        except BaseException:
            opt_job.cancel()
            try:
                opt_job.result()       # never leave the worker running behind an exception of this thread
            except BaseException:
                pass
            raise
        opt_job.result()
        nxt.nclusters = nclusters
        nxt.set_clusterids(clusterids=ids, npoints=len(upoints))
        return nxt
