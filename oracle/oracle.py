"""CPU oracle for the MLFriends hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of ``oracle/mlfriends_oracle.c`` plus numpy restatements of the
reference's host-side stages (bootstrapped enlargement, wrapping ellipsoid,
friends-of-friends clustering, whitening layer).  Every function cites the
reference lines it restates (paths relative to /root/reference).

Parity status: PINNED against golden vectors generated from the real reference
(tests/golden/make_golden.py, checked by tests/test_oracle_golden.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  ``ultranest_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = os.path.join(_HERE, "libmlfriends_oracle.so")

_f64 = np.float64
_i64 = np.int64


def build(force=False):
    """Compile the C restatement with the committed recipe (oracle/Makefile)."""
    src = os.path.join(_HERE, "mlfriends_oracle.c")
    if force or not os.path.exists(_LIBNAME) or os.path.getmtime(_LIBNAME) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIBNAME


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIBNAME)
        dp = ctypes.c_void_p
        sz = ctypes.c_size_t
        L.orc_find_nearby.argtypes = [dp, sz, dp, sz, sz, ctypes.c_double, dp]
        L.orc_count_nearby.argtypes = [dp, sz, dp, sz, sz, ctypes.c_double, dp]
        L.orc_subtract_nearby.argtypes = [dp, sz, sz, ctypes.c_double, dp]
        L.orc_maxradiussq.argtypes = [dp, sz, dp, sz, sz]
        L.orc_maxradiussq.restype = ctypes.c_float
        L.orc_maxradiussq_bootstrap.argtypes = [dp, sz, sz, dp, sz, dp, dp]
        L.orc_mean_pair_distance.argtypes = [dp, sz, sz, dp]
        L.orc_mean_pair_distance.restype = ctypes.c_double
        L.orc_inside_ellipsoid.argtypes = [dp, sz, sz, dp, dp, ctypes.c_double, dp, dp]
        L.orc_affine_transform.argtypes = [dp, sz, sz, dp, dp, dp]
        L.orc_region_inside.argtypes = [dp, sz, sz, dp, sz, dp, dp, dp, dp,
                                        ctypes.c_double, ctypes.c_double, dp]
        L.orc_loglike_gauss.argtypes = [dp, sz, sz, dp, ctypes.c_double, dp]
        L.orc_loglike_eggbox.argtypes = [dp, sz, sz, dp]
        L.orc_loglike_eggbox2.argtypes = [dp, sz, sz, dp]
        L.orc_loglike_rosenbrock.argtypes = [dp, sz, sz, dp]
        _lib = L
    return _lib


def _c(a, dtype=_f64):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------- K rows ----

def find_nearby(apts, bpts, radiussq, out=None):
    """K1, ultranest/mlfriends.pyx:143-183."""
    apts, bpts = _c(apts), _c(bpts)
    if out is None:
        out = np.empty(len(bpts), dtype=_i64)
    d = apts.shape[1] if apts.ndim == 2 else bpts.shape[1]
    lib().orc_find_nearby(_p(apts), len(apts), _p(bpts), len(bpts), d, float(radiussq), _p(out))
    return out


def count_nearby(apts, bpts, radiussq):
    """K2, ultranest/mlfriends.pyx:31-68."""
    apts, bpts = _c(apts), _c(bpts)
    out = np.empty(len(bpts), dtype=_i64)
    lib().orc_count_nearby(_p(apts), len(apts), _p(bpts), len(bpts), apts.shape[1], float(radiussq), _p(out))
    return out


def subtract_nearby(pts, radiussq):
    """K3, ultranest/mlfriends.pyx:73-138."""
    pts = _c(pts)
    out = np.empty_like(pts)
    lib().orc_subtract_nearby(_p(pts), pts.shape[0], pts.shape[1], float(radiussq), _p(out))
    return out


def maxradiussq(apts, bpts):
    """K4, ultranest/mlfriends.pyx:188-224 (returns the float32-rounded value as a Python float)."""
    apts, bpts = _c(apts), _c(bpts)
    return float(lib().orc_maxradiussq(_p(apts), len(apts), _p(bpts), len(bpts), apts.shape[1]))


def maxradiussq_bootstrap(pts, selected):
    """Per-bootstrap K4 over (B, n) selection masks, mlfriends.pyx:1044-1054.

    Returns (maxd[B] float64 already float32-rounded, skipped[B] bool)."""
    pts = _c(pts)
    sel = np.ascontiguousarray(selected, dtype=np.uint8)
    if sel.ndim == 1:
        sel = sel[None, :]
    B = sel.shape[0]
    out = np.empty(B, dtype=_f64)
    skipped = np.empty(B, dtype=np.uint8)
    lib().orc_maxradiussq_bootstrap(_p(pts), pts.shape[0], pts.shape[1], _p(sel), B, _p(out), _p(skipped))
    return out, skipped.astype(bool)


def mean_pair_distance(pts, clusterids):
    """K5, ultranest/mlfriends.pyx:229-270."""
    pts = _c(pts)
    ids = _c(clusterids, _i64)
    return float(lib().orc_mean_pair_distance(_p(pts), pts.shape[0], pts.shape[1], _p(ids)))


def inside_ellipsoid(pts, ctr, invcov, sqradius, return_q=False):
    """H3, ultranest/mlfriends.pyx:882-912."""
    pts, ctr, invcov = _c(pts), _c(ctr), _c(invcov)
    mask = np.empty(len(pts), dtype=np.uint8)
    q = np.empty(len(pts), dtype=_f64)
    lib().orc_inside_ellipsoid(_p(pts), len(pts), pts.shape[1], _p(ctr), _p(invcov), float(sqradius), _p(mask), _p(q))
    return (mask.astype(bool), q) if return_q else mask.astype(bool)


def affine_transform(pts, ctr, T):
    """T1 (no wrapped dims), ultranest/mlfriends.pyx:737-743; FMA-chain restatement (tolerance class)."""
    pts, ctr, T = _c(pts), _c(np.broadcast_to(ctr, (pts.shape[1],))), _c(T)
    out = np.empty_like(pts)
    lib().orc_affine_transform(_p(pts), len(pts), pts.shape[1], _p(ctr), _p(T), _p(out))
    return out


def region_inside(pts, unormed, layer_ctr, layer_T, ell_ctr, ell_invcov, enlarge, radiussq):
    """R3, ultranest/mlfriends.pyx:1186-1211 composed from H3 -> T1 -> K1."""
    pts, unormed = _c(pts), _c(unormed)
    d = pts.shape[1]
    layer_ctr = _c(np.broadcast_to(layer_ctr, (d,)))
    layer_T, ell_ctr, ell_invcov = _c(layer_T), _c(ell_ctr), _c(ell_invcov)
    mask = np.empty(len(pts), dtype=np.uint8)
    lib().orc_region_inside(_p(pts), len(pts), d, _p(unormed), len(unormed), _p(layer_ctr), _p(layer_T),
                            _p(ell_ctr), _p(ell_invcov), float(enlarge), float(radiussq), _p(mask))
    return mask.astype(bool)


# ------------------------------------------------------------ likelihoods ----

def loglike_gauss(params, centers, sigma):
    """L1, docs/gauss.py:25-27."""
    params = _c(params)
    centers = _c(np.broadcast_to(centers, (params.shape[1],)))
    out = np.empty(len(params))
    lib().orc_loglike_gauss(_p(params), params.shape[1], len(params), _p(centers), float(sigma), _p(out))
    return out


def loglike_eggbox(params):
    """L2, examples/testeggbox.py:9-11."""
    params = _c(params)
    out = np.empty(len(params))
    lib().orc_loglike_eggbox(_p(params), params.shape[1], len(params), _p(out))
    return out


def loglike_eggbox2(params):
    """L2', examples/test_PopSliceSampler.py:69-71."""
    params = _c(params)
    out = np.empty(len(params))
    lib().orc_loglike_eggbox2(_p(params), params.shape[1], len(params), _p(out))
    return out


def loglike_rosenbrock(params):
    """L3, examples/testrosenbrock.py:10-13."""
    params = _c(params)
    out = np.empty(len(params))
    lib().orc_loglike_rosenbrock(_p(params), params.shape[1], len(params), _p(out))
    return out


# ------------------------------------------------- numpy host-side stages ----

def draw_bootstrap_masks(rng, n, nbootstraps):
    """Selection masks exactly as the reference draws them: one
    ``rng.randint(N, size=N)`` per bootstrap (mlfriends.pyx:1045-1047)."""
    masks = np.zeros((nbootstraps, n), dtype=bool)
    for b in range(nbootstraps):
        masks[b, rng.randint(n, size=n)] = True
    return masks


def bounding_ellipsoid(x, minvol=0.0):
    """H2, ultranest/mlfriends.pyx:426-476 (+ make_eigvals_positive :389-421)."""
    ndim = x.shape[1]
    ctr = np.mean(x, axis=0)
    cov = np.atleast_2d(np.cov(x - ctr, rowvar=0)) * (ndim + 2)
    if minvol > 0:
        w, v = np.linalg.eigh(cov)
        small = w < max(1e-10, 1e-300 ** (1.0 / ndim))
        if small.any():
            w[small] = (minvol / np.prod(w[~small])) ** (1.0 / small.sum())
            cov = np.dot(np.dot(v, np.diag(w)), np.linalg.inv(v))
    return ctr, cov


def ellipsoid_enlargement(u, mask, minvol=0.0):
    """Per-bootstrap f of R2, ultranest/mlfriends.pyx:1056-1066."""
    ctr, cov = bounding_ellipsoid(u[mask], minvol=minvol)
    a = np.linalg.inv(cov)
    delta = u[~mask] - ctr
    return np.einsum('ij,jk,ik->i', delta, a, delta).max()


def compute_enlargement(u, unormed, masks, minvol=0.0):
    """R2, MLFriends.compute_enlargement (mlfriends.pyx:1017-1070) on pre-drawn masks."""
    maxd = 0.0
    maxf = 0.0
    r, skipped = maxradiussq_bootstrap(unormed, masks)
    for b, m in enumerate(masks):
        if skipped[b]:
            continue
        maxd = max(maxd, r[b])
        f = ellipsoid_enlargement(u, m, minvol)
        if not f > 0:
            raise np.linalg.LinAlgError("Distances are not positive")
        maxf = max(maxf, f)
    return maxd, maxf


def affine_optimize(points, centered_points):
    """T2, AffineLayer.optimize without wraps (mlfriends.pyx:666-710).
    Returns ctr, T, invT, logvolscale, cov."""
    ctr = np.mean(points, axis=0)
    cov = np.cov(centered_points, rowvar=0) * (len(ctr) + 2)
    cov = np.atleast_2d(cov)
    eigval, eigvec = np.linalg.eigh(cov)
    floor = eigval.max() * 1e-40
    eigval[eigval < floor] = floor
    a = np.linalg.inv(cov)
    logvolscale = np.linalg.slogdet(a)[1] * -0.5
    T = eigvec * eigval ** -0.5
    return ctr, T, np.linalg.inv(T), logvolscale, cov


def update_clusters(upoints, tpoints, maxradiussq, clusterids=None):
    """H1, _update_clusters (mlfriends.pyx:275-343): friends-of-friends labels by
    repeated K1 calls (members vs non-members), re-using old ids as seeds."""
    n = len(tpoints)
    if clusterids is None:
        clusterids = np.zeros(n, dtype=_i64)
    old = np.asarray(clusterids)[:n]
    labels = np.zeros(n, dtype=_i64)
    cur = 1
    seed = 0
    hit = np.flatnonzero(old == cur)
    if len(hit):
        seed = hit[0]
    labels[seed] = cur
    while True:
        free = labels == 0
        if not free.any():
            break
        idx = find_nearby(tpoints[labels == cur], tpoints[free], maxradiussq)
        if (idx >= 0).any():
            free_idx = np.flatnonzero(free)
            labels[free_idx[idx >= 0]] = cur
        else:
            cur += 1
            seed = np.flatnonzero(free)[0]
            hit = np.flatnonzero(old == cur)
            if len(hit):
                seed = hit[0]
            labels[seed] = cur
    uniq = np.unique(labels)
    if len(uniq) == 1:
        overlapped = upoints
    else:
        overlapped = np.empty_like(upoints)
        for c in uniq:
            grp = upoints[labels == c]
            mean = grp.mean(axis=0) if len(grp) > 1 else upoints.mean(axis=0)
            overlapped[labels == c] = grp - mean.reshape((1, -1))
    return len(uniq), labels, overlapped
