"""Vectorized step-sampler helpers with the API of the reference's ``ultranest.stepfuncs``
(reference ultranest/stepfuncs.pyx), the walker arithmetic running in HIP kernels.

Drop-in surface (same names, argument order, in-place behaviour and ``np.random`` consumption):
``within_unit_cube``, ``evolve_prepare``, ``evolve_update``, ``evolve``, ``step_back``,
``update_vectorised_slice_sampler`` and the seven ``generate_*direction`` functions.  With the
same global numpy seed ``evolve`` returns bit-identical arrays (tests/test_stepfuncs_golden.py).

These host-array forms pay one PCIe round trip per call; the production path is
``ultranest_amd.popstepsampler.PopulationSliceSampler``, which keeps the whole walker population
in HBM and runs the same device functions on it.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, f64, ptr

int_dtype = np.int64


def _flags(a):
    """numpy bool / uint8 -> contiguous uint8 array sharing memory when possible"""
    a = np.asarray(a)
    if a.dtype == np.bool_ and a.flags.c_contiguous:
        return a.view(np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _inplace(a, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
        raise ValueError("%s must be a C-contiguous float64 array (it is updated in place)" % name)
    return a


def _inplace_flags(a, name):
    if not (isinstance(a, np.ndarray) and a.dtype in (np.bool_, np.uint8) and a.flags.c_contiguous):
        raise ValueError("%s must be a C-contiguous bool array (it is updated in place)" % name)
    return a.view(np.uint8)


def within_unit_cube(u):
    """For each row, whether all coordinates lie strictly between 0 and 1
    (reference stepfuncs.pyx:36-51)."""
    u = f64(u)
    if u.ndim != 2:
        raise ValueError("u must be (npoints, ndim)")
    out = np.empty(u.shape[0], dtype=np.uint8)
    check(_lib.lib().mlf_within_unit_cube(ptr(u), u.shape[0], u.shape[1], ptr(out)))
    return out.view(np.bool_)


def evolve_prepare(searching_left, searching_right):
    """(search_right, bisecting) selectors of the three exclusive walker states
    (reference stepfuncs.pyx:73-95).  Two boolean operations: stays on the host."""
    sl = np.asarray(searching_left, dtype=bool)
    sr = np.asarray(searching_right, dtype=bool)
    return np.logical_and(~sl, sr), ~np.logical_or(sl, sr)


def evolve_update(acceptable, Lnew, Lmin, search_right, bisecting, currentt, current_left, current_right,
                  searching_left, searching_right, success):
    """Slice-sampling state update with stepping out by doubling (reference stepfuncs.pyx:99-183).
    `Lnew` holds one value per acceptable walker; `currentt`, `current_left`, `current_right`,
    `searching_left`, `searching_right` and `success` are written in place.  `search_right` and
    `bisecting` are implied by the searching flags (evolve_prepare) and only checked."""
    acc = _flags(acceptable)
    n = len(acc)
    Lnew = f64(Lnew)
    if int(acc.sum()) != len(Lnew):
        raise ValueError("Lnew must have one entry per acceptable walker")
    sl, sr = _inplace_flags(searching_left, "searching_left"), _inplace_flags(searching_right, "searching_right")
    su = _inplace_flags(success, "success")
    want_right, want_bis = evolve_prepare(sl.view(np.bool_), sr.view(np.bool_))
    if not (np.array_equal(want_right, np.asarray(search_right, dtype=bool))
            and np.array_equal(want_bis, np.asarray(bisecting, dtype=bool))):
        raise ValueError("search_right / bisecting do not match the searching flags")
    Lfull = np.zeros(n)
    Lfull[acc.view(np.bool_)] = Lnew
    check(_lib.lib().mlf_evolve_update(ptr(acc), ptr(Lfull), float(Lmin), ptr(_inplace(currentt, "currentt")),
                                       ptr(_inplace(current_left, "current_left")),
                                       ptr(_inplace(current_right, "current_right")), ptr(sl), ptr(sr), ptr(su), n))


_PNEW_EMPTY = np.empty((0, 1))
_LNEW_EMPTY = np.empty(0)


def evolve(transform, loglike, Lmin, currentu, currentL, currentt, currentv, current_left, current_right,
           searching_left, searching_right):
    """Advance every walker by one likelihood evaluation (reference stepfuncs.pyx:189-282).

    Consumes one ``np.random`` uniform per bisecting walker, like the reference; writes
    `currentu` (it becomes the proposed points), `currentt`, the brackets and the searching flags
    in place and returns ``((currentt, currentv, current_left, current_right, searching_left,
    searching_right), (success, unew, pnew, Lnew), nc)``."""
    del currentL
    search_right, bisecting = evolve_prepare(searching_left, searching_right)
    n, ndim = currentu.shape
    draws = np.zeros(n)
    draws[bisecting] = np.random.random_sample(int(bisecting.sum()))
    unew = _inplace(currentu, "currentu")
    proposed = np.empty_like(unew)
    acceptable = np.empty(n, dtype=np.uint8)
    check(_lib.lib().mlf_evolve_propose(
        ptr(unew), ptr(f64(currentv)), ptr(_inplace(current_left, "current_left")),
        ptr(_inplace(current_right, "current_right")), ptr(_flags(searching_left)), ptr(_flags(searching_right)),
        ptr(draws), ptr(_inplace(currentt, "currentt")), n, ndim, ptr(proposed), ptr(acceptable)))
    unew[...] = proposed
    acceptable = acceptable.view(np.bool_)
    nc = 0
    if acceptable.any():
        pnew = transform(unew[acceptable, :])
        Lnew = loglike(pnew)
        nc += len(pnew)
    else:
        pnew, Lnew = _PNEW_EMPTY, _LNEW_EMPTY
    success = np.zeros_like(searching_left)
    evolve_update(acceptable, Lnew, Lmin, search_right, bisecting, currentt, current_left, current_right,
                  searching_left, searching_right, success)
    took = success[acceptable]
    return ((currentt, currentv, current_left, current_right, searching_left, searching_right),
            (success, unew[success, :], pnew[took, :], Lnew[took]), nc)


def step_back(Lmin, allL, generation, currentt, log=False):
    """Unwind every chain until none of its recorded likelihoods is below `Lmin`
    (reference stepfuncs.pyx:285-334); `allL`, `generation`, `currentt` are updated in place."""
    del log
    if not (isinstance(generation, np.ndarray) and generation.dtype == np.int64 and generation.flags.c_contiguous):
        raise ValueError("generation must be a C-contiguous int64 array")
    _inplace(allL, "allL")
    check(_lib.lib().mlf_step_back(float(Lmin), ptr(allL), allL.shape[0], allL.shape[1], ptr(generation),
                                   ptr(_inplace(currentt, "currentt"))))


def update_vectorised_slice_sampler(t, tleft, tright, proposed_L, proposed_u, proposed_p, worker_running, status,
                                    Likelihood_threshold, shrink_factor, allu, allL, allp, popsize):
    """Shrink the slices of PopulationSimpleSliceSampler with one batch of evaluated workers and
    deal the free workers to the unfinished points (reference stepfuncs.pyx:537-630).  Arrays are
    updated in place and returned; the last element is the number of discarded evaluations."""
    for a, name in ((tleft, "tleft"), (tright, "tright"), (allu, "allu"), (allL, "allL"), (allp, "allp")):
        _inplace(a, name)
    for a, name in ((worker_running, "worker_running"), (status, "status")):
        if not (isinstance(a, np.ndarray) and a.dtype == np.int64 and a.flags.c_contiguous):
            raise ValueError("%s must be a C-contiguous int64 array" % name)
    popsize = int(popsize)
    discarded = ctypes.c_int64(0)
    check(_lib.lib().mlf_update_vectorised_slice_sampler(
        ptr(f64(t)), ptr(tleft), ptr(tright), ptr(f64(proposed_L)), ptr(f64(proposed_u)), ptr(f64(proposed_p)),
        ptr(worker_running), ptr(status), float(Likelihood_threshold), float(shrink_factor), ptr(allu), ptr(allL),
        ptr(allp), popsize, allu.shape[1], allp.shape[1], ctypes.byref(discarded)))
    return tleft, tright, worker_running, status, allu, allL, allp, int(discarded.value)


def unitcube_line_intersection(ray_origin, ray_direction):
    """(tleft, tright): where the lines origin + t*direction leave the unit cube
    (reference popstepsampler.py:26-61); components with zero direction are ignored."""
    o, v = f64(ray_origin), f64(ray_direction)
    assert (o >= 0).all(), o
    assert (o <= 1).all(), o
    assert ((v**2).sum()**0.5 > 1e-200).all(), v
    lo, hi = np.empty(o.shape[0]), np.empty(o.shape[0])
    check(_lib.lib().mlf_unitcube_line_intersection(ptr(o), ptr(v), o.shape[0], o.shape[1], ptr(lo), ptr(hi)))
    return lo, hi


def row_dist2(a, b):
    """Squared Euclidean distance between corresponding rows."""
    a, b = f64(a), f64(b)
    out = np.empty(a.shape[0])
    check(_lib.lib().mlf_row_dist2(ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(out)))
    return out


# ---- slice direction proposals ------------------------------------------------------------------
# Host numpy with the reference's np.random call order (seeded runs draw the same directions).
# `device_kind` names the equivalent Philox generator of the resident sampler
# (include/mlfriends_hip.h, mlf_walkers_brackets_philox); `needs_points` = the starting points'
# values matter (none of the built-in proposals look at them).

def _one_hot(nsamples, ndim, axis, value):
    """Rows of zeros with `value` at column axis[i].  The reference passes the value through a C
    ``float`` argument (stepfuncs.pyx:337-346), i.e. rounds it to single precision first."""
    v = np.zeros((nsamples, ndim))
    v[np.arange(nsamples), axis] = float(np.float32(value))
    return v


def generate_cube_oriented_direction(ui, region, scale=1):
    """Random unit-cube axis, length `scale` (reference stepfuncs.pyx:348-370)."""
    nsamples, ndim = ui.shape
    axis = np.random.randint(ndim, size=nsamples, dtype=int_dtype)
    return _one_hot(nsamples, ndim, axis, scale)


def generate_cube_oriented_direction_scaled(ui, region, scale=1):
    """Random unit-cube axis scaled by the live points' spread on it (reference :373-399)."""
    nsamples, ndim = ui.shape
    spread = region.u.std(axis=0)
    axis = np.random.randint(ndim, size=nsamples, dtype=int_dtype)
    v = _one_hot(nsamples, ndim, axis, scale)
    v *= spread[axis].reshape((-1, 1))
    return v


def generate_random_direction(ui, region, scale=1):
    """Isotropic direction of length `scale` in unit-cube space (reference :401-422)."""
    nsamples, ndim = ui.shape
    v = np.random.normal(size=(nsamples, ndim))
    v *= scale / np.linalg.norm(v, axis=1).reshape((nsamples, 1))
    return v


def generate_region_oriented_direction(ui, region, scale=1):
    """One of the region's principal axes, `scale` long in whitened space (reference :425-450)."""
    nsamples, ndim = ui.shape
    axis = np.random.randint(ndim, size=nsamples, dtype=int_dtype)
    return region.transformLayer.axes[axis] * scale


def generate_region_random_direction(ui, region, scale=1):
    """Isotropic in whitened space, mapped to the unit cube by the layer axes (reference :453-478)."""
    nsamples, ndim = ui.shape
    w = np.random.normal(size=(nsamples, ndim))
    w *= scale / np.linalg.norm(w, axis=1).reshape((nsamples, 1))
    return np.einsum('ij,kj->ki', region.transformLayer.axes, w)


def generate_differential_direction(ui, region, scale=1):
    """Difference of two distinct random live points (reference :480-508)."""
    nsamples, ndim = ui.shape
    nlive = region.u.shape[0]
    first = np.random.randint(nlive, size=nsamples, dtype=int_dtype)
    second = np.random.randint(nlive - 1, size=nsamples, dtype=int_dtype)
    second[second >= first] += 1
    return (region.u[first, :] - region.u[second, :]) * scale


def generate_mixture_random_direction(ui, region, scale=1):
    """Coin flip between the differential and the region-oriented proposal (reference :512-535)."""
    nsamples, ndim = ui.shape
    far = generate_differential_direction(ui, region, scale=scale)
    stiff = generate_region_oriented_direction(ui, region, scale=scale)
    return np.where(np.random.uniform(size=nsamples).reshape((-1, 1)) < 0.5, far, stiff)


for _kind, _fn in enumerate([generate_cube_oriented_direction, generate_cube_oriented_direction_scaled,
                             generate_random_direction, generate_region_oriented_direction,
                             generate_region_random_direction, generate_differential_direction,
                             generate_mixture_random_direction]):
    _fn.device_kind = _kind
    _fn.needs_points = False
