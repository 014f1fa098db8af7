"""CPU oracle for the population step-sampler state machine -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end of ``oracle/stepfuncs_oracle.c`` with the call signatures of the reference's
``ultranest.stepfuncs`` / ``ultranest.popstepsampler`` functions (paths relative to
/root/reference), so the golden tests read like the reference's own.

Parity status: PINNED against vectors recorded from the real reference (group g9 of
tests/golden/make_golden.py, checked by tests/test_stepfuncs_golden.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = os.path.join(_HERE, "libstepfuncs_oracle.so")
_lib = None
int_dtype = np.int64


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "stepfuncs_oracle.c")
        if not os.path.exists(_LIBNAME) or os.path.getmtime(_LIBNAME) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libstepfuncs_oracle.so"])
        L = ctypes.CDLL(_LIBNAME)
        vp, sz, dbl = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double
        L.orc_within_unit_cube.argtypes = [vp, sz, sz, vp]
        L.orc_evolve_prepare.argtypes = [vp, vp, sz, vp, vp]
        L.orc_evolve_propose.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, sz, vp]
        L.orc_bisect_draw.argtypes = [vp, vp, vp, vp, sz, vp]
        L.orc_evolve_update.argtypes = [vp, vp, dbl, vp, vp, vp, vp, vp, vp, vp, vp, sz]
        L.orc_step_back.argtypes = [dbl, vp, sz, sz, vp, vp]
        L.orc_step_back.restype = sz
        L.orc_unitcube_line_intersection.argtypes = [vp, vp, sz, sz, vp, vp]
        L.orc_update_vectorised_slice_sampler.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, dbl, dbl, vp, vp, vp,
                                                          sz, sz, sz]
        L.orc_update_vectorised_slice_sampler.restype = ctypes.c_int64
        L.orc_row_dist2.argtypes = [vp, vp, sz, sz, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _flags(a):
    """bool / uint8 array as a contiguous uint8 view-or-copy"""
    return np.ascontiguousarray(a).astype(np.uint8, copy=False)


def within_unit_cube(u):
    """stepfuncs.pyx:36-51"""
    u = _f(u)
    out = np.empty(len(u), dtype=np.uint8)
    lib().orc_within_unit_cube(_p(u), u.shape[0], u.shape[1], _p(out))
    return out.astype(bool)


def evolve_prepare(searching_left, searching_right):
    """stepfuncs.pyx:73-95"""
    sl, sr = _flags(searching_left), _flags(searching_right)
    a, b = np.empty(len(sl), dtype=np.uint8), np.empty(len(sl), dtype=np.uint8)
    lib().orc_evolve_prepare(_p(sl), _p(sr), len(sl), _p(a), _p(b))
    return a.astype(bool), b.astype(bool)


def evolve_update(acceptable, Lnew, Lmin, search_right, bisecting, currentt, current_left, current_right,
                  searching_left, searching_right, success):
    """stepfuncs.pyx:99-183; writes the same arrays in place (they must be contiguous float64 /
    bool arrays, as in the reference's typed signature)."""
    acc, sright, bis = _flags(acceptable), _flags(search_right), _flags(bisecting)
    Lnew = _f(Lnew)
    for a in (currentt, current_left, current_right):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    sl, sr, su = searching_left.view(np.uint8), searching_right.view(np.uint8), success.view(np.uint8)
    lib().orc_evolve_update(_p(acc), _p(Lnew), float(Lmin), _p(sright), _p(bis), _p(currentt), _p(current_left),
                            _p(current_right), _p(sl), _p(sr), _p(su), len(acc))


def evolve(transform, loglike, Lmin, currentu, currentL, currentt, currentv, current_left, current_right,
           searching_left, searching_right):
    """stepfuncs.pyx:189-282 with the same np.random consumption (one uniform per bisecting
    walker) and the same return structure."""
    search_right, bisecting = evolve_prepare(searching_left, searching_right)
    nbis = int(bisecting.sum())
    draws = np.random.random_sample(nbis)
    lib().orc_bisect_draw(_p(current_left), _p(current_right), _p(_flags(bisecting)), _p(draws), len(currentt),
                          _p(currentt))
    unew = currentu
    src = currentu.copy()
    lib().orc_evolve_propose(_p(src), _p(_f(currentv)), _p(current_left), _p(current_right),
                             _p(_flags(searching_left)), _p(_flags(searching_right)), _p(currentt),
                             unew.shape[0], unew.shape[1], _p(unew))
    acceptable = within_unit_cube(unew)
    nc = 0
    if acceptable.any():
        pnew = transform(unew[acceptable, :])
        Lnew = loglike(pnew)
        nc += len(pnew)
    else:
        pnew = np.empty((0, 1))
        Lnew = np.empty(0)
    success = np.zeros_like(searching_left)
    evolve_update(acceptable, Lnew, Lmin, search_right, bisecting, currentt, current_left, current_right,
                  searching_left, searching_right, success)
    return ((currentt, currentv, current_left, current_right, searching_left, searching_right),
            (success, unew[success, :], pnew[success[acceptable], :], Lnew[success[acceptable]]), nc)


def step_back(Lmin, allL, generation, currentt):
    """stepfuncs.pyx:285-334 (in place)"""
    assert allL.flags.c_contiguous and generation.dtype == np.int64
    return lib().orc_step_back(float(Lmin), _p(allL), allL.shape[0], allL.shape[1], _p(generation), _p(currentt))


def unitcube_line_intersection(ray_origin, ray_direction):
    """popstepsampler.py:26-61"""
    o, v = _f(ray_origin), _f(ray_direction)
    lo, hi = np.empty(len(o)), np.empty(len(o))
    lib().orc_unitcube_line_intersection(_p(o), _p(v), o.shape[0], o.shape[1], _p(lo), _p(hi))
    return lo, hi


def update_vectorised_slice_sampler(t, tleft, tright, proposed_L, proposed_u, proposed_p, worker_running, status,
                                    Likelihood_threshold, shrink_factor, allu, allL, allp, popsize):
    """stepfuncs.pyx:537-630 (arrays updated in place and returned, like the reference)"""
    discarded = lib().orc_update_vectorised_slice_sampler(
        _p(_f(t)), _p(tleft), _p(tright), _p(_f(proposed_L)), _p(_f(proposed_u)), _p(_f(proposed_p)),
        _p(worker_running), _p(status), float(Likelihood_threshold), float(shrink_factor), _p(allu), _p(allL),
        _p(allp), int(popsize), allu.shape[1], allp.shape[1])
    return tleft, tright, worker_running, status, allu, allL, allp, int(discarded)


def row_dist2(a, b):
    a, b = _f(a), _f(b)
    out = np.empty(len(a))
    lib().orc_row_dist2(_p(a), _p(b), a.shape[0], a.shape[1], _p(out))
    return out
