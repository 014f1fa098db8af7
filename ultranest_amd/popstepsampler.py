"""Vectorized population step samplers with the API of the reference's
``ultranest.popstepsampler`` (reference ultranest/popstepsampler.py), built on the device-resident
walker state machine (csrc/mlf_walk.hip).

``PopulationSliceSampler`` is the stepsampler object the reference's driver calls once per
iteration (``sampler.stepsampler = PopulationSliceSampler(...)``; integrator.py:1839-1950 calls
``__next__(region, Lmin, us, Ls, transform, loglike, ...)`` until it returns a point).  Here the
whole population -- chain history ``allu/allL``, slice coordinate, direction, brackets, flags,
transformed point -- lives in HBM; one call uploads only what the host decided (start rows,
directions or random numbers) and downloads one small record.

Two random sources:
  * default: the reference's ``np.random`` stream, drawn on the host in the reference's order
    (start picks, directions, one uniform per bisecting walker).  A seeded run then returns the
    same points as the reference, call by call (tests/test_popstepsampler.py).
  * ``device_rng=DeviceRNG(seed)``: directions and bisection draws come from Philox on the
    device (nothing but the start rows is uploaded); statistically equivalent, not draw-identical.

Likelihood: any ``loglike(p) -> L`` / ``transform(u) -> p`` numpy callbacks (the accepted
proposals then make one round trip), or the device likelihoods of ``ultranest_amd.likelihoods``
(objects with ``device_spec``), which are evaluated in place on the device.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, f64, ptr
from .regions import DeviceRNG
from .stepfuncs import (evolve, generate_cube_oriented_direction,  # noqa: F401
                        generate_cube_oriented_direction_scaled, generate_differential_direction,
                        generate_mixture_random_direction, generate_random_direction,
                        generate_region_oriented_direction, generate_region_random_direction, int_dtype,
                        row_dist2, step_back, unitcube_line_intersection, update_vectorised_slice_sampler)


def diagnose_move_distances(region, ustart, ufinal):
    """Whitened-space distance travelled by each walker against the MLFriends radius
    (reference popstepsampler.py:64-94): ``(far_enough, [distance, radius])``."""
    assert ustart.shape == ufinal.shape, (ustart.shape, ufinal.shape)
    tstart = region.transformLayer.transform(ustart)
    tfinal = region.transformLayer.transform(ufinal)
    d2 = row_dist2(tstart, tfinal) if len(tstart) else np.empty(0)
    return d2 > region.maxradiussq, [d2**0.5, region.maxradiussq**0.5]


def slice_limit_to_unitcube(tleft, tright):
    """Slice limits = intersection with the unit cube (reference popstepsampler.py:700-722)."""
    return np.array(tleft, dtype=float), np.array(tright, dtype=float)


def slice_limit_to_scale(tleft, tright):
    """Slice limits clipped to [-1, +1] (reference popstepsampler.py:725-744)."""
    tleft, tright = np.asarray(tleft, dtype=float), np.asarray(tright, dtype=float)
    return np.fmax(tleft, -1.0), np.fmin(tright, 1.0)


class GenericPopulationSampler(object):
    """Step statistics shared by the population samplers (reference popstepsampler.py:97-194;
    the matplotlib plots of the reference are outside this package's scope)."""

    logstat_labels = ['accept_rate', 'efficiency', 'scale', 'far_enough', 'mean_rel_jump']

    def _acceptance_weighted(self, column, transform=None):
        """average of one logstat column over the recorded steps, weighted by their acceptance rates"""
        if not self.logstat:
            return np.nan
        table = np.asarray(self.logstat, dtype=float)
        values = table[:, column] if transform is None else transform(table[:, column])
        return np.average(values, weights=table[:, 0])

    @property
    def mean_jump_distance(self):
        """Geometric mean jump distance (in units of the MLFriends radius), weighted by acceptance rate."""
        return np.exp(self._acceptance_weighted(-1, lambda rel: np.log(rel + 1e-10)))

    @property
    def far_enough_fraction(self):
        """Fraction of jumps exceeding the MLFriends radius."""
        return self._acceptance_weighted(-2)

    def get_info_dict(self):
        have = len(self.logstat) > 0
        col = lambda i: np.nanmean([row[i] for row in self.logstat]) if have else np.nan  # noqa: E731
        return dict(num_logs=len(self.logstat), rejection_rate=1 - col(0), mean_scale=col(1), mean_nsteps=col(2),
                    mean_distance=self.mean_jump_distance, frac_far_enough=self.far_enough_fraction,
                    last_logstat=dict(zip(self.logstat_labels,
                                          self.logstat[-1] if len(self.logstat) > 1 else [np.nan] * 5)))

    def print_diagnostic(self):
        if len(self.logstat) == 0:
            print("diagnostic unavailable, no recorded steps found")
            return
        frac = self.far_enough_fraction
        if frac < 0.5:
            advice = ': very fishy. Double nsteps and see if fraction and lnZ change)'
        elif frac < 0.66:
            advice = ': fishy. Double nsteps and see if fraction and lnZ change)'
        else:
            advice = ' (should be >50%)'
        print('step sampler diagnostic: jump distance %.2f (should be >1), far enough fraction: %.2f%% %s' % (
            self.mean_jump_distance, frac * 100, advice))

    def region_changed(self, Ls, region):
        pass


class _BatchedPopulationSampler(GenericPopulationSampler):
    """The two samplers of the reference that advance the WHOLE population by `nsteps` moves in one call and then hand the
    walkers out one per call (reference popstepsampler.py:192-358 and :746-1001).  The moves themselves are the vectorised
    step functions of this package (direction generators, line / cube intersection, slice bookkeeping, move diagnostics: HIP
    kernels behind ``ultranest_amd.stepfuncs``); what is restated here is their order and the ``np.random`` / scipy draws,
    so that a seeded run returns the reference's points (tests/test_popstepsampler.py, golden g16)."""

    def region_changed(self, Ls, region):
        """Nothing of the region is cached between calls."""

    def _refill(self, region, Lmin, us, Ls, transform, loglike, **kwargs):
        raise NotImplementedError

    def __next__(self, region, Lmin, us, Ls, transform, loglike, ndraw=10, plot=False, tregion=None, log=False, **kwargs):
        """``(u, p, L, nc)`` of the next prepared walker; the call that finds none left first moves the whole population
        (nc = its likelihood evaluations, 0 otherwise)."""
        nc = 0
        if not self.prepared_samples:
            nc = self._refill(region, Lmin, us, Ls, transform, loglike, **kwargs)
        u, p, L = self.prepared_samples.pop(0)
        return u, p, L, nc


class PopulationRandomWalkSampler(_BatchedPopulationSampler):
    """Vectorized Gaussian random walk (reference popstepsampler.py:192-358; same constructor and ``__next__`` contract)."""

    def __init__(self, popsize, nsteps, generate_direction, scale, scale_adapt_factor=0.9, scale_min=1e-20, scale_max=20,
                 log=False, logfile=None):
        assert scale_adapt_factor <= 1
        self.popsize, self.nsteps, self.generate_direction = popsize, nsteps, generate_direction
        self.scale, self.scale_adapt_factor, self.scale_min, self.scale_max = scale, scale_adapt_factor, scale_min, scale_max
        self.nrejects = self.ncalls = 0
        self.log, self.logfile, self.logstat = log, logfile, []
        self.logstat_labels = ['accept_rate', 'efficiency', 'scale', 'far_enough', 'mean_rel_jump']
        self.prepared_samples = []

    def __str__(self):
        return 'PopulationRandomWalkSampler(popsize=%d, nsteps=%d, generate_direction=%s, scale=%.g)' % (
            self.popsize, self.nsteps, self.generate_direction, self.scale)

    def _refill(self, region, Lmin, us, Ls, transform, loglike):
        import scipy.stats
        nlive = len(us)
        nmoves = self.nsteps * self.popsize
        target_rejects = nmoves * (1 - 0.234)              # the acceptance rate the scale is steered to
        rejects_before = self.nrejects
        start = np.random.randint(0, nlive, size=self.popsize)
        allu, allL, allp = us[start, :], Ls[start], None
        accepted = np.zeros(self.popsize, dtype=bool)
        for _ in range(self.nsteps):
            v = self.generate_direction(allu, region, self.scale)
            tleft, tright = unitcube_line_intersection(allu, v)
            # a unit normal step along v, truncated to the part of the line inside the cube (scipy draws from np.random)
            t = scipy.stats.truncnorm.rvs(tleft, tright, loc=0, scale=1).reshape((-1, 1))
            unew = allu + v * t
            outside = ~np.logical_and(unew > 0, unew < 1).all(axis=1)
            assert not outside.any(), unew[outside, :]
            pnew = transform(unew)
            Lnew = loglike(pnew)
            accepted = Lnew > Lmin
            self.nrejects += (~accepted).sum()
            if allp is None:
                allp = pnew * np.nan
            allu[accepted, :], allp[accepted, :], allL[accepted] = unew[accepted, :], pnew[accepted, :], Lnew[accepted]
        assert np.isfinite(allp).all(), 'some walkers never moved! Double nsteps of PopulationRandomWalkSampler.'
        # (the reference diagnoses the walkers that accepted their LAST move: reference :334)
        far_enough, (moved, radius) = diagnose_move_distances(region, us[start[accepted], :], allu[accepted, :])
        self.prepared_samples = list(zip(allu, allp, allL))
        expected = rejects_before + target_rejects
        # (efficiency in the reference's own arithmetic, :339: the count before this call is recovered from `expected`)
        self.logstat.append([accepted.mean(), 1 - (self.nrejects - (expected - target_rejects)) / nmoves,
                             self.scale, self.nsteps, np.mean(far_enough), np.exp(np.mean(np.log(moved / radius + 1e-10)))])
        if self.logfile:    # (the reference's own format string takes five of the six columns)
            row = self.logstat[-1]
            self.logfile.write("rescale\t%.4f\t%.4f\t%g\t%.4f%g\n" % (row[0], row[1], row[2], row[4], row[5]))
        if self.nrejects > expected and self.scale > self.scale_min:
            self.scale *= self.scale_adapt_factor            # too many rejections: shorter steps
        elif self.nrejects < expected and self.scale < self.scale_max:
            self.scale /= self.scale_adapt_factor
        return nmoves


class PopulationSimpleSliceSampler(_BatchedPopulationSampler):
    """Vectorized slice sampler without stepping out: the slice starts as the line's intersection with the unit cube
    (or ``[-1, 1]`` of the scaled direction, ``slice_limit_to_scale``) and shrinks towards the current point
    (reference popstepsampler.py:746-1001; same constructor and ``__next__`` contract).  The likelihood is always called
    with ``popsize`` points: workers whose point has found its successor are dealt to the points still searching
    (``update_vectorised_slice_sampler``, reference stepfuncs.pyx:537-630 -- one workgroup on the device here)."""

    def __init__(self, popsize, nsteps, generate_direction, scale_adapt_factor=1.0, adapt_slice_scale_target=2.0, scale=1.0,
                 scale_jitter_func=None, slice_limit=slice_limit_to_unitcube, max_it=100, shrink_factor=1.0):
        assert shrink_factor >= 1.0, "The shrink factor should be greater than 1.0 to be efficient"
        self.popsize, self.nsteps, self.generate_direction = popsize, nsteps, generate_direction
        self.max_it, self.shrink_factor = max_it, shrink_factor
        self.scale, self.scale_adapt_factor, self.adapt_slice_scale_target = float(scale), scale_adapt_factor, adapt_slice_scale_target
        self.scale_jitter_func = (lambda: 1.) if scale_jitter_func is None else scale_jitter_func
        self.slice_limit = slice_limit
        self.nrejects = self.ncalls = self.discarded = 0
        self.logstat = []
        self.logstat_labels = ['accept_rate', 'efficiency', 'scale', 'far_enough', 'mean_rel_jump']
        self.prepared_samples = []

    def __str__(self):
        return 'PopulationSimpleSliceSampler(popsize=%d, nsteps=%d, generate_direction=%s, scale=%.g)' % (
            self.popsize, self.nsteps, self.generate_direction, self.scale)

    def _refill(self, region, Lmin, us, Ls, transform, loglike, test=False):
        nlive, ndim = us.shape
        P = self.popsize
        start = np.random.randint(0, nlive, size=P)
        allu = np.array(us if test else us[start, :], dtype=float)        # test: the live points themselves (reversibility checks)
        allp = np.full((P, ndim), np.nan)
        allL = np.array(Ls[start], dtype=float)
        nc = ndiscarded = 0
        width_sum = 0.
        for _ in range(self.nsteps):
            jitter = self.scale_jitter_func()
            v = self.generate_direction(allu, region, scale=1.0) * self.scale * jitter
            cube_left, cube_right = unitcube_line_intersection(allu, v)
            wleft, wright = self.slice_limit(cube_left, cube_right)           # bounds seen by each WORKER (likelihood slot)
            tleft, tright = self.slice_limit(cube_left, cube_right)           # bounds of each POINT's slice
            worker_running = np.arange(P, dtype=int_dtype)                    # which point a worker serves
            status = np.zeros(P, dtype=int_dtype)                            # 1 = the point has its successor
            for _it in range(self.max_it):
                t = wleft + (wright - wleft) * np.random.uniform(size=(P,))
                unew = allu[worker_running, :] + t.reshape((-1, 1)) * v[worker_running, :]
                pnew = transform(unew)
                Lnew = loglike(pnew)
                nc += P
                tleft, tright, worker_running, status, allu, allL, allp, nd = update_vectorised_slice_sampler(
                    t, tleft, tright, Lnew, unew, pnew, worker_running, status, Lmin, self.shrink_factor, allu, allL, allp, P)
                ndiscarded += nd
                wleft, wright = tleft[worker_running], tright[worker_running]
                if not np.any(status == 0):
                    break
            width_sum += np.median(tright - tleft)
        mean_width = width_sum / self.nsteps
        self.discarded += ndiscarded
        self.ncalls += nc
        assert np.isfinite(allp).all(), 'some walkers never moved! Double nsteps of PopulationSimpleSliceSampler.'
        far_enough, (moved, radius) = diagnose_move_distances(region, us[start, :], allu)
        self.prepared_samples = list(zip(allu, allp, allL))
        have = len(far_enough) > 0
        self.logstat.append([P / nc, self.scale, self.nsteps, np.mean(far_enough) if have else 0,
                             np.exp(np.mean(np.log(moved / radius + 1e-10))) if have else 0])
        # steer the scale so that the final slices are 1 / adapt_slice_scale_target wide (reference :990-995)
        if mean_width >= 1. / self.adapt_slice_scale_target:
            self.scale *= 1. / self.scale_adapt_factor
        else:
            self.scale *= self.scale_adapt_factor
        return nc


class _Walkers(object):
    """Owner of one ``mlf_walkers`` handle (include/mlfriends_hip.h)."""

    def __init__(self, popsize, nsteps, ndim):
        self.popsize, self.nsteps, self.ndim = int(popsize), int(nsteps), int(ndim)
        handle = ctypes.c_void_p()
        check(_lib.lib().mlf_walkers_create(ctypes.byref(handle), self.popsize, self.nsteps, self.ndim))
        self._h = handle
        self.nparams = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mlf_walkers_destroy(h)
            except Exception:
                pass

    def begin(self, Lmin):
        generation = np.empty(self.popsize, dtype=np.int64)
        flags = np.empty(self.popsize, dtype=np.uint8)
        check(_lib.lib().mlf_walkers_begin(self._h, float(Lmin), ptr(generation), ptr(flags)))
        return generation, flags

    def start(self, idx, rows, L):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        check(_lib.lib().mlf_walkers_start(self._h, ptr(idx), len(idx), ptr(f64(rows)), ptr(f64(L))))

    def points(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty((len(idx), self.ndim))
        check(_lib.lib().mlf_walkers_points(self._h, ptr(idx), len(idx), ptr(out)))
        return out

    def brackets(self, idx, scale, v):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        v = f64(v)
        if v.shape != (len(idx), self.ndim):
            raise ValueError("generate_direction returned shape %s, expected %s" % (v.shape, (len(idx), self.ndim)))
        check(_lib.lib().mlf_walkers_brackets(self._h, ptr(idx), len(idx), float(scale), ptr(v)))

    def set_direction_data(self, axes=None, live=None, std=None):
        live = None if live is None else f64(live)
        check(_lib.lib().mlf_walkers_set_direction_data(
            self._h, ptr(None if axes is None else f64(axes)), ptr(live), 0 if live is None else len(live),
            ptr(None if std is None else f64(std))))

    def brackets_philox(self, scale, kind, dirscale, rng):
        nxt = ctypes.c_uint64(0)
        check(_lib.lib().mlf_walkers_brackets_philox(self._h, float(scale), int(kind), float(dirscale),
                                                     ctypes.c_uint64(rng.seed), ctypes.c_uint64(rng.offset),
                                                     ctypes.byref(nxt)))
        rng.offset = nxt.value

    def set_layer(self, kind, ctr, mat, wrap, maxradiussq):
        check(_lib.lib().mlf_walkers_set_layer(self._h, int(kind), ptr(None if ctr is None else f64(ctr)),
                                               ptr(None if mat is None else f64(mat)),
                                               ptr(None if wrap is None else f64(wrap)), float(maxradiussq)))

    def propose(self, unif=None, rng=None, fetch=True):
        """unif: one U[0,1) per walker (host stream) or None with rng = DeviceRNG.  Returns the
        acceptable proposals (walker order) when fetch, else None."""
        seed, offset = (rng.seed, rng.offset) if rng is not None else (0, 0)
        if rng is not None:
            rng.offset += self.popsize
        unif = None if unif is None else f64(unif)
        if not fetch:
            check(_lib.lib().mlf_walkers_propose(self._h, ptr(unif), ctypes.c_uint64(seed), ctypes.c_uint64(offset),
                                                 None, None))
            return None
        out = np.empty((self.popsize, self.ndim))
        nacc = ctypes.c_size_t(0)
        check(_lib.lib().mlf_walkers_propose(self._h, ptr(unif), ctypes.c_uint64(seed), ctypes.c_uint64(offset),
                                             ptr(out), ctypes.byref(nacc)))
        return out[:nacc.value]

    def _record(self, rec, nparams):
        d = self.ndim
        return dict(found=rec[0] == 1.0, L=rec[1], left=rec[2], right=rec[3], nc=int(rec[4]), nmovable=int(rec[5]),
                    nsuccess=int(rec[6]), nfar=rec[7], sumlog=rec[8], u=rec[9:9 + d].copy(),
                    p=rec[9 + d:9 + d + nparams].copy())

    def finish(self, Lmin, pnew, Lnew, ringindex):
        pnew, Lnew = f64(pnew), f64(Lnew)
        nacc = len(Lnew)
        if nacc:
            if pnew.ndim != 2 or pnew.shape[0] != nacc:
                raise ValueError("transform must return one row per proposed point")
            self.nparams = pnew.shape[1]
        elif self.nparams is None:
            self.nparams = self.ndim
        rec = np.empty(9 + self.ndim + self.nparams)
        check(_lib.lib().mlf_walkers_finish(self._h, float(Lmin), ptr(pnew), ptr(Lnew), nacc, self.nparams,
                                            int(ringindex), ptr(rec)))
        return self._record(rec, self.nparams)

    def finish_dev(self, Lmin, tspec, lspec, ringindex):
        self.nparams = self.ndim
        tkind, ta, tb = tspec
        lkind, aux, sigma = lspec
        rec = np.empty(9 + 2 * self.ndim)
        check(_lib.lib().mlf_walkers_finish_dev(self._h, float(Lmin), int(tkind), float(ta), float(tb), int(lkind),
                                                ptr(None if aux is None else f64(aux)), float(sigma),
                                                int(ringindex), ptr(rec)))
        return self._record(rec, self.ndim)

    def set_live(self, us, Ls):
        us, Ls = f64(us), f64(Ls)
        check(_lib.lib().mlf_walkers_set_live(self._h, ptr(us), ptr(Ls), len(Ls)))

    def update_live(self, rows, us_rows, Ls_rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        check(_lib.lib().mlf_walkers_update_live(self._h, ptr(rows), len(rows), ptr(f64(us_rows)), ptr(f64(Ls_rows))))

    def step_dev(self, Lmin, scale, kind, dirscale, rng, tspec, lspec, graph=True):
        """Whole sampler step on the device (graph: replayed as one hipGraph launch); returns the record
        (with the ring index after the step)."""
        self.nparams = self.ndim
        tkind, ta, tb = tspec
        lkind, aux, sigma = lspec
        rec = np.empty(10 + 2 * self.ndim)
        nxt = ctypes.c_uint64(0)
        fn = _lib.lib().mlf_walkers_step_graph if graph else _lib.lib().mlf_walkers_step_dev
        check(fn(self._h, float(Lmin), float(scale), int(kind), float(dirscale), ctypes.c_uint64(rng.seed),
                 ctypes.c_uint64(rng.offset), int(tkind), float(ta), float(tb), int(lkind),
                 ptr(None if aux is None else f64(aux)), float(sigma), ptr(rec), ctypes.byref(nxt)))
        rng.offset = nxt.value
        out = self._record(rec, self.ndim)
        out["ring"] = int(rec[9 + 2 * self.ndim])
        return out

    def rounds_dev(self, Lmin, scale, kind, dirscale, rng, tspec, lspec, max_rounds):
        """Whole sampler steps on the device until the walker under the ring index has finished (or `max_rounds`):
        ``mlf_walkers_rounds_dev``.  Returns the record of the LAST round (with "ring" and "rounds") and the per-round
        statistics rows (rounds, 5): what `rounds` consecutive `step_dev` calls would have returned, one host round trip."""
        self.nparams = self.ndim
        tkind, ta, tb = tspec
        lkind, aux, sigma = lspec
        rec = np.empty(10 + 2 * self.ndim)
        if getattr(self, "_round_rows", None) is None or len(self._round_rows) < abs(max_rounds):
            self._round_rows = np.empty((abs(int(max_rounds)), 5))
        nxt = ctypes.c_uint64(0)
        nrounds = ctypes.c_int(0)
        check(_lib.lib().mlf_walkers_rounds_dev(
            self._h, float(Lmin), float(scale), int(kind), float(dirscale), ctypes.c_uint64(rng.seed), ctypes.c_uint64(rng.offset),
            int(tkind), float(ta), float(tb), int(lkind), ptr(None if aux is None else f64(aux)), float(sigma), int(max_rounds),
            ptr(rec), ptr(self._round_rows), ctypes.byref(nrounds), ctypes.byref(nxt)))
        rng.offset = nxt.value
        R = nrounds.value
        rows = self._round_rows[:R]
        d = self.ndim
        out = dict(found=rec[0] == 1.0, L=rec[1], left=rec[2], right=rec[3], u=rec[9:9 + d].copy(), p=rec[9 + d:9 + 2 * d].copy(),
                   ring=int(rec[9 + 2 * d]), rounds=R)
        return out, rows

    def export(self):
        """Host copies of the resident state (tests, debugging)."""
        P, G, d = self.popsize, self.nsteps + 1, self.ndim
        s = dict(allu=np.empty((P, G, d)), allL=np.empty((P, G)), generation=np.empty(P, dtype=np.int64),
                 currentt=np.empty(P), currentv=np.empty((P, d)), current_left=np.empty(P), current_right=np.empty(P),
                 searching_left=np.empty(P, dtype=np.uint8), searching_right=np.empty(P, dtype=np.uint8))
        check(_lib.lib().mlf_walkers_export(self._h, ptr(s["allu"]), ptr(s["allL"]), ptr(s["generation"]),
                                            ptr(s["currentt"]), ptr(s["currentv"]), ptr(s["current_left"]),
                                            ptr(s["current_right"]), ptr(s["searching_left"]),
                                            ptr(s["searching_right"])))
        s["searching_left"] = s["searching_left"].view(np.bool_)
        s["searching_right"] = s["searching_right"].view(np.bool_)
        return s


class PopulationSliceSampler(GenericPopulationSampler):
    """Vectorized slice/HARM sampler over a device-resident walker population
    (reference popstepsampler.py:347-697; same constructor and ``__next__`` contract)."""

    def __init__(self, popsize, nsteps, generate_direction, scale=1.0, scale_adapt_factor=0.9, log=False,
                 logfile=None, device_rng=None):
        self.popsize, self.nsteps, self.generate_direction = popsize, nsteps, generate_direction
        self.scale, self.scale_adapt_factor = scale, scale_adapt_factor
        self.log, self.logfile, self.logstat = log, logfile, []
        self.nrejects = self.ringindex = 0
        if device_rng is not None and not isinstance(device_rng, DeviceRNG):
            raise TypeError("device_rng must be an ultranest_amd.regions.DeviceRNG")
        self.device_rng = device_rng
        self.use_graph = True     # whole-step path: replay the kernel sequence as one hipGraph launch
        # whole-step path: up to this many rounds (= calls of the reference's __next__, which the driver repeats until a
        # point comes back: integrator.py:1839-1950) per device call, each walker's rounds back to back inside the wave
        # that owns it; the call returns when the walker under the ring index has finished.  0 / 1: one round per call
        self.max_rounds = 256
        self._walkers = None
        self._generation = np.zeros(popsize, dtype=int_dtype) - 1
        self._flags = np.ones(popsize, dtype=np.uint8)
        self._seen = dict(region=None, layer=None, r2=None, calls=0)

    def __str__(self):
        return 'PopulationSliceSampler(popsize=%d, nsteps=%d, generate_direction=%s, scale=%.g)' % (
            self.popsize, self.nsteps, self.generate_direction, self.scale)

    # ---- introspection (host copies) -------------------------------------------------------
    @property
    def generation(self):
        return self._generation

    @property
    def status(self):
        """Compact string of the walker states after the last call (reference :472-481)."""
        gen = 'G:' + ''.join(['%d' % g if g >= 0 else '_' for g in self._generation])
        st = 'S:' + ''.join(['S' if f & 1 else 'L' if f & 2 else 'R' if f & 4 else 'B' for f in self._flags])
        return gen + '  ' + st

    def state(self):
        """dict of host copies of allu, allL, generation, currentt, currentv, brackets, flags."""
        if self._walkers is None:
            raise RuntimeError("the sampler has not been called yet")
        return self._walkers.export()

    def region_changed(self, Ls, region):
        """The driver rebuilt the region: refresh the device copies derived from it."""
        self._seen["region"] = None
        self._seen["live_age"] = None
        if self.logfile:
            self.logfile.write("region-update\t%g\t%g\n" % (self.scale, region.u.std(axis=1).mean()))

    def shift(self):
        """Advance the ring index of the walker harvested next (reference :605-609)."""
        self.ringindex = (self.ringindex + 1) % self.popsize

    # ---- device copies of what the region contributes -----------------------------------------
    def _sync_region(self, region, skip_live=False):
        w, seen = self._walkers, self._seen
        layer = region.transformLayer
        r2 = region.maxradiussq
        if seen["region"] is not region or seen["layer"] is not layer or seen["r2"] != r2:
            ndim = self._walkers.ndim
            try:
                kind, ctr, mat = layer.device_params(ndim)
                wrap = layer.wrap_shift_vector(ndim)
            except AttributeError:      # a foreign layer object: whiten through its numpy attributes
                kind, ctr, mat, wrap = 0, np.broadcast_to(layer.ctr, (ndim,)), layer.T, None
            if r2 is None:
                w.set_layer(-1, None, None, None, 1.0)
            else:
                w.set_layer(kind, ctr, mat, wrap, r2)
            seen.update(layer=layer, r2=r2)
        kind = getattr(self.generate_direction, "device_kind", None)
        if self.device_rng is not None and kind is not None:
            fresh = seen["region"] is not region
            if kind in (3, 4, 6) and fresh:
                w.set_direction_data(axes=region.transformLayer.axes)
            if kind == 1 and (fresh or seen["calls"] % 32 == 0):
                w.set_direction_data(std=region.u.std(axis=0))
            # (the whole-step path keeps live points and their likelihoods together: set_live)
            if kind in (5, 6) and not skip_live and (fresh or region.u.size <= 32768 or seen["calls"] % 32 == 0):
                w.set_direction_data(live=region.u)
        seen["region"] = region
        seen["calls"] += 1

    # ---- one sampler step ------------------------------------------------------------------------
    def __next__(self, region, Lmin, us, Ls, transform, loglike, ndraw=10, plot=False, tregion=None, log=False):
        """Advance every walker by one likelihood evaluation; return ``(u, p, L, nc)`` of the
        next finished walker, or ``(None, None, None, nc)`` (reference :610-697)."""
        nlive, ndim = us.shape
        if self._walkers is None:
            self._walkers = _Walkers(self.popsize, self.nsteps, ndim)
        w = self._walkers
        tspec, lspec = getattr(transform, "device_spec", None), getattr(loglike, "device_spec", None)
        device_kind = getattr(self.generate_direction, "device_kind", None)
        whole_step = self.device_rng is not None and device_kind is not None and tspec is not None and lspec is not None
        self._sync_region(region, skip_live=whole_step)
        if whole_step:
            return self._next_on_device(region, Lmin, us, Ls, device_kind, tspec, lspec)

        # step_back on the device; the host learns which walkers need what
        generation, flags = w.begin(Lmin)
        starting = generation < 0
        if starting.any():
            above = np.flatnonzero(Ls > Lmin)
            pick = above[np.random.randint(len(above), size=int(starting.sum()))]
            if not starting.all():
                while starting[self.ringindex]:
                    self.shift()
            w.start(np.flatnonzero(starting), us[pick], Ls[pick])
            generation[starting] = 0
        assert (generation >= 0).all(), generation

        undefined = (flags & 1).astype(bool)            # bracket undefined: new slice
        on_device = self.device_rng is not None and device_kind is not None
        if undefined.any():
            if on_device:
                w.brackets_philox(self.scale, device_kind, 1.0, self.device_rng)
            else:
                idx = np.flatnonzero(undefined)
                if getattr(self.generate_direction, "needs_points", True):
                    start_points = w.points(idx)
                else:
                    start_points = np.zeros((len(idx), ndim))
                w.brackets(idx, self.scale, self.generate_direction(start_points, region))

        movable = generation < self.nsteps
        stepping = np.logical_and((flags & 6) != 0, ~undefined)
        bisecting = np.logical_and(movable, ~np.logical_or(stepping, undefined))
        resident_likelihood = tspec is not None and lspec is not None
        if self.device_rng is not None:
            unif, rng = None, self.device_rng
        else:
            unif, rng = np.zeros(self.popsize), None
            unif[bisecting] = np.random.random_sample(int(bisecting.sum()))
        if resident_likelihood:
            w.propose(unif, rng, fetch=False)
            rec = w.finish_dev(Lmin, tspec, lspec, self.ringindex)
        else:
            unew = w.propose(unif, rng, fetch=True)
            if len(unew):
                pnew = transform(unew)
                Lnew = loglike(pnew)
            else:
                pnew, Lnew = np.empty((0, w.nparams or ndim)), np.empty(0)
            rec = w.finish(Lmin, pnew, Lnew, self.ringindex)
        self._generation, self._flags = generation, flags
        return self._finish_call(rec, region)

    def _next_on_device(self, region, Lmin, us, Ls, device_kind, tspec, lspec):
        """Philox stream + resident likelihood: the whole step is one sequence of kernels
        (``mlf_walkers_step_dev``): restarts draw from a device copy of the live points and the ring
        index lives on the device; one record comes back."""
        w, seen = self._walkers, self._seen
        # the device copy of (us, Ls) follows the host arrays ROW BY ROW: the driver replaces one live point per iteration
        # (integrator.py:2753-2754), so the rows whose likelihood or first coordinate changed since the last call are
        # uploaded (typically one: 8 (d + 2) bytes through pinned staging, no synchronisation) instead of the whole set
        # (two pageable copies per call: ~40 us of an 80 us call at 1000 x 10)
        mirror = seen.get("live_mirror")
        if seen.get("live_age", None) is None or mirror is None or mirror[0].shape != Ls.shape or mirror[2] != us.shape[1]:
            w.set_live(us, Ls)
            seen["live_mirror"] = [np.array(Ls, dtype=float), np.array(us[:, 0], dtype=float), us.shape[1]]
            seen["live_age"] = 0
        else:
            changed = np.flatnonzero(np.logical_or(Ls != mirror[0], us[:, 0] != mirror[1]))
            if len(changed) > max(8, len(Ls) // 8):
                w.set_live(us, Ls)
                mirror[0][:] = Ls
                mirror[1][:] = us[:, 0]
            elif len(changed):
                w.update_live(changed, us[changed], Ls[changed])
                mirror[0][changed] = Ls[changed]
                mirror[1][changed] = us[changed, 0]
        seen["live_age"] += 1
        if abs(self.max_rounds) > 1:      # (negative: the same rounds through the general, memory-resident form -- tests)
            rec, rows = w.rounds_dev(Lmin, self.scale, device_kind, 1.0, self.device_rng, tspec, lspec, self.max_rounds)
            have_diag = region.maxradiussq is not None
            nc = int(rows[:, 0].sum())
            ok = rows[:, 2] > 0            # one logstat row per round with a success, as one call each would have made
            if ok.any():
                r = rows[ok]
                ns = r[:, 2]
                table = np.empty((len(r), 5))
                table[:, 0] = ns / np.maximum(r[:, 1], 1)
                table[:, 1] = self.scale
                table[:, 2] = self.nsteps
                table[:, 3] = r[:, 3] / ns if have_diag else 0
                table[:, 4] = np.exp(r[:, 4] / ns) if have_diag else 0
                new_rows = table.tolist()
                self.logstat.extend(new_rows)
                if self.logfile:
                    for row in new_rows:
                        self.logfile.write("rescale\t%.4f\t%.4f\t%g\t%.4f%g\n" % tuple(row))
            rec.update(nc=nc, nsuccess=0)
            self.rounds_last_call = rec["rounds"]
        else:
            rec = w.step_dev(Lmin, self.scale, device_kind, 1.0, self.device_rng, tspec, lspec, graph=self.use_graph)
        out = self._finish_call(rec, region, shift=False)
        self.ringindex = rec["ring"]
        return out

    def _finish_call(self, rec, region, shift=True):
        nc = rec["nc"]
        if rec["nsuccess"] > 0:
            ns = rec["nsuccess"]
            have_diag = region.maxradiussq is not None
            self.logstat.append([ns / max(rec["nmovable"], 1), self.scale, self.nsteps,
                                 rec["nfar"] / ns if have_diag else 0, np.exp(rec["sumlog"] / ns) if have_diag else 0])
            if self.logfile:
                self.logfile.write("rescale\t%.4f\t%.4f\t%g\t%.4f%g\n" % tuple(self.logstat[-1]))
        if rec["found"]:
            u, p, L = rec["u"], rec["p"], rec["L"]
            assert np.isfinite(u).all(), u
            assert np.isfinite(p).all(), p
            newscale = (rec["right"] - rec["left"]) / 2
            self.scale = self.scale * 0.9 + 0.1 * newscale
            if shift:
                self.shift()
            return u, p, L, nc
        return None, None, None, nc


__all__ = [
    "generate_cube_oriented_direction", "generate_cube_oriented_direction_scaled", "generate_random_direction",
    "generate_region_oriented_direction", "generate_region_random_direction", "generate_differential_direction",
    "generate_mixture_random_direction", "PopulationSliceSampler", "PopulationRandomWalkSampler",
    "PopulationSimpleSliceSampler", "unitcube_line_intersection",
    "diagnose_move_distances",
    "slice_limit_to_unitcube", "slice_limit_to_scale", "int_dtype"]
