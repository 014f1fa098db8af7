"""Counterpart of the reference DRIVER's calls into the region path, for benchmarking and
end-to-end parity of that path without shipping the driver: the region rebuild
(reference integrator.py `_update_region` :1952-2159, call stack A of SURVEY.md 3) and one
proposal batch through the vectorized-likelihood callbacks (`_refill_samples` :1773-1837, stack
B).  Control flow only; every numerical stage is a call into ultranest_amd.mlfriends /
ultranest_amd.distributed.  The reference's own driver can be used instead, unmodified, by
passing these classes through its plug points."""
import gc

import numpy as np

from . import distributed
from .layers import host_worker, single_blas_thread
from .mlfriends import LocalAffineLayer, MLFriends, WrappingEllipsoid, find_nearby, int_dtype


class RegionUpdater(object):
    """Holds (region, transformLayer, tregion) like the driver does and rebuilds them from the
    current live points."""

    _gc_frozen = False

    def __init__(self, x_dim, region_class=MLFriends, transform_layer_class=LocalAffineLayer,
                 wrapped_axes=(), group=None, build_tregion=True, device_resident=False, freeze_gc=False):
        self.x_dim = x_dim
        # The objects the imports leave behind (~170000) make CPython's first generation-2 collector pass cost 37-42 ms,
        # somewhere in the first ~40 rebuilds (scripts/rebuild_rounds.py).  Moving what exists NOW into the permanent
        # generation keeps that pass out of the rebuild loop.  That is a PROCESS-WIDE side effect (everything alive in the
        # host application stops being collectable), so it is opt-in (`freeze_gc=True`: bench.py and the timing scripts pass
        # it); a library object leaves the collector alone by default.
        if freeze_gc and not RegionUpdater._gc_frozen:
            gc.collect()
            gc.freeze()
            RegionUpdater._gc_frozen = True
        self.region_class = region_class
        self.transform_layer_class = transform_layer_class
        self.wrapped_axes = list(wrapped_axes)
        self.group = group
        self.build_tregion = build_tregion
        # opt-in: the steady-state rebuild with the live points resident in HBM (ultranest_amd.device_rebuild: tolerance
        # class 1e-10 on T / radius instead of the default path's bit parity with the reference's numpy calls)
        self.device_resident = bool(device_resident)
        self._device_rebuild = None
        self.region = None
        self.transformLayer = None
        self.tregion = None

    def _use_device_rebuild(self, minvol):
        if not self.device_resident:
            return False
        from . import device_rebuild
        size = distributed.world(self.group)[1]
        if not device_rebuild.supported(self.transformLayer, self.region_class, self.x_dim, minvol, size):
            return False
        if self._device_rebuild is None:
            self._device_rebuild = device_rebuild.DeviceRebuild()
        return True

    def _bootstrap(self, region, nbootstraps, minvol, masks=None):
        return distributed.update_region_bootstrap(region, nbootstraps, minvol, group=self.group, masks=masks)

    def _revalidate_radius(self, active_u, nbootstraps, minvol):
        """Radius was invalidated (driver sets maxradiussq = None, :2827): bootstrap it again on
        the current layer and carry the cluster labels over by proximity (:2009-2053)."""
        old_u = self.region.u
        self.region.u = active_u
        self.region.set_transformLayer(self.transformLayer)
        self._bootstrap(self.region, nbootstraps, minvol)
        old_t = self.transformLayer.transform(old_u)
        old_ids = self.transformLayer.clusterids
        labels = np.zeros(len(active_u), dtype=int_dtype)
        hit = np.empty(len(self.region.unormed), dtype=int_dtype)
        for cid in np.unique(old_ids):
            if cid == 0:
                continue
            find_nearby(old_t[old_ids == cid], self.region.unormed, self.region.maxradiussq, hit)
            near = hit != 0           # the driver's test (:2042), see SURVEY.md appendix A5
            labels[near] = np.where(labels[near] == 0, cid, -1)
        labels[labels == -1] = 0
        self.transformLayer.clusterids = labels
        self.region.create_ellipsoid(minvol=minvol)
        return (labels == 0).any()

    @single_blas_thread
    def update(self, active_u, nbootstraps=30, minvol=0., active_p=None):
        """Returns True if the region object was (re)built or replaced."""
        assert nbootstraps > 0
        updated = False
        if self.region is None:
            self.transformLayer = self.transform_layer_class(wrapped_dims=self.wrapped_axes)
            self.transformLayer.optimize(active_u, active_u, minvol=minvol)
            self.region = self.region_class(active_u, self.transformLayer)
            self._bootstrap(self.region, nbootstraps, minvol)
            self.region.create_ellipsoid(minvol=minvol)
            updated = True
        need_accept = False
        if self.region.maxradiussq is None:
            need_accept = self._revalidate_radius(active_u, nbootstraps, minvol)
            updated = True

        with np.errstate(all='raise'):
            try:
                if self._use_device_rebuild(minvol):
                    nxt_layer, nxt, contains_live = self._device_rebuild.next_region(
                        active_u, self.transformLayer, self.region.maxradiussq, nbootstraps)
                    assert not (nxt_layer.clusterids == 0).any()
                    _, sizes = np.unique(nxt_layer.clusterids, return_counts=True)
                else:
                    nxt_layer = self.transformLayer.create_new(active_u, self.region.maxradiussq, minvol=minvol)
                    assert not (nxt_layer.clusterids == 0).any()
                    _, sizes = np.unique(nxt_layer.clusterids, return_counts=True)
                    # the bootstrap's selection masks are drawn on the worker thread (the compiled walk of numpy's MT19937
                    # stream: 0.2 ms for 30 x 4000 draws) while this thread builds the region object (its `transform` product and
                    # range checks: 0.7 ms).  Nothing between here and the bootstrap touches np.random, so the draws are the ones
                    # the reference's order makes; if the constructor raises, the generator is put back where it was
                    # -- for THIS package's region classes, whose constructors never touch np.random.  A user-supplied
                    # region_class may (the reference's duck-typed contract says nothing about it): its constructor runs
                    # first and the masks are drawn afterwards, on this thread, exactly in the reference's order
                    if _builtin_region_class(self.region_class):
                        rng_state = np.random.get_state()
                        draw = host_worker().submit(_draw, len(active_u), nbootstraps)
                        try:
                            nxt = self.region_class(active_u, nxt_layer)
                        except BaseException:
                            try:
                                draw.result()
                            finally:
                                np.random.set_state(rng_state)
                            raise
                        masks = draw.result()
                    else:
                        nxt = self.region_class(active_u, nxt_layer)
                        masks = _draw(len(active_u), nbootstraps)
                    self._bootstrap(nxt, nbootstraps, minvol, masks=masks)   # starts create_ellipsoid's host LAPACK on the worker thread
                    nxt.create_ellipsoid(minvol=minvol)
                    contains_live = nxt.inside(active_u).all()
                sensible = nxt_layer.nclusters < len(nxt.u) and sizes.max() >= nxt.u.shape[1]
                shrinks = need_accept or nxt.estimate_volume() <= self.region.estimate_volume()
                if contains_live and shrinks and sensible:
                    self.region = nxt
                    self.transformLayer = nxt_layer
                    updated = True
            except (Warning, FloatingPointError, np.linalg.LinAlgError):
                pass                  # keep the previous region, as the driver does (:2123-2131)

        if active_p is None or not self.build_tregion:
            self.tregion = None
        else:
            try:
                with np.errstate(invalid='raise'):
                    tregion = WrappingEllipsoid(active_p)
                    rank, size = distributed.world(self.group)
                    masks = distributed.broadcast_masks(       # every rank draws (streams stay in step), rank 0's counts
                        _draw(len(active_p), nbootstraps), len(active_p), nbootstraps, group=self.group)
                    lo, hi = distributed.shard_bounds(nbootstraps, rank, size)
                    f, failed, err = 0.0, 0.0, None
                    try:      # the shard may fail on ONE rank only: every rank still joins the all-reduce
                        f = tregion.enlargement_from_masks(masks[lo:hi]) if hi > lo else 0.0
                    except distributed.SHARD_ERRORS as e:
                        failed, err = distributed._error_class(e), e
                    f, failed = distributed.allreduce_max([f, failed], group=self.group)
                    if failed > 0:
                        if size == 1 and err is not None:
                            raise err     # a single process keeps the exception's own type, as the reference does
                        if failed >= 2:   # not a numerical failure: do not let the handler below swallow it
                            raise RuntimeError("tregion bootstrap: a rank failed (not a numerical error)") from err
                        raise np.linalg.LinAlgError("tregion bootstrap failed on a rank") from err
                    tregion.enlarge = float(f)
                    tregion.create_ellipsoid()
                    self.tregion = tregion
            except (FloatingPointError, np.linalg.LinAlgError):
                self.tregion = None       # on every rank alike
        return updated


def _builtin_region_class(cls):
    """True for the region classes of this package (exact types, not subclasses: a subclass may override __init__)."""
    from . import regions
    return cls in (regions.MLFriends, regions.RobustEllipsoidRegion, regions.SimpleRegion, regions.WrappingEllipsoid)


def _draw(npoints, nbootstraps):
    from .regions import _draw_selection
    return _draw_selection(np.random, npoints, nbootstraps)


def refill_samples(region, tregion, transform, loglike, Lmin, ndraw, pointstore=None, ncall=0):
    """One proposal batch (reference `_refill_samples`, integrator.py:1773-1837 with
    draw_multiple=True): region.sample -> transform -> tregion.inside -> loglike on the accepted
    rows -> keep logl > Lmin.  Returns (u, v, logl, ncalls)."""
    if tregion is None and pointstore is None and hasattr(region, "refill"):
        got = region.refill(ndraw, Lmin, transform, loglike)     # device-resident batch when possible
        if got is not None:
            return got
    u = region.sample(nsamples=ndraw)
    assert np.logical_and(u > 0, u < 1).all(), u
    nu = u.shape[0]
    if nu == 0:
        return u, np.empty((0, 0)), np.empty(0), 0
    v = transform(u)
    logl = np.ones(nu) * -np.inf
    accepted = tregion.inside(v) if tregion is not None else np.ones(nu, dtype=bool)
    nc = int(accepted.sum())
    if nc > 0:
        logl[accepted] = loglike(v[accepted, :])
    if pointstore is not None:   # every evaluated proposal: [Lmin, L, quality, u..., p...] (integrator.py:1937-1939)
        for ui, vi, li in zip(u[accepted], v[accepted], logl[accepted]):
            pointstore.add([Lmin, li, ndraw] + list(ui) + list(vi), ncall + nc)
    keep = logl > Lmin
    return u[keep, :], v[keep, :], logl[keep], nc


class StaticNestedSampler(object):
    """Minimal static nested sampler over the GPU region path -- NOT a re-implementation of the
    reference's ReactiveNestedSampler (tree search, bootstrapped error bars, dynamic live points,
    resume, MPI, plots are all out of scope).  It exists so that the region path can be driven end to
    end on a machine where the reference cannot travel: fixed number of live points, region proposals
    in large vectorized batches (`draw_multiple=True` style, reference integrator.py:1773-1837),
    region rebuild whenever the remaining volume shrank by 20 % (reference :2555, :2680-2698),
    in-place live-point replacement (:2749-2765), classic ln Z accumulation with X_i = exp(-i/N).

    loglike / transform follow the reference's ``vectorized=True`` callback contract."""

    def __init__(self, x_dim, loglike, transform=None, num_live_points=400, ndraw=4096,
                 region_class=MLFriends, transform_layer_class=LocalAffineLayer, nbootstraps=30, seed=1,
                 device_rng=None, stepsampler=None, pointstore=None, log_dir=None, paramnames=None, keep_tree=False):
        self.x_dim = x_dim
        # optional result files in the reference's layout (ultranest_amd.results): the run then keeps the
        # tree of dead and live points (root / pointpile, as the reference's sampler object does) and
        # replays it through netiter.logz_sequence at the end (reference integrator.py:2933-2995)
        self.log_dir = log_dir
        self.keep_tree = bool(keep_tree or log_dir is not None)
        self.paramnames = paramnames
        self.root = self.pointpile = self.results = self.run_sequence = self.logs = None
        # optional ultranest_amd.store point store: every likelihood evaluation is logged in the
        # reference's row format, and stored points are replayed before new ones are drawn (resume)
        self.pointstore = pointstore
        # optional population step sampler (ultranest_amd.popstepsampler): replaces region
        # rejection sampling by its __next__, called like the reference's driver does
        # (integrator.py:1839-1950)
        self.stepsampler = stepsampler
        # optional regions.DeviceRNG: proposals are then drawn, tested and compacted on the GPU
        self.device_rng = device_rng
        self.loglike = loglike
        self.transform = transform if transform is not None else (lambda u: u)
        self.nlive = num_live_points
        self.ndraw = ndraw
        self.nbootstraps = nbootstraps
        self.updater = RegionUpdater(x_dim, region_class=region_class, transform_layer_class=transform_layer_class,
                                     build_tregion=False)
        self.seed = seed
        self.ncall = 0
        self.ncall_region = 0
        # where a run's wall time goes (seconds and call counts): filled by run(); "host_loop_s" is what is left for the
        # per-iteration bookkeeping of the loop itself (scripts/e2e_run.py reports it)
        self.phases = {}

    def run(self, dlogz=0.5, max_iters=200000):
        import time
        tick = time.perf_counter
        ph = self.phases = dict(initial_points_s=0.0, first_rebuild_s=0.0, rebuild_s=0.0, rebuilds=0, refill_s=0.0, refills=0,
                                first_refill_s=0.0, stepsampler_s=0.0, stepsampler_calls=0, summary_s=0.0)
        t_run = tick()
        np.random.seed(self.seed)
        N = self.nlive
        u = np.random.uniform(size=(N, self.x_dim))
        # initial live points: stored rows with threshold -inf are reused (reference integrator.py:1501),
        # the rest is evaluated and logged as [-inf, L, 0, u, p] (:1556)
        stored = []
        if self.pointstore is not None:
            while len(stored) < N:
                _, row = self.pointstore.pop(-np.inf)
                if row is None:
                    break
                stored.append(np.asarray(row, dtype=float))
        nold = len(stored)
        logl = np.empty(N)
        v_live = None
        if nold:
            rows = np.array(stored)
            u[:nold] = rows[:, 3:3 + self.x_dim]
            logl[:nold] = rows[:, 1]
            v_live = rows[:, 3 + self.x_dim:]
        if nold < N:
            v_new = np.asarray(self.transform(u[nold:]))
            v_live = v_new if v_live is None else np.vstack((v_live, v_new))
            logl[nold:] = np.asarray(self.loglike(v_new), dtype=float)
            self.ncall += N - nold
            if self.pointstore is not None:
                for ui, vi, li in zip(u[nold:], np.asarray(v_new), logl[nold:]):
                    self.pointstore.add([-np.inf, li, 0.0] + list(ui) + list(vi), self.ncall)
        live_nodes = None
        if self.keep_tree:
            from .netiter import PointPile, TreeNode
            self.pointpile = PointPile(self.x_dim, v_live.shape[1])
            live_nodes = [self.pointpile.make_node(li, ui, vi) for li, ui, vi in zip(logl, u, v_live)]
            self.root = TreeNode(id=-1, value=-np.inf, children=list(live_nodes))
        ph["initial_points_s"] = tick() - t_run
        logz = -np.inf
        h_terms = []
        logvol = 0.0
        next_update_logvol = 0.0
        pending_u, pending_l = np.empty((0, self.x_dim)), np.empty(0)
        pending_v = None
        ip = 0      # next unused entry of the pending batch (entries at or below a past threshold stay dead)
        it = 0
        while it < max_iters:
            if logvol <= next_update_logvol:
                t0 = tick()
                self.updater.update(u, nbootstraps=self.nbootstraps, minvol=np.exp(logvol))
                next_update_logvol = logvol + np.log(0.8)
                if self.stepsampler is not None:
                    self.stepsampler.region_changed(logl, self.updater.region)
                ph["first_rebuild_s" if ph["rebuilds"] == 0 and ph["first_rebuild_s"] == 0.0 else "rebuild_s"] += tick() - t0
                ph["rebuilds"] += 1
            region = self.updater.region
            region.device_rng = self.device_rng
            worst = int(np.argmin(logl))
            Lmin = logl[worst]
            # weight of the dying point: shell between X_it and X_(it+1)
            logw = logvol + np.log1p(-np.exp(-1.0 / N)) + Lmin
            logz = np.logaddexp(logz, logw)
            h_terms.append((logw, Lmin))
            logvol -= 1.0 / N
            # remaining evidence bound: live points times remaining volume
            logz_remain = np.max(logl) + logvol
            if np.logaddexp(logz, logz_remain) - logz < dlogz and it > N:
                break
            # a replacement above Lmin from the region
            while self.stepsampler is not None:
                t0 = tick()
                newu, newv, newl, nc = self.stepsampler.__next__(region, Lmin, region.u, logl, self.transform,
                                                                  self.loglike)
                ph["stepsampler_s"] += tick() - t0
                ph["stepsampler_calls"] += 1
                self.ncall += nc
                if newu is not None:
                    break
            while self.stepsampler is None:
                # the threshold only rises: walk the batch once, like the driver's index `ib` (:1942-1950)
                while ip < len(pending_l) and not (pending_l[ip] > Lmin):
                    ip += 1
                if ip < len(pending_l):
                    newu, newl = pending_u[ip].copy(), pending_l[ip]
                    newv = pending_v[ip] if pending_v is not None else None
                    ip += 1
                    break
                if self.pointstore is not None and not self.pointstore.stack_empty:
                    _, row = self.pointstore.pop(Lmin)       # resume: replay stored evaluations first
                    if row is not None:
                        pending_u = np.array([row[3:3 + self.x_dim]])
                        pending_v = np.array([row[3 + self.x_dim:]])
                        pending_l = np.array([row[1]])
                        ip = 0
                        continue
                t0 = tick()
                nu, nv, nl, nc = refill_samples(region, None, self.transform, self.loglike, Lmin, self.ndraw,
                                                pointstore=self.pointstore, ncall=self.ncall)
                ph["first_refill_s" if ph["refills"] == 0 else "refill_s"] += tick() - t0
                ph["refills"] += 1
                self.ncall += nc
                self.ncall_region += self.ndraw
                pending_u, pending_v, pending_l = nu, nv, nl
                ip = 0
            # in-place replacement exactly as the driver does it
            region.u[worst] = newu
            region.unormed[worst] = region.transformLayer.transform(newu)
            region.ellipsoid_center = np.mean(region.u, axis=0)
            region.transformLayer.clusterids[worst] = 0
            u = region.u
            logl[worst] = newl
            if live_nodes is not None:      # the replacement hangs below the point it replaced (integrator.py:2749-2765)
                if newv is None:
                    newv = np.asarray(self.transform(newu[None, :]))[0]
                child = self.pointpile.make_node(newl, newu, np.asarray(newv))
                live_nodes[worst].children.append(child)
                live_nodes[worst] = child
            it += 1
        # remainder: the live points share the remaining volume equally
        logz_live = np.logaddexp.reduce(logl) + logvol - np.log(N)
        logz = np.logaddexp(logz, logz_live)
        logws = np.array([w for w, _ in h_terms])
        Ls = np.array([l for _, l in h_terms])
        p = np.exp(logws - logz)
        info = float(np.sum(p * (Ls - logz)))
        out = dict(logz=float(logz), logzerr=float(np.sqrt(max(info, 0.0) / N)), niter=it, ncall=self.ncall,
                   ncall_region=self.ncall_region, nclusters=int(self.updater.transformLayer.nclusters))
        if self.keep_tree:
            t0 = tick()
            out.update(self._summarize())
            ph["summary_s"] = tick() - t0
        ph["total_s"] = tick() - t_run
        ph["host_loop_s"] = ph["total_s"] - sum(v for k, v in ph.items() if k.endswith("_s") and k not in ("total_s", "host_loop_s"))
        ph["host_loop_us_per_iteration"] = 1e6 * ph["host_loop_s"] / max(it, 1)
        return out

    def _summarize(self):
        """Replay the tree through the bootstrapped counters and (with `log_dir`) write the result files
        exactly as the reference's `_update_results` does (integrator.py:2933-2995)."""
        from . import netiter, results as R
        with np.errstate(all="ignore"):
            sequence, res = netiter.logz_sequence(self.root, self.pointpile, random=True, check_insertion_order=True)
        names = self.paramnames
        if names is None:
            names = ["param%d" % (i + 1) for i in range(self.pointpile.pdim)]
        R.finish_results(res, res, self.ncall, names, float((res["H"] / self.nlive)**0.5))
        self.results, self.run_sequence = res, sequence
        if self.log_dir is not None:
            from .utils import make_run_dir
            self.logs = make_run_dir(self.log_dir)
            R.write_results(self.logs, res, sequence, names)
        return dict(logz_tree=float(res["logz"]), logzerr_tree=float(res["logzerr"]), ess=float(res["ess"]),
                    run_dir=None if self.logs is None else self.logs["run_dir"])
