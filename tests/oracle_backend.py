"""TEST-ONLY stand-in for ultranest_amd.kernels backed by the CPU oracle.

The product has no CPU path.  To exercise the HOST logic (layers, region classes, bootstrap
sharding, device-state diffing) on a machine without a GPU, the `backend` fixture in conftest.py
monkeypatches the kernel entry points of ultranest_amd.kernels with these oracle-backed
functions for the duration of a test.  The same tests run unpatched on the MI355X (`-m gpu`).
"""
import numpy as np

from oracle import oracle as orc


def find_nearby(apts, bpts, radiussq, nnearby):
    nnearby[:] = orc.find_nearby(np.asarray(apts, dtype=float), np.asarray(bpts, dtype=float), radiussq) \
        if len(apts) else -1


def count_nearby(apts, bpts, radiussq, nnearby):
    nnearby[:] = orc.count_nearby(apts, bpts, radiussq)


def cluster_labels(tpoints, radiussq, previous=None):
    tpoints = np.asarray(tpoints, dtype=float)
    ncl, labels, _ = orc.update_clusters(tpoints, tpoints, radiussq, previous)
    return int(ncl), np.asarray(labels)


def subtract_nearby(upoints, maxradiussq):
    return orc.subtract_nearby(upoints, maxradiussq)


def maxradiussq_bootstrap(unormed, selected, rows=None):
    if rows is None:
        return orc.maxradiussq_bootstrap(unormed, selected)
    # one rank's share under row-block sharding: the left-out points of rows [lo, hi) only
    sel = np.asarray(selected, dtype=bool)
    inrange = np.zeros(sel.shape[1], dtype=bool)
    inrange[int(rows[0]):int(rows[1])] = True
    r2 = np.zeros(len(sel))
    skipped = sel.all(axis=1) | ~sel.any(axis=1)
    for b, m in enumerate(sel):
        out = ~m & inrange
        if not skipped[b] and out.any():
            r2[b] = orc.maxradiussq(unormed[m], unormed[out])
    return r2, skipped


def compute_mean_pair_distance(pts, clusterids):
    return orc.mean_pair_distance(pts, clusterids)


def inside_ellipsoid(points, ctr, invcov, sqradius, return_q=False):
    return orc.inside_ellipsoid(points, ctr, invcov, sqradius, return_q=return_q)


def affine_transform(points, ctr, T, wrap_shift=None):
    pts = np.array(points, dtype=float)
    if wrap_shift is not None:
        for k, sh in enumerate(wrap_shift):
            if sh == sh:
                pts[:, k] = np.fmod(pts[:, k] + sh, 1)
    return orc.affine_transform(pts, ctr, T)


def bootstrap_moments(u, selected):
    sel = np.asarray(selected, dtype=bool)
    mean = np.array([u[m].mean(axis=0) for m in sel])
    cov = np.array([np.atleast_2d(np.cov(u[m] - u[m].mean(axis=0), rowvar=0)) for m in sel])
    return mean, cov


def bootstrap_quadform_max(u, selected, ctr, invcov):
    sel = np.asarray(selected, dtype=bool)
    out = np.empty(len(sel))
    for b, m in enumerate(sel):
        delta = u[~m] - ctr[b]
        out[b] = np.einsum('ij,jk,ik->i', delta, invcov[b], delta).max() if len(delta) else -np.inf
    return out


def bootstrap_factor(u, selected, scale):
    """The reference's own sequence for one round (mlfriends.pyx:1056-1066 with minvol = 0): covariance of the
    selected rows, LAPACK inverse, einsum over the left-out rows."""
    mean, cov = bootstrap_moments(u, selected)
    return bootstrap_quadform_max(u, selected, mean, np.linalg.inv(cov * scale))


class DeviceRegion(object):
    """Mimics the mlf_region handle semantics (set / update_point / thresholds / inside)."""
    n_full_sets = 0
    n_row_updates = 0

    def __init__(self):
        self.s = None

    def close(self):
        self.s = None

    def _whiten(self, w):
        s = self.s
        w = np.array(w, dtype=float).reshape((-1, len(s["ctr"])))
        if s["shift"] is not None:
            for k, sh in enumerate(s["shift"]):
                if sh == sh:
                    w[:, k] = np.fmod(w[:, k] + sh, 1)
        if s["kind"] == 0:
            return orc.affine_transform(w, s["lctr"], s["lmat"])
        return (w - s["lctr"]) / s["lmat"]

    def set(self, unormed, layer_kind, layer_ctr, layer_T, wrap_shift, ell_center, ell_invcov, enlarge, radiussq,
            use_scan=True, live_space=0):
        DeviceRegion.n_full_sets += 1
        self.live_space = live_space
        self.s = dict(unormed=None if unormed is None else np.array(unormed, dtype=float), kind=layer_kind,
                      lctr=None if layer_ctr is None else np.array(layer_ctr, dtype=float),
                      lmat=None if layer_T is None else np.array(layer_T, dtype=float),
                      shift=None if wrap_shift is None else np.array(wrap_shift, dtype=float),
                      ctr=np.array(ell_center, dtype=float), inv=np.array(ell_invcov, dtype=float),
                      enlarge=float(enlarge), r2=float(radiussq), scan=bool(use_scan))
        if use_scan and live_space:
            self.s["unormed"] = self._whiten(self.s["unormed"])

    def set_thresholds(self, enlarge, radiussq):
        self.s.update(enlarge=float(enlarge), r2=float(radiussq))

    def set_ellipsoid_center(self, ctr):
        self.s["ctr"] = np.array(ctr, dtype=float)

    def update_point(self, row, unormed_row):
        DeviceRegion.n_row_updates += 1
        self.s["unormed"][row] = self._whiten(unormed_row)[0] if self.live_space else unormed_row

    def update_points(self, rows, live_rows):
        assert len(set(int(r) for r in rows)) == len(rows)
        for row, value in zip(rows, np.asarray(live_rows, dtype=float)):
            self.update_point(int(row), value)

    def inside(self, pts):
        s = self.s
        pts = np.ascontiguousarray(pts, dtype=float)
        mask = orc.inside_ellipsoid(pts, s["ctr"], s["inv"], s["enlarge"])
        if not s["scan"] or not mask.any():
            return mask
        t = self._whiten(pts[mask])
        mask[mask] = orc.find_nearby(s["unormed"], t, s["r2"]) >= 0
        return mask


PATCHED = ["find_nearby", "count_nearby", "cluster_labels", "subtract_nearby", "maxradiussq_bootstrap", "compute_mean_pair_distance",
           "inside_ellipsoid", "affine_transform", "bootstrap_moments", "bootstrap_quadform_max", "bootstrap_factor",
           "DeviceRegion"]


def install(monkeypatch):
    import ultranest_amd.kernels as K
    import ultranest_amd.mlfriends as M
    g = globals()
    for name in PATCHED:
        monkeypatch.setattr(K, name, g[name])
        if hasattr(M, name):
            monkeypatch.setattr(M, name, g[name])
    install_stepfuncs(monkeypatch)


# ---- population step sampler (row f1) ------------------------------------------------------------
class OracleWalkers(object):
    """Test-only stand-in for ultranest_amd.popstepsampler._Walkers: the resident state as numpy
    arrays, advanced with the CPU oracle's restatement of the reference functions.  Lets the
    sampler's host logic (draw order, ring index, bookkeeping) run without a GPU."""

    def __init__(self, popsize, nsteps, ndim):
        from oracle import stepfuncs as osf
        self.osf = osf
        self.popsize, self.nsteps, self.ndim, self.nparams = popsize, nsteps, ndim, None
        P, G = popsize, nsteps + 1
        self.allu = np.full((P, G, ndim), np.nan)
        self.allL = np.full((P, G), np.nan)
        self.generation = np.zeros(P, dtype=np.int64) - 1
        self.currentt = np.full(P, np.nan)
        self.currentv = np.full((P, ndim), np.nan)
        self.left, self.right = np.zeros(P), np.zeros(P)
        self.sl, self.sr = np.zeros(P, dtype=bool), np.zeros(P, dtype=bool)
        self.currentp = None
        self.layer = None

    def begin(self, Lmin):
        self.osf.step_back(Lmin, self.allL, self.generation, self.currentt)
        flags = (~np.isfinite(self.currentt)) * 1 + self.sl * 2 + self.sr * 4
        return self.generation.copy(), flags.astype(np.uint8)

    def start(self, idx, rows, L):
        self.allu[idx, 0] = rows
        self.allL[idx, 0] = L
        self.generation[idx] = 0

    def points(self, idx):
        return self.allu[idx, self.generation[idx]]

    def brackets(self, idx, scale, v):
        self.left[idx], self.right[idx] = -scale, scale
        self.sl[idx] = self.sr[idx] = True
        self.currentt[idx] = 0
        self.currentv[idx] = v

    def set_layer(self, kind, ctr, mat, wrap, r2):
        self.layer = None if kind < 0 else (kind, np.array(ctr), np.array(mat), wrap, r2)

    def set_direction_data(self, **kw):
        raise AssertionError("device directions need the GPU")

    def propose(self, unif=None, rng=None, fetch=True):
        assert rng is None, "the Philox stream lives on the GPU"
        self.movable = self.generation < self.nsteps
        m = np.flatnonzero(self.movable)
        bis = ~np.logical_or(self.sl, self.sr)
        t = self.currentt.copy()
        draw = np.logical_and(bis, self.movable)
        scaled = (self.right - self.left) * unif
        t[draw] = (self.left + scaled)[draw]
        self.currentt = t
        u0 = np.ascontiguousarray(self.allu[m, self.generation[m]])
        unew = np.empty_like(u0)
        # named temporaries: the raw addresses must stay alive for the duration of the call
        v, lo, hi = (np.ascontiguousarray(a[m]) for a in (self.currentv, self.left, self.right))
        sl, sr = np.ascontiguousarray(self.sl[m]).view(np.uint8), np.ascontiguousarray(self.sr[m]).view(np.uint8)
        tm = np.ascontiguousarray(self.currentt[m])
        self.osf.lib().orc_evolve_propose(u0.ctypes.data, v.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                          sl.ctypes.data, sr.ctypes.data, tm.ctypes.data, len(m), self.ndim,
                                          unew.ctypes.data)
        self.unew = np.full((self.popsize, self.ndim), np.nan)
        self.unew[m] = unew
        self.acceptable = np.zeros(self.popsize, dtype=bool)
        self.acceptable[m] = self.osf.within_unit_cube(unew) if len(m) else False
        return self.unew[self.acceptable] if fetch else None

    def finish(self, Lmin, pnew, Lnew, ringindex):
        P = self.popsize
        if len(Lnew):
            self.nparams = pnew.shape[1]
        elif self.nparams is None:
            self.nparams = self.ndim
        if self.currentp is None:
            self.currentp = np.full((P, self.nparams), np.nan)
        m = self.movable
        sright, bis = self.osf.evolve_prepare(self.sl[m], self.sr[m])
        t, lo, hi = (np.ascontiguousarray(a[m]) for a in (self.currentt, self.left, self.right))
        sl, sr = np.ascontiguousarray(self.sl[m]), np.ascontiguousarray(self.sr[m])
        success_m = np.zeros(int(m.sum()), dtype=bool)
        self.osf.evolve_update(self.acceptable[m], Lnew, Lmin, sright, bis, t, lo, hi, sl, sr, success_m)
        self.currentt[m], self.left[m], self.right[m], self.sl[m], self.sr[m] = t, lo, hi, sl, sr
        success = np.zeros(P, dtype=bool)
        success[m] = success_m
        Lfull, pfull = np.full(P, np.nan), np.full((P, self.nparams), np.nan)
        Lfull[self.acceptable] = Lnew
        if len(Lnew):
            pfull[self.acceptable] = pnew
        uorig = self.allu[success, self.generation[success]]
        self.generation[success] += 1
        self.allu[success, self.generation[success]] = self.unew[success]
        self.allL[success, self.generation[success]] = Lfull[success]
        self.currentp[success] = pfull[success]
        nfar, sumlog = 0.0, 0.0
        if self.layer is not None and success.any():
            kind, ctr, mat, wrap, r2 = self.layer
            tr = (lambda x: np.dot(x - ctr, mat)) if kind == 0 else (lambda x: (x - ctr) / mat)
            d2 = ((tr(uorig) - tr(self.unew[success]))**2).sum(axis=1)
            nfar = float((d2 > r2).sum())
            sumlog = float(np.log(d2**0.5 / r2**0.5 + 1e-10).sum())
        rec = dict(found=bool(self.generation[ringindex] == self.nsteps), L=np.nan, left=self.left[ringindex],
                   right=self.right[ringindex], nc=int(self.acceptable.sum()), nmovable=int(m.sum()),
                   nsuccess=int(success.sum()), nfar=nfar, sumlog=sumlog, u=None, p=None)
        if rec["found"]:
            rec["L"] = self.allL[ringindex, self.nsteps]
            rec["u"] = self.allu[ringindex, self.nsteps].copy()
            rec["p"] = self.currentp[ringindex].copy()
            self.generation[ringindex] = -1
            self.currentt[ringindex] = np.nan
            self.allu[ringindex] = np.nan
            self.allL[ringindex] = np.nan
        return rec

    def export(self):
        return dict(allu=self.allu.copy(), allL=self.allL.copy(), generation=self.generation.copy(),
                    currentt=self.currentt.copy(), currentv=self.currentv.copy(), current_left=self.left.copy(),
                    current_right=self.right.copy(), searching_left=self.sl.copy(), searching_right=self.sr.copy())


def install_stepfuncs(monkeypatch):
    """Route the host-array step functions and the resident walker handle through the oracle."""
    import ultranest_amd.popstepsampler as P
    import ultranest_amd.stepfuncs as S
    from oracle import stepfuncs as osf
    for name in ("within_unit_cube", "evolve_update", "evolve", "step_back", "update_vectorised_slice_sampler",
                 "unitcube_line_intersection", "row_dist2"):
        monkeypatch.setattr(S, name, getattr(osf, name))
        if hasattr(P, name):
            monkeypatch.setattr(P, name, getattr(osf, name))
    monkeypatch.setattr(P, "_Walkers", OracleWalkers)
