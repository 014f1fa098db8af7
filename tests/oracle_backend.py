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


def subtract_nearby(upoints, maxradiussq):
    return orc.subtract_nearby(upoints, maxradiussq)


def maxradiussq_bootstrap(unormed, selected):
    return orc.maxradiussq_bootstrap(unormed, selected)


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

    def inside(self, pts):
        s = self.s
        pts = np.ascontiguousarray(pts, dtype=float)
        mask = orc.inside_ellipsoid(pts, s["ctr"], s["inv"], s["enlarge"])
        if not s["scan"] or not mask.any():
            return mask
        t = self._whiten(pts[mask])
        mask[mask] = orc.find_nearby(s["unormed"], t, s["r2"]) >= 0
        return mask


PATCHED = ["find_nearby", "count_nearby", "subtract_nearby", "maxradiussq_bootstrap", "compute_mean_pair_distance",
           "inside_ellipsoid", "affine_transform", "bootstrap_moments", "bootstrap_quadform_max", "DeviceRegion"]


def install(monkeypatch):
    import ultranest_amd.kernels as K
    import ultranest_amd.mlfriends as M
    g = globals()
    for name in PATCHED:
        monkeypatch.setattr(K, name, g[name])
        if hasattr(M, name):
            monkeypatch.setattr(M, name, g[name])
