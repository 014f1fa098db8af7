"""Opt-in device-resident rebuild (ultranest_amd.device_rebuild) against the default, reference-bit-preserving rebuild:
same cluster labels, same `np.random` consumption, T / cov / unormed / radius / enlargement / ellipsoid within the
tolerance class of the mode (1e-10), same accept decision, masks of a proposal batch equal for the region as built
(checked against the CPU oracle).  Reference: integrator.py:2055-2122, mlfriends.pyx:666-710, 827-850, 1017-1070, 1213-1237."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

pytestmark = pytest.mark.gpu


def _equal_or_adjacent_binary32(a, b):
    fa, fb = np.float32(a), np.float32(b)
    assert float(fa) == a and float(fb) == b, "radii are binary32 values"
    assert fa == fb or np.nextafter(fa, np.float32(np.inf)) == fb or np.nextafter(fa, np.float32(-np.inf)) == fb, (a, b)


def _pair(n, d, seed, layer_name, blobs=1):
    import ultranest_amd.mlfriends as M
    from ultranest_amd.harness import RegionUpdater
    rs = np.random.RandomState(seed)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    if blobs == 2:
        u[: n // 2] += 0.2
        u[n // 2:] -= 0.2
    layer_class = getattr(M, layer_name)
    out = []
    for device_resident in (False, True):
        upd = RegionUpdater(d, region_class=M.MLFriends, transform_layer_class=layer_class, device_resident=device_resident)
        np.random.seed(11)
        upd.update(u, nbootstraps=30, minvol=0.)
        rs2 = np.random.RandomState(seed + 1)
        u2 = u.copy()
        u2[: n // 10] = u[: n // 10] * 0.9 + 0.05 + 0.002 * rs2.normal(size=(n // 10, d))
        changed = upd.update(u2, nbootstraps=30, minvol=0.)
        out.append((upd, changed, np.random.get_state()[2], np.random.uniform()))
    return u2, out


@pytest.mark.parametrize("n,d,layer_name,blobs", [(400, 5, "LocalAffineLayer", 1), (1000, 10, "AffineLayer", 1),
                                                   (600, 4, "AffineLayer", 2), (2000, 20, "LocalAffineLayer", 1),
                                                   (4000, 50, "LocalAffineLayer", 1)])
def test_device_resident_rebuild_matches_default(n, d, layer_name, blobs):
    from oracle import oracle as orc
    u2, ((a, ch_a, pos_a, nxt_a), (b, ch_b, pos_b, nxt_b)) = _pair(n, d, 3, layer_name, blobs)
    assert b._device_rebuild is not None, "the device path did not run"
    assert ch_a == ch_b
    assert (pos_a, nxt_a) == (pos_b, nxt_b), "np.random consumed differently"
    la, lb = a.transformLayer, b.transformLayer
    assert la.nclusters == lb.nclusters
    assert np.array_equal(la.clusterids, lb.clusterids)
    tol = dict(rtol=1e-10, atol=1e-12)
    assert np.allclose(la.ctr, lb.ctr, **tol)
    assert np.allclose(la.cov, lb.cov, **tol)
    assert np.allclose(abs(la.logvolscale - lb.logvolscale), 0, atol=1e-9)
    # eigenvectors are defined up to sign / order within the tolerance: compare what they generate
    assert np.allclose(la.T @ la.T.T, lb.T @ lb.T.T, rtol=1e-8, atol=1e-10 * np.abs(la.T @ la.T.T).max())
    assert np.allclose(lb.T @ lb.invT, np.eye(d), atol=1e-9)
    ra, rb = a.region, b.region
    assert np.allclose(np.asarray(ra.u), np.asarray(rb.u), rtol=0, atol=0)
    assert np.allclose(rb.unormed, lb.transform(np.asarray(rb.u)), rtol=1e-9, atol=1e-11)
    # the radius is a binary32-rounded maximum of distances between whitened points that agree to ~1e-12: the same
    # binary32 value or one of its neighbours (north_star: 1e-10 on radii -- met up to the one rounding the reference's
    # own `cdef float` return applies, mlfriends.pyx:188, 224)
    _equal_or_adjacent_binary32(ra.maxradiussq, rb.maxradiussq)
    # the factor comes from the same device kernel on the same points and draws in both paths (measured: identical,
    # scripts/enlargement_tolerance_probe.py); north_star's class for it is 1e-10
    assert abs(ra.enlarge - rb.enlarge) <= 1e-12 * ra.enlarge
    assert np.allclose(ra.ellipsoid_center, rb.ellipsoid_center, **tol)
    assert np.allclose(ra.ellipsoid_cov, rb.ellipsoid_cov, **tol)
    assert np.allclose(ra.ellipsoid_invcov, rb.ellipsoid_invcov, rtol=1e-8, atol=1e-8 * np.abs(ra.ellipsoid_invcov).max())
    assert np.allclose(ra.estimate_volume(), rb.estimate_volume(), atol=1e-6)
    assert np.allclose(rb.bbox_lo, rb.unormed.min(axis=0)) and np.allclose(rb.bbox_hi, rb.unormed.max(axis=0))
    # the region as built answers exactly like the oracle on ITS OWN attributes
    rs = np.random.RandomState(9)
    pts = np.asarray(rb.u)[rs.randint(n, size=3000)] + 0.01 * rs.normal(size=(3000, d))
    got = rb.inside(pts)
    # live points whitened on the device (FMA chain): the oracle gets the same layer and whitens the same way
    want = orc.region_inside(pts, orc.affine_transform(np.asarray(rb.u), lb.ctr, lb.T), lb.ctr, lb.T, rb.ellipsoid_center,
                             rb.ellipsoid_invcov, rb.enlarge, rb.maxradiussq)
    assert np.array_equal(got, want)
    assert rb.inside(np.asarray(rb.u)).all() == ra.inside(np.asarray(ra.u)).all()
    # in-place replacement of a live point afterwards is tracked like on any region
    rb.u[3] = rb.u[5]
    assert rb.inside(np.asarray(rb.u)[3:4])[0] == rb.inside(np.asarray(rb.u)[5:6])[0]


def test_cluster_labels_equal_update_clusters(oracle):
    """mlf_cluster_labels replays update_clusters' growth rounds (the oracle's: one neighbour scan per round, mlfriends.pyx:
    275-343): labels, numbering, carried-over seeds."""
    import ctypes
    from ultranest_amd import _lib
    update_clusters = oracle.update_clusters
    rs = np.random.RandomState(4)
    for case in range(6):
        n, d = 300 + 50 * case, 3
        centres = rs.uniform(size=(4, d))
        t = centres[rs.randint(4, size=n)] + 0.02 * rs.normal(size=(n, d))
        r2 = 0.0015 * (1 + case)
        prev = None if case % 2 == 0 else rs.randint(1, 6, size=n).astype(np.int64)
        ncl, ids, _ = update_clusters(t, t, r2, prev)
        labels = np.empty(n, dtype=np.int64)
        k = ctypes.c_int64(0)
        _lib.check(_lib.lib().mlf_cluster_labels(_lib.ptr(t), n, d, r2, _lib.ptr(prev), _lib.ptr(labels), ctypes.byref(k)))
        assert np.array_equal(ids, labels), case
        assert ncl == k.value


@pytest.mark.parametrize("lname", ["affine", "local"])
def test_device_resident_rebuild_against_the_reference_generations(lname, golden):
    """The reference's own clustered fixture (eggboxregion.txt) through two layer generations as the driver iterates them
    (integrator.py:2068-2091), recorded from the real reference (golden g5: make_golden.py g456): `create_new` of the
    device-resident path against the recorded next layer -- cluster count and ids exactly, centre / covariance / log volume
    scale to 1e-10, T through what it generates -- and the region it builds, bootstrapped with the SAME draws the
    recording used for the next generation (RandomState(77 + gen + 1): the global stream seeded alike), against the
    recorded radius (equal or adjacent binary32) and enlargement (1e-10: north_star's class; measured 7e-16)."""
    import ultranest_amd.mlfriends as M
    from ultranest_amd import device_rebuild
    g = golden("g456_region")
    u = np.ascontiguousarray(g["g5_u"])
    n, d = u.shape
    cls = M.AffineLayer if lname == "affine" else M.LocalAffineLayer
    layer = cls()
    layer.optimize(u, u)
    rebuild = device_rebuild.DeviceRebuild()
    assert device_rebuild.supported(layer, M.MLFriends, d, 0., 1)
    for gen in (1, 2):
        key = "g5_%s%d_" % (lname, gen)
        r_prev = float(g[key + "r_f"][0])              # the radius the reference handed to create_new
        np.random.seed(77 + gen + 1)                   # the draws of the NEXT generation's compute_enlargement
        nxt, region, contains = rebuild.next_region(u, layer, r_prev, 30)
        assert nxt.nclusters == int(g[key + "nclusters"])
        assert np.array_equal(nxt.clusterids, g[key + "ids"])
        assert np.allclose(nxt.ctr, g[key + "ctr"], rtol=1e-10, atol=1e-13)
        assert np.allclose(nxt.cov, g[key + "cov"], rtol=1e-10, atol=1e-10 * np.abs(g[key + "cov"]).max())
        assert abs(nxt.logvolscale - float(g[key + "logvolscale"])) <= 1e-9
        Tg = g[key + "T"]
        assert np.allclose(nxt.T @ nxt.T.T, Tg @ Tg.T, rtol=1e-8, atol=1e-9 * np.abs(Tg @ Tg.T).max())
        if gen == 1:                                   # generation 2's (r, f) were recorded on THIS layer with these draws
            r_next, f_next = (float(v) for v in g["g5_%s2_r_f" % lname])
            _equal_or_adjacent_binary32(region.maxradiussq, r_next)
            assert abs(region.enlarge - f_next) <= 1e-10 * f_next     # measured against the reference's recording: 7e-16
        assert contains
        layer = nxt
