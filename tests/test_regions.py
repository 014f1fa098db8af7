"""Region / layer classes (ultranest_amd.mlfriends namespace): the assertions of the reference's
own hot-path tests (tests/test_regionsampling.py, test_clustering.py, test_transforms.py,
test_run.py:62-72 -- restated, cited per test) plus golden comparisons against the real
reference (G3, G5, G6).

Every test takes the `backend` fixture: on CPU the kernel entry points are monkeypatched with
the oracle (host logic only, `-m "not gpu"`); on the MI355X they run unpatched (`-m gpu`).
"""
import numpy as np
import pytest

import inputs


def _mlf():
    import ultranest_amd.mlfriends as m
    return m


def _build(points, layer_cls, region_cls=None, nbootstraps=30, **layer_kw):
    m = _mlf()
    layer = layer_cls(**layer_kw)
    layer.optimize(points, points)
    region = (region_cls or m.MLFriends)(points, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=nbootstraps)
    region.create_ellipsoid()
    return layer, region


# reference tests/test_regionsampling.py:11-48 (scaling) and :51-88 (affine)
@pytest.mark.parametrize("kind", ["scaling", "affine"])
def test_region_sampling_methods(kind, backend):
    m = _mlf()
    np.random.seed(1)
    if kind == "scaling":
        upoints = np.random.uniform(0.2, 0.5, size=(1000, 2))
        upoints[:, 1] *= 0.1
        layer, region = _build(upoints, m.ScalingLayer, wrapped_dims=[])
        box = ((0.15, 0.25), (0.015, 0.025), (0.45, 0.55), (0.045, 0.055))
        min_inside = 0.99
    else:
        upoints = np.random.uniform(size=(1000, 2))
        upoints[:, 1] *= 0.5
        layer, region = _build(upoints, m.AffineLayer, wrapped_dims=[])
        box = ((-1e-300, 0.1), (-1e-300, 0.1), (0.95, 1 + 1e-12), (0.45 - 1e-12, 0.55))
        min_inside = 1.0
    assert layer.nclusters == 1
    assert np.allclose(region.unormed, region.transformLayer.transform(upoints))
    assert region.inside(upoints).all(), "live points should lie near live points"
    assert len(region.sampling_methods) == 4
    for method in region.sampling_methods:
        newpoints = method(nsamples=4000)
        lo1, lo2 = newpoints.min(axis=0)
        hi1, hi2 = newpoints.max(axis=0)
        for val, (a, b) in zip((lo1, lo2, hi1, hi2), box):
            assert a < val < b, (method.__name__, lo1, lo2, hi1, hi2)
        assert region.inside(newpoints).mean() >= min_inside, method.__name__
    # the d == 0 <= r2 pin (:46-48, :86-88)
    region.maxradiussq = 1e-90
    assert region.inside(upoints).all(), "live points should lie very near themselves"
    # sample() switches method only on an empty batch (:1162-1184)
    region.maxradiussq, _ = region.compute_enlargement(nbootstraps=5)
    before = region.current_sampling_method
    assert len(region.sample(nsamples=500)) > 0 and region.current_sampling_method == before


# reference tests/test_regionsampling.py:90-110
def test_region_ellipsoid_matches_einsum(backend):
    m = _mlf()
    np.random.seed(1)
    points = np.random.uniform(0.4, 0.6, size=(1000, 2))
    points[:, 1] *= 0.5
    layer, region = _build(points, m.AffineLayer, wrapped_dims=[])
    assert layer.nclusters == 1
    bpts = np.random.uniform(size=(100, 2))
    d = bpts - region.ellipsoid_center
    expect = np.einsum('ij,jk,ik->i', d, region.ellipsoid_invcov, d) <= region.enlarge
    assert np.array_equal(region.inside_ellipsoid(bpts), expect)
    for cls in (m.RobustEllipsoidRegion, m.SimpleRegion):
        _, reg = _build(points, m.AffineLayer, region_cls=cls, wrapped_dims=[])
        assert reg.maxradiussq == 1e300
        assert reg.inside(points).all()
        assert len(reg.sampling_methods) == 2
        for method in reg.sampling_methods:
            pts = method(nsamples=2000)
            assert len(pts) > 0 and reg.inside(pts).all()
        assert np.isfinite(reg.estimate_volume())


# reference tests/test_regionsampling.py:113-142
def test_mean_pair_distance_double_loop(backend):
    m = _mlf()
    np.random.seed(1)
    points = np.random.uniform(size=(60, 3))
    ids = np.random.randint(0, 3, size=60).astype(m.int_dtype)
    total, npairs = 0.0, 0
    for j in range(60):
        if ids[j] == 0:
            continue
        for i in range(j):
            if ids[i] == ids[j]:
                total += ((points[i] - points[j])**2).sum()**0.5
                npairs += 1
    assert np.isclose(m.compute_mean_pair_distance(points, ids), total / npairs, rtol=1e-13)


# reference tests/test_clustering.py:12-35 and :38-48
def test_clustering_counts(backend, golden):
    m = _mlf()
    np.random.seed(1)
    blob = lambda c: np.random.normal(c, 0.01, size=(100, 2))  # noqa: E731
    points = np.vstack((blob(0.2), blob(0.5), blob(0.8)))
    nclusters, ids, overlapped = m.update_clusters(points, points, 0.03**2)
    assert nclusters == 3 and set(np.unique(ids)) == {1, 2, 3}
    assert np.abs(overlapped.mean(axis=0)).max() < 0.01
    nclusters, ids, overlapped = m.update_clusters(points, points, 1.0)
    assert nclusters == 1 and (ids == 1).all() and overlapped is points
    g = golden("g2_clusters")
    pts = g["ref_clusters2_pts"]
    nclusters, ids, _ = m.update_clusters(pts, pts, float(g["ref_clusters2_r2"]))
    assert nclusters == int(g["ref_clusters2_nclusters"]) and np.array_equal(ids, g["ref_clusters2_ids"])
    for tag, n, d in (("a", 300, 4),):
        u = inputs.two_blobs(200, n, d)
        nclusters, ids, overlapped = m.update_clusters(u, u, float(g[tag + "_r2"]))
        assert nclusters == int(g[tag + "_nclusters"])
        assert np.array_equal(ids, g[tag + "_ids"]) and np.array_equal(overlapped, g[tag + "_overlapped"])
        nclusters2, ids2, _ = m.update_clusters(u, u, float(g[tag + "_r2b"]), ids)
        assert nclusters2 == int(g[tag + "_nclusters2"]) and np.array_equal(ids2, g[tag + "_ids2"])


# reference tests/test_clustering.py:58-78
def test_subtract_nearby_bounds(backend):
    m = _mlf()
    np.random.seed(1)
    u = np.random.uniform(size=(400, 2))
    far = m.subtract_nearby(u, 1e-300)
    assert np.array_equal(far, np.zeros_like(u))       # only itself is near: u - u
    overlapped = m.subtract_nearby(u, 1.0)
    assert np.all(np.abs(overlapped) < 0.6)


# reference tests/test_clustering.py:81-99 (the reference's own pins on its own fixture)
def test_clusteringcase_eggbox(backend, golden):
    m = _mlf()
    g = golden("g3_bootstrap")
    points = g["eggbox_pts"]
    layer = m.ScalingLayer()
    layer.optimize(points, points)
    for seed in range(10):
        np.random.seed(seed)
        region = m.MLFriends(points, layer)
        maxr = region.compute_maxradiussq(nbootstraps=30)
        assert 1e-10 < maxr < 6e-10
        assert maxr == g["eggbox_maxr"][seed]                     # exact vs the real reference
        nclusters, clusteridxs, _ = m.update_clusters(points, points, maxr)
        assert 14 < nclusters < 20 and nclusters == g["eggbox_nclusters"][seed]


# reference tests/test_transforms.py:30, :78, :89-124
def test_layer_roundtrips_and_wraps(backend, golden):
    m = _mlf()
    np.random.seed(2)
    pts = np.random.uniform(0.1, 0.9, size=(300, 3))
    s = m.ScalingLayer()
    s.optimize(pts, pts)
    assert np.array_equal(s.untransform(s.transform(pts)), pts) or np.allclose(s.untransform(s.transform(pts)), pts, rtol=0, atol=1e-15)
    a = m.AffineLayer()
    a.optimize(pts, pts)
    assert np.allclose(a.untransform(a.transform(pts)), pts)
    assert a.transform(pts[0]).shape == (3,)
    # wrapped axis: golden from the reference
    g = golden("g456_region")
    uw = g["g5_wrap_u"]
    w = m.AffineLayer(wrapped_dims=[0])
    w.optimize(uw, uw)
    assert np.allclose(w.wrap_cuts, g["g5_wrap_cuts"], rtol=1e-15)
    assert np.allclose(w.transform(uw), g["g5_wrap_t"], rtol=1e-9, atol=1e-12) or \
        np.allclose(np.abs(w.transform(uw)), np.abs(g["g5_wrap_t"]), rtol=1e-9, atol=1e-12)
    assert np.allclose(w.untransform(w.transform(uw)), uw)
    assert np.allclose(w.unwrap(w.wrap(uw)), uw)


# reference tests/test_run.py:62-72
def test_bootstrap_linearly_dependent_points_raise(backend):
    m = _mlf()
    np.random.seed(1)
    points = np.random.uniform(0.3, 0.7, size=(100, 3))
    points[:, 2] = points[:, 0]                     # singular covariance
    layer = m.ScalingLayer()
    layer.optimize(points, points)
    region = m.MLFriends(points, layer)
    with pytest.raises(np.linalg.LinAlgError):
        region.compute_enlargement(nbootstraps=30)
    with pytest.raises(ValueError):
        m.MLFriends(np.array([[0.5, 1.2]]), layer)   # u outside the unit cube (:938-939)
    with pytest.raises(FloatingPointError):
        m.RobustEllipsoidRegion(points[:3], layer).compute_enlargement(nbootstraps=3)   # N < d+1 (:1414)


# G3 at the class level: same RandomState stream as the reference run that made the fixture
@pytest.mark.parametrize("case", [0, 1, 3])
def test_compute_enlargement_golden(case, backend, golden):
    m = _mlf()
    import ultranest_amd.regions as R
    g = golden("g3_bootstrap")
    name, n, d, B = inputs.BOOTSTRAP_CASES[case]
    u = inputs.live_points(300 + case, n, d)
    region = m.MLFriends(u, m.ScalingLayer())
    assert np.array_equal(region.unormed, u)
    r, f = region.compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + case))
    assert r == g[name + "_r"].max()                                  # exact, float32-rounded
    np.testing.assert_allclose(f, g[name + "_f"].max(), rtol=1e-10)
    assert isinstance(r, float) and isinstance(f, float)
    rs = np.random.RandomState(900 + case)
    for b in range(3):                                                # per-round, one draw each
        rb, fb = region.compute_enlargement(nbootstraps=1, rng=rs)
        assert rb == g[name + "_r"][b]
        np.testing.assert_allclose(fb, g[name + "_f"][b], rtol=1e-10)
    old = R.STRICT_HOST_MOMENTS
    R.STRICT_HOST_MOMENTS = True
    try:
        _, f2 = region.compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + case))
        np.testing.assert_allclose(f2, g[name + "_f"].max(), rtol=1e-13)
    finally:
        R.STRICT_HOST_MOMENTS = old
    r_rob, f_rob = m.RobustEllipsoidRegion(u, m.ScalingLayer()).compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + case))
    assert r_rob == g[name + "_rob"][0] == 1e300
    np.testing.assert_allclose(f_rob, g[name + "_rob"][1], rtol=1e-10)
    r_sim, f_sim = m.SimpleRegion(u, m.ScalingLayer()).compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + case))
    assert r_sim == 1e300
    np.testing.assert_allclose(f_sim, g[name + "_sim"][1], rtol=1e-12)
    fw = m.WrappingEllipsoid(u).compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + case))
    np.testing.assert_allclose(fw, float(g[name + "_wrap"]), rtol=1e-10)


# G4 + G6 through the classes (AffineLayer + MLFriends end to end, LAPACK on this host)
@pytest.mark.parametrize("name,n,d,p,case", [("c1", 400, 5, 3000, 0), ("c2", 2000, 20, 3000, 1)])
def test_region_pipeline_golden(name, n, d, p, case, backend, golden):
    m = _mlf()
    g = golden("g456_region")
    u = inputs.live_points(400 + case, n, d)
    layer = m.AffineLayer()
    layer.optimize(u, u)
    np.testing.assert_allclose(layer.ctr, g[name + "_layer_ctr"], rtol=1e-13)
    np.testing.assert_allclose(layer.logvolscale, float(g[name + "_layer_logvolscale"]), rtol=1e-10)
    np.testing.assert_allclose(layer.T @ layer.T.T, g[name + "_layer_T"] @ g[name + "_layer_T"].T, rtol=1e-8, atol=1e-9)
    region = m.MLFriends(u, layer)
    r, f = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(950 + case))
    np.testing.assert_allclose(r, float(g[name + "_r2"]), rtol=1e-6)      # float32 ulp: T is LAPACK-dependent
    np.testing.assert_allclose(f, float(g[name + "_enlarge"]), rtol=1e-10)
    region.maxradiussq, region.enlarge = float(g[name + "_r2"]), float(g[name + "_enlarge"])
    region.create_ellipsoid(minvol=0.0)
    np.testing.assert_allclose(region.ellipsoid_center, g[name + "_ell_center"], rtol=1e-13)
    np.testing.assert_allclose(region.ellipsoid_cov, g[name + "_ell_cov"], rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(region.ellipsoid_invcov, g[name + "_ell_invcov"], rtol=1e-8, atol=1e-6)
    np.testing.assert_allclose(np.sort(region.ellipsoid_axlens), np.sort(g[name + "_ell_axlens"]), rtol=1e-9)
    np.testing.assert_allclose(region.estimate_volume(), float(g[name + "_volume"]), rtol=1e-9)
    pts = inputs.proposal_mix(500 + case, u, p, shell_q=float(g[name + "_enlarge"]))
    mask = region.inside(pts)
    assert np.array_equal(mask, np.unpackbits(g[name + "_mask"])[:p].astype(bool))
    region.maxradiussq = float(g[name + "_r2_tight"])
    assert np.array_equal(region.inside(pts), np.unpackbits(g[name + "_mask_tight"])[:p].astype(bool))
    assert region.inside(u).all()
    # strict mode (host np.dot whitening + GPU scan) gives the same masks
    import ultranest_amd.regions as R
    R.STRICT_HOST_TRANSFORM = True
    try:
        assert np.array_equal(region.inside(pts), np.unpackbits(g[name + "_mask_tight"])[:p].astype(bool))
    finally:
        R.STRICT_HOST_TRANSFORM = False


# G5: layer generations on the reference's eggboxregion fixture, as the driver iterates them
@pytest.mark.parametrize("lname", ["affine", "local", "gap", "scaling"])
def test_layer_generations_golden(lname, backend, golden):
    m = _mlf()
    g = golden("g456_region")
    u = g["g5_u"]
    cls = dict(affine=m.AffineLayer, local=m.LocalAffineLayer, gap=m.MaxPrincipleGapAffineLayer, scaling=m.ScalingLayer)[lname]
    layer = cls()
    layer.optimize(u, u)
    for gen in (1, 2):
        key = "g5_%s%d_" % (lname, gen)
        region = m.MLFriends(u, layer)
        r, f = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(77 + gen))
        np.testing.assert_allclose([r, f], g[key + "r_f"], rtol=2e-6)       # r is f32-rounded downstream of LAPACK
        nxt = layer.create_new(u, float(g[key + "r_f"][0]))
        assert nxt.nclusters == int(g[key + "nclusters"])
        assert np.array_equal(nxt.clusterids, g[key + "ids"])
        np.testing.assert_allclose(nxt.logvolscale, float(g[key + "logvolscale"]), rtol=1e-8)
        if lname == "scaling":
            np.testing.assert_allclose(nxt.mean, g[key + "mean"], rtol=1e-13)
            np.testing.assert_allclose(nxt.std, g[key + "std"], rtol=1e-12)
        else:
            np.testing.assert_allclose(nxt.ctr, g[key + "ctr"], rtol=1e-13)
            np.testing.assert_allclose(nxt.cov, g[key + "cov"], rtol=1e-9, atol=1e-22)
        layer = nxt


# driver-style in-place mutation (reference integrator.py:2749-2765) reaches the device copy
def test_inplace_live_point_replacement(backend):
    m = _mlf()
    np.random.seed(4)
    u = inputs.live_points(5, 300, 4)
    layer, region = _build(u.copy(), m.AffineLayer)
    pts = inputs.proposal_mix(6, u, 2000, shell_q=region.enlarge)
    region.inside(pts)
    for it in range(3):
        worst = 7 + it
        newu = pts[region.inside(pts)][it]
        region.u[worst] = newu
        region.unormed[worst] = layer.transform(newu)
        region.ellipsoid_center = np.mean(region.u, axis=0)
        layer.clusterids[worst] = 0
        got = region.inside(pts)
        fresh = m.MLFriends(region.u.copy(), layer)
        fresh.maxradiussq, fresh.enlarge = region.maxradiussq, region.enlarge
        fresh.ellipsoid_center = region.ellipsoid_center
        fresh.ellipsoid_invcov = region.ellipsoid_invcov
        assert np.array_equal(got, fresh.inside(pts))
    region.maxradiussq = None                       # driver invalidates the radius (:2827)
    with pytest.raises(TypeError):
        region.inside(pts)


# reference tests/test_regionsampling.py:145-192
def test_ellipsoids_contain_their_points(backend):
    """Every region class contains its own live points after compute_enlargement + create_ellipsoid, including a
    WrappingEllipsoid with a constant (zero-variance) coordinate and a 1-d one."""
    import ultranest_amd.mlfriends as m
    np.random.seed(4)
    tpoints = np.random.uniform(0.4, 0.6, size=(1000, 1))
    tregion = m.WrappingEllipsoid(tpoints)
    tregion.enlarge = tregion.compute_enlargement(nbootstraps=30)
    tregion.create_ellipsoid()
    assert tregion.inside(tpoints).all()
    for umax in 0.6, 0.5:
        points = np.random.uniform(0.4, 0.6, size=(1000, 3))
        points = points[points[:, 0] < umax]
        tpoints = points * 10
        tpoints[:, 0] = np.floor(tpoints[:, 0])      # umax = 0.5: this coordinate is constant
        layer = m.AffineLayer(wrapped_dims=[])
        layer.optimize(points, points)
        for cls in (m.MLFriends, m.RobustEllipsoidRegion, m.SimpleRegion):
            region = cls(points, layer)
            region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=30)
            region.create_ellipsoid()
            inside = region.inside(points)
            assert inside.shape == (len(points),), (cls, inside.shape, points.shape)
            assert inside.all(), cls
        tregion = m.WrappingEllipsoid(tpoints)
        if umax == 0.5:
            assert list(tregion.variable_dims) == [False, True, True]
        tregion.enlarge = tregion.compute_enlargement(nbootstraps=30)
        tregion.create_ellipsoid()
        inside = tregion.inside(tpoints)
        assert inside.shape == (len(tpoints),), (inside.shape, tpoints.shape)
        assert inside.all()
        moved = tpoints.copy()
        moved[:, 0] += 1                             # off the constant coordinate: outside
        if umax == 0.5:
            assert not tregion.inside(moved).any()


def test_ellipsoid_started_by_the_bootstrap_is_the_one_create_ellipsoid_computes(backend):
    """The bootstrap starts `create_ellipsoid`'s host LAPACK on the worker thread (regions.MLFriends._start_ellipsoid_parts);
    `create_ellipsoid` takes that result only for the SAME live points and the same `minvol` -- in every case the
    attributes are those of a region that computes them on the spot (reference mlfriends.pyx:1213-1237)."""
    import ultranest_amd.mlfriends as M
    from ultranest_amd import regions as R
    rs = np.random.RandomState(3)
    u = 0.5 + 0.05 * rs.normal(size=(300, 4))
    layer = M.AffineLayer()
    layer.optimize(u, u)

    def attributes(region):
        return [np.array(getattr(region, k)) for k in ("ellipsoid_center", "ellipsoid_cov", "ellipsoid_invcov", "ellipsoid_axlens",
                                                      "ellipsoid_axes", "ellipsoid_axes_T", "ellipsoid_inv_axlens", "ellipsoid_inv_axes")]

    def on_the_spot(points, minvol):
        other = M.MLFriends(np.array(points), layer)
        other.enlarge = 1.0
        other.create_ellipsoid(minvol=minvol)       # nothing was started for this region
        assert other not in R._ELLIPSOID_JOBS
        return attributes(other)

    region = M.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=5, rng=np.random.RandomState(1))
    assert region in R._ELLIPSOID_JOBS                          # on its way (or done)
    region.create_ellipsoid()
    assert region not in R._ELLIPSOID_JOBS
    assert all(np.array_equal(a, b) for a, b in zip(attributes(region), on_the_spot(u, 0.0)))
    # live points written to between the bootstrap and the ellipsoid: the started result is not taken
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=5, rng=np.random.RandomState(1))
    region.u[7] = np.clip(region.u[7] + 0.01, 1e-6, 1 - 1e-6)
    region.create_ellipsoid()
    assert all(np.array_equal(a, b) for a, b in zip(attributes(region), on_the_spot(region.u, 0.0)))
    # another minvol than the bootstrap's: computed again
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=5, minvol=0.0, rng=np.random.RandomState(1))
    region.create_ellipsoid(minvol=1e-3)
    assert all(np.array_equal(a, b) for a, b in zip(attributes(region), on_the_spot(region.u, 1e-3)))
    # an error of the LAPACK calls surfaces in create_ellipsoid, as it would without the worker
    flat = np.array(u)
    flat[:, 0] = 0.25
    bad = M.MLFriends(flat, layer)
    bad.enlarge = 1.0
    bad._start_ellipsoid_parts(0.0)
    with np.errstate(all="raise"):
        with pytest.raises((np.linalg.LinAlgError, FloatingPointError)):
            bad.create_ellipsoid()


def test_host_worker_is_per_process():
    """a forked child must not inherit the parent's worker (its thread does not exist there: a job would never run)"""
    import os
    from ultranest_amd import layers
    parent = layers.host_worker()
    assert parent.submit(os.getpid).result() == os.getpid()
    if not hasattr(os, "fork"):
        return
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:      # child: a job submitted here must run
        ok = b"0"
        try:
            child = layers.host_worker()
            if child is not parent and child.submit(lambda: 41 + 1).result(timeout=20) == 42:
                ok = b"1"
        finally:
            os.write(w, ok)
            os._exit(0)
    os.close(w)
    assert os.read(r, 1) == b"1"
    os.waitpid(pid, 0)


def test_update_clusters_both_label_routes_agree(backend, monkeypatch):
    """layers.update_clusters takes its labels from one all-pairs pass (kernels.cluster_labels) up to 32768 points and from one
    neighbour scan per growth round above: same labels, numbering, carried-over seeds and cluster means either way."""
    from ultranest_amd import layers
    rs = np.random.RandomState(12)
    centres = rs.uniform(size=(5, 3))
    t = centres[rs.randint(5, size=700)] + 0.02 * rs.normal(size=(700, 3))
    prev = rs.randint(1, 7, size=700)
    for ids in (None, prev):
        want = layers.update_clusters(t, t, 0.003, ids)
        with monkeypatch.context() as m:
            m.setattr(layers, "_ADJACENCY_MAX_POINTS", 10)
            got = layers.update_clusters(t, t, 0.003, ids)
        assert got[0] == want[0] > 1
        assert np.array_equal(got[1], want[1])
        assert np.array_equal(got[2], want[2])
