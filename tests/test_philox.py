"""Device-side proposal generation: the Philox restatement against the
Random123 known-answer vectors (CPU), the device stream against the restatement (bit-exact), and
the sampled proposals against the reference's distributions (statistical)."""
import numpy as np
import pytest

from oracle import philox

from golden import inputs

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key, expected)
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,expected", KAT)
def test_restatement_known_answers(ctr, key, expected):
    out = philox.philox4x32_10(np.array([ctr], dtype=np.uint32), key)
    assert tuple(int(x) for x in out[0]) == expected


def test_restatement_uniform_range_and_moments():
    pts, nxt = philox.cube_points(7, 0, 20000, 5)
    assert nxt == 50000
    assert pts.min() > 0 and pts.max() < 1
    assert abs(pts.mean() - 0.5) < 0.005
    assert abs(pts.var() - 1 / 12.) < 0.002
    # odd element count: the last block contributes one double only
    a, na = philox.cube_points(7, 3, 3, 3)
    assert a.shape == (3, 3) and na == 3 + 5


def test_restatement_ball_is_uniform():
    z, nxt = philox.ball_points(3, 11, 20000, 7, 2.5)
    assert nxt == 11 + 20000 * 5
    rad = np.sqrt((z**2).sum(axis=1) / 2.5)
    assert rad.max() <= 1.0
    from scipy import stats
    assert stats.kstest(rad**7, "uniform").pvalue > 1e-3
    assert np.abs(z.mean(axis=0)).max() < 0.02


def _region(kind, n, d, seed):
    import ultranest_amd.mlfriends as m
    from ultranest_amd.regions import DeviceRNG
    rng = np.random.RandomState(seed)
    u = 0.5 + 0.08 * rng.normal(size=(n, d)) * np.linspace(0.5, 1.5, d)
    u = u[np.logical_and(u > 0, u < 1).all(axis=1)]
    layer = m.AffineLayer()
    layer.optimize(u, u)
    region = getattr(m, kind)(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rng)
    region.create_ellipsoid()
    return region, DeviceRNG


@pytest.mark.gpu
def test_device_stream_matches_restatement():
    from ultranest_amd import kernels
    got = kernels.philox_blocks(0, 0, 1)
    assert tuple(int(x) for x in got[0]) == KAT[0][2]
    for seed, stream in [(0, 0), (12345678901234567, 1), (2**64 - 1, 2)]:
        got = kernels.philox_blocks(seed, stream, 5000)
        want = philox.blocks(seed, stream, np.arange(5000, dtype=np.uint64))
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["MLFriends", "RobustEllipsoidRegion"])
@pytest.mark.parametrize("d", [2, 7, 50])
def test_cube_sampling_is_the_host_pipeline_on_the_philox_batch(kind, d):
    """method 0: the accepted rows are exactly inside() applied to the restated Philox batch, in
    draw order; the counter advances by the blocks consumed."""
    region, DeviceRNG = _region(kind, 400, d, d)
    nsamples = 20011
    for seed, offset in [(5, 0), (99, 123456789012)]:
        pts, nxt = philox.cube_points(seed, offset, nsamples, d)
        want = pts[region.inside(pts)]
        region.device_rng = DeviceRNG(seed)
        region.device_rng.offset = offset
        got = region.sample_from_boundingbox(nsamples)
        assert region.device_rng.offset == nxt
        assert got.shape == want.shape
        assert np.array_equal(got, want)
        if d == 2:
            assert len(got) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["MLFriends", "RobustEllipsoidRegion"])
@pytest.mark.parametrize("d", [2, 3, 8, 50, 100])
def test_ellipsoid_sampling_matches_restatement_and_region(kind, d):
    region, DeviceRNG = _region(kind, 400 if d <= 50 else 1200, d, 100 + d)   # (d = 100: 32 outputs per wave, 52 KiB of LDS per workgroup)
    nsamples = 30000
    region.device_rng = DeviceRNG(17)
    got = region.sample_from_wrapping_ellipsoid(nsamples)
    z, nxt = philox.ball_points(17, 0, nsamples, d, region.enlarge)
    assert region.device_rng.offset == nxt
    w = region.ellipsoid_center + np.dot(z, region.ellipsoid_axes_T)
    ok = np.logical_and(w > 0, w < 1).all(axis=1)
    region.device_rng = None
    ok[ok] = region.inside(w[ok])
    want = w[ok]
    # libm vs device transcendental functions: a handful of boundary points may flip
    assert abs(len(got) - len(want)) <= max(3, len(want) // 2000)
    if len(got) == len(want):
        assert np.allclose(got, want, rtol=0, atol=1e-12)
    assert len(got) > 0
    assert region.inside(got).mean() > 0.999
    assert np.logical_and(got > 0, got < 1).all()


@pytest.mark.gpu
def test_ellipsoid_sampling_is_uniform_in_the_ellipsoid():
    region, DeviceRNG = _region("RobustEllipsoidRegion", 400, 6, 3)
    region.device_rng = DeviceRNG(2)
    got = region.sample_from_wrapping_ellipsoid(40000)
    delta = got - region.ellipsoid_center
    q = np.einsum("ij,jk,ik->i", delta, region.ellipsoid_invcov, delta) / region.enlarge
    assert q.max() <= 1 + 1e-9
    if len(got) > 39000:     # ellipsoid (almost) entirely in the cube: radial law is exact
        from scipy import stats
        assert stats.kstest(q**3, "uniform").pvalue > 1e-3


@pytest.mark.gpu
def test_capacity_and_empty():
    from ultranest_amd._lib import HipLibraryError
    region, DeviceRNG = _region("MLFriends", 200, 2, 9)
    handle = region._dev.sync(region, True)
    full, nxt = handle.sample(0, 5000, 1, 0)
    part, nxt2 = handle.sample(0, 5000, 1, 0, capacity=7)
    assert nxt == nxt2 == 5000
    assert len(part) == 7 and np.array_equal(part, full[:7])
    none, nxt3 = handle.sample(0, 0, 1, 0, capacity=4)
    assert len(none) == 0 and nxt3 == 0
    with pytest.raises((ValueError, HipLibraryError)):
        handle.sample(1, 10, 1, 0)       # axes not provided


@pytest.mark.gpu
def test_sampler_with_device_rng_recovers_gaussian_evidence():
    from ultranest_amd.harness import StaticNestedSampler
    from ultranest_amd.regions import DeviceRNG
    d = 4
    sigma = 0.05

    def ll(v):
        return -0.5 * (((v - 0.5) / sigma)**2).sum(axis=1)

    s = StaticNestedSampler(d, ll, num_live_points=400, ndraw=8192, seed=3, device_rng=DeviceRNG(11))
    res = s.run(dlogz=0.1)
    want = d * np.log(sigma * np.sqrt(2 * np.pi))
    assert abs(res["logz"] - want) < 5 * res["logzerr"] + 0.1, (res, want)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [2, 7, 30, 70])
def test_transformed_boundingbox_sampling_matches_restatement(d):
    """method 2: the restated Philox batch in whitened space pushed through the host pipeline of the
    reference (neighbour test, untransform, cube, ellipsoid) gives the device's accepted points."""
    region, DeviceRNG = _region("MLFriends", 400, d, 200 + d)
    nsamples = 40000
    region.device_rng = DeviceRNG(23)
    region.device_rng.offset = 77
    got = region.sample_from_transformed_boundingbox(nsamples)
    pad = region.maxradiussq**0.5
    tpts, nxt = philox.tbox_points(23, 77, nsamples, d, region.bbox_lo, region.bbox_hi, pad)
    assert region.device_rng.offset == nxt
    region.device_rng = None
    near = region._near_live_points(tpts)
    w = region.transformLayer.untransform(tpts[near])
    ok = np.logical_and(w > 0, w < 1).all(axis=1)
    ok[ok] = region.inside_ellipsoid(w[ok])
    want = w[ok]
    assert abs(len(got) - len(want)) <= max(2, len(want) // 2000)      # np.dot vs the device FMA chain at the borders
    if len(got) == len(want):
        assert np.allclose(got, want, rtol=0, atol=1e-12)
    if d <= 7:
        assert len(got) > 50
    assert region.inside(got).mean() > 0.999 if len(got) else True


@pytest.mark.gpu
@pytest.mark.parametrize("d", [2, 6])
def test_sampling_from_points_matches_restatement_and_is_uniform(d):
    """method 3: same draws as the restatement; thinning by multiplicity makes the accepted points
    uniform over the union of balls -- compared with the bounding-box method on the same region."""
    region, DeviceRNG = _region("MLFriends", 300, d, 300 + d)
    nsamples = 60000
    region.device_rng = DeviceRNG(31)
    got = region.sample_from_points(nsamples)
    tpts, thin, which, nxt = philox.around_points(31, 0, nsamples, d, region.unormed, region.maxradiussq)
    assert region.device_rng.offset == nxt
    from ultranest_amd import kernels
    mult = np.empty(nsamples, dtype=np.int64)
    kernels.count_nearby(region.unormed, tpts, region.maxradiussq, mult)
    keep = thin * mult < 1
    region.device_rng = None
    w = region.transformLayer.untransform(tpts[keep])
    ok = np.logical_and(w > 0, w < 1).all(axis=1)
    ok[ok] = region.inside_ellipsoid(w[ok])
    want = w[ok]
    assert abs(len(got) - len(want)) <= max(3, len(want) // 500)       # libm vs device log / cos / pow at the ball borders
    assert len(got) > 500
    assert region.inside(got).mean() > 0.999
    # uniformity: first and second moments agree with the (independent) bounding-box sampler
    region.device_rng = DeviceRNG(32)
    ref = region.sample_from_boundingbox(400000)
    assert len(ref) > 1000
    se = ref.std(axis=0) * np.sqrt(1.0 / len(ref) + 1.0 / len(got))
    assert (np.abs(got.mean(axis=0) - ref.mean(axis=0)) < 5 * se).all()
    assert np.allclose(got.std(axis=0), ref.std(axis=0), rtol=0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("method_name", ["sample_from_boundingbox", "sample_from_wrapping_ellipsoid",
                                         "sample_from_transformed_boundingbox", "sample_from_points"])
def test_device_refill_equals_sample_then_callbacks(method_name):
    """mlf_region_refill = the same draw as region.sample() followed by the prior transform, the
    likelihood and the L > Lmin cut, with only the survivors copied back."""
    from ultranest_amd import likelihoods
    region, DeviceRNG = _region("MLFriends", 400, 4, 77)
    region.current_sampling_method = getattr(region, method_name)
    loglike = likelihoods.GaussLikelihood(0.5, 0.08, 4)
    nsamples = 30000
    region.device_rng = DeviceRNG(41)
    pts = region.sample(nsamples)
    nxt = region.device_rng.offset
    p_host = likelihoods.rosenbrock_transform(pts)
    L_host = likelihoods.rosenbrock_loglike(p_host)
    Lmin = np.median(L_host)
    region.device_rng = DeviceRNG(41)
    region.current_sampling_method = getattr(region, method_name)
    u, p, L, nc = region.refill(nsamples, Lmin, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
    assert region.device_rng.offset == nxt and nc == len(pts)
    keep = L_host > Lmin
    assert abs(len(u) - keep.sum()) <= 2
    if len(u) == keep.sum():
        assert np.array_equal(u, pts[keep]) and np.array_equal(p, p_host[keep])
        assert np.allclose(L, L_host[keep], rtol=1e-12, atol=1e-12)
    assert (L > Lmin).all() and len(u) > 100
    # callbacks without a device form: the caller is told to take the host route
    assert region.refill(100, Lmin, lambda x: x, loglike) is None


@pytest.mark.gpu
def test_fully_resident_nested_sampling_run():
    """Region proposals, prior transform, likelihood and threshold cut all on the device: only live-point
    replacements reach the host.  5-d Gaussian, analytic evidence 0."""
    from ultranest_amd import likelihoods
    from ultranest_amd.harness import StaticNestedSampler
    from ultranest_amd.regions import DeviceRNG
    d, sigma = 5, 0.05
    loglike = likelihoods.GaussLikelihood(0.5, sigma, d)
    s = StaticNestedSampler(d, loglike, transform=likelihoods.identity_transform, num_live_points=400, ndraw=16384, seed=5,
                            device_rng=DeviceRNG(13))
    res = s.run(dlogz=0.1)
    assert abs(res["logz"]) < 4 * res["logzerr"] + 0.15, res
    assert res["ncall"] > 5000


@pytest.mark.gpu
@pytest.mark.parametrize("method", [2, 3])
def test_tspace_sampling_with_a_circular_axis(method):
    """Methods 2 and 3 on a layer with a circular parameter (AffineLayer.unwrap, mlfriends.pyx:538-545, inside untransform :745-752):
    the live points straddle the 0 / 1 border of axis 0; the device's accepted points are the host pipeline's on the restated draws."""
    import ultranest_amd.mlfriends as m
    from ultranest_amd import kernels
    from ultranest_amd.regions import DeviceRNG
    rng = np.random.RandomState(58)
    n, d = 500, 5
    u = 0.5 + 0.08 * rng.normal(size=(n, d))
    u[:, 0] = (0.98 + 0.05 * rng.normal(size=n)) % 1.0
    u = u[np.logical_and(u > 0, u < 1).all(axis=1)]
    layer = m.AffineLayer(wrapped_dims=[0])
    layer.optimize(u, u)
    region = m.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rng)
    region.create_ellipsoid()
    assert (u[:, 0] < 0.2).any() and (u[:, 0] > 0.8).any()
    nsamples = 60000
    region.device_rng = DeviceRNG(9)
    if method == 2:
        got = region.sample_from_transformed_boundingbox(nsamples)
        tpts, nxt = philox.tbox_points(9, 0, nsamples, d, region.bbox_lo, region.bbox_hi, region.maxradiussq**0.5)
        assert region.device_rng.offset == nxt
        region.device_rng = None
        cand = tpts[region._near_live_points(tpts)]
    else:
        got = region.sample_from_points(nsamples)
        tpts, thin, which, nxt = philox.around_points(9, 0, nsamples, d, region.unormed, region.maxradiussq)
        assert region.device_rng.offset == nxt
        region.device_rng = None
        mult = np.empty(nsamples, dtype=np.int64)
        kernels.count_nearby(region.unormed, tpts, region.maxradiussq, mult)
        cand = tpts[thin * mult < 1]
    w = region.transformLayer.untransform(cand)
    ok = np.logical_and(w > 0, w < 1).all(axis=1)
    ok[ok] = region.inside_ellipsoid(w[ok])
    want = w[ok]
    assert len(want) > 200 and (want[:, 0] < 0.2).any() and (want[:, 0] > 0.8).any()       # both sides of the cut are populated
    assert abs(len(got) - len(want)) <= max(3, len(want) // 500)
    if len(got) == len(want):
        assert np.allclose(got, want, rtol=0, atol=1e-12)
    # a proposal whose wrapped coordinate lies outside [0, 1) comes back from unwrap -> wrap on the other side of the cut: the
    # reference's pipeline accepts it here and `inside` places it elsewhere (0.15 % of the points of this region; host and device alike)
    assert region.inside(got).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("d", [4, 50])
def test_ellipsoid_sampling_with_the_cube_test_inside_the_fused_first_launch(d):
    """Large batches of method 1 take the phased routing, whose first launch (k_prep_sweep) honours the cube test itself (the
    pre-gate arrives in the gate array); forced here at a small size.  The live points sit near a corner, so that a good part of
    the ellipsoid lies outside the cube.  Same accepted points as the routing with a separate per-proposal stage."""
    import ultranest_amd.mlfriends as m
    from ultranest_amd import _lib
    from ultranest_amd.regions import DeviceRNG
    rng = np.random.RandomState(400 + d)
    u = 0.93 + 0.03 * rng.normal(size=(600, d))
    u = u[np.logical_and(u > 0, u < 1).all(axis=1)]
    layer = m.AffineLayer()
    layer.optimize(u, u)
    region = m.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rng)
    region.create_ellipsoid()
    nsamples = 50000
    got = {}
    try:
        for phase_min in (32768, 64):
            _lib.set_option("filter_phase_min_queries", phase_min)
            region.device_rng = DeviceRNG(77)
            got[phase_min] = region.sample_from_wrapping_ellipsoid(nsamples)
            if phase_min == 64:
                assert region._dev.handle.debug_stats()["range_cuts"][0] > 0       # the phased routing ran
    finally:
        _lib.set_option("filter_phase_min_queries", 32768)
    z, _ = philox.ball_points(77, 0, nsamples, d, region.enlarge)
    w = region.ellipsoid_center + np.dot(z, region.ellipsoid_axes_T)
    in_cube = np.logical_and(w > 0, w < 1).all(axis=1)
    assert 0.02 < in_cube.mean() < 0.98, in_cube.mean()                            # the cube test decides a good share
    assert len(got[64]) > 20 and np.array_equal(got[64], got[32768])
    assert np.logical_and(got[64] > 0, got[64] < 1).all()
