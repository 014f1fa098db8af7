"""PopulationSliceSampler against call-by-call traces recorded from the
reference sampler (tests/golden/make_golden.py g9): same np.random seed, same live points and
thresholds -> the same (u, p, L, nc) out of every __next__, the same final chain state and the
same position of the numpy stream.  `backend` = oracle stand-in (CPU, host logic) or the resident
HIP state machine (`-m gpu`)."""
import types

import numpy as np
import pytest

from golden import inputs


def _region(g, real=False):
    u = inputs.live_points(940, 300, 6)
    layer = types.SimpleNamespace(axes=g["region_axes"], T=g["region_T"], ctr=g["region_ctr"])
    return types.SimpleNamespace(u=u, transformLayer=layer, maxradiussq=float(g["region_maxradiussq"]))


TRACES = [("generate_cube_oriented_direction", 13, 7), ("generate_mixture_random_direction", 40, 5),
          ("generate_region_random_direction", 1, 4)]


@pytest.mark.parametrize("name,popsize,nsteps", TRACES)
def test_trace_equals_reference(backend, golden, name, popsize, nsteps):
    import ultranest_amd.popstepsampler as pop
    g = golden("g9_stepfuncs")
    region = _region(g)
    u = region.u
    Ls = inputs.walker_loglike(u)
    order = np.argsort(Ls)
    want = g["trace_%s_rows" % name]
    np.random.seed(950)
    sampler = pop.PopulationSliceSampler(popsize=popsize, nsteps=nsteps, generate_direction=getattr(pop, name),
                                         scale=0.8)
    nfound = 0
    d = u.shape[1]
    for it in range(len(want)):
        Lmin = Ls[order[min(nfound // 3, len(order) - 50)]]
        unew, pnew, Lnew, nc = sampler.__next__(region, Lmin, u, Ls, inputs.walker_transform, inputs.walker_loglike)
        row = want[it]
        assert nc == int(row[1]), (it, nc, row[1])
        if row[0] == 0:
            assert unew is None and pnew is None and Lnew is None, it
        else:
            nfound += 1
            assert Lnew == row[2], it
            assert np.array_equal(unew, row[3:3 + d]) and np.array_equal(pnew, row[3 + d:3 + 2 * d]), it
    assert np.random.uniform() == float(g["trace_%s_next_random" % name])
    assert sampler.scale == float(g["trace_%s_scale" % name])
    state = sampler.state()
    assert np.array_equal(state["generation"], g["trace_%s_generation" % name])
    assert np.array_equal(state["allL"], g["trace_%s_allL" % name], equal_nan=True)
    assert len(sampler.logstat) > 0 and np.isfinite(sampler.far_enough_fraction)
    assert isinstance(sampler.status, str)


# ---- GPU-only: resident likelihood and the Philox stream -----------------------------------------

def _ball_problem(d, nlive, seed):
    """Gaussian shell threshold: {L > Lmin} is a ball of radius R around 0.5; live points uniform in it."""
    rs = np.random.RandomState(seed)
    R = 0.3
    z = rs.normal(size=(nlive, d))
    z *= (R * rs.uniform(size=(nlive, 1))**(1. / d)) / np.linalg.norm(z, axis=1).reshape((-1, 1))
    u = 0.5 + z
    sigma = 0.1
    Lmin = -0.5 * (R / sigma)**2
    return u, sigma, Lmin, R


def _gpu_region(u):
    import ultranest_amd.mlfriends as m
    layer = m.AffineLayer()
    layer.optimize(u, u)
    region = m.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
    region.create_ellipsoid()
    return region


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["numpy", "philox"])
@pytest.mark.parametrize("direction", ["generate_random_direction", "generate_mixture_random_direction",
                                       "generate_region_random_direction", "generate_cube_oriented_direction"])
def test_slice_sampling_is_uniform_under_the_threshold(mode, direction):
    """Slice sampling inside a hard likelihood contour leaves the uniform distribution invariant:
    the points returned are uniform in the ball, for both random sources and a resident
    (device-evaluated) Gaussian likelihood."""
    from scipy import stats
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    from ultranest_amd.regions import DeviceRNG
    d = 3
    u, sigma, Lmin, R = _ball_problem(d, 400, 11)
    region = _gpu_region(u)
    loglike = likelihoods.GaussLikelihood(0.5, sigma, d)
    norm = -0.5 * np.log(2 * np.pi * sigma**2) * d
    Ls = loglike(u)
    np.random.seed(21)
    sampler = pop.PopulationSliceSampler(popsize=256, nsteps=12, generate_direction=getattr(pop, direction), scale=0.3,
                                         device_rng=DeviceRNG(5) if mode == "philox" else None)
    pts, Lout, ncall = [], [], 0
    for _ in range(6000):
        unew, pnew, Lnew, nc = sampler.__next__(region, Lmin + norm, u, Ls, likelihoods.identity_transform, loglike)
        ncall += nc
        if unew is not None:
            pts.append(unew)
            Lout.append(Lnew)
            assert np.array_equal(unew, pnew)
        if len(pts) >= 1500:
            break
    pts, Lout = np.array(pts), np.array(Lout)
    assert len(pts) >= 1500, len(pts)
    assert (Lout > Lmin + norm).all()
    assert np.allclose(Lout, loglike(pts), rtol=1e-12, atol=1e-12)
    r = np.linalg.norm(pts - 0.5, axis=1) / R
    assert r.max() < 1
    # thin to reduce the correlation between walkers that shared a start point
    assert stats.kstest(r[::3]**d, "uniform").pvalue > 1e-3
    assert np.abs((pts - 0.5).mean(axis=0)).max() < 0.03
    assert sampler.far_enough_fraction > 0.1 and np.isfinite(sampler.mean_jump_distance)


@pytest.mark.gpu
def test_resident_likelihood_equals_host_callbacks_on_rosenbrock():
    """Same numpy stream, once with numpy callbacks and once with the device-evaluated Rosenbrock
    likelihood + affine transform: transformed points are bit-identical, likelihoods agree to
    1e-12, so the trajectories coincide."""
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    d = 6
    rs = np.random.RandomState(3)
    u = 0.5 + 0.04 * rs.normal(size=(300, d))
    region = _gpu_region(u)

    def host_transform(x):
        return x * 20 - 10

    def host_loglike(theta):
        a, b = theta[:, :-1], theta[:, 1:]
        return -2 * (100 * (b - a**2)**2 + (1 - a)**2).sum(axis=1)

    Ls = host_loglike(host_transform(u))
    Lmin = np.sort(Ls)[5]
    runs = []
    for transform, loglike in [(host_transform, host_loglike),
                               (likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)]:
        np.random.seed(8)
        sampler = pop.PopulationSliceSampler(popsize=32, nsteps=6, generate_direction=pop.generate_mixture_random_direction,
                                             scale=0.5)
        out = [sampler.__next__(region, Lmin, u, Ls, transform, loglike) for _ in range(300)]
        runs.append(out)
    nfound = 0
    for (ua, pa, La, nca), (ub, pb, Lb, ncb) in zip(*runs):
        assert nca == ncb
        assert (ua is None) == (ub is None)
        if ua is not None:
            nfound += 1
            assert np.array_equal(ua, ub) and np.array_equal(pa, pb)
            assert abs(La - Lb) <= 1e-12 * abs(La)
    assert nfound > 20


@pytest.mark.gpu
def test_nested_sampling_with_the_population_slice_sampler():
    """End to end: static nested sampling of a 6-d Gaussian with step-sampled replacements
    (Philox stream, resident likelihood) recovers the analytic evidence."""
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    from ultranest_amd.harness import StaticNestedSampler
    from ultranest_amd.regions import DeviceRNG
    d, sigma = 6, 0.05
    loglike = likelihoods.GaussLikelihood(0.5, sigma, d)
    step = pop.PopulationSliceSampler(popsize=128, nsteps=2 * d, generate_direction=pop.generate_mixture_random_direction,
                                      scale=0.5, device_rng=DeviceRNG(4))
    s = StaticNestedSampler(d, loglike, transform=likelihoods.identity_transform, num_live_points=200, seed=2,
                            stepsampler=step)
    res = s.run(dlogz=0.2)
    assert abs(res["logz"] - 0.0) < 4 * res["logzerr"] + 0.2, res
    assert 0 < step.far_enough_fraction <= 1 and step.mean_jump_distance > 0


@pytest.mark.gpu
@pytest.mark.parametrize("direction", ["generate_mixture_random_direction", "generate_cube_oriented_direction"])
def test_graph_replay_equals_kernel_by_kernel_launches(direction):
    """The whole-step path as one hipGraph launch (per-call scalars through a pinned parameter block) gives
    the same points, likelihoods and counts as launching the kernels one by one."""
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    from ultranest_amd.regions import DeviceRNG
    d = 5
    u, sigma, Lmin, R = _ball_problem(d, 300, 21)
    region = _gpu_region(u)
    loglike = likelihoods.GaussLikelihood(0.5, sigma, d)
    Ls = loglike(u)
    thresholds = np.sort(Ls)
    runs = []
    for graph in (True, False):
        sampler = pop.PopulationSliceSampler(popsize=96, nsteps=8, generate_direction=getattr(pop, direction), scale=0.3,
                                             device_rng=DeviceRNG(77))
        sampler.use_graph = graph
        sampler.max_rounds = 1                                  # one round per call: the path the graph captures
        out, nfound = [], 0
        for it in range(400):
            Lcut = thresholds[min(nfound // 4, 200)]          # rising threshold: step_back and restarts happen
            res = sampler.__next__(region, Lcut, u, Ls, likelihoods.identity_transform, loglike)
            nfound += res[0] is not None
            out.append(res)
        runs.append((out, sampler.scale, sampler.ringindex))
    (a, scale_a, ring_a), (b, scale_b, ring_b) = runs
    assert scale_a == scale_b and ring_a == ring_b
    nfound = 0
    for (ua, pa, La, nca), (ub, pb, Lb, ncb) in zip(a, b):
        assert nca == ncb and (ua is None) == (ub is None)
        if ua is not None:
            nfound += 1
            assert np.array_equal(ua, ub) and np.array_equal(pa, pb) and La == Lb
    assert nfound > 30


@pytest.mark.gpu
@pytest.mark.parametrize("problem,d,direction,max_rounds,popsize", [
    ("gauss", 5, "generate_mixture_random_direction", 64, 64),      # odd d: the general form (likelihood row on one lane)
    ("gauss", 8, "generate_cube_oriented_direction", 3, 64),        # the cap is hit: __next__ returns None and is called again
    ("eggbox", 10, "generate_mixture_random_direction", 256, 64),   # even d <= 64: rounds after the first with the state in registers
    ("eggbox", 10, "generate_mixture_random_direction", -256, 64),  # ... and the same rounds forced through the general form
    ("rosenbrock", 6, "generate_region_random_direction", 17, 64),
    ("rosenbrock", 50, "generate_differential_direction", 40, 64),
    ("gauss", 64, "generate_region_oriented_direction", 9, 64),
    ("gauss", 66, "generate_random_direction", 9, 64),              # above 64: general form
    ("eggbox", 2, "generate_random_direction", 2, 64),
    ("eggbox", 4, "generate_mixture_random_direction", 64, 1500),   # two chunks of 1024 walkers: statistics per chunk, then added up
    ("gauss", 6, "generate_region_random_direction", 32, 4500),     # above 4096 walkers: the ring walker's launch, then everybody else's
    ("gauss", 5, "generate_random_direction", 16, 4500)])           # ... in the general form
def test_rounds_equal_single_steps(problem, d, direction, max_rounds, popsize):
    """mlf_walkers_rounds_dev (every walker's rounds back to back inside its wave, the ring walker deciding how many) against
    one mlf_walkers_step_dev call per round, driven as the reference's driver drives __next__ (integrator.py:1839-1950: call
    until a point comes back, then raise the threshold): same points, likelihoods, evaluation counts, scale, ring index,
    Philox offset, per-round statistics rows and resident state, bit for bit."""
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    from ultranest_amd.regions import DeviceRNG
    rs = np.random.RandomState(100 + d)
    if problem == "gauss":
        u = 0.5 + 0.08 * rs.normal(size=(300, d))
        loglike, transform = likelihoods.GaussLikelihood(0.5, 0.1, d), likelihoods.identity_transform
    elif problem == "eggbox":
        u = rs.uniform(size=(400, d))
        loglike, transform = likelihoods.eggbox_loglike, likelihoods.eggbox_transform
    else:
        u = 0.5 + 0.04 * rs.normal(size=(300, d))
        loglike, transform = likelihoods.rosenbrock_loglike, likelihoods.rosenbrock_transform
    u = np.clip(u, 1e-3, 1 - 1e-3)
    region = _gpu_region(u)
    Ls = loglike(transform(u))
    thresholds = np.sort(Ls)
    runs = []
    for mr in (1, max_rounds):
        sampler = pop.PopulationSliceSampler(popsize=popsize, nsteps=7, generate_direction=getattr(pop, direction), scale=0.4,
                                             device_rng=DeviceRNG(31))
        sampler.max_rounds = mr
        out = []
        for it in range(120 if popsize <= 64 else 40):
            Lcut = thresholds[min(it // 3, len(Ls) // 2)]          # rising threshold: step_back and restarts happen
            nc_total, calls = 0, 0
            while True:
                unew, pnew, Lnew, nc = sampler.__next__(region, Lcut, u, Ls, transform, loglike)
                nc_total += nc
                calls += 1
                assert calls < 5000
                if unew is not None:
                    break
            out.append((unew, pnew, Lnew, nc_total))
        runs.append((out, sampler.scale, sampler.ringindex, sampler.device_rng.offset, np.array(sampler.logstat), sampler.state()))
    (a, scale_a, ring_a, off_a, log_a, st_a), (b, scale_b, ring_b, off_b, log_b, st_b) = runs
    for (ua, pa, La, nca), (ub, pb, Lb, ncb) in zip(a, b):
        assert np.array_equal(ua, ub) and np.array_equal(pa, pb) and La == Lb and nca == ncb
    assert scale_a == scale_b and ring_a == ring_b and off_a == off_b
    assert log_a.shape == log_b.shape and np.array_equal(log_a, log_b)
    for key in st_a:
        assert np.array_equal(st_a[key], st_b[key], equal_nan=True), key


@pytest.mark.gpu
def test_live_point_copy_follows_row_replacements():
    """mlf_walkers_update_live (the rows the driver replaced since the last call) leaves the device copy of the live points
    exactly where a full mlf_walkers_set_live of the new arrays leaves it: two populations, same Philox stream, restarts and
    differential directions drawn from the copy -- same resident state afterwards; and through the sampler object: live points
    replaced between calls (as integrator.py:2753-2754 does) against a sampler that re-uploads everything."""
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    from ultranest_amd.regions import DeviceRNG
    d, nlive = 6, 200
    rs = np.random.RandomState(9)
    u = np.clip(0.5 + 0.1 * rs.normal(size=(nlive, d)), 1e-3, 1 - 1e-3)
    loglike = likelihoods.GaussLikelihood(0.5, 0.1, d)
    region = _gpu_region(u)
    runs = []
    for incremental in (True, False):
        us, Ls = u.copy(), loglike(u)
        rs2 = np.random.RandomState(10)
        sampler = pop.PopulationSliceSampler(popsize=48, nsteps=5, generate_direction=pop.generate_differential_direction, scale=0.4,
                                             device_rng=DeviceRNG(13))
        out = []
        for it in range(60):
            if not incremental:
                sampler._seen["live_mirror"] = None          # forces the full upload on every call
            worst = int(np.argmin(Ls))
            Lmin = Ls[worst]
            for calls in range(2000):
                unew, pnew, Lnew, nc = sampler.__next__(region, Lmin, us, Ls, likelihoods.identity_transform, loglike)
                if unew is not None:
                    break
            assert unew is not None, (it, calls)
            out.append((unew, Lnew))
            us[worst], Ls[worst] = unew, Lnew                # the driver's replacement of the dead point
            if it % 7 == 3:                                  # now and then several rows at once
                k = rs2.randint(nlive, size=3)
                us[k] = us[k][:, ::-1].copy()
                Ls[k] = loglike(us[k])
        runs.append((out, sampler.state(), sampler.device_rng.offset))
    (a, sa, oa), (b, sb, ob) = runs
    assert oa == ob
    for (ua, La), (ub, Lb) in zip(a, b):
        assert np.array_equal(ua, ub) and La == Lb
    for key in sa:
        assert np.array_equal(sa[key], sb[key], equal_nan=True), key


@pytest.mark.parametrize("case", ["walk_random", "walk_cube", "slice_random", "slice_mixture_scaled", "slice_region_one"])
def test_batched_population_samplers_equal_reference(backend, golden, case):
    """PopulationRandomWalkSampler and PopulationSimpleSliceSampler (reference popstepsampler.py:192-358, 746-1001): the same
    numpy seed gives the reference's points, likelihoods, evaluation counts, scale, log table and numpy stream position
    (golden g16, recorded from the reference's classes); the vectorised moves underneath are this package's step functions
    (HIP kernels on the `hip` backend, the oracle stub here)."""
    import ultranest_amd.mlfriends as m
    import ultranest_amd.popstepsampler as pop
    from golden import inputs as mk
    g = golden("g16_batched_samplers")
    name, cls, direction, kw, popsize, nsteps, d = [c for c in mk.BATCHED_SAMPLER_CASES if c[0] == case][0]
    u, region, loglike, Ls, Lmin = mk.batched_sampler_problem(m, d)
    kw = dict(kw)
    if kw.get("slice_limit") == "scale":
        kw["slice_limit"] = pop.slice_limit_to_scale
    sampler = getattr(pop, cls)(popsize=popsize, nsteps=nsteps, generate_direction=getattr(pop, direction), **kw)
    np.random.seed(11)
    res = [sampler.__next__(region, Lmin, u, Ls, lambda x: x * 1.0, loglike) for _ in range(40)]
    assert np.array_equal(np.array([r[3] for r in res]), g[name + "_nc"])
    tol = dict(rtol=0, atol=0) if backend == "oracle-stub" else dict(rtol=1e-12, atol=1e-14)   # device whitening of the diagnostics only
    np.testing.assert_allclose(np.array([r[0] for r in res]), g[name + "_u"], rtol=0, atol=0)
    np.testing.assert_allclose(np.array([r[1] for r in res]), g[name + "_p"], rtol=0, atol=0)
    np.testing.assert_allclose(np.array([r[2] for r in res]), g[name + "_L"], rtol=0, atol=0)
    assert sampler.scale == float(g[name + "_scale"])
    np.testing.assert_allclose(np.array(sampler.logstat, dtype=float), g[name + "_logstat"], **(tol if tol["rtol"] else dict(rtol=1e-13, atol=0)))
    assert np.random.uniform() == float(g[name + "_next_random"])
    assert len(sampler.prepared_samples) == (popsize - 40 % popsize) % popsize
    assert str(sampler).startswith(cls)


# ---- the reference's own small tests of this module (tests/test_popstepsampling.py), same recipes -------------
def _make_region(m, ndim, nlive=400):
    us = np.random.uniform(size=(nlive, ndim))
    layer = m.AffineLayer() if ndim > 1 else m.ScalingLayer()
    layer.optimize(us, us)
    region = m.MLFriends(us, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=30)
    region.create_ellipsoid(minvol=1.0)
    return region


# reference tests/test_popstepsampling.py:116-138
def test_direction_proposals_for_every_layer_and_region_class(backend):
    import ultranest_amd.mlfriends as m
    import ultranest_amd.popstepsampler as pop
    proposals = [pop.generate_cube_oriented_direction, pop.generate_random_direction,
                 pop.generate_region_oriented_direction, pop.generate_region_random_direction]
    np.random.seed(2)
    points = np.random.uniform(size=(100, 10))
    for layer_class in m.AffineLayer, m.ScalingLayer:
        layer = layer_class()
        layer.optimize(points, points)
        for region_class in m.MLFriends, m.RobustEllipsoidRegion, m.SimpleRegion:
            region = region_class(points, layer)
            r, f = region.compute_enlargement(minvol=1.0, nbootstraps=30)
            region.maxradiussq, region.enlarge = r, f
            region.create_ellipsoid(minvol=1.0)
            for prop in proposals:
                directions = prop(points, region, scale=1.)
                assert directions.shape == points.shape, (prop, region_class, layer_class)
                assert np.isfinite(directions).all()


# reference tests/test_popstepsampling.py:141-158
def test_slice_limit_functions():
    import ultranest_amd.popstepsampler as pop
    fake_tleft, fake_tright = [-0.5, -0.2, -1.5], [0.2, 2.4, 0.2]
    expected = [(fake_tleft, fake_tright), ([-0.5, -0.2, -1.], [0.2, 1.0, 0.2])]
    for func, (want_left, want_right) in zip((pop.slice_limit_to_unitcube, pop.slice_limit_to_scale), expected):
        tleft, tright = func(fake_tleft, fake_tright)
        assert np.allclose(tleft, want_left) and np.allclose(tright, want_right), func


# reference tests/test_popstepsampling.py:269-292
def test_direction_proposal_values(backend):
    import ultranest_amd.mlfriends as m
    import ultranest_amd.popstepsampler as pop
    np.random.seed(12)
    region = _make_region(m, 10, nlive=400)
    ui = region.u[::2]
    scale = np.random.uniform()
    vcube = pop.generate_cube_oriented_direction(ui, region, scale)
    assert vcube.shape == ui.shape
    assert ((vcube != 0).sum(axis=1) == 1).all(), vcube
    assert np.allclose(np.linalg.norm(vcube, axis=1), scale)
    assert (pop.generate_random_direction(ui, region, scale) != 0).all()
    assert (pop.generate_region_oriented_direction(ui, region, scale) != 0).all()
    assert (pop.generate_region_random_direction(ui, region, scale) != 0).all()
    vcubestd = pop.generate_cube_oriented_direction_scaled(ui, region, scale)
    assert vcubestd.shape == ui.shape
    assert ((vcubestd != 0).sum(axis=1) == 1).all(), vcubestd
