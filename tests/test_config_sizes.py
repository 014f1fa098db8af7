"""BASELINE.json configurations at their OWN sizes on the MI355X (`-m gpu`), each checked against the CPU
oracle or against the exact FP64 scan:

  C3  eggbox d = 10 (reference examples/testeggbox.py:9-26), N = 1000 live points, the HIP eggbox kernel as the
      vectorized likelihood, through the nested-sampling harness with a fixed iteration budget: the run on the
      HIP region path and the run on the oracle stand-in (same numpy stream) must be the same run
      (equal-budget comparison, not convergence).
  C5  sampler leg: 50-d Rosenbrock (examples/testrosenbrock.py:10-16) with PopulationSliceSampler wired as
      in examples/test_PopSliceSampler.py:103-123, N = 4000 live points: resident device walkers against the
      oracle's restatement of stepfuncs.pyx, call by call.
  C5  membership leg: full-size (P = 10^6, N = 4000, d = 50) batches with varied scale / offset / cluster
      structure and r^2 at the routing edges of the MFMA pre-filter: filtered masks == exact-scan masks.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c3_eggbox_d10_n1000_equal_budget(monkeypatch):
    from ultranest_amd.harness import StaticNestedSampler
    from ultranest_amd.likelihoods import eggbox_loglike, eggbox_transform
    import oracle_backend

    def run():
        s = StaticNestedSampler(10, eggbox_loglike, transform=eggbox_transform, num_live_points=1000, ndraw=8192, seed=1)
        res = s.run(dlogz=0.5, max_iters=2500)
        return res, s.updater.region

    got, region = run()
    assert got["niter"] == 2500 and got["ncall"] > 3000, got
    assert region.u.shape == (1000, 10)
    oracle_backend.install(monkeypatch)      # region kernels -> CPU oracle; the likelihood stays the HIP kernel
    want, region_o = run()
    assert got["niter"] == want["niter"] and got["ncall"] == want["ncall"], (got, want)
    assert got["ncall_region"] == want["ncall_region"] and got["nclusters"] == want["nclusters"], (got, want)
    assert abs(got["logz"] - want["logz"]) < 1e-9 and abs(got["logzerr"] - want["logzerr"]) < 1e-9, (got, want)
    assert np.array_equal(region.u, region_o.u)
    assert region.maxradiussq == region_o.maxradiussq
    assert abs(region.enlarge - region_o.enlarge) <= 1e-10 * region_o.enlarge


def _c5_region(u, nboot=10, seed=2):
    import ultranest_amd.mlfriends as m
    layer = m.AffineLayer()
    layer.optimize(u, u)
    region = m.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=nboot, rng=np.random.RandomState(seed))
    region.create_ellipsoid()
    return region


def test_c5_rosenbrock_d50_population_slice_sampler_vs_oracle(monkeypatch):
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    import oracle_backend
    d, nlive, ncalls = 50, 4000, 240
    rs = np.random.RandomState(31)
    u = 0.5 + 0.02 * rs.normal(size=(nlive, d))            # around the Rosenbrock ridge after x*20-10: |theta| < ~2
    region = _c5_region(u)
    # the vectorized-likelihood callbacks are the HIP kernels behind plain callables (numpy in / numpy out, the
    # reference's callback surface); without a `device_spec` the sampler fetches the proposals like the reference does
    def transform(x):
        return likelihoods.rosenbrock_transform(x)

    def loglike(theta):
        return likelihoods.rosenbrock_loglike(theta)

    Ls = loglike(transform(u))
    thresholds = np.sort(Ls)

    def run():
        np.random.seed(77)
        sampler = pop.PopulationSliceSampler(popsize=64, nsteps=10, generate_direction=pop.generate_mixture_random_direction,
                                             scale=1.0)
        out, nfound = [], 0
        for _ in range(ncalls):
            Lmin = thresholds[min(nfound // 2, nlive // 2)]
            res = sampler.__next__(region, Lmin, u, Ls, transform, loglike)
            nfound += res[0] is not None
            out.append(res)
        return out, sampler.scale, sampler.state(), np.random.uniform()

    got, scale_g, state_g, next_g = run()
    oracle_backend.install(monkeypatch)
    want, scale_w, state_w, next_w = run()
    nfound = 0
    for it, ((ua, pa, La, nca), (ub, pb, Lb, ncb)) in enumerate(zip(got, want)):
        assert nca == ncb, it
        assert (ua is None) == (ub is None), it
        if ua is not None:
            nfound += 1
            assert np.array_equal(ua, ub) and np.array_equal(pa, pb) and La == Lb, it
    assert nfound >= 20, nfound
    assert scale_g == scale_w and next_g == next_w
    assert np.array_equal(state_g["generation"], state_w["generation"])
    assert np.array_equal(state_g["allL"], state_w["allL"], equal_nan=True)
    assert np.array_equal(state_g["allu"], state_w["allu"], equal_nan=True)


def _inside_resident(region, pts):
    """MLFriends.inside on a batch that is resident in HBM, as bench.py times it (a host batch of this size is sent in chunks of
    131 072 rows, each of which takes the single-sweep routing): mask + the batch counters"""
    import torch
    handle = region._dev.sync(region, True)
    dev = torch.device("cuda", 0)
    d_pts = torch.as_tensor(np.ascontiguousarray(pts), device=dev)
    d_mask = torch.zeros(len(pts), dtype=torch.uint8, device=dev)
    handle.inside_dev(d_pts.data_ptr(), len(pts), d_mask.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_mask.cpu().numpy().astype(bool), handle.debug_stats()


def _ellipsoid_draws(region, z, rad):
    """set E of SURVEY.md 8d from pre-drawn normals `z` and radial uniforms `rad`"""
    zz = z / np.linalg.norm(z, axis=1, keepdims=True)
    zz *= region.enlarge ** 0.5 * rad ** (1.0 / z.shape[1])
    return region.ellipsoid_center + zz @ region.ellipsoid_axes_T


@pytest.fixture(scope="module")
def fuzz_draws():
    rs = np.random.RandomState(2024 + FUZZ_OFFSET)
    P, d = 1000000, 50
    return rs.normal(size=(P, d)), rs.uniform(size=(P, 1))


# soak runs: MLF_FUZZ_OFFSET=k shifts the seeds (the default run is offset 0)
FUZZ_OFFSET = int(os.environ.get("MLF_FUZZ_OFFSET", "0"))
FUZZ_CASES = ["baseline", "tiny-scale", "offset-mixed", "two-clusters", "r2-edge-hi", "r2-edge-lo"]


@pytest.mark.parametrize("case", FUZZ_CASES)
def test_c5_full_size_filter_equals_exact_scan(case, fuzz_draws):
    from ultranest_amd import _lib
    z, rad = fuzz_draws
    P, d = z.shape
    N = 4000
    rs = np.random.RandomState(FUZZ_CASES.index(case) + 5 + FUZZ_OFFSET)
    if case == "tiny-scale":
        u = 0.5 + 1e-4 * rs.normal(size=(N, d)) * np.linspace(0.2, 3.0, d)
    elif case == "offset-mixed":
        u = 0.9 + 0.01 * rs.normal(size=(N, d))
    elif case == "two-clusters":
        u = np.where(rs.uniform(size=(N, 1)) < 0.5, 0.3, 0.7) + 0.02 * rs.normal(size=(N, d))
    else:
        u = 0.5 + 0.05 * rs.normal(size=(N, d))
    assert np.logical_and(u > 0, u < 1).all()
    region = _c5_region(u)
    pts = _ellipsoid_draws(region, z, rad)
    if case == "offset-mixed":            # a third of the batch are uniform-cube draws: gated out by the ellipsoid
        pts[::3] = rs.uniform(size=pts[::3].shape)
    pts[5::1000] = u[rs.randint(N, size=len(pts[5::1000]))]       # exact copies of live points: distance 0
    amax = np.abs(region.unormed - region.unormed.mean(axis=0)).max()
    sigma = 2.0 ** -np.ceil(np.log2(amax))
    r2_list = [region.maxradiussq]
    if case == "r2-edge-hi":              # sigma^2 r^2 around 4096: the host-side switch between filter and exact scan
        r2_list = [3600.0 / sigma**2, 4090.0 / sigma**2, 4100.0 / sigma**2]
    if case == "r2-edge-lo":              # ... and around 1e-30; only the exact copies can hit
        r2_list = [3e-30 / sigma**2, 0.5e-30 / sigma**2, 1e-26 / sigma**2]
    for r2 in r2_list:
        region.maxradiussq = float(r2)
        got = region.inside(pts)
        _lib.set_option("filter", 0)
        try:
            want = region.inside(pts)
        finally:
            _lib.set_option("filter", 1)
        assert np.array_equal(got, want), (case, r2, int((got != want).sum()))
        got_res, stats = _inside_resident(region, pts)          # the same batch resident in HBM: the phased min-only routing
        assert np.array_equal(got_res, want), (case, r2, int((got_res != want).sum()), stats)
        if case == "baseline":
            assert stats["range_cuts"][1] > 0 and stats["same_quadratic_form"] == 1, stats
            # ... and the CPU oracle itself on a 200 000-row slice of the SAME full-size batch (strided, so that the slice
            # crosses every workgroup / compaction set of the batch): the filtered mask of a 10^6 batch is then pinned to
            # the oracle inside pytest, not only to the HIP exact scan (~4 s of one host core)
            from oracle import oracle as orc
            layer = region.transformLayer
            sl = slice(3, None, 5)
            want_o = orc.region_inside(np.ascontiguousarray(pts[sl]), region.unormed, layer.ctr, layer.T, region.ellipsoid_center,
                                       region.ellipsoid_invcov, region.enlarge, region.maxradiussq).astype(bool)
            assert len(want_o) == 200000 and np.array_equal(got[sl], want_o), int((got[sl] != want_o).sum())
        assert got[5::1000].mean() > 0.9, got[5::1000].mean()      # copies of live points: distance exactly 0
        if case == "two-clusters":
            assert 0.0 < got.mean() < 1.0, got.mean()
        if case == "r2-edge-lo":
            assert got.sum() <= len(pts[5::1000]) + 5, got.sum()


@pytest.mark.parametrize("name,opts", [("per-proposal stage in its own launch (k_prep4, then three ranges)", {"fused_first_range": 0}),
                                       ("two ranges", {"filter_second_range_pct": 0}),
                                       ("two ranges, per-proposal stage in its own launch", {"filter_second_range_pct": 0, "fused_first_range": 0}),
                                       ("three ranges cut at 8 % and 40 %", {"filter_first_range_pct": 20, "filter_second_range_pct": 40}),
                                       ("three ranges cut at 45 % and 90 %, per-proposal stage in its own launch",
                                        {"fused_first_range": 0, "filter_first_range_pct": 50, "filter_second_range_pct": 90}),
                                       ("per-tile band test", {"sweep_min": 0}),
                                       ("binary64 per-proposal stage", {"prep_bounded": 0}),
                                       ("single sweep", {"filter_phases": 0}),
                                       ("storage order of the mask-mode operand, two ranges, first range 50 %",
                                        {"filter_order": 0, "filter_first_range_pct": 50, "filter_second_range_pct": 0}),
                                       ("two ranges, first range 15 %", {"filter_first_range_pct": 15, "filter_second_range_pct": 0}),
                                       ("unfused per-proposal stage (k_prep + k_quant_queries)", {"fused_prep": 0}),
                                       ("per-tile band test, wide later range", {"sweep_min": 0, "filter_narrow_tail": 0}),
                                       ("single sweep in 512-wave tile ranges", {"filter_phases": 0, "filter_split_waves": 512}),
                                       ("three ranges, per-tile band test", {"filter_phases": 3, "sweep_min": 0})])
def test_c5_full_size_optional_routings(name, opts, fuzz_draws):
    """The routings that are off by default (k_prep_sweep; k_sweep with its own re-check; k_prep3; one sweep over all tiles; round
    4's order of the live points and share of the first range; a short first range) on a full-size batch, against the default
    routing and the exact scan: results never depend on options."""
    from ultranest_amd import _lib
    z, rad = fuzz_draws
    N, d = 4000, z.shape[1]
    rs = np.random.RandomState(77 + FUZZ_OFFSET)
    u = np.where(rs.uniform(size=(N, 1)) < 0.5, 0.35, 0.65) + 0.03 * rs.normal(size=(N, d))
    region = _c5_region(u)
    pts = _ellipsoid_draws(region, z, rad)
    pts[5::1000] = u[rs.randint(N, size=len(pts[5::1000]))]
    default = region.inside(pts)
    restore = {"fused_first_range": 1, "sweep_min": 1, "prep_bounded": 1, "filter_phases": 1, "filter": 1, "filter_order": 1,
               "filter_first_range_pct": 30, "filter_second_range_pct": 50, "fused_prep": 1, "filter_narrow_tail": 1, "filter_split_waves": 2048}
    try:
        for k, v in opts.items():
            _lib.set_option(k, v)
        got = region.inside(pts)
        _lib.set_option("filter", 0)
        want = region.inside(pts)
    finally:
        for k, v in restore.items():
            _lib.set_option(k, v)
    assert np.array_equal(default, want), (name, int((default != want).sum()))
    assert np.array_equal(got, want), (name, int((got != want).sum()))
    assert 0.0 < got.mean() < 1.0


@pytest.mark.parametrize("case", ["one-cluster", "shifted-centre", "other-matrix"])
def test_c5_full_size_same_quadratic_form(case, fuzz_draws):
    """k_prep_sweep<.., SQ>: with an AffineLayer fitted to ONE cluster the wrapping ellipsoid's matrix is T T^T of the layer up to
    LAPACK rounding (mlfriends.pyx:684-706 against :447-452, :1040-1048) and both share the mean as their centre, so the first
    launch reads the ellipsoid form off the whitening chain ("fused_variant" bit 1).  Proposals hug the ellipsoid's boundary
    (|q / enlarge - 1| up to 4e-4, a few widths of the band the bounded form leaves to the exact test: the only place where that
    decision is delicate) from both sides; the masks of both forms equal
    the exact scan's and the oracle's.  A region whose centres differ by one ulp, or whose ellipsoid is not the layer's, must
    not take that form."""
    from ultranest_amd import _lib
    from oracle import oracle as orc
    z, rad = fuzz_draws
    P, d = z.shape
    N = 4000
    rs = np.random.RandomState(91 + FUZZ_OFFSET)
    u = 0.5 + 0.05 * rs.normal(size=(N, d)) * np.linspace(0.5, 1.5, d)
    region = _c5_region(u)
    if case == "shifted-centre":
        region.ellipsoid_center = region.ellipsoid_center.copy()
        region.ellipsoid_center[7] = np.nextafter(region.ellipsoid_center[7], 1.0)
    if case == "other-matrix":
        region.ellipsoid_invcov = region.ellipsoid_invcov * (1 + 1e-6)
    zz = z / np.linalg.norm(z, axis=1, keepdims=True)
    radius = np.where(np.arange(P)[:, None] % 4 == 0, rad ** (1.0 / d), 1.0 + 4e-4 * (rad - 0.5))
    pts = region.ellipsoid_center + (zz * (region.enlarge ** 0.5 * radius)) @ region.ellipsoid_axes_T
    pts[5::1000] = u[rs.randint(N, size=len(pts[5::1000]))]
    got = {}
    try:
        for variant in (3, 1):
            _lib.set_option("fused_variant", variant)
            got[variant], stats = _inside_resident(region, pts)
            assert stats["range_cuts"][1] > 0, stats               # the headline routing: k_prep_sweep + two more ranges
            assert stats["same_quadratic_form"] == (1 if variant == 3 and case == "one-cluster" else 0), (variant, stats)
            # three quarters of the batch lie within 2e-4 (relative, in the radius) of the boundary: part of them inside the band
            # that goes to the exact test, the rest decided by the bounded form right next to it
            assert 0.02 * P < stats["ellipsoid_band"] < 0.6 * P, stats
        _lib.set_option("filter", 0)
        want = region.inside(pts)
    finally:
        _lib.set_option("filter", 1)
        _lib.set_option("fused_variant", 3)
    assert np.array_equal(got[3], want), int((got[3] != want).sum())
    assert np.array_equal(got[1], want), int((got[1] != want).sum())
    layer = region.transformLayer
    sl = slice(1, None, 10)
    gate_o = orc.inside_ellipsoid(np.ascontiguousarray(pts[sl]), region.ellipsoid_center, region.ellipsoid_invcov, region.enlarge).astype(bool)
    assert 0.3 < gate_o.mean() < 0.9, gate_o.mean()                # both sides of the boundary are populated
    want_o = orc.region_inside(np.ascontiguousarray(pts[sl]), region.unormed, layer.ctr, layer.T, region.ellipsoid_center,
                               region.ellipsoid_invcov, region.enlarge, region.maxradiussq).astype(bool)
    assert np.array_equal(got[3][sl], want_o), int((got[3][sl] != want_o).sum())
