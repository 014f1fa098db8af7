"""Population step-sampler state machine against vectors recorded from the
real reference (tests/golden/make_golden.py, group g9).

``impl`` = the CPU oracle (pins the restatement; runs everywhere) or ``ultranest_amd.stepfuncs``
(the HIP kernels; `-m gpu`).  Both expose the reference's function names and signatures.
Bit-exact: all state arrays, flags, indices.  Tolerance: move distances (BLAS + pairwise sums)."""
import types

import numpy as np
import pytest

from golden import inputs


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def impl(request):
    if request.param == "oracle":
        from oracle import stepfuncs as mod
        return mod
    import ultranest_amd.stepfuncs as mod
    return mod


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("seed,n,d", inputs.STEP_CASES)
def test_evolve_update(impl, golden, seed, n, d):
    g = golden("g9_stepfuncs")
    s, acceptable, Lnew = inputs.update_inputs(seed, n)
    search_right, bisecting = impl.evolve_prepare(s["searching_left"], s["searching_right"])
    k = "upd%d_" % seed
    assert same(search_right, g[k + "search_right"]) and same(bisecting, g[k + "bisecting"])
    success = np.zeros(n, dtype=bool)
    impl.evolve_update(acceptable, Lnew, s["Lmin"], search_right, bisecting, s["currentt"], s["current_left"],
                       s["current_right"], s["searching_left"], s["searching_right"], success)
    for name, arr in [("t", s["currentt"]), ("left", s["current_left"]), ("right", s["current_right"]),
                      ("sl", s["searching_left"]), ("sr", s["searching_right"]), ("success", success)]:
        assert same(arr, g[k + name]), name


@pytest.mark.parametrize("seed,n,d", inputs.STEP_CASES)
def test_evolve(impl, golden, seed, n, d):
    """Full evolve() on the global numpy stream: identical state, identical accepted points and
    identical stream position afterwards."""
    g = golden("g9_stepfuncs")
    s = inputs.walker_state(seed, n, d)
    np.random.seed(seed)
    args = [s[key] for key in ("currentu", "currentL", "currentt", "currentv", "current_left", "current_right",
                               "searching_left", "searching_right")]
    (t, v, lo, hi, sl, sr), (success, unew, pnew, Lnew), nc = impl.evolve(
        inputs.walker_transform, inputs.walker_loglike, s["Lmin"], *args)
    k = "evo%d_" % seed
    assert np.random.uniform() == float(g[k + "next_random"])
    assert nc == int(g[k + "nc"])
    for name, arr in [("t", t), ("left", lo), ("right", hi), ("sl", sl), ("sr", sr), ("success", success),
                      ("unew", unew), ("pnew", pnew), ("Lnew", Lnew), ("currentu_after", args[0])]:
        assert same(arr, g[k + name]), name
    assert same(impl.within_unit_cube(args[0]), g[k + "cube"])
    # the reference hands back the arrays it was given
    assert t is args[2] and lo is args[4] and hi is args[5] and sl is args[6] and sr is args[7]


@pytest.mark.parametrize("seed,n,ngen", [(911, 40, 6), (912, 300, 21), (913, 5, 1)])
def test_step_back(impl, golden, seed, n, ngen):
    g = golden("g9_stepfuncs")
    allL, generation, currentt, Lmin = inputs.step_back_state(seed, n, ngen)
    impl.step_back(Lmin, allL, generation, currentt)
    k = "back%d_" % seed
    assert same(allL, g[k + "allL"]) and same(generation, g[k + "generation"]) and same(currentt, g[k + "t"])


def test_step_back_nothing_to_do(impl):
    allL = np.full((4, 3), np.nan)
    allL[:, 0] = 1.0
    generation = np.zeros(4, dtype=np.int64)
    currentt = np.arange(4.0)
    impl.step_back(0.5, allL, generation, currentt)
    assert (generation == 0).all() and same(currentt, np.arange(4.0))
    generation[:] = -1
    impl.step_back(0.5, allL, generation, currentt)        # max generation -1: empty window
    assert (generation == -1).all()


@pytest.mark.parametrize("seed,n,d", [(921, 200, 3), (922, 50, 50), (923, 7, 1)])
def test_unitcube_line_intersection(impl, golden, seed, n, d):
    g = golden("g9_stepfuncs")
    origin, direction = inputs.line_inputs(seed, n, d)
    lo, hi = impl.unitcube_line_intersection(origin, direction)
    assert same(lo, g["line%d_left" % seed]) and same(hi, g["line%d_right" % seed])


@pytest.mark.parametrize("seed,popsize,d,nparams,busy", [(931, 12, 1, 1, 3), (932, 200, 7, 9, 40), (933, 64, 3, 3, 1)])
@pytest.mark.parametrize("shrink", [1.0, 1.5])
def test_update_vectorised_slice_sampler(impl, golden, seed, popsize, d, nparams, busy, shrink):
    g = golden("g9_stepfuncs")
    a = inputs.slice_update_inputs(seed, popsize, d, nparams, busy)
    res = impl.update_vectorised_slice_sampler(
        a["t"], a["tleft"], a["tright"], a["proposed_L"], a["proposed_u"], a["proposed_p"], a["worker_running"],
        a["status"], a["threshold"], shrink, a["allu"], a["allL"], a["allp"], popsize)
    k = "slice%d_%d_" % (seed, int(shrink * 10))
    for name, val in zip(("tleft", "tright", "worker_running", "status", "allu", "allL", "allp", "discarded"), res):
        assert same(val, g[k + name]), name


def test_update_slice_sampler_reference_case(impl):
    """The worked example of the reference's own test (tests/test_popstepsampling.py:161-213)."""
    worker_running = np.array([0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2], dtype=np.int64)
    status = np.zeros(12, dtype=np.int64)
    status[3:] = 1
    proposed_L = np.array([-12., 0.5, 0.09, -2., 0.4, -5, 2.4, 0.3, -3.4, 1.2, 0.1, 0.5])
    t = np.array([-0.8, -0.2, 0.4, -0.5, -0.3, 0.9, -0.7, 0.2, -0.8, 0.5, -0.4, 0.6])
    pu = np.array([[0., 0., 0., 0., 1., 1., 1., 1., 2., 2.5, 2., 2.]]).T
    res = impl.update_vectorised_slice_sampler(t, -np.ones(12), np.ones(12), proposed_L, pu, pu.copy(), worker_running,
                                               status, 1., 1.0, np.zeros((12, 1)), np.zeros(12), np.zeros((12, 1)), 12)
    tleft, tright, worker_running, status, allu, allL, allp, discarded = res
    assert same(worker_running, [0, 1] * 6)
    assert same(status, [0, 0] + [1] * 10)
    assert same(allL, [0, 0, 1.2] + [0] * 9) and same(allu[:, 0], [0, 0, 2.5] + [0] * 9)
    assert discarded == 1
    assert np.allclose(tleft[:3], [-0.2, -.3, -0.4]) and np.allclose(tright[:3], [0.4, 0.2, 0.5])


def test_row_dist2(impl, golden):
    g = golden("g9_stepfuncs")
    u = inputs.live_points(940, 300, 6)[:50]
    tstart = np.dot(u - g["region_ctr"], g["region_T"])
    tfinal = np.dot(g["diag_ufinal"] - g["region_ctr"], g["region_T"])
    d2 = impl.row_dist2(tstart, tfinal)
    assert np.allclose(d2**0.5, g["diag_dist"], rtol=1e-12, atol=0)
    assert same(d2 > float(g["region_maxradiussq"]), g["diag_far"])


def _region(g):
    u = inputs.live_points(940, 300, 6)
    layer = types.SimpleNamespace(axes=g["region_axes"], T=g["region_T"], ctr=g["region_ctr"])
    return types.SimpleNamespace(u=u, transformLayer=layer, maxradiussq=float(g["region_maxradiussq"]))


DIRECTIONS = ["generate_cube_oriented_direction", "generate_cube_oriented_direction_scaled",
              "generate_random_direction", "generate_region_oriented_direction",
              "generate_region_random_direction", "generate_differential_direction",
              "generate_mixture_random_direction"]


@pytest.mark.parametrize("name", DIRECTIONS)
def test_direction_generators_follow_the_reference_stream(golden, name):
    """Host numpy on both sides: same draws in the same order give the same directions."""
    import ultranest_amd.stepfuncs as sf
    g = golden("g9_stepfuncs")
    region = _region(g)
    np.random.seed(941)
    v = getattr(sf, name)(region.u[:50], region, scale=0.7)
    if name == "generate_region_random_direction":      # einsum summation order: tolerance class
        assert np.allclose(v, g["dir_" + name], rtol=1e-13, atol=1e-15)
    else:
        assert same(v, g["dir_" + name])
