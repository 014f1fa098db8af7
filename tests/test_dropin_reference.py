"""Drop-in check against the REAL reference driver (dev container only: needs /root/reference and
the scratch Cython build made by tests/golden/make_golden.py; skipped elsewhere, never on the GPU
box).  The stock ``ReactiveNestedSampler`` is run UNMODIFIED with this package's region / layer
classes plugged in through its own plug points (``run(region_class=...)``, the
``transform_layer_class`` attribute, and the names integrator.py imports from .mlfriends), with
``vectorized=True``.  Kernels are the oracle stub here (no GPU in this container); because masks
are bit-exact and the np.random call order is preserved, the run must reproduce the stock run's
trajectory EXACTLY (same ncall, niter, logz as fixture g8_c1_run.json)."""
import json
import os
import sys
import tempfile

import numpy as np
import pytest

REF = "/root/reference"
SCRATCH = os.environ.get("ULTRANEST_REFBUILD") or os.path.join(tempfile.gettempdir(), "ultranest_refbuild")
have_ref = os.path.isdir(REF) and os.path.isdir(os.path.join(SCRATCH, "ultranest"))

pytestmark = pytest.mark.skipif(not have_ref, reason="reference tree / scratch build not present (dev container only)")


def test_c1_run_reproduces_stock_trajectory(monkeypatch):
    import oracle_backend
    oracle_backend.install(monkeypatch)
    sys.path.insert(0, SCRATCH)
    try:
        import ultranest
        import ultranest.integrator as integ
    finally:
        sys.path.remove(SCRATCH)
    import ultranest_amd.mlfriends as mine
    for name in ("AffineLayer", "LocalAffineLayer", "MLFriends", "RobustEllipsoidRegion", "ScalingLayer",
                 "WrappingEllipsoid", "find_nearby"):
        assert hasattr(integ, name)
        monkeypatch.setattr(integ, name, getattr(mine, name))

    ndim, sigma = 5, 0.01
    centers = np.ones(ndim) * 0.5
    ncalls = [0]

    def loglike(theta):
        assert theta.ndim == 2 and theta.flags.c_contiguous
        ncalls[0] += len(theta)
        return -0.5 * (((theta - centers) / sigma) ** 2).sum(axis=1) - 0.5 * np.log(2 * np.pi * sigma ** 2) * ndim

    np.random.seed(1)
    sampler = ultranest.ReactiveNestedSampler(["p%d" % i for i in range(ndim)], loglike,
                                              transform=lambda x: x, vectorized=True, log_dir=None)
    assert sampler.transform_layer_class is mine.LocalAffineLayer
    res = sampler.run(min_num_live_points=400, viz_callback=None, show_status=False, region_class=mine.MLFriends)
    assert isinstance(sampler.region, mine.MLFriends)
    assert isinstance(sampler.transformLayer, mine.LocalAffineLayer)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_c1_run.json")))
    assert int(res["ncall"]) == gold["ncall"]
    assert ncalls[0] == gold["ncall"] + 2          # + the 2 test points of _check_likelihood_function (integrator.py:1270)
    assert int(res["niter"]) == gold["niter"]
    assert res["logz"] == gold["logz"] and res["logzerr"] == gold["logzerr"]
    assert abs(res["logz"]) < 3 * res["logzerr"] + 0.5          # analytic logZ = 0


def test_population_slice_sampler_under_the_stock_driver(monkeypatch):
    """The stock driver with ``sampler.stepsampler = <this package's PopulationSliceSampler>``
    (the reference's own plug point, cf. its tests/test_popstepsampling.py:41-52) against the same
    run with the reference's sampler: identical trajectories (ncall, niter, logz).  The bimodal
    likelihood is the one of the reference's test."""
    import oracle_backend
    oracle_backend.install(monkeypatch)
    sys.path.insert(0, SCRATCH)
    try:
        import ultranest
        import ultranest.popstepsampler as refpop
    finally:
        sys.path.remove(SCRATCH)
    import ultranest_amd.popstepsampler as mypop

    def loglike(z):
        a = -0.5 * (((z - 0.7 + np.arange(z.shape[1]) * 0.001) / 0.1)**2).sum(axis=1)
        b = -0.5 * (((z - 0.3 - np.arange(z.shape[1]) * 0.001) / 0.1)**2).sum(axis=1)
        return np.logaddexp(a, b)

    results = []
    for module in (refpop, mypop):
        np.random.seed(3)
        nsteps = np.random.randint(10, 50)
        popsize = np.random.randint(1, 20)
        sampler = ultranest.ReactiveNestedSampler(["a", "b", "c"], loglike, transform=lambda x: x, vectorized=True,
                                                  log_dir=None)
        sampler.stepsampler = module.PopulationSliceSampler(
            popsize=popsize, nsteps=nsteps, generate_direction=module.generate_cube_oriented_direction)
        res = sampler.run(viz_callback=None, show_status=False, max_iters=1500, max_num_improvement_loops=0)
        results.append((int(res["ncall"]), int(res["niter"]), res["logz"], res["logzerr"], sampler.stepsampler.scale))
        assert np.isfinite(sampler.stepsampler.far_enough_fraction)
    assert results[0] == results[1], results


def test_compiled_counter_under_the_stock_driver(monkeypatch):
    """integrator.py's names ``MultiCounter`` / ``BreadthFirstIterator`` (imported from .netiter,
    integrator.py:31) replaced by this package's: the C1 run must keep its trajectory (same ncall and
    niter as fixture g8) with logz equal to rounding (libm vs numpy exp/log)."""
    sys.path.insert(0, SCRATCH)
    try:
        import ultranest
        import ultranest.integrator as integ
    finally:
        sys.path.remove(SCRATCH)
    import ultranest_amd.netiter as mine
    monkeypatch.setattr(integ, "MultiCounter", mine.MultiCounter)
    monkeypatch.setattr(integ, "BreadthFirstIterator", mine.BreadthFirstIterator)
    ndim, sigma = 5, 0.01
    centers = np.ones(ndim) * 0.5

    def loglike(theta):
        return -0.5 * (((theta - centers) / sigma) ** 2).sum(axis=1) - 0.5 * np.log(2 * np.pi * sigma ** 2) * ndim

    np.random.seed(1)
    sampler = ultranest.ReactiveNestedSampler(["p%d" % i for i in range(ndim)], loglike,
                                              transform=lambda x: x, vectorized=True, log_dir=None)
    res = sampler.run(min_num_live_points=400, viz_callback=None, show_status=False)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_c1_run.json")))
    assert int(res["ncall"]) == gold["ncall"] and int(res["niter"]) == gold["niter"]
    assert abs(res["logz"] - gold["logz"]) < 1e-9 and abs(res["logzerr"] - gold["logzerr"]) < 1e-9


def test_run_summaries_and_stores_under_the_stock_driver(monkeypatch, tmp_path):
    """Every name integrator.py takes from .netiter / .utils / .store that this package provides (SURVEY.md 8f
    rows f3, f4) replaced at once, with `log_dir` set and the text point store: the C1 run keeps its trajectory
    and its summary and writes the reference's result files and point log."""
    sys.path.insert(0, SCRATCH)
    try:
        import ultranest
        import ultranest.integrator as integ
    finally:
        sys.path.remove(SCRATCH)
    import ultranest_amd.netiter as mynet
    import ultranest_amd.store as mystore
    import ultranest_amd.utils as myutils
    for name in ("BreadthFirstIterator", "MultiCounter", "PointPile", "TreeNode", "combine_results",
                 "count_tree_between", "find_nodes_before", "logz_sequence"):
        assert hasattr(integ, name)
        monkeypatch.setattr(integ, name, getattr(mynet, name))
    for name in ("make_run_dir", "resample_equal"):
        assert hasattr(integ, name)
        monkeypatch.setattr(integ, name, getattr(myutils, name))
    for name in ("NullPointStore", "TextPointStore"):
        monkeypatch.setattr(integ, name, getattr(mystore, name))
    ndim, sigma = 5, 0.01
    centers = np.ones(ndim) * 0.5

    def loglike(theta):
        return -0.5 * (((theta - centers) / sigma) ** 2).sum(axis=1) - 0.5 * np.log(2 * np.pi * sigma ** 2) * ndim

    names = ["p%d" % i for i in range(ndim)]
    np.random.seed(1)
    sampler = ultranest.ReactiveNestedSampler(names, loglike, transform=lambda x: x, vectorized=True,
                                              log_dir=str(tmp_path), resume="overwrite", storage_backend="tsv")
    assert isinstance(sampler.pointstore, mystore.TextPointStore)
    res = sampler.run(min_num_live_points=400, viz_callback=None, show_status=False)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_c1_run.json")))
    assert int(res["ncall"]) == gold["ncall"] and int(res["niter"]) == gold["niter"]
    assert abs(res["logz"] - gold["logz"]) < 1e-9 and abs(res["logzerr"] - gold["logzerr"]) < 1e-9
    for sub, fn in (("chains", "equal_weighted_post.txt"), ("chains", "weighted_post.txt"), ("chains", "run.txt"),
                    ("info", "results.json"), ("info", "post_summary.csv"), ("results", "points.tsv")):
        assert os.path.getsize(os.path.join(str(tmp_path), sub, fn)) > 0, fn
    stored = json.load(open(os.path.join(str(tmp_path), "info", "results.json")))
    assert stored["niter"] == int(res["niter"]) and stored["paramnames"] == names
    sampler.pointstore.close()
    # the driver switches the text store to one value per line (integrator.py:1191): records of 3 + 2 ndim values
    with open(os.path.join(str(tmp_path), "results", "points.tsv")) as f:
        nvalues = sum(1 for _ in f)
    assert nvalues % (3 + 2 * ndim) == 0 and nvalues // (3 + 2 * ndim) == sampler.pointstore.nrows > int(res["niter"])
