"""Pins the CPU oracle (oracle/) against golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only; bit-exact for integer /
mask / radius outputs, stated tolerances for LAPACK-dependent floats."""
import numpy as np
import pytest

import inputs


@pytest.mark.parametrize("case", range(len(inputs.FIND_NEARBY_CASES)))
def test_g1_find_nearby_bit_exact(case, golden, oracle):
    g = golden("g1_find_nearby")
    name, n, d, p = inputs.FIND_NEARBY_CASES[case]
    apts, bpts = inputs.find_nearby_inputs(100 + case, n, d, p)
    for tag in ("mid", "none", "all", "big"):
        r2 = float(g["%s_%s_r2" % (name, tag)])
        idx = oracle.find_nearby(apts, bpts, r2)
        assert np.array_equal(idx, g["%s_%s_idx" % (name, tag)]), (name, tag)
    # structural pins of the reference semantics
    assert (g[name + "_all_idx"] == 0).all()                  # first index wins
    none = g[name + "_none_idx"]
    assert none[p // 3] == n // 2 and (np.delete(none, p // 3) == -1).all()   # d == 0 <= r2


def test_g1_count_consistent_with_find(golden, oracle):
    # count_nearby is cdef-only in the reference (no direct golden); it shares
    # the distance loop with find_nearby: count > 0 <=> find >= 0, and a brute
    # force numpy count with the same sequential arithmetic must agree.
    g = golden("g1_find_nearby")
    name, n, d, p = inputs.FIND_NEARBY_CASES[0]
    apts, bpts = inputs.find_nearby_inputs(100, n, d, p)
    r2 = float(g[name + "_mid_r2"])
    cnt = oracle.count_nearby(apts, bpts, r2)
    assert np.array_equal(cnt > 0, g[name + "_mid_idx"] >= 0)
    acc = np.zeros((p, n))
    for k in range(d):
        diff = apts[None, :, k] - bpts[:, None, k]
        acc = acc + diff * diff
    assert np.array_equal(cnt, (acc <= r2).sum(axis=1))


@pytest.mark.parametrize("tag,n,d", [("a", 300, 4), ("b", 1000, 7)])
def test_g2_subtract_clusters_pairdist(tag, n, d, golden, oracle):
    g = golden("g2_clusters")
    u = inputs.two_blobs(200, n, d)
    r2 = float(g[tag + "_r2"])
    assert np.array_equal(oracle.subtract_nearby(u, r2), g[tag + "_subtract"])
    nclusters, ids, overlapped = oracle.update_clusters(u, u, r2)
    assert nclusters == int(g[tag + "_nclusters"])
    assert np.array_equal(ids, g[tag + "_ids"])
    assert np.array_equal(overlapped, g[tag + "_overlapped"])
    nclusters2, ids2, _ = oracle.update_clusters(u, u, float(g[tag + "_r2b"]), ids)
    assert nclusters2 == int(g[tag + "_nclusters2"])
    assert np.array_equal(ids2, g[tag + "_ids2"])
    assert oracle.mean_pair_distance(u, ids) == float(g[tag + "_mpd"])
    ids_z = ids.copy()
    ids_z[::5] = 0
    assert oracle.mean_pair_distance(u, ids_z) == float(g[tag + "_mpd_z"])


def test_g2_reference_fixture_clusters2(golden, oracle):
    g = golden("g2_clusters")
    pts = g["ref_clusters2_pts"]
    nclusters, ids, _ = oracle.update_clusters(pts, pts, float(g["ref_clusters2_r2"]))
    assert nclusters == int(g["ref_clusters2_nclusters"])
    assert np.array_equal(ids, g["ref_clusters2_ids"])


@pytest.mark.parametrize("case", range(len(inputs.BOOTSTRAP_CASES)))
def test_g3_bootstrap_radius_bit_exact(case, golden, oracle):
    g = golden("g3_bootstrap")
    name, n, d, B = inputs.BOOTSTRAP_CASES[case]
    u = inputs.live_points(300 + case, n, d)
    masks = oracle.draw_bootstrap_masks(np.random.RandomState(900 + case), n, B)
    r, skipped = oracle.maxradiussq_bootstrap(u, masks)
    assert not skipped.any()
    assert np.array_equal(r, g[name + "_r"])          # float32-rounded, exact
    if n <= 2000:
        f = np.array([oracle.ellipsoid_enlargement(u, m) for m in masks])
        np.testing.assert_allclose(f, g[name + "_f"], rtol=1e-10)
        maxd, maxf = oracle.compute_enlargement(u, u, masks)
        assert maxd == g[name + "_r"].max()
        np.testing.assert_allclose(maxf, g[name + "_f"].max(), rtol=1e-10)


def test_g3_reference_fixture_eggboxregion(golden, oracle):
    # recipe of reference tests/test_clustering.py:81-99, which pins
    # 1e-10 < maxr < 6e-10 and 14 < nclusters < 20 for seeds 0-9
    g = golden("g3_bootstrap")
    pts = g["eggbox_pts"]
    unormed = (pts - pts.mean(axis=0).reshape((1, -1))) / pts.std(axis=0).reshape((1, -1))
    assert np.array_equal(unormed, g["eggbox_unormed"])
    for seed in range(10):
        masks = oracle.draw_bootstrap_masks(np.random.RandomState(seed), len(pts), 30)
        r, _ = oracle.maxradiussq_bootstrap(unormed, masks)
        assert r.max() == g["eggbox_maxr"][seed]
        assert 1e-10 < r.max() < 6e-10
        nclusters, ids, _ = oracle.update_clusters(pts, pts, r.max())
        assert nclusters == g["eggbox_nclusters"][seed] and 14 < nclusters < 20
    assert np.array_equal(ids, g["eggbox_ids_seed9"])


@pytest.mark.parametrize("name,n,d,p,case", [("c1", 400, 5, 3000, 0), ("c2", 2000, 20, 3000, 1), ("c5", 4000, 50, 1536, 2)])
def test_g4_inside_masks(name, n, d, p, case, golden, oracle):
    g = golden("g456_region")
    u = inputs.live_points(400 + case, n, d)
    pts = inputs.proposal_mix(500 + case, u, p, shell_q=float(g[name + "_enlarge"]))
    ctr, T = g[name + "_layer_ctr"], g[name + "_layer_T"]
    unormed = np.dot(u - ctr, T)
    emask, q = oracle.inside_ellipsoid(pts, g[name + "_ell_center"], g[name + "_ell_invcov"],
                                       float(g[name + "_enlarge"]), return_q=True)
    assert np.array_equal(emask, np.unpackbits(g[name + "_emask"])[:p].astype(bool))
    # H3 restatement is bit-identical to numpy's einsum (numpy 2.2.6 probed)
    delta = pts - g[name + "_ell_center"]
    if np.__version__.startswith("2.2"):
        assert np.array_equal(q, np.einsum('ij,jk,ik->i', delta, g[name + "_ell_invcov"], delta))
    for r2key, mkey in (("_r2", "_mask"), ("_r2_tight", "_mask_tight")):
        mask = oracle.region_inside(pts, unormed, ctr, T, g[name + "_ell_center"], g[name + "_ell_invcov"],
                                    float(g[name + "_enlarge"]), float(g[name + r2key]))
        assert np.array_equal(mask, np.unpackbits(g[name + mkey])[:p].astype(bool)), (name, mkey)
    assert float(g[name + "_margin"]) > 1e-9


def test_g5_affine_optimize(golden, oracle):
    g = golden("g456_region")
    u = inputs.live_points(400, 400, 5)
    ctr, T, invT, logvolscale, cov = oracle.affine_optimize(u, u)
    np.testing.assert_allclose(ctr, g["c1_layer_ctr"], rtol=1e-12)
    np.testing.assert_allclose(logvolscale, float(g["c1_layer_logvolscale"]), rtol=1e-10)
    # eigenvector signs are LAPACK-dependent: compare the sign-free product T T^T = inv(cov)
    np.testing.assert_allclose(T @ T.T, g["c1_layer_T"] @ g["c1_layer_T"].T, rtol=1e-8, atol=1e-10)


def test_g7_likelihoods(golden, oracle):
    g = golden("g7_likelihoods")
    n = 1024
    x = inputs.likelihood_inputs(700, n, 5, 0.45, 0.55)
    np.testing.assert_allclose(oracle.loglike_gauss(x, 0.5, 0.01), g["gauss5"], rtol=1e-12)
    x = inputs.likelihood_inputs(701, n, 20, 0, 1)
    np.testing.assert_allclose(oracle.loglike_gauss(x, g["gauss20_centers"], 0.1), g["gauss20"], rtol=1e-12)
    for d in (2, 10):
        z = inputs.likelihood_inputs(702 + d, n, d, 0, 1) * 10 * np.pi
        np.testing.assert_allclose(oracle.loglike_eggbox(z), g["eggbox%d" % d], rtol=1e-12)
        np.testing.assert_allclose(oracle.loglike_eggbox2(z), g["eggboxsq%d" % d], rtol=1e-12, atol=1e-300)
    for d in (2, 50):
        theta = inputs.likelihood_inputs(720 + d, n, d, 0, 1) * 20 - 10
        np.testing.assert_allclose(oracle.loglike_rosenbrock(theta), g["rosenbrock%d" % d], rtol=1e-12)
