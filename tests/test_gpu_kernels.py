"""GPU parity tests: the HIP kernels, called through the C ABI (ctypes), against
  (a) the golden vectors produced by the real reference (tests/golden/*.npz), and
  (b) the CPU oracle on seeded inputs at sizes it finishes in seconds.
Bit-exact for indices, counts, masks, radii; stated tolerances elsewhere.  Run with `-m gpu`."""
import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from ultranest_amd import _lib, kernels
    assert _lib.device_count() >= 1, "no MI355X visible"
    print("device:", _lib.device_name())
    return kernels


def _find(K, apts, bpts, r2):
    out = np.empty(len(bpts), dtype=np.int64)
    K.find_nearby(apts, bpts, r2, out)
    return out


# ------------------------------------------------------------------ K1 / K2 ------------------
@pytest.mark.parametrize("case", range(len(inputs.FIND_NEARBY_CASES)))
def test_find_nearby_golden(case, golden, K):
    g = golden("g1_find_nearby")
    name, n, d, p = inputs.FIND_NEARBY_CASES[case]
    apts, bpts = inputs.find_nearby_inputs(100 + case, n, d, p)
    for tag in ("mid", "none", "all", "big"):
        r2 = float(g["%s_%s_r2" % (name, tag)])
        assert np.array_equal(_find(K, apts, bpts, r2), g["%s_%s_idx" % (name, tag)]), (name, tag)


@pytest.mark.parametrize("n,d,p", [(1, 1, 1), (63, 2, 65), (64, 7, 64), (65, 9, 1), (1000, 13, 777),
                                   (130, 64, 50), (200, 70, 33), (90, 128, 40)])
def test_find_and_count_vs_oracle(n, d, p, K, oracle):
    rs = np.random.RandomState(n * 1000 + d)
    apts = rs.normal(size=(n, d))
    bpts = rs.normal(size=(p, d)) * 0.7
    bpts[0] = apts[n - 1]
    r2 = float(np.median(((bpts[:, None, :] - apts[None, :50, :]) ** 2).sum(axis=2).min(axis=1)))
    for rr in (r2, 0.0, 1e300):
        assert np.array_equal(_find(K, apts, bpts, rr), oracle.find_nearby(apts, bpts, rr)), rr
        cnt = np.empty(p, dtype=np.int64)
        K.count_nearby(apts, bpts, rr, cnt)
        assert np.array_equal(cnt, oracle.count_nearby(apts, bpts, rr)), rr


def test_find_nearby_edge_cases(K):
    out = np.empty(0, dtype=np.int64)
    K.find_nearby(np.zeros((5, 3)), np.zeros((0, 3)), 1.0, out)          # no queries
    out = np.empty(4, dtype=np.int64)
    K.find_nearby(np.zeros((0, 3)), np.zeros((4, 3)), 1.0, out)          # no live points
    assert (out == -1).all()
    with pytest.raises(ValueError):
        K.find_nearby(np.zeros((5, 1025)), np.zeros((4, 1025)), 1.0, out)  # d > MLF_MAX_DIM (1024), loud
    # non-contiguous inputs are accepted like the reference's strided buffers
    rs = np.random.RandomState(5)
    a = rs.normal(size=(40, 6))[:, ::2]
    b = rs.normal(size=(10, 6))[:, ::2]
    K.find_nearby(a, b, 2.0, out := np.empty(10, dtype=np.int64))
    ref = np.array([next((i for i in range(40) if ((a[i] - b[j]) ** 2).sum() <= 2.0), -1) for j in range(10)])
    assert np.array_equal(out, ref)


# ------------------------------------------------------------------ K3 / K5 / H1 --------------
@pytest.mark.parametrize("tag,n,d", [("a", 300, 4), ("b", 1000, 7)])
def test_subtract_nearby_and_pairdist_golden(tag, n, d, golden, K):
    g = golden("g2_clusters")
    u = inputs.two_blobs(200, n, d)
    assert np.array_equal(K.subtract_nearby(u, float(g[tag + "_r2"])), g[tag + "_subtract"])
    assert K.compute_mean_pair_distance(u, g[tag + "_ids"]) == float(g[tag + "_mpd"])
    ids_z = g[tag + "_ids"].copy()
    ids_z[::5] = 0
    assert K.compute_mean_pair_distance(u, ids_z) == float(g[tag + "_mpd_z"])


def test_subtract_nearby_vs_oracle_dense(K, oracle):
    rs = np.random.RandomState(3)
    u = rs.uniform(size=(333, 11))
    for r2 in (0.0, 0.9, 1e300):     # self only / some / everybody is a neighbour
        assert np.array_equal(K.subtract_nearby(u, r2), oracle.subtract_nearby(u, r2)), r2


@pytest.mark.parametrize("n,d", [(333, 11), (130, 64), (257, 65), (500, 100), (64, 128), (1, 3), (17, 1)])
def test_subtract_nearby_dimensions_and_ragged_tiles(n, d, K, oracle):
    """Both column-half variants of the accumulate kernel (d <= 64, d <= 128), tile counts that are not a multiple
    of the points per workgroup, and mixed neighbourhoods (some points see a whole 64-row tile, others do not)."""
    rs = np.random.RandomState(100 + n + d)
    u = rs.uniform(size=(n, d))
    u[: n // 3] = 0.5 + 1e-3 * rs.normal(size=(n // 3, d))      # a tight clump: dense neighbourhoods
    typical = d / 6.0                                            # mean squared distance of uniform points
    for r2 in (0.0, 0.25 * typical, typical, 1e300):
        assert np.array_equal(K.subtract_nearby(u, r2), oracle.subtract_nearby(u, r2)), (n, d, r2)


@pytest.mark.parametrize("n,d,B", [(300, 70, 5), (1000, 128, 3), (100, 64, 33), (65, 1, 2), (2000, 50, 30)])
def test_bootstrap_moments_dimensions(n, d, B, K, oracle):
    """Index-list moment kernels in both column-half variants against numpy (tolerance class)."""
    rs = np.random.RandomState(7 + n + d)
    u = rs.uniform(size=(n, d))
    masks = oracle.draw_bootstrap_masks(rs, n, B)
    masks[0, :] = False
    masks[0, : d + 5] = True                                     # a short list (fewer rows than one batch)
    mean, cov = K.bootstrap_moments(u, masks)
    for b in range(B):
        sel = u[masks[b]]
        np.testing.assert_allclose(mean[b], sel.mean(axis=0), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(cov[b], np.atleast_2d(np.cov(sel, rowvar=0)), rtol=1e-9, atol=1e-15)


# ------------------------------------------------------------------ K4 ------------------------
@pytest.mark.parametrize("case", range(len(inputs.BOOTSTRAP_CASES)))
def test_bootstrap_radius_golden(case, golden, K, oracle):
    g = golden("g3_bootstrap")
    name, n, d, B = inputs.BOOTSTRAP_CASES[case]
    u = inputs.live_points(300 + case, n, d)
    masks = oracle.draw_bootstrap_masks(np.random.RandomState(900 + case), n, B)
    r, skipped = K.maxradiussq_bootstrap(u, masks)
    assert not skipped.any()
    assert np.array_equal(r, g[name + "_r"])


@pytest.mark.parametrize("n,d", [(150, 1), (9, 3), (130, 15), (200, 16), (131, 17), (257, 31), (190, 33), (333, 47), (129, 63),
                                 (300, 64), (140, 65), (128, 100), (170, 127), (90, 128)])
def test_bootstrap_radius_every_size_class(n, d, K, oracle):
    """k_boot holds a live point 16 coordinates per register pair and hands them round by DPP row broadcast; one
    instance per padded dimension, one wave per SIMD (and scratch) above 64 dimensions: every chunk boundary, live-point
    counts that are no multiple of the wave or of the per-wave share.  Bit-exact against the oracle."""
    rs = np.random.RandomState(1000 * d + n)
    u = rs.uniform(size=(n, d))
    u[n // 2] = u[0]                                              # a duplicate: distance exactly 0
    masks = oracle.draw_bootstrap_masks(rs, n, 7)
    r, skipped = K.maxradiussq_bootstrap(u, masks)
    ro, so = oracle.maxradiussq_bootstrap(u, masks)
    assert np.array_equal(skipped, so)
    assert np.array_equal(r, ro)


@pytest.mark.parametrize("n,d,W", [(300, 4, 2), (1000, 12, 3), (4000, 50, 8), (131, 17, 8), (64, 3, 4), (700, 33, 5)])
def test_bootstrap_radius_row_block_shares(n, d, W, K, oracle):
    """mlf_maxradiussq_bootstrap_rows: rank r of W takes the 64-row blocks shard_bounds(nblocks, r, W) as left-out points
    (all rounds, every live point).  Each share against the oracle on exactly those left-out rows, and the element-wise
    maximum of the shares against the one-rank call -- bit for bit (what the all-reduce MAX of C1 delivers)."""
    from ultranest_amd.distributed import shard_bounds
    import oracle_backend
    rs = np.random.RandomState(31 * n + d)
    u = rs.uniform(size=(n, d))
    masks = oracle.draw_bootstrap_masks(rs, n, 30)
    masks[3] = True                                               # a skipped round rides along
    full, skipped = K.maxradiussq_bootstrap(u, masks)
    nblocks = (n + 63) // 64
    acc = np.zeros(len(masks))
    for r in range(W):
        blo, bhi = shard_bounds(nblocks, r, W)
        rows = (min(blo * 64, n), min(bhi * 64, n))
        share, sk = K.maxradiussq_bootstrap(u, masks, rows=rows)
        want, wsk = oracle_backend.maxradiussq_bootstrap(u, masks, rows=rows)
        assert np.array_equal(sk, skipped) and np.array_equal(wsk, skipped)
        assert np.array_equal(share, want), (r, rows)
        acc = np.maximum(acc, share)
    assert np.array_equal(acc, full)
    # an arbitrary (unaligned) row range and an empty one
    share, _ = K.maxradiussq_bootstrap(u, masks, rows=(5, min(n, 77)))
    want, _ = oracle_backend.maxradiussq_bootstrap(u, masks, rows=(5, min(n, 77)))
    assert np.array_equal(share, want)
    share, _ = K.maxradiussq_bootstrap(u, masks, rows=(n, n))
    assert not share.any()


@pytest.mark.parametrize("n,d,B", [(2880, 3, 30), (3000, 17, 7), (4000, 50, 30), (4033, 64, 9), (5000, 2, 40)])
def test_bootstrap_radius_pairs_once(n, d, B, K, oracle):
    """k_boot_sym (whole-range passes over > 2816 live points: every pair distance once, used for both points of the pair
    through an LDS tile) against k_boot (option boot_symmetric = 0) and against the oracle, bit for bit: tile runs that change
    the row block mid-way, a last block that is not full, more rounds than one pass holds (B = 40), duplicates."""
    from ultranest_amd import _lib
    rs = np.random.RandomState(7 * n + d)
    u = rs.uniform(size=(n, d))
    u[n // 2] = u[0]
    u[n - 1] = u[65]
    masks = oracle.draw_bootstrap_masks(rs, n, B)
    masks[1] = True                                               # a skipped round rides along
    try:
        _lib.set_option("boot_symmetric", 0)
        r0, s0 = K.maxradiussq_bootstrap(u, masks)
        _lib.set_option("boot_symmetric", 1)
        r1, s1 = K.maxradiussq_bootstrap(u, masks)
    finally:
        _lib.set_option("boot_symmetric", 1)
    ro, so = oracle.maxradiussq_bootstrap(u, masks)
    assert np.array_equal(s0, so) and np.array_equal(s1, so)
    assert np.array_equal(r0, ro)
    assert np.array_equal(r1, ro)


@pytest.mark.parametrize("n,d", [(9, 3), (64, 2), (65, 5), (150, 1), (200, 16), (257, 31), (333, 47), (300, 64), (1000, 50)])
def test_bootstrap_radius_pairs_once_small(n, d, K, oracle):
    """The same kernel forced onto small arrays (boot_symmetric = 2): one tile, a lone partial block, a diagonal tile only."""
    from ultranest_amd import _lib
    rs = np.random.RandomState(11 * n + d)
    u = rs.uniform(size=(n, d))
    u[n // 2] = u[0]
    masks = oracle.draw_bootstrap_masks(rs, n, 11)
    try:
        _lib.set_option("boot_symmetric", 2)
        r, skipped = K.maxradiussq_bootstrap(u, masks)
    finally:
        _lib.set_option("boot_symmetric", 1)
    ro, so = oracle.maxradiussq_bootstrap(u, masks)
    assert np.array_equal(skipped, so)
    assert np.array_equal(r, ro)


def test_bootstrap_radius_degenerate_masks(K, oracle):
    u = inputs.live_points(1, 100, 3)
    masks = np.zeros((3, 100), dtype=bool)
    masks[0] = True                  # all selected -> skipped (reference :1048)
    masks[2, ::2] = True             # row 1: none selected -> skipped
    r, skipped = K.maxradiussq_bootstrap(u, masks)
    ro, so = oracle.maxradiussq_bootstrap(u, masks)
    assert list(skipped) == [True, True, False] == list(so)
    assert np.array_equal(r, ro)


def test_bootstrap_eggboxregion_reference_recipe(golden, K, oracle):
    g = golden("g3_bootstrap")
    pts = g["eggbox_pts"]
    unormed = g["eggbox_unormed"]
    for seed in range(10):
        masks = oracle.draw_bootstrap_masks(np.random.RandomState(seed), len(pts), 30)
        r, _ = K.maxradiussq_bootstrap(unormed, masks)
        assert r.max() == g["eggbox_maxr"][seed]
        assert 1e-10 < r.max() < 6e-10      # the reference test's own pin


# ------------------------------------------------------------------ H3 / T1 / R3 --------------
@pytest.mark.parametrize("name,n,d,p,case", [("c1", 400, 5, 3000, 0), ("c2", 2000, 20, 3000, 1), ("c5", 4000, 50, 1536, 2)])
def test_region_inside_golden(name, n, d, p, case, golden, K, oracle):
    g = golden("g456_region")
    u = inputs.live_points(400 + case, n, d)
    pts = inputs.proposal_mix(500 + case, u, p, shell_q=float(g[name + "_enlarge"]))
    ctr, T = g[name + "_layer_ctr"], g[name + "_layer_T"]
    ell_c, ell_a, enlarge = g[name + "_ell_center"], g[name + "_ell_invcov"], float(g[name + "_enlarge"])
    # H3: bit-exact with the oracle (= numpy's einsum order), mask equal to the reference's
    emask, q = K.inside_ellipsoid(pts, ell_c, ell_a, enlarge, return_q=True)
    emask_o, q_o = oracle.inside_ellipsoid(pts, ell_c, ell_a, enlarge, return_q=True)
    assert np.array_equal(q, q_o)
    assert np.array_equal(emask, np.unpackbits(g[name + "_emask"])[:p].astype(bool))
    # T1: FMA chain == oracle restatement bit for bit; within 1e-13 of numpy's BLAS
    t = K.affine_transform(pts, ctr, T)
    assert np.array_equal(t, oracle.affine_transform(pts, ctr, T))
    np.testing.assert_allclose(t, np.dot(pts - ctr, T), rtol=0, atol=1e-12)
    # R3: device-resident region, whole pipeline
    unormed = np.dot(u - ctr, T)
    reg = K.DeviceRegion()
    for r2key, mkey in (("_r2", "_mask"), ("_r2_tight", "_mask_tight")):
        reg.set(unormed, 0, ctr, T, None, ell_c, ell_a, enlarge, float(g[name + r2key]))
        mask = reg.inside(pts)
        assert np.array_equal(mask, np.unpackbits(g[name + mkey])[:p].astype(bool)), (name, mkey)
    # live points are inside their own region (reference integrator.py:2102)
    reg.set(unormed, 0, ctr, T, None, ell_c, ell_a, enlarge, float(g[name + "_r2"]))
    assert reg.inside(u).all() == bool(g[name + "_inside_live"])
    # in-place live point replacement keeps device state in sync (integrator.py:2753)
    t_new = K.affine_transform(pts[:1], ctr, T)[0]
    reg.update_point(3, t_new)
    un2 = unormed.copy()
    un2[3] = t_new
    ref = oracle.region_inside(pts, un2, ctr, T, ell_c, ell_a, enlarge, float(g[name + "_r2_tight"]))
    reg.set_thresholds(enlarge, float(g[name + "_r2_tight"]))
    got = reg.inside(pts)
    assert np.array_equal(got, ref)
    reg.close()


def test_region_ellipsoid_only_and_scaling_layer(K, oracle):
    rs = np.random.RandomState(11)
    n, d, p = 500, 6, 2000
    u = inputs.live_points(77, n, d)
    mean, std = u.mean(axis=0), u.std(axis=0)
    unormed = (u - mean) / std
    ctr, cov = oracle.bounding_ellipsoid(u)
    inv = np.linalg.inv(cov)
    pts = inputs.proposal_mix(78, u, p, shell_q=1.5)
    reg = K.DeviceRegion()
    reg.set(None, 0, None, None, None, ctr, inv, 1.5, 1e300, use_scan=False)   # RobustEllipsoidRegion
    assert np.array_equal(reg.inside(pts), oracle.inside_ellipsoid(pts, ctr, inv, 1.5))
    reg.set(unormed, 1, mean, std, None, ctr, inv, 1.5, 0.8)                    # ScalingLayer
    emask = oracle.inside_ellipsoid(pts, ctr, inv, 1.5)
    idx = oracle.find_nearby(unormed, (pts - mean) / std, 0.8)
    assert np.array_equal(reg.inside(pts), emask & (idx >= 0))
    reg.close()


def test_affine_transform_wrapped_dims(golden, K):
    g = golden("g456_region")
    uw = g["g5_wrap_u"]
    shift = np.array([1 - g["g5_wrap_cuts"][0], np.nan])
    t = K.affine_transform(uw, g["g5_wrap_ctr"], g["g5_wrap_T"], wrap_shift=shift)
    np.testing.assert_allclose(t, g["g5_wrap_t"], rtol=0, atol=1e-12)


def test_bootstrap_enlargement_f(golden, K, oracle):
    g = golden("g3_bootstrap")
    name, n, d, B = inputs.BOOTSTRAP_CASES[1]
    u = inputs.live_points(301, n, d)
    masks = oracle.draw_bootstrap_masks(np.random.RandomState(901), n, B)
    mean, cov = K.bootstrap_moments(u, masks)
    ctrs, invs = [], []
    for b in range(B):
        c_np, cov_np = oracle.bounding_ellipsoid(u[masks[b]])
        np.testing.assert_allclose(mean[b], c_np, rtol=1e-13)
        np.testing.assert_allclose(cov[b] * (d + 2), cov_np, rtol=1e-9, atol=1e-18)
        ctrs.append(c_np)
        invs.append(np.linalg.inv(cov_np))
    f = K.bootstrap_quadform_max(u, masks, np.array(ctrs), np.array(invs))
    np.testing.assert_allclose(f, g[name + "_f"], rtol=1e-10)


@pytest.mark.parametrize("case", range(len(inputs.BOOTSTRAP_CASES)))
def test_bootstrap_factor_on_device(case, golden, K, oracle):
    """Moments + Cholesky + quadratic form in one device call (mlf_bootstrap_factor) against the reference's
    per-round enlargement f (fixture G3; tolerance class: 1e-10 relative), and the failure path."""
    g = golden("g3_bootstrap")
    name, n, d, B = inputs.BOOTSTRAP_CASES[case]
    u = inputs.live_points(300 + case, n, d)
    masks = oracle.draw_bootstrap_masks(np.random.RandomState(900 + case), n, B)
    if d > 64:
        with pytest.raises(ValueError):
            K.bootstrap_factor(u, masks, d + 2)
        return
    f = K.bootstrap_factor(u, masks, d + 2)
    np.testing.assert_allclose(f, g[name + "_f"], rtol=1e-10)
    # a round whose selected rows span no volume: NaN for that round only
    flat = u.copy()
    flat[:, 0] = 0.25
    bad = K.bootstrap_factor(flat, masks[:2], d + 2)
    assert np.isnan(bad).all()


@pytest.mark.parametrize("n,d", [(90, 2), (300, 8), (257, 9), (500, 16), (1000, 23), (255, 24), (700, 33), (513, 40), (400, 41),
                                 (600, 48), (900, 55), (1025, 57), (300, 64)])
def test_bootstrap_factor_every_size_class(n, d, K, oracle):
    """k_boot_solvemax: one lane per row, the factor's column through the DPP operand of v_fmac_f64, one instance per
    multiple of 8 dimensions; row counts on both sides of the 256-row workgroup.  Against numpy's inverse + quadratic form
    (the reference's formulation, mlfriends.pyx:1056-1066) to 1e-9: numpy's explicit inverse of these random covariances
    (condition numbers up to ~1e3 x d) is itself only that accurate; the golden fixture g3 is held to 1e-10."""
    rs = np.random.RandomState(77 * d + n)
    u = rs.uniform(size=(n, d)) * rs.uniform(0.2, 1.0, size=d)
    masks = oracle.draw_bootstrap_masks(rs, n, 5)
    f = K.bootstrap_factor(u, masks, d + 2)
    for b in range(len(masks)):
        sel = u[masks[b]]
        ctr = sel.mean(axis=0)
        cov = np.atleast_2d(np.cov(sel, rowvar=0)) * (d + 2)
        delta = u[~masks[b]] - ctr
        want = np.einsum("ij,jk,ik->i", delta, np.linalg.inv(cov), delta).max()
        np.testing.assert_allclose(f[b], want, rtol=1e-9)


# ------------------------------------------------------------------ likelihoods ---------------
def test_likelihoods_golden(golden):
    from ultranest_amd import likelihoods as L
    g = golden("g7_likelihoods")
    n = 1024
    x = inputs.likelihood_inputs(700, n, 5, 0.45, 0.55)
    np.testing.assert_allclose(L.GaussLikelihood(0.5, 0.01, 5)(x), g["gauss5"], rtol=1e-12)
    x = inputs.likelihood_inputs(701, n, 20, 0, 1)
    np.testing.assert_allclose(L.GaussLikelihood(g["gauss20_centers"], 0.1, 20)(x), g["gauss20"], rtol=1e-12)
    for d in (2, 10):
        z = inputs.likelihood_inputs(702 + d, n, d, 0, 1) * 10 * np.pi
        np.testing.assert_allclose(L.eggbox_loglike(z), g["eggbox%d" % d], rtol=1e-12)
        np.testing.assert_allclose(L.eggbox2_loglike(z), g["eggboxsq%d" % d], rtol=1e-12, atol=1e-300)
    for d in (2, 50):
        theta = inputs.likelihood_inputs(720 + d, n, d, 0, 1) * 20 - 10
        np.testing.assert_allclose(L.rosenbrock_loglike(theta), g["rosenbrock%d" % d], rtol=1e-12)


# ------------------------------------------------------------------ full-size properties ------
def test_full_size_properties_c5(K):
    """BASELINE full size (N=4000, d=50, P=1e5 here x10 in bench): size-independent checks --
    live points find themselves at index i with r2=0, permuting queries permutes results,
    a superset radius never loses hits, count>0 <=> find>=0."""
    rs = np.random.RandomState(2024)
    n, d, p = 4000, 50, 100000
    a = rs.normal(size=(n, d))
    out = _find(K, a, a, 0.0)
    assert np.array_equal(out, np.arange(n))
    b = a[rs.randint(n, size=p)] + 0.55 * rs.normal(size=(p, d))
    r2 = 16.0
    idx = _find(K, a, b, r2)
    frac = (idx >= 0).mean()
    assert 0.02 < frac < 0.98, frac
    perm = rs.permutation(p)
    assert np.array_equal(_find(K, a, b[perm], r2), idx[perm])
    idx_big = _find(K, a, b, r2 * 1.5)
    assert ((idx_big >= 0) >= (idx >= 0)).all() and (idx_big[idx >= 0] <= idx[idx >= 0]).all()
    cnt = np.empty(p, dtype=np.int64)
    K.count_nearby(a, b, r2, cnt)
    assert np.array_equal(cnt > 0, idx >= 0)
    # spot-check 64 hits and 64 misses exactly
    hit = np.flatnonzero(idx >= 0)[:64]
    for j in hit:
        acc = np.zeros(n)
        for k in range(d):
            diff = a[:, k] - b[j, k]
            acc = acc + diff * diff
        assert np.flatnonzero(acc <= r2)[0] == idx[j]


# ------------------------------------------------------------------ INTEGRATION.md stub -------
def test_integration_md_ctypes_stub_runs_verbatim(K):
    """The ctypes stub INTEGRATION.md section B shows a maintainer is executed as printed (only the
    library name is made absolute) and checked against the oracle."""
    import os
    import re
    from ultranest_amd import _lib
    import oracle.oracle as orc
    text = open(os.path.join(os.path.dirname(__file__), "..", "INTEGRATION.md")).read()
    section = text[text.index("## B. ctypes stub"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    assert '"libmlfriends_hip.so"' in code
    ns = {}
    exec(compile(code.replace('"libmlfriends_hip.so"', repr(_lib.LIB_PATH)), "INTEGRATION.md", "exec"), ns)
    rng = np.random.RandomState(5)
    a = rng.uniform(size=(300, 7))
    b = rng.uniform(size=(1000, 7))
    near = np.empty(len(b), dtype=np.int64)
    ns["find_nearby"](a, b, 0.3, near)
    np.testing.assert_array_equal(near, orc.find_nearby(a, b, 0.3))
    sel = rng.randint(0, len(a), size=(6, len(a)))
    masks = np.zeros((6, len(a)), dtype=bool)
    for i in range(6):
        masks[i, sel[i]] = True
    r2, skipped = ns["maxradiussq_rounds"](a, masks)
    o_r2, o_sk = orc.maxradiussq_bootstrap(a, masks)
    np.testing.assert_array_equal(skipped, o_sk)
    np.testing.assert_array_equal(r2, o_r2)
    z = rng.uniform(0, 10 * np.pi, size=(500, 4))
    np.testing.assert_allclose(ns["loglike"](z), orc.loglike_eggbox(z), rtol=1e-12, atol=0)
    wide = rng.uniform(size=(4, 1025))     # above MLF_MAX_DIM: the stub surfaces mlf_last_error()
    with pytest.raises(RuntimeError, match="MLF_MAX_DIM"):
        ns["find_nearby"](wide, wide, 0.3, near[:4].copy())


def test_library_and_torch_share_one_hip_runtime_in_either_import_order():
    """ultranest_amd first, torch second (and the reverse) in fresh processes: both see the device and a
    torch tensor's memory is usable through the device-pointer ABI."""
    import subprocess
    import sys
    import os
    root = os.path.join(os.path.dirname(__file__), "..")
    body = """
import numpy as np
{first}
{second}
from ultranest_amd import _lib, kernels
import torch
assert _lib.device_count() >= 1 and torch.cuda.is_available()
a = np.random.RandomState(1).uniform(size=(200, 5)); b = np.random.RandomState(2).uniform(size=(300, 5))
out = np.empty(300, dtype=np.int64)
kernels.find_nearby(a, b, 0.2, out)
t = torch.arange(6, dtype=torch.float64, device="cuda").reshape(3, 2)
assert float((t * 2).sum().item()) == 30.0
kernels.find_nearby(a, b, 0.2, out)
print("ok", int((out >= 0).sum()))
"""
    outs = []
    for first, second in (("from ultranest_amd import _lib, kernels; _lib.device_name()", "import torch; torch.zeros(1, device='cuda')"),
                          ("import torch; torch.zeros(1, device='cuda')", "from ultranest_amd import _lib, kernels; _lib.device_name()")):
        r = subprocess.run([sys.executable, "-c", body.format(first=first, second=second)], cwd=root, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and outs[0].startswith("ok ")


# ------------------------------------------------------------------ above 128 dimensions (mlf_wide.hip) -------------
# The reference has no dimension limit (every loop runs `for k in range(ndim)`, mlfriends.pyx:63-66, 178-180, 217-219; its own
# benchmark goes to d = 256, tests/benchmark_maxradius.py:36-98).  Round 4 stopped at 128 with a loud error.
WIDE_DIMS = [129, 200, 256, 515]


@pytest.mark.parametrize("d", WIDE_DIMS)
def test_wide_find_count_subtract_vs_oracle(d, K, oracle):
    """K1 / K2 / K3 above 128 dimensions: first-hit indices, counts and neighbour means bit for bit"""
    rs = np.random.RandomState(9000 + d)
    n, p = 333, 150
    apts = rs.normal(size=(n, d))
    bpts = apts[rs.randint(n, size=p)] + rs.normal(size=(p, d)) * rs.uniform(0.2, 1.2, size=(p, 1))
    bpts[0] = apts[n - 1]
    d2 = ((bpts[:, None, :] - apts[None, :60, :]) ** 2).sum(axis=2).min(axis=1)
    r2 = float(np.median(d2))
    for rr in (r2, 0.0, 1e300):
        assert np.array_equal(_find(K, apts, bpts, rr), oracle.find_nearby(apts, bpts, rr)), (d, rr)
        cnt = np.empty(p, dtype=np.int64)
        K.count_nearby(apts, bpts, rr, cnt)
        assert np.array_equal(cnt, oracle.count_nearby(apts, bpts, rr)), (d, rr)
    pd = ((apts[:, None, :] - apts[None, :, :]) ** 2).sum(axis=2)
    rq = float(np.quantile(pd[pd > 0], 0.1))
    assert np.array_equal(K.subtract_nearby(apts, rq), oracle.subtract_nearby(apts, rq)), d


@pytest.mark.parametrize("d", WIDE_DIMS)
def test_wide_bootstrap_radius_and_moments_vs_oracle(d, K, oracle):
    """K4 above 128 dimensions: per-round radii (binary32-narrowed) bit for bit, all 32-round passes incl. a second group;
    row-block shares combine to the full pass; the moments within their tolerance class"""
    rs = np.random.RandomState(9100 + d)
    n, B = 300, 35
    pts = rs.normal(size=(n, d))
    masks = oracle.draw_bootstrap_masks(np.random.RandomState(d), n, B)
    masks[3] = True
    r2, skipped = K.maxradiussq_bootstrap(pts, masks)
    r2_o, skipped_o = oracle.maxradiussq_bootstrap(pts, masks)
    assert np.array_equal(skipped, skipped_o) and np.array_equal(r2, r2_o), d
    parts = [K.maxradiussq_bootstrap(pts, masks, rows=(lo, hi))[0] for lo, hi in ((0, 128), (128, 256), (256, n))]
    assert np.array_equal(np.max(parts, axis=0), r2)
    ctrs, covs = K.bootstrap_moments(pts, masks[:4])
    for b in range(4):
        sel = pts[masks[b]]
        np.testing.assert_allclose(ctrs[b], sel.mean(axis=0), rtol=0, atol=1e-12)
        np.testing.assert_allclose(covs[b], np.cov(sel, rowvar=0), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("d", [65, 72, 80, 100, 127, 128, 129, 200, 256])
def test_wide_region_inside_vs_oracle(d, K, oracle):
    """H3 + T1 + R3 above 64 dimensions (65 ... 128: the FP64 matrix-core stage of mlf_prep64.hip in front of the f16 sweep with
    K up to 144 columns; above 128: mlf_wide.hip): quadratic form in the einsum order and the whitening chain bit for bit, the
    membership mask of a device-resident region (live points whitened on the device) equal to the oracle's, small calls
    (which take the single-launch path below 129 dimensions) included, a live point replaced in place"""
    rs = np.random.RandomState(9200 + d)
    n, p = 400, 700
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    ctr = u.mean(axis=0)
    cov = np.cov(u, rowvar=0) * (d + 2) + 1e-6 * np.eye(d)      # n > d: full rank; the jitter keeps eigh well away from zero
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    pts = inputs.proposal_mix(9300 + d, u, p, shell_q=2.0)
    emask, q = K.inside_ellipsoid(pts, ctr, inv, 2.0 * d, return_q=True)
    emask_o, q_o = oracle.inside_ellipsoid(pts, ctr, inv, 2.0 * d, return_q=True)
    assert np.array_equal(q, q_o) and np.array_equal(emask, emask_o)
    t = K.affine_transform(pts, ctr, T)
    assert np.array_equal(t, oracle.affine_transform(pts, ctr, T))
    tl = K.affine_transform(u, ctr, T)
    dd = ((tl[:150, None, :] - tl[None, :150, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    r2 = float(np.quantile(dd.min(axis=1), 0.8))
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, 2.0 * d, r2, live_space=1)
    want = oracle.region_inside(pts, tl, ctr, T, ctr, inv, 2.0 * d, r2)
    assert np.array_equal(reg.inside(pts), want)
    assert 0 < want.sum() < p
    assert np.array_equal(reg.inside(pts[:10]), want[:10])          # 10 host points: no single-launch path up here
    assert reg.inside(u).all()
    from ultranest_amd import _lib
    for name, opts in (("exact scan only", {"filter": 0}), ("vector per-proposal stage", {"fused_prep": 0})):
        for k_, v_ in opts.items():
            _lib.set_option(k_, v_)
        try:
            assert np.array_equal(reg.inside(pts), want), name
        finally:
            for k_ in opts:
                _lib.set_option(k_, 1)
    un2 = tl.copy()
    un2[7] = t[1]
    reg.update_point(7, pts[1])
    assert np.array_equal(reg.inside(pts), oracle.region_inside(pts, un2, ctr, T, ctr, inv, 2.0 * d, r2))
    reg.close()


@pytest.mark.parametrize("d", [70, 100, 128])
def test_prep64_proposals_on_the_ellipsoid_boundary_and_wrapped_axes(d, K, oracle):
    """the bounded quadratic form of mlf_prep64.hip must hand every proposal inside its band to the einsum-order evaluation:
    proposals placed ON the ellipsoid (q within a few ulp of the enlargement, both sides), and a layer with wrapped axes"""
    rs = np.random.RandomState(9400 + d)
    n, p = 300, 4000
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    ctr = u.mean(axis=0)
    cov = np.cov(u, rowvar=0) * (d + 2) + 1e-6 * np.eye(d)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    enlarge = 1.7
    z = rs.normal(size=(p, d))
    pts = ctr + z * 0.05
    # scale every second proposal onto the boundary: q(x) = enlarge up to rounding
    dl = pts - ctr
    q = np.einsum('ij,jk,ik->i', dl, inv, dl)
    scale = np.sqrt(enlarge / q)
    scale[1::2] *= 1.0 + rs.uniform(-3e-16, 3e-16, size=len(scale[1::2]))
    pts[::1] = ctr + dl * np.where(np.arange(p) % 4 < 2, scale, 1.0)[:, None]
    emask_o = oracle.inside_ellipsoid(pts, ctr, inv, enlarge)
    assert 0.2 < emask_o.mean() < 0.9
    tl = K.affine_transform(u, ctr, T)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, enlarge, 1e300, live_space=1)      # radius huge: the mask IS the ellipsoid test
    assert np.array_equal(reg.inside(pts), emask_o)
    # wrapped axes: the proposals' wrap happens inside the stage
    shift = np.full(d, np.nan)
    shift[[0, 3, d - 1]] = [0.3, 0.9, 0.55]
    def wrapped(x):      # the wrap of circular axes (mlfriends.pyx:529-536), then the plain chain
        w = np.array(x, dtype=float)
        for k_, sh in enumerate(shift):
            if sh == sh:
                w[:, k_] = np.fmod(w[:, k_] + sh, 1.0)
        return w
    tl_w = oracle.affine_transform(wrapped(u), ctr, T)
    dd = ((tl_w[:100, None, :] - tl_w[None, :100, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    r2 = float(np.quantile(dd.min(axis=1), 0.8))
    reg.set(tl_w, 0, ctr, T, shift, ctr, inv, 3.0, r2)
    pts2 = inputs.proposal_mix(9500 + d, u, 3000, shell_q=2.0)
    want = oracle.inside_ellipsoid(pts2, ctr, inv, 3.0) & (oracle.find_nearby(tl_w, oracle.affine_transform(wrapped(pts2), ctr, T), r2) >= 0)
    assert np.array_equal(reg.inside(pts2), want)
    reg.close()


def test_wide_reference_classes_at_d_200(K, oracle):
    """the drop-in classes above 128 dimensions: AffineLayer + MLFriends, bootstrapped radius / enlargement (moments on the
    device, LAPACK inverse on the host, quadratic form on the device), create_ellipsoid, inside"""
    import ultranest_amd.mlfriends as M
    rs = np.random.RandomState(77)
    n, d = 450, 200
    u = 0.5 + 0.04 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=6, rng=np.random.RandomState(5))
    masks = oracle.draw_bootstrap_masks(np.random.RandomState(5), n, 6)
    r_o, f_o = oracle.compute_enlargement(u, region.unormed, masks)
    assert region.maxradiussq == r_o
    assert abs(region.enlarge - f_o) <= 1e-9 * f_o
    region.create_ellipsoid()
    pts = inputs.proposal_mix(78, u, 600, shell_q=region.enlarge)
    want = oracle.region_inside(pts, region.unormed, layer.ctr, layer.T, region.ellipsoid_center, region.ellipsoid_invcov,
                                region.enlarge, region.maxradiussq)
    assert np.array_equal(region.inside(pts), want)
    assert region.inside(u).all()
