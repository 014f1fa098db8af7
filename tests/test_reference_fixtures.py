"""The reference's own over-clustering regression fixtures (its tests/overclustered_u_*.txt and
tests/overclustered_*.npz, copied as data into tests/golden/g13_overclustered.npz) through the reference's own
recipes (tests/test_clustering.py:116-225).  The reference only checks ranges there (14 < nclusters < 20, no
singleton cluster); the fixture also records what the reference computed at every step
(tests/golden/make_golden.py g13), so the same recipes are exact known-answer tests here: radii bit for bit
(float32-representable), cluster labels and counts exactly, the enlargement in its tolerance class, and the
np.random stream position at the end.

Runs on the CPU with the kernels replaced by the oracle (host logic + call order) and on the MI355X (`-m gpu`)."""
import numpy as np
import pytest

TXT_CASES = [20, 23, 24, 27, 49]
NPZ_CASES = [20, 23, 24, 27, 42]


def test_overclustering_txt_recipe(backend, golden):
    """reference tests/test_clustering.py:116-149"""
    import ultranest_amd.mlfriends as M
    g = golden("g13_overclustered")
    np.random.seed(1)
    for i in TXT_CASES:
        points = g["txt%d_u" % i]
        radii, counts = [], []
        for k in range(3):
            layer = M.ScalingLayer(wrapped_dims=[])
            layer.optimize(points, points)
            region = M.MLFriends(points, layer)
            maxr = region.compute_maxradiussq(nbootstraps=30)
            region.maxradiussq = maxr
            nclusters, clusteridxs, overlapped = M.update_clusters(points, points, maxr)
            radii.append(maxr)
            counts.append(nclusters)
        for j in range(3):
            nclusters, clusteridxs, overlapped = M.update_clusters(points, points, maxr)
            counts.append(nclusters)
        assert np.array_equal(np.array(radii), g["txt%d_radii" % i]), (i, radii)
        assert counts == list(g["txt%d_nclusters" % i]), (i, counts)
        assert np.array_equal(clusteridxs, g["txt%d_ids" % i]), i
        assert 14 < nclusters < 20                                  # the reference's own pin


def _state(upd):
    return np.array([upd.region.maxradiussq, upd.region.enlarge, upd.transformLayer.nclusters], dtype=float)


def _same_state(got, want, tag):
    assert got[0] == want[0], (tag, "radius", got, want)              # float32-rounded: bit exact
    np.testing.assert_allclose(got[1], want[1], rtol=1e-9, err_msg=str(tag))      # enlargement: tolerance class
    assert got[2] == want[2], (tag, "nclusters", got, want)


def test_overclustering_update_recipe(backend, golden):
    """reference tests/test_clustering.py:152-225: the driver's `_update_region` (here harness.RegionUpdater) on u0,
    create_new on the same and on the new points, radius invalidated, `_update_region` on the new points."""
    import ultranest_amd.mlfriends as M
    from ultranest_amd.harness import RegionUpdater
    g = golden("g13_overclustered")
    np.random.seed(1)
    for i in NPZ_CASES:
        u0, u1 = g["npz%d_u0" % i], g["npz%d_u" % i]
        upd = RegionUpdater(u0.shape[1], region_class=M.MLFriends, transform_layer_class=M.AffineLayer, build_tregion=False)
        upd.update(u0, nbootstraps=30, minvol=0.)
        _same_state(_state(upd), g["npz%d_first" % i], (i, "first"))
        assert np.array_equal(upd.transformLayer.clusterids, g["npz%d_first_ids" % i]), i
        same = upd.transformLayer.create_new(u0, upd.region.maxradiussq)
        assert np.array_equal(same.clusterids, g["npz%d_same_ids" % i]), i
        _, sizes = np.unique(same.clusterids, return_counts=True)
        assert sizes.min() > 1                                       # the reference's own pin
        new = upd.transformLayer.create_new(u1, upd.region.maxradiussq)
        assert np.array_equal(new.clusterids, g["npz%d_new_ids" % i]), i
        upd.region.maxradiussq = None                                # the live points changed: radius invalid
        updated = upd.update(u1, nbootstraps=30, minvol=0.)
        assert bool(updated) == bool(g["npz%d_updated" % i])
        _same_state(_state(upd), g["npz%d_second" % i], (i, "second"))
        assert np.array_equal(upd.transformLayer.clusterids, g["npz%d_second_ids" % i]), i
        _, sizes = np.unique(upd.transformLayer.clusterids, return_counts=True)
        assert 14 < upd.transformLayer.nclusters < 20 and sizes.min() > 1            # the reference's own pins
        assert np.random.uniform() == float(g["npz%d_next_random" % i]), i           # same stream consumption
