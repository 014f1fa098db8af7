"""Single-launch MLFriends.inside for 1 ... 256 proposals (csrc/mlf_small.hip) against the batched pipeline and the
oracle: masks must be identical whatever path a call takes (reference mlfriends.pyx:1186-1211; callers with such
batches: stepsampler.py:296-330, 1060-1071, integrator.py:1854-1855)."""
import os

import numpy as np
import pytest

# soak runs: MLF_FUZZ_OFFSET=k shifts the seeds of the random tests (the default run is offset 0)
FUZZ_OFFSET = int(os.environ.get("MLF_FUZZ_OFFSET", "0"))

import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from ultranest_amd import _lib, kernels
    assert _lib.device_count() >= 1, "no MI355X visible"
    return kernels


def _both_paths(reg, pts):
    from ultranest_amd import _lib
    out = {}
    for small in (1, 0):
        _lib.set_option("small_path", small)
        out[small] = reg.inside(pts)
    _lib.set_option("small_path", 1)
    return out[1], out[0]


def _affine_region(seed, n, d, wrapped=False):
    u = inputs.live_points(seed, n, d)
    shift = None
    w = u
    if wrapped:
        u[:, 0] = np.fmod(u[:, 0] + 0.55, 1.0)
        shift = np.full(d, np.nan)
        shift[0] = 0.5
        w = u.copy()
        w[:, 0] = np.fmod(w[:, 0] + shift[0], 1)
    ctr = w.mean(axis=0)
    cov = np.atleast_2d(np.cov(w, rowvar=0)) * (d + 2) if n > d + 1 else np.eye(d) * 0.1
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    ectr = u.mean(axis=0)
    einv = np.linalg.inv(np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)) if n > d + 1 else np.eye(d) * 10.0
    return u, w, shift, ctr, T, ectr, einv


@pytest.mark.parametrize("n,d", [(1, 1), (5, 2), (400, 5), (257, 7), (2000, 20), (4000, 50), (300, 64), (513, 70), (200, 128)])
def test_small_path_equals_batched_path_and_oracle(n, d, K, oracle):
    wrapped = d in (7, 20)
    u, w, shift, ctr, T, ectr, einv = _affine_region(n + d, n, d, wrapped)
    enlarge, r2 = (60.0, 1.1) if n > d + 1 else (5.0, 0.5)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, shift, ectr, einv, enlarge, r2, live_space=1)
    rs = np.random.RandomState(n)
    tl = oracle.affine_transform(w, ctr, T)
    for p in (1, 2, 10, 128, 256):
        pts = inputs.proposal_mix(11 + p, w, p, shell_q=2.0)
        pts[0] = w[rs.randint(n)]                    # a live point itself: distance exactly 0
        if wrapped:
            pts[:, 0] = np.fmod(pts[:, 0] + 0.5 + 1.0, 1.0)
        small, batched = _both_paths(reg, pts)
        wp = pts.copy()
        if wrapped:
            wp[:, 0] = np.fmod(wp[:, 0] + shift[0], 1)
        want = oracle.inside_ellipsoid(pts, ectr, einv, enlarge) & \
            (oracle.find_nearby(tl, oracle.affine_transform(wp, ctr, T), r2) >= 0)
        assert np.array_equal(small, want), (p, "small path")
        assert np.array_equal(batched, want), (p, "batched path")
    # 257 proposals take the batched path whatever the option says
    pts = inputs.proposal_mix(5, w, 257, shell_q=2.0)
    a, b = _both_paths(reg, pts)
    assert np.array_equal(a, b)
    reg.close()


def test_small_path_on_the_ellipsoid_boundary_and_after_updates(K, oracle):
    """proposals ON the boundary q = enlarge (the exact tier decides) and the incremental region updates of the driver
    (one live point replaced in place, ellipsoid centre moved, thresholds changed: integrator.py:2749-2765)"""
    n, d = 600, 12
    u, w, shift, ctr, T, ectr, einv = _affine_region(3, n, d)
    enlarge, r2 = 9.0, 0.8
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ectr, einv, enlarge, r2, live_space=1)
    rs = np.random.RandomState(5)
    L = np.linalg.cholesky(einv)
    z = rs.normal(size=(200, d))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    pts = ectr + (z * np.sqrt(enlarge) * (1 + 1e-15 * rs.randint(-4, 5, size=(200, 1)))) @ np.linalg.inv(L)
    small, batched = _both_paths(reg, pts)
    want = oracle.region_inside(pts, oracle.affine_transform(u, ctr, T), ctr, T, ectr, einv, enlarge, r2)
    assert np.array_equal(small, want) and np.array_equal(batched, want)
    # driver-style updates
    u2 = u.copy()
    for row in (0, 77, n - 1):
        u2[row] = np.clip(u[row] + 0.01 * rs.normal(size=d), 1e-6, 1 - 1e-6)
        reg.update_point(row, u2[row])
    ectr2 = u2.mean(axis=0)
    reg.set_ellipsoid_center(ectr2)
    reg.set_thresholds(enlarge * 1.05, r2 * 0.9)
    pts = inputs.proposal_mix(8, u2, 100, shell_q=2.0)
    small, batched = _both_paths(reg, pts)
    want = oracle.region_inside(pts, oracle.affine_transform(u2, ctr, T), ctr, T, ectr2, einv, enlarge * 1.05, r2 * 0.9)
    assert np.array_equal(small, want) and np.array_equal(batched, want)
    assert 0.02 < want.mean() < 0.98
    reg.close()


def test_small_path_scaling_layer_asymmetric_ellipsoid_and_no_scan(K, oracle):
    n, d = 333, 6
    u = inputs.live_points(21, n, d)
    mean, std = u.mean(axis=0), u.std(axis=0)
    ectr = mean
    einv = np.linalg.inv(np.cov(u, rowvar=0) * (d + 2))
    einv_asym = einv.copy()
    einv_asym[0, 1] *= 1.0 + 1e-9                       # not symmetric: no Cholesky bound, the exact order decides
    pts = inputs.proposal_mix(22, u, 200, shell_q=2.0)
    tl = (u - mean) / std
    for inv in (einv, einv_asym):
        reg = K.DeviceRegion()
        reg.set(u, 1, mean, std, None, ectr, inv, 20.0, 0.9, live_space=1)
        small, batched = _both_paths(reg, pts)
        want = oracle.inside_ellipsoid(pts, ectr, inv, 20.0) & (oracle.find_nearby(tl, (pts - mean) / std, 0.9) >= 0)
        assert np.array_equal(small, want) and np.array_equal(batched, want)
        reg.close()
    # a region without live points (wrapping ellipsoid only: RobustEllipsoidRegion)
    reg = K.DeviceRegion()
    reg.set(None, 0, None, None, None, ectr, einv, 8.0, 1e300, use_scan=False)
    small, batched = _both_paths(reg, pts)
    want = oracle.inside_ellipsoid(pts, ectr, einv, 8.0)
    assert np.array_equal(small, want) and np.array_equal(batched, want)
    assert 0.02 < want.mean() < 0.98
    reg.close()


def test_small_path_nonfinite_rows(K):
    n, d = 300, 4
    u, w, shift, ctr, T, ectr, einv = _affine_region(9, n, d)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ectr, einv, 30.0, 1.0, live_space=1)
    pts = u[:6].copy()
    pts[1, 2] = np.nan
    pts[3, 0] = np.inf
    pts[4, 3] = -np.inf
    small, batched = _both_paths(reg, pts)
    assert np.array_equal(small, batched)
    assert list(small) == [True, False, True, False, False, True]
    reg.close()


def test_reference_api_follows_every_kind_of_write_to_region_u(K):
    """MLFriends.inside through the reference API: the device copy of the live points follows row assignments
    (re-sent by row), slices, ufunc out=, np.copyto and attribute re-assignment; a fresh region built from the same
    host state must agree after each step (integrator.py:2749-2765 is the row-assignment case)."""
    import ultranest_amd.mlfriends as M
    rs = np.random.RandomState(3)
    n, d = 300, 12
    u = 0.5 + 0.08 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    # thresholds chosen so that many proposals pass the ellipsoid but have no neighbour (the bootstrap would make the
    # balls overlap everywhere at this size)
    t = region.unormed
    dd = ((t[:, None, :] - t[None, :, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    region.maxradiussq = float(np.sort(dd.min(axis=1))[int(0.9 * n)])
    region.enlarge = 2.6
    region.create_ellipsoid()
    pts = np.clip(u[rs.randint(n, size=150)] + 0.08 * rs.normal(size=(150, d)), 1e-6, 1 - 1e-6)

    def fresh():
        other = M.MLFriends(np.array(region.u), region.transformLayer)
        other.maxradiussq, other.enlarge = region.maxradiussq, region.enlarge
        other.ellipsoid_center, other.ellipsoid_invcov = region.ellipsoid_center, region.ellipsoid_invcov
        return other.inside(pts)

    base = region.inside(pts)
    assert np.array_equal(base, fresh()) and 0.05 < base.mean() < 0.95
    # proposals that are outside only for want of a neighbour: writing one of them INTO the live set must turn it on
    lonely = list(np.flatnonzero(~base & region.inside_ellipsoid(pts)))
    assert len(lonely) >= 8

    def write_row(i, q):
        region.u[i] = q                                                        # the driver's write

    def write_rows_array(i, q):
        region.u[np.array([i])] = q

    def write_view(i, q):
        region.u[i][:] = q

    def write_slice(i, q):
        region.u[i:i + 2] = q

    def write_ufunc_out(i, q):
        delta = np.zeros_like(np.asarray(region.u))
        delta[i] = q - np.asarray(region.u)[i]
        np.add(region.u, delta, out=region.u)

    def write_copyto(i, q):
        new = np.array(region.u)
        new[i] = q
        np.copyto(region.u, new)

    def reassign(i, q):
        new = np.array(region.u)
        new[i] = q
        region.u = new

    def inplace_op(i, q):
        u_ = region.u
        delta = np.zeros_like(np.asarray(u_))
        delta[i] = q - np.asarray(u_)[i]
        u_ += delta

    for step, (write, row) in enumerate(zip((write_row, write_rows_array, write_view, write_slice, write_ufunc_out,
                                             write_copyto, reassign, inplace_op), (7, 11, 13, 20, 33, 44, 55, 66))):
        j = lonely[step]
        write(row, pts[j])
        got = region.inside(pts)
        assert got[j], "step %d: the device did not see the write" % step
        assert np.array_equal(got, region.inside(pts))                          # second call: nothing moved
        assert np.array_equal(got, fresh()), "step %d" % step
    region.ellipsoid_center = np.asarray(region.u).mean(axis=0)
    assert np.array_equal(region.inside(pts), fresh())


@pytest.mark.parametrize("nrep", [1, 2, 9, 37, 38, 120])
def test_many_replaced_rows_between_two_calls_go_in_one_update(nrep, K):
    """the driver replaces one live point per iteration and tests membership once per refill (integrator.py:2753,
    1776-1804): all rows written since the last call are re-sent together (mlf_region_update_points), above
    n / 8 rows the whole set; either way the answers are those of a region built from scratch"""
    import ultranest_amd.mlfriends as M
    from ultranest_amd import kernels
    rs = np.random.RandomState(40 + nrep)
    n, d = 300, 7
    u = 0.5 + 0.08 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    t = region.unormed
    dd = ((t[:, None, :] - t[None, :, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    region.maxradiussq = float(np.sort(dd.min(axis=1))[int(0.8 * n)])
    region.enlarge = 2.6
    region.create_ellipsoid()
    calls = {"set": 0, "rows": []}
    real_set, real_rows = kernels.DeviceRegion.set, kernels.DeviceRegion.update_points

    def counted_set(self, *a, **kw):
        calls["set"] += 1
        return real_set(self, *a, **kw)

    def counted_rows(self, rows, live_rows):
        calls["rows"].append(len(rows))
        return real_rows(self, rows, live_rows)
    kernels.DeviceRegion.set, kernels.DeviceRegion.update_points = counted_set, counted_rows
    try:
        for batch in (40, 700):      # single-launch path and the batched path
            pts = np.clip(u[rs.randint(n, size=batch)] + 0.1 * rs.normal(size=(batch, d)), 1e-6, 1 - 1e-6)
            region.inside(pts)
            calls["set"], calls["rows"] = 0, []
            rows = rs.choice(n, size=nrep, replace=False)
            for i in rows:
                region.u[i] = np.clip(region.u[rs.randint(n)] + 0.05 * rs.normal(size=d), 1e-6, 1 - 1e-6)
            region.u[rows[0]] = np.clip(region.u[rows[0]] + 0.01, 1e-6, 1 - 1e-6)     # one row written twice
            got = region.inside(pts)
            if nrep <= max(8, n // 8):
                assert calls["set"] == 0 and calls["rows"] == [nrep]
            else:
                assert calls["set"] == 1 and calls["rows"] == []
            other = M.MLFriends(np.array(region.u), region.transformLayer)
            other.maxradiussq, other.enlarge = region.maxradiussq, region.enlarge
            other.ellipsoid_center, other.ellipsoid_invcov = region.ellipsoid_center, region.ellipsoid_invcov
            kernels.DeviceRegion.set, kernels.DeviceRegion.update_points = real_set, real_rows
            want = other.inside(pts)
            kernels.DeviceRegion.set, kernels.DeviceRegion.update_points = counted_set, counted_rows
            assert np.array_equal(got, want) and 0.02 < want.mean() < 0.98
    finally:
        kernels.DeviceRegion.set, kernels.DeviceRegion.update_points = real_set, real_rows


def test_row_update_larger_than_the_pinned_staging_buffer(K):
    """700 rows x 50 doubles do not fit the 256 KB staging buffer: plain copies on the stream instead; an empty
    update is accepted and changes nothing"""
    rs = np.random.RandomState(77)
    n, d = 6000, 50
    u = inputs.live_points(5, n, d)
    ctr = u.mean(axis=0)
    cov = np.cov(u, rowvar=0) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    tl = (u[:200] - ctr) @ T
    dd = ((tl[:, None, :] - tl[None, :, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    r2 = float(np.median(dd.min(axis=1))) * 0.5
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, 1.5 * d, r2, live_space=1)
    cur = u.copy()
    rows = rs.choice(n, size=700, replace=False)
    cur[rows] = np.clip(cur[rs.randint(n, size=700)] + 0.02 * rs.normal(size=(700, d)), 1e-6, 1 - 1e-6)
    reg.update_points(rows, cur[rows])
    reg.update_points(np.zeros(0, dtype=np.int64), np.zeros((0, d)))
    fresh = K.DeviceRegion()
    fresh.set(cur, 0, ctr, T, None, ctr, inv, 1.5 * d, r2, live_space=1)
    for p in (64, 5000):
        pts = inputs.proposal_mix(3 + p, cur, p, shell_q=2.0)
        got, want = reg.inside(pts), fresh.inside(pts)
        assert np.array_equal(got, want) and want.any() and not want.all()
    # the replaced rows themselves are members (distance exactly 0 to their own new position)
    assert reg.inside(cur[rows[:50]]).all()
    with pytest.raises(Exception):
        reg.update_points(np.array([n]), cur[:1])
    reg.close()
    fresh.close()


@pytest.mark.parametrize("seed", range(10))
def test_interleaved_row_updates_and_calls_of_every_size(seed, K):
    """live points replaced one at a time (mlf_region_update_point: the pre-filter operands are requantised lazily)
    between membership calls of every size class -- single launch, exact scan, pre-filter with tile ranges, phased
    pre-filter -- must always agree with a region set up from scratch on the same live points"""
    seed += FUZZ_OFFSET
    rs = np.random.RandomState(900 + seed)
    d = int(rs.choice([2, 5, 9, 16, 33, 50]))
    n = int(rs.randint(300, 2500))
    u = inputs.live_points(seed + 70, n, d)
    ctr = u.mean(axis=0)
    cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    tl = (u - ctr) @ T
    dd = ((tl[:150, None, :] - tl[None, :150, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    r2 = float(np.sort(dd.min(axis=1))[100])
    enlarge = float(d) * 1.5
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, enlarge, r2, live_space=1)
    cur = u.copy()
    sizes = [1, 7, 200, 256, 257, 1000, 3000, 40000]
    for step in range(8):
        if step % 3 == 2:      # several rows in one call (mlf_region_update_points), sometimes many
            rows = rs.choice(n, size=int(rs.choice([1, 2, 17, min(n, 300)])), replace=False)
            cur[rows] = np.clip(cur[rs.randint(n, size=len(rows))] + 0.02 * rs.normal(size=(len(rows), d)), 1e-6, 1 - 1e-6)
            reg.update_points(rows, cur[rows])
        else:
            for _ in range(int(rs.randint(1, 4))):
                row = int(rs.randint(n))
                cur[row] = np.clip(cur[int(rs.randint(n))] + 0.02 * rs.normal(size=d), 1e-6, 1 - 1e-6)
                reg.update_point(row, cur[row])
        p = sizes[(step + seed) % len(sizes)]
        pts = inputs.proposal_mix(seed * 31 + step, cur, p, shell_q=2.0)
        got = reg.inside(pts)
        fresh = K.DeviceRegion()
        fresh.set(cur, 0, ctr, T, None, ctr, inv, enlarge, r2, live_space=1)
        want = fresh.inside(pts)
        fresh.close()
        assert np.array_equal(got, want), (d, n, p, step, np.flatnonzero(got != want)[:5])
    reg.close()
