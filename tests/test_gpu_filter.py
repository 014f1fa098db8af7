"""MFMA pre-filter (csrc/mlf_filter.hip) vs the exact scan and the CPU oracle: the filter may only
change speed, never a result bit.  Adversarial cases: radii that coincide exactly with a pair
distance (and one ulp below), badly scaled / offset data, queries outside the binary16 range,
uncertain-pair list overflow."""
import os

import numpy as np
import pytest

# soak runs: MLF_FUZZ_OFFSET=k shifts the seeds of the random-shape tests (the default run is offset 0)
FUZZ_OFFSET = int(os.environ.get("MLF_FUZZ_OFFSET", "0"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["phased", "single-sweep", "one-launch"])
def K(request):
    """Every test runs three times: with the phased sweep (undecided queries compacted between live-point ranges; the
    min-only kernels of mlf_sweepmin.hip) forced on for small batches, with the single sweep (k_sweep with its own
    re-check), and with the one-launch path of mlf_mid.hip that batches of this size take by default."""
    from ultranest_amd import _lib, kernels
    assert _lib.device_count() >= 1
    _lib.set_option("filter", 1)
    _lib.set_option("filter_min_queries", 64)      # let small test batches take the filter path
    _lib.set_option("filter_phases", 0 if request.param == "single-sweep" else 1)
    _lib.set_option("filter_phase_min_queries", 64)
    _lib.set_option("mid_max_queries", 131072 if request.param.startswith("one-launch") else 0)
    _MODE["param"] = request.param
    yield kernels
    _lib.set_option("filter_min_queries", 257)
    _lib.set_option("filter_phase_min_queries", 32768)
    _lib.set_option("filter_phases", 1)
    _lib.set_option("mid_max_queries", 2048)
    _lib.set_option("filter", 1)


_MODE = {"param": "phased"}


def request_param_one_launch():
    return _MODE["param"].startswith("one-launch")


def _find(K, apts, bpts, r2, filt):
    from ultranest_amd import _lib
    _lib.set_option("filter", 1 if filt else 0)
    out = np.empty(len(bpts), dtype=np.int64)
    K.find_nearby(apts, bpts, r2, out)
    _lib.set_option("filter", 1)
    return out


def _exact_d2(a, b):
    acc = np.zeros(len(a))
    for k in range(a.shape[1]):
        diff = a[:, k] - b[k]
        acc = acc + diff * diff
    return acc


@pytest.mark.parametrize("n,d,p", [(256, 1, 300), (300, 2, 257), (1000, 10, 2100), (4000, 50, 2500),
                                   (2000, 20, 3000), (777, 58, 400), (640, 59, 333), (500, 90, 200),
                                   (300, 122, 100), (260, 128, 70)])
def test_filter_equals_exact_and_oracle(n, d, p, K, oracle):
    rs = np.random.RandomState(n + d)
    a = rs.normal(size=(n, d))
    b = np.where((np.arange(p) % 2 == 0)[:, None], a[rs.randint(n, size=p)] + 0.4 * rs.normal(size=(p, d)),
                 rs.normal(size=(p, d)))
    nn = np.array([_exact_d2(a, b[j]).min() for j in range(0, p, max(1, p // 64))])
    for r2 in (float(np.median(nn)), float(np.percentile(nn, 10)), float(nn.max() * 1.5)):
        want = oracle.find_nearby(a, b, r2)
        assert np.array_equal(_find(K, a, b, r2, True), want), (n, d, p, r2)
        assert np.array_equal(_find(K, a, b, r2, False), want)


def test_radius_exactly_on_a_pair_distance(K, oracle):
    """r2 == the reference's own value for one pair: that pair must hit (<=); one ulp lower: miss."""
    rs = np.random.RandomState(11)
    n, d, p = 600, 20, 512
    a = rs.normal(size=(n, d))
    b = rs.normal(size=(p, d))
    for j in rs.randint(p, size=12):
        d2 = _exact_d2(a, b[j])
        i = int(np.argsort(d2)[rs.randint(3)])
        for r2 in (d2[i], np.nextafter(d2[i], 0.0), np.nextafter(d2[i], np.inf)):
            want = oracle.find_nearby(a, b, float(r2))
            got = _find(K, a, b, float(r2), True)
            assert np.array_equal(got, want), (j, i, r2)


@pytest.mark.parametrize("scale,offset", [(1e-6, 0.0), (1e6, 0.0), (1.0, 1000.0), (1e-5, 0.5), (3.0, -70000.0)])
def test_scaled_and_offset_data(scale, offset, K, oracle):
    rs = np.random.RandomState(5)
    n, d, p = 900, 7, 1000
    a = offset + scale * rs.normal(size=(n, d))
    b = offset + scale * rs.normal(size=(p, d))
    b[::50] = offset + scale * 1e6 * rs.normal(size=(len(b[::50]), d))     # far outside the binary16 range
    b[7] = np.inf
    b[9, 2] = np.nan
    r2 = float(np.median([_exact_d2(a, b[j]).min() for j in range(1, 200, 3)]))
    want = oracle.find_nearby(a, b, r2)
    assert np.array_equal(_find(K, a, b, r2, True), want)


def test_uncertain_list_overflow_falls_back_exactly(K, oracle):
    """all live points coincide and every query sits at the threshold: every pair is uncertain,
    the list overflows and the exact scan must take over."""
    n, d, p = 512, 4, 4096
    a = np.tile(np.array([[0.25, 0.5, 0.75, 0.125]]), (n, 1))
    rs = np.random.RandomState(1)
    dirs = rs.normal(size=(p, d))
    dirs /= np.sqrt((dirs ** 2).sum(axis=1, keepdims=True))
    b = a[0] + dirs * (1.0 + 1e-7 * rs.normal(size=(p, 1)))
    want = oracle.find_nearby(a, b, 1.0)
    assert 0.2 < (want >= 0).mean() < 0.8
    assert np.array_equal(_find(K, a, b, 1.0, True), want)


def test_region_inside_filter_vs_exact(K, oracle):
    import inputs
    from ultranest_amd import _lib
    n, d, p = 3000, 30, 20000
    u = inputs.live_points(3, n, d)
    ctr = u.mean(axis=0)
    cov = np.cov(u, rowvar=0) * (d + 2)
    w, v = np.linalg.eigh(cov)
    T = v * w ** -0.5
    inv = np.linalg.inv(cov)
    pts = inputs.proposal_mix(4, u, p, shell_q=2.0)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, 2.0, 1.2, live_space=1)
    res = {}
    for filt in (1, 0):
        _lib.set_option("filter", filt)
        res[filt] = reg.inside(pts)
        for row in (0, 17, 2999):                      # in-place replacement requantises the live set
            reg.update_point(row, pts[row])
        res[filt, "upd"] = reg.inside(pts)
        reg.set(u, 0, ctr, T, None, ctr, inv, 2.0, 1.2, live_space=1)
    _lib.set_option("filter", 1)
    assert np.array_equal(res[1], res[0]) and np.array_equal(res[1, "upd"], res[0, "upd"])
    assert 0.05 < res[1].mean() < 0.95
    unormed = oracle.affine_transform(u, ctr, T)
    assert np.array_equal(res[1], oracle.region_inside(pts, unormed, ctr, T, ctr, inv, 2.0, 1.2))
    reg.close()


@pytest.mark.parametrize("p", [12000, 70001])
def test_exact_scan_tail_of_the_unbounded_path_covers_the_whole_batch(p, K, oracle):
    """Round 5 regression: behind the binary64 per-proposal stage (d = 65 ... 128, wrapped axes, prep_bounded = 0) the exact
    scan that takes what the pre-filter cannot (route 2) is a GATED launch of the plain k_scan instance, one query block per
    workgroup -- and its grid was capped at 512 workgroups, so route-2 proposals past the first 8192 (small batches) / 32768
    never got their answer.  Here EVERY proposal is route 2: a live cloud of extent 1e-6 and proposals at distance ~1 (their
    scaled coordinates do not fit binary16), half of them within the radius."""
    from ultranest_amd import _lib
    n, d = 512, 4
    rs = np.random.RandomState(2)
    c = np.array([0.25, 0.5, 0.75, 0.125])
    u = c + 1e-6 * rs.normal(size=(n, d))
    dirs = rs.normal(size=(p, d))
    dirs /= np.sqrt((dirs ** 2).sum(axis=1, keepdims=True))
    pts = c + dirs * (1.0 + 1e-3 * rs.normal(size=(p, 1)))
    ctr, T, inv = np.zeros(d), np.eye(d), np.eye(d)
    want = oracle.region_inside(pts, u, ctr, T, c, inv, 1e6, 1.0)
    assert 0.3 < want.mean() < 0.7
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, c, inv, 1e6, 1.0)
    got = {}
    for name, opts in (("default", {}), ("binary64 stage", {"prep_bounded": 0}), ("exact", {"filter": 0})):
        for k, v in opts.items():
            _lib.set_option(k, v)
        # poison the answer buffer first: a proposal nobody answers must not pass by luck
        reg.inside(np.tile(u[:1], (p, 1)))
        got[name] = reg.inside(pts)
        for k in opts:
            _lib.set_option(k, 1)
    reg.close()
    for name, m in got.items():
        assert np.array_equal(m, want), (name, p, np.flatnonzero(m != want)[:5])


@pytest.mark.parametrize("n,d,p", [(500, 3, 4097), (700, 1, 3000), (900, 17, 2049), (1200, 20, 5000), (4000, 50, 6001),
                                   (300, 64, 2500), (300, 70, 2500)])
def test_fused_prep_matches_unfused(n, d, p, K, oracle):
    """k_prep2 (coalesced staging, coordinate-major whitened output, fused binary16 quantisation)
    vs the unfused kernels, filter on and off, ragged batch sizes, wrapped axis; all four
    combinations must give the oracle's mask."""
    import inputs
    from ultranest_amd import _lib
    u = inputs.live_points(9, n, d)
    u[:, 0] = np.fmod(u[:, 0] + 0.55, 1.0)                   # straddles the border: wrapped axis
    cut = 0.5
    shift = np.full(d, np.nan)
    shift[0] = 1 - cut
    w = u.copy()
    w[:, 0] = np.fmod(w[:, 0] + shift[0], 1)
    ctr = w.mean(axis=0)
    cov = np.atleast_2d(np.cov(w, rowvar=0)) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    ectr = u.mean(axis=0)
    einv = np.linalg.inv(np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2))
    pts = inputs.proposal_mix(10, w, p, shell_q=2.0)
    pts[:, 0] = np.fmod(pts[:, 0] + cut + 1.0, 1.0)          # back to the unwrapped cube
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, shift, ectr, einv, 60.0, 1.1, live_space=1)
    got = {}
    # matrix = k_prep3 (FP64 matrix cores), unfused = k_prep + separate quantisation; set on THIS handle only
    for mode, fused in dict(matrix=1, unfused=0).items():
        for filt in (1, 0):
            reg.set_option("fused_prep", fused)
            reg.set_option("filter", filt)
            got[mode, filt] = reg.inside(pts)
    reg.set_option("fused_prep")
    reg.set_option("filter")
    wp = pts.copy()
    wp[:, 0] = np.fmod(wp[:, 0] + shift[0], 1)
    emask = oracle.inside_ellipsoid(pts, ectr, einv, 60.0)
    tl = oracle.affine_transform(w, ctr, T)
    idx = oracle.find_nearby(tl, oracle.affine_transform(wp, ctr, T), 1.1)
    want = emask & (idx >= 0)
    for key, m in got.items():
        assert np.array_equal(m, want), key
    assert 0.02 < want.mean() < 0.98
    reg.close()


@pytest.mark.parametrize("n,d", [(400, 7), (300, 1), (500, 5), (350, 33), (600, 63)])
def test_bounded_stage_with_odd_batch_lengths(n, d, K, oracle):
    """np * d odd: the proposals end in the middle of one of the 16-byte pieces k_prep4 fetches them in (the last
    coordinate of the last proposal was left out once); every ragged size through the filter path must give the exact
    scan's mask"""
    import inputs
    from ultranest_amd import _lib
    u = inputs.live_points(41 + d, n, d)
    ctr = u.mean(axis=0)
    cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, 40.0, 1.2, live_space=1)
    tl = oracle.affine_transform(u, ctr, T)
    _lib.set_option("small_path", 0)
    try:
        for p in (257, 385, 1001, 2049, 4097, 33333):
            pts = inputs.proposal_mix(50 + p, u, p, shell_q=2.0)
            pts[-1] = u[p % n]                       # the last proposal: a live point itself, inside for certain
            got = {}
            for filt in (1, 0):
                _lib.set_option("filter", filt)
                got[filt] = reg.inside(pts)
            _lib.set_option("filter", 1)
            assert np.array_equal(got[1], got[0]), (p, np.flatnonzero(got[1] != got[0])[:5])
            assert got[1][-1]
            want = oracle.inside_ellipsoid(pts, ctr, inv, 40.0) & (oracle.find_nearby(tl, oracle.affine_transform(pts, ctr, T), 1.2) >= 0)
            assert np.array_equal(got[1], want), p
    finally:
        _lib.set_option("small_path", 1)
        reg.close()


@pytest.mark.parametrize("seed", range(24))
def test_region_inside_random_shapes_filter_equals_exact_scan(seed, K):
    """random (live points, dimension, batch length, geometry): whatever kernels a call is routed through -- bounded or
    binary64 per-proposal stage, pre-filter with one or several tile ranges, phases, exact scan -- the mask is the same"""
    import inputs
    from ultranest_amd import _lib
    seed += FUZZ_OFFSET
    rs = np.random.RandomState(1000 + seed)
    d = int(rs.choice([1, 2, 3, 5, 7, 10, 13, 20, 31, 50, 63, 64]))
    n = int(rs.randint(max(d + 2, 40), 3000))
    p = int(rs.choice([257, 300, 511, 1000, 2047, 2049, 4001, 9999, 40001]))
    u = inputs.live_points(seed, n, d)
    if seed % 3 == 0:                        # strongly anisotropic cloud
        u = 0.5 + (u - 0.5) * np.geomspace(1.0, 0.02, d)
    ctr = u.mean(axis=0)
    cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    tl = (u - ctr) @ T
    dd = ((tl[:200, None, :] - tl[None, :200, :]) ** 2).sum(axis=2)
    np.fill_diagonal(dd, np.inf)
    r2 = float(np.sort(dd.min(axis=1))[int(0.7 * min(n, 200))]) * float(rs.uniform(0.5, 2.0))
    enlarge = float(d) * float(rs.uniform(1.0, 3.0))
    pts = inputs.proposal_mix(seed + 7, u, p, shell_q=2.0)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, enlarge, r2, live_space=1)
    got = {}
    for name, opts in (("default", {}), ("exact", {"filter": 0}), ("binary64 stage", {"prep_bounded": 0}),
                       ("single sweep", {"filter_phases": 0}), ("wide tail", {"filter_narrow_tail": 0}),
                       ("one launch", {"mid_max_queries": 131072}), ("three launches", {"mid_max_queries": 0}),
                       ("per-tile band test", {"mid_max_queries": 0, "sweep_min": 0}),
                       ("storage order", {"filter_order": 0}), ("storage order, three launches", {"filter_order": 0, "mid_max_queries": 0}),
                       ("per-proposal stage in its own launch", {"fused_first_range": 0}),
                       ("three ranges", {"filter_third_range_min_work": 0}),
                       ("three ranges, per-proposal stage in its own launch", {"filter_third_range_min_work": 0, "fused_first_range": 0})):
        mid_before = 131072 if request_param_one_launch() else 0
        for k, v in opts.items():
            if k == "filter_order":      # per handle: the next batch rebuilds (or drops) the centre-first operand
                reg.set_option(k, v)
            else:
                _lib.set_option(k, v)
        got[name] = reg.inside(pts)
        reg.set_option("filter_order")
        for k, v in (("filter", 1), ("prep_bounded", 1), ("filter_phases", 1), ("filter_narrow_tail", 1), ("sweep_min", 1),
                     ("mid_max_queries", mid_before), ("fused_first_range", 1), ("filter_third_range_min_work", 100000000)):
            _lib.set_option(k, v)
    reg.close()
    wrong = {name: (np.flatnonzero(m != got["exact"])[:5].tolist(), m[np.flatnonzero(m != got["exact"])[:5]].tolist())
             for name, m in got.items() if not np.array_equal(m, got["exact"])}
    assert not wrong, (d, n, p, wrong)


@pytest.mark.parametrize("seed", range(12))
def test_find_nearby_random_shapes_vs_oracle(seed, K, oracle):
    """first-hit indices (reference mlfriends.pyx:143-183) for random live sets / batches, pre-filter on and off: the
    lowest index must survive tile ranges swept by different waves, phases and the re-check"""
    rs = np.random.RandomState(500 + seed + FUZZ_OFFSET)
    d = int(rs.choice([1, 2, 4, 9, 16, 27, 50, 64, 70]))
    na = int(rs.randint(33, 5000))
    nb = int(rs.choice([65, 300, 1025, 4099, 20001]))
    a = rs.normal(size=(na, d))
    b = rs.normal(size=(nb, d)) * rs.uniform(0.5, 1.2)
    b[::7] = a[rs.randint(na, size=len(b[::7]))]                  # exact copies: distance 0
    d2 = ((b[:200, None, :] - a[None, :500, :]) ** 2).sum(axis=2).min(axis=1)
    r2 = float(np.quantile(d2, rs.uniform(0.2, 0.8))) + 1e-12
    want = oracle.find_nearby(a, b, r2)
    for filt in (True, False):
        assert np.array_equal(_find(K, a, b, r2, filt), want), (filt, d, na, nb)


def test_second_range_ignores_stale_slots_of_an_unpadded_last_group(K):
    """the compacted query set of the second range is not padded: what an earlier batch -- here of another
    dimensionality, i.e. another operand layout -- left in the slots past the count must not act (it once reported
    "certain hits" to stale query numbers)"""
    import inputs
    from ultranest_amd import _lib
    reg = K.DeviceRegion()
    rs = np.random.RandomState(77)
    for rnd in range(6):
        for d, n, p in ((50, 1500, 20000), (10, 2744, 511), (3, 900, 300), (31, 700, 1000)):
            u = inputs.live_points(rnd * 10 + d, n, d)
            if d == 10:
                u = 0.5 + (u - 0.5) * np.geomspace(1.0, 0.02, d)
            ctr = u.mean(axis=0)
            cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
            ev, evec = np.linalg.eigh(cov)
            T = evec * ev ** -0.5
            inv = np.linalg.inv(cov)
            tl = (u - ctr) @ T
            dd = ((tl[:150, None, :] - tl[None, :150, :]) ** 2).sum(axis=2)
            np.fill_diagonal(dd, np.inf)
            r2 = float(np.sort(dd.min(axis=1))[100]) * float(rs.uniform(0.7, 1.5))
            reg.set(u, 0, ctr, T, None, ctr, inv, float(d) * 2.0, r2, live_space=1)
            pts = inputs.proposal_mix(rnd + d, u, p, shell_q=2.0)
            got = reg.inside(pts)
            _lib.set_option("filter", 0)
            want = reg.inside(pts)
            _lib.set_option("filter", 1)
            assert np.array_equal(got, want), (rnd, d, n, p, np.flatnonzero(got != want)[:5])
    reg.close()


def test_three_range_min_sweep_alternates_with_two_ranges(K):
    """"filter_second_range_pct": the min-only sweep in three ranges (the middle range carries the minima through a
    second array, the uncertain set lands in the first set's arrays, three slot counters return to zero in the tail).
    Batches alternate between two and three ranges on one handle, with the per-proposal stage inside the first launch and
    in front of it, cuts at both ends of what is accepted, and cuts the library ignores; every mask equals the exact
    scan's (MLFriends.inside, mlfriends.pyx:1186-1211)."""
    import inputs
    from ultranest_amd import _lib
    reg = K.DeviceRegion()
    rs = np.random.RandomState(91 + FUZZ_OFFSET)
    _lib.set_option("filter_phase_min_queries", 1024)
    _lib.set_option("filter_phases", 1)
    _lib.set_option("mid_max_queries", 0)
    _lib.set_option("filter_third_range_min_work", 0)
    try:
        for d, n, p in ((20, 3000, 9000), (50, 4000, 40001), (5, 700, 3001), (33, 1400, 5000)):
            u = inputs.live_points(5 + d, n, d)
            ctr = u.mean(axis=0)
            cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
            ev, evec = np.linalg.eigh(cov)
            T = evec * ev ** -0.5
            inv = np.linalg.inv(cov)
            tl = (u - ctr) @ T
            dd = ((tl[:150, None, :] - tl[None, :150, :]) ** 2).sum(axis=2)
            np.fill_diagonal(dd, np.inf)
            r2 = float(np.sort(dd.min(axis=1))[100]) * float(rs.uniform(0.7, 1.5))
            reg.set(u, 0, ctr, T, None, ctr, inv, float(d) * 2.0, r2, live_space=1)
            pts = inputs.proposal_mix(3 + d, u, p, shell_q=2.0)
            _lib.set_option("filter", 0)
            want = reg.inside(pts)
            _lib.set_option("filter", 1)
            ntiles = (n + 31) // 32
            for i, (first, second) in enumerate(((30, 0), (30, 50), (30, 0), (20, 40), (40, 50), (50, 90), (90, 90), (10, 10),
                                                 (60, 60), (30, 0), (30, 50))):
                _lib.set_option("filter_first_range_pct", first)
                _lib.set_option("filter_second_range_pct", second)
                _lib.set_option("fused_first_range", i % 3 != 1)
                got = reg.inside(pts)
                stats = reg.debug_stats()
                assert np.array_equal(got, want), (d, n, p, first, second, np.flatnonzero(got != want)[:5])
                c2 = ntiles * second // 100
                c1 = max(4, c2 * first // 100)
                three = second > 0 and c1 + 4 <= c2 <= ntiles - 4
                assert stats["range_cuts"] == ([c1, c2] if three else [max(4, min(ntiles - 4, ntiles * first // 100)), 0]), (first, second, stats)
                if three:
                    assert 0 < stats["third_range_groups"] <= stats["second_range_groups"], (first, second, stats)
        _lib.set_option("filter_third_range_min_work", 100000000)      # the default rule: these batches are too small for a third range
        _lib.set_option("filter_first_range_pct", 30)
        _lib.set_option("filter_second_range_pct", 50)
        assert np.array_equal(reg.inside(pts), want) and reg.debug_stats()["range_cuts"][1] == 0
    finally:
        _lib.set_option("filter_phase_min_queries", 64)
        _lib.set_option("filter_phases", 0 if _MODE["param"] == "single-sweep" else 1)
        _lib.set_option("mid_max_queries", 131072 if request_param_one_launch() else 0)
        _lib.set_option("filter_first_range_pct", 30)
        _lib.set_option("filter_second_range_pct", 50)
        _lib.set_option("filter_third_range_min_work", 100000000)
        _lib.set_option("fused_first_range", 1)
        reg.close()


def test_options_are_per_region_handle(K, oracle):
    """mlf_region_set_option overrides the process default for ONE handle: two regions of a process run different
    routings (the switches were process-global), give the same mask, and a released handle starts clean."""
    import inputs
    from ultranest_amd import _lib
    n, d, p = 500, 6, 4000
    u = inputs.live_points(77, n, d)
    ctr = u.mean(axis=0)
    cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    pts = inputs.proposal_mix(78, u, p, shell_q=2.0)
    a, b = K.DeviceRegion(), K.DeviceRegion()
    for reg in (a, b):
        reg.set(u, 0, ctr, T, None, ctr, inv, 40.0, 1.2, live_space=1)
    assert a.filter_info(p)[0] and b.filter_info(p)[0]
    b.set_option("filter", 0)
    assert a.filter_info(p)[0] and not b.filter_info(p)[0], "the override must reach one handle only"
    # the getter resolves through the handle (mlf_get_option alone reports the process default)
    assert a.get_option("filter") == 1 and b.get_option("filter") == 0 and _lib.get_option("filter") == 1
    ma, mb = a.inside(pts), b.inside(pts)
    tl = oracle.affine_transform(u, ctr, T)
    want = oracle.inside_ellipsoid(pts, ctr, inv, 40.0) & (oracle.find_nearby(tl, oracle.affine_transform(pts, ctr, T), 1.2) >= 0)
    assert np.array_equal(ma, want) and np.array_equal(mb, want)
    _lib.set_option("filter", 0)           # the process default moves: the handle without an override follows it
    try:
        assert not a.filter_info(p)[0]
        b.set_option("filter", 1)
        assert b.filter_info(p)[0]
    finally:
        _lib.set_option("filter", 1)
    b.set_option("filter")                 # back to the default
    assert b.filter_info(p)[0]
    b.set_option("filter", 0)
    b.release()
    c = K.DeviceRegion()                   # recycles b's handle: no override left
    c.set(u, 0, ctr, T, None, ctr, inv, 40.0, 1.2, live_space=1)
    assert c.filter_info(p)[0]
    with pytest.raises(ValueError, match="unknown option"):
        c.set_option("no_such_switch", 1)
    with pytest.raises(ValueError, match="unknown option"):
        c.get_option("no_such_switch")
    a.close()
    c.close()


def test_lds_grants_are_per_device_and_reissued(K, oracle):
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to (function, DEVICE).  Every kernel instance keeps a bit per
    device ordinal (csrc/mlf_common.hpp: DeviceGrant); mlf_debug_forget_grants clears them as a second device would find
    them.  A batch through the > 64 KiB-LDS kernels, `mlf_set_device` again, the grants forgotten, the same batch: the
    launches must ask again (the counter of issued grants moves) and give the same mask."""
    import inputs
    from ultranest_amd import _lib
    n, d, p = 2000, 20, 300000 if _MODE["param"] == "phased" else 5000
    u = inputs.live_points(91, n, d)
    ctr = u.mean(axis=0)
    cov = np.atleast_2d(np.cov(u, rowvar=0)) * (d + 2)
    ev, evec = np.linalg.eigh(cov)
    T = evec * ev ** -0.5
    inv = np.linalg.inv(cov)
    pts = inputs.proposal_mix(92, u, p, shell_q=2.0)
    reg = K.DeviceRegion()
    reg.set(u, 0, ctr, T, None, ctr, inv, 40.0, 1.1, live_space=1)
    first = reg.inside(pts)
    small = reg.inside(pts[:40])          # k_inside_small: its own grant
    before = _lib.forget_grants()
    _lib.set_device(0)
    again = reg.inside(pts)
    small_again = reg.inside(pts[:40])
    after = _lib.forget_grants()
    assert after > before, (before, after)
    assert np.array_equal(first, again) and np.array_equal(small, small_again)
    _lib.set_option("filter", 0)
    try:
        assert np.array_equal(reg.inside(pts), first)
    finally:
        _lib.set_option("filter", 1)
    reg.close()
