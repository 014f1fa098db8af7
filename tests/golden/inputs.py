"""Seeded input generators shared by tests/golden/make_golden.py (which runs the
real reference in the dev container) and the parity tests (which run anywhere).

All draws come from legacy ``np.random.RandomState`` streams, which numpy keeps
frozen across versions, and use no BLAS/LAPACK, so inputs regenerate bit-exactly
on any machine and only the reference's OUTPUTS need to be committed.
"""
import numpy as np

# (name, N live points, d dims, P query points)
FIND_NEARBY_CASES = [
    ("c1", 400, 5, 1000),
    ("c2", 2000, 20, 4096),
    ("c5", 4000, 50, 2048),
    ("odd", 257, 3, 129),      # ragged sizes: not multiples of 64
    ("wide", 300, 33, 200),    # odd dimensionality
]

BOOTSTRAP_CASES = [
    ("c1", 400, 5, 30),
    ("c2", 2000, 20, 30),
    ("c4", 4000, 50, 30),
    ("tiny", 70, 2, 40),       # B > 32 exercises bootstrap-group chunking
]


def live_points(seed, n, d):
    """Live points in the unit cube, SURVEY.md 8(d): 0.5 + 0.05*normal."""
    rs = np.random.RandomState(seed)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    assert np.logical_and(u > 0, u < 1).all()
    return u


def find_nearby_inputs(seed, n, d, p):
    """apts/bpts drawn directly from the stream (no LAPACK in between)."""
    rs = np.random.RandomState(seed)
    apts = rs.normal(size=(n, d))
    # half the queries sit close to an a-point (so that roughly half hit),
    # the other half are fresh draws
    bpts = rs.normal(size=(p, d))
    near = rs.randint(n, size=p)
    scale = rs.uniform(0.2, 1.2, size=(p, 1))
    bpts = np.where((np.arange(p) % 2 == 0)[:, None], apts[near] + scale * 0.5 * bpts, bpts)
    # one query coincides with an a-point: distance exactly 0 (the `<=` pin,
    # reference tests/test_regionsampling.py:46-48)
    bpts[p // 3] = apts[n // 2]
    return np.ascontiguousarray(apts), np.ascontiguousarray(bpts)


def median_nn_radius(apts, bpts):
    """A radius^2 that makes about half of bpts hit: median over queries of the
    squared distance to the nearest a-point (numpy, any summation order --
    only used to PICK r2, which is then stored in the fixture)."""
    best = np.full(len(bpts), np.inf)
    for lo in range(0, len(apts), 512):
        blk = apts[lo:lo + 512]
        d2 = ((bpts[:, None, :] - blk[None, :, :]) ** 2).sum(axis=2)
        best = np.minimum(best, d2.min(axis=1))
    return float(np.median(best))


def two_blobs(seed, n, d, sep=0.3, sigma=0.02):
    """Two well separated blobs in the unit cube (cf. reference tests/test_run.py:14-23)."""
    rs = np.random.RandomState(seed)
    u = 0.5 + sigma * rs.normal(size=(n, d))
    u[: n // 2, 0] -= sep / 2
    u[n // 2:, 0] += sep / 2
    assert np.logical_and(u > 0, u < 1).all()
    return u


def proposal_mix(seed, u, p, shell_q=2.0):
    """Proposal batch around a live-point set, four interleaved components:
    jittered copies of live points, a wider Gaussian, uniform cube draws, and a
    shell whose Mahalanobis radius^2 straddles ``shell_q`` (the wrapping
    ellipsoid's enlargement) -- the shell is where the neighbour scan rejects."""
    rs = np.random.RandomState(seed)
    n, d = u.shape
    sig = u.std(axis=0).mean()
    ctr = u.mean(axis=0)
    a = u[rs.randint(n, size=p)] + rs.normal(size=(p, d)) * sig * rs.uniform(0.05, 0.6, size=(p, 1))
    b = ctr + rs.normal(size=(p, d)) * sig * 1.3
    c = rs.uniform(size=(p, d))
    z = rs.normal(size=(p, d))
    z /= np.sqrt((z ** 2).sum(axis=1, keepdims=True))
    q = shell_q * rs.uniform(0.88, 1.04, size=(p, 1))
    s = ctr + z * sig * np.sqrt(q * (d + 2))
    which = (np.arange(p) % 4)[:, None]
    pts = np.where(which == 0, a, np.where(which == 1, b, np.where(which == 2, c, s)))
    return np.ascontiguousarray(pts)


def likelihood_inputs(seed, n, d, lo, hi):
    rs = np.random.RandomState(seed)
    return rs.uniform(lo, hi, size=(n, d))
