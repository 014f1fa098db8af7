import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import inputs
from ultranest_amd import _lib, kernels as K
_lib.set_option("filter_min_queries", 64); _lib.set_option("filter_phase_min_queries", 64)
seed = 3
rs = np.random.RandomState(1000 + seed)
d = int(rs.choice([1, 2, 3, 5, 7, 10, 13, 20, 31, 50, 63, 64]))
n = int(rs.randint(max(d + 2, 40), 3000))
p = int(rs.choice([257, 300, 511, 1000, 2047, 2049, 4001, 9999, 40001]))
u = inputs.live_points(seed, n, d)
if seed % 3 == 0:
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
print("d", d, "n", n, "p", p)
reg = K.DeviceRegion()
reg.set(u, 0, ctr, T, None, ctr, inv, enlarge, r2, live_space=1)
_lib.set_option("filter", 0)
exact = reg.inside(pts)
for k in range(5):
    assert np.array_equal(reg.inside(pts), exact)
_lib.set_option("filter", 1)
bad = {}
variants = {"default": {}, "no fold/wide": {"filter_narrow_tail": 0}, "binary64": {"prep_bounded": 0}, "single": {"filter_phases": 0}}
for rep in range(300):
    for name, opts in variants.items():
        for k_, v in opts.items(): _lib.set_option(k_, v)
        m = reg.inside(pts)
        for k_, v in (("prep_bounded", 1), ("filter_phases", 1), ("filter_narrow_tail", 1)): _lib.set_option(k_, v)
        if not np.array_equal(m, exact):
            bad.setdefault(name, []).append((rep, np.flatnonzero(m != exact).tolist(), m[np.flatnonzero(m != exact)].tolist()))
print({k: (len(v), v[:3]) for k, v in bad.items()})
print(reg.debug_stats())
