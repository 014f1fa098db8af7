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


# ---------------------------------------------------------------- G9: population step sampler ----

STEP_CASES = [(901, 257, 5), (902, 64, 50), (903, 1, 3), (904, 1000, 2)]


def walker_state(seed, n, d):
    """One population of slice-sampling walkers in a mix of the three states (stepping out left,
    stepping out right, bisecting), some of them about to leave the unit cube."""
    rs = np.random.RandomState(seed)
    s = dict()
    s["currentu"] = rs.uniform(0.02, 0.98, size=(n, d))
    s["currentv"] = rs.normal(size=(n, d)) * 0.3
    s["current_left"] = -rs.uniform(0.01, 2.0, size=n)
    s["current_right"] = rs.uniform(0.01, 2.0, size=n)
    state = rs.randint(4, size=n)          # 0 both, 1 right only, 2 bisecting, 3 left only
    s["searching_left"] = np.logical_or(state == 0, state == 3)
    s["searching_right"] = np.logical_or(state == 0, state == 1)
    s["currentt"] = rs.uniform(-1, 1, size=n) * np.where(state == 2, 1.0, 0.0)
    s["currentL"] = rs.normal(size=n)
    s["Lmin"] = -0.3
    return s


def walker_loglike(p):
    """Likelihood callback used for the evolve / sampler-trace vectors (numpy on both sides)."""
    return -0.5 * (((p - 0.5) / 0.2)**2).sum(axis=1)


def walker_transform(u):
    return u * 1.0


def update_inputs(seed, n):
    """Inputs of evolve_update drawn independently of any geometry."""
    rs = np.random.RandomState(seed)
    s = walker_state(seed + 1000, n, 2)
    acceptable = rs.uniform(size=n) < 0.8
    Lnew = rs.normal(size=int(acceptable.sum()))
    return s, acceptable, Lnew


def step_back_state(seed, n, ngen):
    """Chains of different lengths whose likelihoods partly fall below the new threshold."""
    rs = np.random.RandomState(seed)
    generation = rs.randint(-1, ngen, size=n).astype(np.int64)
    allL = np.full((n, ngen), np.nan)
    for i in range(n):
        g = generation[i]
        if g >= 0:
            allL[i, :g + 1] = np.sort(rs.normal(size=g + 1)) if rs.uniform() < 0.5 else rs.normal(size=g + 1)
    currentt = rs.normal(size=n)
    currentt[rs.uniform(size=n) < 0.2] = np.nan
    return allL, generation, currentt, 0.1


def line_inputs(seed, n, d):
    rs = np.random.RandomState(seed)
    origin = rs.uniform(size=(n, d))
    direction = rs.normal(size=(n, d))
    direction[rs.uniform(size=(n, d)) < 0.15] = 0.0       # axis-parallel components
    direction[0, :] = 0.0
    direction[0, 0] = 1.0
    origin[1, 0] = 0.5
    direction[1, 0] = 0.0                                  # 0 * inf -> NaN entry, must be skipped
    origin[2 % n, :] = 0.0                                 # starts on the boundary
    return origin, direction


def slice_update_inputs(seed, popsize, d, nparams, npoints_busy):
    """State of PopulationSimpleSliceSampler in the middle of its shrinking loop: several
    workers serve the same unfinished point."""
    rs = np.random.RandomState(seed)
    status = np.ones(popsize, dtype=np.int64)
    busy = rs.choice(popsize, size=npoints_busy, replace=False)
    status[busy] = 0
    worker_running = np.sort(busy[rs.randint(npoints_busy, size=popsize)]).astype(np.int64)
    tleft = -rs.uniform(0.2, 1.0, size=popsize)
    tright = rs.uniform(0.2, 1.0, size=popsize)
    t = rs.uniform(-1.1, 1.1, size=popsize)
    proposed_L = rs.normal(size=popsize)
    proposed_u = rs.uniform(size=(popsize, d))
    proposed_p = rs.normal(size=(popsize, nparams))
    allu = rs.uniform(size=(popsize, d))
    allL = rs.normal(size=popsize)
    allp = rs.normal(size=(popsize, nparams))
    return dict(t=t, tleft=tleft, tright=tright, proposed_L=proposed_L, proposed_u=proposed_u,
                proposed_p=proposed_p, worker_running=worker_running, status=status, allu=allu, allL=allL,
                allp=allp, threshold=0.2)


# ---------------------------------------------------------------- G10: tree + counters ----------

def random_tree(seed, nroots, nnodes):
    """A nested-sampling-like tree as plain lists: node k has value values[k] and the children
    children[k] (indices); roots are nodes 0 .. nroots-1.  Mostly single-child chains with rising
    values, some branchings (added live points) and some dead ends, like a reactive run."""
    rs = np.random.RandomState(seed)
    values = list(np.sort(rs.normal(size=nroots)) - 5.0)
    rs.shuffle(values)
    children = [[] for _ in range(nroots)]
    active = list(range(nroots))
    while len(values) < nnodes and active:
        k = min(active, key=lambda i: values[i])
        active.remove(k)
        roll = rs.uniform()
        nkids = 1 if roll < 0.93 else (2 if roll < 0.97 else 0)
        for _ in range(nkids):
            top = max(values[i] for i in active) if active else values[k] + 1
            v = values[k] + (top - values[k]) * rs.uniform() + 1e-3 * rs.uniform()
            values.append(float(v))
            children.append([])
            children[k].append(len(values) - 1)
            active.append(len(values) - 1)
    return np.array(values), children


def build_nodes(cls, values, children, nroots):
    nodes = [cls(value=float(v), id=i) for i, v in enumerate(values)]
    for node, kids in zip(nodes, children):
        node.children = [nodes[j] for j in kids]
    return nodes[:nroots]


def tree_points(seed, nnodes, udim=3):
    """Coordinates for the nodes of `random_tree(seed, ...)`: unit-cube points and a 'transformed' copy
    with one derived column (pdim = udim + 1)."""
    rs = np.random.RandomState(seed + 77)
    u = rs.uniform(size=(nnodes, udim))
    p = np.hstack((10. * u - 5., (u ** 2).sum(axis=1, keepdims=True)))
    return u, p


RESULT_TREES = [(1201, 20, 160, 6), (1202, 8, 90, 3)]      # (seed, nroots, nnodes, nbootstraps of the run counter)
RESULT_PARAMNAMES = (["a", "b", "c"], ["r2"])


# ---- dump_tree through a recording stand-in for h5py (g14; h5py is not in this image) ----
class RecordingH5File(object):
    """Stand-in for ``h5py.File`` (h5py is not in this image): records what the writer hands to ``create_dataset``."""
    written = {}

    def __init__(self, filename, mode):
        RecordingH5File.written = dict(filename=filename, mode=mode, datasets={})

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def create_dataset(self, name, data=None, **options):
        RecordingH5File.written["datasets"][name] = (np.array(data), options)


def recorded_tree_dump(netiter, seed, nroots, nnodes):
    """`dump_tree` of a seeded tree (the trees of g12) with the recording stand-in as h5py"""
    import sys
    import types
    values, children = random_tree(seed, nroots, nnodes)
    us, ps = tree_points(seed, len(values))
    pile = netiter.PointPile(us.shape[1], ps.shape[1])
    for urow, prow in zip(us, ps):
        pile.add(urow, prow)
    roots = build_nodes(netiter.TreeNode, values, children, nroots)
    fake = types.ModuleType("h5py")
    fake.File = RecordingH5File
    saved = sys.modules.get("h5py")
    sys.modules["h5py"] = fake
    try:
        netiter.dump_tree("tree_%d.hdf5" % seed, roots, pile)
    finally:
        if saved is None:
            del sys.modules["h5py"]
        else:
            sys.modules["h5py"] = saved
    return RecordingH5File.written


# ---- g16: the two batched population samplers (reference popstepsampler.py:192-358, 746-1001) ----
BATCHED_SAMPLER_CASES = [
    # name, class, direction generator, constructor keywords (slice_limit: "scale" = slice_limit_to_scale), popsize, nsteps, d
    ("walk_random", "PopulationRandomWalkSampler", "generate_random_direction", dict(scale=0.05), 12, 25, 4),
    ("walk_cube", "PopulationRandomWalkSampler", "generate_cube_oriented_direction", dict(scale=0.05, scale_adapt_factor=0.8), 7, 30, 3),
    ("slice_random", "PopulationSimpleSliceSampler", "generate_random_direction", dict(), 12, 6, 4),
    ("slice_mixture_scaled", "PopulationSimpleSliceSampler", "generate_mixture_random_direction",
     dict(scale=0.5, scale_adapt_factor=0.9, shrink_factor=1.5, slice_limit="scale"), 20, 5, 6),
    ("slice_region_one", "PopulationSimpleSliceSampler", "generate_region_oriented_direction", dict(max_it=7), 1, 4, 2)]


def batched_sampler_problem(mod, d, seed=3, n=200):
    """live points, region, Gaussian likelihood and threshold of the batched-sampler traces (shared with the tests)"""
    rs = np.random.RandomState(seed)
    u = np.clip(0.5 + 0.1 * rs.normal(size=(n, d)), 1e-3, 1 - 1e-3)
    layer = mod.AffineLayer()
    layer.optimize(u, u)
    region = mod.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
    region.create_ellipsoid()

    def loglike(p):
        return -0.5 * (((p - 0.5) / 0.1) ** 2).sum(axis=1)

    Ls = loglike(u)
    return u, region, loglike, Ls, float(np.sort(Ls)[20])
