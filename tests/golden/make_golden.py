#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the dev container (needs /root/reference, cython, gcc).  It
cythonizes ultranest/mlfriends.pyx + stepfuncs.pyx into a throw-away directory
OUTSIDE the repository (nothing of the reference -- source, generated C, .so or
bytecode -- is written into the repo or travels to the GPU box), imports the
package from there and records input/output vectors as small .npz fixtures.

    python tests/golden/make_golden.py            # regenerate everything
    python tests/golden/make_golden.py g1 g3      # only some groups

Inputs are regenerated from seeds by tests/golden/inputs.py (RandomState, no
LAPACK) so mostly OUTPUTS are stored.  Where a reference stage depends on
BLAS/LAPACK (np.dot, eigh, inv, cov) the stage's OUTPUT matrices are stored and
fed to the later stages, as SURVEY.md 8(c) prescribes.
"""
import glob
import json
import os
import subprocess
import sys
import sysconfig
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs  # noqa: E402

REF = os.environ.get("ULTRANEST_REFERENCE", "/root/reference")


def build_reference():
    """Cythonize + compile the reference's two extension modules in a scratch dir."""
    scratch = os.environ.get("ULTRANEST_REFBUILD") or os.path.join(tempfile.gettempdir(), "ultranest_refbuild")
    pkg = os.path.join(scratch, "ultranest")
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    if not os.path.exists(os.path.join(pkg, "mlfriends" + suffix)):
        os.makedirs(pkg, exist_ok=True)
        for f in glob.glob(os.path.join(REF, "ultranest", "*.py")):
            dst = os.path.join(pkg, os.path.basename(f))
            if not os.path.lexists(dst):
                os.symlink(f, dst)
        inc = [np.get_include(), sysconfig.get_paths()["include"]]
        for mod in ("mlfriends", "stepfuncs"):
            c = os.path.join(pkg, mod + ".c")
            subprocess.check_call(["cython", "-3", os.path.join(REF, "ultranest", mod + ".pyx"), "-o", c])
            subprocess.check_call(["gcc", "-O3", "-shared", "-fPIC", "-w"] + ["-I" + i for i in inc]
                                  + [c, "-o", os.path.join(pkg, mod + suffix)])
    sys.path.insert(0, scratch)
    import ultranest.mlfriends as ref
    return ref


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("  wrote %-28s %7.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024.))


# ------------------------------------------------------------------ G1 ------

def g1(ref):
    """find_nearby / count_nearby (via sample_from_points is indirect, so the
    count golden comes from find_nearby-consistent full scans below)."""
    out = {}
    for i, (name, n, d, p) in enumerate(inputs.FIND_NEARBY_CASES):
        apts, bpts = inputs.find_nearby_inputs(100 + i, n, d, p)
        r2 = inputs.median_nn_radius(apts, bpts)
        for tag, rr in (("mid", r2), ("none", 1e-300), ("all", 1e300), ("big", 4 * r2)):
            idx = np.empty(p, dtype=np.int64)
            ref.find_nearby(apts, bpts, rr, idx)
            out["%s_%s_idx" % (name, tag)] = idx
            out["%s_%s_r2" % (name, tag)] = np.float64(rr)
        hit = out[name + "_mid_idx"] >= 0
        print("  G1 %-5s N=%d d=%d P=%d  r2=%.6g hit fraction %.3f" % (name, n, d, p, r2, hit.mean()))
        assert out[name + "_none_idx"][p // 3] >= 0, "coincident point must hit at r2=1e-300"
    save("g1_find_nearby", **out)


# ------------------------------------------------------------------ G2 ------

def g2(ref):
    out = {}
    for tag, n, d in (("a", 300, 4), ("b", 1000, 7)):
        u = inputs.two_blobs(200, n, d)
        # radius: a bit more than the typical nearest-neighbour distance
        nn = []
        for j in range(n):
            d2 = ((u - u[j]) ** 2).sum(axis=1)
            d2[j] = np.inf
            nn.append(d2.min())
        r2 = float(np.max(nn)) * 2.0
        out[tag + "_r2"] = np.float64(r2)
        out[tag + "_subtract"] = ref.subtract_nearby(u, r2)
        nclusters, ids, overlapped = ref.update_clusters(u, u, r2)
        out[tag + "_nclusters"] = np.int64(nclusters)
        out[tag + "_ids"] = ids
        out[tag + "_overlapped"] = overlapped
        # re-clustering with previous ids as seeds and a smaller radius
        nclusters2, ids2, overlapped2 = ref.update_clusters(u, u, r2 / 6, ids)
        out[tag + "_r2b"] = np.float64(r2 / 6)
        out[tag + "_nclusters2"] = np.int64(nclusters2)
        out[tag + "_ids2"] = ids2
        out[tag + "_mpd"] = np.float64(ref.compute_mean_pair_distance(u, ids))
        ids_z = ids.copy()
        ids_z[::5] = 0
        out[tag + "_mpd_z"] = np.float64(ref.compute_mean_pair_distance(u, ids_z))
        print("  G2 %s N=%d d=%d r2=%.4g nclusters=%d/%d mpd=%.6g" % (tag, n, d, r2, nclusters, nclusters2, out[tag + "_mpd"]))
    # the reference's own fixture (tests/test_clustering.py:38-48)
    pts = np.loadtxt(os.path.join(REF, "tests", "clusters2.txt"))
    rad = float(np.loadtxt(os.path.join(REF, "tests", "clusters2_radius.txt")))
    nclusters, ids, overlapped = ref.update_clusters(pts, pts, rad)
    out["ref_clusters2_pts"] = pts
    out["ref_clusters2_r2"] = np.float64(rad)
    out["ref_clusters2_nclusters"] = np.int64(nclusters)
    out["ref_clusters2_ids"] = ids
    save("g2_clusters", **out)


# ------------------------------------------------------------------ G3 ------

def g3(ref):
    """Bootstrapped radius/enlargement.  ScalingLayer() (mean 0, std 1) makes
    unormed == u bit-exactly, so K4's inputs carry no LAPACK dependence."""
    out = {}
    for i, (name, n, d, B) in enumerate(inputs.BOOTSTRAP_CASES):
        u = inputs.live_points(300 + i, n, d)
        region = ref.MLFriends(u, ref.ScalingLayer())
        assert np.array_equal(region.unormed, u)
        rs = np.random.RandomState(900 + i)
        rr, ff = [], []
        t0 = time.time()
        for b in range(B):
            r, f = region.compute_enlargement(nbootstraps=1, minvol=0., rng=rs)
            rr.append(r)
            ff.append(f)
        rr, ff = np.array(rr), np.array(ff)
        assert np.array_equal(rr, rr.astype(np.float32).astype(np.float64)), "r2 must be float32-representable"
        # aggregate call on a fresh stream with the same seed
        r30, f30 = region.compute_enlargement(nbootstraps=B, minvol=0., rng=np.random.RandomState(900 + i))
        assert r30 == rr.max() and f30 == ff.max()
        out[name + "_r"] = rr
        out[name + "_f"] = ff
        # ellipsoid-only regions on the same stream
        rob = ref.RobustEllipsoidRegion(u, ref.ScalingLayer())
        r_rob, f_rob = rob.compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + i))
        out[name + "_rob"] = np.array([r_rob, f_rob])
        sim = ref.SimpleRegion(u, ref.ScalingLayer())
        r_sim, f_sim = sim.compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + i))
        out[name + "_sim"] = np.array([r_sim, f_sim])
        wrap = ref.WrappingEllipsoid(u)
        out[name + "_wrap"] = np.float64(wrap.compute_enlargement(nbootstraps=B, rng=np.random.RandomState(900 + i)))
        print("  G3 %-5s N=%d d=%d B=%d  r2=%.9g f=%.9g  (%.1fs)" % (name, n, d, B, r30, f30, time.time() - t0))
    # the reference's own fixture and recipe: tests/test_clustering.py:81-99
    # (ScalingLayer.optimize, global-RNG compute_maxradiussq(30) for seeds 0-9,
    #  then update_clusters(points, points, maxr); pinned 1e-10 < maxr < 6e-10, 14 < nclusters < 20)
    pts = np.loadtxt(os.path.join(REF, "tests", "eggboxregion.txt"))
    layer = ref.ScalingLayer()
    layer.optimize(pts, pts)
    maxrs, ncl = [], []
    for seed in range(10):
        np.random.seed(seed)
        region = ref.MLFriends(pts, layer)
        maxr = region.compute_maxradiussq(nbootstraps=30)
        assert 1e-10 < maxr < 6e-10
        nclusters, clusteridxs, _ = ref.update_clusters(pts, pts, maxr)
        assert 14 < nclusters < 20
        maxrs.append(maxr)
        ncl.append(nclusters)
    out["eggbox_pts"] = pts
    out["eggbox_mean"] = layer.mean
    out["eggbox_std"] = layer.std
    out["eggbox_unormed"] = region.unormed
    out["eggbox_maxr"] = np.array(maxrs)
    out["eggbox_nclusters"] = np.array(ncl)
    out["eggbox_ids_seed9"] = clusteridxs
    print("  G3 eggboxregion: maxr", maxrs[:3], "nclusters", ncl)
    save("g3_bootstrap", **out)


# ------------------------------------------------------------- G4/G5/G6 ------

def _region_with_layer(ref, u, layer_cls, seed, B=30):
    layer = layer_cls()
    layer.optimize(u, u)
    region = ref.MLFriends(u, layer)
    r, f = region.compute_enlargement(nbootstraps=B, rng=np.random.RandomState(seed))
    region.maxradiussq, region.enlarge = r, f
    region.create_ellipsoid(minvol=0.0)
    return layer, region


def g456(ref):
    out = {}
    for i, (name, n, d, p) in enumerate((("c1", 400, 5, 3000), ("c2", 2000, 20, 3000), ("c5", 4000, 50, 1536))):
        u = inputs.live_points(400 + i, n, d)
        layer, region = _region_with_layer(ref, u, ref.AffineLayer, 950 + i)
        pts = inputs.proposal_mix(500 + i, u, p, shell_q=region.enlarge)
        mask = region.inside(pts)
        emask = region.inside_ellipsoid(pts)
        # margins to both thresholds (numpy, float64): min relative distance
        delta = pts - region.ellipsoid_center
        q = np.einsum('ij,jk,ik->i', delta, region.ellipsoid_invcov, delta)
        m_ell = np.abs(q - region.enlarge) / region.enlarge
        t = layer.transform(pts[emask])
        m_scan = np.full(emask.sum(), np.inf)
        for lo in range(0, n, 256):
            d2 = ((t[:, None, :] - region.unormed[None, lo:lo + 256, :]) ** 2).sum(axis=2)
            m_scan = np.minimum(m_scan, (np.abs(d2 - region.maxradiussq) / region.maxradiussq).min(axis=1))
        margin = min(m_ell.min(), m_scan.min())
        assert margin > 1e-9, margin
        # same region with an artificially tighter radius: many scan rejections
        r2_full = region.maxradiussq
        for frac in (0.7, 0.55, 0.4):
            region.maxradiussq = r2_full * frac
            mask_t = region.inside(pts)
            m_t = np.full(emask.sum(), np.inf)
            for lo in range(0, n, 256):
                d2 = ((t[:, None, :] - region.unormed[None, lo:lo + 256, :]) ** 2).sum(axis=2)
                m_t = np.minimum(m_t, (np.abs(d2 - region.maxradiussq) / region.maxradiussq).min(axis=1))
            if 0.15 < mask_t[emask].mean() < 0.85:
                break
        assert m_t.min() > 1e-9, m_t.min()
        margin = min(margin, m_t.min())
        out[name + "_r2_tight"] = np.float64(region.maxradiussq)
        out[name + "_mask_tight"] = np.packbits(mask_t)
        print("     tight radius frac %.2f: scan accepts %.3f of ellipsoid-passing points" % (frac, mask_t[emask].mean()))
        region.maxradiussq = r2_full
        out[name + "_layer_ctr"] = layer.ctr
        out[name + "_layer_T"] = layer.T
        out[name + "_layer_invT"] = layer.invT
        out[name + "_layer_logvolscale"] = np.float64(layer.logvolscale)
        out[name + "_r2"] = np.float64(region.maxradiussq)
        out[name + "_enlarge"] = np.float64(region.enlarge)
        out[name + "_ell_center"] = region.ellipsoid_center
        out[name + "_ell_cov"] = region.ellipsoid_cov
        out[name + "_ell_invcov"] = region.ellipsoid_invcov
        out[name + "_ell_axes"] = region.ellipsoid_axes
        out[name + "_ell_axlens"] = region.ellipsoid_axlens
        out[name + "_ell_inv_axes"] = region.ellipsoid_inv_axes
        out[name + "_volume"] = np.float64(region.estimate_volume())
        out[name + "_mask"] = np.packbits(mask)
        out[name + "_emask"] = np.packbits(emask)
        out[name + "_margin"] = np.float64(margin)
        out[name + "_inside_live"] = np.bool_(region.inside(u).all())
        print("  G4 %-3s N=%d d=%d P=%d  r2=%.6g enlarge=%.6g  ellipsoid pass %.3f  inside %.3f  margin %.2e" % (
            name, n, d, p, region.maxradiussq, region.enlarge, emask.mean(), mask.mean(), margin))
    # G5: metric-learning layers on clustered data = the reference's own
    # eggboxregion.txt fixture (tests/test_clustering.py:81-99), two generations
    # as the integrator iterates them (integrator.py:2068-2091)
    u = np.loadtxt(os.path.join(REF, "tests", "eggboxregion.txt"))
    out["g5_u"] = u
    for lname, cls in (("affine", ref.AffineLayer), ("local", ref.LocalAffineLayer),
                       ("gap", ref.MaxPrincipleGapAffineLayer), ("scaling", ref.ScalingLayer)):
        layer = cls()
        layer.optimize(u, u)
        for gen in (1, 2):
            region = ref.MLFriends(u, layer)
            r, f = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(77 + gen))
            nxt = layer.create_new(u, r)
            key = "g5_%s%d_" % (lname, gen)
            out[key + "r_f"] = np.array([r, f])
            out[key + "nclusters"] = np.int64(nxt.nclusters)
            out[key + "ids"] = nxt.clusterids
            out[key + "logvolscale"] = np.float64(nxt.logvolscale)
            if lname == "scaling":
                out[key + "mean"] = nxt.mean
                out[key + "std"] = nxt.std
            else:
                out[key + "ctr"] = nxt.ctr
                out[key + "cov"] = nxt.cov
                out[key + "T"] = nxt.T
            print("  G5 %-7s gen %d r2=%.6g nclusters=%d logvolscale=%.6f" % (lname, gen, r, nxt.nclusters, nxt.logvolscale))
            layer = nxt
    # wrapped dims round trip (tests/test_transforms.py:89-124 style)
    rs = np.random.RandomState(8)
    uw = np.column_stack([np.fmod(0.9 + 0.15 * rs.uniform(size=400), 1.0), 0.3 + 0.2 * rs.uniform(size=400)])
    layer = ref.AffineLayer(wrapped_dims=[0])
    layer.optimize(uw, uw)
    out["g5_wrap_u"] = uw
    out["g5_wrap_cuts"] = np.array(layer.wrap_cuts)
    out["g5_wrap_t"] = layer.transform(uw)
    out["g5_wrap_ctr"] = layer.ctr
    out["g5_wrap_T"] = layer.T
    save("g456_region", **out)


# ------------------------------------------------------------------ G7 ------

def g7(ref):
    out = {}
    n = 1024
    centers5 = np.ones(5) * 0.5
    x = inputs.likelihood_inputs(700, n, 5, 0.45, 0.55)
    out["gauss5"] = -0.5 * (((x - centers5) / 0.01) ** 2).sum(axis=1) - 0.5 * np.log(2 * np.pi * 0.01 ** 2) * 5
    ndim = 20
    sigma = 0.1
    width = max(0, 1 - 5 * sigma)
    centers = (np.sin(np.arange(ndim) / 2.) * width + 1.) / 2.
    x = inputs.likelihood_inputs(701, n, ndim, 0, 1)
    out["gauss20_centers"] = centers
    out["gauss20"] = -0.5 * (((x - centers) / sigma) ** 2).sum(axis=1) - 0.5 * np.log(2 * np.pi * sigma ** 2) * ndim
    for d in (2, 10):
        z = inputs.likelihood_inputs(702 + d, n, d, 0, 1) * 10 * np.pi
        out["eggbox%d" % d] = (2. + (np.cos(z / 2.)).prod(axis=1)) ** 5
        out["eggboxsq%d" % d] = np.cos(z).prod(axis=1) ** 2
    for d in (2, 50):
        theta = inputs.likelihood_inputs(720 + d, n, d, 0, 1) * 20 - 10
        a, b = theta[:, :-1], theta[:, 1:]
        out["rosenbrock%d" % d] = -2 * (100 * (b - a ** 2) ** 2 + (1 - a) ** 2).sum(axis=1)
    save("g7_likelihoods", **out)


# ------------------------------------------------------------------ G8 ------

def g8(ref):
    """End-to-end C1 plumbing run on the stock reference (5-d Gaussian, sigma
    0.01, 400 live points, vectorized=True)."""
    import ultranest
    ndim = 5
    centers = np.ones(ndim) * 0.5
    sigma = 0.01

    def loglike(theta):
        return -0.5 * (((theta - centers) / sigma) ** 2).sum(axis=1) - 0.5 * np.log(2 * np.pi * sigma ** 2) * ndim

    np.random.seed(1)
    t0 = time.time()
    sampler = ultranest.ReactiveNestedSampler(["p%d" % i for i in range(ndim)], loglike,
                                              transform=lambda x: x, vectorized=True, log_dir=None)
    res = sampler.run(min_num_live_points=400, viz_callback=None, show_status=False)
    meta = dict(logz=res["logz"], logzerr=res["logzerr"], ncall=int(res["ncall"]), niter=int(res["niter"]),
                seconds=time.time() - t0, ultranest=ultranest.__version__, numpy=np.__version__)
    print("  G8", meta)
    with open(os.path.join(HERE, "g8_c1_run.json"), "w") as f:
        json.dump(meta, f, indent=1)


# ------------------------------------------------------------------ G9 ------

DIRECTIONS = ["generate_cube_oriented_direction", "generate_cube_oriented_direction_scaled",
              "generate_random_direction", "generate_region_oriented_direction",
              "generate_region_random_direction", "generate_differential_direction",
              "generate_mixture_random_direction"]


def g9(ref):
    """Population step-sampler state machine: every function of
    ultranest/stepfuncs.pyx plus the geometry helpers and the PopulationSliceSampler loop of
    ultranest/popstepsampler.py, run on seeded inputs."""
    import ultranest.popstepsampler as pop
    import ultranest.stepfuncs as sf
    out = {}
    # evolve_update / evolve_prepare / within_unit_cube
    for seed, n, d in inputs.STEP_CASES:
        s, acceptable, Lnew = inputs.update_inputs(seed, n)
        search_right, bisecting = sf.evolve_prepare(s["searching_left"], s["searching_right"])
        success = np.zeros(n, dtype=bool)
        t, lo, hi = s["currentt"].copy(), s["current_left"].copy(), s["current_right"].copy()
        sl, sr = s["searching_left"].copy(), s["searching_right"].copy()
        sf.evolve_update(acceptable, Lnew, s["Lmin"], search_right, bisecting, t, lo, hi, sl, sr, success)
        k = "upd%d_" % seed
        out.update({k + "search_right": search_right, k + "bisecting": bisecting, k + "t": t, k + "left": lo,
                    k + "right": hi, k + "sl": sl, k + "sr": sr, k + "success": success})
        # full evolve with the global numpy stream
        s = inputs.walker_state(seed, n, d)
        np.random.seed(seed)
        args = [s[key].copy() for key in ("currentu", "currentL", "currentt", "currentv", "current_left",
                                          "current_right", "searching_left", "searching_right")]
        (t, v, lo, hi, sl, sr), (success, unew, pnew, Lnew), nc = sf.evolve(
            inputs.walker_transform, inputs.walker_loglike, s["Lmin"], *args)
        k = "evo%d_" % seed
        out.update({k + "t": t, k + "left": lo, k + "right": hi, k + "sl": sl, k + "sr": sr,
                    k + "success": success, k + "unew": unew, k + "pnew": pnew, k + "Lnew": Lnew,
                    k + "nc": np.int64(nc), k + "currentu_after": args[0],
                    k + "cube": sf.within_unit_cube(args[0]), k + "next_random": np.random.uniform()})
    # step_back
    for seed, n, ngen in [(911, 40, 6), (912, 300, 21), (913, 5, 1)]:
        allL, generation, currentt, Lmin = inputs.step_back_state(seed, n, ngen)
        sf.step_back(Lmin, allL, generation, currentt)
        k = "back%d_" % seed
        out.update({k + "allL": allL, k + "generation": generation, k + "t": currentt})
    # unit-cube line intersection
    for seed, n, d in [(921, 200, 3), (922, 50, 50), (923, 7, 1)]:
        origin, direction = inputs.line_inputs(seed, n, d)
        with np.errstate(all="ignore"):
            lo, hi = pop.unitcube_line_intersection(origin, direction)
        out["line%d_left" % seed], out["line%d_right" % seed] = lo, hi
    # shrinking loop of the simple slice sampler
    for seed, popsize, d, nparams, busy in [(931, 12, 1, 1, 3), (932, 200, 7, 9, 40), (933, 64, 3, 3, 1)]:
        for shrink in (1.0, 1.5):
            a = inputs.slice_update_inputs(seed, popsize, d, nparams, busy)
            res = sf.update_vectorised_slice_sampler(
                a["t"], a["tleft"], a["tright"], a["proposed_L"], a["proposed_u"], a["proposed_p"],
                a["worker_running"], a["status"], a["threshold"], shrink, a["allu"], a["allL"], a["allp"], popsize)
            k = "slice%d_%d_" % (seed, int(shrink * 10))
            for name, val in zip(("tleft", "tright", "worker_running", "status", "allu", "allL", "allp", "discarded"), res):
                out[k + name] = np.asarray(val)
    # direction generators + move diagnostics on a real region
    u = inputs.live_points(940, 300, 6)
    rng = np.random.RandomState(940)
    layer = ref.AffineLayer()
    layer.optimize(u, u)
    region = ref.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rng)
    region.create_ellipsoid()
    out["region_axes"] = np.array(layer.axes)
    out["region_T"], out["region_ctr"] = np.array(layer.T), np.array(layer.ctr)
    out["region_maxradiussq"] = np.float64(region.maxradiussq)
    ui = u[:50]
    for name in DIRECTIONS:
        np.random.seed(941)
        out["dir_" + name] = getattr(sf, name)(ui, region, scale=0.7)
    ufinal = np.clip(ui + rng.normal(size=ui.shape) * 0.02, 1e-3, 1 - 1e-3)
    far, (dist, refdist) = pop.diagnose_move_distances(region, ui, ufinal)
    out["diag_ufinal"], out["diag_far"], out["diag_dist"], out["diag_ref"] = ufinal, far, dist, np.float64(refdist)
    # PopulationSliceSampler traces: fixed live points, rising threshold
    Ls = inputs.walker_loglike(u)
    for name, popsize, nsteps in [("generate_cube_oriented_direction", 13, 7),
                                  ("generate_mixture_random_direction", 40, 5),
                                  ("generate_region_random_direction", 1, 4)]:
        np.random.seed(950)
        sampler = pop.PopulationSliceSampler(popsize=popsize, nsteps=nsteps, generate_direction=getattr(sf, name),
                                             scale=0.8)
        order = np.argsort(Ls)
        rows = []
        nfound = 0
        for it in range(400):
            Lmin = Ls[order[min(nfound // 3, len(order) - 50)]]
            unew, pnew, Lnew, nc = sampler.__next__(region, Lmin, u, Ls, inputs.walker_transform,
                                                     inputs.walker_loglike)
            if unew is None:
                rows.append(np.concatenate([[0, nc, np.nan], np.full(2 * u.shape[1], np.nan)]))
            else:
                nfound += 1
                rows.append(np.concatenate([[1, nc, Lnew], unew, pnew]))
        k = "trace_%s_" % name
        out[k + "rows"] = np.array(rows)
        out[k + "scale"] = np.float64(sampler.scale)
        out[k + "generation"] = sampler.generation
        out[k + "allL"] = sampler.allL
        out[k + "next_random"] = np.float64(np.random.uniform())
        print("  trace", name, "found", nfound, "of 400 calls")
    save("g9_stepfuncs", **out)


# ------------------------------------------------------------------ G10 -----

def explore(netiter, seed, nroots, nnodes, nbootstraps, random):
    """Run the driver's explorer / counter loop (integrator.py:2650-2832) over a seeded tree."""
    values, children = inputs.random_tree(seed, nroots, nnodes)
    roots = inputs.build_nodes(netiter.TreeNode, values, children, nroots)
    np.random.seed(seed)
    explorer = netiter.BreadthFirstIterator(roots)
    counter = netiter.MultiCounter(nroots=nroots, nbootstraps=nbootstraps, random=random, check_insertion_order=True)
    trace = []
    while True:
        nxt = explorer.next_node()
        if nxt is None:
            break
        rootid, node, (_, active_rootids, active_values, active_node_ids) = nxt
        counter.passing_node(rootid, node, active_rootids, active_values)
        trace.append([node.id, rootid, len(active_values), counter.logZ, counter.logZerr, counter.logVolremaining,
                      counter.logZremain, counter.logZremainMax, counter.remainder_ratio, counter.remainder_fraction,
                      len(counter.insertion_order_accumulator), counter.insertion_order_accumulator.zscore])
        explorer.expand_children_of(rootid, node)
    return dict(trace=np.array(trace, dtype=float), logweights=np.array(counter.logweights),
                istail=np.array(counter.istail), all_logZ=np.array(counter.all_logZ), all_H=np.array(counter.all_H),
                all_logVolremaining=np.array(counter.all_logVolremaining), rootids=np.array(counter.rootids),
                runs=np.array(counter.insertion_order_runs, dtype=np.int64), logZ_bs=np.float64(counter.logZ_bs),
                logZerr_bs=np.float64(counter.logZerr_bs), next_random=np.float64(np.random.uniform()))


G10_CASES = [(1001, 40, 900, 10, False), (1002, 400, 4000, 30, False), (1003, 25, 600, 5, True), (1004, 1, 50, 3, False)]


def g10(ref):
    """Integrator bookkeeping: BreadthFirstIterator + MultiCounter traces."""
    import ultranest.netiter as netiter
    out = {}
    for seed, nroots, nnodes, nboot, random in G10_CASES:
        with np.errstate(all="ignore"):
            res = explore(netiter, seed, nroots, nnodes, nboot, random)
        if len(res["trace"]) > 1000:       # keep the fixture small: every 16th iteration of the big case
            res["logweights"] = res["logweights"][::16]
            res["trace"] = res["trace"][::16]
        for k, v in res.items():
            out["c%d_%s" % (seed, k)] = v
        print("  tree", seed, "iterations", len(res["trace"]), "logZ", res["trace"][-1][3], "runs", res["runs"])
    save("g10_netiter", **out)


# ------------------------------------------------------------------ G11 -----

def g11(ref):
    """Point-store file written by the reference's TextPointStore, and its pop order."""
    import importlib
    import ultranest.store as store
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    rows = importlib.import_module("test_store")._rows()
    path = os.path.join(HERE, "g11_pointstore.tsv")
    if os.path.exists(path):
        os.remove(path)
    s = store.TextPointStore(path, 8)
    for i, row in enumerate(rows):
        s.add(row, 10 + i)
    s.close()
    s = store.TextPointStore(path, 8)
    pops = []
    for Lmin in [-np.inf, -np.inf, -1.0, -0.5, -1.0, 0.0, 0.3, 0.5, 0.5, 2.0, -np.inf, 1e-300, 5.0] * 4:
        idx, row = s.pop(Lmin)
        pops.append([Lmin, -1 if idx is None else idx])
    s.close()
    np.savetxt(os.path.join(HERE, "g11_pointstore_pops.txt"), np.array(pops))
    print("  wrote g11_pointstore.tsv", os.path.getsize(path), "bytes;", len(pops), "pops")


# ------------------------------------------------------------------ G12 -----

def run_and_write(netiter, update_results, make_run_dir, seed, nroots, nnodes, nboot, log_dir):
    """The tail of a run as the reference's driver performs it (integrator.py:2650-2832, :2933-2995):
    explore the tree with the run's counter, then `_update_results` (combine_results + logz_sequence +
    result files).  `update_results(state, counter, saved_logl, saved_nodeids)` is the driver method."""
    import types
    values, children = inputs.random_tree(seed, nroots, nnodes)
    us, ps = inputs.tree_points(seed, len(values))
    pile = netiter.PointPile(us.shape[1], ps.shape[1])
    for urow, prow in zip(us, ps):
        pile.add(urow, prow)
    roots = inputs.build_nodes(netiter.TreeNode, values, children, nroots)
    root = netiter.TreeNode(id=-1, value=-np.inf, children=roots)
    np.random.seed(seed)
    explorer = netiter.BreadthFirstIterator(roots)
    counter = netiter.MultiCounter(nroots=nroots, nbootstraps=nboot, random=False, check_insertion_order=False)
    saved_logl, saved_nodeids = [], []
    while True:
        nxt = explorer.next_node()
        if nxt is None:
            break
        rootid, node, (_, active_rootids, active_values, _) = nxt
        saved_logl.append(node.value)
        saved_nodeids.append(node.id)
        counter.passing_node(rootid, node, active_rootids, active_values)
        explorer.expand_children_of(rootid, node)
    names, derived = inputs.RESULT_PARAMNAMES
    state = types.SimpleNamespace(
        log=False, logger=None, pointpile=pile, use_mpi=False, comm=None, ncall=7 * len(values), paramnames=list(names),
        derivedparamnames=list(derived), min_num_live_points=nroots, root=root, log_to_disk=True,
        logs=make_run_dir(log_dir, run_num=1), num_params=len(names) + len(derived))
    update_results(state, counter, saved_logl, saved_nodeids)
    return state, np.float64(np.random.uniform())


def g12(ref):
    """Run summaries and result files: the reference's `_update_results` on seeded
    trees; inputs of the file writer, sha256 of every file it wrote, results.json verbatim."""
    import hashlib
    import shutil
    import ultranest.netiter as netiter
    from ultranest.integrator import ReactiveNestedSampler
    from ultranest.utils import make_run_dir, resample_equal
    out = {}
    for seed, nroots, nnodes, nboot in inputs.RESULT_TREES:
        tmp = tempfile.mkdtemp()
        with np.errstate(all="ignore"):
            state, next_random = run_and_write(netiter, ReactiveNestedSampler._update_results, make_run_dir, seed, nroots,
                                               nnodes, nboot, tmp)
        res, seq = state.results, state.run_sequence
        k = "t%d_" % seed
        out[k + "next_random"] = next_random
        for name in ("upoints", "points", "weights", "logw", "bootstrapped_weights", "logl"):
            out[k + "ws_" + name] = np.asarray(res["weighted_samples"][name])
        out[k + "samples"] = np.asarray(res["samples"])
        for name in ("logz", "logzerr", "logvol", "nlive", "insert_order", "logwt", "logl", "weights"):
            out[k + "seq_" + name] = np.asarray(seq[name])
        out[k + "results_json"] = np.array(open(os.path.join(state.logs["info"], "results.json")).read())
        for sub, fn in (("chains", "equal_weighted_post.txt"), ("chains", "weighted_post.txt"),
                        ("chains", "weighted_post_untransformed.txt"), ("chains", "run.txt"),
                        ("info", "results.json"), ("info", "post_summary.csv")):
            data = open(os.path.join(state.logs[sub], fn), "rb").read()
            out[k + "sha256_" + fn] = np.array(hashlib.sha256(data).hexdigest())
            out[k + "head_" + fn] = np.array(data[:400].decode())
        vals = np.sort([n.value for n in state.root.children])
        lo, hi = float(vals[len(vals) // 2]), float(vals[-1] + 1.0)
        out[k + "count_tree"] = np.array(netiter.count_tree(state.root.children))
        out[k + "count_between"] = np.array([lo, hi] + list(netiter.count_tree_between(state.root.children, lo, hi)))
        for tag, thresh in (("mid", lo), ("high", hi), ("first", float(vals[0]))):
            parents, weights = netiter.find_nodes_before(state.root, thresh)
            out[k + "before_%s" % tag] = np.array([[thresh, n.id, w] for n, w in zip(parents, weights)], dtype=float).reshape(-1, 3)
        print("  tree", seed, "niter", res["niter"], "logz", res["logz"], "+-", res["logzerr"], "ess", res["ess"])
        shutil.rmtree(tmp)
    # systematic resampling alone, incl. a weight vector whose cumulative sum stays below the last position
    rs = np.random.RandomState(12)
    w = rs.uniform(size=500) ** 6
    w /= w.sum()
    x = rs.normal(size=(500, 2))
    out["resample_x"], out["resample_w"] = x, w
    out["resample_out"] = resample_equal(x, w, rstate=np.random.RandomState(99))
    save("g12_results", **out)


# ------------------------------------------------------------------ G13 -----

OVERCLUSTERED_TXT = [20, 23, 24, 27, 49]
OVERCLUSTERED_NPZ = [20, 23, 24, 27, 42]


def g13(ref):
    """The reference's over-clustering regression fixtures (tests/overclustered_u_*.txt, tests/overclustered_*.npz)
    run through the reference's own recipes (tests/test_clustering.py:116-225), with every intermediate value
    recorded: the range checks of those tests become exact known answers."""
    import types
    from ultranest import ReactiveNestedSampler
    from ultranest.utils import create_logger
    tests_dir = os.path.join(REF, "tests")
    out = {}
    # recipe 1 (test_overclustering_eggbox_txt): ScalingLayer + MLFriends.compute_maxradiussq (global np.random) +
    # update_clusters called with the u-space points as transformed points, as the reference's test does
    np.random.seed(1)
    for i in OVERCLUSTERED_TXT:
        points = np.loadtxt(os.path.join(tests_dir, "overclustered_u_%d.txt" % i))
        out["txt%d_u" % i] = points
        radii, counts = [], []
        for k in range(3):
            layer = ref.ScalingLayer(wrapped_dims=[])
            layer.optimize(points, points)
            region = ref.MLFriends(points, layer)
            maxr = region.compute_maxradiussq(nbootstraps=30)
            region.maxradiussq = maxr
            nclusters, clusteridxs, overlapped = ref.update_clusters(points, points, maxr)
            radii.append(maxr)
            counts.append(nclusters)
        for j in range(3):
            nclusters, clusteridxs, overlapped = ref.update_clusters(points, points, maxr)
            counts.append(nclusters)
        assert 14 < nclusters < 20
        out["txt%d_radii" % i] = np.array(radii)
        out["txt%d_nclusters" % i] = np.array(counts)
        out["txt%d_ids" % i] = np.asarray(clusteridxs)
        print("  txt", i, len(points), "points; r2", radii, "nclusters", counts)

    # recipe 2 (test_overclustering_eggbox_update): the driver's _update_region on u0, then on u after invalidating
    # the radius; AffineLayer, 30 bootstraps
    class MockIntegrator(ReactiveNestedSampler):
        def __init__(self):
            self.use_mpi = False
            self.mpi_size = 1
            self.mpi_rank = 0
            self.region = None
            self.transformLayer = None
            self.wrapped_axes = []
            self.log = False
            self.logger = create_logger("mock")
            self.region_class = ref.MLFriends
            self.transform_layer_class = ref.AffineLayer

    def state(mock):
        return np.array([mock.region.maxradiussq, mock.region.enlarge, mock.transformLayer.nclusters], dtype=float)

    np.random.seed(1)
    for i in OVERCLUSTERED_NPZ:
        data = np.load(os.path.join(tests_dir, "overclustered_%d.npz" % i))
        u0, u1 = data["u0"], data["u"]
        out["npz%d_u0" % i], out["npz%d_u" % i] = u0, u1
        mock = MockIntegrator()
        mock.x_dim = u0.shape[1]
        mock._update_region(u0, u0)
        out["npz%d_first" % i] = state(mock)
        out["npz%d_first_ids" % i] = np.asarray(mock.transformLayer.clusterids)
        same = mock.transformLayer.create_new(u0, mock.region.maxradiussq)
        out["npz%d_same_ids" % i] = np.asarray(same.clusterids)
        new = mock.transformLayer.create_new(u1, mock.region.maxradiussq)
        out["npz%d_new_ids" % i] = np.asarray(new.clusterids)
        mock.region.maxradiussq = None
        updated = mock._update_region(u1, u1)
        out["npz%d_second" % i] = state(mock)
        out["npz%d_second_ids" % i] = np.asarray(mock.transformLayer.clusterids)
        out["npz%d_updated" % i] = np.array(bool(updated))
        out["npz%d_next_random" % i] = np.float64(np.random.uniform())
        _, sizes = np.unique(mock.transformLayer.clusterids, return_counts=True)
        assert 14 < mock.transformLayer.nclusters < 20 and sizes.min() > 1
        print("  npz", i, u0.shape, "->", u1.shape, "first", out["npz%d_first" % i], "second", out["npz%d_second" % i], updated)
    save("g13_overclustered", **out)


# ------------------------------------------------------------------ G14 -----

def g14(ref):
    """`dump_tree` (reference netiter.py:220-256): the arrays and dataset options the reference
    hands to h5py for the seeded trees of g12."""
    import ultranest.netiter as netiter
    out = {}
    for seed, nroots, nnodes, _ in inputs.RESULT_TREES:
        w = inputs.recorded_tree_dump(netiter, seed, nroots, nnodes)
        k = "t%d_" % seed
        out[k + "mode"] = np.array(w["mode"])
        out[k + "names"] = np.array(list(w["datasets"]))
        for name, (data, options) in w["datasets"].items():
            out[k + name] = data
            out[k + name + "_options"] = np.array(json.dumps(options, sort_keys=True))
        print("  tree", seed, {n: d.shape for n, (d, _) in w["datasets"].items()})
    save("g14_tree_dump", **out)


def g16(ref):
    """PopulationRandomWalkSampler / PopulationSimpleSliceSampler (reference popstepsampler.py:192-358, 746-1001) on a fixed
    region: 40 calls each from one numpy seed -- points, likelihoods, evaluation counts, final scale, log table, and where
    the numpy stream stands afterwards."""
    import ultranest.popstepsampler as pop
    out = {}
    for name, cls, direction, kw, popsize, nsteps, d in inputs.BATCHED_SAMPLER_CASES:
        u, region, loglike, Ls, Lmin = inputs.batched_sampler_problem(ref, d)
        kw = dict(kw)
        if kw.get("slice_limit") == "scale":
            kw["slice_limit"] = pop.slice_limit_to_scale
        sampler = getattr(pop, cls)(popsize=popsize, nsteps=nsteps, generate_direction=getattr(pop, direction), **kw)
        np.random.seed(11)
        res = [sampler.__next__(region, Lmin, u, Ls, lambda x: x * 1.0, loglike) for _ in range(40)]
        out[name + "_u"] = np.array([r[0] for r in res])
        out[name + "_p"] = np.array([r[1] for r in res])
        out[name + "_L"] = np.array([r[2] for r in res])
        out[name + "_nc"] = np.array([r[3] for r in res], dtype=np.int64)
        out[name + "_scale"] = np.float64(sampler.scale)
        out[name + "_logstat"] = np.array(sampler.logstat, dtype=float)
        out[name + "_next_random"] = np.float64(np.random.uniform())
    save("g16_batched_samplers", **out)


def timing(ref):
    """SURVEY 8(d) / BASELINE.md 3.2: how good a TIMING stand-in is the C restatement (oracle/) for the real Cython path?
    Both on this container's CPU, one core, the 8(d) inputs (N = 4000, d = 50, seed 1), a bounded query sample.  Stored in
    META.json as `timing`; bench.py quotes `cython_over_port` next to `cpu_baseline` (the port is what travels to the
    GPU box)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle as orc
    rs = np.random.RandomState(1)
    N, d, P = 4000, 50, 2000
    u = 0.5 + 0.05 * rs.normal(size=(N, d))
    layer = ref.AffineLayer()
    layer.optimize(u, u)
    region = ref.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=30, minvol=0., rng=np.random.RandomState(1))
    region.create_ellipsoid()
    z = rs.normal(size=(P, d))
    z /= np.linalg.norm(z, axis=1)[:, None]
    w = region.ellipsoid_center + (z * region.enlarge**0.5 * rs.uniform(size=(P, 1))**(1. / d)) @ region.ellipsoid_axes_T
    tq = np.ascontiguousarray(layer.transform(w))
    un = np.ascontiguousarray(region.unormed)

    def best(f, reps=3):
        t = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            t = min(t, time.perf_counter() - t0)
        return t
    out = {}
    idx = np.empty(P, dtype=int)
    for name, r2 in (("set_E_early_exit", region.maxradiussq), ("set_F_full_scan", 1e-300)):
        t_ref = best(lambda: ref.find_nearby(un, tq, r2, idx))
        want = idx.copy()
        t_port = best(lambda: orc.find_nearby(un, tq, r2))
        assert np.array_equal(orc.find_nearby(un, tq, r2), want)
        out["find_nearby_" + name] = dict(cython_s=t_ref, port_s=t_port, cython_over_port=t_ref / t_port,
                                          queries=P, cython_queries_per_s=P / t_ref, port_queries_per_s=P / t_port)
    # K4 is a `cdef` function (mlfriends.pyx:188): reached through MLFriends.compute_maxradiussq (:996-1016), which draws
    # the selection from np.random -- the port gets the same draws from a generator seeded alike
    def ref_rounds():
        np.random.seed(2)
        return region.compute_maxradiussq(nbootstraps=3)

    def port_rounds():
        rs2 = np.random.RandomState(2)
        m = 0.0
        for _ in range(3):
            mask = np.zeros(N, dtype=bool)
            mask[rs2.randint(N, size=N)] = True
            m = max(m, orc.maxradiussq(un[mask], un[~mask]))
        return m
    t_ref = best(ref_rounds, reps=2)
    t_port = best(port_rounds, reps=2)
    assert port_rounds() == ref_rounds()
    out["compute_maxradiussq_3_rounds"] = dict(cython_s=t_ref, port_s=t_port, cython_over_port=t_ref / t_port)
    t_ref = best(lambda: region.compute_enlargement(nbootstraps=30, minvol=0., rng=np.random.RandomState(3)), reps=1)
    out["compute_enlargement_30_rounds_cython_s"] = t_ref
    out["note"] = ("one core of the dev container; cython = the reference's mlfriends.pyx built -O3 (setup.py:21-25); "
                   "port = oracle/mlfriends_oracle.c (-O3 -ffp-contract=off); a ratio > 1 means the port is FASTER, "
                   "i.e. cpu_baseline.kind = 'port' overstates the reference's speed by that factor (conservative)")
    import platform
    out["cpu"] = platform.processor() or platform.machine()
    print(json.dumps(out, indent=1))
    return out


GROUPS = dict(g16=g16, g14=g14, g13=g13, g12=g12, g1=g1, g2=g2, g3=g3, g456=g456, g7=g7, g8=g8, g9=g9, g10=g10, g11=g11)

if __name__ == "__main__":
    want = sys.argv[1:] or (list(GROUPS) + ["timing"])
    ref = build_reference()
    meta = dict(numpy=np.__version__, reference="UltraNest 4.5.0 (/root/reference)",
                generated=time.strftime("%Y-%m-%d"))
    meta_path = os.path.join(HERE, "META.json")
    if os.path.exists(meta_path):        # keep what this invocation does not regenerate (the timing block)
        with open(meta_path) as f:
            old_meta = json.load(f)
        if "timing" in old_meta:
            meta["timing"] = old_meta["timing"]
    for g in want:
        print(g)
        if g == "timing":
            meta["timing"] = timing(ref)
        else:
            GROUPS[g](ref)
    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1)
