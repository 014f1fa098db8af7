"""The reference's OWN benchmark grid (tests/benchmark_maxradius.py) through ultranest_amd.mlfriends on the GPU, with the
CPU port (oracle/, test infrastructure) timed next to it on the host of the same box.

  benchmark_maxradius  (:6-33)  `MLFriends(points, ScalingLayer()).compute_maxradiussq(nbootstraps=20)` for
                                ndim in 2 .. 64 x npts in 100, 400, 1000, 4000, uniform points, np.random.seed(ndim)
  benchmark_transform  (:36-98) npts = 400, ndim in 2 .. 256, ScalingLayer and AffineLayer: untransform + transform of ONE
                                point, and region.inside of 10 points

Same loops and the same stopping rule as the reference (repeat until 1 s / 0.1 s of accumulated time), JSON instead of the
plots.  d = 256 runs on the run-time-dimensionality kernels of mlf_wide.hip (round 5; until round 4: MLF_E_DIM above 128).

    python scripts/reference_grid_bench.py [--no-cpu] > profiles/r04_reference_grid.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultranest_amd.mlfriends import AffineLayer, MLFriends, ScalingLayer  # noqa: E402

CPU = "--no-cpu" not in sys.argv


def until(budget, body):
    niter, total = 0, 0.0
    out = None
    while total < budget:
        start = time.time()
        out = body()
        total += time.time() - start
        niter += 1
    return total * 1000 / niter, niter, out


def port_maxradius(points, layer, nbootstraps, seed):
    """what MLFriends.compute_maxradiussq does (mlfriends.pyx:996-1016) on the C port: the same draws, one round at a time"""
    from oracle import oracle as orc
    unormed = layer.transform(points)
    N = len(points)
    rs = np.random.RandomState(seed)
    t0 = time.time()
    maxd = 0.0
    for _ in range(nbootstraps):
        sel = np.zeros(N, dtype=bool)
        sel[rs.randint(N, size=N)] = True
        maxd = max(maxd, orc.maxradiussq(unormed[sel], unormed[~sel]))
    return (time.time() - t0) * 1000, maxd


def benchmark_maxradius():
    rows = []
    for ndim in 2, 4, 8, 16, 32, 64:
        np.random.seed(ndim)
        for npts in 100, 400, 1000, 4000:
            points = np.random.uniform(size=(npts, ndim))
            layer = ScalingLayer()
            region = MLFriends(points, layer)
            region.compute_maxradiussq(nbootstraps=20)       # first call of a size class: code objects, buffers
            ms, niter, maxr = until(1.0, lambda: region.compute_maxradiussq(nbootstraps=20))
            row = dict(ndim=ndim, npts=npts, gpu_ms=ms, calls=niter, value=float(maxr))
            if CPU:
                # the same selection draws for both: the value must agree bit for bit
                np.random.seed(1000 + ndim)
                want = region.compute_maxradiussq(nbootstraps=20)
                cms, got = port_maxradius(points, layer, 20, 1000 + ndim)
                row.update(cpu_port_ms=cms, cpu_over_gpu=cms / ms, same_value_as_port=bool(got == want))
            rows.append(row)
            print("maxradius", row, file=sys.stderr)
    return rows


def benchmark_transform():
    rows = []
    npts = 400
    for lname in "scale", "affine":
        for ndim in 2, 4, 8, 16, 32, 64, 128, 256:
            np.random.seed(ndim)
            points = np.random.uniform(0.4, 0.6, size=(npts, ndim))
            layer = ScalingLayer() if lname == "scale" else AffineLayer()
            row = dict(layer=lname, ndim=ndim, npts=npts)
            try:
                region = MLFriends(points, layer)
                region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=30)
                region.create_ellipsoid()
                region.inside(np.random.normal(0.5, 0.1, size=(10, ndim)))
            except Exception as e:       # d > 128: MLF_E_DIM
                row["unsupported"] = "%s: %s" % (type(e).__name__, str(e)[:120])
                rows.append(row)
                print("transform", row, file=sys.stderr)
                continue

            def tr():
                u = region.transformLayer.untransform(np.random.normal(size=(ndim)))
                region.transformLayer.transform(u)
            row["transform_ms"], _, _ = until(0.1, tr)
            niter, total = 0, 0.0
            while total < 0.1:
                u = np.random.normal(0.5, 0.1, size=(10, ndim))
                start = time.time()
                region.inside(u)
                total += time.time() - start
                niter += 1
            row["inside10_ms"] = total * 1000 / niter
            if CPU:
                from oracle import oracle as orc
                niter, total = 0, 0.0
                ctr = np.ascontiguousarray(np.broadcast_to(layer.ctr if lname == "affine" else layer.mean, (ndim,)), dtype=float)

                # the library whitens the live points with the same k-ascending FMA chain as the proposals (np.dot's
                # order is BLAS's own business); the port gets the live points whitened that way too
                # (the benchmark's AffineLayer() is never optimised: ctr = 0, T = 1 -- the identity, spelled out for the C port)
                Tm = np.ascontiguousarray(layer.T if np.ndim(layer.T) == 2 else np.eye(ndim) * float(layer.T)) if lname == "affine" else None
                live_t = orc.affine_transform(np.asarray(region.u), ctr, Tm) if lname == "affine" else region.unormed

                def port_inside(u):      # R3 = H3 -> T1 -> K1 (mlfriends.pyx:1186-1211) on the C port
                    mask = orc.inside_ellipsoid(u, region.ellipsoid_center, region.ellipsoid_invcov, region.enlarge)
                    if mask.any():
                        w = u[mask]
                        t = orc.affine_transform(w, ctr, Tm) if lname == "affine" else (w - ctr) / np.asarray(layer.std).reshape((1, -1))
                        mask[mask] = orc.find_nearby(live_t, t, region.maxradiussq) >= 0
                    return mask
                while total < 0.1:
                    u = np.random.normal(0.5, 0.1, size=(10, ndim))
                    start = time.time()
                    got = port_inside(u)
                    total += time.time() - start
                    niter += 1
                row["inside10_cpu_port_ms"] = total * 1000 / niter
                row["last_mask_equal"] = bool(np.array_equal(got, region.inside(u)))
            rows.append(row)
            print("transform", row, file=sys.stderr)
    return rows


if __name__ == "__main__":
    from ultranest_amd import _lib
    out = dict(device=_lib.device_name(), host_cores=os.cpu_count(),
               reference="tests/benchmark_maxradius.py:6-33 (compute_maxradiussq grid), :36-98 (transform / inside of 10 points)",
               maxradius=benchmark_maxradius(), transform=benchmark_transform())
    print(json.dumps(out, indent=1))
