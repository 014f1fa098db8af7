"""Latency of MLFriends.inside through the reference's own API (host numpy in, host mask out) at the batch sizes
the stock driver and the scalar step samplers use (reference stepsampler.py:296-330, 1060-1071, 1124: 1-10 points
per call; integrator.py:1854-1855: all N live points every iteration).  Writes profiles/r02_small_batch.json
when called with --save (run on the MI355X box)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ultranest_amd.mlfriends as M  # noqa: E402

out = []
for label, (n, d) in [("C1", (400, 5)), ("C2", (2000, 20)), ("C5", (4000, 50))]:
    rs = np.random.RandomState(1)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rs)
    region.create_ellipsoid()
    for p in (1, 10, 128, n, 65536):
        pts = u[rs.randint(n, size=p)] + 0.01 * rs.normal(size=(p, d))
        region.inside(pts)
        reps = 200 if p <= 4096 else 30
        each = []
        for _ in range(reps):
            t0 = time.perf_counter()
            m = region.inside(pts)
            each.append(time.perf_counter() - t0)
        dt = float(np.mean(each))
        med = float(np.median(each))
        # the driver's pattern: one live point replaced in place between calls (integrator.py:2749-2765); medians,
        # write included (a generation-2 pass of the cyclic collector inside the loop costs 37 ms once)
        each = []
        for i in range(reps):
            t0 = time.perf_counter()
            region.u[i % n] = np.clip(region.u[i % n] + 1e-9, 1e-6, 1 - 1e-6)
            region.unormed[i % n] = region.transformLayer.transform(region.u[i % n])
            m = region.inside(pts)
            each.append(time.perf_counter() - t0)
        dt_upd = float(np.median(each))
        # a refill's worth of iterations between two calls: 20 rows replaced, then one call
        each = []
        for i in range(reps // 4):
            t0 = time.perf_counter()
            for k in range(20):
                j = (i * 20 + k) % n
                region.u[j] = np.clip(region.u[j] + 1e-9, 1e-6, 1 - 1e-6)
            m = region.inside(pts)
            each.append(time.perf_counter() - t0)
        dt_upd20 = float(np.median(each))
        out.append(dict(config=label, n_live=n, d=d, batch=p, us_per_call=dt * 1e6, us_per_call_median=med * 1e6,
                        us_per_call_with_one_row_replaced=dt_upd * 1e6,
                        us_per_call_with_20_rows_replaced=dt_upd20 * 1e6, accept=float(m.mean())))
        print(json.dumps(out[-1]), flush=True)
if "--save" in sys.argv:
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/small_batch.json", "w"), indent=1)
