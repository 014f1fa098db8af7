"""Latency of MLFriends.inside through the host-pointer API at the batch sizes the stock driver uses."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ultranest_amd.mlfriends as M  # noqa: E402

out = []
for (n, d) in [(400, 5), (400, 20), (4000, 50)]:
    rs = np.random.RandomState(1)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rs)
    region.create_ellipsoid()
    for p in (128, 4096, 65536):
        pts = rs.uniform(0.3, 0.7, size=(p, d))
        region.inside(pts)
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            m = region.inside(pts)
        dt = (time.perf_counter() - t0) / reps
        out.append(dict(n=n, d=d, p=p, us_per_call=dt * 1e6, accept=float(m.mean())))
        print(json.dumps(out[-1]), flush=True)
