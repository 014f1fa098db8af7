"""Device-side region.sample() vs the host-draw path (row f2), C5 shape (N=4000, d=50)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from golden import inputs  # noqa: E402

import ultranest_amd.mlfriends as m  # noqa: E402
from ultranest_amd.regions import DeviceRNG  # noqa: E402

out = {}
for (n, d, nsamples) in [(400, 2, 1 << 20), (4000, 50, 1 << 20)]:
    u = inputs.live_points(5, n, d)
    layer = m.AffineLayer()
    layer.optimize(u, u)
    region = m.MLFriends(u, layer)
    rng = np.random.RandomState(1)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rng)
    region.create_ellipsoid()
    res = {}
    for name, fn in [("boundingbox", region.sample_from_boundingbox), ("wrapping_ellipsoid", region.sample_from_wrapping_ellipsoid)]:
        region.device_rng = None
        np.random.seed(1)
        fn(nsamples)
        t0 = time.perf_counter()
        host = fn(nsamples)
        t_host = time.perf_counter() - t0
        region.device_rng = DeviceRNG(1)
        fn(nsamples)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dev = fn(nsamples)
        t_dev = (time.perf_counter() - t0) / reps
        res[name] = dict(host_ms=t_host * 1e3, device_ms=t_dev * 1e3, host_accept=len(host) / nsamples,
                         device_accept=len(dev) / nsamples)
    out["N%d_d%d" % (n, d)] = res
print(json.dumps(out, indent=1))
