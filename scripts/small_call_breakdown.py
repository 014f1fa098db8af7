"""Where the time of a tiny MLFriends.inside call goes: the C-ABI call alone (DeviceRegion.inside) against the
reference-API call (MLFriends.inside with its device-state check), small path on and off."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ultranest_amd.mlfriends as M  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

for label, (n, d) in [("C1", (400, 5)), ("C5", (4000, 50))]:
    rs = np.random.RandomState(1)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=rs)
    region.create_ellipsoid()
    for p in (1, 10, 128):
        pts = u[rs.randint(n, size=p)] + 0.01 * rs.normal(size=(p, d))
        row = dict(config=label, batch=p)
        for small in (1, 0):
            _lib.set_option("small_path", small)
            region.inside(pts)
            handle = region._dev.sync(region, True)
            for name, fn in (("abi", lambda: handle.inside(pts)), ("api", lambda: region.inside(pts))):
                fn()
                t0 = time.perf_counter()
                for _ in range(300):
                    fn()
                row["%s_us_small%d" % (name, small)] = (time.perf_counter() - t0) / 300 * 1e6
        _lib.set_option("small_path", 1)
        print(json.dumps(row), flush=True)
