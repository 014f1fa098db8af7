#!/usr/bin/env python3
"""Wall time of 40 consecutive steady-state rebuilds (C5) with the cyclic garbage collector's full passes logged.
`python scripts/rebuild_rounds.py freeze` calls gc.freeze() first: the one 37 ms outlier (a generation-2 pass over
the 170000 objects the imports leave behind) disappears."""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ultranest_amd.mlfriends as M  # noqa: E402
from ultranest_amd.harness import RegionUpdater  # noqa: E402

u, region = bench.build_region(None)
rs = np.random.RandomState(7)
upd = RegionUpdater(bench.NDIM, region_class=M.MLFriends, transform_layer_class=M.LocalAffineLayer, freeze_gc=True)
np.random.seed(11)
upd.update(u, nbootstraps=bench.NBOOT, minvol=0.)

passes = []
started = [0.0]


def on_gc(phase, info):
    if phase == "start":
        started[0] = time.perf_counter()
    elif info["generation"] >= 1:
        passes.append((info["generation"], round((time.perf_counter() - started[0]) * 1e3, 2), info["collected"]))


gc.callbacks.append(on_gc)
print("tracked objects:", len(gc.get_objects()), "thresholds", gc.get_threshold())
if len(sys.argv) > 1 and sys.argv[1] == "freeze":
    gc.collect()
    gc.freeze()
    print("frozen:", gc.get_freeze_count())
times = []
for rep in range(40):
    v = u.copy()
    v[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
    passes.clear()
    t0 = time.perf_counter()
    upd.update(v, nbootstraps=bench.NBOOT, minvol=0.)
    times.append((time.perf_counter() - t0) * 1e3)
    print("round %2d: %6.2f ms; collector passes (generation, ms, collected): %s" % (rep, times[-1], passes))
print("median %.2f ms, max %.2f ms" % (np.median(times), max(times)))
