#!/usr/bin/env python3
"""cProfile of the steady-state region rebuild bench.py times (C5: N = 4000, d = 50, 30 bootstrap rounds)."""
import cProfile
import io
import os
import pstats
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
for rep in range(3):
    u2 = u.copy()
    u2[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
    upd.update(u2, nbootstraps=bench.NBOOT, minvol=0.)
u2 = u.copy()
u2[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
upd.update(u2, nbootstraps=bench.NBOOT, minvol=0.)
pr.disable()
print("rebuild ms:", (time.perf_counter() - t0) * 1e3)
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumtime").print_stats(45)
print("\n".join(l[:160] for l in out.getvalue().splitlines()[:70]))
