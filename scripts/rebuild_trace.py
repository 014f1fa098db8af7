#!/usr/bin/env python3
"""A few steady-state rebuilds (C5) and nothing else: the command for `rocprofv3 --kernel-trace --memory-copy-trace` (timeline of
one rebuild: kernels, copies and the gaps between them).  scripts/rebuild_timeline.py prints the merged timeline."""
import os
import sys

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
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    u2 = u.copy()
    u2[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
    upd.update(u2, nbootstraps=bench.NBOOT, minvol=0.)
