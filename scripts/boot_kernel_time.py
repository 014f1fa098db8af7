"""The bootstrap-radius call at C5 (N = 4000, d = 50, 30 rounds): median wall time of K.maxradiussq_bootstrap (selection
upload, k_pack_selection, k_boot, k_boot_final, results back); run under rocprofv3 --kernel-trace --stats for k_boot alone."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ultranest_amd import kernels as K  # noqa: E402

u, region = bench.build_region(None)
rs = np.random.RandomState(3)
n = len(u)
masks = np.zeros((30, n), dtype=bool)
for b in range(30):
    masks[b, np.unique(rs.randint(n, size=n))] = True
t = region.unormed
for _ in range(30):
    K.maxradiussq_bootstrap(t, masks)
ts = []
for _ in range(40):
    t0 = time.perf_counter()
    r, sk = K.maxradiussq_bootstrap(t, masks)
    ts.append(time.perf_counter() - t0)
print(json.dumps({"median_call_us": round(float(np.median(ts)) * 1e6, 1), "r2max": float(r.max())}))
