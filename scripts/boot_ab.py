"""K4 A/B: mlf_maxradiussq_bootstrap (30 rounds over 4000 x 50 live points resident in HBM) with k_boot (both orders of every
pair) and k_boot_sym (every pair once).  Under rocprofv3 --kernel-trace --stats the two kernels show up by name.
    python scripts/boot_ab.py [n d B reps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ultranest_amd import _lib, kernels  # noqa: E402

n, d, B, reps = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (4000, 50, 30, 50)
rs = np.random.RandomState(5)
u = rs.uniform(size=(n, d))
masks = np.zeros((B, n), dtype=np.uint8)
for b in range(B):
    masks[b, rs.randint(n, size=n)] = 1
U = kernels.DevArray.from_host(u)
Mk = kernels.DevArray.from_host(masks)
L = _lib.lib()
out = {}
res = {}
for mode in (0, 1, 0, 1):
    _lib.set_option("boot_symmetric", mode)
    r2 = np.empty(B)
    sk = np.empty(B, dtype=np.uint8)
    for _ in range(5):
        _lib.check(L.mlf_maxradiussq_bootstrap(U.data_ptr(), n, d, Mk.data_ptr(), B, _lib.ptr(r2), _lib.ptr(sk)))
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(L.mlf_maxradiussq_bootstrap(U.data_ptr(), n, d, Mk.data_ptr(), B, _lib.ptr(r2), _lib.ptr(sk)))
    dt = (time.perf_counter() - t0) / reps
    out.setdefault("call_us_boot_symmetric_%d" % mode, []).append(round(dt * 1e6, 2))
    res[mode] = r2.copy()
out["equal"] = bool(np.array_equal(res[0], res[1]))
out["config"] = dict(n=n, d=d, B=B, reps=reps)
print(json.dumps(out))
