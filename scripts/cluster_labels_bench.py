"""mlf_cluster_labels and the pieces around it in the device-resident rebuild (C5 sizes)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

u, region = bench.build_region(None)
dev = torch.device("cuda", 0)
L = _lib.lib()
n, d = u.shape
layer = region.transformLayer
U = torch.from_numpy(u).to(dev)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def consts():
    c = torch.from_numpy(np.ascontiguousarray(layer.ctr, dtype=np.float64)).to(dev)
    T = torch.from_numpy(np.ascontiguousarray(layer.T, dtype=np.float64)).to(dev)
    return c, T


c_d, T_d = consts()
t_old = torch.mm(U - c_d, T_d)
labels = np.empty(n, dtype=np.int64)
ncl = ctypes.c_int64(0)
prev = np.ones(n, dtype=np.int64)
out = {
    "small uploads (ctr, T)": timed(consts),
    "whitening (sub + mm)": timed(lambda: torch.mm(U - c_d, T_d)),
    "mlf_cluster_labels (device in)": timed(lambda: _lib.check(L.mlf_cluster_labels(ctypes.c_void_p(t_old.data_ptr()), n, d, float(region.maxradiussq), _lib.ptr(prev), _lib.ptr(labels), ctypes.byref(ncl)))),
    "nclusters": int(ncl.value),
}
cen = torch.empty_like(U)
out["mlf_subtract_nearby (device in/out)"] = timed(lambda: _lib.check(L.mlf_subtract_nearby(ctypes.c_void_p(U.data_ptr()), n, d, float(region.maxradiussq), ctypes.c_void_p(cen.data_ptr()))))
print(json.dumps(out, indent=1))
