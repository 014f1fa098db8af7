import time, numpy as np, sys
sys.path.insert(0, ".")
from ultranest_amd import kernels
rs = np.random.RandomState(1)
u = rs.uniform(size=(4000, 50))
sel = rs.randint(0, 2, size=(30, 4000)).astype(bool)
for pause in (0.0, 0.002, 0.02, 0.0):
    kernels.bootstrap_moments(u, sel)
    ts = []
    for i in range(30):
        if pause: time.sleep(pause)
        t0 = time.perf_counter(); kernels.bootstrap_moments(u, sel); ts.append(time.perf_counter() - t0)
    print("pause %.3f s: median call %.3f ms  min %.3f" % (pause, np.median(ts) * 1e3, min(ts) * 1e3))
