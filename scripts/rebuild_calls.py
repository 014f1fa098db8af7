#!/usr/bin/env python3
"""Wall time of every C-ABI call and of the host numpy between them during one steady-state rebuild (C5)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ultranest_amd.mlfriends as M  # noqa: E402
from ultranest_amd import _lib  # noqa: E402
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
real = _lib.lib()
log = []


class Timed:
    def __getattr__(self, name):
        fn = getattr(real, name)

        def call(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            log.append((name, t0, time.perf_counter()))
            return r
        return call


_lib._lib = Timed()
u2 = u.copy()
u2[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
t_start = time.perf_counter()
upd.update(u2, nbootstraps=bench.NBOOT, minvol=0.)
t_end = time.perf_counter()
prev = t_start
for name, a, b in log:
    print("%8.3f ms host | %8.3f ms %s" % ((a - prev) * 1e3, (b - a) * 1e3, name))
    prev = b
print("%8.3f ms host (tail)" % ((t_end - prev) * 1e3))
print("total %.3f ms, in C calls %.3f ms" % ((t_end - t_start) * 1e3, sum(b - a for _, a, b in log) * 1e3))

# the same rebuild under cProfile, averaged over several rounds: which host functions the gaps consist of
import cProfile  # noqa: E402
import io  # noqa: E402
import pstats  # noqa: E402

_lib._lib = real
ROUNDS = 20
inputs = []
for rep in range(ROUNDS):
    v = u.copy()
    v[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
    inputs.append(v)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for v in inputs:
    upd.update(v, nbootstraps=bench.NBOOT, minvol=0.)
pr.disable()
print("under cProfile: %.3f ms per rebuild" % ((time.perf_counter() - t0) / ROUNDS * 1e3))
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:32]
for (fname, line, func), (cc, nc, tt, ct, _) in rows:
    print("%7.3f ms self %7.3f ms cum %5.1f calls  %s:%d %s" % (tt / ROUNDS * 1e3, ct / ROUNDS * 1e3, nc / ROUNDS,
                                                            os.path.basename(fname), line, func))
