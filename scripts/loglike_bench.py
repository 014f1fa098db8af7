"""Likelihood kernels on device-resident batches (10^6 x d): time per launch (hipEvents on the launch stream, best of 10
after a warm-up), bytes moved, and the largest relative deviation from a binary64 torch evaluation of the same formula.
    python scripts/loglike_bench.py"""
import ctypes
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = torch.cuda.current_stream().cuda_stream
out = {}
for kind, name, d in ((0, "gauss", 50), (3, "rosenbrock", 50), (1, "eggbox", 50), (1, "eggbox", 10), (2, "eggbox2", 10),
                      (0, "gauss", 5), (0, "gauss", 51), (3, "rosenbrock", 128), (3, "rosenbrock", 2)):
    n = 1000000
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.rand(n, d, dtype=torch.float64, device=dev, generator=g)
    if kind == 3:
        x = x * 4 - 2
    if kind in (1, 2):
        x = x * 10 * math.pi
    aux = torch.full((d,), 0.5, dtype=torch.float64, device=dev)
    like = torch.empty(n, dtype=torch.float64, device=dev)
    best = 1e9
    for it in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.mlf_loglike_dev(kind, ctypes.c_void_p(x.data_ptr()), d, n, ctypes.c_void_p(aux.data_ptr()), 0.1,
                                       ctypes.c_void_p(like.data_ptr()), ctypes.c_void_p(stream)))
        e1.record()
        torch.cuda.synchronize()
        if it >= 5:
            best = min(best, e0.elapsed_time(e1))
    if kind == 0:
        ref = -0.5 * (((x - aux) / 0.1) ** 2).sum(dim=1) - 0.5 * math.log(2 * math.pi * 0.01) * d
    elif kind == 1:
        ref = (2 + torch.cos(x / 2).prod(dim=1)) ** 5
    elif kind == 2:
        ref = torch.cos(x).prod(dim=1) ** 2
    else:
        a, b = x[:, :-1], x[:, 1:]
        ref = -2 * (100 * (b - a * a) ** 2 + (1 - a) ** 2).sum(dim=1)
    err = float(((like - ref).abs() / ref.abs().clamp_min(1e-300)).max().item())
    out["%s d=%d" % (name, d)] = {"ms": round(best, 4), "GBps": round(n * (8 * d + 8) / (best * 1e-3) / 1e9, 1),
                                  "frac_of_8TBps": round(n * (8 * d + 8) / (best * 1e-3) / 8e12, 3), "max_rel_dev_vs_torch": err}
print(json.dumps(out, indent=1))
