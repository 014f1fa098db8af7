#!/usr/bin/env python3
"""Kernel-level micro-benchmark (GPU box): neighbour-scan throughput at the BASELINE sizes,
FP64 VALU rate probe, bootstrap kernel timing.  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultranest_amd import kernels as K  # noqa: E402


def region_state(n, d, seed=1):
    rs = np.random.RandomState(seed)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    ctr = u.mean(axis=0)
    cov = np.cov(u, rowvar=0) * (d + 2)
    w, v = np.linalg.eigh(cov)
    T = v * w ** -0.5
    unormed = np.dot(u - ctr, T)
    masks = np.zeros((30, n), dtype=bool)
    for b in range(30):
        masks[b, rs.randint(n, size=n)] = True
    t0 = time.time()
    r, _ = K.maxradiussq_bootstrap(unormed, masks)
    t1 = time.time()
    r, _ = K.maxradiussq_bootstrap(unormed, masks)
    t2 = time.time()
    print(json.dumps(dict(what="maxradiussq_bootstrap host call (30 rounds)", n=n, d=d, first_ms=(t1 - t0) * 1e3,
                          second_ms=(t2 - t1) * 1e3, r2=r.max())), flush=True)
    inv = np.linalg.inv(cov)
    enlarge = 1.95 if d == 50 else 2.4
    return u, ctr, T, unormed, cov, inv, float(r.max()), enlarge


def main():
    dev = torch.device("cuda:0")
    print(json.dumps(dict(what="fp64_valu_probe_TFLOPs", value=K.bench_fp64_valu())), flush=True)
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    for (n, d, p) in (((4000, 50, 1000000),) if quick else ((4000, 50, 1000000), (2000, 20, 100000), (4000, 50, 100000))):
        u, ctr, T, unormed, cov, inv, r2, enlarge = region_state(n, d)
        reg = K.DeviceRegion()
        reg.set(unormed, 0, ctr, T, None, ctr, inv, enlarge, r2)
        # set E: uniform in the wrapping ellipsoid
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        z = torch.randn(p, d, dtype=torch.float64, device=dev, generator=g)
        z /= z.norm(dim=1, keepdim=True)
        rad = torch.rand(p, 1, dtype=torch.float64, device=dev, generator=g) ** (1.0 / d)
        lam, vec = np.linalg.eigh(inv)
        axes_T = torch.tensor((vec * (1 / np.sqrt(lam))).T.copy(), device=dev)
        pts = torch.tensor(ctr, device=dev) + (z * (enlarge ** 0.5) * rad) @ axes_T
        pts = pts.contiguous()
        mask = torch.empty(p, dtype=torch.uint8, device=dev)
        idx = torch.empty(p, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        from ultranest_amd import _lib
        for label, rr in ((("E", r2),) if quick else (("E", r2), ("F", 1e-300))):
            reg.set_thresholds(enlarge, rr)
            reg.first_index_dev(pts.data_ptr(), p, idx.data_ptr(), stream)
            torch.cuda.synchronize()
            work = torch.where(idx >= 0, idx + 1, torch.full_like(idx, n))
            work = torch.where(idx == -2, torch.zeros_like(idx), work).sum().item()
            tot, scan = reg.time_inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream, reps=3)
            flops = 3.0 * d * work
            print("%-12s prep %.3f ms  scan-stage %.3f ms" % (label, tot - scan, scan), flush=True)
            print(json.dumps(dict(what="inside", set=label, n=n, d=d, p=p, accept=float(mask.float().mean()),
                                  ell_pass=float((idx != -2).float().mean()),
                                  ms_total=tot, ms_scan=scan, proposals_per_s=p / (tot * 1e-3),
                                  scan_alg_TFLOPs=flops / (scan * 1e-3) / 1e12,
                                  mean_scan_fraction=work / (p * n))), flush=True)
        reg.close()


if __name__ == "__main__":
    main()
