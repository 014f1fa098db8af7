"""What a plain device copy / read reaches on this part (context for roofline_prep): torch kernels, 400 MB buffers."""
import json
import time

import torch

dev = torch.device("cuda", 0)
x = torch.rand(50_000_000, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
out = {}
for name, fn, nbytes in (("copy (read + write)", lambda: y.copy_(x), 2 * x.numel() * 8), ("sum (read only)", lambda: x.sum(), x.numel() * 8),
                         ("fill (write only)", lambda: y.fill_(1.0), x.numel() * 8)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out[name] = {"ms": dt * 1e3, "TBps": nbytes / dt / 1e12}
print(json.dumps(out))
