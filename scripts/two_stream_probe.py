"""Does the membership pipeline gain from running two half batches on two streams (two region handles with their own
scratch), so that the HBM-bound stage of one half overlaps the matrix-bound sweeps of the other and the ramps / tails
of the launches fill each other?   python scripts/two_stream_probe.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
P = int(os.environ.get("MLF_TS_P", bench.NPROPOSALS))
u, region = bench.build_region(None)
h0 = region._dev.sync(region, True)
u2, region2 = bench.build_region(None)
h1 = region2._dev.sync(region2, True)
pts = bench.proposals_in_ellipsoid(region, P, 1000, dev)
mask = torch.empty(P, dtype=torch.uint8, device=dev)
ref = torch.empty(P, dtype=torch.uint8, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream().cuda_stream
for _ in range(100):
    h0.inside_dev(pts.data_ptr(), P, ref.data_ptr(), cur)
torch.cuda.synchronize()


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def one():
    h0.inside_dev(pts.data_ptr(), P, mask.data_ptr(), cur)


def split(nchunk):
    def run():
        c = P // nchunk
        for i in range(nchunk):
            h = h0 if i % 2 == 0 else h1
            st = s0 if i % 2 == 0 else s1
            h.inside_dev(pts.data_ptr() + i * c * 50 * 8, c, mask.data_ptr() + i * c, st.cuda_stream)
    return run


def serial(nchunk):
    def run():
        c = P // nchunk
        for i in range(nchunk):
            h0.inside_dev(pts.data_ptr() + i * c * 50 * 8, c, mask.data_ptr() + i * c, cur)
    return run


out = {"one_stream_ms": timed(one)}
for n in (2, 4, 8):
    out["two_streams_%d_chunks_ms" % n] = timed(split(n))
    out["equal_%d" % n] = bool((mask == ref).all().item())
    out["one_stream_%d_chunks_ms" % n] = timed(serial(n))
# WHOLE batches alternating between two handles on two streams: the tail of one batch (k_uncertain: a latency chain on 181 of 256
# CUs, k_scan tail) under the first launch of the next.  ms per BATCH.
pts_b = bench.proposals_in_ellipsoid(region, P, 2000, dev)
mask_b = torch.empty(P, dtype=torch.uint8, device=dev)
ref_b = torch.empty(P, dtype=torch.uint8, device=dev)
h0.inside_dev(pts_b.data_ptr(), P, ref_b.data_ptr(), cur)
torch.cuda.synchronize()


def alternate():
    h0.inside_dev(pts.data_ptr(), P, mask.data_ptr(), s0.cuda_stream)
    h1.inside_dev(pts_b.data_ptr(), P, mask_b.data_ptr(), s1.cuda_stream)


def back_to_back():
    h0.inside_dev(pts.data_ptr(), P, mask.data_ptr(), cur)
    h0.inside_dev(pts_b.data_ptr(), P, mask_b.data_ptr(), cur)


out["two_batches_one_stream_ms_per_batch"] = timed(back_to_back) / 2
out["two_batches_two_streams_ms_per_batch"] = timed(alternate) / 2
out["alternate_equal"] = bool((mask == ref).all().item() and (mask_b == ref_b).all().item())
out["two_batches_one_stream_again_ms_per_batch"] = timed(back_to_back) / 2
out["one_stream_again_ms"] = timed(one)
print(json.dumps(out))
