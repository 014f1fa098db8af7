"""N > 1 path on CPU: world_size-2/3 `gloo` process groups (one process per would-be GPU).  The
kernels are the oracle stub (no GPU here); what is under test is the HOST logic of
ultranest_amd.distributed: mask broadcast, balanced sharding, the MAX all-reduce, error
propagation -- and that the result is bit-identical to the single-process run for any world size.
"""
import os
import socket
import sys

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_exactly():
    from ultranest_amd.distributed import shard_bounds
    for n in (0, 1, 7, 30, 64, 1000003):
        for w in (1, 2, 3, 8, 64):
            seen = []
            for r in range(w):
                lo, hi = shard_bounds(n, r, w)
                assert 0 <= hi - lo <= n // w + 1
                seen.extend(range(lo, hi)) if n < 2000 else None
                if r == w - 1:
                    assert hi == n
            if n < 2000:
                assert seen == list(range(n))
    assert [shard_bounds(30, r, 8) for r in range(8)] == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20), (20, 24), (24, 27), (27, 30)]


@pytest.mark.parametrize("n,W", [(300, 2), (300, 3), (131, 8), (64, 4), (1000, 8)])
def test_row_block_shares_combine_to_the_full_bootstrap(n, W, monkeypatch):
    """MLFriends.enlargement_share: radius by 64-row blocks (all rounds), factor by rounds; the element-wise maximum over
    the ranks' shares (the all-reduce of C1) equals enlargement_from_masks on all rounds, bit for bit -- also when a rank's
    block range is empty (more ranks than blocks) or a round is skipped."""
    import oracle_backend
    oracle_backend.install(monkeypatch)
    import ultranest_amd.mlfriends as M
    from ultranest_amd import regions
    u = inputs.live_points(77 + n, n, 4)
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    masks = regions._draw_selection(np.random.RandomState(n), n, 30)
    masks[7] = True      # skipped round (reference :1048)
    r1, f1 = region.enlargement_from_masks(masks)
    shares = [region.enlargement_share(masks, r, W) for r in range(W)]
    assert max(s[0] for s in shares) == r1
    assert max(s[1] for s in shares) == f1
    if n <= 64 * (W - 1):                   # fewer row blocks than ranks: the surplus ranks contribute the neutral 0
        assert min(s[0] for s in shares) == 0.0


def _free_port():
    """rendezvous token for the spawned ranks: a fresh temporary FILE (file:// store) -- a port found free here could be
    taken by the time the ranks bind it (seen once in a few hundred runs: EADDRINUSE)"""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="mlf_rdzv_")
    os.close(fd)
    os.unlink(path)          # the store creates it
    return path


def _init(dist, backend, token, rank, world_size, **kw):
    if backend == "gloo":
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # the container's hostname may not resolve
    dist.init_process_group(backend, init_method="file://" + token, rank=rank, world_size=world_size, **kw)


def _worker(rank, world_size, port, singular, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    import oracle_backend
    import ultranest_amd.kernels as K
    import ultranest_amd.mlfriends as M
    for name in oracle_backend.PATCHED:                     # plain (non-pytest) install of the stub
        setattr(K, name, getattr(oracle_backend, name))
        if hasattr(M, name):
            setattr(M, name, getattr(oracle_backend, name))
    from ultranest_amd import distributed
    _init(dist, "gloo", port, rank, world_size)
    try:
        u = inputs.live_points(31, 300, 4)
        if singular:
            u[:, 3] = u[:, 0]
        layer = M.AffineLayer() if not singular else M.ScalingLayer()
        layer.optimize(u, u)
        region = M.MLFriends(u, layer)
        # every rank passes a DIFFERENT stream: only rank 0's draws may matter
        rng = np.random.RandomState(1234 if rank == 0 else 999 + rank)
        try:
            r, f = distributed.update_region_bootstrap(region, 30, minvol=0., rng=rng)
            out[rank] = (r, f, region.maxradiussq, region.enlarge)
        except np.linalg.LinAlgError:
            out[rank] = "LinAlgError"
        t = distributed.allreduce_max([float(rank), -float(rank)])
        assert list(t) == [world_size - 1.0, 0.0]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_sharded_bootstrap_bit_identical_to_single_process(world_size, monkeypatch):
    import torch.multiprocessing as mp
    import oracle_backend
    oracle_backend.install(monkeypatch)
    import ultranest_amd.mlfriends as M
    u = inputs.live_points(31, 300, 4)
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    r1, f1 = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(1234))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world_size, _free_port(), False, out), nprocs=world_size, join=True)
    for rank in range(world_size):
        assert out[rank] == (r1, f1, r1, f1), (rank, out[rank], (r1, f1))


def test_error_flag_reaches_every_rank(monkeypatch):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), True, out), nprocs=2, join=True)
    assert out[0] == "LinAlgError" and out[1] == "LinAlgError"


def _gpu_worker(rank, world_size, port, out):
    """two ranks, gloo for the (tiny) collectives, REAL HIP kernels for the compute (both ranks share
    GPU 0 on the single-GPU test box; on an 8-GPU node each rank would own one GPU and RCCL)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    import ultranest_amd.mlfriends as M
    from ultranest_amd import distributed
    _init(dist, "gloo", port, rank, world_size)
    try:
        u = inputs.live_points(31, 1500, 12)
        layer = M.AffineLayer()
        layer.optimize(u, u)
        region = M.MLFriends(u, layer)
        rng = np.random.RandomState(1234 if rank == 0 else 5 + rank)
        r, f = distributed.update_region_bootstrap(region, 30, minvol=0., rng=rng)
        out[rank] = (r, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_bootstrap_with_hip_kernels():
    import torch.multiprocessing as mp
    import ultranest_amd.mlfriends as M
    u = inputs.live_points(31, 1500, 12)
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    r1, f1 = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(1234))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == out[1] == (r1, f1), (dict(out), (r1, f1))


def _rccl_worker(rank, world_size, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch
    import torch.distributed as dist
    import ultranest_amd.mlfriends as M
    from ultranest_amd import distributed
    os.environ["MLF_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(0)
    _init(dist, "nccl", port, rank, world_size, device_id=torch.device("cuda", 0))
    try:
        u = inputs.live_points(31, 1500, 12)
        layer = M.AffineLayer()
        layer.optimize(u, u)
        region = M.MLFriends(u, layer)
        out["pair"] = distributed.update_region_bootstrap(region, 30, minvol=0., rng=np.random.RandomState(1234))
        out["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_collectives_on_rccl_with_one_rank():
    """The broadcast of the masks and the MAX all-reduce as RCCL calls on device tensors (one rank: all a 1-GPU box
    can do), same result as the plain call."""
    import torch.multiprocessing as mp
    import ultranest_amd.mlfriends as M
    u = inputs.live_points(31, 1500, 12)
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    want = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(1234))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert out["backend"] == "nccl"
    assert tuple(out["pair"]) == tuple(want), (dict(out), want)


def _one_rank_fails_worker(rank, world_size, port, out, exc_name="FloatingPointError"):
    """rank 1's shard raises a FloatingPointError (what np.errstate(all='raise') turns a numpy warning into under the
    driver, integrator.py:2066); rank 0's shard is fine.  Every rank must still reach the all-reduce and leave with the
    same exception type (a rank that skips the collective leaves the others blocked in it)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    import oracle_backend
    import ultranest_amd.kernels as K
    import ultranest_amd.mlfriends as M
    for name in oracle_backend.PATCHED:
        setattr(K, name, getattr(oracle_backend, name))
        if hasattr(M, name):
            setattr(M, name, getattr(oracle_backend, name))
    from ultranest_amd import distributed
    from ultranest_amd.harness import RegionUpdater
    from ultranest_amd._lib import HipLibraryError
    exc = {"FloatingPointError": FloatingPointError, "RuntimeError": RuntimeError, "HipLibraryError": HipLibraryError,
           "MemoryError": MemoryError}[exc_name]
    _init(dist, "gloo", port, rank, world_size)
    try:
        u = inputs.live_points(31, 300, 4)
        layer = M.AffineLayer()
        layer.optimize(u, u)
        region = M.MLFriends(u, layer)
        good = region.enlargement_share

        def shard(masks, rank_, size_, minvol=0.):
            if rank == 1:
                raise exc("invalid value encountered in this rank's shard")
            return good(masks, rank_, size_, minvol=minvol)
        region.enlargement_share = shard
        try:
            distributed.update_region_bootstrap(region, 30, minvol=0., rng=np.random.RandomState(3))
            out[rank] = "no error"
        except np.linalg.LinAlgError:
            out[rank] = "LinAlgError"
        except RuntimeError:
            out[rank] = "RuntimeError"
        # the wrapping ellipsoid of the harness: same hazard, same cure
        upd = RegionUpdater(4)
        np.random.seed(5)
        real = M.WrappingEllipsoid.enlargement_from_masks

        def tshard(self, masks):
            if rank == 1:
                raise exc("shard")
            return real(self, masks)
        import ultranest_amd.harness as H
        H.WrappingEllipsoid.enlargement_from_masks = tshard
        try:
            upd.update(u, nbootstraps=10, active_p=u * 3.0)
            out[10 + rank] = upd.tregion is None
        except RuntimeError:
            out[10 + rank] = "RuntimeError"
        t = distributed.allreduce_max([float(rank)])       # the group is still in step
        out[20 + rank] = float(t[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exc_name", ["FloatingPointError", "RuntimeError", "HipLibraryError", "MemoryError"])
def test_failure_on_one_rank_only_keeps_the_group_in_step(exc_name):
    """whatever a shard raises -- a numerical error, a device-layer RuntimeError (HipLibraryError was not in
    the caught set, the failing rank skipped the all-reduce and the others hung), an allocation failure"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_one_rank_fails_worker, args=(2, _free_port(), out, exc_name), nprocs=2, join=True)
    # a NUMERICAL failure comes back as LinAlgError on every rank (the caller keeps its old region, integrator.py:385-411);
    # anything else as RuntimeError on every rank -- it must not be swallowed by that handler
    if exc_name == "FloatingPointError":
        assert out[0] == "LinAlgError" and out[1] == "LinAlgError", dict(out)
        assert out[10] is True and out[11] is True, dict(out)
    else:
        assert out[0] == "RuntimeError" and out[1] == "RuntimeError", dict(out)
        assert out[10] == "RuntimeError" and out[11] == "RuntimeError", dict(out)
    assert out[20] == 1.0 and out[21] == 1.0, dict(out)


# ---- the C-ABI exchange step (mlf_comm_* / mlf_allreduce_max over librccl; include/mlfriends_hip.h) ------------------
def _abi_comm_worker(rank, world_size, idpath, out):
    sys.path.insert(0, ROOT)
    import ctypes
    import time
    from ultranest_amd import _lib
    L = _lib.lib()
    _lib.set_device(rank)
    buf = ctypes.create_string_buffer(128)
    if rank == 0:
        _lib.check(L.mlf_comm_unique_id(buf, 128))
        with open(idpath + ".tmp", "wb") as fh:
            fh.write(buf.raw)
        os.replace(idpath + ".tmp", idpath)
    else:
        for _ in range(600):
            if os.path.exists(idpath):
                break
            time.sleep(0.05)
        buf = ctypes.create_string_buffer(open(idpath, "rb").read(), 128)
    _lib.check(L.mlf_comm_init_rank(buf, 128, world_size, rank))
    v = np.array([1.0 + rank, -float(rank), 0.5], dtype=np.float64)
    _lib.check(L.mlf_allreduce_max(v.ctypes.data, 3))
    out[rank] = list(v)
    _lib.check(L.mlf_comm_destroy())


@pytest.mark.gpu
def test_abi_allreduce_max_one_rank_and_one_process(tmp_path):
    import ctypes
    from ultranest_amd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(128)
    _lib.check(L.mlf_comm_unique_id(buf, 128))
    _lib.check(L.mlf_comm_init_rank(buf, 128, 1, 0))
    v = np.array([3.0, -2.0, np.inf], dtype=np.float64)
    _lib.check(L.mlf_allreduce_max(v.ctypes.data, 3))
    assert list(v) == [3.0, -2.0, np.inf]
    # one process driving all visible devices: ndev rows, the maximum lands in every row
    ndev = _lib.device_count()
    _lib.check(L.mlf_comm_init(ndev))
    rows = np.arange(ndev * 4, dtype=np.float64).reshape(ndev, 4) * np.array([1.0, -1.0, 1.0, -1.0])
    want = rows.max(axis=0)
    _lib.check(L.mlf_allreduce_max(rows.ctypes.data, 4))
    assert np.array_equal(rows, np.broadcast_to(want, rows.shape))
    _lib.check(L.mlf_comm_destroy())
    with pytest.raises(ValueError):
        _lib.check(L.mlf_allreduce_max(v.ctypes.data, 3))      # no communicator


@pytest.mark.gpu
def test_abi_allreduce_max_two_ranks_over_rccl(tmp_path):
    """one process per GPU, the 128-byte id handed over through a file (what an MPI_Bcast would do)"""
    from ultranest_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_abi_comm_worker, args=(2, str(tmp_path / "id.bin"), out), nprocs=2, join=True)
    assert out[0] == out[1] == [2.0, 0.0, 0.5], dict(out)


def _rccl_two_rank_worker(rank, world_size, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch
    import torch.distributed as dist
    import ultranest_amd.mlfriends as M
    from ultranest_amd import _lib, distributed
    torch.cuda.set_device(rank)
    _lib.set_device(rank)
    _init(dist, "nccl", port, rank, world_size, device_id=torch.device("cuda", rank))
    try:
        u = inputs.live_points(31, 1500, 12)
        layer = M.AffineLayer()
        layer.optimize(u, u)
        region = M.MLFriends(u, layer)
        rng = np.random.RandomState(1234 if rank == 0 else 77)
        out[rank] = distributed.update_region_bootstrap(region, 30, minvol=0., rng=rng)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_bootstrap_two_ranks_over_rccl():
    """the rebuild's exchange step on two GPUs: device-side mask broadcast + MAX all-reduce over RCCL"""
    from ultranest_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    import ultranest_amd.mlfriends as M
    u = inputs.live_points(31, 1500, 12)
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    want = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(1234))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_two_rank_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert tuple(out[0]) == tuple(out[1]) == tuple(want), (dict(out), want)


def _gather_worker(rank, world_size, port, out):
    sys.path.insert(0, os.path.dirname(__file__))
    import torch.distributed as dist
    from ultranest_amd import distributed
    _init(dist, "gloo", port, rank, world_size)
    rs = np.random.RandomState(10 + rank)
    k = [3, 0, 5][rank]                                   # rank 1 accepted nothing in this round
    u, v, logl = rs.uniform(size=(k, 4)), rs.normal(size=(k, 6)), rs.normal(size=k)
    su, sv, sl, nc = distributed.allgather_samples(u, v, logl, 100 + rank)
    out.put((rank, su, sv, sl, nc, u, v, logl))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_samples_uneven_ranks():
    """reference integrator.py:1916-1928: rank-ordered concatenation of what every rank accepted, summed call counts"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    mp.spawn(_gather_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    res = sorted((out.get() for _ in range(3)), key=lambda t: t[0])
    want_u = np.concatenate([r[5] for r in res])
    want_v = np.concatenate([r[6] for r in res])
    want_l = np.concatenate([r[7] for r in res])
    for rank, su, sv, sl, nc, *_ in res:
        assert nc == 100 + 101 + 102
        assert np.array_equal(su, want_u) and np.array_equal(sv, want_v) and np.array_equal(sl, want_l)
    # one process: passthrough
    from ultranest_amd import distributed
    u, v, l_, nc = distributed.allgather_samples(np.ones((2, 3)), np.zeros((2, 1)), [1.0, 2.0], 7)
    assert u.shape == (2, 3) and v.shape == (2, 1) and list(l_) == [1.0, 2.0] and nc == 7


# ---- round 5: every region class through a real group; a world-size-8 rehearsal at C4 -------------------------------
def _plain_install():
    """install the oracle stand-in in a spawned rank (no pytest there)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import oracle_backend
    import ultranest_amd.kernels as K
    import ultranest_amd.mlfriends as M
    for name in oracle_backend.PATCHED:
        setattr(K, name, getattr(oracle_backend, name))
        if hasattr(M, name):
            setattr(M, name, getattr(oracle_backend, name))
    return M


def _make_region(M, cls_name, seed, n, d):
    u = inputs.live_points(seed, n, d)
    layer = M.AffineLayer()
    layer.optimize(u, u)
    return getattr(M, cls_name)(u, layer)


def _class_worker(rank, world_size, port, out, cls_name, n, d, nboot, failing_rank, exc_name):
    M = _plain_install()
    import torch.distributed as dist
    from ultranest_amd import distributed
    _init(dist, "gloo", port, rank, world_size)
    try:
        region = _make_region(M, cls_name, 31, n, d)
        if failing_rank is not None and rank == failing_rank:
            exc = {"AssertionError": AssertionError, "FloatingPointError": FloatingPointError,
                   "ZeroDivisionError": ZeroDivisionError}[exc_name]

            def shard(masks, rank_, size_, minvol=0.):
                raise exc("this rank's shard only")
            region.enlargement_share = shard
        rng = np.random.RandomState(1234 if rank == 0 else 999 + rank)     # only rank 0's draws may matter
        try:
            r, f = distributed.update_region_bootstrap(region, nboot, minvol=0., rng=rng)
            out[rank] = (r, f, region.maxradiussq, region.enlarge)
        except np.linalg.LinAlgError:
            out[rank] = "LinAlgError"
        except RuntimeError:
            out[rank] = "RuntimeError"
        t = distributed.allreduce_max([float(rank), 1.0])                    # still in step; every rank is seen
        ones = distributed.allreduce_max([1.0])
        out[100 + rank] = (float(t[0]), float(ones[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cls_name", ["RobustEllipsoidRegion", "SimpleRegion", "MLFriends"])
def test_every_region_class_keeps_its_own_rule_across_ranks(cls_name, monkeypatch):
    """with more than one rank RobustEllipsoidRegion / SimpleRegion went through the inherited MLFriends
    share -- friends radius instead of 1e300, full-covariance factor instead of the per-axis one.  Two ranks must return
    what one process returns, bit for bit, for every region class (reference: mlfriends.pyx:1392-1440, 1511-1548)."""
    import torch.multiprocessing as mp
    import oracle_backend
    oracle_backend.install(monkeypatch)
    import ultranest_amd.mlfriends as M
    region = _make_region(M, cls_name, 31, 300, 4)
    r1, f1 = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(1234))
    if cls_name != "MLFriends":
        assert r1 == 1e300
    out = mp.Manager().dict()
    mp.spawn(_class_worker, args=(2, _free_port(), out, cls_name, 300, 4, 30, None, None), nprocs=2, join=True)
    for rank in range(2):
        assert out[rank] == (r1, f1, r1, f1), (rank, out[rank], (r1, f1))


def test_world_size_8_rehearsal_at_c4(monkeypatch):
    """C4's shape (N = 4000, d = 50, 30 rounds) through EIGHT real ranks and their collectives --
    the (4,4,4,4,4,4,3,3) round split and the 63-row-block split (8,8,8,8,8,8,8,7) -- bit-identical to one process
    (integrator.py:375-415 is the step this replaces)."""
    import torch.multiprocessing as mp
    import oracle_backend
    oracle_backend.install(monkeypatch)
    import ultranest_amd.mlfriends as M
    from ultranest_amd.distributed import shard_bounds
    assert [b - a for a, b in (shard_bounds(63, r, 8) for r in range(8))] == [8, 8, 8, 8, 8, 8, 8, 7]
    region = _make_region(M, "MLFriends", 31, 4000, 50)
    r1, f1 = region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(1234))
    out = mp.Manager().dict()
    mp.spawn(_class_worker, args=(8, _free_port(), out, "MLFriends", 4000, 50, 30, None, None), nprocs=8, join=True)
    for rank in range(8):
        assert out[rank] == (r1, f1, r1, f1), (rank, out[rank], (r1, f1))
        assert out[100 + rank] == (7.0, 1.0)


@pytest.mark.parametrize("exc_name,want", [("FloatingPointError", "LinAlgError"), ("AssertionError", "RuntimeError"),
                                           ("ZeroDivisionError", "RuntimeError")])
def test_error_class_from_rank_5_of_8_reaches_every_rank(exc_name, want):
    """only rank 5 fails; all eight leave with the same exception type and the group stays in step.  A numerical failure
    (the three types of the driver's handlers, integrator.py:2123-2131) -> LinAlgError; an AssertionError or a
    ZeroDivisionError is NOT one of them (a single process propagates those) -> RuntimeError everywhere."""
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_class_worker, args=(8, _free_port(), out, "MLFriends", 300, 4, 30, 5, exc_name), nprocs=8, join=True)
    for rank in range(8):
        assert out[rank] == want, dict(out)
        assert out[100 + rank] == (7.0, 1.0)
