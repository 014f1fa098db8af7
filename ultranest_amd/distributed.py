"""Multi-GPU sharding of the region rebuild and of proposal batches: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

What the reference does (integrator.py:375-415, `_update_region_bootstrap`): every MPI rank runs
``nbootstraps // mpi_size`` rounds on identical live points with its OWN random seed, then the
per-rank maxima are exchanged with pickle gather + bcast and maximised.  Consequences: the
number of rounds actually run depends on the rank count (30 // 8 * 8 = 24) and results are not
reproducible across rank counts.

What this module does instead: rank 0 draws all B selection masks from ONE stream (the same
draws a single-process run makes) and broadcasts them (B*N bytes); ranks take contiguous shards
of the rounds (sizes differ by at most one: 30 over 8 GPUs = 4,4,4,4,4,4,3,3); each computes its
shard's maxima on its GPU; ONE all-reduce(MAX) of three doubles (radius^2, enlargement, error
flag) finishes the step.  max is exact and order independent and the float32 rounding of the
radius is monotone, so the result is bit-identical for every world size.  An error on any rank
travels as the explicit flag (not as a NaN through MAX) and is re-raised on all ranks after the
collective, which keeps the ranks in step -- the property the reference protects with its
try/except around compute_enlargement (integrator.py:385-411).

Proposal batches shard by rows with NO collective: rows are independent and every rank holds the
(1.6 MB) region state; `shard_bounds` gives the slice.
"""
import numpy as np

from . import regions


# A shard of bootstrap rounds can raise numerical errors (the driver runs the rebuild under np.errstate(all='raise'),
# integrator.py:2066: LinAlgError, FloatingPointError, Warning), assertions of its own and anything the device layer reports
# (HipLibraryError is a RuntimeError; an out-of-memory allocation a MemoryError).  EVERY exception of a shard is caught
# per rank, exchanged as a flag and re-raised on every rank after the all-reduce: a rank that skipped the collective
# would leave the other ranks blocked in it.
SHARD_ERRORS = (Exception,)
# what a shard's failure means for the caller: numerical failures -- exactly the three types the driver's own handlers keep
# the old region on (integrator.py:2123-2131: Warning, FloatingPointError, LinAlgError) -- travel as class 1; everything else,
# AssertionError and ZeroDivisionError included (a single process lets those propagate, so must a group), as class 2 and
# comes back as a RuntimeError on every rank
NUMERICAL_ERRORS = (np.linalg.LinAlgError, FloatingPointError, Warning)


def _error_class(error):
    if error is None:
        return 0.0
    return 1.0 if isinstance(error, NUMERICAL_ERRORS) else 2.0


def _dist():
    import torch.distributed as dist
    return dist


def _forced():
    """MLF_FORCE_COLLECTIVES=1: issue the broadcast / all-reduce also in a one-rank group (lets a single GPU
    exercise the RCCL code path; the results are unchanged)."""
    import os
    return os.environ.get("MLF_FORCE_COLLECTIVES", "") not in ("", "0")


def _initialised():
    try:
        dist = _dist()
    except ImportError:
        return False
    return dist.is_available() and dist.is_initialized()


def world(group=None):
    """(rank, world_size) of the default/group process group; (0, 1) when not initialised."""
    try:
        dist = _dist()
    except ImportError:
        return 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_bounds(nitems, rank, world_size):
    """Contiguous, balanced [lo, hi) slice of `nitems` work items for `rank`; the first
    ``nitems % world_size`` ranks take one extra item."""
    base, extra = divmod(int(nitems), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _tensor_device(group=None):
    import torch
    dist = _dist()
    backend = dist.get_backend(group)
    if backend == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_masks(masks, npoints, nbootstraps, group=None, src=0, keep_on_device=False):
    """Rank `src` provides the (B, N) bool masks; every rank returns the same matrix.  With `keep_on_device` and a
    device backend (RCCL) the result is the uint8 device tensor of the broadcast itself -- the bootstrap kernels take
    device pointers for the masks, so nothing travels device -> host -> device; otherwise a numpy bool array."""
    rank, size = world(group)
    if size == 1 and not (_forced() and _initialised()):
        return masks
    import torch
    dist = _dist()
    dev = _tensor_device(group)
    if rank == src:
        buf = torch.from_numpy(np.ascontiguousarray(masks, dtype=np.uint8)).to(dev)
    else:
        buf = torch.empty((nbootstraps, npoints), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src, group=group)
    if keep_on_device and dev.type == "cuda":
        # stream contract: the consumers (mlf_* calls) copy from this tensor on the LIBRARY's stream, which knows nothing
        # of torch's streams -- the broadcast (queued behind torch's current stream) has to be complete before they start
        torch.cuda.current_stream(dev).synchronize()
        return buf
    return buf.cpu().numpy().astype(bool)


def allreduce_max(values, group=None):
    """Element-wise MAX of a small float64 vector over all ranks (RCCL ncclMax on the GPU box)."""
    rank, size = world(group)
    values = np.asarray(values, dtype=np.float64)
    if size == 1 and not (_forced() and _initialised()):
        return values
    import torch
    dist = _dist()
    t = torch.from_numpy(values.copy()).to(_tensor_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy()


def sharded_enlargement(region, nbootstraps, minvol=0., rng=np.random, group=None, masks=None):
    """(maxradiussq, enlarge) over `nbootstraps` rounds sharded across the process group.
    Single process: identical to ``region.compute_enlargement`` (same draws, same bits).
    `masks`: the (B, N) selection matrix if the caller has drawn it already -- from the same stream position this call would
    have drawn it from (harness.RegionUpdater draws on its worker thread while the region object is being built)."""
    rank, size = world(group)
    npoints = len(region.u)
    start = getattr(region, "_start_ellipsoid_parts", None)
    if start is not None:
        start(minvol)      # the host LAPACK of the create_ellipsoid that follows: on the worker thread from here on
    # every rank draws (so that rank-replicated host logic that uses the same stream afterwards stays in step across
    # the ranks when they are seeded alike); rank 0's draw is the one that counts
    if masks is None:
        masks = regions._draw_selection(rng, npoints, nbootstraps)
    masks = broadcast_masks(masks, npoints, nbootstraps, group=group, keep_on_device=True)
    lo, hi = shard_bounds(nbootstraps, rank, size)
    error = None
    r = f = 0.0
    try:
        share = getattr(region, "enlargement_share", None)
        if share is not None and size > 1:
            # MLFriends: the radius by ROW BLOCKS (all rounds, 1/size of the pair distances), the factor by rounds;
            # RobustEllipsoidRegion / SimpleRegion override the share with their OWN per-round rule (round shards)
            out = share(masks, rank, size, minvol=minvol)
        else:
            out = region.enlargement_from_masks(masks[lo:hi], minvol=minvol) if hi > lo else (0.0, 0.0)
        r, f = out if isinstance(out, tuple) else (0.0, out)
    except SHARD_ERRORS as e:     # whatever happens in this rank's shard, the rank still takes part in the collective
        error = e
    r, f, flag = allreduce_max([r, f, _error_class(error)], group=group)
    if flag > 0:
        # a failure in a single-process run keeps its own type
        if size == 1 and error is not None:
            raise error
        # the same exception type on every rank (the ranks must take the same branch in the caller).  Class 1 = a
        # NUMERICAL failure somewhere (the caller keeps its old region, as the reference does: integrator.py:385-411);
        # class 2 = anything else -- a device error, out of memory, a plain bug -- must not be swallowed by that handler
        if flag >= 2:
            raise RuntimeError("a rank of the group failed in its bootstrap shard (not a numerical error)") from error
        raise np.linalg.LinAlgError("compute_enlargement failed on rank(s) of the group") from error
    return float(r), float(f)


def update_region_bootstrap(region, nbootstraps, minvol=0., group=None, rng=np.random, masks=None):
    """Counterpart of the reference's ``_update_region_bootstrap`` (integrator.py:375-415): sets
    ``region.maxradiussq`` and ``region.enlarge`` and returns them."""
    assert nbootstraps > 0, nbootstraps
    r, f = sharded_enlargement(region, nbootstraps, minvol=minvol, rng=rng, group=group, masks=masks)
    if not (r > 0 and f > 0 and np.isfinite(r) and np.isfinite(f)):
        raise np.linalg.LinAlgError("compute_enlargement failed")
    region.maxradiussq = r
    region.enlarge = f
    return r, f


def allgather_samples(u, v, logl, ncall, group=None):
    """Every rank contributes the points it accepted in this round -- `u` (k, x_dim), `v` (k, num_params), `logl`
    (k,), k may differ from rank to rank and may be 0 -- and its number of likelihood calls; every rank gets the
    rank-ordered concatenation and the summed call count.  Counterpart of the reference's gather + bcast of pickled
    arrays (integrator.py:1916-1928), as two collectives on one padded tensor: an all-gather of the row counts and
    an all-gather of the rows."""
    u = np.atleast_2d(np.asarray(u, dtype=np.float64))
    v = np.atleast_2d(np.asarray(v, dtype=np.float64))
    logl = np.asarray(logl, dtype=np.float64).reshape(-1)
    if not (len(u) == len(v) == len(logl)):
        raise ValueError("u, v and logl must have one row per accepted point")
    rank, size = world(group)
    if size == 1 and not (_forced() and _initialised()):
        return u, v, logl, int(ncall)
    import torch
    dist = _dist()
    dev = _tensor_device(group)
    xdim, npar = u.shape[1], v.shape[1]
    mine = torch.tensor([len(u), int(ncall)], dtype=torch.int64, device=dev)
    heads = [torch.empty_like(mine) for _ in range(size)]
    dist.all_gather(heads, mine, group=group)
    counts = [int(h[0].item()) for h in heads]
    total_calls = sum(int(h[1].item()) for h in heads)
    width = xdim + npar + 1
    rows = max(max(counts), 1)
    pack = torch.zeros((rows, width), dtype=torch.float64, device=dev)
    if len(u):
        pack[:len(u)] = torch.from_numpy(np.concatenate((u, v, logl[:, None]), axis=1)).to(dev)
    parts = [torch.empty_like(pack) for _ in range(size)]
    dist.all_gather(parts, pack, group=group)
    merged = torch.cat([p[:k] for p, k in zip(parts, counts)], dim=0).cpu().numpy()
    return merged[:, :xdim], merged[:, xdim:xdim + npar], merged[:, -1], total_calls
