"""numpy-facing wrappers of the HIP kernels: the functions the reference exposes from
``ultranest/mlfriends.pyx`` as module-level (c)def kernels, same names and argument meaning.

Every function here runs on the MI355X through libmlfriends_hip.so; none has a CPU path.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, f64, ptr

int_dtype = np.int64   # reference mlfriends.pyx:26


def find_nearby(apts, bpts, radiussq, nnearby):
    """Index of the first point of `apts` within square radius `radiussq` of each `bpts` row,
    -1 if none; written into `nnearby` (int64, len(bpts)).  Reference mlfriends.pyx:143-183."""
    apts, bpts = f64(apts), f64(bpts)
    if apts.ndim != 2 or bpts.ndim != 2:
        raise ValueError("apts and bpts must be 2-d")
    nb = bpts.shape[0]
    d = bpts.shape[1] if nb else apts.shape[1]
    if nb and apts.shape[0] and apts.shape[1] != bpts.shape[1]:
        raise ValueError("dimensionality mismatch")
    if nnearby.dtype != int_dtype or nnearby.ndim != 1 or len(nnearby) != nb:
        raise ValueError("nnearby must be an int64 vector of len(bpts)")
    out = nnearby if nnearby.flags.c_contiguous else np.empty(nb, dtype=int_dtype)
    check(_lib.lib().mlf_find_nearby(ptr(apts), apts.shape[0], ptr(bpts), nb, d, float(radiussq), ptr(out)))
    if out is not nnearby:
        nnearby[:] = out


def count_nearby(apts, bpts, radiussq, nnearby):
    """Number of `apts` within `radiussq` of each `bpts` row.  Reference mlfriends.pyx:31-68."""
    apts, bpts = f64(apts), f64(bpts)
    nb = bpts.shape[0]
    d = bpts.shape[1] if nb else apts.shape[1]
    out = nnearby if nnearby.flags.c_contiguous else np.empty(nb, dtype=int_dtype)
    check(_lib.lib().mlf_count_nearby(ptr(apts), apts.shape[0], ptr(bpts), nb, d, float(radiussq), ptr(out)))
    if out is not nnearby:
        nnearby[:] = out


def subtract_nearby(upoints, maxradiussq):
    """upoints minus the mean of the points within `maxradiussq`.  Reference mlfriends.pyx:118-138."""
    pts = f64(upoints)
    out = np.empty_like(pts)
    if pts.shape[0]:
        check(_lib.lib().mlf_subtract_nearby(ptr(pts), pts.shape[0], pts.shape[1], float(maxradiussq), ptr(out)))
    return out


def cluster_labels(tpoints, radiussq, previous=None):
    """Friends-of-friends labels of `tpoints` (linking length sqrt(`radiussq`)): ``(nclusters, labels)`` with the
    numbering of the reference's growth loop (mlfriends.pyx:275-343: labels from 1, cluster c seeded at the first point
    that carried c in `previous`).  One all-pairs pass of exact distances on the device (hit bits), the growth rounds
    replayed on the bit rows (mlf_cluster_labels)."""
    import ctypes
    tpoints = f64(tpoints)
    n, d = tpoints.shape
    prev = None if previous is None else np.ascontiguousarray(np.asarray(previous)[:n], dtype=int_dtype)
    if prev is not None and len(prev) < n:   # fewer old ids than points: the rest carried none
        prev = np.concatenate((prev, np.zeros(n - len(prev), dtype=int_dtype)))
    labels = np.empty(n, dtype=int_dtype)
    ncl = ctypes.c_int64(0)
    check(_lib.lib().mlf_cluster_labels(ptr(tpoints), n, d, float(radiussq), ptr(prev), ptr(labels), ctypes.byref(ncl)))
    return int(ncl.value), labels


def maxradiussq_bootstrap(unormed, selected, rows=None):
    """Per-bootstrap ``compute_maxradiussq(unormed[sel], unormed[~sel])`` (reference
    mlfriends.pyx:188-224, called at :1012 and :1052) for a (B, N) boolean selection matrix.

    Returns (r2[B] -- already float32-rounded like the reference's ``cdef float`` --, skipped[B]).
    ``rows=(lo, hi)``: only the rows lo <= j < hi count as left-out points (one rank's share of a bootstrap sharded by
    row blocks, ``mlf_maxradiussq_bootstrap_rows``); the maximum over the shares is the full result, bit for bit."""
    pts = f64(unormed)
    sel_ptr, B, n, _keep = _mask_arg(selected)
    if n != pts.shape[0]:
        raise ValueError("selection mask length != number of points")
    r2 = np.empty(B, dtype=np.float64)
    skipped = np.empty(B, dtype=np.uint8)
    if rows is None:
        check(_lib.lib().mlf_maxradiussq_bootstrap(ptr(pts), n, pts.shape[1], sel_ptr, B, ptr(r2), ptr(skipped)))
    else:
        lo, hi = int(rows[0]), int(rows[1])
        check(_lib.lib().mlf_maxradiussq_bootstrap_rows(ptr(pts), n, pts.shape[1], sel_ptr, B, lo, hi, ptr(r2), ptr(skipped)))
    return r2, skipped.astype(bool)


class DevArray(object):
    """A block of device memory owned by the library's allocator (mlf_dev_alloc): what ultranest_amd.device_rebuild keeps
    its live points in between calls.  `ptr` goes wherever an entry point documents "host or device"."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(_lib.lib().mlf_dev_alloc(self.nbytes, ctypes.byref(p)))
        self.ptr = p

    @classmethod
    def from_host(cls, array):
        a = np.ascontiguousarray(array)
        self = cls(a.nbytes)
        # synchronous: `a` may be the CALLER's own array (ascontiguousarray copies nothing then) and pageable memory is
        # read when the stream reaches the copy -- the call returns only after that, so the caller may write to its array
        # again at once (sync = 0 saved nothing, no upload here is followed by another host-side copy)
        check(_lib.lib().mlf_dev_copy(self.ptr, ptr(a), a.nbytes, 1))
        return self

    def to_host(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(_lib.lib().mlf_dev_copy(ptr(out), self.ptr, out.nbytes, 1))
        return out

    def data_ptr(self):
        return self.ptr.value

    def free(self):
        if self.ptr is not None and self.ptr.value:
            _lib.lib().mlf_dev_free(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def col_extent(pts, n=None, d=None):
    """(lo[d], hi[d]) of the columns of an (n, d) float64 array: a numpy array or a DevArray (then n, d are required)."""
    if isinstance(pts, DevArray):
        p = pts.ptr
    else:
        pts = f64(pts)
        n, d = pts.shape
        p = ptr(pts)
    lo, hi = np.empty(d), np.empty(d)
    check(_lib.lib().mlf_col_extent(p, int(n), int(d), ptr(lo), ptr(hi)))
    return lo, hi


def _mask_arg(selected):
    """(pointer, B, n, keep-alive) of a (B, n) selection matrix: numpy bool / uint8 on the host, or a uint8 torch
    tensor on the device (the masks of a device-side broadcast stay where they are)."""
    if hasattr(selected, "data_ptr"):          # torch tensor
        t = selected if selected.dim() == 2 else selected[None, :]
        if not t.is_contiguous():
            t = t.contiguous()
        assert t.element_size() == 1, t.dtype
        return ctypes.c_void_p(t.data_ptr()), int(t.shape[0]), int(t.shape[1]), t
    sel = np.ascontiguousarray(selected, dtype=np.uint8)
    if sel.ndim == 1:
        sel = sel[None, :]
    return ptr(sel), sel.shape[0], sel.shape[1], sel


def compute_mean_pair_distance(pts, clusterids):
    """Mean distance between same-cluster pairs (cluster id 0 = unassigned is excluded).
    Reference mlfriends.pyx:229-270.  The N^2 d squared distances are computed on the GPU; the
    square root and the reference's single running sum (j outer, i<j inner) are applied on the
    host, which keeps the result bit-identical."""
    pts = f64(pts)
    ids = np.asarray(clusterids)
    n = pts.shape[0]
    if n < 2:
        return np.float64(np.nan)
    d2 = np.empty(n * (n - 1) // 2, dtype=np.float64)
    check(_lib.lib().mlf_pair_dist2_lower(ptr(pts), n, pts.shape[1], ptr(d2)))
    jj, ii = np.tril_indices(n, -1)            # row-major lower triangle == (j outer, i inner)
    keep = (ids[jj] != 0) & (ids[jj] == ids[ii])
    vals = np.sqrt(d2[keep])
    npairs = int(keep.sum())
    total = np.cumsum(vals)[-1] if npairs else 0.0   # cumsum = strictly sequential summation
    assert np.isfinite(total), total
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.float64(total) / npairs if npairs else np.float64(np.nan)


def inside_ellipsoid(points, ellipsoid_center, ellipsoid_invcov, square_radius, return_q=False):
    """``einsum('ij,jk,ik->i', d, invcov, d) <= square_radius``; reference mlfriends.pyx:882-912."""
    pts = f64(points)
    ctr, inv = f64(ellipsoid_center), f64(ellipsoid_invcov)
    n = pts.shape[0]
    mask = np.empty(n, dtype=np.uint8)
    q = np.empty(n, dtype=np.float64) if return_q else None
    if n:
        check(_lib.lib().mlf_inside_ellipsoid(ptr(pts), n, pts.shape[1], ptr(ctr), ptr(inv),
                                              float(square_radius), ptr(mask), ptr(q)))
    mask = mask.view(np.bool_)
    return (mask, q) if return_q else mask


def affine_transform(points, ctr, T, wrap_shift=None):
    """``np.dot(wrap(points) - ctr, T)``; reference mlfriends.pyx:737-743."""
    pts = f64(points)
    d = pts.shape[1]
    ctr = f64(np.broadcast_to(ctr, (d,)))
    T = f64(T)
    out = np.empty_like(pts)
    ws = None if wrap_shift is None else f64(wrap_shift)
    if pts.shape[0]:
        check(_lib.lib().mlf_affine_transform(ptr(pts), pts.shape[0], d, ptr(ctr), ptr(T), ptr(ws), ptr(out)))
    return out


def bootstrap_moments(u, selected):
    """mean (B, d) and ddof=1 sample covariance (B, d, d) of the selected rows per bootstrap."""
    u = f64(u)
    sel = np.ascontiguousarray(selected, dtype=np.uint8)
    B, n = sel.shape
    d = u.shape[1]
    mean = np.empty((B, d))
    cov = np.empty((B, d, d))
    check(_lib.lib().mlf_bootstrap_moments(ptr(u), n, d, ptr(sel), B, ptr(mean), ptr(cov)))
    return mean, cov


def bootstrap_factor(u, selected, scale):
    """f[B]: per round the largest quadratic form of the left-out rows with the (scale x) covariance of
    the selected rows, all on the device (d <= 64); NaN marks a round with a singular matrix."""
    u = f64(u)
    sel_ptr, B, n, _keep = _mask_arg(selected)
    f = np.empty(B)
    check(_lib.lib().mlf_bootstrap_factor(ptr(u), n, u.shape[1], sel_ptr, B, float(scale), ptr(f)))
    return f


def bootstrap_quadform_max(u, selected, ctr, invcov):
    """f[b] = max over rows NOT selected in bootstrap b of (u-ctr_b)^T invcov_b (u-ctr_b);
    reference mlfriends.pyx:1060-1062."""
    u = f64(u)
    sel = np.ascontiguousarray(selected, dtype=np.uint8)
    B, n = sel.shape
    ctr, invcov = f64(ctr), f64(invcov)
    f = np.empty(B)
    check(_lib.lib().mlf_bootstrap_quadform_max(ptr(u), n, u.shape[1], ptr(sel), B, ptr(ctr), ptr(invcov), ptr(f)))
    return f


class DeviceRegion(object):
    """Device-resident state of one region (live points in both layouts, layer, wrapping
    ellipsoid, thresholds) behind the opaque ``mlf_region`` handle."""

    # Handles of regions that went out of use, with their device buffers.  A rebuild makes a new region object
    # every time and the old ones sit in reference cycles (region -> bound methods -> region) until the cyclic
    # collector runs: five at once, ~50 hipFree calls each -- a 55 ms stall in every fifth rebuild.  A recycled
    # handle keeps its allocations; mlf_region_set overwrites everything they hold.
    _idle = []
    _IDLE_MAX = 8

    def __init__(self):
        if DeviceRegion._idle:
            self._h = DeviceRegion._idle.pop()
        else:
            self._h = ctypes.c_void_p()
            check(_lib.lib().mlf_region_create(ctypes.byref(self._h)))

    def close(self):
        """free the device buffers now"""
        if self._h:
            _lib.lib().mlf_region_destroy(self._h)
            self._h = ctypes.c_void_p()

    def set_option(self, name, value=None):
        """Tuning option of THIS region (mlf_region_set_option); value None = back to the process default
        (_lib.set_option).  Results never depend on options."""
        check(_lib.lib().mlf_region_set_option(self._h, name.encode(), 0 if value is None else int(value), int(value is None)))

    def get_option(self, name):
        """The value of a tuning option IN FORCE for this region: its own override, else the process default
        (mlf_region_get_option)."""
        v = ctypes.c_longlong(0)
        check(_lib.lib().mlf_region_get_option(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def release(self):
        """hand the handle (and its buffers) to the next region"""
        if self._h:
            _lib.lib().mlf_region_set_option(self._h, None, 0, 1)     # the next owner starts from the process defaults
            if len(DeviceRegion._idle) < DeviceRegion._IDLE_MAX:
                DeviceRegion._idle.append(self._h)
                self._h = ctypes.c_void_p()
            else:
                self.close()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def set(self, live, layer_kind, layer_ctr, layer_T, wrap_shift, ell_center, ell_invcov,
            enlarge, radiussq, use_scan=True, live_space=0):
        """`live` holds region.unormed (live_space=0) or region.u (live_space=1: whitened on the
        device with the proposal kernel, see include/mlfriends_hip.h)."""
        d = len(ell_center)
        self._d = d
        un = None if live is None else f64(live)
        n = 0 if un is None else un.shape[0]
        lc = None if layer_ctr is None else f64(np.broadcast_to(layer_ctr, (d,)))
        if layer_T is None:
            lt = None
        elif layer_kind == 1:
            lt = f64(np.broadcast_to(layer_T, (d,)))
        else:
            lt = f64(layer_T)
        ws = None if wrap_shift is None else f64(wrap_shift)
        self._keep = (un, lc, lt, ws)
        check(_lib.lib().mlf_region_set(self._h, ptr(un), n, d, int(live_space), int(layer_kind), ptr(lc), ptr(lt), ptr(ws),
                                        ptr(f64(ell_center)), ptr(f64(ell_invcov)), float(enlarge),
                                        float(radiussq), int(bool(use_scan))))

    def set_from_device(self, live_dev, n, d, layer_kind, layer_ctr, layer_T, wrap_shift, ell_center, ell_invcov,
                        enlarge, radiussq, use_scan=True, live_space=1, live_amax=None):
        """`set` with the live points already on the device (`live_dev`: a C-contiguous float64 (n, d) DevArray or torch tensor;
        mlf_region_set takes host or device pointers for the point array).  `live_amax`: the largest |u - layer centre|
        coordinate if the caller knows it (otherwise the rows are fetched back once to find it)."""
        if live_amax is not None:
            check(_lib.lib().mlf_region_hint_live_extent(self._h, float(live_amax)))
        self._d = d
        lc = f64(np.broadcast_to(layer_ctr, (d,)))
        lt = f64(np.broadcast_to(layer_T, (d,))) if layer_kind == 1 else f64(layer_T)
        ws = None if wrap_shift is None else f64(wrap_shift)
        self._keep = (live_dev, lc, lt, ws)
        check(_lib.lib().mlf_region_set(self._h, ctypes.c_void_p(live_dev.data_ptr()), n, d, int(live_space), int(layer_kind),
                                        ptr(lc), ptr(lt), ptr(ws), ptr(f64(ell_center)), ptr(f64(ell_invcov)), float(enlarge),
                                        float(radiussq), int(bool(use_scan))))

    def set_thresholds(self, enlarge, radiussq):
        check(_lib.lib().mlf_region_set_thresholds(self._h, float(enlarge), float(radiussq)))

    def set_ellipsoid_center(self, ctr):
        check(_lib.lib().mlf_region_set_ellipsoid_center(self._h, ptr(f64(ctr))))

    def update_point(self, row, unormed_row):
        check(_lib.lib().mlf_region_update_point(self._h, int(row), ptr(f64(unormed_row))))

    def update_points(self, rows, live_rows):
        """Replace the DISTINCT live points `rows` by the rows of `live_rows` in one call."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        live_rows = f64(live_rows)
        assert live_rows.shape[0] == rows.shape[0]
        check(_lib.lib().mlf_region_update_points(self._h, rows.shape[0], ptr(rows), ptr(live_rows)))

    def inside(self, pts):
        pts = f64(pts)
        mask = np.empty(pts.shape[0], dtype=np.uint8)
        if pts.shape[0]:
            check(_lib.lib().mlf_region_inside(self._h, ptr(pts), pts.shape[0], ptr(mask)))
        return mask.view(np.bool_)

    # device-pointer entry points (integers from e.g. torch.Tensor.data_ptr())
    def inside_dev(self, d_pts, npts, d_mask, stream=0):
        check(_lib.lib().mlf_region_inside_dev(self._h, ctypes.c_void_p(d_pts), npts, ctypes.c_void_p(d_mask),
                                               ctypes.c_void_p(stream)))

    def find_nearby_dev(self, d_tpts, npts, d_idx, stream=0):
        check(_lib.lib().mlf_region_find_nearby_dev(self._h, ctypes.c_void_p(d_tpts), npts,
                                                    ctypes.c_void_p(d_idx), ctypes.c_void_p(stream)))

    def set_axes(self, axes_T):
        check(_lib.lib().mlf_region_set_axes(self._h, ptr(f64(axes_T))))

    def set_sampling_data(self, invT, bbox_lo, bbox_hi):
        check(_lib.lib().mlf_region_set_sampling_data(self._h, ptr(f64(invT)), ptr(f64(bbox_lo)), ptr(f64(bbox_hi))))

    def sample(self, method, nsamples, seed, offset, capacity=None):
        """Device-side draw + membership test + compaction.  Returns (accepted rows (k, d), next offset)."""
        d = self._d
        cap = int(nsamples if capacity is None else capacity)
        out = np.empty((cap, d), dtype=np.float64)
        nacc, nxt = ctypes.c_size_t(0), ctypes.c_uint64(0)
        check(_lib.lib().mlf_region_sample(self._h, int(method), int(nsamples), ctypes.c_uint64(int(seed)),
                                           ctypes.c_uint64(int(offset)), ptr(out), cap, ctypes.byref(nacc),
                                           ctypes.byref(nxt)))
        return out[:nacc.value], nxt.value

    def refill(self, method, nsamples, seed, offset, Lmin, tspec, lspec, capacity=None):
        """Device-resident proposal batch: draw + region test + prior transform + likelihood; returns the
        points above `Lmin` as (u, p, L), the number of likelihood evaluations and the next offset."""
        d = self._d
        cap = int(nsamples if capacity is None else capacity)
        u, p, L = np.empty((cap, d)), np.empty((cap, d)), np.empty(cap)
        nev, nkept, nxt = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_uint64(0)
        tkind, ta, tb = tspec
        lkind, aux, sigma = lspec
        check(_lib.lib().mlf_region_refill(self._h, int(method), int(nsamples), ctypes.c_uint64(int(seed)),
                                           ctypes.c_uint64(int(offset)), float(Lmin), int(tkind), float(ta), float(tb),
                                           int(lkind), ptr(None if aux is None else f64(aux)), float(sigma), ptr(u), ptr(p),
                                           ptr(L), cap, ctypes.byref(nev), ctypes.byref(nkept), ctypes.byref(nxt)))
        k = nkept.value
        return u[:k], p[:k], L[:k], nev.value, nxt.value

    def first_index_dev(self, d_pts, npts, d_idx, stream=0):
        check(_lib.lib().mlf_region_first_index_dev(self._h, ctypes.c_void_p(d_pts), npts,
                                                    ctypes.c_void_p(d_idx), ctypes.c_void_p(stream)))

    def inside_dev_timed(self, d_pts, npts, d_mask, stream=0):
        check(_lib.lib().mlf_region_inside_dev_timed(self._h, ctypes.c_void_p(d_pts), npts,
                                                     ctypes.c_void_p(d_mask), ctypes.c_void_p(stream)))

    def timing_collect(self):
        """(ncalls, ms per-proposal stage, ms scan kernel, ms rest of scan stage) since last collect."""
        n, a, b, c = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        check(_lib.lib().mlf_region_timing_collect(self._h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b),
                                                   ctypes.byref(c)))
        return n.value, a.value, b.value, c.value

    def timing_filter_launches(self):
        """(number of k_filter launches, their summed ms) of the timed calls since the last collect"""
        n, ms = ctypes.c_int(0), ctypes.c_double(0)
        check(_lib.lib().mlf_region_timing_filter_launches(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def timing_filter_launch_ms(self, cap=4096):
        """durations (ms) of the k_filter launches of the timed calls since the last collect, in launch order"""
        ms = np.empty(cap)
        n = ctypes.c_int(0)
        check(_lib.lib().mlf_region_timing_filter_launch_ms(self._h, ptr(ms), cap, ctypes.byref(n)))
        return ms[:min(n.value, cap)].copy()

    def filter_info(self, npts):
        """(filter active for this batch size, K columns of the f16 GEMM, number of 32-row live tiles)"""
        act, k, t = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        check(_lib.lib().mlf_region_filter_info(self._h, int(npts), ctypes.byref(act), ctypes.byref(k), ctypes.byref(t)))
        return bool(act.value), k.value, t.value

    def debug_stats(self):
        """Counters of the last filtered batch (see mlf_region_debug_stats in include/mlfriends_hip.h)."""
        out = np.zeros(19, dtype=np.uint64)
        check(_lib.lib().mlf_region_debug_stats(self._h, ptr(out), 19))
        keys = ("ellipsoid_band", None, "uncertain_pairs", "largest_segment", "segments", "second_range_groups",
                "uncertain_queries", "third_range_groups")
        stats = {k: int(v) for k, v in zip(keys, out[:8]) if k}
        stats["range_cuts"] = [int(out[16]), int(out[17])]      # tile cuts of the last min-only batch (second: 0 = two ranges)
        stats["same_quadratic_form"] = int(out[18])             # k_prep_sweep read the ellipsoid form off the whitening chain
        if out[8]:     # k_uncertain, workgroup 0: shader cycles between the stage boundaries of its first set
            st = out[8:16].astype(np.int64)
            stats["uncertain_stage_cycles"] = [int((st[i + 1] - st[i]) & 0xffffffff) for i in range(6) if st[i + 1]]
            stats["stamp7"] = int(st[7])
        return stats

    def fused_stamps(self, block):
        """Stage stamps (shader cycles, differences to the first) that wave 0 of the k_prep_sweep workgroup selected by the
        previous call wrote during the last phased batch; selects workgroup `block` for the next batches (None: off).
        block = 1 000 000 + b / 2 000 000 + b: workgroup b of the second / third range's k_sweep_min launch (entries 8 ... 11 then
        hold the shared 100 MHz clock, see include/mlfriends_hip.h)."""
        out = np.zeros(16, dtype=np.uint64)
        check(_lib.lib().mlf_region_debug_fused_stamps(self._h, -1 if block is None else int(block), ptr(out), 16))
        st = out[:16].astype(np.int64)
        return [int(x - st[0]) if x else None for x in st] if st[0] else None

    def time_inside_dev(self, d_pts, npts, d_mask, stream=0, reps=3):
        tot, scan = ctypes.c_float(0), ctypes.c_float(0)
        check(_lib.lib().mlf_region_time_inside_dev(self._h, ctypes.c_void_p(d_pts), npts, ctypes.c_void_p(d_mask),
                                                    ctypes.c_void_p(stream), int(reps), ctypes.byref(tot),
                                                    ctypes.byref(scan)))
        return tot.value, scan.value


def bench_fp64_valu():
    """Measured non-fused FP64 vector issue rate in TFLOP/s (one flop per v_add_f64/v_mul_f64 lane)."""
    t = ctypes.c_double(0)
    check(_lib.lib().mlf_bench_fp64_valu(ctypes.byref(t)))
    return t.value


def philox_blocks(seed, stream, nblocks):
    """Raw Philox-4x32-10 output blocks of the device generator (known-answer tests)."""
    out = np.empty((int(nblocks), 4), dtype=np.uint32)
    check(_lib.lib().mlf_debug_philox(ctypes.c_uint64(int(seed)), int(stream), int(nblocks), ptr(out)))
    return out
