"""Region classes with the API of the reference's ``MLFriends``, ``RobustEllipsoidRegion``,
``SimpleRegion`` and ``WrappingEllipsoid`` (reference ultranest/mlfriends.pyx:915-1649), built on
the MI355X kernels.

The driver (reference integrator.py, not part of this package) only ever does
``region_class(active_u, transformLayer)`` (:1996, :2079) and then reads / MUTATES attributes
(``u``, ``unormed``, ``maxradiussq``, ``enlarge``, ``ellipsoid_center`` ... :2749-2765, :2827), so
these classes keep plain numpy attributes on the host as the source of truth and mirror them to
the device lazily: before every device call the host arrays are diffed against the last uploaded
snapshot and only what changed is sent (usually one live-point row per iteration).

What runs where:
  GPU   pair distances (bootstrapped radius K4, neighbour scans K1/K2, subtract_nearby K3),
        ellipsoid quadratic forms over batches (H3), affine transform of proposal batches (T1),
        bootstrap moments + enlargement factor
  host  d x d LAPACK (cov inverse, eigh), random draws (the reference's ``np.random`` call
        order is preserved so seeded runs follow the same trajectory)
"""
import weakref

import numpy as np

from . import _lib, kernels
from .layers import host_worker, int_dtype, single_blas_thread

# When True, ``MLFriends.inside`` transforms ellipsoid-passing points on the host with the same
# ``np.dot`` the reference uses (bit-identical t-space points on the same machine) and only the
# neighbour scan runs on the GPU.  Default: fully device-side pipeline.
STRICT_HOST_TRANSFORM = False
# When True, bootstrap ellipsoid moments come from numpy (mean / cov exactly as the reference
# computes them); default: GPU two-pass moments (agree to ~1e-13 relative).
STRICT_HOST_MOMENTS = False


class DeviceRNG(object):
    """Opt-in counter-based random stream for DEVICE-SIDE proposal generation (SURVEY.md 8f row
    f2): Philox-4x32-10 keyed by `seed`; the counter advances by what every draw consumed, so a
    run is reproducible for a given seed.  Assign an instance to ``region.device_rng`` (or pass
    ``device_rng=`` to the harness) and ``sample_from_boundingbox`` /
    ``sample_from_wrapping_ellipsoid`` draw, test and compact on the GPU; only the accepted
    points come back.  The stream is not numpy's MT19937: agreement with the reference is then
    statistical (same distribution), not draw by draw."""

    def __init__(self, seed=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0


def vol_prefactor(n):
    """Volume of the unit n-ball (reference mlfriends.pyx:853-879): recurrence
    V_n = V_{n-2} * 2 pi / n from V_0 = 1, V_1 = 2."""
    n = int(n)
    vol = 1. if n % 2 == 0 else 2.
    for i in range(2 if n % 2 == 0 else 3, n + 1, 2):
        vol *= 2. / i * np.pi
    return vol


def make_eigvals_positive(a, targetprod):
    """Lift (near-)zero eigenvalues of the symmetric matrix `a` so that the product of all
    eigenvalues reaches `targetprod` (reference mlfriends.pyx:389-421)."""
    assert np.isfinite(a).all(), a
    w, v = np.linalg.eigh(a)
    tiny = w < max(1.e-10, 1e-300**(1. / len(a)))
    if np.any(tiny):
        w[tiny] = (targetprod / np.prod(w[~tiny])) ** (1. / tiny.sum())
        a = np.dot(np.dot(v, np.diag(w)), np.linalg.inv(v))
    return a


@single_blas_thread
def bounding_ellipsoid(x, minvol=0.):
    """Centre and (d+2)-inflated sample covariance of the points `x`
    (reference mlfriends.pyx:426-476)."""
    ndim = x.shape[1]
    ctr = np.mean(x, axis=0)
    cov = np.cov(x - ctr, rowvar=0)
    assert np.isfinite(cov).all(), (cov, x)
    if ndim == 1:
        cov = np.atleast_2d(cov)
    cov *= (ndim + 2)
    if minvol > 0:
        cov = make_eigvals_positive(cov, minvol)
    return ctr, cov


_ELLIPSOID_JOBS = weakref.WeakKeyDictionary()      # region -> (future of ellipsoid_parts, minvol, write count of region.u)


def _principal_axes(precision, cov):
    """(axlens, axes, axes_T, inv_axlens, inv_axes) of an ellipsoid matrix and its inverse (reference :1226-1235)."""
    lam, vec = np.linalg.eigh(precision)
    axlens = 1. / np.sqrt(lam)
    axes = np.dot(vec, np.diag(axlens))
    lam2, vec2 = np.linalg.eigh(cov)
    inv_axlens = 1. / np.sqrt(lam2)
    inv_axes = np.dot(vec2, np.diag(inv_axlens))
    return axlens, axes, axes.transpose(), inv_axlens, inv_axes


def _inside_ellipsoid(points, ellipsoid_center, ellipsoid_invcov, square_radius):
    """``(p-c)^T invcov (p-c) <= square_radius`` per row (reference mlfriends.pyx:882-912); the
    quadratic form is evaluated on the GPU in numpy's einsum order."""
    return kernels.inside_ellipsoid(points, ellipsoid_center, ellipsoid_invcov, square_radius)


def _draw_selection(rng, npoints, nbootstraps):
    """(B, N) bootstrap selection masks; ONE ``rng.randint(N, size=N)`` per round, drawn before
    any validity check, exactly like the reference (mlfriends.pyx:1044-1047)."""
    masks = _lib.draw_selection(rng, npoints, nbootstraps)   # numpy's MT19937 stream, compiled
    if masks is not None:
        return masks
    masks = np.zeros((nbootstraps, npoints), dtype=bool)
    for b in range(nbootstraps):
        masks[b, rng.randint(npoints, size=npoints)] = True
    return masks


def _shard_bounds(nitems, rank, world_size):
    """contiguous, balanced [lo, hi) slice (the same split as ``distributed.shard_bounds``)"""
    base, extra = divmod(int(nitems), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _host_masks(masks):
    """the (B, N) selection matrix as a numpy bool array (a device-side broadcast hands over a uint8 torch tensor)"""
    if hasattr(masks, "data_ptr"):
        return masks.cpu().numpy().astype(bool)
    return np.asarray(masks, dtype=bool)


def _select_rounds(masks, use):
    """rows `use` of the (B, N) selection matrix (numpy on the host or a torch tensor on the device)"""
    if use.all():
        return masks
    if hasattr(masks, "data_ptr"):
        import torch
        picked = masks[torch.from_numpy(np.flatnonzero(use)).to(masks.device)]
        # the library copies from this tensor on ITS stream: the index kernel (torch's current stream) has to be done
        torch.cuda.current_stream(masks.device).synchronize()
        return picked
    return masks[use]


@single_blas_thread
def _bootstrap_enlargement(u, masks, minvol):
    """Per-round wrapping-ellipsoid enlargement f_b (reference mlfriends.pyx:1056-1066):
    ellipsoid of the selected points, largest Mahalanobis distance of the left-out ones."""
    nrounds, ndim = len(masks), u.shape[1]
    on_device = hasattr(masks, "data_ptr")        # uint8 torch tensor from a device-side broadcast
    if on_device and (STRICT_HOST_MOMENTS or not (minvol == 0 and ndim <= 64)):
        masks, on_device = masks.cpu().numpy().astype(bool), False
    if STRICT_HOST_MOMENTS:
        ctrs = np.empty((nrounds, ndim))
        covs = np.empty((nrounds, ndim, ndim))
        for b, m in enumerate(masks):
            ctrs[b], covs[b] = bounding_ellipsoid(u[m], minvol=minvol)
    elif minvol == 0 and ndim <= 64:
        # moments, factorisation and the quadratic form in one device call (no LAPACK inverse on the host)
        f = kernels.bootstrap_factor(u, masks, ndim + 2)
        if not np.isfinite(f).all():
            raise np.linalg.LinAlgError("Singular matrix")     # what np.linalg.inv raises in the reference
        return f
    else:
        ctrs, covs = kernels.bootstrap_moments(u, masks)
        assert np.isfinite(covs).all(), (covs, u)
        covs *= (ndim + 2)
        if minvol > 0:
            for b in range(nrounds):
                covs[b] = make_eigvals_positive(covs[b], minvol)
    precisions = np.linalg.inv(covs)      # LinAlgError on a singular round, as in the reference
    return kernels.bootstrap_quadform_max(u, masks, ctrs, precisions)


# ---- write-tracked live points --------------------------------------------------------------------
_IN_PLACE_FUNCTIONS = frozenset(f for f in (getattr(np, n, None) for n in (
    "copyto", "put", "place", "putmask", "put_along_axis", "fill_diagonal")) if f is not None)


class _LiveArray(np.ndarray):
    """``region.u`` as an ndarray that COUNTS its in-place writes.

    The driver replaces one live point per iteration with ``region.u[worst] = u`` (reference
    integrator.py:2753) and the step samplers then call ``region.inside`` with 1-10 points many times
    before the next write (stepsampler.py:296-330, 1060-1071).  Comparing 4000 x 50 doubles with the last
    uploaded snapshot on every one of those calls cost 60 us of a 120 us call; with the counter an
    untouched array is recognised in O(1), and a plain row assignment names the rows to re-send.

    The array owns its memory (``region.u = x`` copies x), so the only ways to write into it are through
    this object and its views, which all share one counter cell:  __setitem__, ufunc ``out=`` (hence
    ``+=`` and friends), the in-place methods, and numpy's in-place functions (np.copyto ...).  What the
    counter cannot see -- writes through ``np.asarray(region.u)`` / ``region.u.view(np.ndarray)`` or a
    ``memoryview`` (``.flat``, ``.ctypes`` and ``.data`` count as a write when they are asked for) -- needs ``region.invalidate_device_state()``; nothing in the reference does that."""

    def __array_finalize__(self, obj):
        self._cell = getattr(obj, "_cell", None)
        self._root = False

    def _touch(self, key=None):
        cell = self._cell
        if cell is None:
            return
        cell[0] += 1
        rows = cell[1]
        if rows is None:
            return
        if len(rows) > 4096:   # nobody asked the device for a long time: stop collecting, diff at the next call
            cell[1] = None
            return
        if self._root:   # a plain row assignment: remember which rows
            if isinstance(key, tuple) and key:
                key = key[0]
            if isinstance(key, (int, np.integer)) and not isinstance(key, (bool, np.bool_)):   # u[True] = x writes everywhere
                rows.append(int(key))
                return
            if isinstance(key, np.ndarray) and key.ndim == 1 and key.dtype.kind in "iu" and key.size <= 16:
                rows.extend(int(k) for k in key)
                return
        cell[1] = None   # anything else: the next device call diffs the whole array

    def __setitem__(self, key, value):
        self._touch(key)
        np.ndarray.__setitem__(self, key, value)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        if out is not None:
            for o in out:
                if isinstance(o, _LiveArray):
                    o._touch()
            kwargs["out"] = tuple(o.view(np.ndarray) if isinstance(o, _LiveArray) else o for o in out)
        args = tuple(x.view(np.ndarray) if isinstance(x, _LiveArray) else x for x in inputs)
        return getattr(ufunc, method)(*args, **kwargs)   # plain ndarrays out: results are not live points

    def __array_function__(self, func, types, args, kwargs):
        # numpy functions that write into one of their arguments: the in-place family (first positional argument, or the
        # keyword numpy gives it: dst / a / arr), and any function handed a live array as `out=` (np.dot, np.take,
        # np.cumsum ... are not ufuncs and do not pass through __array_ufunc__)
        if func in _IN_PLACE_FUNCTIONS:
            target = args[0] if args else next((kwargs[k] for k in ("dst", "a", "arr") if k in kwargs), None)
            if isinstance(target, _LiveArray):
                target._touch()
        out = kwargs.get("out") if kwargs else None
        if out is not None:
            for o in (out if isinstance(out, tuple) else (out,)):
                if isinstance(o, _LiveArray):
                    o._touch()
        return super().__array_function__(func, types, args, kwargs)

    def __reduce__(self):   # pickles as the plain array it holds; a region re-wraps it on assignment
        return np.asarray(self).__reduce__()


def _in_place_method(name):
    base = getattr(np.ndarray, name)

    def method(self, *args, **kwargs):
        self._touch()
        return base(self, *args, **kwargs)
    method.__name__ = name
    method.__doc__ = base.__doc__
    return method


for _name in ("fill", "sort", "partition", "put", "itemset", "setfield", "byteswap", "resize"):
    if hasattr(np.ndarray, _name):
        setattr(_LiveArray, _name, _in_place_method(_name))


def _raw_handle(name):
    """``flat``, ``ctypes`` and ``data`` hand out something that writes without passing through the array
    object: asking for one of them counts as a write (a whole-array diff at the next device call)."""
    base = getattr(np.ndarray, name)

    def getter(self):
        self._touch()
        return base.__get__(self, type(self))

    def setter(self, value):   # only ``flat`` has one (a.flat = x assigns through it)
        self._touch()
        base.__set__(self, value)
    return property(getter, setter, doc=base.__doc__)


for _name in ("flat", "ctypes", "data"):
    setattr(_LiveArray, _name, _raw_handle(_name))


class _LivePoints(object):
    """Mixin: the ``u`` attribute of a region is a private, write-tracked copy (see _LiveArray)."""

    @property
    def u(self):
        return self._u

    @u.setter
    def u(self, value):
        cell = self.__dict__.get("_u_cell")
        if cell is None:
            cell = self.__dict__["_u_cell"] = [0, None]
        if isinstance(value, _LiveArray) and value._cell is cell and value._root:
            self._u = value
            return
        arr = np.array(value, order="C", copy=True).view(_LiveArray)
        arr._cell = cell
        arr._root = True
        cell[0] += 1
        cell[1] = None
        self._u = arr

    # ``sampling_methods`` / ``current_sampling_method`` hold bound methods of the region itself in the reference
    # (mlfriends.pyx:944-950): a reference cycle, so a replaced region -- and its device buffers -- lives until the
    # cyclic collector happens to run.  Here the own methods are kept by NAME and bound on access; anything else a
    # caller assigns is kept as it is.
    def _unbind(self, method):
        return method.__name__ if getattr(method, "__self__", None) is self else method

    def _bind(self, item):
        return getattr(self, item) if isinstance(item, str) else item

    @property
    def sampling_methods(self):
        return [self._bind(m) for m in self._sampling_methods]

    @sampling_methods.setter
    def sampling_methods(self, methods):
        self._sampling_methods = [self._unbind(m) for m in methods]

    @property
    def current_sampling_method(self):
        return self._bind(self._current_sampling_method)

    @current_sampling_method.setter
    def current_sampling_method(self, method):
        self._current_sampling_method = self._unbind(method)

    def invalidate_device_state(self):
        """Forget what the device holds: the next device call re-sends the region.  Only needed after
        writing into ``region.u`` (or the ellipsoid / layer arrays) behind numpy's back."""
        self._dev.fast_key = None
        cell = self.__dict__.get("_u_cell")
        if cell is not None:
            cell[0] += 1
            cell[1] = None


class _DeviceState(object):
    """Lazy mirror of a region's host attributes on the GPU (see module docstring).

    The small constants (layer, ellipsoid matrix) are compared BY VALUE with the last upload;
    the live points are diffed row-wise so that the driver's in-place replacement of one live
    point per iteration costs one row upload."""

    def __init__(self):
        self.handle = None
        self.consts = None
        self.live = None
        self.ell_center = None
        self.thresholds = None
        self.axes_T = None
        self.sampling_data = None
        self.fast_key = None      # identities + write counter of everything the device state was built from

    @staticmethod
    def _key(region, use_scan):
        """What the device state depends on, by IDENTITY (arrays) and value (scalars): equal keys mean nothing
        was re-assigned and region.u was not written to since the last sync."""
        layer = region.transformLayer if use_scan else None
        ld = layer.__dict__ if layer is not None else {}
        cell = region.__dict__.get("_u_cell")
        live = region.__dict__.get("_u")
        tracked = cell is not None and isinstance(live, _LiveArray) and live._cell is cell   # not e.g. after unpickling
        return (use_scan, live, cell[0] if tracked else -1, layer,
                ld.get("T"), ld.get("ctr"), ld.get("mean"), ld.get("std"), ld.get("wrap_cuts"),
                region.ellipsoid_center, region.ellipsoid_invcov, region.enlarge, region.maxradiussq if use_scan else None)

    def _key_unchanged(self, key):
        old = self.fast_key
        if old is None or len(old) != len(key):
            return False
        for a, b in zip(old, key):
            if a is b:
                continue
            if isinstance(a, (float, int)) and isinstance(b, (float, int)) and not isinstance(a, bool) and a == b:
                continue
            return False
        return True

    def refill(self, region, use_scan, method, nsamples, Lmin, tspec, lspec):
        """Device-resident proposal batch (draw, region test, prior transform, likelihood, threshold)."""
        handle = self._prepare_sampling(region, use_scan, method)
        rng = region.device_rng
        u, p, L, nev, rng.offset = handle.refill(method, nsamples, rng.seed, rng.offset, Lmin, tspec, lspec)
        return u, p, L, nev

    def sample(self, region, use_scan, method, nsamples):
        """Device-side draw + membership + compaction with the region's ``device_rng``."""
        handle = self._prepare_sampling(region, use_scan, method)
        rng = region.device_rng
        pts, rng.offset = handle.sample(method, nsamples, rng.seed, rng.offset)
        return pts

    def _prepare_sampling(self, region, use_scan, method):
        handle = self.sync(region, use_scan)
        if method == 1:
            axes_T = np.asarray(region.ellipsoid_axes_T, dtype=float)
            if self.axes_T is None or not np.array_equal(self.axes_T, axes_T):
                handle.set_axes(axes_T)
                self.axes_T = axes_T.copy()
        if method >= 2:
            layer = region.transformLayer
            data = (np.asarray(layer.invT, dtype=float), np.asarray(region.bbox_lo, dtype=float),
                    np.asarray(region.bbox_hi, dtype=float))
            if self.sampling_data is None or not all(np.array_equal(a, b) for a, b in zip(self.sampling_data, data)):
                handle.set_sampling_data(*data)
                self.sampling_data = tuple(a.copy() for a in data)
        return handle

    @staticmethod
    def _same(a, b):
        if len(a) != len(b):
            return False
        for x, y in zip(a, b):
            if x is None or y is None or np.isscalar(x) or np.isscalar(y):
                if not (x is None and y is None) and not (np.isscalar(x) and np.isscalar(y) and x == y):
                    return False
            elif x.shape != y.shape or not np.array_equal(x, y, equal_nan=True):
                return False
        return True

    def adopt(self, region, use_scan):
        """The handle has just been set from data that never left the device (device_rebuild): record what a full
        `_sync_slow` would have recorded, so that the next call finds the device state current."""
        ndim = region.u.shape[1]
        kind, lctr, lmat = region.transformLayer.device_params(ndim)
        shift = region.transformLayer.wrap_shift_vector(ndim)
        consts = (kind, lctr, lmat, shift, np.asarray(region.ellipsoid_invcov), int(use_scan), len(region.u))
        self.consts = tuple(None if c is None else (c if np.isscalar(c) else np.array(c, dtype=float)) for c in consts)
        self.live = None      # region.u has just been assigned: its writes are counted (see _sync_slow)
        self.ell_center = np.array(region.ellipsoid_center, dtype=float)
        self.thresholds = (float(region.enlarge), float(region.maxradiussq))
        cell = region.__dict__.get("_u_cell")
        if cell is not None:
            cell[1] = []
        self.fast_key = self._key(region, use_scan)

    def sync(self, region, use_scan):
        key = self._key(region, use_scan)
        if self.handle is not None and key[2] >= 0 and self._key_unchanged(key):
            return self.handle
        handle = self._sync_slow(region, use_scan, key)
        cell = region.__dict__.get("_u_cell")
        if cell is not None:
            cell[1] = []      # from here on plain row assignments are remembered
        self.fast_key = key
        return handle

    def _sync_slow(self, region, use_scan, key):
        ndim = region.u.shape[1]
        if use_scan and region.maxradiussq is None:
            raise TypeError("region.maxradiussq is None: bootstrap the radius before testing membership")
        r2 = float(region.maxradiussq) if use_scan else 1e300
        nlive = len(region.u) if use_scan else 0
        thresholds = (float(region.enlarge), r2)
        if self.handle is None:
            self.handle = kernels.DeviceRegion()
        # layer, ellipsoid matrix and live-point array are the objects of the last sync: their values need no comparing
        old = self.fast_key
        untouched = (old is not None and self.consts is not None and old[0] == key[0] and key[2] >= 0
                     and all(old[i] is key[i] for i in (1, 3, 4, 5, 6, 7, 8, 10)))
        def current_consts():
            if use_scan:
                kind, lctr, lmat = region.transformLayer.device_params(ndim)
                shift = region.transformLayer.wrap_shift_vector(ndim)
            else:
                kind, lctr, lmat, shift = 0, None, None, None
            return (kind, lctr, lmat, shift, np.asarray(region.ellipsoid_invcov), int(use_scan), nlive)

        consts = None if untouched else current_consts()
        full = not untouched and (self.consts is None or not self._same(consts, self.consts))
        changed = ()
        cell = region.__dict__.get("_u_cell")
        rows = cell[1] if cell is not None else None
        distinct = None
        if (not full and use_scan and rows is not None and key[2] >= 0 and self.fast_key is not None
                and self.fast_key[1] is key[1]):
            distinct = set(r % nlive for r in rows)      # a row written twice counts once
        known = distinct is not None and len(distinct) <= max(8, nlive // 8)
        if known:      # only plain row assignments since the last sync: no diff
            changed = sorted(distinct)
        elif not full and use_scan and self.live is None:
            # writes the counter could not name, and no host snapshot to diff against (a snapshot of a TRACKED array cost a
            # 1.6 MB copy per rebuild at N = 4000, d = 50 for a case nothing in the reference produces): send everything
            full = True
        elif not full and use_scan:
            u_now = np.asarray(region.u)
            if u_now.dtype == np.float64 and u_now.flags.c_contiguous and u_now.shape == self.live.shape:
                limit = max(8, nlive // 8)
                changed, nchanged = _lib.changed_rows(self.live, u_now, capacity=limit)
                full = nchanged > limit
            else:
                changed = np.flatnonzero((self.live != u_now).any(axis=1))
                full = len(changed) > max(8, nlive // 8)
        if full:
            if consts is None:
                consts = current_consts()
            kind, lctr, lmat, shift = consts[:4]
            # the device whitens region.u itself (live_space=1): live points and proposals then go
            # through the same arithmetic, so a live point is at distance exactly 0 from itself
            self.handle.set(region.u if use_scan else None, kind, lctr, lmat, shift,
                            region.ellipsoid_center, region.ellipsoid_invcov, thresholds[0], thresholds[1],
                            use_scan=use_scan, live_space=1)
            self.consts = tuple(None if c is None else (c if np.isscalar(c) else np.array(c, dtype=float)) for c in consts)
            # host snapshot only for an array whose writes are NOT counted (e.g. a region that came out of a pickle):
            # there every call diffs against it; a counted array names its rows or is sent again
            self.live = np.array(region.u, dtype=float) if (use_scan and key[2] < 0) else None
            self.ell_center = np.array(region.ellipsoid_center, dtype=float)
            self.thresholds = thresholds
            return self.handle
        if len(changed):
            rows = np.asarray(changed, dtype=np.int64)
            fresh = np.asarray(region.u)[rows]
            self.handle.update_points(rows, fresh)
            if self.live is not None:
                self.live[rows] = fresh
        if not np.array_equal(self.ell_center, region.ellipsoid_center):
            self.handle.set_ellipsoid_center(region.ellipsoid_center)
            self.ell_center = np.array(region.ellipsoid_center, dtype=float)
        if thresholds != self.thresholds:
            self.handle.set_thresholds(*thresholds)
            self.thresholds = thresholds
        return self.handle


class MLFriends(_LivePoints):
    """MLFriends region: the union of balls of radius sqrt(maxradiussq) around the whitened
    live points, intersected with a wrapping ellipsoid (reference mlfriends.pyx:915-1257)."""

    def __init__(self, u, transformLayer):
        plain = np.asarray(u)
        if not (plain.size == 0 or (plain.min() > 0 and plain.max() < 1)):   # NaN fails both comparisons, as in the reference's test
            ok = np.logical_and(u > 0, u < 1)
            raise ValueError("not all u values are between 0 and 1: %s" % u[~ok.all(axis=1)])
        self.u = u
        self.enlarge = None
        self.device_rng = None
        self._dev = _DeviceState()
        self.set_transformLayer(transformLayer)
        self.sampling_methods = [
            self.sample_from_transformed_boundingbox,
            self.sample_from_boundingbox,
            self.sample_from_points,
            self.sample_from_wrapping_ellipsoid,
        ]
        self.current_sampling_method = self.sample_from_boundingbox
        self.vol_prefactor = vol_prefactor(self.u.shape[1])

    # ---- geometry state ------------------------------------------------------------------
    def set_transformLayer(self, transformLayer):
        """Adopt a new whitening layer; the radius becomes invalid (reference :972-986)."""
        self.transformLayer = transformLayer
        self.unormed = self.transformLayer.transform(self.u)
        self.bbox_lo = self.unormed.min(axis=0)
        self.bbox_hi = self.unormed.max(axis=0)
        # every element finite <=> every column's extremes finite (min / max propagate NaN; an infinity is an extreme)
        assert np.isfinite(self.bbox_lo).all() and np.isfinite(self.bbox_hi).all(), (self.unormed, self.u)
        self.maxradiussq = None

    def estimate_volume(self):
        """log-volume scale of one ball in u-space: logvolscale + d*log(r) (reference :953-970)."""
        r = self.maxradiussq**0.5
        ndim = self.u.shape[1]
        return self.transformLayer.logvolscale + np.log(r) * ndim

    @staticmethod
    def ellipsoid_parts(u, minvol=0.0, errstate=None):
        """What `create_ellipsoid` derives from the live points ALONE (centre, inflated covariance, its inverse, the axes):
        the numpy / LAPACK calls of reference :1213-1237 in their order.  `_start_ellipsoid_parts` runs this on the worker
        thread while the GPU bootstraps the radius; `errstate` = the caller's ``np.geterr()``, which numpy keeps per
        thread."""
        with np.errstate(**(errstate or np.geterr())):
            ctr, cov = bounding_ellipsoid(u, minvol=minvol)
            precision = np.linalg.inv(cov)
            return ctr, cov, precision, _principal_axes(precision, cov)

    def _write_count(self):
        cell = self.__dict__.get("_u_cell")
        return (id(self.__dict__.get("_u")), cell[0] if cell is not None else None)

    def _start_ellipsoid_parts(self, minvol):
        """Every caller of the bootstrap (the reference's driver, `integrator.py:375-415` then `:2098`; the harness) calls
        `create_ellipsoid` next, and that call's numpy / LAPACK work -- `cov`, `inv`, two `eigh`: 0.7 ms at N = 4000,
        d = 50 -- depends on the live points alone.  It starts here, on the worker thread, while the calling thread waits
        for the GPU's bootstrap passes inside the C-ABI calls; `create_ellipsoid` takes the result if the live points
        have not been written to since and `minvol` is the same.  Same calls, same inputs, same results."""
        if type(self).ellipsoid_parts is not MLFriends.ellipsoid_parts:
            return
        stamp = self._write_count()
        started = _ELLIPSOID_JOBS.get(self)
        if started is not None and started[1] == minvol and started[2] == stamp:
            return      # on its way already
        # the worker's cov / inv / eigh under the same one-thread BLAS limit as the caller's scopes (counted, layers.single_blas_thread)
        job = host_worker().submit(single_blas_thread(self.ellipsoid_parts), self.u, minvol, np.geterr())
        _ELLIPSOID_JOBS[self] = (job, minvol, stamp)

    @single_blas_thread
    def create_ellipsoid(self, minvol=0.0):
        """Wrapping ellipsoid of all live points and its principal axes (reference :1213-1237)."""
        assert self.enlarge is not None
        started = _ELLIPSOID_JOBS.pop(self, None)
        if started is not None and started[1] == minvol and started[2] == self._write_count():
            ctr, cov, precision, axes = started[0].result()      # raises here what the calls raised there
        else:
            ctr, cov, precision, axes = self.ellipsoid_parts(self.u, minvol)
        self.ellipsoid_center = ctr
        self.ellipsoid_invcov = precision
        self.ellipsoid_cov = cov
        (self.ellipsoid_axlens, self.ellipsoid_axes, self.ellipsoid_axes_T, self.ellipsoid_inv_axlens,
         self.ellipsoid_inv_axes) = axes

    def _set_axes(self, precision, cov):
        (self.ellipsoid_axlens, self.ellipsoid_axes, self.ellipsoid_axes_T, self.ellipsoid_inv_axlens,
         self.ellipsoid_inv_axes) = _principal_axes(precision, cov)

    # ---- bootstrapping ---------------------------------------------------------------------
    def compute_maxradiussq(self, nbootstraps=50):
        """Bootstrapped MLFriends radius only, global ``np.random`` stream, no all/none guard
        (reference :988-1015).  All rounds run in one GPU pass."""
        npoints = len(self.u)
        masks = _draw_selection(np.random, npoints, nbootstraps)
        r2, _ = kernels.maxradiussq_bootstrap(self.unormed, masks)
        maxd = float(max(0.0, r2.max())) if len(r2) else 0.0
        assert maxd > 0, (maxd, self.u)
        return maxd

    def compute_enlargement(self, nbootstraps=50, minvol=0., rng=np.random):
        """(maxradiussq, enlarge) from `nbootstraps` leave-out rounds (reference :1017-1070).
        Rounds that select all or no points contribute nothing (:1048)."""
        assert np.isfinite(self.unormed).all(), self.unormed
        self._start_ellipsoid_parts(minvol)
        masks = _draw_selection(rng, len(self.u), nbootstraps)
        maxd, maxf = self.enlargement_from_masks(masks, minvol=minvol)
        assert maxd > 0, (maxd, self.u, self.unormed)
        assert maxf > 0, (maxf, self.u, self.unormed)
        return maxd, maxf

    def enlargement_from_masks(self, masks, minvol=0.):
        """(max radius^2, max enlargement) over the given pre-drawn (B, N) selection masks; 0.0
        for an empty set.  This is the unit of work that ``ultranest_amd.distributed`` shards
        over GPUs (max is exact and order independent, so any sharding gives identical bits)."""
        maxd, maxf = 0.0, 0.0
        if len(masks) == 0:
            return maxd, maxf
        self._start_ellipsoid_parts(minvol)      # host LAPACK of the create_ellipsoid that follows, behind the GPU passes below
        r2, skipped = kernels.maxradiussq_bootstrap(self.unormed, masks)
        use = ~skipped
        if use.any():
            maxd = float(r2[use].max())
            f = _bootstrap_enlargement(self.u, _select_rounds(masks, use), minvol)
            assert np.isfinite(f).all(), (f, self.unormed)
            if not (f > 0).all():
                raise np.linalg.LinAlgError("Distances are not positive")
            maxf = float(f.max())
        return maxd, maxf

    def enlargement_share(self, masks, rank, size, minvol=0.):
        """Rank `rank`'s share of ``enlargement_from_masks(masks)`` in a group of `size` GPUs; the element-wise maximum of
        the shares over the ranks is that call's result, bit for bit (max is exact and order independent, the binary32
        narrowing of the radius monotone).

        The RADIUS is sharded by ROW BLOCKS: every rank runs all rounds and every live point i, but only its own 64-row
        blocks of left-out points j -- 1/size of the pair distances.  (Sharding the rounds divides nothing there: the
        kernel computes a distance once for all 32 rounds of a pass, so a rank with 4 of 30 rounds would still do all of
        them.)  The ellipsoid factor f is sharded by ROUNDS (moments, Cholesky factor and substitution are per round)."""
        nrounds, npoints = len(masks), len(self.u)
        if nrounds == 0:
            return 0.0, 0.0
        self._start_ellipsoid_parts(minvol)
        nblocks = (npoints + 63) // 64
        blo, bhi = _shard_bounds(nblocks, rank, size)
        rows = (min(blo * 64, npoints), min(bhi * 64, npoints))
        r2, skipped = kernels.maxradiussq_bootstrap(self.unormed, masks, rows=rows)
        use = ~skipped
        maxd = float(r2[use].max()) if use.any() else 0.0
        lo, hi = _shard_bounds(nrounds, rank, size)
        mine = np.zeros(nrounds, dtype=bool)
        mine[lo:hi] = use[lo:hi]
        maxf = 0.0
        if mine.any():
            f = _bootstrap_enlargement(self.u, _select_rounds(masks, mine), minvol)
            assert np.isfinite(f).all(), (f, self.unormed)
            if not (f > 0).all():
                raise np.linalg.LinAlgError("Distances are not positive")
            maxf = float(f.max())
        return maxd, maxf

    # ---- membership ------------------------------------------------------------------------
    def inside_ellipsoid(self, u):
        return _inside_ellipsoid(u, self.ellipsoid_center, self.ellipsoid_invcov, self.enlarge)

    def inside(self, pts):
        """True where `pts` lie in the wrapping ellipsoid AND within the radius of some live
        point (reference :1186-1211).  One device pipeline: quadratic form -> whitening ->
        neighbour scan gated on the ellipsoid result."""
        pts = np.asarray(pts)
        if STRICT_HOST_TRANSFORM:
            mask = self.inside_ellipsoid(pts)
            if mask.any():
                tpts = self.transformLayer.transform(pts[mask, :])
                idnearby = np.empty(len(tpts), dtype=int_dtype)
                kernels.find_nearby(self.unormed, tpts, self.maxradiussq, idnearby)
                mask[mask] = idnearby >= 0
            return mask
        return self._dev.sync(self, True).inside(pts)

    def compute_mean_pair_distance(self):
        return kernels.compute_mean_pair_distance(self.unormed, self.transformLayer.clusterids)

    # ---- proposal generators (np.random call order = reference) -------
    def _near_live_points(self, tpts):
        idnearby = np.empty(len(tpts), dtype=int_dtype)
        kernels.find_nearby(self.unormed, tpts, self.maxradiussq, idnearby)
        return idnearby >= 0

    def sample_from_points(self, nsamples=100):
        """Pick live points at random, draw uniformly in their balls, thin by 1/multiplicity
        (reference :1072-1094)."""
        if self._device_tspace():
            return self._dev.sample(self, True, 3, nsamples)
        npoints, ndim = self.u.shape
        which = np.random.randint(npoints, size=nsamples)
        direction = np.random.normal(size=(nsamples, ndim))
        direction *= (np.random.uniform(size=nsamples)**(1. / ndim) / np.linalg.norm(direction, axis=1)).reshape((-1, 1))
        tpts = self.unormed[which, :] + direction * self.maxradiussq**0.5
        multiplicity = np.empty(nsamples, dtype=int_dtype)
        kernels.count_nearby(self.unormed, tpts, self.maxradiussq, multiplicity)
        keep = np.random.uniform(high=multiplicity) < 1
        w = self.transformLayer.untransform(tpts[keep, :])
        ok = np.logical_and(w > 0, w < 1).all(axis=1)
        ok[ok] = self.inside_ellipsoid(w[ok])
        return w[ok, :]

    def sample_from_boundingbox(self, nsamples=100):
        """Uniform in the unit cube, filtered by the region (reference :1096-1112)."""
        if self.device_rng is not None:
            return self._dev.sample(self, True, 0, nsamples)
        ndim = self.u.shape[1]
        u = np.random.uniform(size=(nsamples, ndim))
        return u[self.inside(u), :]

    def sample_from_transformed_boundingbox(self, nsamples=100):
        """Uniform in the padded t-space bounding box (reference :1114-1133)."""
        if self._device_tspace():
            return self._dev.sample(self, True, 2, nsamples)
        ndim = self.u.shape[1]
        pad = self.maxradiussq**0.5
        tpts = np.random.uniform(self.bbox_lo - pad, self.bbox_hi + pad, size=(nsamples, ndim))
        w = self.transformLayer.untransform(tpts[self._near_live_points(tpts), :])
        ok = np.logical_and(w > 0, w < 1).all(axis=1)
        ok[ok] = self.inside_ellipsoid(w[ok])
        return w[ok, :]

    def _device_tspace(self):
        """Device-side t-space draws need the Philox stream and an affine layer (ctr, T, invT)."""
        return self.device_rng is not None and np.ndim(getattr(self.transformLayer, "invT", 1)) == 2

    def _draw_in_ellipsoid(self, nsamples):
        ndim = self.u.shape[1]
        z = np.random.normal(size=(nsamples, ndim))
        norm2 = (z**2).sum(axis=1)
        assert (norm2 > 0).all(), norm2
        z /= (norm2**0.5).reshape((nsamples, 1))
        assert self.enlarge > 0, self.enlarge
        z = z * self.enlarge**0.5 * np.random.uniform(size=(nsamples, 1))**(1. / ndim)
        w = self.ellipsoid_center + np.dot(z, self.ellipsoid_axes_T)
        return w[np.logical_and(w > 0, w < 1).all(axis=1), :]

    def sample_from_wrapping_ellipsoid(self, nsamples=100):
        """Uniform in the wrapping ellipsoid, filtered by cube and neighbour scan
        (reference :1135-1160)."""
        if self.device_rng is not None:
            return self._dev.sample(self, True, 1, nsamples)
        w = self._draw_in_ellipsoid(nsamples)
        return w[self._near_live_points(self.transformLayer.transform(w)), :]

    def sample(self, nsamples=100):
        """Accepted draws of the current method; an empty batch re-rolls the method
        (reference :1162-1184)."""
        samples = self.current_sampling_method(nsamples=nsamples)
        if len(samples) == 0:
            self.current_sampling_method = self.sampling_methods[np.random.randint(len(self.sampling_methods))]
        return samples


    _DEVICE_METHOD = dict(sample_from_boundingbox=0, sample_from_wrapping_ellipsoid=1,
                          sample_from_transformed_boundingbox=2, sample_from_points=3)

    def refill(self, nsamples, Lmin, transform, loglike):
        """One proposal batch of the driver's ``_refill_samples`` (reference integrator.py:1773-1837)
        without leaving the device: draw with the current sampling method, region test, prior
        transform, likelihood, and only the points with L > Lmin come back as ``(u, p, L, nc)``.
        Needs ``device_rng`` and ``device_spec`` on both callbacks (ultranest_amd.likelihoods);
        returns None if that does not hold (the caller then uses sample() + callbacks)."""
        tspec, lspec = getattr(transform, "device_spec", None), getattr(loglike, "device_spec", None)
        method = self._DEVICE_METHOD.get(getattr(self.current_sampling_method, "__name__", ""), None)
        if self.device_rng is None or tspec is None or lspec is None or method is None:
            return None
        if method >= 2 and not self._device_tspace():
            return None
        u, p, L, nc = self._dev.refill(self, self._uses_scan(), method, nsamples, Lmin, tspec, lspec)
        if nc == 0:   # the region accepted nothing: re-roll the method like sample() does (:1180-1183)
            self.current_sampling_method = self.sampling_methods[np.random.randint(len(self.sampling_methods))]
        return u, p, L, nc

    def _uses_scan(self):
        return True


class RobustEllipsoidRegion(MLFriends):
    """Single bootstrapped ellipsoid; no neighbour scan (reference mlfriends.pyx:1260-1457)."""

    def __init__(self, u, transformLayer):
        MLFriends.__init__(self, u, transformLayer)
        self.sampling_methods = [
            self.sample_from_boundingbox,
            self.sample_from_wrapping_ellipsoid,
        ]
        self.current_sampling_method = self.sample_from_boundingbox

    def _uses_scan(self):
        return False

    def sample_from_boundingbox(self, nsamples=100):
        if self.device_rng is not None:
            return self._dev.sample(self, False, 0, nsamples)
        ndim = self.u.shape[1]
        u = np.random.uniform(size=(nsamples, ndim))
        return u[self.inside_ellipsoid(u), :]

    def sample_from_transformed_boundingbox(self, nsamples=100):
        # pads by maxradiussq (not its root) like the reference (:1319); not in sampling_methods
        ndim = self.u.shape[1]
        tpts = np.random.uniform(self.bbox_lo - self.maxradiussq, self.bbox_hi + self.maxradiussq, size=(nsamples, ndim))
        w = self.transformLayer.untransform(tpts)
        ok = np.logical_and(w > 0, w < 1).all(axis=1)
        ok[ok] = self.inside_ellipsoid(w[ok])
        return w[ok, :]

    def sample_from_wrapping_ellipsoid(self, nsamples=100):
        if self.device_rng is not None:
            return self._dev.sample(self, False, 1, nsamples)
        return self._draw_in_ellipsoid(nsamples)

    def inside(self, pts):
        return self.inside_ellipsoid(pts)

    def inside_ellipsoid(self, u):
        return self._dev.sync(self, False).inside(np.asarray(u))

    def _selected_stats(self, sel):
        return bounding_ellipsoid(self.u[sel, :])

    def compute_enlargement(self, nbootstraps=50, minvol=0., rng=np.random):
        """Bootstrapped ellipsoid enlargement; the radius is reported as 1e300
        (reference :1392-1440)."""
        npoints, ndim = self.u.shape
        if npoints < ndim + 1:
            raise FloatingPointError('not enough live points to compute covariance')
        assert np.isfinite(self.unormed).all(), self.unormed
        masks = _draw_selection(rng, npoints, nbootstraps)
        maxd, maxf = self.enlargement_from_masks(masks)
        assert maxf > 0, (maxf, self.u, self.unormed)
        return maxd, maxf

    def enlargement_from_masks(self, masks, minvol=0.):
        if len(masks) == 0:
            return 1e300, 0.0
        f = _bootstrap_enlargement(self.u, masks, 0.)
        assert np.isfinite(f).all(), (f, self.unormed)
        if not (f > 0).all():
            raise np.linalg.LinAlgError("Distances are not positive")
        return 1e300, float(f.max())

    def enlargement_share(self, masks, rank, size, minvol=0.):
        """Rank `rank`'s share in a group of `size`: a contiguous shard of the ROUNDS through this class's own
        ``enlargement_from_masks`` (the inherited MLFriends share would apply the friends radius and the
        full-covariance factor to the ellipsoid-only classes).  The element-wise maximum over the ranks is the one-process
        result bit for bit: the radius is the constant 1e300 on every rank that holds a round, f a maximum over rounds."""
        lo, hi = _shard_bounds(len(masks), rank, size)
        if hi <= lo:
            return 0.0, 0.0
        return self.enlargement_from_masks(_host_masks(masks)[lo:hi], minvol=minvol)

    def estimate_volume(self):
        """log-volume of the ellipsoid (reference :1442-1457)."""
        ndim = len(self.ellipsoid_cov)
        sign, logvol = np.linalg.slogdet(self.ellipsoid_cov)
        if sign > 0:
            return logvol + ndim * np.log(self.enlarge)
        return -1e300


class SimpleRegion(RobustEllipsoidRegion):
    """Axis-aligned ellipsoid (reference mlfriends.pyx:1460-1548)."""

    def create_ellipsoid(self, minvol=0.0):
        assert self.enlarge is not None
        var = np.var(self.u, axis=0)
        self.ellipsoid_center = np.mean(self.u, axis=0)
        self.ellipsoid_invcov = np.diag(1. / var)
        self.ellipsoid_cov = np.diag(var)
        self._set_axes(self.ellipsoid_invcov, self.ellipsoid_cov)

    def compute_enlargement(self, nbootstraps=50, minvol=0., rng=np.random):
        """Per-axis variant.  NOTE the reference sums the normalised squared offsets over the
        POINTS (axis 0) and maximises over dimensions (:1540); reproduced as is."""
        npoints, ndim = self.u.shape
        assert np.isfinite(self.u).all(), self.u
        assert np.isfinite(self.unormed).all(), self.unormed
        if npoints < ndim + 1:
            raise FloatingPointError('not enough live points to compute variance')
        maxd, maxf = self.enlargement_from_masks(_draw_selection(rng, npoints, nbootstraps))
        assert maxf > 0, (maxf, self.u, self.unormed)
        return maxd, maxf

    def enlargement_from_masks(self, masks, minvol=0.):
        maxf = 0.0
        for sel in _host_masks(masks):
            ctr = np.mean(self.u[sel, :], axis=0)
            var = np.var(self.u[sel, :], axis=0)
            f = np.sum((self.u[~sel, :] - ctr.reshape((1, -1)))**2 / var, axis=0).max()
            assert np.isfinite(f), (self.u, ctr, var, self.unormed, f)
            if not f > 0:
                raise np.linalg.LinAlgError("Distances are not positive")
            maxf = max(maxf, f)
        return 1e300, maxf


class WrappingEllipsoid(object):
    """Bootstrapped ellipsoid around points in the user's parameter space, applied to every
    proposal batch by the driver (reference mlfriends.pyx:1551-1649, integrator.py:1794).
    Dimensions in which all points coincide are checked for equality instead."""

    def __init__(self, u):
        self.u = u
        self.variable_dims = np.std(self.u, axis=0) > 0
        if self.variable_dims.all():
            self.variable_dims = Ellipsis
        self.enlarge = None

    def compute_enlargement(self, nbootstraps=50, rng=np.random):
        masks = _draw_selection(rng, len(self.u), nbootstraps)
        maxf = self.enlargement_from_masks(masks)
        assert maxf > 0, (maxf, self.u)
        return maxf

    def enlargement_from_masks(self, masks):
        if len(masks) == 0:
            return 0.0
        v = np.ascontiguousarray(self.u[:, self.variable_dims])
        f = _bootstrap_enlargement(v, masks, 0.)
        if not (f > 0).all():
            raise np.linalg.LinAlgError("Distances are not positive")
        return float(f.max())

    def create_ellipsoid(self, minvol=0.0):
        assert self.enlarge is not None
        ctr, cov = bounding_ellipsoid(self.u[:, self.variable_dims], minvol=minvol)
        precision = np.linalg.inv(cov)
        self.ellipsoid_center = ctr
        self.ellipsoid_invcov = precision
        self.ellipsoid_cov = cov
        lam, vec = np.linalg.eigh(precision)
        self.ellipsoid_axlens = 1. / np.sqrt(lam)
        self.ellipsoid_axes = np.dot(vec, np.diag(self.ellipsoid_axlens))

    def update_center(self, ctr):
        if self.variable_dims is Ellipsis:
            self.ellipsoid_center = ctr
        else:
            self.ellipsoid_center = ctr[self.variable_dims]

    def inside(self, u):
        u = np.asarray(u)
        inside_variable = _inside_ellipsoid(u[:, self.variable_dims], self.ellipsoid_center,
                                            self.ellipsoid_invcov, self.enlarge)
        if self.variable_dims is Ellipsis:
            return inside_variable
        inside_fixed = np.all(self.u[0, ~self.variable_dims] == u[:, ~self.variable_dims], axis=1)
        return np.logical_and(inside_fixed, inside_variable)
