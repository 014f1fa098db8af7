"""The write counter behind ``region.u`` (regions._LiveArray): every way numpy offers to write into the array must
bump it -- the device copy of the live points is re-sent only when it moved -- and reads must not."""
import pickle

import numpy as np
import pytest

from ultranest_amd import regions


class _Holder(regions._LivePoints):
    def __init__(self, u):
        self._dev = regions._DeviceState()
        self.u = u


def _gen(h):
    return h.__dict__["_u_cell"][0]


def test_assignment_copies_and_wraps():
    src = np.random.RandomState(1).uniform(size=(20, 3))
    h = _Holder(src)
    assert isinstance(h.u, regions._LiveArray) and isinstance(h.u, np.ndarray)
    assert np.array_equal(h.u, src) and not np.shares_memory(h.u, src)
    g = _gen(h)
    src[0] = 7.0                      # the caller's array is not the region's
    assert h.u[0, 0] != 7.0 and _gen(h) == g
    same = h.u
    h.u = same                        # re-assigning the region's own array keeps it
    assert h.u is same and _gen(h) == g
    h.u = src                         # a new array: new copy, counter moves
    assert _gen(h) > g and h.u is not same


def test_every_in_place_write_is_counted_and_reads_are_not():
    h = _Holder(np.random.RandomState(2).uniform(size=(30, 4)))
    u = h.u
    g = _gen(h)
    # reads
    _ = u[3], u[3, 1], u[2:5], u.mean(axis=0), np.std(u, axis=0), u.T @ u, (u > 0.5).sum(), u * 2, np.asarray(u).sum()
    _ = u.copy(), u.astype(np.float32), u.reshape(-1)[:5].tolist(), np.concatenate((u, u)), u[np.array([1, 2])]
    assert _gen(h) == g
    writes = [
        lambda: u.__setitem__(3, 0.5),
        lambda: u.__setitem__((slice(None), 1), 0.25),
        lambda: u[4].__setitem__(slice(None), 0.125),        # through a row view
        lambda: u[:, 2][5:7].__setitem__(0, 0.75),           # through a view of a view
        lambda: u.__iadd__(0.0),
        lambda: np.add(u, 0.0, out=u),
        lambda: np.clip(u, 0.1, 0.9, out=u),
        lambda: np.copyto(u, u * 1.0),
        lambda: np.put(u, [0, 1], [0.3, 0.4]),
        lambda: np.putmask(u, u > 2, 0.0),
        lambda: u.fill(0.5),
        lambda: u.sort(axis=0),
        lambda: u.T.__setitem__(0, 0.2),
        lambda: u.reshape(-1).__setitem__(7, 0.9),
        lambda: u.flat.__setitem__(11, 0.6),                 # the raw handles count when they are asked for
        lambda: setattr(u, "flat", 0.35),
        lambda: u[2:4].flat.__setitem__(0, 0.65),
        lambda: u.ctypes.data,
        lambda: u.data,
        # writes that round 2 missed
        lambda: np.copyto(dst=u, src=u * 1.0),               # the target as a keyword
        lambda: np.put(a=u, ind=[0], v=[0.3]),
        lambda: np.dot(u, np.eye(4), out=u[:, :]),           # out= of functions that are not ufuncs
        lambda: np.cumsum(u, axis=0, out=u),
        lambda: np.take(u, np.arange(30), axis=0, out=u),
    ]
    for i, w in enumerate(writes):
        before = _gen(h)
        w()
        assert _gen(h) > before, "write %d was not counted" % i
    assert np.asarray(u)[3, 0] != 123.0


def test_boolean_scalar_key_is_not_a_row_number():
    """u[True] = x writes EVERY row (numpy treats a scalar bool as a mask); it must not be remembered as 'row 1'."""
    h = _Holder(np.random.RandomState(3).uniform(size=(10, 2)))
    cell = h.__dict__["_u_cell"]
    cell[1] = []                      # as after a device sync: row assignments are being remembered
    h.u[3] = 0.5
    assert cell[1] == [3]
    h.u[True] = 0.25
    assert cell[1] is None, "a whole-array write must force a diff"
    assert (np.asarray(h.u) == 0.25).all()


def test_results_of_arithmetic_are_plain_arrays():
    h = _Holder(np.random.RandomState(3).uniform(size=(10, 2)))
    assert type(h.u + 1) is np.ndarray and type(np.sqrt(h.u)) is np.ndarray
    assert type(h.u.mean(axis=0)) is np.ndarray
    r = pickle.loads(pickle.dumps(h.u))
    assert type(r) is np.ndarray and np.array_equal(r, h.u)


def test_row_assignments_are_remembered_until_something_else_writes():
    h = _Holder(np.random.RandomState(4).uniform(size=(50, 3)))
    cell = h.__dict__["_u_cell"]
    assert cell[1] is None            # before the first device sync every write means "diff everything"
    cell[1] = []                      # what _DeviceState.sync does after an upload
    h.u[7] = 0.5
    h.u[np.array([3])] = 0.25
    h.u[-1, :] = 0.125
    assert cell[1] == [7, 3, -1]
    h.u[2:4] = 0.1                    # a slice: not tracked by row
    assert cell[1] is None


class _FakeRegion(regions._LivePoints):
    def __init__(self, u):
        self._dev = regions._DeviceState()
        self.u = u
        self.transformLayer = type("L", (), {})()
        self.transformLayer.T = np.eye(3)
        self.transformLayer.ctr = np.zeros(3)
        self.ellipsoid_center = np.zeros(3)
        self.ellipsoid_invcov = np.eye(3)
        self.enlarge = 2.0
        self.maxradiussq = 0.5


def test_device_state_is_reused_only_while_nothing_moved(monkeypatch):
    reg = _FakeRegion(np.random.RandomState(5).uniform(size=(40, 3)))
    st = reg._dev
    calls = []
    monkeypatch.setattr(st, "_sync_slow", lambda region, use_scan, key: calls.append(1) or "handle")
    st.handle = "handle"
    assert st.sync(reg, True) == "handle" and len(calls) == 1        # first call: slow
    assert st.sync(reg, True) == "handle" and len(calls) == 1        # nothing moved: fast
    reg.u[3] = 0.5
    st.sync(reg, True)
    assert len(calls) == 2                                           # live point written
    st.sync(reg, True)
    assert len(calls) == 2
    for change in (lambda: setattr(reg, "ellipsoid_center", np.zeros(3)),
                   lambda: setattr(reg, "ellipsoid_invcov", np.eye(3)),
                   lambda: setattr(reg, "enlarge", 2.5),
                   lambda: setattr(reg, "maxradiussq", 0.25),
                   lambda: setattr(reg.transformLayer, "T", np.eye(3)),
                   lambda: setattr(reg, "transformLayer", reg.transformLayer.__class__()),
                   lambda: setattr(reg, "u", np.asarray(reg.u) * 1.0),
                   lambda: reg.invalidate_device_state()):
        n = len(calls)
        change()
        if not hasattr(reg.transformLayer, "T"):
            reg.transformLayer.T, reg.transformLayer.ctr = np.eye(3), np.zeros(3)
        st.sync(reg, True)
        assert len(calls) == n + 1
        st.sync(reg, True)
        assert len(calls) == n + 1
    st.sync(reg, False)                                              # the ellipsoid-only use is a different state
    assert len(calls) == n + 2


def test_untracked_live_points_never_take_the_fast_path(monkeypatch):
    """a region whose ``_u`` is a plain array (e.g. restored by pickle: _LiveArray pickles as the array it holds) must be
    diffed on every call"""
    reg = _FakeRegion(np.random.RandomState(6).uniform(size=(10, 3)))
    st = reg._dev
    calls = []
    monkeypatch.setattr(st, "_sync_slow", lambda region, use_scan, key: calls.append(1) or "handle")
    st.handle = "handle"
    st.sync(reg, True)
    st.sync(reg, True)
    assert len(calls) == 1
    reg.__dict__["_u"] = np.array(reg.u)          # what unpickling leaves behind
    st.sync(reg, True)
    st.sync(reg, True)
    assert len(calls) == 3
    reg.u = reg.__dict__["_u"]                    # assigning through the property tracks it again
    st.sync(reg, True)
    st.sync(reg, True)
    assert len(calls) == 4
