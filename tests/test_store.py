"""Point-store format: the tab-separated store of the reference
(ultranest/store.py:109-158) written and read byte-compatibly; resume of the harness from a store.
tests/golden/g11_pointstore.tsv was written by the reference's TextPointStore (make_golden.py g11)."""
import os

import numpy as np
import pytest

from golden import inputs

HERE = os.path.dirname(os.path.abspath(__file__))


def _rows():
    rs = np.random.RandomState(1100)
    rows = []
    Lmin = -np.inf
    for i in range(40):
        L = rs.normal()
        rows.append([Lmin, L, float(rs.randint(1, 400))] + list(rs.uniform(size=3)) + list(rs.normal(size=2) * 1e3))
        if i % 3 == 2:
            Lmin = min(L, 0.5) if np.isfinite(Lmin) else -1.0
    rows.append([1e-300, 1.7976931348623157e308, 0.0, 0.1, 0.2, 0.3, -0.0, 5e-324])
    return rows


def test_text_store_bytes_equal_reference(tmp_path):
    from ultranest_amd.store import TextPointStore
    path = str(tmp_path / "points.tsv")
    store = TextPointStore(path, 8)
    assert store.stack_empty and store.ncalls == 0
    for i, row in enumerate(_rows()):
        assert store.add(row, 10 + i) == i
    assert store.nrows == 41 and store.ncalls == 50
    with pytest.raises(ValueError):
        store.add([1.0, 2.0], 1)
    store.close()
    want = open(os.path.join(HERE, "golden", "g11_pointstore.tsv"), "rb").read()
    assert open(path, "rb").read() == want


def test_text_store_reads_reference_file_and_pops_like_it(tmp_path):
    from ultranest_amd.store import TextPointStore
    import shutil
    path = str(tmp_path / "points.tsv")
    shutil.copy(os.path.join(HERE, "golden", "g11_pointstore.tsv"), path)
    with open(path, "a") as f:
        f.write("not a number\n1.0\t2.0\n")                  # tolerated like in the reference: skipped with a warning
    with pytest.warns(UserWarning):
        store = TextPointStore(path, 8)
    assert len(store.stack) == 41 and store.ncalls == 41 and not store.stack_empty
    order = np.loadtxt(os.path.join(HERE, "golden", "g11_pointstore_pops.txt"), ndmin=2)
    for Lmin, want_idx in order:
        idx, row = store.pop(Lmin)
        assert (idx if idx is not None else -1) == int(want_idx)
        if idx is not None:
            assert row[0] <= Lmin and row[1] > Lmin
    store.close()


def test_null_store():
    from ultranest_amd.store import NullPointStore
    s = NullPointStore(5)
    assert s.add([1, 2, 3, 4, 5], 7) == 0 and s.ncalls == 7 and s.pop(0.0) == (None, None)
    s.reset(); s.flush(); s.close()


def test_harness_resumes_from_store(backend, tmp_path):
    """A finished run replayed from its store needs no new likelihood evaluation until the store is
    exhausted and gives the same evidence."""
    from ultranest_amd.harness import StaticNestedSampler
    from ultranest_amd.store import TextPointStore
    d, sigma = 3, 0.1
    calls = [0]

    def loglike(v):
        calls[0] += len(v)
        return -0.5 * (((v - 0.5) / sigma)**2).sum(axis=1)

    path = str(tmp_path / "run.tsv")
    store = TextPointStore(path, 3 + 2 * d)
    first = StaticNestedSampler(d, loglike, num_live_points=100, ndraw=256, seed=4, pointstore=store).run(dlogz=0.5)
    store.close()
    ncalls_first = calls[0]
    assert ncalls_first == first["ncall"] and store.nrows == ncalls_first
    store = TextPointStore(path, 3 + 2 * d)
    assert len(store.stack) == ncalls_first
    calls[0] = 0
    again = StaticNestedSampler(d, loglike, num_live_points=100, ndraw=256, seed=4, pointstore=store).run(dlogz=0.5)
    store.close()
    assert abs(again["logz"] - first["logz"]) < 0.5
    assert calls[0] < 0.2 * ncalls_first, (calls[0], ncalls_first)
