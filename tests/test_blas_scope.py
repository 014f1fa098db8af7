"""The one-thread BLAS limit of the host linear algebra (layers.single_blas_thread) is taken by the first scope that is
entered and given back by the last one that is left -- the scopes overlap across the calling thread and the host worker
(regions._start_ellipsoid_parts) and are not nested."""
import threading

import pytest


def _blas_threads():
    from threadpoolctl import threadpool_info
    return [i["num_threads"] for i in threadpool_info() if i["user_api"] == "blas"]


def test_overlapping_scopes_on_two_threads_restore_the_pool():
    pytest.importorskip("threadpoolctl")
    import numpy as np
    from ultranest_amd import layers
    np.dot(np.ones((4, 4)), np.ones((4, 4)))          # the BLAS library is loaded
    before = _blas_threads()
    if not before or max(before) == 1:
        pytest.skip("no multi-threaded BLAS pool to limit")
    entered, release = threading.Event(), threading.Event()
    seen = {}

    @layers.single_blas_thread
    def outer():
        entered.set()
        release.wait(5)
        seen["outer"] = _blas_threads()

    @layers.single_blas_thread
    def inner():
        return _blas_threads()

    t = threading.Thread(target=outer)
    t.start()
    assert entered.wait(5)
    assert set(inner()) == {1}
    assert set(_blas_threads()) == {1}                # the inner scope left while the outer one is still inside
    release.set()
    t.join()
    assert set(seen["outer"]) == {1}
    assert _blas_threads() == before                  # the last one out gives the pool back


def test_scope_survives_an_exception():
    pytest.importorskip("threadpoolctl")
    from ultranest_amd import layers
    before = _blas_threads()

    @layers.single_blas_thread
    def boom():
        raise ValueError("x")

    with pytest.raises(ValueError):
        boom()
    assert _blas_threads() == before
    assert layers._blas_scope["depth"] == 0
