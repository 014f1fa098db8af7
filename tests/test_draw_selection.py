"""The compiled bootstrap draw (mlf_host_draw_selection) against numpy: B calls of ``rng.randint(N, size=N)`` on a
legacy MT19937 stream (reference mlfriends.pyx:1044-1047), including where the generator is left afterwards."""
import numpy as np
import pytest

from ultranest_amd import _lib, regions


def _numpy_rounds(rng, npoints, nrounds):
    masks = np.zeros((nrounds, npoints), dtype=bool)
    for b in range(nrounds):
        masks[b, rng.randint(npoints, size=npoints)] = True
    return masks


@pytest.mark.parametrize("npoints", [1, 2, 3, 77, 400, 4000, 4096, 4097, 65536, 100003])
def test_same_draws_and_same_generator_state(npoints):
    for seed, nrounds in ((0, 1), (1, 7), (2, 30)):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        for rng in (a, b):
            rng.normal()                       # a cached gaussian has to survive
            rng.randint(9, size=seed * 211)    # start in the middle of a 624-word block
        want = _numpy_rounds(a, npoints, nrounds)
        got = _lib.draw_selection(b, npoints, nrounds)
        assert got is not None and got.dtype == bool and np.array_equal(want, got)
        assert a.normal() == b.normal() and np.array_equal(a.randint(1 << 30, size=700), b.randint(1 << 30, size=700))


def test_global_stream_and_other_generators():
    np.random.seed(3)
    want = _numpy_rounds(np.random, 500, 30)
    after = np.random.uniform()
    np.random.seed(3)
    got = regions._draw_selection(np.random, 500, 30)
    assert np.array_equal(want, got) and after == np.random.uniform()
    assert _lib.draw_selection(np.random.default_rng(1), 10, 2) is None   # not a legacy MT19937 stream

    class Fixed:                                   # anything with randint still works (numpy loop)
        def randint(self, n, size):
            return np.arange(size) % 3
    m = regions._draw_selection(Fixed(), 10, 2)
    assert m[:, :3].all() and not m[:, 3:].any()


def test_bad_arguments_are_refused():
    key = np.zeros(624, dtype=np.uint32)
    import ctypes
    pos = ctypes.c_int32(700)
    out = np.zeros(4, dtype=np.uint8)
    assert _lib.lib().mlf_host_draw_selection(key.ctypes.data, ctypes.byref(pos), 4, 1, out.ctypes.data) != 0
    pos = ctypes.c_int32(0)
    assert _lib.lib().mlf_host_draw_selection(key.ctypes.data, ctypes.byref(pos), 0, 1, out.ctypes.data) != 0
