"""ctypes binding of libmlfriends_hip.so (C ABI: include/mlfriends_hip.h).

The library is the ONLY compute backend of this package.  If it is missing, or if no MI355X is
visible when a kernel is called, an exception is raised -- there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmlfriends_hip.so")

ABI_VERSION = 1

_c_double_p = ctypes.c_void_p
_sz = ctypes.c_size_t
_dbl = ctypes.c_double
_int = ctypes.c_int
_vp = ctypes.c_void_p

# symbol -> argtypes; this table is what tests/test_abi.py checks against include/mlfriends_hip.h
SIGNATURES = {
    "mlf_abi_version": [],
    "mlf_last_error": [],
    "mlf_device_count": [_vp],
    "mlf_set_device": [_int],
    "mlf_device_name": [_vp, _sz],
    "mlf_synchronize": [],
    "mlf_set_option": [ctypes.c_char_p, ctypes.c_longlong],
    "mlf_get_option": [ctypes.c_char_p, _vp],
    "mlf_debug_forget_grants": [_vp],
    "mlf_region_get_option": [_vp, ctypes.c_char_p, _vp],
    "mlf_find_nearby": [_vp, _sz, _vp, _sz, _sz, _dbl, _vp],
    "mlf_count_nearby": [_vp, _sz, _vp, _sz, _sz, _dbl, _vp],
    "mlf_subtract_nearby": [_vp, _sz, _sz, _dbl, _vp],
    "mlf_cluster_labels": [_vp, _sz, _sz, _dbl, _vp, _vp, _vp],
    "mlf_adjacency_bits": [_vp, _sz, _sz, _dbl, _vp],
    "mlf_host_cluster_replay": [_vp, _sz, _vp, _vp, _vp],
    "mlf_region_hint_live_extent": [_vp, _dbl],
    "mlf_region_set_option": [_vp, ctypes.c_char_p, ctypes.c_longlong, ctypes.c_int],
    "mlf_option_name": [ctypes.c_int, _vp, _sz],
    "mlf_dev_alloc": [_sz, _vp],
    "mlf_dev_free": [_vp],
    "mlf_dev_copy": [_vp, _vp, _sz, ctypes.c_int],
    "mlf_col_extent": [_vp, _sz, _sz, _vp, _vp],
    "mlf_maxradiussq_bootstrap": [_vp, _sz, _sz, _vp, _sz, _vp, _vp],
    "mlf_maxradiussq_bootstrap_rows": [_vp, _sz, _sz, _vp, _sz, _sz, _sz, _vp, _vp],
    "mlf_pair_dist2_lower": [_vp, _sz, _sz, _vp],
    "mlf_inside_ellipsoid": [_vp, _sz, _sz, _vp, _vp, _dbl, _vp, _vp],
    "mlf_affine_transform": [_vp, _sz, _sz, _vp, _vp, _vp, _vp],
    "mlf_bootstrap_moments": [_vp, _sz, _sz, _vp, _sz, _vp, _vp],
    "mlf_bootstrap_factor": [_vp, _sz, _sz, _vp, _sz, _dbl, _vp],
    "mlf_bootstrap_quadform_max": [_vp, _sz, _sz, _vp, _sz, _vp, _vp, _vp],
    "mlf_region_create": [_vp],
    "mlf_region_destroy": [_vp],
    "mlf_region_set": [_vp, _vp, _sz, _sz, _int, _int, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _int],
    "mlf_region_update_point": [_vp, _sz, _vp],
    "mlf_region_update_points": [_vp, _sz, _vp, _vp],
    "mlf_region_set_thresholds": [_vp, _dbl, _dbl],
    "mlf_region_set_ellipsoid_center": [_vp, _vp],
    "mlf_region_inside": [_vp, _vp, _sz, _vp],
    "mlf_region_inside_dev": [_vp, _vp, _sz, _vp, _vp],
    "mlf_region_find_nearby_dev": [_vp, _vp, _sz, _vp, _vp],
    "mlf_loglike_gauss": [_vp, _sz, _sz, _vp, _dbl, _vp],
    "mlf_loglike_eggbox": [_vp, _sz, _sz, _vp],
    "mlf_loglike_eggbox2": [_vp, _sz, _sz, _vp],
    "mlf_loglike_rosenbrock": [_vp, _sz, _sz, _vp],
    "mlf_loglike_dev": [_int, _vp, _sz, _sz, _vp, _dbl, _vp, _vp],
    "mlf_region_time_inside_dev": [_vp, _vp, _sz, _vp, _vp, _int, _vp, _vp],
    "mlf_region_first_index_dev": [_vp, _vp, _sz, _vp, _vp],
    "mlf_region_set_axes": [_vp, _vp],
    "mlf_region_set_sampling_data": [_vp, _vp, _vp, _vp],
    "mlf_region_sample": [_vp, _int, _sz, ctypes.c_uint64, ctypes.c_uint64, _vp, _sz, _vp, _vp],
    "mlf_region_refill": [_vp, _int, _sz, ctypes.c_uint64, ctypes.c_uint64, _dbl, _int, _dbl, _dbl, _int, _vp, _dbl, _vp, _vp,
                          _vp, _sz, _vp, _vp, _vp],
    "mlf_debug_philox": [ctypes.c_uint64, ctypes.c_uint, _sz, _vp],
    "mlf_within_unit_cube": [_vp, _sz, _sz, _vp],
    "mlf_evolve_propose": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp],
    "mlf_evolve_update": [_vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _sz],
    "mlf_step_back": [_dbl, _vp, _sz, _sz, _vp, _vp],
    "mlf_unitcube_line_intersection": [_vp, _vp, _sz, _sz, _vp, _vp],
    "mlf_update_vectorised_slice_sampler": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp, _vp, _vp, _sz,
                                            _sz, _sz, _vp],
    "mlf_row_dist2": [_vp, _vp, _sz, _sz, _vp],
    "mlf_walkers_create": [_vp, _sz, _sz, _sz],
    "mlf_walkers_destroy": [_vp],
    "mlf_walkers_reset": [_vp],
    "mlf_walkers_begin": [_vp, _dbl, _vp, _vp],
    "mlf_walkers_start": [_vp, _vp, _sz, _vp, _vp],
    "mlf_walkers_points": [_vp, _vp, _sz, _vp],
    "mlf_walkers_brackets": [_vp, _vp, _sz, _dbl, _vp],
    "mlf_walkers_set_direction_data": [_vp, _vp, _vp, _sz, _vp],
    "mlf_walkers_brackets_philox": [_vp, _dbl, _int, _dbl, ctypes.c_uint64, ctypes.c_uint64, _vp],
    "mlf_walkers_set_layer": [_vp, _int, _vp, _vp, _vp, _dbl],
    "mlf_walkers_propose": [_vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp],
    "mlf_walkers_finish": [_vp, _dbl, _vp, _vp, _sz, _sz, ctypes.c_int64, _vp],
    "mlf_walkers_finish_dev": [_vp, _dbl, _int, _dbl, _dbl, _int, _vp, _dbl, ctypes.c_int64, _vp],
    "mlf_walkers_set_live": [_vp, _vp, _vp, _sz],
    "mlf_walkers_update_live": [_vp, _vp, _sz, _vp, _vp],
    "mlf_walkers_step_dev": [_vp, _dbl, _dbl, _int, _dbl, ctypes.c_uint64, ctypes.c_uint64, _int, _dbl, _dbl, _int, _vp, _dbl,
                             _vp, _vp],
    "mlf_walkers_step_graph": [_vp, _dbl, _dbl, _int, _dbl, ctypes.c_uint64, ctypes.c_uint64, _int, _dbl, _dbl, _int, _vp, _dbl,
                               _vp, _vp],
    "mlf_walkers_rounds_dev": [_vp, _dbl, _dbl, _int, _dbl, ctypes.c_uint64, ctypes.c_uint64, _int, _dbl, _dbl, _int, _vp, _dbl,
                               _int, _vp, _vp, _vp, _vp],
    "mlf_walkers_export": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlf_host_changed_rows": [_vp, _vp, _sz, _sz, _vp, _sz, _vp],
    "mlf_host_draw_selection": [_vp, _vp, _sz, _sz, _vp],
    "mlf_counter_create": [_vp, _sz, _sz, _vp, _int, _int],
    "mlf_counter_destroy": [_vp],
    "mlf_counter_reset": [_vp],
    "mlf_counter_passing_node": [_vp, ctypes.c_int64, _dbl, _sz, _vp, _vp, _vp, _sz, _vp, _vp],
    "mlf_counter_state": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mlf_region_inside_dev_timed": [_vp, _vp, _sz, _vp, _vp],
    "mlf_region_timing_collect": [_vp, _vp, _vp, _vp, _vp],
    "mlf_region_timing_filter_launches": [_vp, _vp, _vp],
    "mlf_region_timing_filter_launch_ms": [_vp, _vp, ctypes.c_int, _vp],
    "mlf_region_filter_info": [_vp, _sz, _vp, _vp, _vp],
    "mlf_bench_fp64_valu": [_vp],
    "mlf_region_debug_stats": [_vp, _vp, _int],
    "mlf_region_debug_fused_stamps": [_vp, _int, _vp, _int],
    "mlf_comm_unique_id": [_vp, _sz],
    "mlf_comm_init_rank": [_vp, _sz, _int, _int],
    "mlf_comm_init": [_int],
    "mlf_allreduce_max": [_vp, _sz],
    "mlf_comm_destroy": [],
}

_lib = None


class HipLibraryError(RuntimeError):
    """The HIP library is missing or a HIP call failed (never silently replaced by CPU code)."""


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and load them by path.  If this
    library pulled in the system copy first and torch were imported later, the process would hold two HIP
    runtimes and the second one finds no device.  Loading torch's copy first (without importing torch) makes
    both orders end up on one runtime: the dynamic loader satisfies this library's libamdhip64.so.N dependency
    from the already loaded object, and torch's later load of the same file is a no-op."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass    # fall back to the system runtime; a conflict then shows up as a loud "no device" error


def lib():
    """Load (once) and return the ctypes handle.  Loading needs no GPU; calling kernels does."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "%s not found: build it with `python ultranest_amd/csrc/build.py` "
                "(ultranest_amd has no CPU fallback)" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a symbol is missing
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name == "mlf_last_error" else ctypes.c_int
        if L.mlf_abi_version() != ABI_VERSION:
            raise HipLibraryError("libmlfriends_hip.so ABI %d != expected %d" % (L.mlf_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    """Map a status code to an exception (0 = ok)."""
    if rc == 0:
        return
    msg = lib().mlf_last_error()
    msg = msg.decode() if msg else ""
    if rc < 0:
        raise HipLibraryError("HIP failure (%d): %s" % (rc, msg))
    if rc == 3:
        raise HipLibraryError(msg or "no HIP device")
    raise ValueError("mlfriends_hip: %s (code %d)" % (msg, rc))


def f64(a):
    """C-contiguous float64 view/copy (the reference accepts arbitrary strides; the ABI does not)."""
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def device_count():
    n = ctypes.c_int(0)
    check(lib().mlf_device_count(ctypes.byref(n)))
    return n.value


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(lib().mlf_device_name(buf, 256))
    return buf.value.decode()


def set_device(i):
    check(lib().mlf_set_device(int(i)))


def set_option(name, value):
    """Tuning switch of the library (e.g. set_option("filter", 0) forces the exact scan only)."""
    check(lib().mlf_set_option(name.encode(), int(value)))


def get_option(name):
    """The process default of a tuning switch that is in force (mlf_get_option)."""
    v = ctypes.c_longlong(0)
    check(lib().mlf_get_option(name.encode(), ctypes.byref(v)))
    return int(v.value)


def forget_grants():
    """Test hook (mlf_debug_forget_grants): every kernel asks for its dynamic-LDS grant again at its next launch; returns
    the number of grants the process had issued so far."""
    n = ctypes.c_ulonglong(0)
    check(lib().mlf_debug_forget_grants(ctypes.byref(n)))
    return int(n.value)


def draw_selection(rng, npoints, nbootstraps):
    """(B, N) boolean selection masks from a legacy MT19937 stream (``np.random`` or a RandomState): what B
    calls of ``rng.randint(N, size=N)`` select, with the generator left where those calls leave it.  None if
    `rng` is not such a generator."""
    try:
        state = rng.get_state()
    except (AttributeError, TypeError):
        return None
    if not isinstance(state, tuple) or state[0] != "MT19937" or not 0 < npoints <= 1 << 31:
        return None
    key = np.array(state[1], dtype=np.uint32)
    pos = ctypes.c_int32(int(state[2]))
    masks = np.empty((nbootstraps, npoints), dtype=bool)
    check(lib().mlf_host_draw_selection(key.ctypes.data, ctypes.byref(pos), npoints, nbootstraps, masks.ctypes.data))
    rng.set_state((state[0], key, pos.value) + tuple(state[3:]))
    return masks


def changed_rows(a, b, capacity=64):
    """Indices of the rows where two C-contiguous float64 (n, d) arrays differ bytewise, and their total
    number (compiled memcmp per row)."""
    rows = np.empty(capacity, dtype=np.int64)
    count = ctypes.c_size_t(0)
    check(lib().mlf_host_changed_rows(a.ctypes.data, b.ctypes.data, a.shape[0], a.shape[1], rows.ctypes.data, capacity,
                                      ctypes.byref(count)))
    return rows[:min(count.value, capacity)], count.value
