"""C-ABI boundary (CPU): libmlfriends_hip.so loads without a GPU, exports every symbol that
include/mlfriends_hip.h declares, the ctypes table binds each with the declared number of
arguments, and compute calls FAIL LOUDLY when no device is present (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mlfriends_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(mlf_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return decls


def test_header_symbols_exported_and_bound():
    from ultranest_amd import _lib
    decls = _declared()
    assert len(decls) >= 30
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(L, name), "symbol %s declared in the header but not exported" % name
        assert name in _lib.SIGNATURES, "symbol %s not bound by ultranest_amd._lib" % name
        assert len(_lib.SIGNATURES[name]) == nargs, (name, nargs, len(_lib.SIGNATURES[name]))
    assert set(_lib.SIGNATURES) == set(decls), set(_lib.SIGNATURES) ^ set(decls)
    assert _lib.lib().mlf_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback_without_device():
    from ultranest_amd import _lib, kernels
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    out = np.empty(2, dtype=np.int64)
    with pytest.raises(_lib.HipLibraryError):
        kernels.find_nearby(np.zeros((3, 2)), np.zeros((2, 2)), 1.0, out)
    with pytest.raises(_lib.HipLibraryError):
        kernels.maxradiussq_bootstrap(np.zeros((4, 2)), np.ones((1, 4), dtype=bool))
    with pytest.raises(_lib.HipLibraryError):
        kernels.DeviceRegion()
    import ultranest_amd.likelihoods as L
    with pytest.raises(_lib.HipLibraryError):
        L.eggbox_loglike(np.zeros((2, 2)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ultranest_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), "product file %s mentions the oracle" % f


def test_argument_validation_is_host_side():
    from ultranest_amd import kernels
    with pytest.raises(ValueError):
        kernels.find_nearby(np.zeros((3, 2)), np.zeros((2, 3)), 1.0, np.empty(2, dtype=np.int64))
    with pytest.raises(ValueError):
        kernels.find_nearby(np.zeros((3, 2)), np.zeros((2, 2)), 1.0, np.empty(2, dtype=np.int32))
    with pytest.raises(ValueError):
        kernels.maxradiussq_bootstrap(np.zeros((4, 2)), np.ones((1, 5), dtype=bool))


def test_step_sampler_path_fails_loudly_without_device():
    from ultranest_amd import _lib
    import ultranest_amd.popstepsampler as pop
    import ultranest_amd.stepfuncs as sf
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.HipLibraryError):
        sf.within_unit_cube(np.zeros((3, 2)))
    with pytest.raises(_lib.HipLibraryError):
        sf.step_back(0.0, np.zeros((2, 3)), np.zeros(2, dtype=np.int64), np.zeros(2))
    with pytest.raises(_lib.HipLibraryError):
        pop._Walkers(4, 3, 2)


def test_host_helpers_need_no_device():
    """changed_rows (lazy device mirror) and the compiled MultiCounter are host code."""
    from ultranest_amd import _lib
    a = np.random.RandomState(1).uniform(size=(50, 7))
    b = a.copy()
    rows, n = _lib.changed_rows(a, b)
    assert n == 0 and len(rows) == 0
    b[[3, 17, 49]] += 1e-16 * (1 + b[[3, 17, 49]])       # one-ulp changes are seen
    b[5, 2] = -0.0 if b[5, 2] == 0 else b[5, 2]
    rows, n = _lib.changed_rows(a, b)
    assert n == 3 and list(rows) == [3, 17, 49]
    rows, n = _lib.changed_rows(a, b, capacity=2)          # more changes than capacity: count is still exact
    assert n == 3 and list(rows) == [3, 17]


def test_every_documented_option_is_accepted_without_a_device():
    """mlf_set_option only stores a process default: every name the header documents is accepted (and restored)
    without a GPU, the library enumerates exactly those names, an unknown name is an error."""
    import ctypes
    import re
    from ultranest_amd import _lib
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "mlfriends_hip.h")).read()
    block = header[header.index("Tuning options."):header.index("int mlf_set_option")]
    names = re.findall(r'^ \*   "([a-z_]+)"', block, flags=re.M)
    listed = []
    buf = ctypes.create_string_buffer(64)
    while _lib.lib().mlf_option_name(len(listed), buf, 64) == 0:
        listed.append(buf.value.decode())
    assert sorted(names) == sorted(listed) and len(listed) == 20
    defaults = {"filter_min_queries": 257, "filter_phase_min_queries": 32768, "filter_first_range_pct": 30,
                "filter_split_waves": 2048, "time_filter_launches": 0, "mid_max_queries": 2048, "filter_second_range_pct": 50,
                "filter_third_range_min_work": 100000000, "fused_waves": 4, "fused_variant": 3}
    for name in names:
        _lib.set_option(name, defaults.get(name, 1))
    with pytest.raises(ValueError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)
