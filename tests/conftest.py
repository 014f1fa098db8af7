import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture(autouse=True)
def _watchdog():
    """A test that has not returned after 15 minutes dumps the stacks of all its threads and ends the run (the watchdog is a
    C thread of faulthandler: it also fires when the interpreter itself is stuck, e.g. in a lock held across threads)."""
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)
    yield
    faulthandler.cancel_dump_traceback_later()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # MLF_TEST_OPTIONS="name=value,name=value": process-wide tuning options for this run of the GPU suite (results never depend
    # on them: that is what running the suite under a non-default routing checks)
    spec = os.environ.get("MLF_TEST_OPTIONS", "")
    if spec:
        from ultranest_amd import _lib
        for kv in spec.split(","):
            name, value = kv.split("=")
            _lib.set_option(name.strip(), int(value))


@pytest.fixture(scope="session")
def golden():
    """name -> lazily loaded npz fixture generated from the real reference."""
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(params=["oracle-stub", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """Host-logic tests run twice: on CPU with the kernels monkeypatched to the oracle (no GPU
    needed; exercises the Python layer only) and unpatched on the MI355X (`-m gpu`)."""
    if request.param == "oracle-stub":
        import oracle_backend
        oracle_backend.install(monkeypatch)
    else:
        from ultranest_amd import _lib
        assert _lib.device_count() >= 1
    return request.param
