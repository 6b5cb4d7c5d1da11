import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # The checker libraries must exist before collection: test modules decide at import time whether the compiled reference
    # (oracle/_ref, buildable only where /root/reference exists) is there.  `make` is a no-op when everything is up to date.
    import subprocess

    try:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=False, capture_output=True, timeout=600)
    except (OSError, subprocess.TimeoutExpired):
        pass


@pytest.fixture(scope="session")
def asm():
    """The product package (ctypes over the C ABI).  Building is `__graft_entry__.build()`."""
    import approximate_string_matching_amd as m

    if not os.path.exists(m.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return m


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_binding

    return oracle_binding.load_oracle()


@pytest.fixture(scope="session")
def engine(asm):
    """One engine (asm_handle) on cuda:0 for the whole GPU session."""
    if asm.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the GPU box; there is no CPU fallback")
    eng = asm.Engine(0)
    yield eng
    eng.close()
