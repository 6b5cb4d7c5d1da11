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

    # A compiler error in the checker must not turn the reference-pinned tests into silent skips: only a missing
    # /root/reference (the GPU box, where the prebuilt oracle/_ref travels with the snapshot) leads to a skip.
    try:
        r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=900)
    except (OSError, subprocess.TimeoutExpired) as exc:
        pytest.exit(f"could not build the checker libraries under oracle/: {exc!r}", returncode=3)
    if r.returncode != 0:
        pytest.exit("`make -C oracle` failed — the oracle / reference checker does not build:\n" + r.stdout[-2000:] + r.stderr[-4000:],
                    returncode=3)


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
