import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Everything native is built in-tree by __graft_entry__.build(); build on demand if a library is missing."""
    from hipstr_amd import build, capi
    if not (os.path.exists(capi.ORACLE_LIB) and os.path.exists(capi.SYNTH_LIB)):
        build.build_synth()
        build.build_oracle()
    if not os.path.exists(capi.HMM_LIB):
        build.build_hmm()


@pytest.fixture(scope="session")
def oracle():
    from hipstr_amd import capi
    return capi.load_oracle()


@pytest.fixture(scope="session")
def hmm_host():
    """The product library loaded WITHOUT touching a device (host-only entry points)."""
    from hipstr_amd import capi
    return capi.load_hmm()


@pytest.fixture(scope="session")
def hmm():
    """The product library with device 0 initialised; fails loudly when there is no GPU."""
    from hipstr_amd import capi
    lib = capi.load_hmm()
    rc = lib.hipstr_hmm_init(0)
    assert rc == 0, "hipstr_hmm_init failed: " + lib.hipstr_last_error().decode()
    return lib
