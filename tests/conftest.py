import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libfm_oracle.so), built on demand with gcc."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def fmd():
    """The product package (ctypes binding over the HIP C-ABI library)."""
    import rtl_sdr_rs_amd
    return rtl_sdr_rs_amd


def _has_gpu():
    try:
        import rtl_sdr_rs_amd
        return rtl_sdr_rs_amd.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A `-m gpu` run on a box without a GPU must fail loudly, not skip: the product has no CPU path.
    pass
