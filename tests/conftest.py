import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
SAMPLE = os.path.join(GOLDEN, "sample.fasta")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def kmc():
    """The product package.  Builds libkmc.so in-tree if it is missing (hipcc cross-compiles)."""
    m = importlib.import_module("k-mer-count_amd")
    # (re)build when the library is missing OR older than any of its sources: a stale libkmc.so must
    # never be what the tests exercise.  `make` decides (it knows the dependencies); where the
    # toolchain is absent an up-to-date library is used as it is and a stale one is an error.
    here = os.path.dirname(m.LIB_PATH)
    srcs = [os.path.join(here, "csrc", f) for f in os.listdir(os.path.join(here, "csrc"))] + [os.path.join(ROOT, "include", "kmc.h")]
    stale = (not os.path.exists(m.LIB_PATH)) or any(os.path.getmtime(f) > os.path.getmtime(m.LIB_PATH) for f in srcs)
    if stale and "KMC_LIB_PATH" not in os.environ:
        m.build()
    m.lib()
    return m


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def sample_path():
    return SAMPLE
