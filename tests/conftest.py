import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The native pieces are built in-tree by __graft_entry__.build() / `make -C sfmlocalization_amd/csrc`; if a
    checkout arrives without them (they are not in git), build them once here rather than failing every test."""
    import subprocess
    lib = os.path.join(ROOT, "sfmlocalization_amd", "lib", "libsfmloc_hip.so")
    cli = os.path.join(ROOT, "sfmlocalization_amd", "bin", "OpenMVGLocalization_AKAZE")
    smoke = os.path.join(ROOT, "sfmlocalization_amd", "bin", "engine_smoke")
    extf = os.path.join(ROOT, "sfmlocalization_amd", "bin", "ExtFeatAndMatch")
    if not (os.path.exists(lib) and os.path.exists(cli) and os.path.exists(smoke) and os.path.exists(extf)):
        subprocess.call(["make", "-C", os.path.join(ROOT, "sfmlocalization_amd", "csrc")])


@pytest.fixture(scope="session")
def oracle_c():
    from oracle import oracle_c as oc
    oc.build()
    return oc


@pytest.fixture(scope="session")
def gpu_available():
    import sfmlocalization_amd as s
    return s.device_count() > 0
