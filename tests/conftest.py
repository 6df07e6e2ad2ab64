import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multigpu: needs at least two MI355X on one node (real RCCL at world size > 1); skipped elsewhere")


def _built():
    lib = os.path.join(ROOT, "dm-vio_amd", "lib", "libdmvio_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    return os.path.exists(lib) and os.path.exists(orc)


@pytest.fixture(scope="session")
def pkg():
    # a fresh checkout has no binaries (they are git-ignored): build once — cross-compiling for gfx950 needs no GPU
    if not _built() and os.path.exists("/opt/rocm/bin/hipcc"):
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle(pkg):
    return graft.load_oracle()


@pytest.fixture(scope="session")
def synth(pkg):
    import dmvio_amd.synth as s
    return s


def _have_gpu():
    try:
        p = graft.load_package()
        return p.load_library().dmvio_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required(pkg):
    # -m gpu tests must FAIL (not skip) when the HIP library or the device is missing: no silent fallback.
    lib = pkg.load_library()
    assert lib.dmvio_hip_device_count() > 0, "no HIP device visible"
    return True
