"""The drop-in adapter (tests/dropin/dmvio_hip_adapter.cpp: the INTEGRATION.md adapter compiled against the reference's own headers) without a device:
it builds, exports the five members it replaces under the reference's exact mangled names, the reference's library reaches those members through its PLT (so that
loading the adapter first really puts it in the reference's call path), and with the adapter switched OFF — every member forwarding to the reference's own
definition through dlsym(RTLD_NEXT) — the reference's FullSystem produces bit for bit what it produces without the adapter in the process."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_py as R  # noqa: E402

REF_DIR = os.path.join(ROOT, "oracle", "_ref")
DROPIN = os.path.join(REF_DIR, "libdropin_hip.so")
MEMBERS = ["_ZN3dso12FrameHessian10makeImagesEPfPNS_12CalibHessianE",
           "_ZN3dso13CoarseTracker20setCoarseTrackingRefESt6vectorIPNS_12FrameHessianESaIS3_EE",
           "_ZN3dso13CoarseTracker17trackNewestCoarseEPNS_12FrameHessianERN6Sophus8SE3GroupIdLi0EEERNS_8AffLightEiN5Eigen6MatrixIdLi5ELi1ELi0ELi5ELi1EEEPNS_6IOWrap15Output3DWrapperE",
           "_ZN3dso10FullSystem14traceNewCoarseEPNS_12FrameHessianE",
           "_ZN3dso10FullSystem8optimizeEi"]

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so not built and /root/reference absent")


@pytest.fixture(scope="module")
def dropin(pkg):
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "dropin"), "-s"])
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/libdropin_hip.so not built and /root/reference absent")
    return DROPIN


def _dynsyms(path, kind):
    out = subprocess.check_output(["nm", "-D", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if len(line.split()) >= 2 and line.split()[-2] == kind}


def test_adapter_defines_the_members_under_the_references_names(dropin):
    mine = _dynsyms(dropin, "T")
    theirs = _dynsyms(os.path.join(REF_DIR, "libref.so"), "T")
    for m in MEMBERS:
        assert m in mine and m in theirs, m
    # the reference's library is not bound to its own definitions at link time (-Bsymbolic would defeat the interposition) ...
    dyn = subprocess.check_output(["readelf", "-d", os.path.join(REF_DIR, "libref.so")], text=True)
    assert "SYMBOLIC" not in dyn
    # ... and calls the members through the PLT (relocations against the member symbols)
    rel = subprocess.check_output(["readelf", "-r", "-W", os.path.join(REF_DIR, "libref.so")], text=True)
    for m in MEMBERS:
        assert any("JUMP_SLO" in line and m in line for line in rel.splitlines()), m
    # the adapter needs libdmvio_hip.so and libref.so, and nothing of the product needs the reference
    need = subprocess.check_output(["readelf", "-d", dropin], text=True)
    assert "libdmvio_hip.so" in need and "libref.so" in need
    prod = subprocess.check_output(["readelf", "-d", os.path.join(ROOT, "dm-vio_amd", "lib", "libdmvio_hip.so")], text=True)
    assert "libref" not in prod and "dropin" not in prod


def _run(mode, out, frames=30):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "dropin", "run_dropin.py"), "--mode", mode, "--out", str(out), "--frames", str(frames)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return np.load(out)


def test_switched_off_the_adapter_is_transparent(dropin, tmp_path):
    """`plain` (libref.so alone) against `cpu` (adapter interposed, switched off).  The reference's initialiser is multi-threaded with dynamic chunking, so two runs of the
    SAME binary can differ in the last bits; runs are paired by the signature of what the initialiser handed over (run_dropin.py) — everything after it is deterministic."""
    plain, cpu = {}, {}
    for attempt in range(6):
        a = _run("plain", tmp_path / ("plain%d.npz" % attempt)); plain[str(a["init_signature"][0])] = a
        b = _run("cpu", tmp_path / ("cpu%d.npz" % attempt)); cpu[str(b["init_signature"][0])] = b
        common = set(plain) & set(cpu)
        if common:
            break
    assert common, "no pair of runs started from the same initialisation in 6 attempts"
    a, b = plain[sorted(common)[0]], cpu[sorted(common)[0]]
    assert b["stat_calls"].min() > 0 and b["failures"][0] == 0           # all five members were reached through the adapter
    assert b["stat_calls"][0] == 30 and b["stat_calls"][4] == len(b["opt_rmse"])
    for k in ("camToWorld", "valid", "keyframeId", "trackingRef", "aff", "opt_rmse", "opt_resInA", "opt_N", "opt_R"):
        assert np.array_equal(a[k], b[k]), k
