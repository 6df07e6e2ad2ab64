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
MEMBERS = ["_ZN3dso10FullSystem25activatePointsMT_ReductorEPSt6vectorIPNS_12PointHessianESaIS3_EEPS1_IPNS_13ImmaturePointESaIS8_EEiiPN5Eigen6MatrixIdLi10ELi1ELi0ELi10ELi1EEEi",
           "_ZN3dso17CoarseInitializer12calcResAndGSEiRN5Eigen6MatrixIfLi8ELi8ELi0ELi8ELi8EEERNS2_IfLi8ELi1ELi0ELi8ELi1EEES4_S6_RKN6Sophus8SE3GroupIdLi0EEENS_8AffLightEb",
           "_ZN3dso12FrameHessian10makeImagesEPfPNS_12CalibHessianE",
           "_ZN3dso13CoarseTracker20setCoarseTrackingRefESt6vectorIPNS_12FrameHessianESaIS3_EE",
           "_ZN3dso13CoarseTracker17trackNewestCoarseEPNS_12FrameHessianERN6Sophus8SE3GroupIdLi0EEERNS_8AffLightEiN5Eigen6MatrixIdLi5ELi1ELi0ELi5ELi1EEEPNS_6IOWrap15Output3DWrapperE",
           "_ZN3dso10FullSystem14traceNewCoarseEPNS_12FrameHessianE",
           "_ZN3dso10FullSystem8optimizeEi",
           # round 5: the try loop of trackNewCoarse as a member (one device batch for the motion hypotheses)
           "_ZN3dso10FullSystem14trackNewCoarseEPNS_12FrameHessianEPN6Sophus8SE3GroupIdLi0EEE",
           # round 4: EnergyFunctional's graph mutators (forwarded to the resident window graph) and marginalizePointsF (a real member on top of it)
           "_ZN3dso16EnergyFunctional11insertFrameEPNS_12FrameHessianEPNS_12CalibHessianE", "_ZN3dso16EnergyFunctional11insertPointEPNS_12PointHessianE",
           "_ZN3dso16EnergyFunctional14insertResidualEPNS_18PointFrameResidualE", "_ZN3dso16EnergyFunctional12dropResidualEPNS_10EFResidualE",
           "_ZN3dso16EnergyFunctional11removePointEPNS_7EFPointE", "_ZN3dso16EnergyFunctional16marginalizeFrameEPNS_7EFFrameE",
           "_ZN3dso16EnergyFunctional18marginalizePointsFEv"]

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
        assert any(("JUMP_SLO" in line or "GLOB_DAT" in line) and m in line for line in rel.splitlines()), m    # (GLOB_DAT: a member whose address is also taken, bound through the GOT)
    # the adapter needs libdmvio_hip.so and libref.so, and nothing of the product needs the reference
    need = subprocess.check_output(["readelf", "-d", dropin], text=True)
    assert "libdmvio_hip.so" in need and "libref.so" in need
    prod = subprocess.check_output(["readelf", "-d", os.path.join(ROOT, "dm-vio_amd", "lib", "libdmvio_hip.so")], text=True)
    assert "libref" not in prod and "dropin" not in prod


def _run(mode, out, *extra):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "dropin", "run_dropin.py"), "--mode", mode, "--out", str(out)] + list(extra),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return np.load(out)


def test_switched_off_the_adapter_is_transparent(dropin, tmp_path):
    """`plain` (libref.so alone in the process) against `cpu` (adapter interposed, switched off): makeImages, traceNewCoarse, optimize, setCoarseTrackingRef and
    trackNewestCoarse of the reference reached through the adapter's forwarding definitions give bit for bit what they give without it (one window, the members called one by
    one through oracle/ref_py.py: single-threaded code, deterministic)."""
    a = _run("plain", tmp_path / "plain.npz", "--window")
    b = _run("cpu", tmp_path / "cpu.npz", "--window")
    assert b["stat_calls"][:5].min() > 0 and b["failures"][0] == 0       # the five members this sequence uses were reached through the adapter
    assert b["stat_calls"][0] >= 4 and b["stat_calls"][4] == 1
    for k in ("rmse", "poses", "idepth", "imm_min", "imm_max", "imm_status", "track_pose", "track_res"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k


def test_whole_run_through_the_switched_off_adapter(dropin, tmp_path):
    """The reference's whole FullSystem with the adapter interposed and switched off, the initialiser's calcResAndGS through the oracle's single-threaded restatement (the
    reference's own is multi-threaded with dynamic chunking: two runs of the SAME binary differ in the last bits, and with them every discrete decision downstream — see
    tests/dropin/run_dropin.py): deterministic, initialises, never loses track, every replaced member on the call path."""
    a = _run("cpu", tmp_path / "a.npz", "--init", "seq", "--frames", "30")
    b = _run("cpu", tmp_path / "b.npz", "--init", "seq", "--frames", "30")
    assert a["initialized"][-1] and not a["lost"][-1] and len(a["opt_rmse"]) >= 2
    assert a["stat_calls"].min() > 0 and a["stat_calls"][0] == 30 and a["stat_calls"][4] == len(a["opt_rmse"])
    for k in ("camToWorld", "opt_rmse", "opt_N", "opt_R", "init_signature"):
        assert np.array_equal(a[k], b[k]), k


def test_real_time_mode_through_the_switched_off_adapter(dropin, tmp_path):
    """FullSystem(linearizeOperation = false): frames arrive on one thread (tracking), the reference's own mapping thread makes the keyframes.  The adapter's definitions are then
    entered from two threads at once (makeImages / trackNewestCoarse on one, makeKeyFrame / makeNonKeyFrame / traceNewCoarse / optimize / setCoarseTrackingRef on the other);
    switched off they only forward, but its bookkeeping (slots, timers, pending images) runs.  Timing decides which frames become keyframes: the run is compared with the
    linearised one loosely."""
    a = _run("cpu", tmp_path / "lin.npz", "--init", "seq")
    b = _run("cpu", tmp_path / "rt.npz", "--init", "ref", "--realtime", "60")
    assert b["initialized"][-1] and not b["lost"][-1] and b["failures"][0] == 0
    assert b["stat_calls"][0] == len(b["valid"]) and b["stat_calls"][4] >= 3 and b["stat_calls"][1] >= 3       # frames, keyframe optimisations, tracking references
    v = (a["valid"] != 0) & (b["valid"] != 0)
    d = a["camToWorld"][v, :3] - b["camToWorld"][v, :3]
    assert v.sum() >= 40 and np.sqrt((d ** 2).sum(1).mean()) < 2e-2


def test_the_references_default_vio_configuration_runs_live_against_the_stand_in(dropin, tmp_path):
    """setting_useIMU / setting_useGTSAMIntegration ON — what dmvio_dataset runs by default — with the live stand-in for the absent IMU / GTSAM side behind the facade's hooks
    (oracle/ref_glue.cpp: VioStandIn; keyframe bookkeeping restated from src/IMU/IMUIntegration.cpp in oracle/ref_shim/IMU/IMUIntegration.hpp): the reference's own
    FullSystem initialises, walks prepareKeyframe / keyframeCreated / initCoarseGraph, switches trackNewestCoarse to its computeCoarseUpdate branch after three keyframe
    optimisations and runs every optimize through computeBAUpdate / getBAEnergy / acceptBAUpdate / canBreak — deterministically (this is the all-CPU side of
    tests/test_dropin_gpu.py's VIO comparisons), and differently from the visual-only run (the stand-in's factors really enter the solves)."""
    a = _run("cpu", tmp_path / "a.npz", "--init", "seq", "--vio")
    b = _run("cpu", tmp_path / "b.npz", "--init", "seq", "--vio")
    v = _run("cpu", tmp_path / "v.npz", "--init", "seq")
    assert a["initialized"][-1] and not a["lost"][-1] and len(a["opt_rmse"]) >= 6
    c = a["vio_counters"]
    # addIMUData per tracked frame; computeCoarseUpdate / acceptCoarseUpdate / addVisualToCoarseGraph; computeBAUpdate = Gauss-Newton iterations; one postOptimization per keyframe
    assert c[0] >= 50 and c[1] > 100 and c[2] > 100 and c[3] >= 50 and c[5] >= len(a["opt_rmse"]) and c[7] >= len(a["opt_rmse"]) and c[10] == len(a["opt_rmse"]) and c[15] == 1, c
    for k in ("camToWorld", "opt_rmse", "opt_N", "opt_R", "vio_counters"):
        assert np.array_equal(a[k], b[k]), k
    ok = (a["valid"] != 0) & (v["valid"] != 0)
    d = a["camToWorld"][ok, :3] - v["camToWorld"][ok, :3]
    assert 1e-6 < np.sqrt((d ** 2).sum(1).mean()) < 5e-3       # another estimator than the visual-only one, on the same trajectory
