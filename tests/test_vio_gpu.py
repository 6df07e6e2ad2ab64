"""The reference's DEFAULT tracking path on the GPU: trackNewestCoarse with every LM step handed to the host
(setting_useIMU, CoarseTracker.cpp:612-637) — dmvio_hip_tracker_track_vio / dmvio_hip::CoarseTracker::trackNewestCoarse(..., hooks)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IDENT = np.array([0, 0, 0, 0, 0, 0, 1.0])


@pytest.fixture(scope="module")
def setup(pkg, oracle, synth, gpu_required):
    w = h = 512
    case = synth.tracking_case(w, h, n_ref=2000, n_frames=3, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=8)
    trk = pkg.CoarseTrackerHip(ctx)
    trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"])
    for k, f in enumerate(case["frames"]):
        ctx.frame_upload(1 + k, f["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    dIr, _ = oracle.make_images(case["ref_img"], w, h)
    T = oracle.Tracker(w, h)
    T.make_k(case["K4"])
    T.set_ref(dIr, case["u"], case["v"], case["idepth"], case["hdiF"])
    return dict(case=case, ctx=ctx, trk=trk, T=T, w=w, h=h)


@pytest.mark.parametrize("frame", [0, 1, 2])
def test_default_update_reproduces_the_device_resident_lm(setup, frame):
    """With the library's visual-only step as the update the host-driven loop walks the same path as k_track_lm: the evaluations are
    split over the same number of workgroups (bit-identical sums), so poses / residuals agree to the last bits of the 8x8 solve."""
    trk = setup["trk"]
    a = trk.trackNewestCoarse(1 + frame, IDENT, [0.0, 0.0])
    b = trk.trackNewestCoarseVIO(1 + frame, IDENT, [0.0, 0.0])
    assert a["good"] == b["good"] is True
    assert np.abs(a["pose7"] - b["pose7"]).max() < 1e-9
    assert np.abs(a["aff"] - b["aff"]).max() < 1e-7
    assert np.allclose(a["lastResiduals"][:4], b["lastResiduals"][:4], rtol=1e-9)
    assert np.allclose(a["H"], b["H"], rtol=1e-7, atol=1e-9 * np.abs(a["H"]).max()) and np.allclose(a["b"], b["b"], rtol=1e-6, atol=1e-9 * np.abs(a["b"]).max())
    assert b["n_evals"] == a["iterations"] + 4   # one initial evaluation per level + one per LM iteration


def test_python_update_callback_sees_the_reference_hand_off(setup, oracle):
    """A computeCoarseUpdate written by the caller (here: the oracle's LDLT + SE3 exp in Python) receives exactly calcGSSSE's scaled
    H, b; acceptCoarseUpdate fires on every accepted step; addVisualToCoarseGraph gets the final level-0 system."""
    trk, T = setup["trk"], setup["T"]
    calls = []
    accepts = []
    vis = []

    def update(H, b, extrapFac, lam, pose_cur, aff_cur):
        calls.append((H.copy(), b.copy(), lam, extrapFac, pose_cur.copy(), aff_cur.copy()))
        Hl = H.copy()
        Hl[np.diag_indices(8)] *= float(np.float32(1) + np.float32(lam))   # (1+lambda) is a float expression in the reference (CoarseTracker.cpp:602)
        inc = oracle.ldlt_solve(Hl, -b) * extrapFac
        pose_new = oracle.se3_mul(oracle.se3_exp(inc[:6]), pose_cur)
        return pose_new, inc[6], inc[7], np.linalg.norm(inc)

    r = trk.trackNewestCoarseVIO(1, IDENT, [0.0, 0.0], update=update, accept=lambda: accepts.append(len(calls)), visual=lambda H, b, good: vis.append((H, b, good)))
    ref = trk.trackNewestCoarse(1, IDENT, [0.0, 0.0])
    assert r["good"] and len(calls) == ref["iterations"] and 3 <= len(accepts) <= len(calls)
    assert np.abs(r["pose7"] - ref["pose7"]).max() < 1e-7 and np.abs(r["aff"] - ref["aff"]).max() < 1e-5
    assert len(vis) == 1 and vis[0][2] is True and np.array_equal(vis[0][0], r["H"]) and np.array_equal(vis[0][1], r["b"])
    # every system handed over is the device evaluation at the pose the callback was told is current
    for (H, b, lam, ef, pose_cur, aff_cur) in calls[:6]:
        assert H.shape == (8, 8) and np.allclose(H, H.T) and np.all(np.diag(H) > 0)
    # against the CPU oracle at the first hand-over (coarsest level, start pose): same system to summation-order accuracy
    dIn, _ = oracle.make_images(setup["case"]["frames"][0]["img"], setup["w"], setup["h"])
    T.set_new(dIn)
    lvl = setup["ctx"].levels - 1
    T.calc_res(lvl, IDENT, [0.0, 0.0])
    Ho, bo = T.calc_gs(lvl, [0.0, 0.0])
    assert np.allclose(calls[0][0], Ho, rtol=2e-5, atol=2e-5 * np.abs(Ho).max()) and np.allclose(calls[0][1], bo, rtol=2e-5, atol=2e-5 * np.abs(bo).max())


def test_against_the_references_own_vio_branch(setup):
    """The reference's sources (oracle/_ref/libref.so), setting_useIMU = true with a coarse-initialised IMU facade whose
    computeCoarseUpdate is the visual step: same number of hand-overs, same accepted steps, same final pose within 1e-6."""
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_py as R
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not present")
    case, trk = setup["case"], setup["trk"]
    RT = R.Tracker(setup["w"], setup["h"], case["K4"])
    RT.set_ref(case["ref_img"], case["u"], case["v"], case["idepth"], case["hdiF"])
    for k in range(3):
        RT.set_new(case["frames"][k]["img"])
        rr = RT.track(IDENT, [0.0, 0.0], vio=True)
        accepts = []
        calls = []
        g = trk.trackNewestCoarseVIO(1 + k, IDENT, [0.0, 0.0], update=lambda H, b, ef, lam, pc, ac: (calls.append(lam), trk.coarse_update_visual(H, b, ef, lam, pc))[1],
                                     accept=lambda: accepts.append(1))
        assert g["good"] == rr["good"]
        assert len(calls) == rr["vio_calls"] and len(accepts) == rr["vio_accepts"]
        assert np.allclose([c for c in calls], rr["vio_log"][:, 72], rtol=0, atol=0)     # the same lambda schedule = the same accept / reject sequence
        assert np.abs(g["pose7"] - rr["pose7"]).max() < 1e-6 and np.abs(g["aff"] - rr["aff"]).max() < 1e-4
        assert np.allclose(g["lastResiduals"][:4], rr["lastResiduals"][:4], rtol=1e-5)


def test_cpp_hooks_compile_and_track(setup, pkg, tmp_path):
    """include/dmvio_hip.hpp: CoarseTracker::trackNewestCoarse(..., CoarseIMUHooks) from C++11, hooks = lambdas (visual-only fallback inside)."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "vio_hooks.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dmvio_hip.hpp"
int main(int argc, char** argv) {
  const int w = 256, h = 256;
  dmvio_hip::FrameStore fs(0, w, h, 2);
  if (!fs.valid()) { std::printf("no device: %s\n", dmvio_hip::lastError().c_str()); return 2; }
  std::vector<float> ref(w * h), cur(w * h);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    ref[y * w + x] = 128 + 60 * std::sin(0.11f * x) * std::cos(0.07f * y) + 30 * std::sin(0.031f * (x + 2 * y));
    cur[y * w + x] = 128 + 60 * std::sin(0.11f * (x + 1.5f)) * std::cos(0.07f * (y - 0.75f)) + 30 * std::sin(0.031f * ((x + 1.5f) + 2 * (y - 0.75f)));
  }
  fs.makeImages(0, ref.data()); fs.makeImages(1, cur.data());
  dmvio_hip::CoarseTracker trk(fs);
  trk.makeK(0.2f * w, 0.2f * h, 0.499f * w - 0.5f, 0.499f * h - 0.5f);
  std::vector<dmvio_hip::RefPoint> pts;
  for (int y = 12; y < h - 12; y += 6) for (int x = 12; x < w - 12; x += 6) { dmvio_hip::RefPoint p = {(float)x, (float)y, 0.3f, 1e-4f}; pts.push_back(p); }
  trk.setCoarseTrackingRef(0, 0, 1.0f, dmvio_hip::AffLight(0, 0), pts);
  const double minRes[5] = {NAN, NAN, NAN, NAN, NAN};
  dmvio_hip::SE3 T1, T2; dmvio_hip::AffLight a1, a2;
  const bool g1 = trk.trackNewestCoarse(1, 1.0f, T1, a1, fs.pyrLevelsUsed() - 1, minRes);
  int updates = 0, accepts = 0, visuals = 0;
  dmvio_hip::CoarseTracker::CoarseIMUHooks hooks;
  hooks.computeCoarseUpdate = [&](const double* H, const double* b, float extrapFac, float lambda, const dmvio_hip::SE3& cur, double& incA, double& incB, double& incNorm) {
    updates++;
    double p[7], q[7]; cur.toPose7(p);
    dmvio_hip_coarse_update_visual(nullptr, H, b, extrapFac, lambda, p, q, &incA, &incB, &incNorm);
    dmvio_hip::SE3 n; n.fromPose7(q); return n;
  };
  hooks.acceptCoarseUpdate = [&]() { accepts++; };
  hooks.addVisualToCoarseGraph = [&](const double*, const double*, bool) { visuals++; };
  const bool g2 = trk.trackNewestCoarse(1, 1.0f, T2, a2, fs.pyrLevelsUsed() - 1, minRes, hooks);
  double d = 0; for (int i = 0; i < 3; i++) d += (T1.t[i] - T2.t[i]) * (T1.t[i] - T2.t[i]);
  std::printf("good %d %d updates %d accepts %d visuals %d evals %d dpos %.3e\n", (int)g1, (int)g2, updates, accepts, visuals, trk.lastEvaluations, std::sqrt(d));
  return (g1 == g2 && updates > 3 && accepts > 0 && visuals == 1 && std::sqrt(d) < 1e-8) ? 0 : 1;
}
''')
    exe = tmp_path / "vio_hooks"
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", os.path.join(ROOT, "dm-vio_amd", "lib"), "-ldmvio_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "dm-vio_amd", "lib")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_evaluation_server_restarts_after_a_slow_callback(setup):
    """The resident evaluation kernel leaves by itself after 5 ms without a request; a computeCoarseUpdate that takes longer (a factor-graph solve) finds it gone, the library
    starts it again and the call goes on: same result as with a fast callback."""
    import time
    trk = setup["trk"]
    ref = trk.trackNewestCoarseVIO(1, IDENT, [0.0, 0.0])
    calls = []

    def slow_update(H, b, ef, lam, pc, ac):
        calls.append(lam)
        if len(calls) in (2, 5):
            time.sleep(0.02)
        return trk.coarse_update_visual(H, b, ef, lam, pc)

    r = trk.trackNewestCoarseVIO(1, IDENT, [0.0, 0.0], update=slow_update)
    assert len(calls) >= 6 and r["good"] == ref["good"]
    assert np.array_equal(r["pose7"], ref["pose7"]) and np.array_equal(r["lastResiduals"], ref["lastResiduals"], equal_nan=True)
    # and a batch launched right after a server session runs behind it on the same stream
    b = trk.track_batch([1, 2, 3], [IDENT] * 3, [(0.0, 0.0)] * 3)
    assert b["good"].all() and np.abs(b["pose7"][0] - ref["pose7"]).max() < 1e-14


def test_back_to_back_single_frame_tracks_from_c_keep_their_sessions_apart(pkg, gpu_required, tmp_path):
    """tests/server_session_harness.c: 400 dmvio_hip_tracker_track calls in a C loop, two different new frames alternating — the evaluation server of one call must never
    serve a request of the next (each launch carries a session number)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(pkg.LIB_PATH)
    exe = tmp_path / "server_session_harness"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", os.path.join(root, "tests", "server_session_harness.c"), "-I" + os.path.dirname(pkg.INCLUDE_PATH),
                           "-o", str(exe), "-L" + libdir, "-ldmvio_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
