// The same frame as c_abi_demo.c through the C++ mirror of the reference's member surface (include/dmvio_hip.hpp): code that reads like
// the tracking thread of FullSystem — makeImages, makeK, setCoarseTrackingRef, trackNewestCoarse (visual-only and with the IMU hooks of the reference's default
// branch) — with the wall and the motion known.
//
//   g++ -std=c++11 -O2 examples/cpp_adapter_demo.cpp -Iinclude -Ldm-vio_amd/lib -ldmvio_hip -o cpp_adapter_demo
#include <cmath>
#include <cstdio>
#include <vector>
#include "dmvio_hip.hpp"

static float texture(double x, double y) {
  return (float)(128.0 + 30.0 * std::sin(0.11 * x + 0.3) + 25.0 * std::sin(0.07 * y + 1.1) + 20.0 * std::sin(0.05 * (x + y)) + 15.0 * std::sin(0.13 * (x - 0.6 * y) + 0.7));
}

// The mapping thread's side: three keyframes looking at the same wall from x = 0, 5 cm, 10 cm; points hosted in the first two with their
// inverse depths 8 % off and the last pose 4 mm off — FullSystem::optimize through dmvio_hip::WindowOptimizer brings the photometric energy down.
static int mappingDemo() {
  const int w = 256, h = 256, F = 3;
  const double fx = 200, fy = 200, cx = 127.5, cy = 127.5, idepth = 0.5;
  dmvio_hip::FrameStore frames(0, w, h, F);
  if (!frames.valid()) return 2;
  std::vector<float> img(w * h);
  std::vector<dmvio_hip::KeyFrame> frameHessians(F);
  for (int k = 0; k < F; k++) {
    const double camX = 0.05 * k;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) img[y * w + x] = texture(x + fx * camX * idepth, y);
    if (!frames.makeImages(k, img.data())) return 1;
    frameHessians[k].slot = k; frameHessians[k].frameID = k; frameHessians[k].ab_exposure = 1.0f;
    frameHessians[k].worldToCam_evalPT.t[0] = -camX + (k == 2 ? 0.004 : 0.0);      // the newest pose is off
  }
  // colour / weights of the pattern pixels as ImmaturePoint's constructor computes them (ImmaturePoint.cpp:34-62)
  dmvio_hip_immature* imm = dmvio_hip_immature_create(frames.handle(), 4096);
  if (!imm) return 1;
  std::vector<int> pu, pv;
  for (int y = 24; y < h - 24; y += 8) for (int x = 24; x < w - 24; x += 8) { pu.push_back(x); pv.push_back(y); }
  const int per = (int)pu.size();
  for (int host = 0; host < 2; host++) if (dmvio_hip_immature_add_points(imm, host, host, per, pu.data(), pv.data()) < 0) return 1;
  const int N = dmvio_hip_immature_count(imm);
  std::vector<float> u(N), v(N), col(8 * N), wts(8 * N);
  std::vector<int> tag(N);
  if (dmvio_hip_immature_get_static(imm, u.data(), v.data(), tag.data(), col.data(), wts.data(), nullptr, nullptr) < 0) return 1;
  dmvio_hip_immature_destroy(imm);
  std::vector<dmvio_hip::ActivePoint> points(N);
  for (int i = 0; i < N; i++) {
    dmvio_hip::ActivePoint& p = points[i];
    p.host = tag[i]; p.u = u[i]; p.v = v[i]; p.idepth = (float)(idepth * 1.08); p.hasDepthPrior = false;
    for (int k = 0; k < 8; k++) { p.color[k] = col[8 * i + k]; p.weights[k] = wts[8 * i + k]; }
    for (int t = 0; t < F; t++) if (t != p.host) p.targets.push_back(t);
  }
  dmvio_hip::WindowOptimizer ef(frames);
  if (!ef.setWindow(frameHessians, fx, fy, cx, cy) || !ef.setPoints(points)) { std::fprintf(stderr, "window set-up failed: %s\n", dmvio_hip::lastError().c_str()); return 1; }
  const float rmse = ef.optimize(6);
  const double E0 = ef.energyTrace[0] + ef.energyTrace[1] + ef.energyTrace[2];
  std::vector<float> id;
  ef.idepths(id);
  double mean = 0; for (int i = 0; i < N; i++) mean += id[i]; mean /= N;
  std::printf("optimize: %d points, %d iterations, rmse %.3f, energy %.1f -> %.1f, mean idepth %.4f (start %.4f)\n", N, ef.lastIterations, rmse, E0, ef.lastEnergy, mean, idepth * 1.08);
  if (!(rmse >= 0) || !(ef.lastEnergy < 0.5 * E0)) { std::fprintf(stderr, "bundle adjustment did not reduce the energy\n"); return 5; }
  std::printf("ok: energy reduced to %.0f %%\n", 100.0 * ef.lastEnergy / E0);
  // the reference's default branch: every solve handed to a hook (here the library's own LDLT, so the result must be the one above)
  dmvio_hip::WindowOptimizer ef2(frames);
  if (!ef2.setWindow(frameHessians, fx, fy, cx, cy) || !ef2.setPoints(points)) return 1;
  dmvio_hip_ba_callbacks hooks = dmvio_hip_ba_callbacks();
  hooks.computeBAUpdate = dmvio_hip_ba_hook_ldlt;
  dmvio_hip_ba_vio_options opt = dmvio_hip_ba_vio_options();
  opt.coarseTrackingWasGood = 1; opt.minOptIterations = -1; opt.resInA_at_entry = -1;
  const float rmse2 = ef2.optimize(6, hooks, opt);
  if (rmse2 != rmse || ef2.lastEnergy != ef.lastEnergy) { std::fprintf(stderr, "hook path differs: rmse %.9g vs %.9g, energy %.9g vs %.9g\n", rmse2, rmse, ef2.lastEnergy, ef.lastEnergy); return 6; }
  std::printf("ok: the same window through the computeBAUpdate hook: identical result\n");
  // the window graph kept resident (EnergyFunctional's own mutators, forwarded one by one): inserted with one residual too many per point and in another order,
  // then brought to the same graph by dropResidual — the LAST residual takes the dropped one's index, as in EnergyFunctional::dropResidual —; same result
  dmvio_hip::WindowGraph graph;
  for (int f = 0; f < F; f++) graph.insertFrame();
  for (int i = 0; i < N; i++) {
    const dmvio_hip::ActivePoint& p = points[i];
    const int idx = graph.insertPoint(p);
    if (idx < 0) return 1;
    // wanted: [t0, t1, ..., t(k-1)].  Inserted: a placeholder at index 0, then t1 ... t(k-1), then t0; dropping index 0 moves the last residual (t0) into its place
    const size_t k = p.targets.size();
    if (k == 0) continue;
    graph.insertResidual(p.host, idx, p.targets[0]);
    for (size_t r = 1; r < k; r++) graph.insertResidual(p.host, idx, p.targets[r]);
    graph.insertResidual(p.host, idx, p.targets[0]);
    if (!graph.dropResidual(p.host, idx, 0)) return 1;
  }
  if (graph.nPoints() != N) return 1;
  dmvio_hip::WindowOptimizer ef3(frames);
  if (!ef3.setWindow(frameHessians, fx, fy, cx, cy) || !ef3.setPoints(graph)) { std::fprintf(stderr, "window graph hand-over failed: %s\n", dmvio_hip::lastError().c_str()); return 1; }
  const float rmse3 = ef3.optimize(6);
  if (rmse3 != rmse || ef3.lastEnergy != ef.lastEnergy) { std::fprintf(stderr, "window graph path differs: rmse %.9g vs %.9g, energy %.9g vs %.9g\n", rmse3, rmse, ef3.lastEnergy, ef.lastEnergy); return 7; }
  std::vector<float> id3;
  ef3.idepths(id3);
  if (!graph.setIdepths(id3)) return 1;
  std::printf("ok: the same window from the resident graph (%d residuals through insertResidual / dropResidual): identical result\n", graph.nResiduals());
  return 0;
}

int main() {
  const int w = 256, h = 256;
  const float fx = 200, fy = 200, cx = 127.5f, cy = 127.5f;
  const double tx = 0.02, ty = -0.01, idepth = 0.5;
  std::vector<float> ref(w * h), cur(w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) { ref[y * w + x] = texture(x, y); cur[y * w + x] = texture(x - fx * tx * idepth, y - fy * ty * idepth); }
  std::vector<dmvio_hip::RefPoint> points;
  for (int y = 16; y < h - 16; y += 6)
    for (int x = 16; x < w - 16; x += 6) points.push_back(dmvio_hip::RefPoint{(float)x, (float)y, (float)idepth, 1e-4f});

  dmvio_hip::FrameStore frames(0, w, h, 2);
  if (!frames.valid()) { std::fprintf(stderr, "no device: %s\n", dmvio_hip::lastError().c_str()); return 2; }
  dmvio_hip::CoarseTracker coarseTracker(frames);
  if (!frames.makeImages(0, ref.data()) || !frames.makeImages(1, cur.data()) || !coarseTracker.makeK(fx, fy, cx, cy) ||
      !coarseTracker.setCoarseTrackingRef(/*slot*/ 0, /*frameID*/ 0, /*ab_exposure*/ 1.0f, dmvio_hip::AffLight(0, 0), points)) {
    std::fprintf(stderr, "set-up failed: %s\n", dmvio_hip::lastError().c_str());
    return 1;
  }
  dmvio_hip::SE3 lastF_2_fh;            // initial guess: identity (FullSystem::trackNewCoarse's first try)
  dmvio_hip::AffLight aff_g2l;
  const double achievedRes[5] = {NAN, NAN, NAN, NAN, NAN};
  const bool trackingIsGood = coarseTracker.trackNewestCoarse(1, 1.0f, lastF_2_fh, aff_g2l, frames.pyrLevelsUsed() - 1, achievedRes);
  std::printf("trackingIsGood %d  t = (%.5f, %.5f, %.5f)  lastResiduals[0] = %.3f  flow = (%.3f, %.3f, %.3f)\n", (int)trackingIsGood, lastF_2_fh.t[0], lastF_2_fh.t[1],
              lastF_2_fh.t[2], coarseTracker.lastResiduals[0], coarseTracker.lastFlowIndicators[0], coarseTracker.lastFlowIndicators[1], coarseTracker.lastFlowIndicators[2]);
  const double err = std::sqrt((lastF_2_fh.t[0] - tx) * (lastF_2_fh.t[0] - tx) + (lastF_2_fh.t[1] - ty) * (lastF_2_fh.t[1] - ty) + lastF_2_fh.t[2] * lastF_2_fh.t[2]);
  if (!trackingIsGood || err > 2e-3) { std::fprintf(stderr, "pose not recovered (%.2e m): %s\n", err, dmvio_hip::lastError().c_str()); return 3; }
  // an out-of-range slot reads as "tracking failed", not as a crash or an exception
  dmvio_hip::SE3 T; dmvio_hip::AffLight a;
  if (coarseTracker.trackNewestCoarse(7, 1.0f, T, a, frames.pyrLevelsUsed() - 1, achievedRes)) { std::fprintf(stderr, "bad slot accepted\n"); return 4; }
  std::printf("ok: translation error %.2e m; bad slot -> false (%s)\n", err, dmvio_hip::lastError().c_str());

  // The reference's DEFAULT branch (setting_useIMU, CoarseTracker.cpp:612-637): every LM step comes from the host — here a stand-in for dmvio::IMUIntegration that
  // computes the visual-only step (what the reference runs until the IMU is initialised) and counts what the tracker calls on it.
  struct FakeIMUIntegration { int updates = 0, accepts = 0, visuals = 0; bool lastGood = false; } imu;
  dmvio_hip::CoarseTracker::CoarseIMUHooks hooks;
  hooks.computeCoarseUpdate = [&imu](const double* H, const double* b, float extrapFac, float lambda, const dmvio_hip::SE3& refToNew_current, double& incA, double& incB,
                                     double& incNorm) {
    double cur[7], nxt[7];
    refToNew_current.toPose7(cur);
    dmvio_hip_coarse_update_visual(nullptr, H, b, extrapFac, lambda, cur, nxt, &incA, &incB, &incNorm);
    imu.updates++;
    dmvio_hip::SE3 r; r.fromPose7(nxt);
    return r;
  };
  hooks.acceptCoarseUpdate = [&imu]() { imu.accepts++; };
  hooks.addVisualToCoarseGraph = [&imu](const double*, const double*, bool trackingGood) { imu.visuals++; imu.lastGood = trackingGood; };
  dmvio_hip::SE3 T_vio; dmvio_hip::AffLight aff_vio;
  const bool goodVio = coarseTracker.trackNewestCoarse(1, 1.0f, T_vio, aff_vio, frames.pyrLevelsUsed() - 1, achievedRes, hooks);
  double dT = 0;
  for (int i = 0; i < 3; i++) dT = std::fmax(dT, std::fabs(T_vio.t[i] - lastF_2_fh.t[i]));
  std::printf("hand-off: good %d, %d computeCoarseUpdate, %d acceptCoarseUpdate, %d addVisualToCoarseGraph, %d evaluations, max |dt| vs the visual-only call %.1e\n", (int)goodVio,
              imu.updates, imu.accepts, imu.visuals, coarseTracker.lastEvaluations, dT);
  if (!goodVio || imu.updates < 4 || imu.accepts < 1 || imu.accepts > imu.updates || imu.visuals != 1 || !imu.lastGood || dT > 1e-12) {
    std::fprintf(stderr, "hand-off path disagrees with the visual-only path\n");
    return 6;
  }
  return mappingDemo();
}
