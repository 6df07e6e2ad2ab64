// The same frame as c_abi_demo.c through the C++ mirror of the reference's member surface (include/dmvio_hip.hpp): code that reads like
// the tracking thread of FullSystem — makeImages, makeK, setCoarseTrackingRef, trackNewestCoarse — with the wall and the motion known.
//
//   g++ -std=c++11 -O2 examples/cpp_adapter_demo.cpp -Iinclude -Ldm-vio_amd/lib -ldmvio_hip -o cpp_adapter_demo
#include <cmath>
#include <cstdio>
#include <vector>
#include "dmvio_hip.hpp"

static float texture(double x, double y) {
  return (float)(128.0 + 30.0 * std::sin(0.11 * x + 0.3) + 25.0 * std::sin(0.07 * y + 1.1) + 20.0 * std::sin(0.05 * (x + y)) + 15.0 * std::sin(0.13 * (x - 0.6 * y) + 0.7));
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
  if (coarseTracker.trackNewestCoarse(7, 1.0f, T, a, 3, achievedRes)) { std::fprintf(stderr, "bad slot accepted\n"); return 4; }
  std::printf("ok: translation error %.2e m; bad slot -> false (%s)\n", err, dmvio_hip::lastError().c_str());
  return 0;
}
