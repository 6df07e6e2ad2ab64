// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY PINNED: bit / byte identical to the reference's util/Undistort.cpp (own tables from getUndistorterForFile; the OpenCV image reader replaced by oracle/ref_imagerw.cpp) and to FullSystem::printResult of a live run (oracle/_ref, tests/test_ref_pin_cpu.py::test_undistort_bitwise, ::test_print_result_bytes).
// CPU restatement of the image input edge of the hot path:
//   orc_undistort  <- PhotometricUndistorter::processFrame (src/dso/util/Undistort.cpp:214-250) + Undistort::undistort (:386-481, without the
//                     benchmark noise options)
//   orc_write_result_txt <- FullSystem::printResult (src/dso/FullSystem/FullSystem.cpp:256-298)
#include <cstring>
#include <vector>
#include <fstream>
#include <iomanip>
#include "lie.h"

extern "C" {

// raw: wOrg*hOrg pixels of `bits` (8 / 16) bits; G: response table or NULL (then data = factor * raw); vig: vignetteMapInv or NULL;
// remapX / remapY: w*h or NULL (passthrough, wOrg x hOrg == w x h); out: w*h floats
void orc_undistort(const void* raw, int bits, int wOrg, int hOrg, const float* G, const float* vig, const float* remapX, const float* remapY, int w, int h,
                   float factor, float* out) {
  const int wh = wOrg * hOrg;
  std::vector<float> data(wh);
  const unsigned char* r8 = (const unsigned char*)raw; const unsigned short* r16 = (const unsigned short*)raw;
  if (!G) { for (int i = 0; i < wh; i++) data[i] = factor * (bits == 8 ? (float)r8[i] : (float)r16[i]); }
  else {
    for (int i = 0; i < wh; i++) data[i] = G[bits == 8 ? (int)r8[i] : (int)r16[i]];
    if (vig) for (int i = 0; i < wh; i++) data[i] *= vig[i];
  }
  if (!remapX) { memcpy(out, data.data(), sizeof(float) * (size_t)w * h); return; }
  for (int idx = w * h - 1; idx >= 0; idx--) {
    float xx = remapX[idx], yy = remapY[idx];
    if (xx < 0) out[idx] = 0;
    else {
      const int xxi = xx, yyi = yy;
      xx -= xxi; yy -= yyi;
      const float xxyy = xx * yy;
      const float* src = data.data() + xxi + yyi * wOrg;
      out[idx] = xxyy * src[1 + wOrg] + (yy - xxyy) * src[wOrg] + (xx - xxyy) * src[1] + (1 - xx - yy + xxyy) * src[0];
    }
  }
}

// FullSystem::printResult (FullSystem.cpp:256-298); std::setprecision(15) on a default-format ostream == "%.15g"
int orc_write_result_txt(const char* path, int n, const double* timestamps, const double* camToWorld7, const unsigned char* pose_valid,
                         const int* tracking_ref, const double* camToTrackingRef7, const double* firstPose7) {
  std::ofstream out(path);
  if (!out) return -1;
  out << std::setprecision(15);
  auto from7 = [](const double* p) { orc::SE3 T; T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2]; T.q = orc::qimport(orc::Quat{p[6], p[3], p[4], p[5]}); return T; };
  const orc::SE3 firstInv = orc::se3Inv(from7(firstPose7));
  for (int i = 0; i < n; i++) {
    if (pose_valid && !pose_valid[i]) continue;
    orc::SE3 c2w = from7(camToWorld7 + 7 * i);
    if (tracking_ref && tracking_ref[i] >= 0) c2w = orc::se3Mul(from7(camToWorld7 + 7 * tracking_ref[i]), from7(camToTrackingRef7 + 7 * i));
    const orc::SE3 T = orc::se3Mul(firstInv, c2w);
    out << timestamps[i] << " " << T.t[0] << " " << T.t[1] << " " << T.t[2] << " " << T.q.x << " " << T.q.y << " " << T.q.z << " " << T.q.w << "\n";
  }
  return 0;
}

}  // extern "C"
