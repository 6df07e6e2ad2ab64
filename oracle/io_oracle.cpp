// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED.
// CPU restatement of the image input edge of the hot path:
//   orc_undistort  <- PhotometricUndistorter::processFrame (src/dso/util/Undistort.cpp:214-250) + Undistort::undistort (:386-481, without the
//                     benchmark noise options)
#include <cstring>
#include <vector>

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

}  // extern "C"
