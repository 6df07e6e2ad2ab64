/* Back-to-back single-frame tracking calls from plain C (no interpreter latency between them): every dmvio_hip_tracker_track call runs its LM loop on the host against
 * the evaluation server (one resident kernel per call).  serverStop only POSTS the quit ticket; the next call's first tickets can overwrite it before every workgroup of
 * the previous launch has polled it.  Each launch therefore carries a session number (mailbox dword + kernel argument): a workgroup that reads another session's ticket
 * leaves instead of serving the next frame against ITS slot.  This harness alternates between two different new frames 400 times and demands the bits of the first
 * answer for each of them.  Built and run by tests/test_vio_gpu.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dmvio_hip.h"

#define W 256
#define H 256
static float texture(double x, double y) {
  return (float)(128.0 + 30.0 * sin(0.11 * x + 0.3) + 25.0 * sin(0.07 * y + 1.1) + 20.0 * sin(0.05 * (x + y)) + 15.0 * sin(0.13 * (x - 0.6 * y) + 0.7));
}
#define CHECK(call) do { if ((call) < 0) { fprintf(stderr, "%s failed: %s\n", #call, dmvio_hip_last_error()); return 1; } } while (0)

int main(void) {
  const float K4[4] = {200.0f, 200.0f, 127.5f, 127.5f};
  const double t[2][2] = {{0.02, -0.01}, {-0.015, 0.025}}, idepth = 0.5;
  float* img = (float*)malloc(sizeof(float) * W * H);
  enum { STEP = 6, MARGIN = 16, NMAX = ((W - 2 * MARGIN) / STEP + 1) * ((H - 2 * MARGIN) / STEP + 1) };
  static float u[NMAX], v[NMAX], id[NMAX], hdiF[NMAX];
  int n = 0;
  for (int y = MARGIN; y < H - MARGIN; y += STEP)
    for (int x = MARGIN; x < W - MARGIN; x += STEP) { u[n] = (float)x; v[n] = (float)y; id[n] = (float)idepth; hdiF[n] = 1e-4f; n++; }
  if (dmvio_hip_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", dmvio_hip_last_error()); return 2; }
  dmvio_hip_ctx* ctx = dmvio_hip_create(0, W, H, 3);
  if (!ctx) { fprintf(stderr, "dmvio_hip_create: %s\n", dmvio_hip_last_error()); return 1; }
  for (int s = 0; s < 3; s++) {
    const double tx = s ? t[s - 1][0] : 0.0, ty = s ? t[s - 1][1] : 0.0;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) img[y * W + x] = texture(x - K4[0] * tx * idepth, y - K4[1] * ty * idepth);
    CHECK(dmvio_hip_frame_upload(ctx, s, img));
  }
  dmvio_hip_tracker* trk = dmvio_hip_tracker_create(ctx);
  if (!trk) { fprintf(stderr, "dmvio_hip_tracker_create: %s\n", dmvio_hip_last_error()); return 1; }
  CHECK(dmvio_hip_tracker_make_k(trk, K4));
  CHECK(dmvio_hip_tracker_set_ref(trk, 0, 1.0f, 0.0, 0.0, n, u, v, id, hdiF));
  const double minRes[5] = {NAN, NAN, NAN, NAN, NAN};
  double first[2][7 + 5];
  int bad = 0;
  for (int it = 0; it < 400; it++) {
    const int s = it & 1;
    double pose7[7] = {0, 0, 0, 0, 0, 0, 1}, aff[2] = {0, 0}, lastRes[5], flow[3], Hm[64], b[8];
    int good = 0;
    CHECK(dmvio_hip_tracker_track(trk, 1 + s, 1.0f, pose7, aff, dmvio_hip_pyr_levels(ctx) - 1, minRes, lastRes, flow, Hm, b, &good));
    double cur[12];
    memcpy(cur, pose7, sizeof(pose7)); memcpy(cur + 7, lastRes, sizeof(lastRes));
    if (it < 2) {
      memcpy(first[s], cur, sizeof(cur));
      const double err = sqrt((pose7[0] - t[s][0]) * (pose7[0] - t[s][0]) + (pose7[1] - t[s][1]) * (pose7[1] - t[s][1]) + pose7[2] * pose7[2]);
      if (!good || err > 2e-3) { fprintf(stderr, "frame %d: pose not recovered (%.2e m)\n", s, err); return 3; }
    } else if (memcmp(first[s], cur, sizeof(cur)) != 0) {
      if (bad < 5) fprintf(stderr, "call %d (frame %d): translation (%.6f, %.6f, %.6f), first answer (%.6f, %.6f, %.6f)\n", it, s, pose7[0], pose7[1], pose7[2], first[s][0], first[s][1], first[s][2]);
      bad++;
    }
  }
  dmvio_hip_tracker_destroy(trk);
  dmvio_hip_destroy(ctx);
  free(img);
  if (bad) { fprintf(stderr, "%d of 398 repeated calls differ from the first answer for their frame\n", bad); return 4; }
  printf("ok: 400 back-to-back single-frame tracks, two frames alternating, every answer bit-identical to the first for its frame\n");
  return 0;
}
