/* Plain-C use of libdmvio_hip: the calls a maintainer's adapter makes for one tracked frame (INTEGRATION.md sections 2-3), on a synthetic
 * fronto-parallel plane so that the answer is known — a camera translation of (2 cm, -1 cm, 0) in front of a textured wall 2 m away.
 *
 *   gcc -std=c99 -O2 examples/c_abi_demo.c -Iinclude -Ldm-vio_amd/lib -ldmvio_hip -lm -o c_abi_demo
 *   LD_LIBRARY_PATH=dm-vio_amd/lib ./c_abi_demo          (needs an MI355X; exits 0 when the pose is recovered) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "dmvio_hip.h"

#define W 256
#define H 256

static float texture(double x, double y) {
  return (float)(128.0 + 30.0 * sin(0.11 * x + 0.3) + 25.0 * sin(0.07 * y + 1.1) + 20.0 * sin(0.05 * (x + y)) + 15.0 * sin(0.13 * (x - 0.6 * y) + 0.7));
}

#define CHECK(call) do { if ((call) < 0) { fprintf(stderr, "%s failed: %s\n", #call, dmvio_hip_last_error()); return 1; } } while (0)

int main(void) {
  const float K4[4] = {200.0f, 200.0f, 127.5f, 127.5f};
  const double tx = 0.02, ty = -0.01, idepth = 0.5;
  float* ref = (float*)malloc(sizeof(float) * W * H);
  float* cur = (float*)malloc(sizeof(float) * W * H);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      ref[y * W + x] = texture(x, y);
      cur[y * W + x] = texture(x - K4[0] * tx * idepth, y - K4[1] * ty * idepth);   /* the wall as seen after the translation */
    }
  /* active points of the reference keyframe: a grid with the wall's inverse depth */
  enum { STEP = 6, MARGIN = 16, NMAX = ((W - 2 * MARGIN) / STEP + 1) * ((H - 2 * MARGIN) / STEP + 1) };
  static float u[NMAX], v[NMAX], id[NMAX], hdiF[NMAX];
  int n = 0;
  for (int y = MARGIN; y < H - MARGIN; y += STEP)
    for (int x = MARGIN; x < W - MARGIN; x += STEP) { u[n] = (float)x; v[n] = (float)y; id[n] = (float)idepth; hdiF[n] = 1e-4f; n++; }

  if (dmvio_hip_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", dmvio_hip_last_error()); return 2; }
  dmvio_hip_ctx* ctx = dmvio_hip_create(0, W, H, 2);
  if (!ctx) { fprintf(stderr, "dmvio_hip_create: %s\n", dmvio_hip_last_error()); return 1; }
  CHECK(dmvio_hip_frame_upload(ctx, 0, ref));                       /* fh->makeImages of the keyframe */
  CHECK(dmvio_hip_frame_upload(ctx, 1, cur));                       /* ... and of the new frame */
  dmvio_hip_tracker* trk = dmvio_hip_tracker_create(ctx);
  if (!trk) { fprintf(stderr, "dmvio_hip_tracker_create: %s\n", dmvio_hip_last_error()); return 1; }
  CHECK(dmvio_hip_tracker_make_k(trk, K4));                          /* coarseTracker->makeK(&Hcalib) */
  CHECK(dmvio_hip_tracker_set_ref(trk, 0, 1.0f, 0.0, 0.0, n, u, v, id, hdiF));   /* setCoarseTrackingRef(frameHessians) */
  double pose7[7] = {0, 0, 0, 0, 0, 0, 1}, aff[2] = {0, 0}, lastRes[5], flow[3], Hm[64], b[8];
  const double minRes[5] = {NAN, NAN, NAN, NAN, NAN};
  int good = 0;
  CHECK(dmvio_hip_tracker_track(trk, 1, 1.0f, pose7, aff, dmvio_hip_pyr_levels(ctx) - 1, minRes, lastRes, flow, Hm, b, &good));
  printf("levels %d, template points (level 0) %d, trackingGood %d\n", dmvio_hip_pyr_levels(ctx), dmvio_hip_tracker_pc_n(trk, 0), good);
  printf("refToNew translation (%.5f, %.5f, %.5f)  expected (%.5f, %.5f, 0)   quaternion (%.5f, %.5f, %.5f, %.5f)\n", pose7[0], pose7[1], pose7[2], tx, ty,
         pose7[3], pose7[4], pose7[5], pose7[6]);
  printf("rmse per level:"); for (int l = 0; l < dmvio_hip_pyr_levels(ctx); l++) printf(" %.3f", lastRes[l]); printf("\n");
  const double err = sqrt((pose7[0] - tx) * (pose7[0] - tx) + (pose7[1] - ty) * (pose7[1] - ty) + pose7[2] * pose7[2]);
  const double rot = sqrt(pose7[3] * pose7[3] + pose7[4] * pose7[4] + pose7[5] * pose7[5]);
  dmvio_hip_tracker_destroy(trk);
  dmvio_hip_destroy(ctx);
  free(ref); free(cur);
  if (!good || err > 2e-3 || rot > 2e-3) { fprintf(stderr, "pose not recovered (translation error %.2e m, rotation %.2e)\n", err, rot); return 3; }
  printf("ok: translation error %.2e m\n", err);
  return 0;
}
