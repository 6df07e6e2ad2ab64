/*
 * dmvio_hip.h — C ABI of libdmvio_hip.so: the MI355X (gfx950) implementation of DM-VIO's photometric
 * direct-alignment hot path (coarse tracking + sliding-window photometric bundle adjustment).
 *
 * The reference (lukasvst/dm-vio) has no FFI on this path; its boundary is the C++ member surface that
 * FullSystem uses.  Each entry point below names the reference interface it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the C++ adapter a maintainer adds on the reference side.
 *
 * Conventions
 *   - opaque handles, POD in / POD out, caller owns every host array, no pointers retained after return
 *   - poses are pose7 = [tx ty tz qx qy qz qw] (double), the reference's result.txt order
 *     (FullSystem.cpp:256-298); an SE3 "refToNew" maps reference-frame points into the new frame
 *   - return value: 0 = ok, <0 = error (dmvio_hip_last_error() describes it).  The reference's own
 *     convention on this path is bool / NaN / isLost; the adapter maps <0 to "tracking failed"/isLost
 *   - a dmvio_hip_tracker / dmvio_hip_ba is used by one thread at a time (tracking thread resp. mapping
 *     thread under mapMutex — FullSystem.h:273-332); the ctx is shared and internally locked
 *   - calls are synchronous on return unless named *_async
 */
#ifndef DMVIO_HIP_H
#define DMVIO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMVIO_HIP_MAX_LEVELS 6 /* PYR_LEVELS, src/dso/util/settings.h:52 */

typedef struct dmvio_hip_ctx dmvio_hip_ctx;
typedef struct dmvio_hip_tracker dmvio_hip_tracker;
typedef struct dmvio_hip_ba dmvio_hip_ba;
typedef struct dmvio_hip_graph dmvio_hip_graph;   /* host-side mirror of the window's point / residual graph ("window graph" below) */

/* ------------------------------------------------------------------ context ------------------ */
const char* dmvio_hip_last_error(void);
int dmvio_hip_device_count(void);

/* Replaces the allocations of FullSystem::FullSystem (FullSystem.cpp:73-187) + setGlobalCalib's level rule
 * (src/dso/util/globalCalib.cpp:45-105).  n_frame_slots image pyramids are kept resident in HBM. */
dmvio_hip_ctx* dmvio_hip_create(int device, int w, int h, int n_frame_slots);
void dmvio_hip_destroy(dmvio_hip_ctx* ctx);
int dmvio_hip_pyr_levels(const dmvio_hip_ctx* ctx);
/* Run all work of this ctx on a caller-provided hipStream_t (e.g. torch's current stream); NULL = own stream. */
int dmvio_hip_set_stream(dmvio_hip_ctx* ctx, void* hip_stream);
/* A stream of their own for the BATCHED pyramid builds (dmvio_hip_frames_from_device_batch / _attach_device_batch / _from_raw_device_batch); NULL (default) = the context's
 * stream.  Lets the build of batch k+1 overlap the tracking of batch k (different bounds: HBM vs L1 miss path / VALU).  The caller orders the two streams with events: a
 * build must not start before the consumers of the slots it rewrites are done, a consumer not before the build of its slots.  dmvio_hip_synchronize waits for both. */
int dmvio_hip_set_build_stream(dmvio_hip_ctx* ctx, void* hip_stream);
/* Tile shape of the LDS-tile pyramid builds: 2^tw_log2 pixels wide x 4096 / 2^tw_log2 high (7 .. 9, at least 2^(levels-1) rows).  A measurement knob; default: as wide as the image allows. */
int dmvio_hip_set_pyramid_tile_log2(dmvio_hip_ctx* ctx, int tw_log2);
int dmvio_hip_synchronize(dmvio_hip_ctx* ctx);

/* ------------------------------------------------------------------ frames ------------------- */
/* FrameHessian::makeImages (src/dso/FullSystem/HessianBlocks.cpp:128-191): irradiance w*h floats (0..255)
 * -> dIp[lvl] = (I, dx, dy) pyramids resident in slot.  *_upload copies from host first; *_from_device
 * consumes a device pointer already in HBM (no PCIe traffic). */
int dmvio_hip_frame_upload(dmvio_hip_ctx* ctx, int slot, const float* irradiance_host);
int dmvio_hip_frame_from_device(dmvio_hip_ctx* ctx, int slot, const float* irradiance_dev);
/* Raw camera image on the upload path (data-format edge): PhotometricUndistorter::processFrame
 * (src/dso/util/Undistort.cpp:214-250: G[raw] * vignetteMapInv, or factor * raw when G == NULL) and Undistort::undistort
 * (Undistort.cpp:386-481: bilinear remap through remapX / remapY, xx < 0 -> 0; NULL maps = passthrough) run on the device, then
 * makeImages.  bits = 8 / 16 (unsigned char / unsigned short raw pixels, G with 256 / 65536 entries).  1 B/px crosses PCIe instead of
 * 4 B/px.  undistorted_out (may be NULL) receives the w*h irradiance image the reference would hand to makeImages. */
typedef struct dmvio_hip_undistorter dmvio_hip_undistorter;
dmvio_hip_undistorter* dmvio_hip_undistorter_create(dmvio_hip_ctx* ctx, int wOrg, int hOrg, int bits, const float* G, const float* vignetteMapInv,
                                                    const float* remapX, const float* remapY);
void dmvio_hip_undistorter_destroy(dmvio_hip_undistorter* und);
int dmvio_hip_frame_upload_raw(dmvio_hip_ctx* ctx, dmvio_hip_undistorter* und, int slot, const void* raw, float factor, float* undistorted_out);
/* FullSystem::printResult (src/dso/FullSystem/FullSystem.cpp:256-298): "timestamp tx ty tz qx qy qz qw" per frame with a valid pose,
 * 15 significant digits, camToFirst = firstPose^-1 * camToWorld.  pose7 = [tx ty tz qx qy qz qw].  pose_valid (n, may be NULL = all
 * valid): FrameShell::poseValid.  tracking_ref (n, may be NULL): index of the frame's tracking reference, or -1 for keyframes; when
 * >= 0 the pose written is camToWorld[tracking_ref] * camToTrackingRef (useCamToTrackingRef).  Host-only, no device needed. */
int dmvio_hip_write_result_txt(const char* path, int n, const double* timestamps, const double* camToWorld7, const unsigned char* pose_valid,
                               const int* tracking_ref, const double* camToTrackingRef7, const double firstPose7[7]);
/* makeImages for B frames in 4 launches: frame i is read from dev_base + i*stride_bytes and written to slots[i].
 * Asynchronous on the ctx stream (ordering with later tracker calls is by stream order). */
int dmvio_hip_frames_from_device_batch(dmvio_hip_ctx* ctx, int B, const int* slots, const float* dev_base, size_t stride_bytes);
/* Zero-copy form of the call above for images that stay resident: this library keeps the intensity plane only (the gradient channels of
 * dIp are rebuilt at the taps), so level 0 of FrameHessian::makeImages IS the input image — the slots reference the caller's images in
 * place and only the coarser levels are built (1.33 B written per input byte less).  The images (row-major, w floats per row) must stay
 * valid and unchanged until their slots are rebuilt or the context is destroyed. */
int dmvio_hip_frames_attach_device_batch(dmvio_hip_ctx* ctx, int B, const int* slots, const float* dev_base, size_t stride_bytes);
/* Raw-image form of the batched build: B raw camera images (8- or 16-bit as the undistorter was created, wOrg x hOrg each, frame i at
 * raw_dev_base + i * stride_bytes) ALREADY IN DEVICE MEMORY -> PhotometricUndistorter::processFrame + Undistort::undistort (src/dso/util/Undistort.cpp:214-250,
 * 386-481) + FrameHessian::makeImages (src/dso/FullSystem/HessianBlocks.cpp:128-191) of slots[i], fused into one launch: a frame crosses PCIe and
 * enters the kernel as 1 (2) bytes per pixel instead of 4, and the undistorted fp32 image is written once, as level 0.  `factor` as in
 * dmvio_hip_frame_upload_raw.  Bit-identical to B calls of dmvio_hip_frame_upload_raw.  Asynchronous on the ctx stream. */
int dmvio_hip_frames_from_raw_device_batch(dmvio_hip_ctx* ctx, dmvio_hip_undistorter* und, int B, const int* slots, const void* raw_dev_base, size_t stride_bytes, float factor);
/* Layout of the level-0 plane dmvio_hip_frames_from_raw_device_batch writes: tiled == 0 (default): row-major; tiled != 0: 8x4-pixel tiles (one 128-byte line each), which the
 * coarse tracker's batch kernel (dmvio_hip_tracker_track_batch with >= 2 problems) gathers from directly — the 4x4 footprint of a bilinear tap (getInterpolatedElement33,
 * util/globalFuncs.h:103-118) then touches 2.4 lines on average instead of 4.3, at the price of twelve single loads with their own tile addresses per tap.  Measured on MI355X
 * the price is the larger term (k_track_lm 8 % slower at 4096 frames), which is why the layout is an option and not the default.  Needs w % 8 == 0 and h % 4 == 0 (other sizes
 * are always row-major).  Every other consumer of a tiled slot (setCoarseTrackingRef, single-frame tracking, the window optimiser, immature points, the initializer, downloads)
 * converts it back to row-major on first use — same values, bit for bit.  dmvio_hip_frame_level0_is_tiled reports a slot's state. */
int dmvio_hip_set_raw_batch_layout(dmvio_hip_ctx* ctx, int tiled);
int dmvio_hip_frame_level0_is_tiled(dmvio_hip_ctx* ctx, int slot);
/* Kernels behind dmvio_hip_frames_from_raw_device_batch and dmvio_hip_frames_attach_device_batch: variant 1 (default) = a 4 x 8 pixel block per thread, every level formed
 * in registers, no workgroup barrier (pyramids of at most four levels on images whose sides are multiples of 8; other geometries always take variant 0); variant 0 = the
 * general LDS-tile builds (which also serve single uploads and dmvio_hip_frames_from_device_batch).  Identical output, bit for bit; for A/B measurements. */
int dmvio_hip_set_raw_batch_kernel(dmvio_hip_ctx* ctx, int variant);
/* Diagnostics.  Every pyramid build stamps its slot "clean" when all pixels are finite (|I| <= 1e30): consumers then run without the
 * reference's isfinite guards (HessianBlocks.cpp:172-181, CoarseTracker.cpp:455), which cannot fire on such a frame.  This call withdraws
 * the stamp so that the guarded code path runs (tests compare the two). */
int dmvio_hip_frame_mark_unclean(dmvio_hip_ctx* ctx, int slot);
/* 1: the slot carries the clean stamp of its last build, 0: it does not (a pixel was not finite, or the stamp was withdrawn), < 0: error.  Waits for the context's stream. */
int dmvio_hip_frame_is_clean(dmvio_hip_ctx* ctx, int slot);
/* dIp[lvl] back on host as w_l*h_l*3 floats (AoS, the reference's Eigen::Vector3f layout). */
int dmvio_hip_frame_download(dmvio_hip_ctx* ctx, int slot, int lvl, float* dIp_host);
/* absSquaredGrad[0..n_levels-1] of FrameHessian::makeImages (src/dso/FullSystem/HessianBlocks.cpp:169-189) — the planes PixelSelector::makeMaps /
 * makeHists read (src/dso/FullSystem/PixelSelector2.cpp) — computed from the resident intensity planes of `slot` and copied to out_host[l] (w_l*h_l floats each,
 * NULL entries are skipped).  B_lut256 = CalibHessian::B (256 floats) applies the getBGradOnly weights of setting_gammaWeightsPixelSelect == 1
 * (HessianBlocks.h:394-400); NULL = no response weighting.  Rows 0 and h_l-1 (never written by the reference) are zero.  Bit-identical to the CPU path. */
int dmvio_hip_frame_abs_squared_grad(dmvio_hip_ctx* ctx, int slot, int n_levels, const float* B_lut256, float* const* out_host);

/* ------------------------------------------------------------------ coarse tracker ----------- */
typedef struct dmvio_hip_tracker_settings {
  float huberTH;          /* setting_huberTH          = 9    settings.cpp:148 */
  float coarseCutoffTH;   /* setting_coarseCutoffTH   = 20   settings.cpp:160 */
  float affineOptModeA;   /* setting_affineOptModeA   = 1e12 settings.cpp:139 */
  float affineOptModeB;   /* setting_affineOptModeB   = 1e8  settings.cpp:140 */
} dmvio_hip_tracker_settings;

/* CoarseTracker::CoarseTracker (CoarseTracker.cpp:62-97) */
dmvio_hip_tracker* dmvio_hip_tracker_create(dmvio_hip_ctx* ctx);
void dmvio_hip_tracker_destroy(dmvio_hip_tracker* trk);
int dmvio_hip_tracker_set_settings(dmvio_hip_tracker* trk, const dmvio_hip_tracker_settings* s);
/* CoarseTracker::makeK (CoarseTracker.cpp:105-134): fx, fy, cx, cy of level 0 */
int dmvio_hip_tracker_make_k(dmvio_hip_tracker* trk, const float fxfycxcy[4]);
/* CoarseTracker::setCoarseTrackingRef + makeCoarseDepthL0 (CoarseTracker.cpp:524-538, 138-295).
 * ref_slot: frame slot of lastRef.  The n points are the active points whose newest residual
 * (target == lastRef) is IN: (u,v,idepth) = centerProjectedTo, hdiF = efPoint->HdiF (:144-161). */
int dmvio_hip_tracker_set_ref(dmvio_hip_tracker* trk, int ref_slot, float ref_exposure, double ref_aff_a, double ref_aff_b,
                              int n, const float* u, const float* v, const float* idepth, const float* hdiF);
/* pc_n[lvl] / pc_u,pc_v,pc_idepth,pc_color[lvl] (CoarseTracker.h:113-118) — parity / debugPlotIDepthMap */
int dmvio_hip_tracker_pc_n(dmvio_hip_tracker* trk, int lvl);
int dmvio_hip_tracker_get_pc(dmvio_hip_tracker* trk, int lvl, float* u, float* v, float* idepth, float* color);
/* CoarseTracker::idepth[lvl] / weightSums[lvl] (w_lvl * h_lvl floats each) as makeCoarseDepthL0 leaves them (CoarseTracker.cpp:249-293): what debugPlotIDepthMap /
 * debugPlotIDepthMapFloat (:772-880) read when output wrappers exist.  Debug path (two plane downloads); weightSums_out may be NULL. */
int dmvio_hip_tracker_get_idepth_map(dmvio_hip_tracker* trk, int lvl, float* idepth_out, float* weightSums_out);
/* One fused CoarseTracker::calcRes + calcGSSSE evaluation (CoarseTracker.cpp:361-517, 299-356) at refToNew.
 * res6 = {E, numTermsInE, flowT, 0, flowRT, saturatedRatio}; H (8x8 row-major) and b are the SCALE_*-scaled
 * system handed to IMUIntegration::computeCoarseUpdate in VIO mode (CoarseTracker.cpp:612-637). */
int dmvio_hip_tracker_eval(dmvio_hip_tracker* trk, int lvl, int new_slot, float new_exposure,
                           const double pose7_ref_to_new[7], const double aff_g2l_new[2], float cutoffTH,
                           double res6[6], double H[64], double b[8]);
/* CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:539-770), visual-only (useimu=0) branch, whole LM loop
 * resident on the device.  pose7/aff are in/out (written only when every level finished, like the reference);
 * returns trackingGood in *good; lastResiduals(5)/lastFlow(3) = CoarseTracker::lastResiduals/lastFlowIndicators;
 * H,b = system at the accepted state of the last level. */
int dmvio_hip_tracker_track(dmvio_hip_tracker* trk, int new_slot, float new_exposure,
                            double pose7_io[7], double aff_io[2], int coarsestLvl, const double minResForAbort[5],
                            double lastResiduals[5], double lastFlow[3], double H[64], double b[8], int* good);
/* How ONE alignment problem (dmvio_hip_tracker_track, or a batch of one) is run: 1 (default) = the LM loop on the host against the evaluation server — one kernel launch per
 * frame, every calcRes + calcGSSSE evaluation a request through host-coherent memory, the 8x8 solve / SE3 exp on the CPU (sub-microsecond there, 6.5 us per iteration on one
 * wavefront); 0 = the device-resident LM (cluster mode).  Same evaluation sums and iteration counts either way; batches of two and more always run device-resident. */
int dmvio_hip_tracker_set_single_frame_mode(dmvio_hip_tracker* trk, int host_lm);
/* Measurement knobs as explicit calls (the library reads no environment variable that changes what or how it computes).  0 = the library's own choice.
 * eval_blocks: workgroups per fused evaluation / evaluation server; lm_threads (256 | 512) and lm_waves (1 | 2 | 4): workgroup shape of the device-resident LM; lm_cluster
 * (2 .. 32): workgroups sharing one alignment problem.  eval_blocks and lm_cluster change how the fp32 partial sums are grouped (results move in the last bits). */
int dmvio_hip_tracker_set_launch_shape(dmvio_hip_tracker* trk, int eval_blocks, int lm_threads, int lm_waves, int lm_cluster);
/* 1 (default): a host-driven LM (dmvio_hip_tracker_track of one frame, dmvio_hip_tracker_track_vio) posts its evaluations to the resident evaluation server; 0: one launch each */
int dmvio_hip_tracker_set_eval_server(dmvio_hip_tracker* trk, int on);
/* Kernel of full batches (>= 512 alignment problems per launch): 0 = four wavefronts per problem (they evaluate, then three wait while the first runs the LM control step),
 * 1 = a four-wavefront workgroup holds two problems: wavefront 0 runs the control step of one problem and then joins the other three in the evaluation of the other;
 * problems dealt out to a persistent grid by a device-wide counter.
 * Per problem the same arithmetic in the same order: identical results. */
int dmvio_hip_tracker_set_batch_kernel(dmvio_hip_tracker* trk, int mode);
/* Storage order of the template points (from the next dmvio_hip_tracker_set_ref on): 0 (default) = 8x8-pixel tiles, Z-ordered inside 16x16 blocks; 1 = the reference's
 * row-major order (CoarseTracker.cpp:249-293).  Same points, same per-point arithmetic; the fp32 partial sums are grouped differently (results agree to rounding). */
int dmvio_hip_tracker_set_template_order(dmvio_hip_tracker* trk, int row_major);
/* Diagnostics (profiles/r05_tracker_floor.md): mode 1 = the following dmvio_hip_tracker_track_batch_launch calls record the parameters of every evaluation they run (full
 * batches: one 256-thread workgroup per problem); mode 2 = they run the recorded evaluations again without the LM control steps between them (same points, same taps, same
 * fused reductions; no results to fetch — time the launch on the context's stream); 0 = normal operation. */
int dmvio_hip_tracker_debug_record_replay(dmvio_hip_tracker* trk, int mode);
/* Idle limit of the evaluation server (the resident kernel behind a single-frame dmvio_hip_tracker_track / _track_vio call): it leaves after this long without a request
 * and is started again by the next one.  Default 5000 us; raise it when the computeCoarseUpdate hook (IMUIntegration's factor-graph solve) regularly takes longer, so
 * that an LM iteration does not pay a relaunch.  100 us .. 2 s. */
int dmvio_hip_tracker_set_server_idle_us(dmvio_hip_tracker* trk, int microseconds);
/* The reference's DEFAULT tracking path (settings.cpp:36 setting_useIMU = true): trackNewestCoarse with every LM step handed to the
 * host (CoarseTracker.cpp:612-637).  The callbacks mirror the three members of dmvio::IMUIntegration the tracker calls
 * (src/IMU/IMUIntegration.hpp:106-112):
 *   update  <- computeCoarseUpdate(H, b, extrapFac, lambda, incA, incB, incNorm) -> refToNew_new.  H (8x8 row-major) and b are the
 *              SCALE_*-scaled system in the order [trans3 rot3 a b]; pose7_cur / aff_cur = the current estimate (for callers that do not
 *              keep it themselves); pose7_new = the new refToNew; *incA, *incB = affine increments BEFORE the SCALE_A / SCALE_B the
 *              tracker applies (CoarseTracker.cpp:633-637); *incNorm ends the level when it is not > 1e-3.  Return 0; nonzero aborts the call.
 *   accept  <- acceptCoarseUpdate()            (may be NULL)
 *   visual  <- addVisualToCoarseGraph(H, b, trackingGood), called when the finest level was reached (may be NULL)
 * update == NULL (or cb == NULL) runs dmvio_hip_coarse_update_visual: the reference's own visual-only step, which is also what it
 * executes while the IMU is not yet initialised.  One kernel launch per CALL (the evaluation server, k_eval_server): every evaluation is a request posted into host-coherent
 * memory and its sums are polled from there — no launch, no stream synchronisation per LM iteration; with the default update the results are those of dmvio_hip_tracker_track.
 * The callbacks run on the calling thread with the context locked and the server resident on the context's stream: they must not call entry points of the SAME context
 * (dmvio_hip_coarse_update_visual and other contexts are fine).  A callback that takes longer than the server's 5 ms idle limit only costs a relaunch of the kernel. */
typedef int (*dmvio_hip_coarse_update_fn)(void* user, const double H[64], const double b[8], float extrapFac, float lambda, const double pose7_cur[7],
                                          const double aff_cur[2], double pose7_new[7], double* incA, double* incB, double* incNorm);
typedef void (*dmvio_hip_coarse_accept_fn)(void* user);
typedef void (*dmvio_hip_coarse_visual_fn)(void* user, const double H[64], const double b[8], int trackingGood);
typedef struct dmvio_hip_coarse_callbacks {
  void* user;
  dmvio_hip_coarse_update_fn update;
  dmvio_hip_coarse_accept_fn accept;
  dmvio_hip_coarse_visual_fn visual;
} dmvio_hip_coarse_callbacks;
int dmvio_hip_tracker_track_vio(dmvio_hip_tracker* trk, int new_slot, float new_exposure, double pose7_io[7], double aff_io[2], int coarsestLvl,
                                const double minResForAbort[5], const dmvio_hip_coarse_callbacks* cb, double lastResiduals[5], double lastFlow[3],
                                double H[64], double b[8], int* good, int* n_evals);
/* The visual-only LM step of CoarseTracker.cpp:639-682 (damped H, 6 / 7 / 8-dof LDL^T by affineOptModeA / B, extrapolation, SE3::exp),
 * host-only; st == NULL = the reference's default settings.  For callbacks that fall back to the visual step. */
int dmvio_hip_coarse_update_visual(const dmvio_hip_tracker_settings* st, const double H[64], const double b[8], float extrapFac, float lambda,
                                   const double pose7_cur[7], double pose7_new[7], double* incA, double* incB, double* incNorm);
/* Same for B independent alignment problems against the current reference in one launch: the pose-hypothesis
 * list of FullSystem::trackNewCoarse (FullSystem.cpp:364-402) and/or a batch of new frames.  All arrays are
 * B-major (pose7_io[B*7], aff_io[B*2], minResForAbort[B*5], lastResiduals[B*5], lastFlow[B*3], H[B*64], b[B*8],
 * good[B], iterations[B] (may be NULL)). */
int dmvio_hip_tracker_track_batch(dmvio_hip_tracker* trk, int B, const int* new_slots, const float* new_exposures,
                                  double* pose7_io, double* aff_io, int coarsestLvl, const double* minResForAbort,
                                  double* lastResiduals, double* lastFlow, double* H, double* b, int* good, int* iterations);
/* Enqueue-only variant: inputs must have been staged with a previous _track_batch call of the same B or
 * _track_batch_stage; results stay on the device until _track_batch_fetch.  Used for kernel timing. */
int dmvio_hip_tracker_track_batch_stage(dmvio_hip_tracker* trk, int B, const int* new_slots, const float* new_exposures,
                                        const double* pose7_in, const double* aff_in, int coarsestLvl, const double* minResForAbort);
int dmvio_hip_tracker_track_batch_launch(dmvio_hip_tracker* trk);
/* Marks the last launch's results as "fetched later"; a following _track_batch_fetch returns THOSE results.  Between the two calls the
 * next batch (not a larger one) may be staged and launched: the kernel writes its results straight into one of two pinned host
 * buffers, so the device stays busy while the host unpacks the previous batch. */
int dmvio_hip_tracker_track_batch_fetch_begin(dmvio_hip_tracker* trk);
int dmvio_hip_tracker_track_batch_fetch(dmvio_hip_tracker* trk, double* pose7_out, double* aff_out, double* lastResiduals,
                                        double* lastFlow, double* H, double* b, int* good, int* iterations);
/* FullSystem::trackNewCoarse (FullSystem.cpp:300-539), visual-only path without IMU hint:
 *  - dmvio_hip_make_track_hypotheses builds lastF_2_fh_tries (:364-402: constant / double / half / zero motion, zero motion from the
 *    keyframe, 26 small rotations) from the camToWorld poses of the last two frames and of the reference keyframe; returns the count (31);
 *  - dmvio_hip_tracker_track_new_coarse runs the try loop (:419-489) and returns what the reference writes into fh->shell:
 *    the winning refToNew pose, aff_g2l, flow indicators; lastCoarseRMSE is in/out (achievedRes); *winner = index of the winning try or -1. */
int dmvio_hip_make_track_hypotheses(const double slast_c2w[7], const double sprelast_c2w[7], const double lastF_c2w[7], double* out7, int max_out);
int dmvio_hip_tracker_track_new_coarse(dmvio_hip_tracker* trk, int new_slot, float new_exposure, int n_tries, const double* tries7, const double aff_last[2],
                                       double lastCoarseRMSE_io[5], double reTrackThreshold, double pose7_out[7], double aff_out[2], double flow_out[3],
                                       int* winner, int* tries_used, int* tracking_good);
/* Work counters of the last batch launch: evals (calcRes+calcGS passes) and point-evaluations
 * (sum over evals of pc_n[lvl]) — the unit count behind the roofline's algorithmic bytes. */
int dmvio_hip_tracker_last_work(dmvio_hip_tracker* trk, long long* n_evals, long long* n_point_evals);
/* In-kernel time split of the last batch launch, summed over problems, in 100 MHz wall_clock64 ticks:
 * LM control steps (solve, pose update, bookkeeping) vs evaluations (calcRes+calcGS). Diagnostics only. */
int dmvio_hip_tracker_last_ticks(dmvio_hip_tracker* trk, long long* ticks_step, long long* ticks_eval);
/* Shape of the last batch launch: workgroups sharing one alignment problem (cluster mode for small batches) and threads per
 * workgroup.  Diagnostics only. */
int dmvio_hip_tracker_last_launch(dmvio_hip_tracker* trk, int* workgroups_per_problem, int* threads_per_workgroup);
/* Diagnostics: for n operand pairs, the quotient a/b as the tracker's evaluation loop computes it (refined reciprocal shared by the
 * quotients of one denominator, CoarseTracker.cpp:381-384 / :465 being the divisions in question) and as the compiler's IEEE division does. */
int dmvio_hip_selftest_divide(dmvio_hip_ctx* ctx, int n, const float* a, const float* b, float* q_shared, float* q_ieee);

/* ------------------------------------------------------------------ sliding-window BA -------- */
/* The window FullSystem::optimize works on (FullSystemOptimize.cpp:417-647): F <= 8 keyframes, N active points, R residuals.
 * Frame order = frameHessians order (newest last); frame f's image pyramid must be resident in slots[f].
 *   pose7_w2c[F*7]  worldToCam_evalPT of every frame (HessianBlocks.h:160), aff_ab[F*2] = aff_g2l (a, b), exposures[F] = ab_exposure,
 *   frameIDs[F]     FrameHessian::frameID (0 = very first keyframe -> strong pose prior, HessianBlocks.h:264-299),
 *   fxfycxcy        CalibHessian::value_scaled (HessianBlocks.h:318).
 * Corresponds to EnergyFunctional::insertFrame + setAdjointsF + FullSystem::setPrecalcValues. */
dmvio_hip_ba* dmvio_hip_ba_create(dmvio_hip_ctx* ctx);
void dmvio_hip_ba_destroy(dmvio_hip_ba* ba);
/* Every dmvio_hip_ba handle enqueues on a HIP stream of its own and has its own lock, so the mapping thread (optimize, marginalisation)
 * overlaps on the device with the tracking thread, which uses the context's stream — the reference's two-thread structure
 * (FullSystem.cpp:980-985: coarseTracker on the tracking thread, mapping under mapMutex).  Frames must be resident (upload calls return
 * after their stream synchronised) before a window refers to them.  dmvio_hip_ba_set_stream replaces the stream (NULL: own stream). */
int dmvio_hip_ba_set_stream(dmvio_hip_ba* ba, void* hip_stream);
/* The window: F keyframes, oldest first (FullSystem::frameHessians), 1 <= F <= dmvio_hip_ba_max_frames().  The window size is a RUN-TIME setting of the reference
 * (setting_maxFrames, util/settings.cpp:100, registered as `maxFrames` in util/MainSettings.cpp:223,246; default 7, i.e. 8 keyframes while the newest is optimised):
 * every kernel takes F at run time; dmvio_hip_ba_max_frames() is the library's upper bound (12: the per-pair kernel arguments exist in an 8- and a 12-keyframe form,
 * the stitch workgroup is 64 F threads).  A larger F is refused with an error (never truncated). */
int dmvio_hip_ba_max_frames(void);
int dmvio_hip_ba_set_window(dmvio_hip_ba* ba, int F, const int* slots, const double* pose7_w2c, const double* aff_ab, const float* exposures,
                            const int* frameIDs, const double fxfycxcy[4]);
/* Marginalisation prior HM (n x n row-major), bM (n), n = 4 + 8F (EnergyFunctional.h:129-131); zero when never called. */
int dmvio_hip_ba_set_marg_prior(dmvio_hip_ba* ba, const double* HM, const double* bM);
/* FrameHessian::setState (HessianBlocks.h:179-199): state10 = [xi (6, left increment on the evaluation point) | a, b | 0, 0] in the
 * reference's unscaled units, followed by FullSystem::setPrecalcValues (FullSystem.cpp:1670-1680). */
int dmvio_hip_ba_set_frame_state(dmvio_hip_ba* ba, int frame, const double state10[10]);
/* For a window taken over from a running system: a keyframe's linearisation state FrameHessian::state_zero (HessianBlocks.cpp:74-107; the
 * pose part is zero by construction, the affine part is the brightness at the evaluation point), the outlier thresholds
 * FrameHessian::frameEnergyTH of all keyframes (th[F]) and CalibHessian::value / value_zero (the reference's unscaled units: fx,fy / SCALE_F,
 * cx,cy / SCALE_C; `value` is the primary quantity of a running system, value_scaled its product with the float SCALE_* constants). */
int dmvio_hip_ba_set_frame_zero(dmvio_hip_ba* ba, int frame, const double state_zero10[10]);
/* The two calls above for ALL keyframes of the window at once (state_zero10 / state10: F x 10 doubles each, either may be NULL): one setPrecalcValues instead of one per call —
 * what an adapter that takes a window over from a running FullSystem uses (tests/dropin). */
int dmvio_hip_ba_set_frame_states(dmvio_hip_ba* ba, const double* state_zero10, const double* state10);
int dmvio_hip_ba_set_frame_energy_th(dmvio_hip_ba* ba, const float* th);
/* IMUIntegration::newFrameEnergyTH (src/IMU/IMUIntegration.cpp:365-373; FullSystemOptimize.cpp:136-140, setting_useIMU only): IMUSettings::maxFrameEnergyThreshold caps the
 * threshold setNewFrameEnergyTH selects for the newest keyframe.  <= 0 (the default, and the reference's): no cap. */
int dmvio_hip_ba_set_frame_energy_th_cap(dmvio_hip_ba* ba, float maxFrameEnergyThreshold);
int dmvio_hip_ba_set_calib_values(dmvio_hip_ba* ba, const double value[4], const double value_zero[4]);
/* Point marginalisation: the relinearisation branch of FullSystem::flagPointsForRemoval (FullSystem.cpp:829-859: resetOOB, linearize,
 * applyRes, EFResidual::fixLinearizationF EnergyFunctionalStructs.cpp:76-106) for the points with candidates[i] != 0, the
 * marginalise-or-drop decision (idepth_hessian > setting_minIdepthH_marg), then EnergyFunctional::marginalizePointsF
 * (EnergyFunctional.cpp:678-742): AccumulatedTopHessianSSE::addPoint<2> + AccumulatedSCHessianSSE::addPoint(p, false) over the
 * marginalised points, stitched.  decision[i]: 0 untouched, 1 marginalised, 2 dropped.  Hadd / badd (may be NULL) = the increment
 * setting_margWeightFac * (M - Msc) of HM / bM; with update_prior != 0 it is also added to the handle's marginalisation prior.
 * The caller then rebuilds the graph without the decided points (dmvio_hip_ba_set_graph), as the reference's removePoint does. */
int dmvio_hip_ba_marginalize_points(dmvio_hip_ba* ba, const unsigned char* candidates, unsigned char* decision, double* Hadd, double* badd, int* resInM,
                                    int update_prior);
/* EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:520-660, visual-only branch :570-640): Schur complement of keyframe `frame`
 * (its block moved last, its prior added, diagonal pre-scaling) on the handle's marginalisation prior; returns the prior of the
 * remaining window, (n-8) x (n-8) row-major and n-8.  The caller re-creates the window without the frame (dmvio_hip_ba_set_window)
 * and installs the result with dmvio_hip_ba_set_marg_prior.  dmvio_hip_ba_get_marg_prior reads the current HM, bM. */
int dmvio_hip_ba_marginalize_frame(dmvio_hip_ba* ba, int frame, double* HM_new, double* bM_new);
int dmvio_hip_ba_get_marg_prior(dmvio_hip_ba* ba, double* HM, double* bM);
/* Flattened point / residual graph (EnergyFunctional::makeIDX, EnergyFunctional.cpp:997-1017): point p is hosted in frame host[p],
 * has PointHessian::u, v, idepth, color[8], weights[8] (HessianBlocks.h:419-436) and hasDepthPrior; residual r observes point
 * res_point[r] in frame res_target[r].  Residuals must be sorted by point (points in allPoints order). */
int dmvio_hip_ba_set_graph(dmvio_hip_ba* ba, int N, const int* host, const float* u, const float* v, const float* idepth, const float* color8,
                           const float* weights8, const unsigned char* hasDepthPrior, int R, const int* res_point, const int* res_target);
/* ---- window graph: the point / residual graph kept RESIDENT across keyframes -----------------------------------------------------------------------------------------
 * The reference never rebuilds its graph; EnergyFunctional mutates it in place (EnergyFunctional.cpp: insertResidual :435-446, insertFrame :447-484, insertPoint :485-499,
 * dropResidual :500-518, removePoint :766-782, marginalizeFrame :641-646, makeIDX :997-1017).  A dmvio_hip_graph is a host-side mirror with exactly those mutators,
 * addressed the way the reference addresses its own objects — a keyframe by EFFrame::idx, a point by (host keyframe, EFPoint::idxInPoints), a residual by (point,
 * EFResidual::idxInAll) — and with the reference's order semantics (append; the LAST element takes a removed one's place; frames keep their order).  An adapter forwards
 * each EnergyFunctional call with the indices it already holds (one line per member) and calls dmvio_hip_ba_set_graph_from once per keyframe
 * instead of flattening its pointer graph for dmvio_hip_ba_set_graph: the library walks compact records in makeIDX order, which is the order every accumulator adds in.
 * Host only: none of the dmvio_hip_graph_* calls needs a device.  One graph may be used from several threads (internally locked).
 *   insert_* return the new element's index (>= 0); every call returns < 0 on error (dmvio_hip_last_error). */
dmvio_hip_graph* dmvio_hip_graph_create(void);
void dmvio_hip_graph_destroy(dmvio_hip_graph* g);
int dmvio_hip_graph_clear(dmvio_hip_graph* g);
int dmvio_hip_graph_insert_frame(dmvio_hip_graph* g);                                           /* EnergyFunctional::insertFrame: appended; returns EFFrame::idx */
/* EnergyFunctional::marginalizeFrame's effect on the graph: keyframe idx leaves (it must host no point any more), later keyframes move down by one.  Residuals that still
 * target it — FullSystem::marginalizeFrame drops them right AFTER ef->marginalizeFrame (FullSystemMarginalize.cpp:162-196) — stay as dangling until their
 * dmvio_hip_graph_drop_residual arrives; dmvio_hip_ba_set_graph_from refuses a graph that still has one. */
int dmvio_hip_graph_remove_frame(dmvio_hip_graph* g, int idx);
/* EnergyFunctional::insertPoint (+ EFPoint::takeData): PointHessian::u, v, idepth, color[8], weights[8], hasDepthPrior; appended to its host's points; returns idxInPoints */
int dmvio_hip_graph_insert_point(dmvio_hip_graph* g, int host, float u, float v, float idepth, const float* color8, const float* weights8, int hasDepthPrior);
int dmvio_hip_graph_remove_point(dmvio_hip_graph* g, int host, int idxInPoints);                /* EnergyFunctional::removePoint: its residuals go with it, the host's last point takes its index */
int dmvio_hip_graph_insert_residual(dmvio_hip_graph* g, int host, int idxInPoints, int target);  /* EnergyFunctional::insertResidual: appended; returns idxInAll */
int dmvio_hip_graph_drop_residual(dmvio_hip_graph* g, int host, int idxInPoints, int idxInAll);  /* EnergyFunctional::dropResidual: the point's last residual takes its index */
/* EFResidual::fixLinearizationF's result (EnergyFunctionalStructs.cpp:85-113) of one residual: isLinearized = true with its frozen Jacobian (74 floats, layout above) and
 * res_toZeroF (8); J74 == NULL: isLinearized = false again (FullSystem.cpp:840-843).  The record moves with the residual (dropResidual) and goes with its point (removePoint). */
int dmvio_hip_graph_set_residual_linearized(dmvio_hip_graph* g, int host, int idxInPoints, int idxInAll, const float* J74, const float* res_toZeroF);
int dmvio_hip_graph_linearized_count(dmvio_hip_graph* g);
int dmvio_hip_graph_export_linearized(dmvio_hip_graph* g, unsigned char* isLinearized, float* J74, float* res_toZeroF);   /* flat order of dmvio_hip_graph_export; any may be NULL */
int dmvio_hip_graph_set_idepth(dmvio_hip_graph* g, int host, int idxInPoints, float idepth);     /* PointHessian::setIdepth of one point */
int dmvio_hip_graph_set_idepths(dmvio_hip_graph* g, int N, const float* idepth);                /* ... of all points in makeIDX order: what dmvio_hip_ba_get_points returns after an optimisation; refused once the graph's structure changed since it was flattened */
int dmvio_hip_graph_counts(dmvio_hip_graph* g, int* F, int* N, int* R);                         /* EnergyFunctional::nFrames, nPoints, nResiduals */
int dmvio_hip_graph_frame_points(dmvio_hip_graph* g, int host);                                 /* EFFrame::points.size() */
int dmvio_hip_graph_point_residuals(dmvio_hip_graph* g, int host, int idxInPoints);             /* EFPoint::residualsAll.size() */
/* The graph as the flat arrays dmvio_hip_ba_set_graph takes, in makeIDX order (any output may be NULL; sizes from dmvio_hip_graph_counts; a dangling residual has target -1). */
int dmvio_hip_graph_export(dmvio_hip_graph* g, int* host, float* u, float* v, float* idepth, float* color8, float* weights8, unsigned char* hasDepthPrior, int* res_point,
                           int* res_target);
/* dmvio_hip_ba_set_graph from the resident graph (the window set by dmvio_hip_ba_set_window must have the graph's number of keyframes).  Stream-ordered: unlike
 * dmvio_hip_ba_set_graph it does not wait for its uploads (they are staged in the handle's pinned memory and enqueued in front of whatever uses them).  After an optimisation the caller
 * hands the new inverse depths back with dmvio_hip_graph_set_idepths(g, N, <idepth of dmvio_hip_ba_get_points>): same order. */
int dmvio_hip_ba_set_graph_from(dmvio_hip_ba* ba, dmvio_hip_graph* g);
/* Residuals that ARRIVE linearised (EFResidual::isLinearized with EFResidual::J and ::res_toZeroF, EnergyFunctionalStructs.h:63-87 — linearised by the reference itself, or by
 * an earlier window of this library) right after dmvio_hip_ba_set_graph: R flags in the graph's residual order, J74 = R x 74 floats (RawResidualJacobian in
 * dmvio_hip_ba_get_full_jacobians' layout: resF 8 | Jpdxi 2x6 | Jpdc 2x4 | Jpdd 2 | JIdx 2x8 | JabF 2x8 | JIdx2 4 | JabJIdx 4 | Jab2 4), res_toZeroF = R x 8; only the rows
 * of flagged residuals are read.  They become what dmvio_hip_ba_fix_linearization leaves behind: active, ResState::IN, out of activeResiduals, served by accumulateLF_MT /
 * addPoint<1> and calcLEnergyPt on the host loop, the device loop, in a batch and on a sharded window (collective there, like _fix_linearization).  The record the
 * accumulation reads (JpJdF of EFResidual::takeDataF, EnergyFunctionalStructs.cpp:39-49, Hdd / Hcd contributions) is formed from J by the linearisation kernel's own
 * expressions: a graph that carries them over gives the bits of the graph they were linearised on (tests/test_ba_gpu.py).  With a resident graph:
 * dmvio_hip_graph_set_residual_linearized per residual, then dmvio_hip_ba_set_graph_from hands them over in the same call.  n_linearized may be NULL. */
int dmvio_hip_ba_set_linearized_residuals(dmvio_hip_ba* ba, int R, const unsigned char* isLinearized, const float* J74, const float* res_toZeroF, int* n_linearized);
/* ... and back: flags / J / res_toZeroF of the graph's residuals as they stand (rows of the others zeroed; any output may be NULL) */
int dmvio_hip_ba_get_linearized_residuals(dmvio_hip_ba* ba, int R, unsigned char* isLinearized, float* J74, float* res_toZeroF);
/* A bare EFResidual::isLinearized flag (R flags, same order) WITHOUT the frozen Jacobian and res_toZeroF cannot be served — what accumulateLF_MT / addPoint<1> / calcLEnergyPt
 * read (EnergyFunctional.cpp:223-233, 349-431, AccumulatedTopHessian.cpp:84-98) would be missing.  A graph flagged this way is REFUSED: the call returns an error, the graph is
 * dropped (every later call on it fails until the next dmvio_hip_ba_set_graph) — it is never silently optimised without that energy term.  All flags zero: no effect.  Hand
 * linearised residuals over with dmvio_hip_ba_set_linearized_residuals. */
int dmvio_hip_ba_set_residual_flags(dmvio_hip_ba* ba, int R, const unsigned char* isLinearized);
/* EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:85-113) for the ACTIVE residuals with res_mask != 0 (R flags, graph order), at the window's current state:
 * res_toZeroF = resF - [JI Jp | Jab] delta from the applied Jacobian, isLinearized = true.  From then on the residual is no member of activeResiduals
 * (FullSystemOptimize.cpp:436-446: not relinearised, not applied, not removed, outside the photometric energy and the newest keyframe's threshold) and enters
 *   - every system through accumulateLF_MT / AccumulatedTopHessianSSE::addPoint<1> (EnergyFunctional.cpp:223-233, AccumulatedTopHessian.cpp:52-58,84-98): H_L / b_L with
 *     resApprox = res_toZeroF + J delta, Hdd_accLF / bd_accLF / Hcd_accLF into the points' Schur terms,
 *   - the energy through calcLEnergyPt (EnergyFunctional.cpp:349-409),
 * until the next dmvio_hip_ba_set_graph, or until dmvio_hip_ba_marginalize_points relinearises its point (FullSystem.cpp:840-843).  Requires the Jacobians of the APPLIED
 * linearisation: dmvio_hip_ba_keep_jacobians(ba, 1) before the HOST-DRIVEN dmvio_hip_ba_optimize (or linearize(fix) / linearize + apply) that precedes this call (the
 * device-resident loop does not write them).  A graph that carries such residuals is optimised by the host-driven loop, by the device-resident loop
 * (dmvio_hip_ba_set_device_loop: k_ba_solve adds H_L / b_L, the linearised energy and the three accumulation passes run on the device) and inside
 * dmvio_hip_ba_optimize_batch beside windows without them.  On a window sharded over ranks (dmvio_hip_ba_set_comm before this call) the call is COLLECTIVE: every rank
 * passes the flags of its own residuals, the ranks' counts are summed, and from then on every rank runs the three accumulation passes — each with its all-reduce — and one more
 * one-double all-reduce per calcLEnergyF of calcLEnergyPt's term (tests/test_sharded_ba_gpu.py, world 2).  n_linearized (may be NULL): linearised residuals of the graph
 * (of this rank's part of it) after the call. */
int dmvio_hip_ba_fix_linearization(dmvio_hip_ba* ba, int R, const unsigned char* res_mask, int* n_linearized);
/* accumulateLF_MT's result as the reference returns it (EnergyFunctional.cpp:223-233: the stitched system of the linearised residuals plus the frame / calibration priors,
 * AccumulatedTopHessian.cpp:292-302), n x n and n doubles, for the state of the last accumulation (dmvio_hip_ba_accumulate / _solve / _gn_iteration / _optimize) */
int dmvio_hip_ba_get_lf_system(dmvio_hip_ba* ba, double* HL, double* bL);
/* resetOOB of every residual (FullSystemOptimize.cpp:431-448) */
int dmvio_hip_ba_activate_all(dmvio_hip_ba* ba);
/* FullSystem::linearizeAll(fixLinearization) (FullSystemOptimize.cpp:150-218): PointFrameResidual::linearize over all residuals,
 * energy sum, setNewFrameEnergyTH; fix != 0 also applies the results (applyRes(true)) and — like the reference (:176-212) — takes every residual that is not active
 * afterwards out of the graph: it stays OOB / inactive for all later linearisations, activate_all and marginalisations until the next dmvio_hip_ba_set_graph. */
int dmvio_hip_ba_linearize(dmvio_hip_ba* ba, int fix, double* energy);
/* FullSystem::applyRes_Reductor(true) (FullSystemOptimize.cpp:91-95) */
int dmvio_hip_ba_apply(dmvio_hip_ba* ba);
/* per-residual state_NewState (0 IN, 1 OOB, 2 OUTLIER), state_NewEnergy, state_NewEnergyWithOutlier, efResidual->isActive(), centerProjectedTo */
int dmvio_hip_ba_get_res_state(dmvio_hip_ba* ba, unsigned char* newState, float* newEnergy, float* newEnergyWO, unsigned char* active, float* center3);
/* RawResidualJacobian of the last linearisation, 74 floats per residual in the member order of RawResidualJacobian.h:32-61 (parity / debug).
 * The accumulation works on a 52-float compact record; the full Jacobian is only written while dmvio_hip_ba_keep_jacobians(ba, 1) is on. */
int dmvio_hip_ba_keep_jacobians(dmvio_hip_ba* ba, int on);
int dmvio_hip_ba_get_jacobians(dmvio_hip_ba* ba, float* J74);
/* Order of the fp32 accumulation (AccumulatedTopHessianSSE / AccumulatedSCHessianSSE): k partial accumulators per (host,target) /
 * (host,t1,t2) bucket, added in double like the per-worker accumulators of the reference's multi-threaded mode
 * (AccumulatedTopHessian.h:91-139, NUM_THREADS 6) but with a fixed member-to-partial assignment, so results are reproducible run to run.
 * k = 1 replays the reference's single-threaded order bit for bit (incl. the 1k / 1M shift-up) at the price of a k-times longer dependent
 * chain per bucket.  Default 4 (DMVIO_HIP_BA_EXACT=1 in the environment makes 1 the default).  Takes effect with the next set_graph. */
int dmvio_hip_ba_set_accumulators(dmvio_hip_ba* ba, int k);
int dmvio_hip_ba_get_frame_energy_th(dmvio_hip_ba* ba, float* th);
/* accumulateAF_MT + accumulateSCF_MT with adjoint stitching (EnergyFunctional.cpp:201-265): H_A, b_A, H_sc, b_sc ((4+8F)^2 / (4+8F), double)
 * — the matrices handed to BAGTSAMIntegration::computeBAUpdate in VIO mode; resInA = number of active residuals. */
int dmvio_hip_ba_accumulate(dmvio_hip_ba* ba, double* HA, double* bA, double* Hsc, double* bsc, int* resInA);
/* EFPoint::Hdd_accAF, bd_accAF, Hcd_accAF, HdiF, bdSumF (EnergyFunctionalStructs.h:118-128) */
int dmvio_hip_ba_get_point_acc(dmvio_hip_ba* ba, float* Hdd, float* bd, float* Hcd4, float* HdiF, float* bdSumF);
/* EnergyFunctional::solveSystemF (EnergyFunctional.cpp:841-996), no-GTSAM branch: accumulate, damped Jacobi-scaled LDLT,
 * orthogonalisation from iteration 2, resubstituteF_MT.  x_out = lastX (minus the step). */
int dmvio_hip_ba_solve(dmvio_hip_ba* ba, int iteration, double lambda, double* x_out);
/* EnergyFunctional::resubstituteF_MT with a caller-provided x (e.g. from GTSAM) */
int dmvio_hip_ba_resubstitute(dmvio_hip_ba* ba, const double* x);
int dmvio_hip_ba_get_points(dmvio_hip_ba* ba, float* idepth, float* step);
int dmvio_hip_ba_get_frame(dmvio_hip_ba* ba, int f, double pose7_w2c[7], double aff[2], double state10[10]);
int dmvio_hip_ba_get_calib(dmvio_hip_ba* ba, double fxfycxcy[4]);
/* CalibHessian::value / value_zero in the reference's unscaled units (what dmvio_hip_ba_set_calib_values takes; value_scaled = SCALE_* x value is dmvio_hip_ba_get_calib) */
int dmvio_hip_ba_get_calib_values(dmvio_hip_ba* ba, double value[4], double value_zero[4]);
/* EnergyFunctional::resInA: the number of active residuals the accumulation of the last solveSystemF saw (EnergyFunctional.cpp:209; the denominator of
 * statistics_lastFineTrackRMSE, FullSystemOptimize.cpp:620) */
int dmvio_hip_ba_get_res_in_a(dmvio_hip_ba* ba, int* resInA);
/* one Gauss-Newton iteration = the loop body of FullSystem::optimize (FullSystemOptimize.cpp:485-586); lastE = {E_A, E_L, E_M} in/out */
int dmvio_hip_ba_gn_iteration(dmvio_hip_ba* ba, int iteration, double* lambda_io, double lastE[3], int* accepted);
/* Measurement: the five kernels of an ACCEPTED Gauss-Newton iteration on the current state (linearise + decision pass, applyRes + per-point sums, accumulate, adjoint
 * stitch, gather), `reps` times with HIP events recorded on the handle's stream between the launches; us5 = mean microseconds per kernel (each includes the gap to
 * the next launch).  What bench.py's ba.roofline is computed from. */
int dmvio_hip_ba_profile_chain(dmvio_hip_ba* ba, int reps, float us5[5]);
/* Diagnostics: in-kernel timeline of the last decision pass (energy sum, newest keyframe's threshold, accept test — taken by the last
 * workgroup of the linearisation kernel): 100 MHz ticks since that workgroup started: pass begin, energy summed, threshold keys loaded, done. */
int dmvio_hip_ba_last_decide_ticks(dmvio_hip_ba* ba, int ticks4[4]);
/* ---- One window over several GPUs: every rank holds all keyframes of the window and ITS share of the points (set_graph with the
 * rank's points only).  With a communicator set, dmvio_hip_ba_linearize / _accumulate / _gn_iteration / _optimize run the sharded iteration
 * themselves: per linearisation ONE all-gather of the per-rank records [local energy | residual energies of the newest keyframe] (the accept
 * test and setNewFrameEnergyTH, FullSystemOptimize.cpp:96-149,553, are evaluated over their union, identically on every rank), per accumulation ONE
 * all-reduce (fp64 sum) of the packed system [H_A | b_A | H_sc | b_sc | resInA] in HBM — both enqueued on the handle's stream between the
 * kernels that produce and consume the buffers.  Every rank then solves the identical reduced system (EnergyFunctional::solveSystemF replaces
 * nothing here: the reference has no multi-device path; its multi-THREADED accumulation, IndexThreadReduce + per-thread accumulators summed in
 * AccumulatedTopHessian::stitchDoubleMT, AccumulatedTopHessian.h:91-139, is the structure this follows).
 *   nccl_comm: an ncclComm_t of RCCL whose rank `rank` lives on this handle's device (not owned; NULL with world 0 detaches).
 *   Every rank must issue the same sequence of BA calls. */
int dmvio_hip_ba_set_comm(dmvio_hip_ba* ba, void* nccl_comm, int rank, int world);
/* Measurement (RCCL transport): HIP events around the sharded iteration's collectives on the BA stream.  _comm_timing(1) starts collecting and clears the counters;
 * _comm_times waits for the stream and returns the mean microseconds of [all-reduce of the packed system | all-gather of the decision records] over the first 64 of each,
 * and how many of each were issued since. */
int dmvio_hip_ba_comm_timing(dmvio_hip_ba* ba, int on);
int dmvio_hip_ba_comm_times(dmvio_hip_ba* ba, double mean_us2[2], long issued2[2]);
/* The partition policy of the sharded window (north_star: "one keyframe per GPU"): which rank owns which point.  Points follow their HOST keyframe
 * (FrameHessian::pointHessians, HessianBlocks.h:138; EFFrame::points, EnergyFunctionalStructs.h:160): whole keyframes are dealt to the ranks, largest first, each to the
 * currently lightest rank (ties: lower keyframe index / lower rank first); when the heaviest rank would then hold more than max_imbalance x N / world points (the newest
 * keyframe hosts no active points, old ones few: typical for world > 4) the split falls back to `world` equal contiguous point ranges [N r / world, N (r + 1) / world).
 * host[N]: host keyframe index per point (>= 0); owner_out[N]: owning rank per point.  Host-only, needs no device.  Returns 0 (split by keyframe), 1 (equal ranges), < 0 on
 * error.  max_imbalance <= 0 selects the default 1.25.  A rank then passes ITS points (and their residuals) to dmvio_hip_ba_set_graph. */
int dmvio_hip_ba_partition_points(const int* host, int N, int world, double max_imbalance, int* owner_out);
/* Same protocol over a caller-provided transport (MPI, gloo, ...): buffers are staged through host memory.  Both callbacks return 0 on success.
 *   allreduce_sum_f64: element-wise sum over all ranks, in place, identical result on every rank.
 *   allgather: `bytes` bytes per rank, rank order, into out[world * bytes]. */
typedef struct dmvio_hip_comm_callbacks {
  void* user;
  int (*allreduce_sum_f64)(void* user, double* buf, size_t count);
  int (*allgather)(void* user, const void* in, void* out, size_t bytes);
} dmvio_hip_comm_callbacks;
int dmvio_hip_ba_set_comm_callbacks(dmvio_hip_ba* ba, const dmvio_hip_comm_callbacks* cb, int rank, int world);
/* Hypothesis-parallel FullSystem::trackNewCoarse over several GPUs (SURVEY.md 8e — the one split coarse tracking has: it does not shard by point): every rank holds the
 * same reference template and the same new frame; with a communicator set, dmvio_hip_tracker_track_new_coarse runs try 0 on every rank (identical bits; no exchange when
 * it already ends the loop) and splits the remaining tries round-robin over the ranks, ONE all-reduce (fp64 sum of 20 doubles per try, each written by exactly one rank)
 * hands every rank all results, and the reference's sequential abort / winner rule (FullSystem.cpp:419-489) is replayed identically everywhere.  nccl_comm: an
 * ncclComm_t whose rank `rank` lives on this tracker's device (not owned); world <= 1 or NULL detaches.  The callbacks form uses allreduce_sum_f64 of
 * the dmvio_hip_comm_callbacks struct above.  Collective: every rank must make the same call.  Every rank ends with the same bits; against the single-device call the
 * winner and the number of tries are the same and the pose agrees to ~1e-5 (a share of 31 / world tries runs with another cluster size, i.e. another grouping of the fp32
 * partial sums, than one batch of 30).  A rank whose share fails still enters the exchange with its error flagged, and the call then fails on EVERY rank. */
int dmvio_hip_tracker_set_comm(dmvio_hip_tracker* trk, void* nccl_comm, int rank, int world);
int dmvio_hip_tracker_set_comm_callbacks(dmvio_hip_tracker* trk, const dmvio_hip_comm_callbacks* cb, int rank, int world);
/* Test hook (no counterpart in the reference): with on != 0 a LATER dmvio_hip_tracker_set_comm / _set_comm_callbacks with world == 1 keeps the split path — every try is
 * this rank's, the all-reduce over one rank is the identity — so the whole exchange (pinned staging, device buffer, ncclAllReduce on the context's stream) can be
 * exercised on a one-device box.  Off by default; nothing in the environment switches it on. */
int dmvio_hip_tracker_debug_split_single_rank(dmvio_hip_tracker* trk, int on);
/* Convenience wrappers over RCCL for callers without their own communicator: ncclGetUniqueId (128 bytes, to be distributed to all ranks by
 * the caller), ncclCommInitRank on the context's device, ncclCommDestroy. */
int dmvio_hip_comm_unique_id(unsigned char id128[128]);
int dmvio_hip_comm_init_rank(dmvio_hip_ctx* ctx, const unsigned char id128[128], int rank, int world, void** nccl_comm_out);
int dmvio_hip_comm_info(void* nccl_comm, int* n_ranks, int* rank);   /* ncclCommCount, ncclCommUserRank */
int dmvio_hip_comm_destroy(void* nccl_comm);

/* Building blocks of a SHARDED GN iteration driven by the caller (points of one keyframe per GPU; the packed systems / energies are summed by the caller
 * with one RCCL all-reduce): backupState, solve of an externally reduced system + resubstitute, doStepFromBackup (sums6 = frame sums
 * A,B,T,R and the local point sums step^2, |idepth_backup|), loadSateBackup, linearizeAll without the setNewFrameEnergyTH tail
 * (returns the newest-frame energies it would use), the threshold setter and the prior / marginalisation energy terms. */
int dmvio_hip_ba_backup(dmvio_hip_ba* ba);
int dmvio_hip_ba_solve_system(dmvio_hip_ba* ba, int iteration, double lambda, const double* HA, const double* bA, const double* Hsc, const double* bsc, double* x_out);
int dmvio_hip_ba_step(dmvio_hip_ba* ba, float stepfac, float sums6[6]);
int dmvio_hip_ba_restore(dmvio_hip_ba* ba);
int dmvio_hip_ba_linearize_local(dmvio_hip_ba* ba, int fix, double* energy, float* new_frame_energies, int* n_new_frame_energies);
int dmvio_hip_ba_set_new_frame_energy_th(dmvio_hip_ba* ba, float th);
int dmvio_hip_ba_energy_terms(dmvio_hip_ba* ba, double* EL, double* EM);
/* FullSystem::optimize(mnumOptIts) (FullSystemOptimize.cpp:417-647): returns statistics_lastFineTrackRMSE in *rmse; the solver is the reference's
 * non-GTSAM branch (EnergyFunctional.cpp:971-973).  trace (may be NULL): 64 rows [E_A, E_L, E_M, accepted], row 0 = the initial state.  After the call the
 * per-point sums (dmvio_hip_ba_get_point_acc / _get_point_hessian) and resInA are those of the LAST solveSystemF, as in the reference. */
int dmvio_hip_ba_optimize(dmvio_hip_ba* ba, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations, double* trace);
/* The reference's DEFAULT solver branch (settings.cpp:37 setting_useGTSAMIntegration = true) inside the same device-resident loop.  The hooks mirror the members of
 * dmvio::BAGTSAMIntegration that EnergyFunctional / FullSystem::optimize call (src/GTSAMIntegration/BAGTSAMIntegration.h), in the reference's call order:
 *   updateBAValues(frames)                          EnergyFunctional.cpp:339 (calcMEnergyF(false): before the loop and after every rejected step)
 *   getBAEnergy(useNewValues)                       EnergyFunctional.cpp:341; the library adds delta.dot(2 bMForGTSAM + HMForGTSAM delta)
 *   updateDynamicWeight(energy, rmse, trackingGood) FullSystemOptimize.cpp:491-503 (iteration 0, or every iteration and once more before the accept test with
 *                                                   updateDynamicWeightDuringOptimization), :594 after the loop; the accept test divides both M-energies by it (:553)
 *   computeBAUpdate(HPassed, b, lambda, frames, HNoLambda) -> x   EnergyFunctional.cpp:958-969.  n = 4 + 8F, matrices n x n row-major, order [calib4 | per frame
 *                                                   trans3 rot3 a b] in the reference's unscaled units; x = MINUS the step; nonzero return aborts the call.  The
 *                                                   library then orthogonalises x from iteration 2 on (:977-981) and back-substitutes on the device
 *   canBreak()                                      FullSystemOptimize.cpp:523 (and-ed with doStepFromBackup's step-norm test; ends the loop from minOptIterations on)
 *   acceptBAUpdate(energy)                          FullSystemOptimize.cpp:569-572
 *   postOptimization(frames)                        FullSystemOptimize.cpp:641
 * `frames` is what the hooks read through std::vector<EFFrame*> in the reference (BAGTSAMIntegration.cpp:97-120, computeEvaluationPointValues): per keyframe
 * PRE_worldToCam, worldToCam_evalPT, get_state(), get_state_zero(), FrameHessian::frameID; calib_value = CalibHessian::value.  Every hook except computeBAUpdate may
 * be NULL (no-op / 0 energy / weight 1 / never break).  The hooks run on the calling thread with the handle locked: they must not call entry points of this handle.
 * dmvio_hip_ba_solve_ldlt is the reference's own solve of that branch's `else` (diagonal pre-scaling + pivoted LDL^T) for hooks that fall back to it. */
typedef struct dmvio_hip_ba_frame_view {
  int frameID, index;
  double PRE_worldToCam7[7], worldToCam_evalPT7[7], state10[10], state_zero10[10];
} dmvio_hip_ba_frame_view;
typedef struct dmvio_hip_ba_callbacks {
  void* user;
  int (*computeBAUpdate)(void* user, int n, const double* HPassed, const double* b, double lambda, const double* HNoLambda, int F, const dmvio_hip_ba_frame_view* frames,
                         const double calib_value[4], double* x_out);
  void (*acceptBAUpdate)(void* user, double energy);
  double (*getBAEnergy)(void* user, int useNewValues);
  void (*updateBAValues)(void* user, int F, const dmvio_hip_ba_frame_view* frames, const double calib_value[4]);
  double (*updateDynamicWeight)(void* user, double energy, double rmse, int coarseTrackingWasGood);
  int (*canBreak)(void* user);
  void (*postOptimization)(void* user, int F, const dmvio_hip_ba_frame_view* frames, const double calib_value[4]);
} dmvio_hip_ba_callbacks;
typedef struct dmvio_hip_ba_vio_options {
  int coarseTrackingWasGood;                  /* frameHessians.back()->shell->trackingWasGood */
  int updateDynamicWeightDuringOptimization;  /* IMUSettings::updateDynamicWeightDuringOptimization */
  int minOptIterations;                       /* setting_minOptIterations (settings.cpp:101, 1); < 0 = that default */
  int resInA_at_entry;                        /* ef->resInA before the call (left by the previous solveSystemF; enters the rmse of the first updateDynamicWeight; 0 before the
                                                 very first solve, which makes that rmse inf in the reference too); < 0: the current count */
  const double* HMForGTSAM;                   /* EnergyFunctional::HMForGTSAM / bMForGTSAM (n x n row-major / n), NULL = zero */
  const double* bMForGTSAM;
} dmvio_hip_ba_vio_options;
int dmvio_hip_ba_optimize_vio(dmvio_hip_ba* ba, int mnumOptIts, const dmvio_hip_ba_callbacks* cb, const dmvio_hip_ba_vio_options* opt, float* rmse, double* finalEnergy,
                              int* iterations, double* trace);
/* ---- Device-resident Gauss-Newton loop, W windows per launch (round 5).  FullSystem::optimize (FullSystemOptimize.cpp:417-647, the non-GTSAM solver branch
 * EnergyFunctional.cpp:971-973) for W independent windows at once: per iteration ONE sequence of kernels serves all windows (each kernel takes its window from blockIdx.y
 * and is gated on that window's own accept / reject decision), the 68x68 solve, the frame step, FrameFramePrecalc, E_L / E_M and the accept test run on the device — no
 * host round trip per iteration, two waits per call.  windows[W]: distinct handles of the batch's context, each with its window set; windows with different keyframe
 * counts run as separate groups.  rmse / finalEnergy / iterations: W entries (may be NULL); trace: W x 64 x 4 or NULL.  A window's result does not depend on the
 * other windows of the call (bit-identical to a batch of one).  Against dmvio_hip_ba_optimize's host-driven loop the device loop differs in the elementary functions of the
 * frame step (device sin / cos / exp within 1 ulp of the C library's) and, unless dmvio_hip_ba_batch_set_exact_backsub(1), in the association of the back substitution. */
typedef struct dmvio_hip_ba_batch dmvio_hip_ba_batch;
dmvio_hip_ba_batch* dmvio_hip_ba_batch_create(dmvio_hip_ctx* ctx, int max_windows);
void dmvio_hip_ba_batch_destroy(dmvio_hip_ba_batch* batch);
int dmvio_hip_ba_optimize_batch(dmvio_hip_ba_batch* batch, int W, dmvio_hip_ba* const* windows, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations, double* trace);
int dmvio_hip_ba_batch_set_exact_backsub(dmvio_hip_ba_batch* batch, int on);
/* HIP-event times of the last dmvio_hip_ba_optimize_batch call on the batch's stream (ms): [0] initial chain + all iterations, [1] the final fix-linearisation,
 * [2] with dmvio_hip_ba_batch_set_profile(1): the stepped linearisation of the second iteration (k_ba_linearize_b over ALL windows of the call: a profiled call runs as one
 * group on one stream) */
int dmvio_hip_ba_batch_last_ms(dmvio_hip_ba_batch* batch, float ms3[3]);
int dmvio_hip_ba_batch_set_profile(dmvio_hip_ba_batch* batch, int on);
/* 0 (default): from 4 windows on the batch is cut into up to three groups of at least two windows (measured best at 16 and 64 windows), one HIP stream each, their launches enqueued stage by stage, group g
 * started behind group g-1's initial linearisation — so that one group's k_ba_solve (one workgroup per window) runs beside the other groups' linearisations / accumulations;
 * k >= 1: at most k groups, up to 8 (1 = the whole batch on one stream).  A profiled call (dmvio_hip_ba_batch_set_profile) always runs as one group.  Results do not depend on it. */
int dmvio_hip_ba_batch_set_streams(dmvio_hip_ba_batch* batch, int streams);
/* the linearisation kernel of a batch of >= 4 windows: 1 (default) = one lane per residual (k_ba_linearize_b1: the lane walks the eight pattern pixels; no redundant
 * geometry — what a grid that fills the device wants), 8 = eight lanes per residual (k_ba_linearize_b: the single window's latency-hiding form).  Same values, same energy
 * partials, same decisions either way. */
int dmvio_hip_ba_batch_set_linearize_lanes(dmvio_hip_ba_batch* batch, int lanes);
/* diagnostics: in-kernel timeline (100 MHz ticks since kernel start) of the first window's k_ba_solve of the LAST iteration of the last call, 12 phase boundaries */
int dmvio_hip_ba_batch_last_solve_ticks(dmvio_hip_ba_batch* batch, int ticks12[12]);
/* diagnostics: how window w's last k_ba_solve of the last call found Eigen's pivot order (EnergyFunctional.cpp:971-973, ldlt()): 0 = ranks of the scaled diagonal (all
 * |values| distinct), 1 = ties replayed (selection with swaps), 2 = NaN on the diagonal (the literal loop) */
int dmvio_hip_ba_batch_last_pivot_branch(dmvio_hip_ba_batch* batch, int w, int* branch);
/* Diagnostics: host clock (us since the call began) at the phase boundaries of the last call's (last group's) loop: [0] streams handed over, [1] per-window tables prepared,
 * [2] whole loop enqueued, [3] loop finished (first wait), [4] states written back, [5] final linearisation enqueued, [6] finished (second wait), [7] results out */
int dmvio_hip_ba_batch_last_host_us(struct dmvio_hip_ba_batch* batch, double us8[8]);
/* Tests / diagnostics: the solve of EnergyFunctional.cpp:971-973 (diagonal pre-scaling, pivoted LDL^T) for a GIVEN n x n system on the device, exactly as the device-resident
 * loop's k_ba_solve runs it — the device counterpart of dmvio_hip_ba_solve_ldlt.  HPassed row-major (lower triangle read), 2 <= n <= 100.  perm_out[n]: the index the
 * transpositions bring to position k; branch_out: as dmvio_hip_ba_batch_last_pivot_branch; zero_out: the first pivot was zero (x = 0).  The three may be NULL. */
int dmvio_hip_ba_debug_solve(dmvio_hip_ctx* ctx, int n, const double* HPassed, const double* b_in, int exact_backsub, double* x_out, int* perm_out, int* branch_out, int* zero_out);
/* dmvio_hip_ba_optimize of this handle through the device-resident loop (a batch of one); 0 (default) = the host-driven loop */
int dmvio_hip_ba_set_device_loop(dmvio_hip_ba* ba, int on);
/* EnergyFunctional::lastX of the window's last solve (n = 4 + 8F doubles; x = MINUS the step) */
int dmvio_hip_ba_get_last_x(dmvio_hip_ba* ba, double* x_out);
int dmvio_hip_ba_solve_ldlt(int n, const double* HPassed, const double* b, double* x_out);
/* the same with the signature of dmvio_hip_ba_callbacks::computeBAUpdate (user, lambda, HNoLambda, frames, calib_value ignored): usable as the hook itself */
int dmvio_hip_ba_hook_ldlt(void* user, int n, const double* HPassed, const double* b, double lambda, const double* HNoLambda, int F, const dmvio_hip_ba_frame_view* frames,
                           const double calib_value[4], double* x_out);
/* PointHessian::idepth_hessian (set by AccumulatedSCHessianSSE::addPoint, AccumulatedSCHessian.cpp:42,50; read by FullSystem::flagPointsForRemoval, FullSystem.cpp:825) */
int dmvio_hip_ba_get_point_hessian(dmvio_hip_ba* ba, float* idepth_hessian);

/* ------------------------------------------------------------------------------------------------------------------------
 * Immature points: candidate points traced along their epipolar line in every new frame.
 *   ImmaturePoint::ImmaturePoint   src/dso/FullSystem/ImmaturePoint.cpp:34-62   -> dmvio_hip_immature_add_points
 *   ImmaturePoint::traceOn         src/dso/FullSystem/ImmaturePoint.cpp:76-437  -> dmvio_hip_immature_trace
 *   FullSystem::traceNewCoarse     src/dso/FullSystem/FullSystem.cpp:541-584    -> dmvio_hip_trace_new_coarse
 * Every point is independent; results are bit-identical to the CPU path.  The handle keeps the points of all keyframes of the window
 * in one structure-of-arrays; host_tag (0..63) selects the per-host table row (KRKi, Kt, affine) of a trace call.
 * status codes = enum ImmaturePointStatus (ImmaturePoint.h:46-52): 0 GOOD, 1 OOB, 2 OUTLIER, 3 SKIPPED, 4 BADCONDITION, 5 UNINITIALIZED. */
typedef struct dmvio_hip_immature dmvio_hip_immature;
dmvio_hip_immature* dmvio_hip_immature_create(dmvio_hip_ctx* ctx, int capacity);
void dmvio_hip_immature_destroy(dmvio_hip_immature* imm);
int dmvio_hip_immature_clear(dmvio_hip_immature* imm);
int dmvio_hip_immature_count(dmvio_hip_immature* imm);
/* constructs n points at integer pixels (u, v) of the keyframe in host_slot (FullSystem::makeNewTraces, FullSystem.cpp:1465-1490);
 * returns the index of the first new point or <0 */
int dmvio_hip_immature_add_points(dmvio_hip_immature* imm, int host_tag, int host_slot, int n, const int* u, const int* v);
int dmvio_hip_immature_get_static(dmvio_hip_immature* imm, float* u, float* v, int* host_tag, float* color8, float* weights8, float* gradH4, float* energyTH);
int dmvio_hip_immature_get_state(dmvio_hip_immature* imm, float* idepth_min, float* idepth_max, float* quality, float* lastTraceUV2,
                                 float* lastTracePixelInterval, int* lastTraceStatus);
int dmvio_hip_immature_set_state(dmvio_hip_immature* imm, const float* idepth_min, const float* idepth_max, const float* quality, const int* lastTraceStatus);
/* traceOn of every point against the frame in new_slot; tables: hostToFrame_KRKi (row-major 3x3), hostToFrame_Kt, hostToFrame_affine per host_tag */
int dmvio_hip_immature_trace(dmvio_hip_immature* imm, int new_slot, int n_hosts, const float* KRKi9, const float* Kt3, const float* aff2);
/* FullSystem::optimizeImmaturePoint (src/dso/FullSystem/FullSystemOptPoint.cpp:51-205) with ImmaturePoint::linearizeResidual
 * (ImmaturePoint.cpp:498-565) for the points with select[i] != 0 (NULL = all): Gauss-Newton on the inverse depth over the residuals
 * to the other F-1 keyframes of the window (host_tag of a point = its keyframe's index; w2c7 = PRE_worldToCam, aff2 = aff_g2l per frame).
 * result[i]: 1 = activate (idepth[i], res_state valid), 0 = not well constrained (stays immature), -1 = delete;
 * res_state[i*F + t] = ResState of the residual to keyframe t (0 IN, 1 OOB, 2 OUTLIER; -1 for the host itself): the IN entries are
 * the PointFrameResiduals the reference creates (FullSystemOptPoint.cpp:178-197). */
int dmvio_hip_immature_optimize(dmvio_hip_immature* imm, int F, const int* frame_slots, const double* w2c7, const double* aff2, const float* exposure,
                                const double fxfycxcy[4], const unsigned char* select, int minObs, int* result, float* idepth, int* res_state);
/* FullSystem::traceNewCoarse: builds the per-host tables from the poses (new frame worldToCam, hosts camToWorld, pose7 = tx ty tz qx qy qz qw),
 * traces, and returns the status histogram counts6 = {good, oob, outlier, skipped, badcondition, uninitialized} */
int dmvio_hip_trace_new_coarse(dmvio_hip_immature* imm, int new_slot, const double new_w2c7[7], const double new_aff[2], float new_exposure, int n_hosts,
                               const double* host_c2w7, const double* host_aff2, const float* host_exposure, const double fxfycxcy[4], int counts6[6]);

/* ------------------------------------------------------------------------------------------------------------------------
 * Initializer: CoarseInitializer::calcResAndGS  src/dso/FullSystem/CoarseInitializer.cpp:331-624.
 * The handle holds the point set of one pyramid level (struct Pnt, CoarseInitializer.h:44-83); every call evaluates the 8-pixel
 * pattern of all points at (refToNew, aff) for the current idepth_new and returns the 8x8 system H_out, b_out, its Schur part
 * H_out_sc, b_out_sc (row-major floats, Mat88f / Vec8f), res3 = (energy, alphaEnergy, num) and — for non-NULL pointers — the
 * per-point fields calcResAndGS writes (energy_new[2n], isGood_new, maxstep, lastHessian_new, JbBuffer_new[10n]).
 * Ki9: the level's inverse intrinsics (Mat33 Ki[lvl], row-major double); fxfycxcy_lvl: fx[lvl] .. cy[lvl]; aff_ab = (a, b) of refToNew_aff. */
typedef struct dmvio_hip_initializer dmvio_hip_initializer;
dmvio_hip_initializer* dmvio_hip_initializer_create(dmvio_hip_ctx* ctx, int capacity);
void dmvio_hip_initializer_destroy(dmvio_hip_initializer* ini);
int dmvio_hip_initializer_set_points(dmvio_hip_initializer* ini, int n, const float* u, const float* v, const float* iR, const unsigned char* isGood,
                                     const float* energy2, const float* outlierTH);
int dmvio_hip_initializer_calc_res_and_gs(dmvio_hip_initializer* ini, int lvl, int first_slot, int new_slot, const double Ki9[9], const float fxfycxcy_lvl[4],
                                          const double refToNew7[7], const double aff_ab[2], const float* idepth_new, float alphaW, float alphaK,
                                          float couplingWeight, double priorY, double priorX, float* H_out64, float* b_out8, float* H_sc64, float* b_sc8,
                                          float res3[3], float* energy_new2, unsigned char* isGood_new, float* maxstep, float* lastHessian_new, float* JbBuffer_new10);

#ifdef __cplusplus
}
#endif
#endif /* DMVIO_HIP_H */
