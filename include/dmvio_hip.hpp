/* C++11 mirror of the member surface FullSystem uses on the photometric alignment path, over the C ABI of dmvio_hip.h.
 *
 * The reference has no plugin / FFI interface on this path — its boundary is the C++ member surface of CoarseTracker and FrameHessian
 * (SURVEY.md section 8b).  These classes keep the reference's names, argument meaning and error behaviour (bool results, NaN residuals, no
 * exceptions) with plain-old-data in place of Eigen / Sophus types, so that an adapter inside the reference is a field-by-field copy:
 *
 *   dmvio_hip::FrameStore::makeImages(slot, color)          <- FrameHessian::makeImages(float* color, CalibHessian*)   HessianBlocks.cpp:128-191
 *   dmvio_hip::CoarseTracker::makeK(fx, fy, cx, cy)          <- CoarseTracker::makeK(CalibHessian*)                      CoarseTracker.cpp:105-134
 *   dmvio_hip::CoarseTracker::setCoarseTrackingRef(...)      <- CoarseTracker::setCoarseTrackingRef(frameHessians)       CoarseTracker.cpp:524-538
 *   dmvio_hip::CoarseTracker::trackNewestCoarse(...)         <- CoarseTracker::trackNewestCoarse(newFH, lastToNew_out, aff_g2l_out, coarsestLvl,
 *                                                               minResForAbort)                                          CoarseTracker.cpp:539-770
 *   members lastResiduals, lastFlowIndicators, refFrameID    <- CoarseTracker.h:83-91
 *
 * Header-only; link with -ldmvio_hip.  Not thread-safe per object, like the reference's tracker (one tracker per thread). */
#ifndef DMVIO_HIP_HPP
#define DMVIO_HIP_HPP
#include <cmath>
#include <functional>
#include <string>
#include <vector>
#include "dmvio_hip.h"

namespace dmvio_hip {

/* Sophus::SE3d as [translation | unit quaternion x y z w] (the pose7 of the C ABI) */
struct SE3 {
  double t[3];
  double q[4];
  SE3() { t[0] = t[1] = t[2] = 0; q[0] = q[1] = q[2] = 0; q[3] = 1; }
  void toPose7(double p[7]) const { for (int i = 0; i < 3; i++) p[i] = t[i]; for (int i = 0; i < 4; i++) p[3 + i] = q[i]; }
  void fromPose7(const double p[7]) { for (int i = 0; i < 3; i++) t[i] = p[i]; for (int i = 0; i < 4; i++) q[i] = p[3 + i]; }
};
/* dso::AffLight (util/NumType.h:166-192) */
struct AffLight {
  double a, b;
  AffLight() : a(0), b(0) {}
  AffLight(double a_, double b_) : a(a_), b(b_) {}
};

inline std::string lastError() { const char* e = dmvio_hip_last_error(); return e ? e : ""; }

/* The resident image pyramids: one slot per FrameHessian that is alive (its dIp pyramid in the reference). */
class FrameStore {
 public:
  FrameStore(int device, int w, int h, int n_slots) : ctx_(dmvio_hip_create(device, w, h, n_slots)), w_(w), h_(h) {}
  ~FrameStore() { if (ctx_) dmvio_hip_destroy(ctx_); }
  FrameStore(const FrameStore&) = delete;
  FrameStore& operator=(const FrameStore&) = delete;
  bool valid() const { return ctx_ != nullptr; }
  int pyrLevelsUsed() const { return ctx_ ? dmvio_hip_pyr_levels(ctx_) : 0; }
  int w() const { return w_; }
  int h() const { return h_; }
  /* FrameHessian::makeImages: irradiance image (w*h floats) -> pyramid of the slot */
  bool makeImages(int slot, const float* color) { return ctx_ && dmvio_hip_frame_upload(ctx_, slot, color) == 0; }
  /* the same for images that stay resident in device memory: level 0 is the image itself */
  bool makeImagesInPlace(int n, const int* slots, const float* device_images, size_t stride_bytes) {
    return ctx_ && dmvio_hip_frames_attach_device_batch(ctx_, n, slots, device_images, stride_bytes) == 0;
  }
  /* FrameHessian::absSquaredGrad[0..2] of a resident frame (HessianBlocks.cpp:169-189), what PixelSelector::makeMaps reads; B = CalibHessian::B (256 floats) for the
   * getBGradOnly weights of setting_gammaWeightsPixelSelect == 1, or NULL.  out[l] must hold (w >> l) * (h >> l) floats. */
  bool absSquaredGrad(int slot, const float* B, float* const out[3]) { return ctx_ && dmvio_hip_frame_abs_squared_grad(ctx_, slot, 3, B, out) == 0; }
  dmvio_hip_ctx* handle() const { return ctx_; }

 private:
  dmvio_hip_ctx* ctx_;
  int w_, h_;
};

/* One active point of the reference keyframe as setCoarseTrackingRef reads it (CoarseTracker.cpp:144-161): centerProjectedTo of the
 * residual that targets lastRef, and efPoint->HdiF. */
struct RefPoint { float u, v, idepth, HdiF; };

class CoarseTracker {
 public:
  explicit CoarseTracker(FrameStore& frames) : refFrameID(-1), frames_(frames), trk_(frames.valid() ? dmvio_hip_tracker_create(frames.handle()) : nullptr), refSlot_(-1) {
    for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = 1000;
  }
  ~CoarseTracker() { if (trk_) dmvio_hip_tracker_destroy(trk_); }
  CoarseTracker(const CoarseTracker&) = delete;
  CoarseTracker& operator=(const CoarseTracker&) = delete;

  bool makeK(float fx, float fy, float cx, float cy) {
    const float K[4] = {fx, fy, cx, cy};
    return trk_ && dmvio_hip_tracker_make_k(trk_, K) == 0;
  }
  /* lastRef = the newest keyframe (its slot, frameID, ab_exposure, aff_g2l); points = the active points with an IN residual into it */
  bool setCoarseTrackingRef(int refSlot, int frameID, float ab_exposure, const AffLight& aff_g2l, const std::vector<RefPoint>& points) {
    if (!trk_) return false;
    const size_t n = points.size();
    std::vector<float> u(n), v(n), id(n), hd(n);
    for (size_t i = 0; i < n; i++) { u[i] = points[i].u; v[i] = points[i].v; id[i] = points[i].idepth; hd[i] = points[i].HdiF; }
    if (dmvio_hip_tracker_set_ref(trk_, refSlot, ab_exposure, aff_g2l.a, aff_g2l.b, (int)n, u.data(), v.data(), id.data(), hd.data()) != 0) return false;
    refSlot_ = refSlot; refFrameID = frameID; lastRef_aff_g2l = aff_g2l;
    return true;
  }
  /* idepth[lvl] / weightSums[lvl] of the current template (w_lvl * h_lvl floats each): the arrays debugPlotIDepthMap / debugPlotIDepthMapFloat read (CoarseTracker.cpp:772-880) */
  bool idepthMap(int lvl, std::vector<float>& idepth, std::vector<float>& weightSums) const {
    if (!trk_ || lvl < 0 || lvl >= frames_.pyrLevelsUsed()) return false;
    const size_t n = (size_t)(frames_.w() >> lvl) * (frames_.h() >> lvl);
    idepth.resize(n); weightSums.resize(n);
    return dmvio_hip_tracker_get_idepth_map(trk_, lvl, idepth.data(), weightSums.data()) == 0;
  }
  /* Returns trackingGood; lastToNew_out / aff_g2l_out are written only when every level finished (CoarseTracker.cpp:731-760); a device or
   * argument error reads as "tracking failed" (the adapter sets isLost), lastError() tells why. */
  bool trackNewestCoarse(int newSlot, float new_ab_exposure, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, const double minResForAbort[5]) {
    if (!trk_) return false;
    double pose7[7], aff[2] = {aff_g2l_out.a, aff_g2l_out.b};
    lastToNew_out.toPose7(pose7);
    int good = 0;
    if (dmvio_hip_tracker_track(trk_, newSlot, new_ab_exposure, pose7, aff, coarsestLvl, minResForAbort, lastResiduals, lastFlowIndicators, lastH, lastb, &good) != 0) return false;
    lastToNew_out.fromPose7(pose7);
    aff_g2l_out = AffLight(aff[0], aff[1]);
    return good != 0;
  }
  /* The reference's default branch (setting_useIMU, CoarseTracker.cpp:612-637): every LM step is computed by the caller.  The three
   * members mirror what the tracker calls on dmvio::IMUIntegration (IMUIntegration.hpp:106-112):
   *   computeCoarseUpdate(H, b, extrapFac, lambda, incA, incB, incNorm) -> refToNew_new      H: 8x8 row-major, scaled, [trans rot a b]
   *   acceptCoarseUpdate()
   *   addVisualToCoarseGraph(H, b, trackingGood)
   * An empty computeCoarseUpdate runs the visual-only step (what the reference does while !isCoarseInitialized()). */
  struct CoarseIMUHooks {
    std::function<SE3(const double* H, const double* b, float extrapFac, float lambda, const SE3& refToNew_current, double& incA, double& incB, double& incNorm)>
        computeCoarseUpdate;
    std::function<void()> acceptCoarseUpdate;
    std::function<void(const double* H, const double* b, bool trackingGood)> addVisualToCoarseGraph;
  };
  bool trackNewestCoarse(int newSlot, float new_ab_exposure, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, const double minResForAbort[5],
                         const CoarseIMUHooks& imu) {
    if (!trk_) return false;
    dmvio_hip_coarse_callbacks cb;
    cb.user = const_cast<CoarseIMUHooks*>(&imu);
    cb.update = imu.computeCoarseUpdate ? &CoarseTracker::updateThunk : nullptr;
    cb.accept = imu.acceptCoarseUpdate ? &CoarseTracker::acceptThunk : nullptr;
    cb.visual = imu.addVisualToCoarseGraph ? &CoarseTracker::visualThunk : nullptr;
    double pose7[7], aff[2] = {aff_g2l_out.a, aff_g2l_out.b};
    lastToNew_out.toPose7(pose7);
    int good = 0;
    if (dmvio_hip_tracker_track_vio(trk_, newSlot, new_ab_exposure, pose7, aff, coarsestLvl, minResForAbort, &cb, lastResiduals, lastFlowIndicators, lastH, lastb, &good,
                                    &lastEvaluations) != 0)
      return false;
    lastToNew_out.fromPose7(pose7);
    aff_g2l_out = AffLight(aff[0], aff[1]);
    return good != 0;
  }
  int lastEvaluations = 0;   /* calcRes + calcGSSSE passes of the last hand-off track */
  int pc_n(int lvl) const { return trk_ ? dmvio_hip_tracker_pc_n(trk_, lvl) : 0; }

  /* CoarseTracker.h:83-91 */
  double lastResiduals[5];
  double lastFlowIndicators[3];
  AffLight lastRef_aff_g2l;
  int refFrameID;
  /* the scaled 8x8 system at the accepted state of the last level — what IMUIntegration::addVisualToCoarseGraph receives (CoarseTracker.cpp:766) */
  double lastH[64], lastb[8];

 private:
  static int updateThunk(void* user, const double H[64], const double b[8], float extrapFac, float lambda, const double pose7_cur[7], const double*, double pose7_new[7],
                         double* incA, double* incB, double* incNorm) {
    const CoarseIMUHooks* h = static_cast<const CoarseIMUHooks*>(user);
    SE3 cur; cur.fromPose7(pose7_cur);
    const SE3 nxt = h->computeCoarseUpdate(H, b, extrapFac, lambda, cur, *incA, *incB, *incNorm);
    nxt.toPose7(pose7_new);
    return 0;
  }
  static void acceptThunk(void* user) { static_cast<const CoarseIMUHooks*>(user)->acceptCoarseUpdate(); }
  static void visualThunk(void* user, const double H[64], const double b[8], int good) { static_cast<const CoarseIMUHooks*>(user)->addVisualToCoarseGraph(H, b, good != 0); }
  FrameStore& frames_;
  dmvio_hip_tracker* trk_;
  int refSlot_;
};

/* ---- mapping side: the window FullSystem::optimize works on -------------------------------------------------------------------------
 *   KeyFrame      <- FrameHessian (slot of its pyramid, worldToCam_evalPT, aff_g2l, ab_exposure, frameID)       HessianBlocks.h:113-307
 *   ActivePoint   <- PointHessian (host frame index, u, v, idepth, color[8], weights[8], hasDepthPrior) with the targets of its residuals
 *                                                                                                                HessianBlocks.h:413-508
 *   WindowOptimizer::optimize(mnumOptIts) <- float FullSystem::optimize(int)                                     FullSystemOptimize.cpp:417-647 */
struct KeyFrame {
  int slot;
  SE3 worldToCam_evalPT;
  AffLight aff_g2l;
  float ab_exposure;
  int frameID;
};
struct ActivePoint {
  int host;
  float u, v, idepth;
  float color[8], weights[8];
  bool hasDepthPrior;
  std::vector<int> targets;   /* frame indices of its PointFrameResiduals */
};

/* The window's point / residual graph kept resident across keyframes: EnergyFunctional's own mutators over a host-side mirror (dmvio_hip_graph_*, include/dmvio_hip.h
 * "window graph"; EnergyFunctional.cpp:435-518, 641-646, 766-782).  Elements are addressed as the reference addresses its objects: a keyframe by EFFrame::idx, a point by
 * (host keyframe, EFPoint::idxInPoints), a residual by (point, EFResidual::idxInAll); insert* return the index the reference assigns in the same call, -1 on error. */
class WindowGraph {
 public:
  WindowGraph() : g_(dmvio_hip_graph_create()) {}
  ~WindowGraph() { if (g_) dmvio_hip_graph_destroy(g_); }
  WindowGraph(const WindowGraph&) = delete;
  WindowGraph& operator=(const WindowGraph&) = delete;
  bool valid() const { return g_ != nullptr; }
  int insertFrame() { return g_ ? dmvio_hip_graph_insert_frame(g_) : -1; }
  bool marginalizeFrame(int idx) { return g_ && dmvio_hip_graph_remove_frame(g_, idx) == 0; }
  int insertPoint(const ActivePoint& p) { return g_ ? dmvio_hip_graph_insert_point(g_, p.host, p.u, p.v, p.idepth, p.color, p.weights, p.hasDepthPrior ? 1 : 0) : -1; }
  bool removePoint(int host, int idxInPoints) { return g_ && dmvio_hip_graph_remove_point(g_, host, idxInPoints) == 0; }
  int insertResidual(int host, int idxInPoints, int target) { return g_ ? dmvio_hip_graph_insert_residual(g_, host, idxInPoints, target) : -1; }
  bool dropResidual(int host, int idxInPoints, int idxInAll) { return g_ && dmvio_hip_graph_drop_residual(g_, host, idxInPoints, idxInAll) == 0; }
  bool setIdepth(int host, int idxInPoints, float idepth) { return g_ && dmvio_hip_graph_set_idepth(g_, host, idxInPoints, idepth) == 0; }
  bool setIdepths(const std::vector<float>& idepth) { return g_ && dmvio_hip_graph_set_idepths(g_, (int)idepth.size(), idepth.data()) == 0; }   /* makeIDX order */
  /* EFResidual::fixLinearizationF's result of one residual (isLinearized = true with its 74-float RawResidualJacobian and res_toZeroF), or isLinearized = false again
   * (J74 == nullptr): the record follows the residual through dropResidual / removePoint and WindowOptimizer::setGraphFrom hands it over */
  bool setResidualLinearized(int host, int idxInPoints, int idxInAll, const float* J74, const float* resToZeroF8) {
    return g_ && dmvio_hip_graph_set_residual_linearized(g_, host, idxInPoints, idxInAll, J74, resToZeroF8) == 0;
  }
  int nLinearized() const { return g_ ? dmvio_hip_graph_linearized_count(g_) : -1; }
  int nFrames() const { int F = 0; return g_ && dmvio_hip_graph_counts(g_, &F, nullptr, nullptr) == 0 ? F : -1; }
  int nPoints() const { int N = 0; return g_ && dmvio_hip_graph_counts(g_, nullptr, &N, nullptr) == 0 ? N : -1; }
  int nResiduals() const { int R = 0; return g_ && dmvio_hip_graph_counts(g_, nullptr, nullptr, &R) == 0 ? R : -1; }
  dmvio_hip_graph* handle() const { return g_; }

 private:
  dmvio_hip_graph* g_;
};

class WindowOptimizer {
 public:
  explicit WindowOptimizer(FrameStore& frames) : lastEnergy(NAN), lastIterations(0), ba_(frames.valid() ? dmvio_hip_ba_create(frames.handle()) : nullptr), F_(0), N_(0) {}
  ~WindowOptimizer() { if (ba_) dmvio_hip_ba_destroy(ba_); }
  WindowOptimizer(const WindowOptimizer&) = delete;
  WindowOptimizer& operator=(const WindowOptimizer&) = delete;

  /* frameHessians in window order (newest last) + Hcalib: EnergyFunctional::insertFrame / setAdjointsF / FullSystem::setPrecalcValues */
  bool setWindow(const std::vector<KeyFrame>& frameHessians, double fx, double fy, double cx, double cy) {
    if (!ba_) return false;
    const int F = (int)frameHessians.size();
    std::vector<int> slots(F), ids(F);
    std::vector<double> poses(7 * (size_t)F), aff(2 * (size_t)F);
    std::vector<float> expo(F);
    for (int f = 0; f < F; f++) {
      slots[f] = frameHessians[f].slot; ids[f] = frameHessians[f].frameID; expo[f] = frameHessians[f].ab_exposure;
      frameHessians[f].worldToCam_evalPT.toPose7(&poses[7 * (size_t)f]);
      aff[2 * (size_t)f] = frameHessians[f].aff_g2l.a; aff[2 * (size_t)f + 1] = frameHessians[f].aff_g2l.b;
    }
    const double K[4] = {fx, fy, cx, cy};
    if (dmvio_hip_ba_set_window(ba_, F, slots.data(), poses.data(), aff.data(), expo.data(), ids.data(), K) != 0) return false;
    F_ = F;
    return true;
  }
  /* all active points in allPoints order with their residuals: EnergyFunctional::makeIDX */
  bool setPoints(const std::vector<ActivePoint>& points) {
    if (!ba_) return false;
    const size_t N = points.size();
    std::vector<int> host(N), rp, rt;
    std::vector<float> u(N), v(N), id(N), col(8 * N), wts(8 * N);
    std::vector<unsigned char> prior(N);
    for (size_t i = 0; i < N; i++) {
      host[i] = points[i].host; u[i] = points[i].u; v[i] = points[i].v; id[i] = points[i].idepth; prior[i] = points[i].hasDepthPrior ? 1 : 0;
      for (int k = 0; k < 8; k++) { col[8 * i + k] = points[i].color[k]; wts[8 * i + k] = points[i].weights[k]; }
      for (size_t r = 0; r < points[i].targets.size(); r++) { rp.push_back((int)i); rt.push_back(points[i].targets[r]); }
    }
    if (dmvio_hip_ba_set_graph(ba_, (int)N, host.data(), u.data(), v.data(), id.data(), col.data(), wts.data(), prior.data(), (int)rp.size(), rp.data(), rt.data()) != 0) return false;
    N_ = (int)N;
    return true;
  }
  /* the same from a resident WindowGraph (no flat arrays are formed by the caller); after optimize(): graph.setIdepths(<idepths()>) */
  bool setPoints(const WindowGraph& graph) {
    if (!ba_ || !graph.valid() || dmvio_hip_ba_set_graph_from(ba_, graph.handle()) != 0) return false;
    N_ = graph.nPoints();
    return true;
  }
  /* returns sqrt(E / (patternNum * resInA)) like the reference; a negative value on a device / argument error (the adapter sets isLost) */
  float optimize(int mnumOptIts) {
    if (!ba_) return -1.0f;
    float rmse = -1.0f;
    if (dmvio_hip_ba_optimize(ba_, mnumOptIts, &rmse, &lastEnergy, &lastIterations, energyTrace) != 0) return -1.0f;
    return rmse;
  }
  /* FullSystem::optimize on the reference's default branch (setting_useGTSAMIntegration): the members of dmvio::BAGTSAMIntegration the loop calls as hooks
   * (computeBAUpdate / getBAEnergy / updateBAValues / updateDynamicWeight / canBreak / acceptBAUpdate / postOptimization, include/dmvio_hip.h), the options what the
   * reference reads beside them (shell->trackingWasGood, updateDynamicWeightDuringOptimization, setting_minOptIterations, ef->resInA, HMForGTSAM / bMForGTSAM) */
  float optimize(int mnumOptIts, const dmvio_hip_ba_callbacks& hooks, const dmvio_hip_ba_vio_options& options) {
    if (!ba_) return -1.0f;
    float rmse = -1.0f;
    if (dmvio_hip_ba_optimize_vio(ba_, mnumOptIts, &hooks, &options, &rmse, &lastEnergy, &lastIterations, energyTrace) != 0) return -1.0f;
    return rmse;
  }
  /* one window over several GPUs: this optimizer holds all keyframes and the rank's share of the points; with a communicator set, optimize() is collective (every rank
   * calls it) and runs the all-reduce of the system and the all-gather of the decision records itself, on its own stream (include/dmvio_hip.h, dmvio_hip_ba_set_comm) */
  bool setCommunicator(void* ncclComm, int rank, int world) { return ba_ && dmvio_hip_ba_set_comm(ba_, ncclComm, rank, world) == 0; }
  /* 1 = the reference's single-threaded accumulation order (bit-identical sums); default 4 partial accumulators per bucket = its multi-threaded structure */
  bool setAccumulators(int k) { return ba_ && dmvio_hip_ba_set_accumulators(ba_, k) == 0; }
  bool frameState(int f, SE3& worldToCam, AffLight& aff_g2l) const {
    double p[7], a[2], st[10];
    if (!ba_ || dmvio_hip_ba_get_frame(ba_, f, p, a, st) != 0) return false;
    worldToCam.fromPose7(p); aff_g2l = AffLight(a[0], a[1]);
    return true;
  }
  bool idepths(std::vector<float>& idepth) const {
    idepth.resize(N_);
    return ba_ && dmvio_hip_ba_get_points(ba_, idepth.data(), nullptr) == 0;
  }
  /* EFResidual::fixLinearizationF for the active residuals with mask != 0 on the resident graph (outside a marginalisation): from then on they ride in every system through
   * accumulateLF_MT / addPoint<1> and in E_L through calcLEnergyPt; needs keepJacobians(true) before the optimize that precedes it.  Returns the number of linearised
   * residuals of the graph, -1 on error. */
  int fixLinearization(const std::vector<unsigned char>& residualMask) {
    int n = -1;
    if (!ba_ || dmvio_hip_ba_fix_linearization(ba_, (int)residualMask.size(), residualMask.data(), &n) != 0) return -1;
    return n;
  }
  /* residuals that ARRIVE linearised with a flat graph (right after setGraph): flags, 74-float Jacobians and res_toZeroF per residual, graph order; and what the window holds,
   * to be handed to the next one.  Returns the number of linearised residuals, -1 on error. */
  int setLinearizedResiduals(const std::vector<unsigned char>& isLinearized, const std::vector<float>& J74, const std::vector<float>& resToZeroF8) {
    int n = -1;
    const size_t R = isLinearized.size();
    if (!ba_ || J74.size() != 74 * R || resToZeroF8.size() != 8 * R) return -1;
    if (dmvio_hip_ba_set_linearized_residuals(ba_, (int)R, isLinearized.data(), J74.data(), resToZeroF8.data(), &n) != 0) return -1;
    return n;
  }
  bool linearizedResiduals(int R, std::vector<unsigned char>& isLinearized, std::vector<float>& J74, std::vector<float>& resToZeroF8) const {
    isLinearized.resize(R); J74.resize(74 * (size_t)R); resToZeroF8.resize(8 * (size_t)R);
    return ba_ && dmvio_hip_ba_get_linearized_residuals(ba_, R, isLinearized.data(), J74.data(), resToZeroF8.data()) == 0;
  }
  bool keepJacobians(bool on) { return ba_ && dmvio_hip_ba_keep_jacobians(ba_, on ? 1 : 0) == 0; }
  dmvio_hip_ba* handle() const { return ba_; }
  double lastEnergy;           /* E_A + E_L + E_M after the last optimize */
  int lastIterations;
  double energyTrace[64 * 4];  /* per iteration: E_A, E_L, E_M, accepted (row 0: before the first step) */

 private:
  dmvio_hip_ba* ba_;
  int F_, N_;
  friend class WindowBatch;
};

/* FullSystem::optimize for several windows per call (dmvio_hip_ba_optimize_batch): the whole Gauss-Newton loop — the 68x68 solve, the frame step and the accept test included —
 * runs on the device for all of them, two host waits per call.  Each WindowOptimizer is set up as for its own optimize(); its results are read with its own getters afterwards. */
class WindowBatch {
 public:
  WindowBatch(dmvio_hip_ctx* ctx, int maxWindows) : b_(dmvio_hip_ba_batch_create(ctx, maxWindows)) {}
  ~WindowBatch() { if (b_) dmvio_hip_ba_batch_destroy(b_); }
  WindowBatch(const WindowBatch&) = delete;
  WindowBatch& operator=(const WindowBatch&) = delete;
  bool valid() const { return b_ != nullptr; }
  /* rmse[i] = sqrt(E / (patternNum * resInA)) of window i like the reference's return value; false on a device / argument error */
  bool optimize(const std::vector<WindowOptimizer*>& windows, int mnumOptIts, std::vector<float>& rmse) {
    if (!b_ || windows.empty()) return false;
    std::vector<dmvio_hip_ba*> hs(windows.size());
    for (size_t i = 0; i < windows.size(); i++) { if (!windows[i] || !windows[i]->ba_) return false; hs[i] = windows[i]->ba_; }
    rmse.assign(windows.size(), -1.0f);
    std::vector<double> energy(windows.size()), trace(windows.size() * 256);
    std::vector<int> its(windows.size());
    if (dmvio_hip_ba_optimize_batch(b_, (int)hs.size(), hs.data(), mnumOptIts, rmse.data(), energy.data(), its.data(), trace.data()) != 0) return false;
    for (size_t i = 0; i < windows.size(); i++) {
      windows[i]->lastEnergy = energy[i]; windows[i]->lastIterations = its[i];
      for (int k = 0; k < 256; k++) windows[i]->energyTrace[k] = trace[i * 256 + k];
    }
    return true;
  }

 private:
  dmvio_hip_ba_batch* b_;
};

}  // namespace dmvio_hip
#endif
