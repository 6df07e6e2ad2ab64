// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library.  PARITY PINNED against the reference's own compiled sources (oracle/_ref/libref.so, oracle/Makefile.ref): precalc, adjoints,
// linearize, thresholds, accumulation, solve, optimize bit for bit (tests/test_ref_pin_cpu.py), 7 recorded windows of a live run (tests/test_ref_replay_cpu.py);
// unpinned only for Eigen's ldlt / SVD / inverse (Sophus is pinned through lie.h).  Construction checks: tests/test_oracle_ba_cpu.py.
//
// Restates the MAPPING half of the hot path of lukasvst/dm-vio (sliding-window photometric bundle adjustment):
//   OWindow::setPrecalcValues   <- FullSystem::setPrecalcValues        src/dso/FullSystem/FullSystem.cpp:1670-1680
//   precalcSet                  <- FrameFramePrecalc::set              src/dso/FullSystem/HessianBlocks.cpp:193-223
//   frameSetState/StateZero     <- FrameHessian::setState/setStateZero src/dso/FullSystem/HessianBlocks.h:179-227, .cpp:74-107
//   OWindow::setAdjointsF       <- EnergyFunctional::setAdjointsF      src/dso/OptimizationBackend/EnergyFunctional.cpp:48-108
//   OWindow::setDeltaF          <- EnergyFunctional::setDeltaF         EnergyFunctional.cpp:175-198
//   linearizeRes                <- PointFrameResidual::linearize       src/dso/FullSystem/Residuals.cpp:78-274
//   projectPoint (x2)           <- projectPoint                        src/dso/FullSystem/ResidualProjections.h:47-87
//   applyRes / takeDataF        <- Residuals.cpp:306-328, OptimizationBackend/EnergyFunctionalStructs.cpp:39-49
//   OWindow::linearizeAll       <- FullSystem::linearizeAll(+_Reductor) src/dso/FullSystem/FullSystemOptimize.cpp:55-88,150-218
//   OWindow::setNewFrameEnergyTH<- FullSystemOptimize.cpp:96-149
//   AccApprox / AccXX / AccX    <- AccumulatorApprox / AccumulatorXX / AccumulatorX   OptimizationBackend/MatrixAccumulators.h:36-237,595-972
//   topAddPoint<mode>           <- AccumulatedTopHessianSSE::addPoint  OptimizationBackend/AccumulatedTopHessian.cpp:39-159
//   topStitch                   <- stitchDoubleInternal + stitchDoubleMT  AccumulatedTopHessian.cpp:241-303, .h:91-139
//   scAddPoint / scStitch       <- AccumulatedSCHessianSSE::addPoint / stitchDoubleInternal  AccumulatedSCHessian.cpp:34-157, .h:91-133
//   OWindow::solveSystemF       <- EnergyFunctional::solveSystemF      EnergyFunctional.cpp:841-996 (useimu=0 / no-GTSAM branch)
//   OWindow::orthogonalize      <- EnergyFunctional::orthogonalize     EnergyFunctional.cpp:784-838
//   OWindow::getNullspaces      <- FullSystem::getNullspaces           FullSystemOptimize.cpp:704-758
//   OWindow::resubstitute       <- resubstituteF_MT / resubstituteFPt  EnergyFunctional.cpp:267-321
//   OWindow::calcLEnergy/MEnergy<- calcLEnergyF_MT/calcLEnergyPt/calcMEnergyF  EnergyFunctional.cpp:324-431
//   OWindow::doStepFromBackup / backupState / loadSateBackup  <- FullSystemOptimize.cpp:224-388
//   OWindow::optimize           <- FullSystem::optimize                FullSystemOptimize.cpp:417-647
// Third-party arithmetic at the edge (absent from /root/reference): Eigen's pivoted LDLT (dense.h) and JacobiSVD
// (EnergyFunctional.cpp:812): the pseudo-inverse projector is restated with a one-sided Jacobi (Hestenes) SVD.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>
#include <emmintrin.h>
#include "lie.h"
#include "dense.h"

namespace orc {

static const float SCALE_IDEPTH = 1.0f, SCALE_XI_ROT = 1.0f, SCALE_XI_TRANS = 1.0f, SCALE_F = 50.0f, SCALE_C = 50.0f, SCALE_A = 10.0f, SCALE_B = 1000.0f;
static const float SCALE_F_INVERSE = 1.0f / SCALE_F, SCALE_C_INVERSE = 1.0f / SCALE_C, SCALE_A_INVERSE = 1.0f / SCALE_A, SCALE_B_INVERSE = 1.0f / SCALE_B;
static const int CPARS = 4, PATTERN = 8, MAXF = 16;   // MAXF: capacity of the per-frame precalc table (setting_maxFrames is a run-time setting, settings.cpp:100)
static const int patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};  // settings.cpp:296 pattern 8

struct BASettings {
  float huberTH = 9, outlierTHSumComponent = 50 * 50, idepthFixPrior = 50 * 50;
  float initialRotPrior = 1e11, initialTransPrior = 1e10, initialAffBPrior = 1e14, initialAffAPrior = 1e14, initialCalibHessian = 5e9;
  float affineOptModeA = 1e12, affineOptModeB = 1e8;
  float frameEnergyTHConstWeight = 0.5, frameEnergyTHN = 0.7f, frameEnergyTHFacMedian = 1.5, overallEnergyTHWeight = 1;
  float thOptIterations = 1.2;
  int minOptIterations = 1;
  double solverModeDelta = 0.00001;
  bool forceAcceptStep = false;
};

enum { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };

struct V3f { float v[3]; };

struct RawJ {
  float resF[8];
  float Jpdxi[2][6];
  float Jpdc[2][4];
  float Jpdd[2];
  float JIdx[2][8];
  float JabF[2][8];
  float JIdx2[2][2], JabJIdx[2][2], Jab2[2][2];
};

struct Precalc {
  float PRE_RTll[9], PRE_KRKiTll[9], PRE_RKiTll[9], PRE_RTll_0[9];
  float PRE_aff_mode[2], PRE_b0_mode;
  float PRE_tTll[3], PRE_KtTll[3], PRE_tTll_0[3];
};

struct OFrame {
  SE3 evalPT;
  double state[10], state_zero[10], state_scaled[10], step[10], state_backup[10];
  SE3 PRE_worldToCam, PRE_camToWorld;
  float ab_exposure = 1, frameEnergyTH = 8 * 8 * PATTERN;
  int frameID = 0;
  const V3f* dI = nullptr;
  double ns_pose[6][6], ns_scale[6], ns_affine[4][2];
  double prior[8], delta[8], delta_prior[8];
  Precalc pre[MAXF];
};

struct OPoint {
  int host;
  float u, v, idepth, idepth_scaled, idepth_zero, idepth_zero_scaled, step = 0, idepth_backup = 0;
  float color[8], weights[8];
  bool hasDepthPrior = false;
  float priorF = 0, deltaF = 0, HdiF = 0, bdSumF = 0;
  float Hdd_accAF = 0, bd_accAF = 0, Hcd_accAF[4] = {0, 0, 0, 0};
  float Hdd_accLF = 0, bd_accLF = 0, Hcd_accLF[4] = {0, 0, 0, 0};
  float idepth_hessian = 0, maxRelBaseline = 0;
  int numGoodResiduals = 0;
  std::vector<int> residuals;
};

struct ORes {
  int point, host, target;
  int state_state = RS_IN, state_NewState = RS_OUTLIER;
  double state_energy = 0, state_NewEnergy = 0, state_NewEnergyWithOutlier = -1;
  bool isNew = true, isActive = false, isLinearized = false, dropped = false;
  RawJ Jnew, Jef;
  float res_toZeroF[8];
  float JpJdF[8];
  float centerProjectedTo[3];
  float projectedTo[8][2];
};

// AccumulatorApprox: 13x13 = [10x10 sym (55) | 10x3 | 3x3 sym (6)], hierarchical 1k/1M shift-up
struct AccApprox {
  float Data[60], Data1k[60], Data1m[60];
  float TR[32], TR1k[32], TR1m[32];
  float BR[8], BR1k[8], BR1m[8];
  float numIn1, numIn1k, numIn1m;
  size_t num;
  float H[13][13];
  void initialize() { memset(this, 0, sizeof(*this)); }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < 60; i++) Data1k[i] = Data[i] + Data1k[i];
      for (int i = 0; i < 32; i++) TR1k[i] = TR[i] + TR1k[i];
      for (int i = 0; i < 8; i++) BR1k[i] = BR[i] + BR1k[i];
      numIn1k += numIn1; numIn1 = 0;
      memset(Data, 0, sizeof(Data)); memset(TR, 0, sizeof(TR)); memset(BR, 0, sizeof(BR));
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < 60; i++) Data1m[i] = Data1k[i] + Data1m[i];
      for (int i = 0; i < 32; i++) TR1m[i] = TR1k[i] + TR1m[i];
      for (int i = 0; i < 8; i++) BR1m[i] = BR1k[i] + BR1m[i];
      numIn1m += numIn1k; numIn1k = 0;
      memset(Data1k, 0, sizeof(Data1k)); memset(TR1k, 0, sizeof(TR1k)); memset(BR1k, 0, sizeof(BR1k));
    }
  }
  // x = [x4; x6], y = [y4; y6];  H10 += [x y] [a b; b c] [x y]^T
  void update(const float* x4, const float* x6, const float* y4, const float* y6, float a, float b, float c) {
    float x[10], y[10];
    for (int i = 0; i < 4; i++) { x[i] = x4[i]; y[i] = y4[i]; }
    for (int i = 0; i < 6; i++) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
    int idx = 0;
    for (int r = 0; r < 10; r++)
      for (int cc = r; cc < 10; cc++) {
        Data[idx] += a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
        idx++;
      }
    num++; numIn1++;
    shiftUp(false);
  }
  void updateTopRight(const float* x4, const float* x6, const float* y4, const float* y6,
                      float TR00, float TR10, float TR01, float TR11, float TR02, float TR12) {
    float x[10], y[10];
    for (int i = 0; i < 4; i++) { x[i] = x4[i]; y[i] = y4[i]; }
    for (int i = 0; i < 6; i++) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
    for (int r = 0; r < 10; r++) {
      TR[3 * r + 0] += x[r] * TR00 + y[r] * TR10;
      TR[3 * r + 1] += x[r] * TR01 + y[r] * TR11;
      TR[3 * r + 2] += x[r] * TR02 + y[r] * TR12;
    }
  }
  void updateBotRight(float a00, float a01, float a02, float a11, float a12, float a22) {
    BR[0] += a00; BR[1] += a01; BR[2] += a02; BR[3] += a11; BR[4] += a12; BR[5] += a22;
  }
  void finish() {
    memset(H, 0, sizeof(H));
    shiftUp(true);
    int idx = 0;
    for (int r = 0; r < 10; r++)
      for (int c = r; c < 10; c++) { H[r][c] = H[c][r] = Data1m[idx]; idx++; }
    idx = 0;
    for (int r = 0; r < 10; r++)
      for (int c = 0; c < 3; c++) { H[r][c + 10] = H[c + 10][r] = TR1m[idx]; idx++; }
    H[10][10] = BR1m[0]; H[10][11] = H[11][10] = BR1m[1]; H[10][12] = H[12][10] = BR1m[2];
    H[11][11] = BR1m[3]; H[11][12] = H[12][11] = BR1m[4]; H[12][12] = BR1m[5];
    num = (size_t)(numIn1 + numIn1k + numIn1m);
  }
};

template <int I, int J>
struct AccXX {
  float A[I][J], A1k[I][J], A1m[I][J];
  float numIn1, numIn1k, numIn1m;
  size_t num;
  void initialize() { memset(this, 0, sizeof(*this)); }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) {
      for (int i = 0; i < I; i++) for (int j = 0; j < J; j++) { A1k[i][j] += A[i][j]; A[i][j] = 0; }
      numIn1k += numIn1; numIn1 = 0;
    }
    if (numIn1k > 1000 || force) {
      for (int i = 0; i < I; i++) for (int j = 0; j < J; j++) { A1m[i][j] += A1k[i][j]; A1k[i][j] = 0; }
      numIn1m += numIn1k; numIn1k = 0;
    }
  }
  void update(const float* L, const float* R, float w) {
    for (int i = 0; i < I; i++) for (int j = 0; j < J; j++) A[i][j] += w * L[i] * R[j];  // A += w*L*R^T
    numIn1++;
    shiftUp(false);
  }
  void finish() { shiftUp(true); num = (size_t)(numIn1 + numIn1k + numIn1m); }
};
template <int I>
struct AccX {
  float A[I], A1k[I], A1m[I];
  float numIn1, numIn1k, numIn1m;
  size_t num;
  void initialize() { memset(this, 0, sizeof(*this)); }
  void shiftUp(bool force) {
    if (numIn1 > 1000 || force) { for (int i = 0; i < I; i++) { A1k[i] += A[i]; A[i] = 0; } numIn1k += numIn1; numIn1 = 0; }
    if (numIn1k > 1000 || force) { for (int i = 0; i < I; i++) { A1m[i] += A1k[i]; A1k[i] = 0; } numIn1m += numIn1k; numIn1k = 0; }
  }
  void update(const float* L, float w) { for (int i = 0; i < I; i++) A[i] += w * L[i]; numIn1++; shiftUp(false); }
  void finish() { shiftUp(true); num = (size_t)(numIn1 + numIn1k + numIn1m); }
};

static inline V3f interp33f(const V3f* mat, float x, float y, int width) {
  int ix = (int)x, iy = (int)y;
  float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const V3f* bp = mat + ix + iy * width;
  V3f r;
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++) r.v[c] = w11 * bp[1 + width].v[c] + w01 * bp[width].v[c] + w10 * bp[1].v[c] + w00 * bp[0].v[c];
  return r;
}
static inline void affFromToD(float eF, float eT, double aF, double bF, double aT, double bT, double out[2]) {
  if (eF == 0 || eT == 0) { eT = eF = 1; }
  double a = std::exp(aT - aF) * eT / eF;
  out[0] = a; out[1] = bT - a * bF;
}
static inline void m33f(const float* A, const float* B, float* C) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3 + 0] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

typedef std::vector<double> Mat;  // row-major n x n or vectors

// Persistent worker pool (IndexThreadReduce, src/dso/util/IndexThreadReduce.h:40-217: workers parked on a condition variable, woken per
// reduce call) — used only by the multi-threaded timing variant of the oracle (cpu_baseline); parity tests run single-threaded.
struct alignas(64) PaddedDouble { double v = 0; };   // one cache line per worker
struct WorkerPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cvGo, cvDone;
  std::function<void(int)> job;
  int generation = 0, pending = 0;
  bool stop = false;
  explicit WorkerPool(int n) {
    for (int t = 0; t < n; t++)
      th.emplace_back([this, t]() {
        int seen = 0;
        for (;;) {
          std::function<void(int)> fn;
          {
            std::unique_lock<std::mutex> lk(mu);
            cvGo.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation; fn = job;
          }
          fn(t);
          { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cvDone.notify_one(); }
        }
      });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cvGo.notify_all();
    for (auto& x : th) x.join();
  }
  void run(const std::function<void(int)>& fn) {
    std::unique_lock<std::mutex> lk(mu);
    job = fn; pending = (int)th.size(); generation++;
    cvGo.notify_all();
    cvDone.wait(lk, [&] { return pending == 0; });
  }
};

struct OWindow {
  BASettings S;
  int w, h;
  float wM3G, hM3G;
  int nF = 0;
  // calibration (CalibHessian, HessianBlocks.h:309-409)
  double c_value[4], c_value_zero[4], c_value_scaled[4], c_step[4], c_value_backup[4], c_vmz[4];
  float c_f[4], c_i[4];  // value_scaledf / value_scaledi
  std::vector<OFrame> frames;
  std::vector<OPoint> points;
  std::vector<ORes> res;
  std::vector<int> activeResiduals;
  // EnergyFunctional state
  std::vector<double> adHost, adTarget;    // nF*nF blocks of 8x8 (row-major), index h + t*nF
  std::vector<float> adHostF, adTargetF, adHTdeltaF;  // adHTdeltaF: nF*nF x 8
  float cDeltaF[4], cPriorF[4];
  double cPrior[4];
  Mat HM, bM;
  Mat lastHS, lastbS, lastX;
  int resInA = 0, resInL = 0;
  std::vector<Mat> ns_pose, ns_scale;
  int nThreads = 1;
  std::unique_ptr<WorkerPool> pool;
  // statistics
  double lastEnergyTrace[64][4];
  int nIterationsDone = 0;

  void calibSetValue(const double v[4]) {
    for (int i = 0; i < 4; i++) c_value[i] = v[i];
    c_value_scaled[0] = SCALE_F * v[0]; c_value_scaled[1] = SCALE_F * v[1]; c_value_scaled[2] = SCALE_C * v[2]; c_value_scaled[3] = SCALE_C * v[3];
    for (int i = 0; i < 4; i++) c_f[i] = (float)c_value_scaled[i];
    c_i[0] = 1.0f / c_f[0]; c_i[1] = 1.0f / c_f[1]; c_i[2] = -c_f[2] / c_f[0]; c_i[3] = -c_f[3] / c_f[1];
    for (int i = 0; i < 4; i++) c_vmz[i] = c_value[i] - c_value_zero[i];
  }
  void calibInitScaled(const double vs[4]) {  // CalibHessian ctor: setValueScaled + value_zero = value
    for (int i = 0; i < 4; i++) c_value_scaled[i] = vs[i];
    for (int i = 0; i < 4; i++) c_f[i] = (float)vs[i];
    c_value[0] = SCALE_F_INVERSE * vs[0]; c_value[1] = SCALE_F_INVERSE * vs[1]; c_value[2] = SCALE_C_INVERSE * vs[2]; c_value[3] = SCALE_C_INVERSE * vs[3];
    c_i[0] = 1.0f / c_f[0]; c_i[1] = 1.0f / c_f[1]; c_i[2] = -c_f[2] / c_f[0]; c_i[3] = -c_f[3] / c_f[1];
    for (int i = 0; i < 4; i++) { c_value_zero[i] = c_value[i]; c_vmz[i] = 0; c_step[i] = 0; c_value_backup[i] = c_value[i]; }
  }

  // ---- FrameHessian state handling
  static void frameSetState(OFrame& f, const double st[10]) {
    for (int i = 0; i < 10; i++) f.state[i] = st[i];
    for (int i = 0; i < 3; i++) f.state_scaled[i] = SCALE_XI_TRANS * st[i];
    for (int i = 3; i < 6; i++) f.state_scaled[i] = SCALE_XI_ROT * st[i];
    f.state_scaled[6] = SCALE_A * st[6]; f.state_scaled[7] = SCALE_B * st[7];
    f.state_scaled[8] = SCALE_A * st[8]; f.state_scaled[9] = SCALE_B * st[9];
    f.PRE_worldToCam = se3Mul(se3Exp(f.state_scaled), f.evalPT);
    f.PRE_camToWorld = se3Inv(f.PRE_worldToCam);
  }
  static void frameSetStateZero(OFrame& f, const double st0[10]) {
    for (int i = 0; i < 10; i++) f.state_zero[i] = st0[i];
    SE3 Tinv = se3Inv(f.evalPT);
    for (int i = 0; i < 6; i++) {
      double eps[6] = {0, 0, 0, 0, 0, 0}, meps[6] = {0, 0, 0, 0, 0, 0};
      eps[i] = 1e-3; meps[i] = -1e-3;
      SE3 P = se3Mul(se3Mul(f.evalPT, se3Exp(eps)), Tinv);
      SE3 M = se3Mul(se3Mul(f.evalPT, se3Exp(meps)), Tinv);
      double lp[6], lm[6]; se3Log(P, lp); se3Log(M, lm);
      for (int r = 0; r < 6; r++) f.ns_pose[r][i] = (lp[r] - lm[r]) / (2e-3);
    }
    SE3 P = f.evalPT; for (int i = 0; i < 3; i++) P.t[i] *= 1.00001; P = se3Mul(P, Tinv);
    SE3 M = f.evalPT; for (int i = 0; i < 3; i++) M.t[i] /= 1.00001; M = se3Mul(M, Tinv);
    double lp[6], lm[6]; se3Log(P, lp); se3Log(M, lm);
    for (int r = 0; r < 6; r++) f.ns_scale[r] = (lp[r] - lm[r]) / (2e-3);
    memset(f.ns_affine, 0, sizeof(f.ns_affine));
    f.ns_affine[0][0] = 1; f.ns_affine[1][0] = 0;
    f.ns_affine[0][1] = 0; f.ns_affine[1][1] = expf((float)(f.state_zero[6] * SCALE_A)) * f.ab_exposure;
  }
  static void frameSetEvalPT(OFrame& f, const SE3& T, const double st[10]) { f.evalPT = T; frameSetState(f, st); frameSetStateZero(f, st); }
  void frameGetPrior(const OFrame& f, double p[10]) const {
    for (int i = 0; i < 10; i++) p[i] = 0;
    if (f.frameID == 0) {
      for (int i = 0; i < 3; i++) p[i] = S.initialTransPrior;
      for (int i = 3; i < 6; i++) p[i] = S.initialRotPrior;
      p[6] = S.initialAffAPrior; p[7] = S.initialAffBPrior;
    } else {
      p[6] = S.affineOptModeA < 0 ? S.initialAffAPrior : S.affineOptModeA;
      p[7] = S.affineOptModeB < 0 ? S.initialAffBPrior : S.affineOptModeB;
    }
    p[8] = S.initialAffAPrior; p[9] = S.initialAffBPrior;
  }
  void frameTakeData(OFrame& f) {  // EFFrame::takeData
    double p[10]; frameGetPrior(f, p);
    for (int i = 0; i < 8; i++) { f.prior[i] = p[i]; f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
  }

  void precalcSet(Precalc& pc, const OFrame& host, const OFrame& target) {
    SE3 l0 = se3Mul(target.evalPT, se3Inv(host.evalPT));
    double R[9]; qToR(l0.q, R);
    for (int i = 0; i < 9; i++) pc.PRE_RTll_0[i] = (float)R[i];
    for (int i = 0; i < 3; i++) pc.PRE_tTll_0[i] = (float)l0.t[i];
    SE3 l = se3Mul(target.PRE_worldToCam, host.PRE_camToWorld);
    qToR(l.q, R);
    for (int i = 0; i < 9; i++) pc.PRE_RTll[i] = (float)R[i];
    for (int i = 0; i < 3; i++) pc.PRE_tTll[i] = (float)l.t[i];
    float K[9] = {c_f[0], 0, c_f[2], 0, c_f[1], c_f[3], 0, 0, 1};
    // K.inverse() (Eigen cofactor formula in float)
    const float a = K[0], e = K[4], c = K[2], ff = K[5];
    const float det = a * (e * 1.0f - ff * 0.0f), invdet = 1.0f / det;
    float Ki[9] = {(e * 1.0f - ff * 0.0f) * invdet, (c * 0.0f - 0.0f * 1.0f) * invdet, (0.0f * ff - c * e) * invdet,
                   (ff * 0.0f - 0.0f * 1.0f) * invdet, (a * 1.0f - c * 0.0f) * invdet, (c * 0.0f - a * ff) * invdet,
                   (0.0f * 0.0f - e * 0.0f) * invdet, (0.0f * 0.0f - a * 0.0f) * invdet, (a * e - 0.0f * 0.0f) * invdet};
    float KR[9]; m33f(K, pc.PRE_RTll, KR);
    m33f(KR, Ki, pc.PRE_KRKiTll);
    m33f(pc.PRE_RTll, Ki, pc.PRE_RKiTll);
    for (int r = 0; r < 3; r++) pc.PRE_KtTll[r] = K[r * 3 + 0] * pc.PRE_tTll[0] + K[r * 3 + 1] * pc.PRE_tTll[1] + K[r * 3 + 2] * pc.PRE_tTll[2];
    double aff[2];
    affFromToD(host.ab_exposure, target.ab_exposure, host.state_scaled[6], host.state_scaled[7], target.state_scaled[6], target.state_scaled[7], aff);
    pc.PRE_aff_mode[0] = (float)aff[0]; pc.PRE_aff_mode[1] = (float)aff[1];
    pc.PRE_b0_mode = (float)(host.state_zero[7] * SCALE_B);
  }
  void setPrecalcValues() {
    for (int hh = 0; hh < nF; hh++) for (int t = 0; t < nF; t++) precalcSet(frames[hh].pre[t], frames[hh], frames[t]);
    setDeltaF();
  }

  void setAdjointsF() {
    adHost.assign((size_t)nF * nF * 64, 0); adTarget.assign((size_t)nF * nF * 64, 0);
    adHostF.assign((size_t)nF * nF * 64, 0); adTargetF.assign((size_t)nF * nF * 64, 0);
    for (int hh = 0; hh < nF; hh++)
      for (int t = 0; t < nF; t++) {
        const OFrame& host = frames[hh]; const OFrame& target = frames[t];
        SE3 hostToTarget = se3Mul(target.evalPT, se3Inv(host.evalPT));
        double Adj[36]; se3Adj(hostToTarget, Adj);
        double AH[64], AT[64];
        for (int i = 0; i < 64; i++) { AH[i] = 0; AT[i] = 0; }
        for (int i = 0; i < 8; i++) { AH[i * 8 + i] = 1; AT[i * 8 + i] = 1; }
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) AH[r * 8 + c] = -Adj[c * 6 + r];  // -Adj^T
        double aff[2];
        affFromToD(host.ab_exposure, target.ab_exposure, host.state_zero[6] * SCALE_A, host.state_zero[7] * SCALE_B,
                   target.state_zero[6] * SCALE_A, target.state_zero[7] * SCALE_B, aff);
        const float affLL0 = (float)aff[0];
        AT[6 * 8 + 6] = -affLL0; AH[6 * 8 + 6] = affLL0; AT[7 * 8 + 7] = -1; AH[7 * 8 + 7] = affLL0;
        const float rs[8] = {SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_A, SCALE_B};
        for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) { AH[r * 8 + c] *= rs[r]; AT[r * 8 + c] *= rs[r]; }
        const size_t o = ((size_t)hh + (size_t)t * nF) * 64;
        for (int i = 0; i < 64; i++) { adHost[o + i] = AH[i]; adTarget[o + i] = AT[i]; adHostF[o + i] = (float)AH[i]; adTargetF[o + i] = (float)AT[i]; }
      }
    for (int i = 0; i < 4; i++) { cPrior[i] = S.initialCalibHessian; cPriorF[i] = (float)cPrior[i]; }
  }
  void setDeltaF() {
    adHTdeltaF.assign((size_t)nF * nF * 8, 0);
    for (int hh = 0; hh < nF; hh++)
      for (int t = 0; t < nF; t++) {
        const size_t idx = (size_t)hh + (size_t)t * nF;
        float dh[8], dt[8];
        for (int i = 0; i < 8; i++) { dh[i] = (float)(frames[hh].state[i] - frames[hh].state_zero[i]); dt[i] = (float)(frames[t].state[i] - frames[t].state_zero[i]); }
        for (int c = 0; c < 8; c++) {
          float s1 = 0, s2 = 0;
          for (int r = 0; r < 8; r++) { s1 += dh[r] * adHostF[idx * 64 + r * 8 + c]; s2 += dt[r] * adTargetF[idx * 64 + r * 8 + c]; }
          adHTdeltaF[idx * 8 + c] = s1 + s2;
        }
      }
    for (int i = 0; i < 4; i++) cDeltaF[i] = (float)c_vmz[i];
    for (auto& f : frames) for (int i = 0; i < 8; i++) { f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
    for (auto& p : points) p.deltaF = p.idepth - p.idepth_zero;
  }

  // ---- projections (ResidualProjections.h)
  bool projectPointFull(float u_pt, float v_pt, float idepth, const float* R, const float* t, float& drescale, float& u, float& v,
                        float& Ku, float& Kv, float KliP[3], float& new_idepth) const {
    KliP[0] = (u_pt + 0 - c_f[2]) * c_i[0]; KliP[1] = (v_pt + 0 - c_f[3]) * c_i[1]; KliP[2] = 1;
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = R[r * 3 + 0] * KliP[0] + R[r * 3 + 1] * KliP[1] + R[r * 3 + 2] * KliP[2] + t[r] * idepth;
    drescale = 1.0f / ptp[2];
    new_idepth = idepth * drescale;
    if (!(drescale > 0)) return false;
    u = ptp[0] * drescale; v = ptp[1] * drescale;
    Ku = u * c_f[0] + c_f[2]; Kv = v * c_f[1] + c_f[3];
    return Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
  }
  bool projectPointK(float u_pt, float v_pt, float idepth, const float* KRKi, const float* Kt, float& Ku, float& Kv) const {
    float ptp[3];
    for (int r = 0; r < 3; r++) ptp[r] = KRKi[r * 3 + 0] * u_pt + KRKi[r * 3 + 1] * v_pt + KRKi[r * 3 + 2] * 1.0f + Kt[r] * idepth;
    Ku = ptp[0] / ptp[2]; Kv = ptp[1] / ptp[2];
    return Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
  }

  double linearizeRes(ORes& r) {
    r.state_NewEnergyWithOutlier = -1;
    if (r.state_state == RS_OOB) { r.state_NewState = RS_OOB; return r.state_energy; }
    const OPoint& p = points[r.point];
    const OFrame& host = frames[r.host]; const OFrame& target = frames[r.target];
    const Precalc& pre = host.pre[r.target];
    float energyLeft = 0;
    const V3f* dIl = target.dI;
    const float affLL0 = pre.PRE_aff_mode[0], affLL1 = pre.PRE_aff_mode[1], b0 = pre.PRE_b0_mode;
    RawJ& J = r.Jnew;
    float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x, d_d_y;
    {
      float drescale, u, v, new_idepth, Ku, Kv, KliP[3];
      if (!projectPointFull(p.u, p.v, p.idepth_zero_scaled, pre.PRE_RTll_0, pre.PRE_tTll_0, drescale, u, v, Ku, Kv, KliP, new_idepth)) {
        r.state_NewState = RS_OOB; return r.state_energy;
      }
      r.centerProjectedTo[0] = Ku; r.centerProjectedTo[1] = Kv; r.centerProjectedTo[2] = new_idepth;
      const float* R0 = pre.PRE_RTll_0; const float* t0 = pre.PRE_tTll_0;
      const float fxl = c_f[0], fyl = c_f[1], fxli = c_i[0], fyli = c_i[1];
      d_d_x = drescale * (t0[0] - t0[2] * u) * SCALE_IDEPTH * fxl;
      d_d_y = drescale * (t0[1] - t0[2] * v) * SCALE_IDEPTH * fyl;
      d_C_x[2] = drescale * (R0[6] * u - R0[0]);
      d_C_x[3] = fxl * drescale * (R0[7] * u - R0[1]) * fyli;
      d_C_x[0] = KliP[0] * d_C_x[2];
      d_C_x[1] = KliP[1] * d_C_x[3];
      d_C_y[2] = fyl * drescale * (R0[6] * v - R0[3]) * fxli;
      d_C_y[3] = drescale * (R0[7] * v - R0[4]);
      d_C_y[0] = KliP[0] * d_C_y[2];
      d_C_y[1] = KliP[1] * d_C_y[3];
      d_C_x[0] = (d_C_x[0] + u) * SCALE_F; d_C_x[1] *= SCALE_F; d_C_x[2] = (d_C_x[2] + 1) * SCALE_C; d_C_x[3] *= SCALE_C;
      d_C_y[0] *= SCALE_F; d_C_y[1] = (d_C_y[1] + v) * SCALE_F; d_C_y[2] *= SCALE_C; d_C_y[3] = (d_C_y[3] + 1) * SCALE_C;
      d_xi_x[0] = new_idepth * fxl; d_xi_x[1] = 0; d_xi_x[2] = -new_idepth * u * fxl;
      d_xi_x[3] = -u * v * fxl; d_xi_x[4] = (1 + u * u) * fxl; d_xi_x[5] = -v * fxl;
      d_xi_y[0] = 0; d_xi_y[1] = new_idepth * fyl; d_xi_y[2] = -new_idepth * v * fyl;
      d_xi_y[3] = -(1 + v * v) * fyl; d_xi_y[4] = u * v * fyl; d_xi_y[5] = u * fyl;
    }
    for (int i = 0; i < 6; i++) { J.Jpdxi[0][i] = d_xi_x[i]; J.Jpdxi[1][i] = d_xi_y[i]; }
    for (int i = 0; i < 4; i++) { J.Jpdc[0][i] = d_C_x[i]; J.Jpdc[1][i] = d_C_y[i]; }
    J.Jpdd[0] = d_d_x; J.Jpdd[1] = d_d_y;

    float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
    float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
    float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
    float wJI2_sum = 0;
    for (int idx = 0; idx < PATTERN; idx++) {
      float Ku, Kv;
      if (!projectPointK(p.u + patternP[idx][0], p.v + patternP[idx][1], p.idepth_scaled, pre.PRE_KRKiTll, pre.PRE_KtTll, Ku, Kv)) {
        r.state_NewState = RS_OOB; return r.state_energy;
      }
      r.projectedTo[idx][0] = Ku; r.projectedTo[idx][1] = Kv;
      V3f hitColor = interp33f(dIl, Ku, Kv, w);
      float residual = hitColor.v[0] - (float)(affLL0 * p.color[idx] + affLL1);
      float drdA = (p.color[idx] - b0);
      if (!std::isfinite((float)hitColor.v[0])) { r.state_NewState = RS_OOB; return r.state_energy; }
      float wgt = sqrtf(S.outlierTHSumComponent / (S.outlierTHSumComponent + (hitColor.v[1] * hitColor.v[1] + hitColor.v[2] * hitColor.v[2])));
      wgt = 0.5f * (wgt + p.weights[idx]);
      float hw = fabsf(residual) < S.huberTH ? 1 : S.huberTH / fabsf(residual);
      energyLeft += wgt * wgt * hw * residual * residual * (2 - hw);
      {
        if (hw < 1) hw = sqrtf(hw);
        hw = hw * wgt;
        hitColor.v[1] *= hw; hitColor.v[2] *= hw;
        J.resF[idx] = residual * hw;
        J.JIdx[0][idx] = hitColor.v[1]; J.JIdx[1][idx] = hitColor.v[2];
        J.JabF[0][idx] = drdA * hw; J.JabF[1][idx] = hw;
        JIdxJIdx_00 += hitColor.v[1] * hitColor.v[1];
        JIdxJIdx_11 += hitColor.v[2] * hitColor.v[2];
        JIdxJIdx_10 += hitColor.v[1] * hitColor.v[2];
        JabJIdx_00 += drdA * hw * hitColor.v[1];
        JabJIdx_01 += drdA * hw * hitColor.v[2];
        JabJIdx_10 += hw * hitColor.v[1];
        JabJIdx_11 += hw * hitColor.v[2];
        JabJab_00 += drdA * drdA * hw * hw;
        JabJab_01 += drdA * hw * hw;
        JabJab_11 += hw * hw;
        wJI2_sum += hw * hw * (hitColor.v[1] * hitColor.v[1] + hitColor.v[2] * hitColor.v[2]);
        if (S.affineOptModeA < 0) J.JabF[0][idx] = 0;
        if (S.affineOptModeB < 0) J.JabF[1][idx] = 0;
      }
    }
    J.JIdx2[0][0] = JIdxJIdx_00; J.JIdx2[0][1] = JIdxJIdx_10; J.JIdx2[1][0] = JIdxJIdx_10; J.JIdx2[1][1] = JIdxJIdx_11;
    J.JabJIdx[0][0] = JabJIdx_00; J.JabJIdx[0][1] = JabJIdx_01; J.JabJIdx[1][0] = JabJIdx_10; J.JabJIdx[1][1] = JabJIdx_11;
    J.Jab2[0][0] = JabJab_00; J.Jab2[0][1] = JabJab_01; J.Jab2[1][0] = JabJab_01; J.Jab2[1][1] = JabJab_11;
    r.state_NewEnergyWithOutlier = energyLeft;
    const float th = std::max<float>(host.frameEnergyTH, target.frameEnergyTH);
    if (energyLeft > th || wJI2_sum < 2) { energyLeft = th; r.state_NewState = RS_OUTLIER; }
    else r.state_NewState = RS_IN;
    r.state_NewEnergy = energyLeft;
    return energyLeft;
  }

  void takeDataF(ORes& r) {
    std::swap(r.Jef, r.Jnew);
    const RawJ& J = r.Jef;
    const float v0 = J.JIdx2[0][0] * J.Jpdd[0] + J.JIdx2[0][1] * J.Jpdd[1];
    const float v1 = J.JIdx2[1][0] * J.Jpdd[0] + J.JIdx2[1][1] * J.Jpdd[1];
    for (int i = 0; i < 6; i++) r.JpJdF[i] = J.Jpdxi[0][i] * v0 + J.Jpdxi[1][i] * v1;
    r.JpJdF[6] = J.JabJIdx[0][0] * J.Jpdd[0] + J.JabJIdx[0][1] * J.Jpdd[1];
    r.JpJdF[7] = J.JabJIdx[1][0] * J.Jpdd[0] + J.JabJIdx[1][1] * J.Jpdd[1];
  }
  void applyRes(ORes& r) {  // applyRes(true)
    if (r.state_state == RS_OOB) return;
    if (r.state_NewState == RS_IN) { r.isActive = true; takeDataF(r); }
    else r.isActive = false;
    r.state_state = r.state_NewState;
    r.state_energy = r.state_NewEnergy;
  }
  void setNewFrameEnergyTH() {
    std::vector<float> allResVec;
    allResVec.reserve(activeResiduals.size() * 2);
    const int newF = nF - 1;
    for (int ri : activeResiduals) { const ORes& r = res[ri]; if (r.state_NewEnergyWithOutlier >= 0 && r.target == newF) allResVec.push_back((float)r.state_NewEnergyWithOutlier); }
    OFrame& nf = frames[newF];
    if (allResVec.size() == 0) { nf.frameEnergyTH = 12 * 12 * PATTERN; return; }
    int nthIdx = (int)(S.frameEnergyTHN * allResVec.size());
    std::nth_element(allResVec.begin(), allResVec.begin() + nthIdx, allResVec.end());
    float nthElement = sqrtf(allResVec[nthIdx]);
    nf.frameEnergyTH = nthElement * S.frameEnergyTHFacMedian;
    nf.frameEnergyTH = 26.0f * S.frameEnergyTHConstWeight + nf.frameEnergyTH * (1 - S.frameEnergyTHConstWeight);
    nf.frameEnergyTH = nf.frameEnergyTH * nf.frameEnergyTH;
    nf.frameEnergyTH *= S.overallEnergyTHWeight * S.overallEnergyTHWeight;
  }
  double linearizeAll(bool fixLinearization) {
    double lastEnergyP = 0;
    if (!fixLinearization && nThreads > 1) {   // linearizeAll_Reductor over the worker pool (FullSystemOptimize.cpp:55-88,150-171): per-worker energy sums
      // per-worker sums on their own cache lines (the reference's per-thread Vec10 stats are 80 B apart, IndexThreadReduce.h:53-57)
      std::vector<PaddedDouble> part(nThreads);
      parallelChunks((int)activeResiduals.size(), [&](int tid, int b0, int e0) {
        double e = 0;
        for (int k = b0; k < e0; k++) e += linearizeRes(res[activeResiduals[k]]);
        part[tid].v += e;
      });
      for (const PaddedDouble& v : part) lastEnergyP += v.v;
      setNewFrameEnergyTH();
      return lastEnergyP;
    }
    for (int ri : activeResiduals) {
      ORes& r = res[ri];
      lastEnergyP += linearizeRes(r);
      if (fixLinearization) {
        applyRes(r);
        if (r.isActive) {
          if (r.isNew) {
            OPoint& p = points[r.point];
            const Precalc& pre = frames[r.host].pre[r.target];
            float pi[3], pp[3];
            for (int k = 0; k < 3; k++) pi[k] = pre.PRE_KRKiTll[k * 3 + 0] * p.u + pre.PRE_KRKiTll[k * 3 + 1] * p.v + pre.PRE_KRKiTll[k * 3 + 2] * 1.0f;
            for (int k = 0; k < 3; k++) pp[k] = pi[k] + pre.PRE_KtTll[k] * p.idepth_scaled;
            const float dx = pi[0] / pi[2] - pp[0] / pp[2], dy = pi[1] / pi[2] - pp[1] / pp[2];
            float relBS = 0.01 * sqrtf(dx * dx + dy * dy);
            if (relBS > p.maxRelBaseline) p.maxRelBaseline = relBS;
            p.numGoodResiduals++;
          }
        } else {
          // EnergyFunctional::dropResidual (EnergyFunctional.cpp:500-520) and deleteOut (FullSystemOptimize.cpp:174-193): the last residual of the point
          // takes the removed one's slot, which is the order every later per-point sum runs in
          r.dropped = true;
          std::vector<int>& pr = points[r.point].residuals;
          for (size_t k = 0; k < pr.size(); k++) if (pr[k] == ri) { pr[k] = pr.back(); pr.pop_back(); break; }
        }
      }
    }
    setNewFrameEnergyTH();
    return lastEnergyP;
  }

  // ---- accumulation
  void topAddPoint(int mode, OPoint& p, std::vector<AccApprox>& acc, int& nres) {
    float bd_acc = 0, Hdd_acc = 0, Hcd_acc[4] = {0, 0, 0, 0};
    const float dd = p.deltaF;
    for (int ri : p.residuals) {
      ORes& r = res[ri];
      if (r.dropped) continue;
      if (mode == 0) { if (r.isLinearized || !r.isActive) continue; }
      if (mode == 1) { if (!r.isLinearized || !r.isActive) continue; }
      if (mode == 2) { if (!r.isActive) continue; }   // marginalisation: every active residual, all linearised (asserted in the reference)
      const RawJ& rJ = r.Jef;
      const int htIDX = r.host + r.target * nF;
      const float* dp = &adHTdeltaF[(size_t)htIDX * 8];
      float resApprox[8];
      if (mode == 0) for (int i = 0; i < 8; i++) resApprox[i] = rJ.resF[i];
      if (mode == 2) for (int i = 0; i < 8; i++) resApprox[i] = r.res_toZeroF[i];
      if (mode == 1) {
        float jx = rJ.Jpdd[0] * dd, jy = rJ.Jpdd[1] * dd;
        float sx = 0, sy = 0;
        for (int i = 0; i < 6; i++) { sx += rJ.Jpdxi[0][i] * dp[i]; sy += rJ.Jpdxi[1][i] * dp[i]; }
        float cx = 0, cy = 0;
        for (int i = 0; i < 4; i++) { cx += rJ.Jpdc[0][i] * cDeltaF[i]; cy += rJ.Jpdc[1][i] * cDeltaF[i]; }
        const float Jp_delta_x = sx + cx + jx, Jp_delta_y = sy + cy + jy;
        for (int i = 0; i < 8; i++) {
          float rtz = r.res_toZeroF[i];
          rtz = rtz + rJ.JIdx[0][i] * Jp_delta_x; rtz = rtz + rJ.JIdx[1][i] * Jp_delta_y;
          rtz = rtz + rJ.JabF[0][i] * dp[6]; rtz = rtz + rJ.JabF[1][i] * dp[7];
          resApprox[i] = rtz;
        }
      }
      float JI_r[2] = {0, 0}, Jab_r[2] = {0, 0}, rr = 0;
      for (int i = 0; i < PATTERN; i++) {
        JI_r[0] += resApprox[i] * rJ.JIdx[0][i]; JI_r[1] += resApprox[i] * rJ.JIdx[1][i];
        Jab_r[0] += resApprox[i] * rJ.JabF[0][i]; Jab_r[1] += resApprox[i] * rJ.JabF[1][i];
        rr += resApprox[i] * resApprox[i];
      }
      AccApprox& a = acc[htIDX];
      a.update(rJ.Jpdc[0], rJ.Jpdxi[0], rJ.Jpdc[1], rJ.Jpdxi[1], rJ.JIdx2[0][0], rJ.JIdx2[0][1], rJ.JIdx2[1][1]);
      a.updateBotRight(rJ.Jab2[0][0], rJ.Jab2[0][1], Jab_r[0], rJ.Jab2[1][1], Jab_r[1], rr);
      a.updateTopRight(rJ.Jpdc[0], rJ.Jpdxi[0], rJ.Jpdc[1], rJ.Jpdxi[1], rJ.JabJIdx[0][0], rJ.JabJIdx[0][1], rJ.JabJIdx[1][0], rJ.JabJIdx[1][1], JI_r[0], JI_r[1]);
      const float Ji2_Jpdd0 = rJ.JIdx2[0][0] * rJ.Jpdd[0] + rJ.JIdx2[0][1] * rJ.Jpdd[1];
      const float Ji2_Jpdd1 = rJ.JIdx2[1][0] * rJ.Jpdd[0] + rJ.JIdx2[1][1] * rJ.Jpdd[1];
      bd_acc += JI_r[0] * rJ.Jpdd[0] + JI_r[1] * rJ.Jpdd[1];
      Hdd_acc += Ji2_Jpdd0 * rJ.Jpdd[0] + Ji2_Jpdd1 * rJ.Jpdd[1];
      for (int i = 0; i < 4; i++) Hcd_acc[i] += rJ.Jpdc[0][i] * Ji2_Jpdd0 + rJ.Jpdc[1][i] * Ji2_Jpdd1;
      nres++;
    }
    if (mode == 0) { p.Hdd_accAF = Hdd_acc; p.bd_accAF = bd_acc; for (int i = 0; i < 4; i++) p.Hcd_accAF[i] = Hcd_acc[i]; }
    if (mode == 1 || mode == 2) { p.Hdd_accLF = Hdd_acc; p.bd_accLF = bd_acc; for (int i = 0; i < 4; i++) p.Hcd_accLF[i] = Hcd_acc[i]; }
    if (mode == 2) { p.Hdd_accAF = 0; p.bd_accAF = 0; for (int i = 0; i < 4; i++) p.Hcd_accAF[i] = 0; }
  }

  // EnergyFunctional::marginalizeFrame, visual part (EnergyFunctional.cpp:570-640): move the frame's block to the end, add its
  // prior, diagonal pre-scaling, Schur complement of the 8x8 block, unscale, symmetrise.  Returns the (n-8)-dimensional prior.
  void marginalizeFrame(int idx, Mat& HMn, Mat& bMn) const {
    const int odim = nF * 8 + CPARS, ndim = odim - 8;
    Mat Hm = HM, bm = bM;
    if (Hm.size() != (size_t)odim * odim) { Hm.assign((size_t)odim * odim, 0.0); bm.assign(odim, 0.0); }
    // permutation: everything before the frame stays, the tail moves up, the frame goes last
    std::vector<int> perm;
    const int io = idx * 8 + CPARS;
    for (int i = 0; i < io; i++) perm.push_back(i);
    for (int i = io + 8; i < odim; i++) perm.push_back(i);
    for (int i = io; i < io + 8; i++) perm.push_back(i);
    Mat H((size_t)odim * odim), b(odim);
    for (int i = 0; i < odim; i++) { b[i] = bm[perm[i]]; for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = Hm[(size_t)perm[i] * odim + perm[j]]; }
    const OFrame& f = frames[idx];
    for (int i = 0; i < 8; i++) { H[(size_t)(ndim + i) * odim + ndim + i] += f.prior[i]; b[ndim + i] += f.prior[i] * f.delta_prior[i]; }
    Mat SVec(odim), SVecI(odim);
    for (int i = 0; i < odim; i++) { SVec[i] = std::sqrt(std::fabs(H[(size_t)i * odim + i]) + 10); SVecI[i] = 1.0 / SVec[i]; }
    for (int i = 0; i < odim; i++) { b[i] *= SVecI[i]; for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = SVecI[i] * H[(size_t)i * odim + j] * SVecI[j]; }
    // invert the bottom-right 8x8 (Eigen: partial-pivot LU), symmetrised as the reference does (0.5f*(hpi+hpi) is the identity map)
    double A[8][16];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { A[i][j] = H[(size_t)(ndim + i) * odim + ndim + j]; A[i][8 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 8; c++) {
      int piv = c;
      for (int r = c + 1; r < 8; r++) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
      if (piv != c) for (int k = 0; k < 16; k++) std::swap(A[c][k], A[piv][k]);
      const double d = A[c][c];
      for (int k = 0; k < 16; k++) A[c][k] /= d;
      for (int r = 0; r < 8; r++) if (r != c) { const double m = A[r][c]; if (m != 0) for (int k = 0; k < 16; k++) A[r][k] -= m * A[c][k]; }
    }
    // bli = BL^T * hpi (ndim x 8); top-left -= bli * BL; b_head -= bli * b_tail
    std::vector<double> bli((size_t)ndim * 8);
    for (int i = 0; i < ndim; i++) for (int j = 0; j < 8; j++) { double sum = 0; for (int k = 0; k < 8; k++) sum += H[(size_t)(ndim + k) * odim + i] * A[k][8 + j]; bli[(size_t)i * 8 + j] = sum; }
    for (int i = 0; i < ndim; i++) {
      for (int j = 0; j < ndim; j++) { double sum = 0; for (int k = 0; k < 8; k++) sum += bli[(size_t)i * 8 + k] * H[(size_t)(ndim + k) * odim + j]; H[(size_t)i * odim + j] -= sum; }
      double sb = 0; for (int k = 0; k < 8; k++) sb += bli[(size_t)i * 8 + k] * b[ndim + k];
      b[i] -= sb;
    }
    HMn.assign((size_t)ndim * ndim, 0.0); bMn.assign(ndim, 0.0);
    for (int i = 0; i < ndim; i++) {
      bMn[i] = SVec[i] * b[i];
      for (int j = 0; j < ndim; j++) HMn[(size_t)i * ndim + j] = 0.5 * (SVec[i] * H[(size_t)i * odim + j] * SVec[j] + SVec[j] * H[(size_t)j * odim + i] * SVec[i]);
    }
  }

  // EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:76-106): res_toZeroF = resF - [JI*Jp Ja] * delta
  void fixLinearizationF(ORes& r) {
    const RawJ& J = r.Jef;
    const float* dp = &adHTdeltaF[(size_t)(r.host + nF * r.target) * 8];
    const float dd = points[r.point].deltaF;
    float sx = 0, sy = 0, cx = 0, cy = 0;
    for (int i = 0; i < 6; i++) { sx += J.Jpdxi[0][i] * dp[i]; sy += J.Jpdxi[1][i] * dp[i]; }
    for (int i = 0; i < 4; i++) { cx += J.Jpdc[0][i] * cDeltaF[i]; cy += J.Jpdc[1][i] * cDeltaF[i]; }
    const float Jp_delta_x = sx + cx + J.Jpdd[0] * dd, Jp_delta_y = sy + cy + J.Jpdd[1] * dd;
    for (int i = 0; i < 8; i++) {
      float rtz = J.resF[i];
      rtz = rtz - J.JIdx[0][i] * Jp_delta_x; rtz = rtz - J.JIdx[1][i] * Jp_delta_y;
      rtz = rtz - J.JabF[0][i] * dp[6]; rtz = rtz - J.JabF[1][i] * dp[7];
      r.res_toZeroF[i] = rtz;
    }
    r.isLinearized = true;
  }

  // FullSystem::flagPointsForRemoval's relinearisation branch (FullSystem.cpp:829-859) for the candidate points, then
  // EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:678-742).  decision[i]: 0 untouched, 1 marginalised, 2 dropped.
  // Hadd / badd = setting_margWeightFac * (M - Msc), the increment of HM / bM.
  int marginalizePoints(const unsigned char* cand, unsigned char* decision, Mat& Hadd, Mat& badd) {
    const float setting_minIdepthH_marg = 50, setting_idepthFixPriorMargFac = 600 * 600, setting_margWeightFac = 0.5 * 0.5;
    for (size_t i = 0; i < points.size(); i++) {
      decision[i] = 0;
      if (!cand[i]) continue;
      OPoint& p = points[i];
      for (int ri : p.residuals) {
        ORes& r = res[ri];
        if (r.dropped) continue;
        r.state_NewEnergy = r.state_energy = 0; r.state_NewState = RS_OUTLIER; r.state_state = RS_IN;   // resetOOB
        linearizeRes(r);
        r.isLinearized = false;
        applyRes(r);
        if (r.isActive) fixLinearizationF(r);
      }
      decision[i] = (p.idepth_hessian > setting_minIdepthH_marg) ? 1 : 2;
    }
    std::vector<std::vector<AccApprox>> accs(1, std::vector<AccApprox>((size_t)nF * nF));
    for (auto& x : accs[0]) x.initialize();
    std::vector<SCAcc> As(1);
    As[0].init(nF);
    int nres = 0;
    for (size_t i = 0; i < points.size(); i++) {
      if (decision[i] != 1) continue;
      OPoint& p = points[i];
      p.priorF *= setting_idepthFixPriorMargFac;
      topAddPoint(2, p, accs[0], nres);
      scAddPoint(p, false, As[0]);
    }
    Mat M, Mb, Msc, Mbsc;
    topStitch(accs, M, Mb, false);
    scStitch(As, Msc, Mbsc);
    const int n = nF * 8 + CPARS;
    Hadd.assign((size_t)n * n, 0); badd.assign(n, 0);
    for (size_t k = 0; k < Hadd.size(); k++) Hadd[k] = setting_margWeightFac * (M[k] - Msc[k]);
    for (int k = 0; k < n; k++) badd[k] = setting_margWeightFac * (Mb[k] - Mbsc[k]);
    return nres;
  }
  static inline void blkMulAdd(Mat& H, int n, int r0, int c0, const double* A, const double* B, const double* Ct, int inner = 8) {
    // H[r0.., c0..] (8x8) += A(8x8) * B(8x8) * Ct^T, all row-major 8x8
    double T[64];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { double s = 0; for (int k = 0; k < 8; k++) s += A[i * 8 + k] * B[k * 8 + j]; T[i * 8 + j] = s; }
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { double s = 0; for (int k = 0; k < 8; k++) s += T[i * 8 + k] * Ct[j * 8 + k]; H[(size_t)(r0 + i) * n + c0 + j] += s; }
  }
  void topStitch(std::vector<std::vector<AccApprox>>& accs, Mat& H, Mat& b, bool usePrior) {
    const int n = nF * 8 + CPARS;
    H.assign((size_t)n * n, 0); b.assign(n, 0);
    if (nThreads > 1) {   // stitchDoubleMT (AccumulatedTopHessian.h:91-139): the (host,target) pairs split over the workers, per-worker H / b summed afterwards
      mtH.resize(nThreads); mtB.resize(nThreads);
      runWorkers([&](int tid) {
        mtH[tid].assign((size_t)n * n, 0); mtB[tid].assign(n, 0);
        for (int k = tid; k < nF * nF; k += nThreads) topStitchPair(accs, k, mtH[tid], mtB[tid]);
      });
      for (int t = 0; t < nThreads; t++) { for (size_t q = 0; q < H.size(); q++) H[q] += mtH[t][q]; for (int q = 0; q < n; q++) b[q] += mtB[t][q]; }
    } else {
      for (int k = 0; k < nF * nF; k++) topStitchPair(accs, k, H, b);
    }
    topStitchTail(H, b, usePrior);
  }
  void topStitchPair(std::vector<std::vector<AccApprox>>& accs, const int k, Mat& H, Mat& b) {
    const int n = nF * 8 + CPARS;
    {
      const int hh = k % nF, t = k / nF;
      const int hIdx = CPARS + hh * 8, tIdx = CPARS + t * 8;
      double accH[13][13];
      memset(accH, 0, sizeof(accH));
      for (auto& acc : accs) {
        acc[k].finish();
        if (acc[k].num == 0) continue;
        for (int i = 0; i < 13; i++) for (int j = 0; j < 13; j++) accH[i][j] += (double)acc[k].H[i][j];
      }
      double B88[64], B8C[32], b8[8];
      for (int i = 0; i < 8; i++) { for (int j = 0; j < 8; j++) B88[i * 8 + j] = accH[CPARS + i][CPARS + j]; for (int j = 0; j < 4; j++) B8C[i * 4 + j] = accH[CPARS + i][j]; b8[i] = accH[CPARS + i][CPARS + 8]; }
      const double* AH = &adHost[(size_t)k * 64]; const double* AT = &adTarget[(size_t)k * 64];
      blkMulAdd(H, n, hIdx, hIdx, AH, B88, AH);
      blkMulAdd(H, n, tIdx, tIdx, AT, B88, AT);
      blkMulAdd(H, n, hIdx, tIdx, AH, B88, AT);
      for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) {
        double s1 = 0, s2 = 0;
        for (int q = 0; q < 8; q++) { s1 += AH[i * 8 + q] * B8C[q * 4 + j]; s2 += AT[i * 8 + q] * B8C[q * 4 + j]; }
        H[(size_t)(hIdx + i) * n + j] += s1; H[(size_t)(tIdx + i) * n + j] += s2;
      }
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) H[(size_t)i * n + j] += accH[i][j];
      for (int i = 0; i < 8; i++) {
        double s1 = 0, s2 = 0;
        for (int q = 0; q < 8; q++) { s1 += AH[i * 8 + q] * b8[q]; s2 += AT[i * 8 + q] * b8[q]; }
        b[hIdx + i] += s1; b[tIdx + i] += s2;
      }
      for (int i = 0; i < 4; i++) b[i] += accH[i][CPARS + 8];
    }
  }
  void topStitchTail(Mat& H, Mat& b, bool usePrior) {
    const int n = nF * 8 + CPARS;
    if (usePrior) {
      for (int i = 0; i < 4; i++) { H[(size_t)i * n + i] += cPrior[i]; b[i] += cPrior[i] * (double)cDeltaF[i]; }
      for (int hh = 0; hh < nF; hh++)
        for (int i = 0; i < 8; i++) { const int q = CPARS + hh * 8 + i; H[(size_t)q * n + q] += frames[hh].prior[i]; b[q] += frames[hh].prior[i] * frames[hh].delta_prior[i]; }
    }
    // make diagonal by copying over parts (stitchDoubleMT tail)
    for (int hh = 0; hh < nF; hh++) {
      const int hIdx = CPARS + hh * 8;
      for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) H[(size_t)i * n + hIdx + j] = H[(size_t)(hIdx + j) * n + i];
      for (int t = hh + 1; t < nF; t++) {
        const int tIdx = CPARS + t * 8;
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H[(size_t)(hIdx + i) * n + tIdx + j] += H[(size_t)(tIdx + j) * n + hIdx + i];
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H[(size_t)(tIdx + i) * n + hIdx + j] = H[(size_t)(hIdx + j) * n + tIdx + i];
      }
    }
  }

  struct SCAcc {
    std::vector<AccXX<8, 4>> accE;
    std::vector<AccX<8>> accEB;
    std::vector<AccXX<8, 8>> accD;
    AccXX<4, 4> accHcc;
    AccX<4> accbc;
    void init(int nf) {
      accE.resize((size_t)nf * nf); accEB.resize((size_t)nf * nf); accD.resize((size_t)nf * nf * nf);
      for (auto& a : accE) a.initialize(); for (auto& a : accEB) a.initialize(); for (auto& a : accD) a.initialize();
      accHcc.initialize(); accbc.initialize();
    }
  };
  void scAddPoint(OPoint& p, bool shiftPriorToZero, SCAcc& A) {
    int ngoodres = 0;
    for (int ri : p.residuals) if (!res[ri].dropped && res[ri].isActive) ngoodres++;
    if (ngoodres == 0) { p.HdiF = 0; p.bdSumF = 0; p.idepth_hessian = 0; p.maxRelBaseline = 0; return; }
    float H = p.Hdd_accAF + p.Hdd_accLF + p.priorF;
    if (H < 1e-10) H = 1e-10;
    p.idepth_hessian = H;
    p.HdiF = 1.0 / H;
    p.bdSumF = p.bd_accAF + p.bd_accLF;
    if (shiftPriorToZero) p.bdSumF += p.priorF * p.deltaF;
    float Hcd[4];
    for (int i = 0; i < 4; i++) Hcd[i] = p.Hcd_accAF[i] + p.Hcd_accLF[i];
    A.accHcc.update(Hcd, Hcd, p.HdiF);
    A.accbc.update(Hcd, p.bdSumF * p.HdiF);
    const int nFrames2 = nF * nF;
    for (int r1i : p.residuals) {
      const ORes& r1 = res[r1i];
      if (r1.dropped || !r1.isActive) continue;
      const int r1ht = r1.host + r1.target * nF;
      for (int r2i : p.residuals) {
        const ORes& r2 = res[r2i];
        if (r2.dropped || !r2.isActive) continue;
        A.accD[r1ht + r2.target * nFrames2].update(r1.JpJdF, r2.JpJdF, p.HdiF);
      }
      A.accE[r1ht].update(r1.JpJdF, Hcd, p.HdiF);
      A.accEB[r1ht].update(r1.JpJdF, p.HdiF * p.bdSumF);
    }
  }
  void scStitch(std::vector<SCAcc>& As, Mat& H, Mat& b) {
    const int n = nF * 8 + CPARS, nf = nF;
    H.assign((size_t)n * n, 0); b.assign(n, 0);
    if (nThreads > 1) {   // AccumulatedSCHessianSSE::stitchDoubleMT (AccumulatedSCHessian.h:93-133)
      mtH.resize(nThreads); mtB.resize(nThreads);
      runWorkers([&](int tid) {
        mtH[tid].assign((size_t)n * n, 0); mtB[tid].assign(n, 0);
        for (int k = tid; k < nf * nf; k += nThreads) scStitchPair(As, k, mtH[tid], mtB[tid]);
      });
      for (int t = 0; t < nThreads; t++) { for (size_t q = 0; q < H.size(); q++) H[q] += mtH[t][q]; for (int q = 0; q < n; q++) b[q] += mtB[t][q]; }
    } else {
      for (int k = 0; k < nf * nf; k++) scStitchPair(As, k, H, b);
    }
    scStitchTail(As, H, b);
  }
  void scStitchPair(std::vector<SCAcc>& As, const int k, Mat& H, Mat& b) {
    const int n = nF * 8 + CPARS, nf = nF, nframes2 = nF * nF;
    {
      const int i = k % nf, j = k / nf;
      const int iIdx = CPARS + i * 8, jIdx = CPARS + j * 8, ijIdx = i + nf * j;
      double Hpc[32], bp[8];
      for (int q = 0; q < 32; q++) Hpc[q] = 0; for (int q = 0; q < 8; q++) bp[q] = 0;
      for (auto& A : As) {
        A.accE[ijIdx].finish(); A.accEB[ijIdx].finish();
        for (int r = 0; r < 8; r++) { for (int c = 0; c < 4; c++) Hpc[r * 4 + c] += (double)A.accE[ijIdx].A1m[r][c]; bp[r] += (double)A.accEB[ijIdx].A1m[r]; }
      }
      const double* AH = &adHost[(size_t)ijIdx * 64]; const double* AT = &adTarget[(size_t)ijIdx * 64];
      for (int r = 0; r < 8; r++) {
        for (int c = 0; c < 4; c++) {
          double s1 = 0, s2 = 0;
          for (int q = 0; q < 8; q++) { s1 += AH[r * 8 + q] * Hpc[q * 4 + c]; s2 += AT[r * 8 + q] * Hpc[q * 4 + c]; }
          H[(size_t)(iIdx + r) * n + c] += s1; H[(size_t)(jIdx + r) * n + c] += s2;
        }
        double s1 = 0, s2 = 0;
        for (int q = 0; q < 8; q++) { s1 += AH[r * 8 + q] * bp[q]; s2 += AT[r * 8 + q] * bp[q]; }
        b[iIdx + r] += s1; b[jIdx + r] += s2;
      }
      for (int kk = 0; kk < nf; kk++) {
        const int kIdx = CPARS + kk * 8, ijkIdx = ijIdx + kk * nframes2, ikIdx = i + nf * kk;
        double accDM[64];
        for (int q = 0; q < 64; q++) accDM[q] = 0;
        for (auto& A : As) {
          A.accD[ijkIdx].finish();
          if (A.accD[ijkIdx].num == 0) continue;
          for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) accDM[r * 8 + c] += (double)A.accD[ijkIdx].A1m[r][c];
        }
        const double* AHk = &adHost[(size_t)ikIdx * 64]; const double* ATk = &adTarget[(size_t)ikIdx * 64];
        blkMulAdd(H, n, iIdx, iIdx, AH, accDM, AHk);
        blkMulAdd(H, n, jIdx, kIdx, AT, accDM, ATk);
        blkMulAdd(H, n, jIdx, iIdx, AT, accDM, AHk);
        blkMulAdd(H, n, iIdx, kIdx, AH, accDM, ATk);
      }
    }
  }
  void scStitchTail(std::vector<SCAcc>& As, Mat& H, Mat& b) {
    const int n = nF * 8 + CPARS, nf = nF;
    for (auto& A : As) {
      A.accHcc.finish(); A.accbc.finish();
      for (int r = 0; r < 4; r++) { for (int c = 0; c < 4; c++) H[(size_t)r * n + c] += (double)A.accHcc.A1m[r][c]; b[r] += (double)A.accbc.A1m[r]; }
    }
    for (int hh = 0; hh < nf; hh++) {
      const int hIdx = CPARS + hh * 8;
      for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) H[(size_t)i * n + hIdx + j] = H[(size_t)(hIdx + j) * n + i];
    }
  }

  // run fn(tid, begin, end) over [0, n) in static chunks of 50 round-robin on nThreads workers (IndexThreadReduce stand-in)
  template <class F>
  void parallelChunks(int n, F fn) {
    if (nThreads <= 1) { fn(0, 0, n); return; }
    runWorkers([&](int t) { for (int b0 = t * 50; b0 < n; b0 += nThreads * 50) fn(t, b0, std::min(n, b0 + 50)); });
  }
  template <class F>
  void runWorkers(F fn) {
    if (!pool || (int)pool->th.size() != nThreads) pool.reset(new WorkerPool(nThreads));
    pool->run(fn);
  }
  // per-worker state of the multi-threaded variant, kept between calls like the reference's acc[tid] members
  std::vector<Mat> mtH, mtB;
  std::vector<std::vector<AccApprox>> mtTop;
  std::vector<SCAcc> mtSC;

  void accumulateAF(Mat& H, Mat& b) {
    if ((int)mtTop.size() != nThreads || mtTop[0].size() != (size_t)nF * nF) mtTop.assign(nThreads, std::vector<AccApprox>((size_t)nF * nF));
    std::vector<std::vector<AccApprox>>& accs = mtTop;
    std::vector<PaddedDouble> nres(nThreads);
    if (nThreads > 1) runWorkers([&](int tid) { for (auto& x : accs[tid]) x.initialize(); });   // setZero runs on the workers too (EnergyFunctional.cpp:252)
    else for (auto& x : accs[0]) x.initialize();
    parallelChunks((int)points.size(), [&](int tid, int b0, int e0) {
      int cnt = 0;
      for (int i = b0; i < e0; i++) topAddPoint(0, points[i], accs[tid], cnt);
      nres[tid].v += cnt;
    });
    topStitch(accs, H, b, false);
    resInA = 0; for (const PaddedDouble& v : nres) resInA += (int)v.v;
  }
  void accumulateLF(Mat& H, Mat& b, bool usePrior = true) {
    std::vector<std::vector<AccApprox>> accs(1, std::vector<AccApprox>((size_t)nF * nF));
    for (auto& x : accs[0]) x.initialize();
    int nres = 0;
    for (auto& p : points) topAddPoint(1, p, accs[0], nres);
    topStitch(accs, H, b, usePrior);
    resInL = nres;
  }
  void accumulateSCF(Mat& H, Mat& b) {
    if ((int)mtSC.size() != nThreads) mtSC.assign(nThreads, SCAcc());
    std::vector<SCAcc>& As = mtSC;
    if (nThreads > 1) runWorkers([&](int tid) { As[tid].init(nF); });
    else As[0].init(nF);
    parallelChunks((int)points.size(), [&](int tid, int b0, int e0) { for (int i = b0; i < e0; i++) scAddPoint(points[i], true, As[tid]); });
    scStitch(As, H, b);
  }

  Mat getStitchedDeltaF() const {
    Mat d(CPARS + nF * 8);
    for (int i = 0; i < 4; i++) d[i] = (double)cDeltaF[i];
    for (int hh = 0; hh < nF; hh++) for (int i = 0; i < 8; i++) d[CPARS + 8 * hh + i] = frames[hh].delta[i];
    return d;
  }

  void getNullspaces() {
    const int n = CPARS + nF * 8;
    ns_pose.clear(); ns_scale.clear();
    for (int i = 0; i < 6; i++) {
      Mat v(n, 0.0);
      for (int f = 0; f < nF; f++) for (int r = 0; r < 6; r++) v[CPARS + f * 8 + r] = frames[f].ns_pose[r][i];  // SCALE_XI_*_INVERSE == 1
      ns_pose.push_back(v);
    }
    Mat v(n, 0.0);
    for (int f = 0; f < nF; f++) for (int r = 0; r < 6; r++) v[CPARS + f * 8 + r] = frames[f].ns_scale[r];
    ns_scale.push_back(v);
  }
  // x -= N (N^T N)^+ N^T x  through a one-sided Jacobi SVD of N = [normalised nullspace vectors]
  void orthogonalize(Mat& x) {
    std::vector<Mat> ns(ns_pose); ns.insert(ns.end(), ns_scale.begin(), ns_scale.end());
    const int n = (int)x.size(), m = (int)ns.size();
    std::vector<Mat> U(m);
    for (int i = 0; i < m; i++) { double nn = 0; for (double v : ns[i]) nn += v * v; nn = std::sqrt(nn); U[i] = ns[i]; for (auto& v : U[i]) v /= nn; }
#ifdef ORC_ALT_EIGEN_LEAF
    // liboracle_altleaf.so only (tests/test_leaf_sensitivity_cpu.py): the same projector x -= N (N^T N)^+ N^T x built WITHOUT an SVD — modified Gram-Schmidt over the
    // normalised nullspace vectors (a vector whose remainder falls below solverModeDelta is dropped) — to measure how far the SVD's internals can move a result
    {
      std::vector<Mat> Q;
      for (int i = 0; i < m; i++) {
        Mat v = U[i];
        for (int pass = 0; pass < 2; pass++)
          for (const Mat& q : Q) { double d = 0; for (int k = n - 1; k >= 0; k--) d += q[k] * v[k]; for (int k = 0; k < n; k++) v[k] -= d * q[k]; }
        double nn = 0; for (int k = n - 1; k >= 0; k--) nn += v[k] * v[k]; nn = std::sqrt(nn);
        if (!(nn > S.solverModeDelta)) continue;
        for (auto& e : v) e /= nn;
        Q.push_back(v);
      }
      Mat proj(n, 0.0);
      for (const Mat& q : Q) { double d = 0; for (int k = n - 1; k >= 0; k--) d += q[k] * x[k]; for (int k = 0; k < n; k++) proj[k] += d * q[k]; }
      for (int k = 0; k < n; k++) x[k] -= proj[k];
      return;
    }
#endif
    // Hestenes: orthogonalise the columns by plane rotations; column norms converge to the singular values
    for (int sweep = 0; sweep < 60; sweep++) {
      double off = 0;
      for (int p = 0; p < m; p++)
        for (int q = p + 1; q < m; q++) {
          double a = 0, bb = 0, g = 0;
          for (int k = 0; k < n; k++) { a += U[p][k] * U[p][k]; bb += U[q][k] * U[q][k]; g += U[p][k] * U[q][k]; }
          off = std::max(off, std::fabs(g) / std::sqrt(a * bb + 1e-300));
          if (std::fabs(g) < 1e-300) continue;
          const double zeta = (bb - a) / (2 * g);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
          const double c = 1 / std::sqrt(1 + t * t), s = c * t;
          for (int k = 0; k < n; k++) { const double up = U[p][k], uq = U[q][k]; U[p][k] = c * up - s * uq; U[q][k] = s * up + c * uq; }
        }
      if (off < 1e-15) break;
    }
    std::vector<double> sv(m);
    double maxSv = 0;
    for (int i = 0; i < m; i++) { double nn = 0; for (double v : U[i]) nn += v * v; sv[i] = std::sqrt(nn); maxSv = std::max(maxSv, sv[i]); }
    Mat proj(n, 0.0);
    for (int i = 0; i < m; i++) {
      if (!(sv[i] > S.solverModeDelta * maxSv)) continue;
      double dot = 0;
      for (int k = 0; k < n; k++) dot += U[i][k] * x[k];
      dot /= sv[i] * sv[i];
      for (int k = 0; k < n; k++) proj[k] += U[i][k] * dot;
    }
    for (int k = 0; k < n; k++) x[k] -= proj[k];
  }

  void resubstitute(const Mat& x) {
    const int n = CPARS + nF * 8;
    std::vector<float> xF(n);
    for (int i = 0; i < n; i++) xF[i] = (float)x[i];
    for (int i = 0; i < 4; i++) c_step[i] = -x[i];
    std::vector<float> xAd((size_t)nF * nF * 8);
    for (int hh = 0; hh < nF; hh++) {
      for (int i = 0; i < 8; i++) frames[hh].step[i] = -x[CPARS + 8 * hh + i];
      frames[hh].step[8] = frames[hh].step[9] = 0;
      for (int t = 0; t < nF; t++) {
        const size_t o = ((size_t)hh + (size_t)nF * t) * 64;
        for (int c = 0; c < 8; c++) {
          float s1 = 0, s2 = 0;
          for (int r = 0; r < 8; r++) { s1 += xF[CPARS + 8 * hh + r] * adHostF[o + r * 8 + c]; s2 += xF[CPARS + 8 * t + r] * adTargetF[o + r * 8 + c]; }
          xAd[((size_t)nF * hh + t) * 8 + c] = s1 + s2;
        }
      }
    }
    parallelChunks((int)points.size(), [&](int, int b0, int e0) {
      for (int k = b0; k < e0; k++) {
        OPoint& p = points[k];
        int ngoodres = 0;
        for (int ri : p.residuals) if (!res[ri].dropped && res[ri].isActive) ngoodres++;
        if (ngoodres == 0) { p.step = 0; continue; }
        float b = p.bdSumF;
        float dotc = 0;
        for (int i = 0; i < 4; i++) dotc += xF[i] * (p.Hcd_accAF[i] + p.Hcd_accLF[i]);
        b -= dotc;
        for (int ri : p.residuals) {
          const ORes& r = res[ri];
          if (r.dropped || !r.isActive) continue;
          float d = 0;
          for (int i = 0; i < 8; i++) d += xAd[((size_t)r.host * nF + r.target) * 8 + i] * r.JpJdF[i];
          b -= d;
        }
        p.step = -b * p.HdiF;
      }
    });
  }

  void solveSystemF(int iteration, double lambda) {
    const int n = CPARS + nF * 8;
    Mat HL_top, HA_top, H_sc, bL_top, bA_top, b_sc;
    accumulateAF(HA_top, bA_top);
    accumulateLF(HL_top, bL_top);
    accumulateSCF(H_sc, b_sc);
    Mat d = getStitchedDeltaF();
    Mat bM_top(n);
    for (int i = 0; i < n; i++) { double s = bM[i]; for (int j = 0; j < n; j++) s += HM[(size_t)i * n + j] * d[j]; bM_top[i] = s; }
    Mat HFinal((size_t)n * n), bFinal(n);
    for (size_t i = 0; i < (size_t)n * n; i++) HFinal[i] = HL_top[i] + HM[i] + HA_top[i];
    for (int i = 0; i < n; i++) bFinal[i] = bL_top[i] + bM_top[i] + bA_top[i] - b_sc[i];
    lastHS.resize((size_t)n * n);
    for (size_t i = 0; i < (size_t)n * n; i++) lastHS[i] = HFinal[i] - H_sc[i];
    lastbS = bFinal;
    for (int i = 0; i < n; i++) HFinal[(size_t)i * n + i] *= (1 + lambda);
    const double fac = 1.0f / (1 + lambda);
    for (size_t i = 0; i < (size_t)n * n; i++) HFinal[i] -= H_sc[i] * fac;
    // Jacobi-scaled LDLT (EnergyFunctional.cpp:971-973)
    Mat SVecI(n), Hs((size_t)n * n), bs(n), xs(n), x(n);
    for (int i = 0; i < n; i++) SVecI[i] = 1.0 / std::sqrt(HFinal[(size_t)i * n + i] + 10);
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) Hs[(size_t)i * n + j] = SVecI[i] * HFinal[(size_t)i * n + j] * SVecI[j]; bs[i] = SVecI[i] * bFinal[i]; }
    ldltSolve(Hs.data(), bs.data(), xs.data(), n);
    for (int i = 0; i < n; i++) x[i] = SVecI[i] * xs[i];
    if (iteration >= 2) orthogonalize(x);  // SOLVER_ORTHOGONALIZE_X_LATER
    lastX = x;
    resubstitute(x);
  }

  double calcLEnergy() {
    if (S.forceAcceptStep) return 0;
    double E = 0;
    for (auto& f : frames) for (int i = 0; i < 8; i++) E += f.delta_prior[i] * f.prior[i] * f.delta_prior[i];
    float ec = 0;
    for (int i = 0; i < 4; i++) ec += cDeltaF[i] * cPriorF[i] * cDeltaF[i];
    E += ec;
    // calcLEnergyPt (EnergyFunctional.cpp:349-409) through IndexThreadReduce::reduce with a step of 50 (EnergyFunctional.cpp:427-428; the reduce has no
    // single-threaded shortcut, IndexThreadReduce.h:78-88): every run of 50 points gets its own Accumulator11 (MatrixAccumulators.h:91-172: four fp32 lanes, shifted up
    // into a 1k and a 1M level when more than 1000 updates sit in the level below) that takes, point by point, the energy (2 res_toZeroF + J delta) . (J delta) of the
    // point's linearised active residuals — two 4-lane updates per residual, no shift-up — and then the point's prior term deltaF^2 priorF as a single update WITH
    // shift-up; the fp32 totals of the runs are added in double (in the reference in the order the workers finish; here in run order).
    double A = 0;
    for (size_t p0 = 0; p0 < points.size(); p0 += 50) {
      float d1[4] = {0, 0, 0, 0}, d1k[4] = {0, 0, 0, 0}, d1m[4] = {0, 0, 0, 0};
      float numIn1 = 0, numIn1k = 0, numIn1m = 0;
      auto shiftUp = [&](bool force) {
        if (numIn1 > 1000 || force) { for (int k = 0; k < 4; k++) { d1k[k] = d1[k] + d1k[k]; d1[k] = 0; } numIn1k += numIn1; numIn1 = 0; }
        if (numIn1k > 1000 || force) { for (int k = 0; k < 4; k++) { d1m[k] = d1k[k] + d1m[k]; d1k[k] = 0; } numIn1m += numIn1k; numIn1k = 0; }
      };
      for (size_t pi = p0; pi < std::min(points.size(), p0 + 50); pi++) {
        const OPoint& p = points[pi];
        const float dd = p.deltaF;
        for (int ri : p.residuals) {
          const ORes& r = res[ri];
          if (r.dropped || !r.isLinearized || !r.isActive) continue;
          const RawJ& rJ = r.Jef;
          const float* dp = &adHTdeltaF[(size_t)(r.host + nF * r.target) * 8];
          float sx = 0, sy = 0, cx = 0, cy = 0;
          for (int i = 0; i < 6; i++) { sx += rJ.Jpdxi[0][i] * dp[i]; sy += rJ.Jpdxi[1][i] * dp[i]; }
          for (int i = 0; i < 4; i++) { cx += rJ.Jpdc[0][i] * cDeltaF[i]; cy += rJ.Jpdc[1][i] * cDeltaF[i]; }
          const float Jp_delta_x_1 = sx + cx + rJ.Jpdd[0] * dd, Jp_delta_y_1 = sy + cy + rJ.Jpdd[1] * dd;
          for (int i = 0; i + 3 < PATTERN; i += 4) {
            for (int k = 0; k < 4; k++) {
              float Jdelta = rJ.JIdx[0][i + k] * Jp_delta_x_1;
              Jdelta = Jdelta + rJ.JIdx[1][i + k] * Jp_delta_y_1;
              Jdelta = Jdelta + rJ.JabF[0][i + k] * dp[6];
              Jdelta = Jdelta + rJ.JabF[1][i + k] * dp[7];
              float r0 = r.res_toZeroF[i + k];
              r0 = r0 + r0;
              r0 = r0 + Jdelta;
              d1[k] = d1[k] + Jdelta * r0;
            }
            numIn1++;   // updateSSENoShift
          }
        }
        d1[0] += p.deltaF * p.deltaF * p.priorF; numIn1++;   // updateSingle
        shiftUp(false);
      }
      shiftUp(true);
      A += (double)(d1m[0] + d1m[1] + d1m[2] + d1m[3]);
    }
    return E + A;
  }
  double calcMEnergy() {
    if (S.forceAcceptStep) return 0;
    const int n = CPARS + nF * 8;
    Mat d = getStitchedDeltaF();
    double s = 0;
    for (int i = 0; i < n; i++) { double t = 2 * bM[i]; for (int j = 0; j < n; j++) t += HM[(size_t)i * n + j] * d[j]; s += d[i] * t; }
    return s;
  }

  void backupState() {
    for (int i = 0; i < 4; i++) c_value_backup[i] = c_value[i];
    for (auto& f : frames) for (int i = 0; i < 10; i++) f.state_backup[i] = f.state[i];
    for (auto& p : points) p.idepth_backup = p.idepth;
  }
  static void setIdepth(OPoint& p, float id) { p.idepth = id; p.idepth_scaled = SCALE_IDEPTH * id; }
  static void setIdepthZero(OPoint& p, float id) { p.idepth_zero = id; p.idepth_zero_scaled = SCALE_IDEPTH * id; }
  bool doStepFromBackup(float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD) {
    double pstepfac[10];
    for (int i = 0; i < 3; i++) pstepfac[i] = stepfacT;
    for (int i = 3; i < 6; i++) pstepfac[i] = stepfacR;
    for (int i = 6; i < 10; i++) pstepfac[i] = stepfacA;
    float sumA = 0, sumB = 0, sumT = 0, sumR = 0, sumID = 0, numID = 0, sumNID = 0;
    double nv[4];
    for (int i = 0; i < 4; i++) nv[i] = c_value_backup[i] + stepfacC * c_step[i];
    calibSetValue(nv);
    for (auto& f : frames) {
      double st[10];
      for (int i = 0; i < 10; i++) st[i] = f.state_backup[i] + pstepfac[i] * f.step[i];
      frameSetState(f, st);
      sumA += f.step[6] * f.step[6]; sumB += f.step[7] * f.step[7];
      sumT += f.step[0] * f.step[0] + f.step[1] * f.step[1] + f.step[2] * f.step[2];
      sumR += f.step[3] * f.step[3] + f.step[4] * f.step[4] + f.step[5] * f.step[5];
    }
    for (auto& p : points) {  // same order as the reference: per frame, per hosted point
      setIdepth(p, p.idepth_backup + stepfacD * p.step);
      sumID += p.step * p.step; sumNID += fabsf(p.idepth_backup); numID++;
      setIdepthZero(p, p.idepth_backup + stepfacD * p.step);
    }
    sumA /= frames.size(); sumB /= frames.size(); sumR /= frames.size(); sumT /= frames.size();
    sumID /= numID; sumNID /= numID;
    setPrecalcValues();
    return sqrtf(sumA) < 0.0005 * S.thOptIterations && sqrtf(sumB) < 0.00005 * S.thOptIterations &&
           sqrtf(sumR) < 0.00005 * S.thOptIterations && sqrtf(sumT) * sumNID < 0.00005 * S.thOptIterations;
  }
  void loadSateBackup() {
    calibSetValue(c_value_backup);
    for (auto& f : frames) frameSetState(f, f.state_backup);
    for (auto& p : points) { setIdepth(p, p.idepth_backup); setIdepthZero(p, p.idepth_backup); }
    setPrecalcValues();
  }

  float optimize(int mnumOptIts) {
    if (nF < 2) return 0;
    if (nF < 3) mnumOptIts = 20;
    if (nF < 4) mnumOptIts = 15;
    activeResiduals.clear();
    for (size_t i = 0; i < res.size(); i++) {
      ORes& r = res[i];
      if (r.dropped) continue;
      if (!r.isLinearized) { activeResiduals.push_back((int)i); r.state_NewEnergy = r.state_energy = 0; r.state_NewState = RS_OUTLIER; r.state_state = RS_IN; }
    }
    double lastEnergy = linearizeAll(false);
    double lastEnergyL = calcLEnergy();
    double lastEnergyM = calcMEnergy();
    for (int ri : activeResiduals) applyRes(res[ri]);
    const double minLambda = 1e-5;
    double lambda = minLambda;
    float stepsize = 1;
    nIterationsDone = 0;
    lastEnergyTrace[0][0] = lastEnergy; lastEnergyTrace[0][1] = lastEnergyL; lastEnergyTrace[0][2] = lastEnergyM; lastEnergyTrace[0][3] = 1;
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
      backupState();
      getNullspaces();
      solveSystemF(iteration, lambda);
      bool canbreak = doStepFromBackup(stepsize, stepsize, stepsize, stepsize, stepsize);
      canbreak = false;  // baIntegration->canBreak() stays false without the GTSAM path (BAGTSAMIntegration.h:225)
      double newEnergy = linearizeAll(false);
      double newEnergyL = calcLEnergy();
      double newEnergyM = calcMEnergy();
      const bool accept = S.forceAcceptStep || (newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM);
      if (accept) {
        for (int ri : activeResiduals) applyRes(res[ri]);
        lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
        lambda *= 0.25;
        lambda = std::max(lambda, minLambda);
      } else {
        loadSateBackup();
        lastEnergy = linearizeAll(false);
        lastEnergyL = calcLEnergy();
        lastEnergyM = calcMEnergy();
        lambda *= 1e2;
      }
      nIterationsDone++;
      if (nIterationsDone < 64) { lastEnergyTrace[nIterationsDone][0] = lastEnergy; lastEnergyTrace[nIterationsDone][1] = lastEnergyL; lastEnergyTrace[nIterationsDone][2] = lastEnergyM; lastEnergyTrace[nIterationsDone][3] = accept ? 1 : 0; }
      if (canbreak && iteration >= S.minOptIterations) break;
    }
    OFrame& last = frames[nF - 1];
    double newStateZero[10] = {0, 0, 0, 0, 0, 0, last.state[6], last.state[7], 0, 0};
    frameSetEvalPT(last, last.PRE_worldToCam, newStateZero);
    setAdjointsF();
    setPrecalcValues();
    lastEnergy = linearizeAll(true);
    finalEnergy = lastEnergy;
    return sqrtf((float)(lastEnergy / (PATTERN * resInA)));
  }
  double finalEnergy = 0;
};

}  // namespace orc

using namespace orc;

extern "C" {

// ---- window construction -----------------------------------------------------------------------------------------------
void* orc_ba_create(int w, int h, const double fxfycxcy[4]) {
  OWindow* W = new OWindow();
  W->w = w; W->h = h; W->wM3G = w - 3; W->hM3G = h - 3;
  W->calibInitScaled(fxfycxcy);
  return W;
}
void orc_ba_destroy(void* p) { delete (OWindow*)p; }
void orc_ba_set_threads(void* p, int n) { ((OWindow*)p)->nThreads = n < 1 ? 1 : n; }
// worldToCam pose7, aff (a,b) in scaled units, exposure, frameID (0 = first keyframe: strong priors), dI = level-0 float3 image
int orc_ba_add_frame(void* p, const double pose7_w2c[7], double aff_a, double aff_b, float exposure, int frameID, const float* dI) {
  OWindow* W = (OWindow*)p;
  OFrame f;
  f.ab_exposure = exposure; f.frameID = frameID; f.dI = (const V3f*)dI;
  SE3 T; T.t[0] = pose7_w2c[0]; T.t[1] = pose7_w2c[1]; T.t[2] = pose7_w2c[2];
  T.q = qimport(Quat{pose7_w2c[6], pose7_w2c[3], pose7_w2c[4], pose7_w2c[5]});
  // setEvalPT_scaled (HessianBlocks.h:220-227)
  double st[10] = {0, 0, 0, 0, 0, 0, SCALE_A_INVERSE * aff_a, SCALE_B_INVERSE * aff_b, 0, 0};
  f.evalPT = T;
  for (int i = 0; i < 10; i++) { f.step[i] = 0; f.state_backup[i] = 0; }
  OWindow::frameSetState(f, st);
  OWindow::frameSetStateZero(f, f.state);
  W->frames.push_back(f);
  W->nF = (int)W->frames.size();
  return W->nF - 1;
}
// perturb the current state of a frame by a left increment (scaled units), keeping evalPT / state_zero (FEJ)
void orc_ba_perturb_frame(void* p, int fidx, const double dstate8[8]) {
  OWindow* W = (OWindow*)p; OFrame& f = W->frames[fidx];
  double st[10]; for (int i = 0; i < 10; i++) st[i] = f.state[i];
  for (int i = 0; i < 6; i++) st[i] += dstate8[i];
  st[6] += dstate8[6] * SCALE_A_INVERSE; st[7] += dstate8[7] * SCALE_B_INVERSE;
  OWindow::frameSetState(f, st);
}
void orc_ba_set_frame_state(void* p, int fidx, const double state10[10]) {
  OWindow* W = (OWindow*)p;
  OWindow::frameSetState(W->frames[fidx], state10);
  W->setPrecalcValues();
}
// replay of windows recorded from a running system: a frame's state_zero (FrameHessian::setStateZero, HessianBlocks.cpp:74-107), its
// frameEnergyTH, and CalibHessian::value_zero (unscaled) when they are not the defaults a fresh window starts with
void orc_ba_set_frame_zero(void* p, int fidx, const double state_zero10[10]) {
  OWindow* W = (OWindow*)p;
  OWindow::frameSetStateZero(W->frames[fidx], state_zero10);
}
void orc_ba_set_frame_energy_th(void* p, const float* th) { OWindow* W = (OWindow*)p; for (int i = 0; i < W->nF; i++) W->frames[i].frameEnergyTH = th[i]; }
void orc_ba_set_calib_values(void* p, const double value[4], const double value_zero[4]) {   // CalibHessian::value / value_zero (unscaled)
  OWindow* W = (OWindow*)p;
  for (int i = 0; i < 4; i++) W->c_value_zero[i] = value_zero[i];
  W->calibSetValue(value);
  W->setPrecalcValues();   // -> setDeltaF: cDeltaF
}
int orc_ba_add_point(void* p, int host, float u, float v, float idepth, const float color[8], const float weights[8], int hasDepthPrior) {
  OWindow* W = (OWindow*)p;
  OPoint q; q.host = host; q.u = u; q.v = v;
  OWindow::setIdepth(q, idepth); OWindow::setIdepthZero(q, idepth);
  memcpy(q.color, color, 32); memcpy(q.weights, weights, 32);
  q.hasDepthPrior = hasDepthPrior != 0;
  q.priorF = q.hasDepthPrior ? W->S.idepthFixPrior * SCALE_IDEPTH * SCALE_IDEPTH : 0;  // EFPoint::takeData
  q.deltaF = 0;
  W->points.push_back(q);
  return (int)W->points.size() - 1;
}
int orc_ba_add_residual(void* p, int point, int target) {
  OWindow* W = (OWindow*)p;
  ORes r; r.point = point; r.host = W->points[point].host; r.target = target;
  memset(&r.Jnew, 0, sizeof(RawJ)); memset(&r.Jef, 0, sizeof(RawJ)); memset(r.JpJdF, 0, sizeof(r.JpJdF)); memset(r.res_toZeroF, 0, sizeof(r.res_toZeroF));
  W->res.push_back(r);
  W->points[point].residuals.push_back((int)W->res.size() - 1);
  return (int)W->res.size() - 1;
}
// finish construction: makeIDX/setAdjointsF/setPrecalcValues + zero marginalisation prior
void orc_ba_finalize(void* p) {
  OWindow* W = (OWindow*)p;
  const int n = CPARS + W->nF * 8;
  W->HM.assign((size_t)n * n, 0); W->bM.assign(n, 0);
  for (auto& f : W->frames) W->frameTakeData(f);
  W->setAdjointsF();
  W->setPrecalcValues();
  for (auto& f : W->frames) W->frameTakeData(f);
}
// EFResidual::fixLinearizationF for the active residuals with mask[ri] != 0, at the current state (setDeltaF first): from then on FullSystem::optimize leaves them out of
// activeResiduals (FullSystemOptimize.cpp:436-446), accumulateLF_MT (addPoint<1>) and calcLEnergyPt carry them.  Returns the number of linearised residuals of the window.
int orc_ba_fix_linearization(void* p, const unsigned char* mask) {
  OWindow* W = (OWindow*)p;
  W->setPrecalcValues();
  int n = 0;
  for (size_t ri = 0; ri < W->res.size(); ri++) {
    ORes& r = W->res[ri];
    if (mask[ri] && !r.dropped && r.isActive && !r.isLinearized) W->fixLinearizationF(r);
    if (!r.dropped && r.isLinearized) n++;
  }
  return n;
}
int orc_ba_marginalize_points(void* p, const unsigned char* cand, unsigned char* decision, double* Hadd, double* badd) {
  OWindow* W = (OWindow*)p;
  Mat H, b;
  const int nres = W->marginalizePoints(cand, decision, H, b);
  memcpy(Hadd, H.data(), sizeof(double) * H.size()); memcpy(badd, b.data(), sizeof(double) * b.size());
  return nres;
}
void orc_ba_marginalize_frame(void* p, int idx, double* HMn, double* bMn) {
  OWindow* W = (OWindow*)p;
  Mat H, b;
  W->marginalizeFrame(idx, H, b);
  memcpy(HMn, H.data(), sizeof(double) * H.size()); memcpy(bMn, b.data(), sizeof(double) * b.size());
}
void orc_ba_set_marg_prior(void* p, const double* HM, const double* bM) {
  OWindow* W = (OWindow*)p; const int n = CPARS + W->nF * 8;
  W->HM.assign(HM, HM + (size_t)n * n); W->bM.assign(bM, bM + n);
}
int orc_ba_nframes(void* p) { return ((OWindow*)p)->nF; }
int orc_ba_npoints(void* p) { return (int)((OWindow*)p)->points.size(); }
int orc_ba_nres(void* p) { return (int)((OWindow*)p)->res.size(); }

// ---- single steps (parity units) ---------------------------------------------------------------------------------------
void orc_ba_activate_all(void* p) {
  OWindow* W = (OWindow*)p;
  W->activeResiduals.clear();
  for (size_t i = 0; i < W->res.size(); i++) { ORes& r = W->res[i]; if (r.dropped) continue; W->activeResiduals.push_back((int)i); r.state_NewEnergy = r.state_energy = 0; r.state_NewState = RS_OUTLIER; r.state_state = RS_IN; }
}
double orc_ba_linearize_all(void* p, int fix) { return ((OWindow*)p)->linearizeAll(fix != 0); }
void orc_ba_apply_res(void* p) { OWindow* W = (OWindow*)p; for (int ri : W->activeResiduals) W->applyRes(W->res[ri]); }
// per residual: state_NewState, NewEnergy, NewEnergyWithOutlier, isActive, centerProjectedTo(3)
void orc_ba_get_res_state(void* p, int* newState, double* newEnergy, double* newEnergyWO, int* isActive, float* center3) {
  OWindow* W = (OWindow*)p;
  for (size_t i = 0; i < W->res.size(); i++) {
    const ORes& r = W->res[i];
    newState[i] = r.state_NewState; newEnergy[i] = r.state_NewEnergy; newEnergyWO[i] = r.state_NewEnergyWithOutlier; isActive[i] = r.isActive ? 1 : 0;
    for (int k = 0; k < 3; k++) center3[3 * i + k] = r.centerProjectedTo[k];
  }
}
// RawResidualJacobian of residual i as 74 floats: resF8, Jpdxi 2x6, Jpdc 2x4, Jpdd 2, JIdx 2x8, JabF 2x8, JIdx2 4, JabJIdx 4, Jab2 4 ; which: 0 = new, 1 = ef (applied)
void orc_ba_get_J(void* p, int i, int which, float* out74, float* JpJdF8) {
  OWindow* W = (OWindow*)p; const ORes& r = W->res[i]; const RawJ& J = which ? r.Jef : r.Jnew;
  memcpy(out74, &J, sizeof(RawJ));
  if (JpJdF8) memcpy(JpJdF8, r.JpJdF, 32);
}
void orc_ba_get_frame_energy_th(void* p, float* out) { OWindow* W = (OWindow*)p; for (int i = 0; i < W->nF; i++) out[i] = W->frames[i].frameEnergyTH; }
void orc_ba_get_precalc(void* p, int host, int target, float* out37) {
  const Precalc& pc = ((OWindow*)p)->frames[host].pre[target];
  float* o = out37;
  memcpy(o, pc.PRE_KRKiTll, 36); o += 9; memcpy(o, pc.PRE_KtTll, 12); o += 3; memcpy(o, pc.PRE_RTll_0, 36); o += 9; memcpy(o, pc.PRE_tTll_0, 12); o += 3;
  o[0] = pc.PRE_aff_mode[0]; o[1] = pc.PRE_aff_mode[1]; o[2] = pc.PRE_b0_mode; o += 3;
  memcpy(o, pc.PRE_RTll, 36); o += 9; o[0] = 0;
}
void orc_ba_get_adjoints(void* p, double* adHost, double* adTarget, float* adHTdeltaF) {
  OWindow* W = (OWindow*)p; const size_t n = (size_t)W->nF * W->nF;
  memcpy(adHost, W->adHost.data(), n * 64 * 8); memcpy(adTarget, W->adTarget.data(), n * 64 * 8);
  if (adHTdeltaF) memcpy(adHTdeltaF, W->adHTdeltaF.data(), n * 8 * 4);
}
// the three accumulated systems of solveSystemF, each (4+8F)^2 / (4+8F)
void orc_ba_accumulate(void* p, double* HA, double* bA, double* HL, double* bL, double* Hsc, double* bsc, int* resInA) {
  OWindow* W = (OWindow*)p;
  Mat a, b, c, d, e, f;
  W->accumulateAF(a, b); W->accumulateLF(c, d); W->accumulateSCF(e, f);
  const size_t n = CPARS + W->nF * 8;
  memcpy(HA, a.data(), n * n * 8); memcpy(bA, b.data(), n * 8); memcpy(HL, c.data(), n * n * 8); memcpy(bL, d.data(), n * 8);
  memcpy(Hsc, e.data(), n * n * 8); memcpy(bsc, f.data(), n * 8);
  if (resInA) *resInA = W->resInA;
}
// the stitched system of the linearised residuals WITHOUT the priors stitchDoubleInternal adds last (AccumulatedTopHessian.cpp:292-302): what the library's host side keeps as
// HLraw / bLraw and adds the priors to in the reference's order (tests/test_host_algebra_cpu.py)
void orc_ba_accumulate_lf_raw(void* p, double* HL, double* bL) {
  OWindow* W = (OWindow*)p;
  Mat c, d;
  W->accumulateLF(c, d, false);
  const size_t n = CPARS + W->nF * 8;
  memcpy(HL, c.data(), n * n * 8); memcpy(bL, d.data(), n * 8);
}
void orc_ba_get_point_acc(void* p, float* Hdd, float* bd, float* Hcd4, float* HdiF, float* bdSumF) {
  OWindow* W = (OWindow*)p;
  for (size_t i = 0; i < W->points.size(); i++) {
    const OPoint& q = W->points[i];
    Hdd[i] = q.Hdd_accAF; bd[i] = q.bd_accAF; for (int k = 0; k < 4; k++) Hcd4[4 * i + k] = q.Hcd_accAF[k]; HdiF[i] = q.HdiF; bdSumF[i] = q.bdSumF;
  }
}
void orc_ba_solve(void* p, int iteration, double lambda, double* x_out) {
  OWindow* W = (OWindow*)p;
  W->getNullspaces();
  W->solveSystemF(iteration, lambda);
  memcpy(x_out, W->lastX.data(), W->lastX.size() * 8);
}
void orc_ba_get_last_system(void* p, double* HS, double* bS) {
  OWindow* W = (OWindow*)p;
  memcpy(HS, W->lastHS.data(), W->lastHS.size() * 8); memcpy(bS, W->lastbS.data(), W->lastbS.size() * 8);
}
void orc_ba_resubstitute(void* p, const double* x) { OWindow* W = (OWindow*)p; Mat xv(x, x + CPARS + W->nF * 8); W->resubstitute(xv); }
void orc_ba_get_point_state(void* p, float* idepth, float* step) {
  OWindow* W = (OWindow*)p;
  for (size_t i = 0; i < W->points.size(); i++) { idepth[i] = W->points[i].idepth; step[i] = W->points[i].step; }
}
void orc_ba_get_frame_pose(void* p, int fidx, double pose7_w2c[7], double aff[2], double state10[10]) {
  OWindow* W = (OWindow*)p; const OFrame& f = W->frames[fidx];
  pose7_w2c[0] = f.PRE_worldToCam.t[0]; pose7_w2c[1] = f.PRE_worldToCam.t[1]; pose7_w2c[2] = f.PRE_worldToCam.t[2];
  pose7_w2c[3] = f.PRE_worldToCam.q.x; pose7_w2c[4] = f.PRE_worldToCam.q.y; pose7_w2c[5] = f.PRE_worldToCam.q.z; pose7_w2c[6] = f.PRE_worldToCam.q.w;
  aff[0] = f.state_scaled[6]; aff[1] = f.state_scaled[7];
  if (state10) memcpy(state10, f.state, 80);
}
void orc_ba_get_calib(void* p, double value_scaled[4]) { memcpy(value_scaled, ((OWindow*)p)->c_value_scaled, 32); }
void orc_ba_get_nullspaces(void* p, double* out /* 7 x (4+8F) */) {
  OWindow* W = (OWindow*)p; W->getNullspaces(); const int n = CPARS + W->nF * 8;
  for (int i = 0; i < 6; i++) memcpy(out + (size_t)i * n, W->ns_pose[i].data(), n * 8);
  memcpy(out + (size_t)6 * n, W->ns_scale[0].data(), n * 8);
}
void orc_ba_orthogonalize(void* p, double* x) { OWindow* W = (OWindow*)p; W->getNullspaces(); Mat xv(x, x + CPARS + W->nF * 8); W->orthogonalize(xv); memcpy(x, xv.data(), xv.size() * 8); }
double orc_ba_calc_lenergy(void* p) { return ((OWindow*)p)->calcLEnergy(); }
double orc_ba_calc_menergy(void* p) { return ((OWindow*)p)->calcMEnergy(); }
// ---- the full FullSystem::optimize ------------------------------------------------------------------------------------
float orc_ba_optimize(void* p, int mnumOptIts, double* finalEnergy, int* iterations, double* trace /* 64x4 or NULL */) {
  OWindow* W = (OWindow*)p;
  float rmse = W->optimize(mnumOptIts);
  if (finalEnergy) *finalEnergy = W->finalEnergy;
  if (iterations) *iterations = W->nIterationsDone;
  if (trace) memcpy(trace, W->lastEnergyTrace, sizeof(W->lastEnergyTrace));
  return rmse;
}
// one GN iteration body (FullSystemOptimize.cpp:485-586) for timing; returns 1 when the step was accepted
int orc_ba_gn_iteration(void* p, int iteration, double* lambda_io, double lastE[3]) {
  OWindow* W = (OWindow*)p;
  W->backupState();
  W->getNullspaces();
  W->solveSystemF(iteration, *lambda_io);
  W->doStepFromBackup(1, 1, 1, 1, 1);
  double nE = W->linearizeAll(false), nL = W->calcLEnergy(), nM = W->calcMEnergy();
  const bool accept = nE + nL + nM < lastE[0] + lastE[1] + lastE[2];
  if (accept) {
    for (int ri : W->activeResiduals) W->applyRes(W->res[ri]);
    lastE[0] = nE; lastE[1] = nL; lastE[2] = nM;
    *lambda_io = std::max(*lambda_io * 0.25, 1e-5);
  } else {
    W->loadSateBackup();
    lastE[0] = W->linearizeAll(false); lastE[1] = W->calcLEnergy(); lastE[2] = W->calcMEnergy();
    *lambda_io *= 1e2;
  }
  return accept ? 1 : 0;
}

// --- primitives exposed for the pinning tests (same record formats as ref_accapprox_stream / ref_accxx_stream in oracle/ref_glue.cpp) ---
void orc_accapprox_stream(int n, const float* rec35, int reps, float* H169, long* num) {
  AccApprox acc; acc.initialize();
  for (int rep = 0; rep < reps; rep++)
  for (int i = 0; i < n; i++) {
    const float* r = rec35 + 35 * i;
    acc.update(r, r + 4, r + 10, r + 14, r[20], r[21], r[22]);
    acc.updateTopRight(r, r + 4, r + 10, r + 14, r[23], r[24], r[25], r[26], r[27], r[28]);
    acc.updateBotRight(r[29], r[30], r[31], r[32], r[33], r[34]);
  }
  acc.finish();
  for (int r = 0; r < 13; r++) for (int c = 0; c < 13; c++) H169[r * 13 + c] = acc.H[r][c];
  *num = (long)acc.num;
}
void orc_accxx_stream(int n, const float* rec21, int reps, float* A88, float* A84, float* A8, long* num) {
  AccXX<8, 8> a88; AccXX<8, 4> a84; AccX<8> a8;
  a88.initialize(); a84.initialize(); a8.initialize();
  for (int rep = 0; rep < reps; rep++)
  for (int i = 0; i < n; i++) { const float* r = rec21 + 21 * i; a88.update(r, r + 8, r[20]); a84.update(r, r + 16, r[20]); a8.update(r, r[20]); }
  a88.finish(); a84.finish(); a8.finish();
  memcpy(A88, a88.A1m, sizeof(a88.A1m)); memcpy(A84, a84.A1m, sizeof(a84.A1m)); memcpy(A8, a8.A1m, sizeof(a8.A1m));
  *num = (long)a88.num;
}
// projectPoint, both forms, on a window that only carries the calibration (w, h, K)
int orc_project_point_short(void* p, float u, float v, float idepth, const float KRKi[9], const float Kt[3], float out2[2]) {
  float Ku = 0, Kv = 0;
  bool ok = ((OWindow*)p)->projectPointK(u, v, idepth, KRKi, Kt, Ku, Kv);
  out2[0] = Ku; out2[1] = Kv; return ok ? 1 : 0;
}
int orc_project_point_long(void* p, float u_pt, float v_pt, float idepth, const float R[9], const float t[3], float out9[9]) {
  float drescale = 0, u = 0, v = 0, Ku = 0, Kv = 0, nid = 0, KliP[3] = {0, 0, 0};
  bool ok = ((OWindow*)p)->projectPointFull(u_pt, v_pt, idepth, R, t, drescale, u, v, Ku, Kv, KliP, nid);
  out9[0] = drescale; out9[1] = u; out9[2] = v; out9[3] = Ku; out9[4] = Kv; out9[5] = KliP[0]; out9[6] = KliP[1]; out9[7] = KliP[2]; out9[8] = nid;
  return ok ? 1 : 0;
}

}  // extern "C"
