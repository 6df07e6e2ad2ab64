// Host side of the BA path of libdmvio_hip: window state the reference keeps in FrameHessian / CalibHessian / EFFrame, the
// tiny dense algebra around the kernels (precalc tables, adjoints, nullspaces, damped Jacobi-scaled LDLT, orthogonalisation,
// step application, accept / reject) and the FullSystem::optimize loop driving the kernels of ba_kernels.hpp.
//
// Reference interfaces mirrored here (all under src/dso/):
//   FrameHessian::setState / setStateZero / setEvalPT / getPrior   FullSystem/HessianBlocks.h:179-299, HessianBlocks.cpp:74-107
//   CalibHessian::setValue                                         FullSystem/HessianBlocks.h:356-372
//   FrameFramePrecalc::set, FullSystem::setPrecalcValues           FullSystem/HessianBlocks.cpp:193-223, FullSystem.cpp:1670-1680
//   EnergyFunctional::setAdjointsF / solveSystemF / orthogonalize / resubstituteF_MT / calcLEnergyF_MT / calcMEnergyF
//                                                                  OptimizationBackend/EnergyFunctional.cpp:48-108,784-996,267-431
//   FullSystem::getNullspaces / linearizeAll / setNewFrameEnergyTH / doStepFromBackup / backupState / loadSateBackup / optimize
//                                                                  FullSystem/FullSystemOptimize.cpp:96-218,224-388,417-647,704-758
// In a drop-in integration the reference's own host code keeps doing this part and calls the kernel-level entry points
// (dmvio_hip_ba_linearize / _accumulate / _resubstitute); dmvio_hip_ba_optimize exists so the whole loop can be verified
// against the oracle and timed end to end.
#pragma once
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <atomic>
#include "common.h"
#include "lie_dev.h"
#include "ba_kernels.hpp"

namespace dmv {

struct BASettingsHost {
  float huberTH = 9, outlierTHSumComponent = 50 * 50, idepthFixPrior = 50 * 50;
  float initialRotPrior = 1e11f, initialTransPrior = 1e10f, initialAffBPrior = 1e14f, initialAffAPrior = 1e14f, initialCalibHessian = 5e9f;
  float affineOptModeA = 1e12f, affineOptModeB = 1e8f;
  float frameEnergyTHConstWeight = 0.5f, frameEnergyTHN = 0.7f, frameEnergyTHFacMedian = 1.5f, overallEnergyTHWeight = 1;
  float thOptIterations = 1.2f;
  int minOptIterations = 1;
  double solverModeDelta = 0.00001;
};

struct BAFrameHost {
  Pose evalPT, w2c, c2w;
  double state[10], state_zero[10], state_scaled[10], step[10], state_backup[10];
  float ab_exposure = 1, frameEnergyTH = 8 * 8 * 8;
  unsigned long long zero_stamp = 0;   // changes whenever frameSetStateZero recomputed this frame's nullspaces
  int frameID = 0, slot = 0;
  double ns_pose[6][6], ns_scale[6];
  double prior[8], delta[8], delta_prior[8];
};

struct BAHost {
  BASettingsHost S;
  int w = 0, h = 0, F = 0, N = 0, R = 0;
  double c_value[4], c_value_zero[4], c_value_scaled[4], c_step[4], c_value_backup[4], c_vmz[4];
  float c_f[4], c_i[4];
  BAFrameHost fr[BA_MAXF_CAP];
  std::vector<double> adHost, adTarget;      // F*F x 64, index h + t*F
  std::vector<float> adHostF, adTargetF;
  std::vector<BAPrecalc> pre;                // F*F, index h + F*t
  float cDeltaF[4], cPriorF[4];
  double cPrior[4];
  std::vector<double> HM, bM, lastX;
  std::vector<double> HLraw, bLraw;   // see lfTop()
  // EnergyFunctional::HMForGTSAM / bMForGTSAM (EnergyFunctional.h:108-111): the marginalisation prior the GTSAM branch of solveSystemF / calcMEnergyF uses instead
  // of HM / bM (only the points marginalised since the last keyframe marginalisation; the rest lives in the GTSAM graph).  Used while `gtsam` is set.
  std::vector<double> HMG, bMG;
  bool gtsam = false;
  const std::vector<double>& priorH() const { return gtsam ? HMG : HM; }
  const std::vector<double>& priorb() const { return gtsam ? bMG : bM; }
  std::vector<std::vector<double>> nsp;      // 7 nullspace vectors
  std::vector<std::vector<double>> orthoBasis;   // unit left singular vectors of the nullspace matrix above the cut (prepareOrthogonalize)
  std::vector<double> bPriorM, hfScratch, htScratch;   // bM + HM * delta (prepareSolve); scratch of solveSystem
  bool solvePrepared = false;
  unsigned long long orthoKey = 0;
  int resInA = 0;

  void calibSetValue(const double v[4]) {
    for (int i = 0; i < 4; i++) c_value[i] = v[i];
    c_value_scaled[0] = 50.0f * v[0]; c_value_scaled[1] = 50.0f * v[1]; c_value_scaled[2] = 50.0f * v[2]; c_value_scaled[3] = 50.0f * v[3];
    for (int i = 0; i < 4; i++) c_f[i] = (float)c_value_scaled[i];
    c_i[0] = 1.0f / c_f[0]; c_i[1] = 1.0f / c_f[1]; c_i[2] = -c_f[2] / c_f[0]; c_i[3] = -c_f[3] / c_f[1];
    for (int i = 0; i < 4; i++) c_vmz[i] = c_value[i] - c_value_zero[i];
  }
  void calibInitScaled(const double vs[4]) {
    const float inv = 1.0f / 50.0f;
    for (int i = 0; i < 4; i++) { c_value_scaled[i] = vs[i]; c_f[i] = (float)vs[i]; c_value[i] = inv * vs[i]; }
    c_i[0] = 1.0f / c_f[0]; c_i[1] = 1.0f / c_f[1]; c_i[2] = -c_f[2] / c_f[0]; c_i[3] = -c_f[3] / c_f[1];
    for (int i = 0; i < 4; i++) { c_value_zero[i] = c_value[i]; c_vmz[i] = 0; c_step[i] = 0; c_value_backup[i] = c_value[i]; }
  }
  static void frameSetState(BAFrameHost& f, const double st[10]) {
    for (int i = 0; i < 10; i++) f.state[i] = st[i];
    for (int i = 0; i < 6; i++) f.state_scaled[i] = 1.0f * st[i];
    f.state_scaled[6] = 10.0f * st[6]; f.state_scaled[7] = 1000.0f * st[7]; f.state_scaled[8] = 10.0f * st[8]; f.state_scaled[9] = 1000.0f * st[9];
    f.w2c = poseMul(poseExp(f.state_scaled), f.evalPT);
    f.c2w = poseInv(f.w2c);
  }
  static void frameSetStateZero(BAFrameHost& f, const double st0[10]) {
    for (int i = 0; i < 10; i++) f.state_zero[i] = st0[i];
    { static std::atomic<unsigned long long> stamp{0}; f.zero_stamp = ++stamp; }   // the nullspaces below change: cached orthogonalisation bases are stale (atomic: handles on several threads)
    const Pose Tinv = poseInv(f.evalPT);
    for (int i = 0; i < 6; i++) {
      double ep[6] = {0, 0, 0, 0, 0, 0}, em[6] = {0, 0, 0, 0, 0, 0};
      ep[i] = 1e-3; em[i] = -1e-3;
      double lp[6], lm[6];
      poseLogHost(poseMul(poseMul(f.evalPT, poseExp(ep)), Tinv), lp);
      poseLogHost(poseMul(poseMul(f.evalPT, poseExp(em)), Tinv), lm);
      for (int r = 0; r < 6; r++) f.ns_pose[r][i] = (lp[r] - lm[r]) / (2e-3);
    }
    Pose P = f.evalPT, M = f.evalPT;
    for (int i = 0; i < 3; i++) { P.t[i] *= 1.00001; M.t[i] /= 1.00001; }
    double lp[6], lm[6];
    poseLogHost(poseMul(P, Tinv), lp); poseLogHost(poseMul(M, Tinv), lm);
    for (int r = 0; r < 6; r++) f.ns_scale[r] = (lp[r] - lm[r]) / (2e-3);
  }
  void frameTakeData(BAFrameHost& f) const {
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (f.frameID == 0) {
      for (int i = 0; i < 3; i++) p[i] = S.initialTransPrior;
      for (int i = 3; i < 6; i++) p[i] = S.initialRotPrior;
      p[6] = S.initialAffAPrior; p[7] = S.initialAffBPrior;
    } else {
      p[6] = S.affineOptModeA < 0 ? S.initialAffAPrior : S.affineOptModeA;
      p[7] = S.affineOptModeB < 0 ? S.initialAffBPrior : S.affineOptModeB;
    }
    for (int i = 0; i < 8; i++) { f.prior[i] = p[i]; f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
  }

  static void m33f(const float* A, const float* B, float* C) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3 + 0] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
  }
  void setPrecalcValues() {
    pre.resize((size_t)F * F);
    const float K[9] = {c_f[0], 0, c_f[2], 0, c_f[1], c_f[3], 0, 0, 1};
    const float a = K[0], e = K[4], c = K[2], ff = K[5];
    const float det = a * (e * 1.0f - ff * 0.0f), invdet = 1.0f / det;
    const float Ki[9] = {(e * 1.0f - ff * 0.0f) * invdet, (c * 0.0f - 0.0f * 1.0f) * invdet, (0.0f * ff - c * e) * invdet,
                         (ff * 0.0f - 0.0f * 1.0f) * invdet, (a * 1.0f - c * 0.0f) * invdet, (c * 0.0f - a * ff) * invdet,
                         (0.0f * 0.0f - e * 0.0f) * invdet, (0.0f * 0.0f - a * 0.0f) * invdet, (a * e - 0.0f * 0.0f) * invdet};
    for (int hh = 0; hh < F; hh++)
      for (int t = 0; t < F; t++) {
        BAPrecalc& pc = pre[(size_t)hh + (size_t)F * t];
        const Pose l0 = poseMul(fr[t].evalPT, poseInv(fr[hh].evalPT));
        double Rd[9];
        quatToR(l0.q, Rd);
        for (int i = 0; i < 9; i++) pc.R0[i] = (float)Rd[i];
        for (int i = 0; i < 3; i++) pc.t0[i] = (float)l0.t[i];
        const Pose l = poseMul(fr[t].w2c, fr[hh].c2w);
        quatToR(l.q, Rd);
        float Rf[9], tf[3], KR[9];
        for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
        for (int i = 0; i < 3; i++) tf[i] = (float)l.t[i];
        m33f(K, Rf, KR);
        m33f(KR, Ki, pc.KRKi);
        for (int r = 0; r < 3; r++) pc.Kt[r] = K[r * 3 + 0] * tf[0] + K[r * 3 + 1] * tf[1] + K[r * 3 + 2] * tf[2];
        double aff[2];
        affFromToHost(fr[hh].ab_exposure, fr[t].ab_exposure, fr[hh].state_scaled[6], fr[hh].state_scaled[7], fr[t].state_scaled[6], fr[t].state_scaled[7], aff);
        pc.aff0 = (float)aff[0]; pc.aff1 = (float)aff[1];
        pc.b0 = (float)(fr[hh].state_zero[7] * 1000.0f);
        pc.pad = 0;
      }
    // setDeltaF: frame deltas (the kernels recompute the per-point deltaF = idepth - idepth_zero themselves)
    for (int i = 0; i < 4; i++) cDeltaF[i] = (float)c_vmz[i];
    for (int f = 0; f < F; f++) for (int i = 0; i < 8; i++) { fr[f].delta[i] = fr[f].state[i] - fr[f].state_zero[i]; fr[f].delta_prior[i] = fr[f].state[i]; }
  }
  // AffLight::fromToVecExposure with the C library's exp: the brightness transfer enters every residual, and the reference
  // evaluates it with std::exp (NumType.h:183).
  static void affFromToHost(float eF, float eT, double aF, double bF, double aT, double bT, double out[2]) {
    if (eF == 0 || eT == 0) { eT = eF = 1; }
    const double a = std::exp(aT - aF) * eT / eF;
    out[0] = a; out[1] = bT - a * bF;
  }
  void setAdjointsF() {
    adHost.assign((size_t)F * F * 64, 0); adTarget.assign((size_t)F * F * 64, 0);
    adHostF.assign((size_t)F * F * 64, 0); adTargetF.assign((size_t)F * F * 64, 0);
    for (int hh = 0; hh < F; hh++)
      for (int t = 0; t < F; t++) {
        const Pose hostToTarget = poseMul(fr[t].evalPT, poseInv(fr[hh].evalPT));
        double Adj[36];
        poseAdj(hostToTarget, Adj);
        double AH[64], AT[64];
        for (int i = 0; i < 64; i++) { AH[i] = 0; AT[i] = 0; }
        for (int i = 0; i < 8; i++) { AH[i * 8 + i] = 1; AT[i * 8 + i] = 1; }
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) AH[r * 8 + c] = -Adj[c * 6 + r];
        double aff[2];
        affFromToHost(fr[hh].ab_exposure, fr[t].ab_exposure, fr[hh].state_zero[6] * 10.0f, fr[hh].state_zero[7] * 1000.0f,
                      fr[t].state_zero[6] * 10.0f, fr[t].state_zero[7] * 1000.0f, aff);
        const float a0 = (float)aff[0];
        AT[6 * 8 + 6] = -a0; AH[6 * 8 + 6] = a0; AT[7 * 8 + 7] = -1; AH[7 * 8 + 7] = a0;
        const float rs[8] = {1, 1, 1, 1, 1, 1, 10.0f, 1000.0f};
        for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) { AH[r * 8 + c] *= rs[r]; AT[r * 8 + c] *= rs[r]; }
        const size_t o = ((size_t)hh + (size_t)t * F) * 64;
        for (int i = 0; i < 64; i++) { adHost[o + i] = AH[i]; adTarget[o + i] = AT[i]; adHostF[o + i] = (float)AH[i]; adTargetF[o + i] = (float)AT[i]; }
      }
    for (int i = 0; i < 4; i++) { cPrior[i] = S.initialCalibHessian; cPriorF[i] = (float)cPrior[i]; }
  }

  int n() const { return 4 + 8 * F; }
  void getNullspaces() {
    nsp.assign(7, std::vector<double>(n(), 0.0));
    for (int i = 0; i < 6; i++) for (int f = 0; f < F; f++) for (int r = 0; r < 6; r++) nsp[i][4 + f * 8 + r] = fr[f].ns_pose[r][i];
    for (int f = 0; f < F; f++) for (int r = 0; r < 6; r++) nsp[6][4 + f * 8 + r] = fr[f].ns_scale[r];
  }
  // x -= N N^+ x : orthonormal basis of span(N) by modified Gram-Schmidt with re-orthogonalisation and the reference's
  // relative singular-value cut (EnergyFunctional.cpp:812-824), evaluated through the Gram matrix's Jacobi eigen-decomposition.
  // The basis depends on the nullspaces only: prepareOrthogonalize() builds it (while the accumulation kernels run), orthogonalize()
  // applies it.
  void prepareOrthogonalize() {
    const int nn = n(), m = 7;
    {   // the basis is a function of the frames' nullspaces only (fixed while a window is optimised): rebuild it when one of them changed
      unsigned long long key = (unsigned long long)F * 0x9E3779B97F4A7C15ull;
      for (int f = 0; f < F; f++) key = key * 1099511628211ull + fr[f].zero_stamp;
      { unsigned long long dbits; static_assert(sizeof(dbits) == sizeof(S.solverModeDelta), "double"); memcpy(&dbits, &S.solverModeDelta, 8); key = key * 1099511628211ull + dbits; }   // the cut enters the basis
      if (key == orthoKey && !orthoBasis.empty()) return;
      orthoKey = key;
    }
    std::vector<std::vector<double>> U(m);
    for (int i = 0; i < m; i++) { double s = 0; for (double v : nsp[i]) s += v * v; s = std::sqrt(s); U[i] = nsp[i]; for (auto& v : U[i]) v /= s; }
    // G = U^T U (7x7), Jacobi eigenvalue iteration: G = V diag(s^2) V^T
    double G[7][7], V[7][7];
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) { double s = 0; for (int k = 0; k < nn; k++) s += U[i][k] * U[j][k]; G[i][j] = s; V[i][j] = i == j ? 1 : 0; }
    for (int sweep = 0; sweep < 60; sweep++) {
      double off = 0;
      for (int p = 0; p < m; p++) for (int q = p + 1; q < m; q++) off += G[p][q] * G[p][q];
      if (off < 1e-30) break;
      for (int p = 0; p < m; p++)
        for (int q = p + 1; q < m; q++) {
          if (std::fabs(G[p][q]) < 1e-300) continue;
          const double theta = (G[q][q] - G[p][p]) / (2 * G[p][q]);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
          const double c = 1 / std::sqrt(t * t + 1), s = t * c;
          for (int k = 0; k < m; k++) { const double gkp = G[k][p], gkq = G[k][q]; G[k][p] = c * gkp - s * gkq; G[k][q] = s * gkp + c * gkq; }
          for (int k = 0; k < m; k++) { const double gpk = G[p][k], gqk = G[q][k]; G[p][k] = c * gpk - s * gqk; G[q][k] = s * gpk + c * gqk; }
          for (int k = 0; k < m; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
        }
    }
    double maxSv = 0;
    double sv[7];
    for (int i = 0; i < m; i++) { sv[i] = std::sqrt(std::max(G[i][i], 0.0)); maxSv = std::max(maxSv, sv[i]); }
    orthoBasis.clear();
    for (int i = 0; i < m; i++) {
      if (!(sv[i] > S.solverModeDelta * maxSv)) continue;
      // left singular vector u_i = U V_i / s_i
      std::vector<double> u(nn, 0.0);
      for (int j = 0; j < m; j++) for (int k = 0; k < nn; k++) u[k] += U[j][k] * V[j][i];
      for (int k = 0; k < nn; k++) u[k] /= sv[i];
      orthoBasis.push_back(std::move(u));
    }
  }
  void orthogonalize(std::vector<double>& x) const {
    const int nn = n();
    std::vector<double> proj(nn, 0.0);
    for (const std::vector<double>& u : orthoBasis) {
      double dot = 0;
      for (int k = 0; k < nn; k++) dot += u[k] * x[k];
      for (int k = 0; k < nn; k++) proj[k] += u[k] * dot;
    }
    for (int k = 0; k < nn; k++) x[k] -= proj[k];
  }

  // Symmetric solve by LDL^T with diagonal pivoting — the arithmetic of ldltSolveInPlace (lie_dev.h), element for element and in the
  // same order, on the TRANSPOSED triangle (m[c * ld + r] = lower[r][c]) so that the rank update of step k runs along contiguous rows:
  // acc[r] += L[r][j] * temp[j] for j ascending, vectorised over r.  Host only (the 68x68 system of the window).
  // (A 512-bit build of this body was timed on the GPU box's EPYC 9575F: 12.5 us per solve either way — the factorisation is bound by its dependent add chains, not by vector width.)
  __attribute__((target("avx2"))) static void ldltSolveTransposed(double* m, const int ld, double* d, const int n) {
    constexpr int NMAX = 4 + 8 * BA_MAXF_CAP;
    int tr[NMAX];
    double temp[NMAX], acc[NMAX];
    bool zero = false;
    for (int k = 0; k < n; k++) {
      int big = k;
      double bigv = std::fabs(m[k * ld + k]);
      for (int i = k + 1; i < n; i++) { const double v = std::fabs(m[i * ld + i]); if (v > bigv) { bigv = v; big = i; } }
      tr[k] = big;
      if (k != big) {
        for (int c = 0; c < k; c++) std::swap(m[c * ld + k], m[c * ld + big]);
        for (int r = big + 1; r < n; r++) std::swap(m[k * ld + r], m[big * ld + r]);
        std::swap(m[k * ld + k], m[big * ld + big]);
        for (int i = k + 1; i < big; i++) std::swap(m[k * ld + i], m[i * ld + big]);
      }
      if (k > 0) {
        for (int j = 0; j < k; j++) temp[j] = m[j * ld + j] * m[j * ld + k];
        for (int r = k; r < n; r++) acc[r] = 0;
        int j = 0;
        for (; j + 4 <= k; j += 4) {
          const double t0 = temp[j], t1 = temp[j + 1], t2 = temp[j + 2], t3 = temp[j + 3];
          const double *r0 = m + j * ld, *r1 = r0 + ld, *r2 = r1 + ld, *r3 = r2 + ld;
          for (int r = k; r < n; r++) acc[r] = (((acc[r] + r0[r] * t0) + r1[r] * t1) + r2[r] * t2) + r3[r] * t3;
        }
        for (; j < k; j++) { const double tj = temp[j]; const double* row = m + j * ld; for (int r = k; r < n; r++) acc[r] += row[r] * tj; }
        double* rk = m + k * ld;
        for (int r = k; r < n; r++) rk[r] -= acc[r];
      }
      const double akk = m[k * ld + k];
      const bool ok = std::fabs(akk) > 0;
      if (k == 0 && !ok) { zero = true; break; }
      if (ok) { double* rk = m + k * ld; for (int r = k + 1; r < n; r++) rk[r] /= akk; }
    }
    if (zero) { for (int i = 0; i < n; i++) d[i] = 0; return; }
    for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
    for (int j = 0; j < n; j++) { const double dj = d[j]; const double* row = m + j * ld; for (int i = j + 1; i < n; i++) d[i] -= row[i] * dj; }
    for (int i = 0; i < n; i++) { if (std::fabs(m[i * ld + i]) > 2.2250738585072014e-308) d[i] /= m[i * ld + i]; else d[i] = 0; }
    for (int i = n - 1; i >= 0; i--) { double s = d[i]; const double* row = m + i * ld; for (int j = i + 1; j < n; j++) s -= row[j] * d[j]; d[i] = s; }
    for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  }

  // accumulateLF_MT's result (EnergyFunctional.cpp:223-233): the stitched system of the residuals kept linearised — HLraw / bLraw, filled by the accumulation when the graph
  // carries any (dmvio_hip_ba_fix_linearization), otherwise empty = zero — plus the priors stitchDoubleInternal adds last (AccumulatedTopHessian.cpp:292-302).
  // HLd: the prior diagonal, to be added to HLraw's; bL: the complete right-hand side.  Returns whether HLraw holds a system.
  bool lfTop(double* HLd, double* bL) const {
    const bool haveL = HLraw.size() == (size_t)n() * n();
    for (int i = 0; i < 4; i++) { HLd[i] = cPrior[i]; bL[i] = (haveL ? bLraw[i] : 0.0) + cPrior[i] * (double)cDeltaF[i]; }
    for (int f = 0; f < F; f++) for (int i = 0; i < 8; i++) { const int q = 4 + 8 * f + i; HLd[q] = fr[f].prior[i]; bL[q] = (haveL ? bLraw[q] : 0.0) + fr[f].prior[i] * fr[f].delta_prior[i]; }
    return haveL;
  }
  // Everything of solveSystemF that depends on the window state only (nullspaces, orthogonalisation basis, prior right-hand side):
  // the GN loop runs it on the host while the accumulation kernels are in flight.
  void prepareSolve() {
    const int nn = n();
    getNullspaces();
    prepareOrthogonalize();
    std::vector<double> d(nn);
    for (int i = 0; i < 4; i++) d[i] = (double)cDeltaF[i];
    for (int f = 0; f < F; f++) for (int i = 0; i < 8; i++) d[4 + 8 * f + i] = fr[f].delta[i];
    bPriorM.assign(nn, 0.0);
    const std::vector<double>&HMs = priorH(), &bMs = priorb();   // bM_top resp. bMGTSAM_top (EnergyFunctional.cpp:864-865)
    const bool haveM = HMs.size() == (size_t)nn * nn;
    for (int i = 0; i < nn; i++) { double s = haveM ? bMs[i] : 0.0; if (haveM) for (int j = 0; j < nn; j++) s += HMs[(size_t)i * nn + j] * d[j]; bPriorM[i] = s; }
    solvePrepared = true;
  }

  // EnergyFunctional::solveSystemF after the accumulations: HA,bA / Hsc,bsc come from the device; priors (accumulateLF with no
  // linearised residuals) and the marginalisation prior are added here.
  void solveSystem(int iteration, double lambda, const double* HA, const double* bA, const double* Hsc, const double* bsc, std::vector<double>& x) {
    const int nn = n();
    if (!solvePrepared) prepareSolve();
    solvePrepared = false;
    const bool haveM = HM.size() == (size_t)nn * nn;
    hfScratch.resize((size_t)nn * nn); htScratch.resize((size_t)nn * nn);
    double* HF = hfScratch.data();
    double* Ht = htScratch.data();
    double bF[4 + 8 * BA_MAXF_CAP], sv[4 + 8 * BA_MAXF_CAP], bs[4 + 8 * BA_MAXF_CAP];
    // HFinal_top = HL_top + HM + HA_top, bFinal_top = bL_top + bM_top + bA_top - b_sc (EnergyFunctional.cpp:906-907), summed in that order: HL_top / bL_top hold only the priors
    // (stitchDoubleInternal usePrior, AccumulatedTopHessian.cpp:292-302; no linearised residuals outside marginalisation), zero elsewhere
    double HLd[4 + 8 * BA_MAXF_CAP], bL[4 + 8 * BA_MAXF_CAP];
    const bool haveL = lfTop(HLd, bL);
    // only the lower triangle of HFinal_top - H_sc / (1 + lambda) is read below: one pass over it, per element the operations of the reference's whole-matrix statements
    // ((HL + HM) + HA, the diagonal times (1 + lambda), minus H_sc * fac) in their order
    const double fac = 1.0f / (1 + lambda);
    for (int i = 0; i < nn; i++)
      for (int j = 0; j <= i; j++) {
        const size_t o = (size_t)i * nn + j;
        double v = (((haveL ? HLraw[o] : 0.0) + (i == j ? HLd[i] : 0.0)) + (haveM ? HM[o] : 0.0)) + HA[o];
        if (i == j) v *= (1 + lambda);
        HF[o] = v - Hsc[o] * fac;
      }
    for (int i = 0; i < nn; i++) bF[i] = ((bL[i] + bPriorM[i]) + bA[i]) - bsc[i];
    for (int i = 0; i < nn; i++) sv[i] = 1.0 / std::sqrt(HF[(size_t)i * nn + i] + 10);
    // Jacobi-scaled system, stored transposed for ldltSolveTransposed (which reads the lower triangle of the untransposed matrix)
    for (int i = 0; i < nn; i++) { for (int j = 0; j <= i; j++) Ht[(size_t)j * nn + i] = sv[i] * HF[(size_t)i * nn + j] * sv[j]; bs[i] = sv[i] * bF[i]; }
    ldltSolveTransposed(Ht, nn, bs, nn);
    x.resize(nn);
    for (int i = 0; i < nn; i++) x[i] = sv[i] * bs[i];
    if (iteration >= 2) orthogonalize(x);  // SOLVER_ORTHOGONALIZE_X_LATER (settings.cpp:81)
    lastX = x;
  }
  // The GTSAM branch of solveSystemF (EnergyFunctional.cpp:958-969): what the reference hands to BAGTSAMIntegration::computeBAUpdate —
  //   HPassed   = (HL_top + HMForGTSAM + HA_top) with the diagonal times (1 + lambda), minus H_sc / (1 + lambda)
  //   bPassed   = bL_top + bMGTSAM_top + bA_top - b_sc
  //   HNoLambda = HL_top + HMForGTSAM + HA_top - H_sc
  // each summed in the reference's order.  n x n row-major (the matrices are symmetric).  Requires gtsam == true before prepareSolve().
  void buildGtsamSystem(double lambda, const double* HA, const double* bA, const double* Hsc, const double* bsc, double* HPassed, double* bPassed, double* HNoLambda) {
    const int nn = n();
    if (!solvePrepared) prepareSolve();
    solvePrepared = false;
    const std::vector<double>& HMs = priorH();
    const bool haveM = HMs.size() == (size_t)nn * nn;
    double HLd[4 + 8 * BA_MAXF_CAP], bL[4 + 8 * BA_MAXF_CAP];
    const bool haveL = lfTop(HLd, bL);
    const double fac = 1.0f / (1 + lambda);
    for (int i = 0; i < nn; i++)
      for (int j = 0; j < nn; j++) {
        const size_t o = (size_t)i * nn + j;
        const double top = (((haveL ? HLraw[o] : 0.0) + (i == j ? HLd[i] : 0.0)) + (haveM ? HMs[o] : 0.0)) + HA[o];
        HNoLambda[o] = top - Hsc[o];
        HPassed[o] = (i == j ? top * (1 + lambda) : top) - Hsc[o] * fac;
      }
    for (int i = 0; i < nn; i++) bPassed[i] = ((bL[i] + bPriorM[i]) + bA[i]) - bsc[i];
  }
  // tail of solveSystemF for a step that came from outside (EnergyFunctional.cpp:977-985): orthogonalisation from iteration 2, lastX
  void finishExternalSolve(int iteration, std::vector<double>& x) {
    if (iteration >= 2) orthogonalize(x);
    lastX = x;
  }
  // EnergyFunctional::marginalizeFrame, visual-only branch (EnergyFunctional.cpp:570-640): returns the prior of the window without
  // keyframe idx.  8x8 block inverse: Gauss-Jordan with partial pivoting (the reference: Eigen's PartialPivLU behind Mat88::inverse()).
  void marginalizeFrame(int idx, std::vector<double>& HMn, std::vector<double>& bMn) const {
    const int odim = n(), ndim = odim - 8, io = idx * 8 + 4;
    std::vector<double> Hm = HM, bm = bM;
    if (Hm.size() != (size_t)odim * odim) { Hm.assign((size_t)odim * odim, 0.0); bm.assign(odim, 0.0); }
    std::vector<int> perm;
    for (int i = 0; i < io; i++) perm.push_back(i);
    for (int i = io + 8; i < odim; i++) perm.push_back(i);
    for (int i = io; i < io + 8; i++) perm.push_back(i);
    std::vector<double> Hp((size_t)odim * odim), bp(odim);
    for (int i = 0; i < odim; i++) { bp[i] = bm[perm[i]]; for (int j = 0; j < odim; j++) Hp[(size_t)i * odim + j] = Hm[(size_t)perm[i] * odim + perm[j]]; }
    for (int i = 0; i < 8; i++) { Hp[(size_t)(ndim + i) * odim + ndim + i] += fr[idx].prior[i]; bp[ndim + i] += fr[idx].prior[i] * fr[idx].delta_prior[i]; }
    std::vector<double> sv(odim), svi(odim);
    for (int i = 0; i < odim; i++) { sv[i] = std::sqrt(std::fabs(Hp[(size_t)i * odim + i]) + 10); svi[i] = 1.0 / sv[i]; }
    for (int i = 0; i < odim; i++) { bp[i] *= svi[i]; for (int j = 0; j < odim; j++) Hp[(size_t)i * odim + j] = svi[i] * Hp[(size_t)i * odim + j] * svi[j]; }
    double A[8][16];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { A[i][j] = Hp[(size_t)(ndim + i) * odim + ndim + j]; A[i][8 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 8; c++) {
      int piv = c;
      for (int r = c + 1; r < 8; r++) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
      if (piv != c) for (int k = 0; k < 16; k++) std::swap(A[c][k], A[piv][k]);
      const double d = A[c][c];
      for (int k = 0; k < 16; k++) A[c][k] /= d;
      for (int r = 0; r < 8; r++) if (r != c) { const double m = A[r][c]; if (m != 0) for (int k = 0; k < 16; k++) A[r][k] -= m * A[c][k]; }
    }
    std::vector<double> bli((size_t)ndim * 8);
    for (int i = 0; i < ndim; i++) for (int j = 0; j < 8; j++) { double s = 0; for (int k = 0; k < 8; k++) s += Hp[(size_t)(ndim + k) * odim + i] * A[k][8 + j]; bli[(size_t)i * 8 + j] = s; }
    for (int i = 0; i < ndim; i++) {
      for (int j = 0; j < ndim; j++) { double s = 0; for (int k = 0; k < 8; k++) s += bli[(size_t)i * 8 + k] * Hp[(size_t)(ndim + k) * odim + j]; Hp[(size_t)i * odim + j] -= s; }
      double sb = 0; for (int k = 0; k < 8; k++) sb += bli[(size_t)i * 8 + k] * bp[ndim + k];
      bp[i] -= sb;
    }
    HMn.assign((size_t)ndim * ndim, 0.0); bMn.assign(ndim, 0.0);
    for (int i = 0; i < ndim; i++) {
      bMn[i] = sv[i] * bp[i];
      for (int j = 0; j < ndim; j++) HMn[(size_t)i * ndim + j] = 0.5 * (sv[i] * Hp[(size_t)i * odim + j] * sv[j] + sv[j] * Hp[(size_t)j * odim + i] * sv[i]);
    }
  }
  // adHTdeltaF of EnergyFunctional::setDeltaF (EnergyFunctional.cpp:175-198): F*F x 8, index h + F*t
  void adHTdeltaF(std::vector<float>& out) const {
    out.assign((size_t)F * F * 8, 0.f);
    for (int hh = 0; hh < F; hh++)
      for (int t = 0; t < F; t++) {
        const size_t idx = (size_t)hh + (size_t)t * F;
        float dh[8], dt[8];
        for (int i = 0; i < 8; i++) { dh[i] = (float)(fr[hh].state[i] - fr[hh].state_zero[i]); dt[i] = (float)(fr[t].state[i] - fr[t].state_zero[i]); }
        for (int c = 0; c < 8; c++) {
          float s1 = 0, s2 = 0;
          for (int r = 0; r < 8; r++) { s1 += dh[r] * adHostF[idx * 64 + r * 8 + c]; s2 += dt[r] * adTargetF[idx * 64 + r * 8 + c]; }
          out[idx * 8 + c] = s1 + s2;
        }
      }
  }
  // xc (4) and xAd (F*F x 8, index h*F + t) of resubstituteF_MT; frame / calib steps
  void prepareResubstitute(const std::vector<double>& x, float xc[4], std::vector<float>& xAd) {
    const int nn = n();
    std::vector<float> xF(nn);
    for (int i = 0; i < nn; i++) xF[i] = (float)x[i];
    for (int i = 0; i < 4; i++) { c_step[i] = -x[i]; xc[i] = xF[i]; }
    xAd.assign((size_t)F * F * 8, 0.f);
    for (int hh = 0; hh < F; hh++) {
      for (int i = 0; i < 8; i++) fr[hh].step[i] = -x[4 + 8 * hh + i];
      fr[hh].step[8] = fr[hh].step[9] = 0;
      for (int t = 0; t < F; t++) {
        const size_t o = ((size_t)hh + (size_t)F * t) * 64;
        for (int c = 0; c < 8; c++) {
          float s1 = 0, s2 = 0;
          for (int r = 0; r < 8; r++) { s1 += xF[4 + 8 * hh + r] * adHostF[o + r * 8 + c]; s2 += xF[4 + 8 * t + r] * adTargetF[o + r * 8 + c]; }
          xAd[((size_t)F * hh + t) * 8 + c] = s1 + s2;
        }
      }
    }
  }
  double calcLEnergyFrames() const {
    double E = 0;
    for (int f = 0; f < F; f++) for (int i = 0; i < 8; i++) E += fr[f].delta_prior[i] * fr[f].prior[i] * fr[f].delta_prior[i];
    float ec = 0;
    for (int i = 0; i < 4; i++) ec += cDeltaF[i] * cPriorF[i] * cDeltaF[i];
    return E + ec;  // + sum_p deltaF^2 priorF, which is zero while idepth_zero follows idepth (doStepFromBackup) — added by the caller when not
  }
  double calcMEnergy() const {
    const int nn = n();
    std::vector<double> d(nn);
    for (int i = 0; i < 4; i++) d[i] = (double)cDeltaF[i];
    for (int f = 0; f < F; f++) for (int i = 0; i < 8; i++) d[4 + 8 * f + i] = fr[f].delta[i];
    const std::vector<double>&HMs = priorH(), &bMs = priorb();   // delta.dot(2 bM + HM delta) resp. the ForGTSAM pair (EnergyFunctional.cpp:332-341)
    if (HMs.size() != (size_t)nn * nn) return 0.0;
    double s = 0;
    for (int i = 0; i < nn; i++) { double t = 2 * bMs[i]; for (int j = 0; j < nn; j++) t += HMs[(size_t)i * nn + j] * d[j]; s += d[i] * t; }
    return s;
  }
  void backupFrames() {
    for (int i = 0; i < 4; i++) c_value_backup[i] = c_value[i];
    for (int f = 0; f < F; f++) for (int i = 0; i < 10; i++) fr[f].state_backup[i] = fr[f].state[i];
  }
  // frame / calib part of doStepFromBackup; returns the four frame sums (A, B, T, R) already divided by F
  void stepFrames(float stepfac, float sums[4]) {
    double nv[4];
    for (int i = 0; i < 4; i++) nv[i] = c_value_backup[i] + stepfac * c_step[i];
    calibSetValue(nv);
    float sA = 0, sB = 0, sT = 0, sR = 0;
    for (int f = 0; f < F; f++) {
      double st[10];
      for (int i = 0; i < 10; i++) st[i] = fr[f].state_backup[i] + (double)stepfac * fr[f].step[i];
      frameSetState(fr[f], st);
      sA += fr[f].step[6] * fr[f].step[6]; sB += fr[f].step[7] * fr[f].step[7];
      sT += fr[f].step[0] * fr[f].step[0] + fr[f].step[1] * fr[f].step[1] + fr[f].step[2] * fr[f].step[2];
      sR += fr[f].step[3] * fr[f].step[3] + fr[f].step[4] * fr[f].step[4] + fr[f].step[5] * fr[f].step[5];
    }
    sums[0] = sA / F; sums[1] = sB / F; sums[2] = sT / F; sums[3] = sR / F;
  }
  void restoreFrames() {
    calibSetValue(c_value_backup);
    for (int f = 0; f < F; f++) frameSetState(fr[f], fr[f].state_backup);
  }
  float newFrameEnergyTH(std::vector<float>& allResVec) const {
    if (allResVec.empty()) return 12 * 12 * 8;
    const int nthIdx = (int)(S.frameEnergyTHN * allResVec.size());
    std::nth_element(allResVec.begin(), allResVec.begin() + nthIdx, allResVec.end());
    const float nthElement = sqrtf(allResVec[nthIdx]);
    float th = nthElement * S.frameEnergyTHFacMedian;
    th = 26.0f * S.frameEnergyTHConstWeight + th * (1 - S.frameEnergyTHConstWeight);
    th = th * th;
    th *= S.overallEnergyTHWeight * S.overallEnergyTHWeight;
    return th;
  }
};

}  // namespace dmv
