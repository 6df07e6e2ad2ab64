// Test infrastructure (tests/test_host_algebra_cpu.py builds it, HOST code only, no device touched): csrc/ba_host.hpp's BAHost — the window state and the small dense algebra
// the library runs on the host around the BA kernels — behind a few plain C functions, set up exactly like dmvio_hip_ba_set_window / _set_frame_state do it (csrc/capi_ba.hip),
// so its precalc tables, adjoints, nullspaces, energies and frame steps can be held against the oracle's (pinned to the reference) without a GPU.
#include "ba_host.hpp"
using namespace dmv;

extern "C" {
void* bah_create(int F, const double* pose7_w2c, const double* aff_ab, const float* exposures, const int* frameIDs, const double fxfycxcy[4]) {
  BAHost* H = new BAHost();
  H->F = F;
  H->calibInitScaled(fxfycxcy);
  for (int f = 0; f < F; f++) {
    BAFrameHost& fr = H->fr[f];
    fr = BAFrameHost();
    fr.ab_exposure = exposures ? exposures[f] : 1.0f;
    fr.frameID = frameIDs ? frameIDs[f] : f;
    fr.evalPT = poseFrom7(pose7_w2c + 7 * f);
    const double a = aff_ab ? aff_ab[2 * f] : 0.0, bb = aff_ab ? aff_ab[2 * f + 1] : 0.0;
    double st[10] = {0, 0, 0, 0, 0, 0, (1.0f / 10.0f) * a, (1.0f / 1000.0f) * bb, 0, 0};
    for (int i = 0; i < 10; i++) { fr.step[i] = 0; fr.state_backup[i] = 0; }
    BAHost::frameSetState(fr, st);
    BAHost::frameSetStateZero(fr, fr.state);
  }
  const int n = H->n();
  H->HM.assign((size_t)n * n, 0.0); H->bM.assign(n, 0.0);
  for (int f = 0; f < F; f++) H->frameTakeData(H->fr[f]);
  H->setAdjointsF();
  H->setPrecalcValues();
  return H;
}
void bah_destroy(void* p) { delete (BAHost*)p; }
// dmvio_hip_ba_set_frame_state: state in the reference's unscaled units
void bah_set_frame_state(void* p, int f, const double state10[10]) {
  BAHost* H = (BAHost*)p;
  BAHost::frameSetState(H->fr[f], state10);
  H->frameTakeData(H->fr[f]);
  H->setPrecalcValues();
}
void bah_get_frame(void* p, int f, double pose7_w2c[7], double state10[10]) {
  BAHost* H = (BAHost*)p;
  poseTo7(H->fr[f].w2c, pose7_w2c);
  for (int i = 0; i < 10; i++) state10[i] = H->fr[f].state[i];
}
// 37 floats like orc_ba_get_precalc: KRKi 9, Kt 3, R0 9, t0 3, aff 2, b0 1 (the last 9 of the oracle's record, PRE_RTll, are not kept by the library)
void bah_get_precalc(void* p, int h, int t, float out28[28]) {
  BAHost* H = (BAHost*)p;
  const BAPrecalc& pc = H->pre[(size_t)h + (size_t)H->F * t];
  memcpy(out28, pc.KRKi, 36); memcpy(out28 + 9, pc.Kt, 12); memcpy(out28 + 12, pc.R0, 36); memcpy(out28 + 21, pc.t0, 12);
  out28[24] = pc.aff0; out28[25] = pc.aff1; out28[26] = pc.b0; out28[27] = 0;
}
void bah_get_adjoints(void* p, double* adHost, double* adTarget) {
  BAHost* H = (BAHost*)p;
  const size_t n = (size_t)H->F * H->F * 64;
  memcpy(adHost, H->adHost.data(), n * 8); memcpy(adTarget, H->adTarget.data(), n * 8);
}
void bah_get_nullspaces(void* p, double* out7xn) {
  BAHost* H = (BAHost*)p;
  H->getNullspaces();
  const int n = H->n();
  for (int i = 0; i < 7; i++) memcpy(out7xn + (size_t)i * n, H->nsp[i].data(), sizeof(double) * n);
}
void bah_energies(void* p, double* EL_frames, double* EM) {
  BAHost* H = (BAHost*)p;
  *EL_frames = H->calcLEnergyFrames(); *EM = H->calcMEnergy();
}
void bah_set_marg_prior(void* p, const double* HM, const double* bM) {
  BAHost* H = (BAHost*)p;
  const int n = H->n();
  H->HM.assign(HM, HM + (size_t)n * n); H->bM.assign(bM, bM + n);
}
// the stitched system of the residuals kept linearised (accumulateLF_MT without the priors): what dmvio_hip_ba_fix_linearization's accumulation leaves in BAHost
void bah_set_lf_raw(void* p, const double* HL, const double* bL) {
  BAHost* H = (BAHost*)p;
  const int n = H->n();
  if (HL) { H->HLraw.assign(HL, HL + (size_t)n * n); H->bLraw.assign(bL, bL + n); } else { H->HLraw.clear(); H->bLraw.clear(); }
}
void bah_solve_system(void* p, int iteration, double lambda, const double* HA, const double* bA, const double* Hsc, const double* bsc, double* x) {
  BAHost* H = (BAHost*)p;
  std::vector<double> xv;
  H->solveSystem(iteration, lambda, HA, bA, Hsc, bsc, xv);
  memcpy(x, xv.data(), sizeof(double) * xv.size());
}
}
