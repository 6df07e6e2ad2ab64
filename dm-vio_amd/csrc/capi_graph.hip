// libdmvio_hip.so — the window's point / residual graph kept RESIDENT across keyframes (include/dmvio_hip.h, "window graph" section).
//
// The reference never rebuilds its graph: EnergyFunctional mutates it in place, one call per change —
//     insertResidual  src/dso/OptimizationBackend/EnergyFunctional.cpp:435-446     appended to EFPoint::residualsAll
//     insertFrame     :447-484                                                     appended to EnergyFunctional::frames
//     insertPoint     :485-499                                                     appended to EFFrame::points
//     dropResidual    :500-518                                                     the last residual of the point takes the dropped one's place
//     removePoint     :766-782                                                     its residuals dropped, the frame's last point takes its place
//     marginalizeFrame :641-646                                                    the later frames move down by one (order kept)
//     makeIDX         :997-1017                                                    allPoints = frames in order, their points in order — the order every accumulator adds in
// — while dmvio_hip_ba_set_graph takes the whole graph as flat arrays, which made a host rebuild them from its pointer graph every keyframe (1.6 ms for 2000 points /
// 12.8k residuals in tests/dropin, more than optimize itself).  This file is the missing half: a host-side mirror with exactly those mutators, addressed the way the
// reference addresses its own objects (EFFrame::idx, EFPoint::idxInPoints, EFResidual::idxInAll), so that an adapter forwards each EnergyFunctional call with the
// indices it already holds and never walks the graph again.  dmvio_hip_ba_set_graph_from (capi_ba.hip) flattens the mirror — compact records, no pointer chasing —
// in makeIDX order and builds the device arrays from it.
//
// Host only: no device, no stream; one mutex per graph.
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "internal.h"

using namespace dmv;

extern "C" {

dmvio_hip_graph* dmvio_hip_graph_create(void) { return new (std::nothrow) dmvio_hip_graph(); }
void dmvio_hip_graph_destroy(dmvio_hip_graph* g) { delete g; }

#define G_LOCK(g) if (!(g)) return failmsg("graph: null handle"); std::lock_guard<std::mutex> lk_((g)->mu)
#define G_POINT(g, host, idx, what)                                                                                         \
  if ((host) < 0 || (host) >= (int)(g)->frames.size()) return failmsg(std::string(what) + ": host frame out of range");    \
  if ((idx) < 0 || (idx) >= (int)(g)->frames[host].size()) return failmsg(std::string(what) + ": point index out of range"); \
  DmvGraphPoint& P = (g)->frames[host][idx]

int dmvio_hip_graph_clear(dmvio_hip_graph* g) {
  G_LOCK(g);
  g->frames.clear(); g->linPool.clear(); g->linFree.clear(); g->nLin = 0; g->nPoints = g->nRes = g->nDangling = 0; g->version++;
  return 0;
}

// EnergyFunctional::insertFrame: the new keyframe goes to the end; returns its index (EFFrame::idx)
int dmvio_hip_graph_insert_frame(dmvio_hip_graph* g) {
  G_LOCK(g);
  if ((int)g->frames.size() >= DMV_GRAPH_MAX_FRAMES) return failmsg("graph_insert_frame: more than " + std::to_string(DMV_GRAPH_MAX_FRAMES) + " keyframes");
  g->frames.emplace_back();
  g->version++;
  return (int)g->frames.size() - 1;
}

// EnergyFunctional::marginalizeFrame, the part that touches the graph (:641-646): the frame leaves, later frames move down by one.  The frame hosts no point any more
// (FullSystem::marginalizeFrame asserts it, FullSystemMarginalize.cpp:160); residuals that still TARGET it — the reference drops them right after
// (FullSystemMarginalize.cpp:169-196) — are kept as dangling until their dropResidual arrives, and a graph with dangling residuals is refused by dmvio_hip_ba_set_graph_from.
int dmvio_hip_graph_remove_frame(dmvio_hip_graph* g, int idx) {
  G_LOCK(g);
  if (idx < 0 || idx >= (int)g->frames.size()) return failmsg("graph_remove_frame: index out of range");
  if (!g->frames[idx].empty()) return failmsg("graph_remove_frame: the frame still hosts " + std::to_string(g->frames[idx].size()) + " points");
  g->frames.erase(g->frames.begin() + idx);
  for (auto& fr : g->frames)
    for (DmvGraphPoint& P : fr)
      for (int k = 0; k < P.nres; k++) {
        if (P.target[k] == idx) { P.target[k] = -1; g->nDangling++; }
        else if (P.target[k] > idx) P.target[k]--;
      }
  g->version++;
  return 0;
}

// EnergyFunctional::insertPoint + EFPoint::takeData: appended to its host frame's points; returns EFPoint::idxInPoints
int dmvio_hip_graph_insert_point(dmvio_hip_graph* g, int host, float u, float v, float idepth, const float* color8, const float* weights8, int hasDepthPrior) {
  G_LOCK(g);
  if (!color8 || !weights8) return failmsg("graph_insert_point: null argument");
  if (host < 0 || host >= (int)g->frames.size()) return failmsg("graph_insert_point: host frame out of range");
  DmvGraphPoint P;
  P.u = u; P.v = v; P.idepth = idepth; P.prior = hasDepthPrior ? 1 : 0; P.nres = 0;
  for (int k = 0; k < 8; k++) { P.color[k] = color8[k]; P.weights[k] = weights8[k]; }
  g->frames[host].push_back(P);
  g->nPoints++; g->version++;
  return (int)g->frames[host].size() - 1;
}

// EnergyFunctional::removePoint: whatever residuals it still has go with it, the frame's LAST point takes its place (and its index)
int dmvio_hip_graph_remove_point(dmvio_hip_graph* g, int host, int idxInPoints) {
  G_LOCK(g);
  G_POINT(g, host, idxInPoints, "graph_remove_point");
  for (int k = 0; k < P.nres; k++) {
    if (P.target[k] < 0) g->nDangling--;
    if (P.lin[k] >= 0) { g->linFree.push_back(P.lin[k]); g->nLin--; }
  }
  g->nRes -= P.nres;
  P = g->frames[host].back();
  g->frames[host].pop_back();
  g->nPoints--; g->version++;
  return 0;
}

// EnergyFunctional::insertResidual: appended to the point's residualsAll; returns EFResidual::idxInAll
int dmvio_hip_graph_insert_residual(dmvio_hip_graph* g, int host, int idxInPoints, int target) {
  G_LOCK(g);
  G_POINT(g, host, idxInPoints, "graph_insert_residual");
  if (target < 0 || target >= (int)g->frames.size() || target == host) return failmsg("graph_insert_residual: bad target frame");
  if (P.nres >= DMV_GRAPH_MAX_FRAMES) return failmsg("graph_insert_residual: the point already has " + std::to_string((int)P.nres) + " residuals");
  P.target[P.nres] = (short)target; P.lin[P.nres] = -1;
  g->nRes++; g->version++;
  return P.nres++;
}

// EnergyFunctional::dropResidual: the point's LAST residual takes the dropped one's place (and its idxInAll)
int dmvio_hip_graph_drop_residual(dmvio_hip_graph* g, int host, int idxInPoints, int idxInAll) {
  G_LOCK(g);
  G_POINT(g, host, idxInPoints, "graph_drop_residual");
  if (idxInAll < 0 || idxInAll >= P.nres) return failmsg("graph_drop_residual: residual index out of range");
  if (P.target[idxInAll] < 0) g->nDangling--;
  if (P.lin[idxInAll] >= 0) { g->linFree.push_back(P.lin[idxInAll]); g->nLin--; }
  P.target[idxInAll] = P.target[P.nres - 1]; P.lin[idxInAll] = P.lin[P.nres - 1];
  P.nres--;
  g->nRes--; g->version++;
  return 0;
}

// EFResidual::fixLinearizationF's result on the caller's side (EnergyFunctionalStructs.cpp:85-113): isLinearized = true with the frozen Jacobian (EFResidual::J, 74 floats in
// dmvio_hip_ba_get_full_jacobians' layout) and res_toZeroF (8).  J74 == NULL: isLinearized = false again (FullSystem.cpp:840-843).  A value, not structure: the version stays.
int dmvio_hip_graph_set_residual_linearized(dmvio_hip_graph* g, int host, int idxInPoints, int idxInAll, const float* J74, const float* res_toZeroF) {
  G_LOCK(g);
  G_POINT(g, host, idxInPoints, "graph_set_residual_linearized");
  if (idxInAll < 0 || idxInAll >= P.nres) return failmsg("graph_set_residual_linearized: residual index out of range");
  if (!J74) {
    if (P.lin[idxInAll] >= 0) { g->linFree.push_back(P.lin[idxInAll]); g->nLin--; P.lin[idxInAll] = -1; }
    return 0;
  }
  if (!res_toZeroF) return failmsg("graph_set_residual_linearized: null res_toZeroF");
  int slot = P.lin[idxInAll];
  if (slot < 0) {
    if (!g->linFree.empty()) { slot = g->linFree.back(); g->linFree.pop_back(); }
    else { slot = (int)g->linPool.size(); g->linPool.emplace_back(); }
    P.lin[idxInAll] = slot; g->nLin++;
  }
  memcpy(g->linPool[slot].J, J74, sizeof(float) * 74);
  memcpy(g->linPool[slot].res_toZeroF, res_toZeroF, sizeof(float) * 8);
  return 0;
}
int dmvio_hip_graph_linearized_count(dmvio_hip_graph* g) {
  G_LOCK(g);
  return g->nLin;
}
// ... in flat (makeIDX) order beside dmvio_hip_graph_export's arrays: R flags, R x 74 and R x 8 floats (rows of the residuals that are not linearised are zeroed); any may be NULL
int dmvio_hip_graph_export_linearized(dmvio_hip_graph* g, unsigned char* isLinearized, float* J74, float* res_toZeroF) {
  G_LOCK(g);
  int ri = 0;
  for (auto& fr : g->frames)
    for (const DmvGraphPoint& P : fr)
      for (int k = 0; k < P.nres; k++, ri++) {
        const int slot = P.lin[k];
        if (isLinearized) isLinearized[ri] = slot >= 0 ? 1 : 0;
        if (J74) { if (slot >= 0) memcpy(J74 + (size_t)ri * 74, g->linPool[slot].J, sizeof(float) * 74); else memset(J74 + (size_t)ri * 74, 0, sizeof(float) * 74); }
        if (res_toZeroF) { if (slot >= 0) memcpy(res_toZeroF + (size_t)ri * 8, g->linPool[slot].res_toZeroF, sizeof(float) * 8); else memset(res_toZeroF + (size_t)ri * 8, 0, sizeof(float) * 8); }
      }
  return 0;
}

int dmvio_hip_graph_set_idepth(dmvio_hip_graph* g, int host, int idxInPoints, float idepth) {
  G_LOCK(g);
  G_POINT(g, host, idxInPoints, "graph_set_idepth");
  P.idepth = idepth;   // a value, not structure: the version stays
  return 0;
}

// the inverse depths of ALL points in flat (makeIDX) order — what an optimisation returns (dmvio_hip_ba_get_points on a window built by dmvio_hip_ba_set_graph_from)
int dmvio_hip_graph_set_idepths(dmvio_hip_graph* g, int N, const float* idepth) {
  G_LOCK(g);
  if (!idepth || N != g->nPoints) return failmsg("graph_set_idepths: N differs from the graph's point count");
  if (g->flat_version != g->version)
    return failmsg("graph_set_idepths: the graph changed (or was never flattened) since dmvio_hip_ba_set_graph_from / dmvio_hip_graph_export — the flat order of these values is not the graph's any more");
  int i = 0;
  for (auto& fr : g->frames) for (DmvGraphPoint& P : fr) P.idepth = idepth[i++];
  return 0;
}

int dmvio_hip_graph_counts(dmvio_hip_graph* g, int* F, int* N, int* R) {
  G_LOCK(g);
  if (F) *F = (int)g->frames.size();
  if (N) *N = g->nPoints;
  if (R) *R = g->nRes;
  return 0;
}
// points hosted by a frame / residuals of a point: the sizes of EFFrame::points and EFPoint::residualsAll
int dmvio_hip_graph_frame_points(dmvio_hip_graph* g, int host) {
  G_LOCK(g);
  if (host < 0 || host >= (int)g->frames.size()) return failmsg("graph_frame_points: host frame out of range");
  return (int)g->frames[host].size();
}
int dmvio_hip_graph_point_residuals(dmvio_hip_graph* g, int host, int idxInPoints) {
  G_LOCK(g);
  G_POINT(g, host, idxInPoints, "graph_point_residuals");
  return P.nres;
}

// The graph as the flat arrays of dmvio_hip_ba_set_graph, in makeIDX order (frames, their points, each point's residualsAll).  Any output may be NULL; sizes from
// dmvio_hip_graph_counts.  A dangling residual (its target frame removed, its dropResidual not yet seen) is exported with target -1.
int dmvio_hip_graph_export(dmvio_hip_graph* g, int* host, float* u, float* v, float* idepth, float* color8, float* weights8, unsigned char* hasDepthPrior, int* res_point,
                           int* res_target) {
  G_LOCK(g);
  g->flat_version = g->version;
  int pi = 0, ri = 0;
  for (int f = 0; f < (int)g->frames.size(); f++)
    for (const DmvGraphPoint& P : g->frames[f]) {
      if (host) host[pi] = f;
      if (u) u[pi] = P.u;
      if (v) v[pi] = P.v;
      if (idepth) idepth[pi] = P.idepth;
      if (color8) for (int k = 0; k < 8; k++) color8[8 * pi + k] = P.color[k];
      if (weights8) for (int k = 0; k < 8; k++) weights8[8 * pi + k] = P.weights[k];
      if (hasDepthPrior) hasDepthPrior[pi] = P.prior;
      for (int k = 0; k < P.nres; k++, ri++) {
        if (res_point) res_point[ri] = pi;
        if (res_target) res_target[ri] = P.target[k];
      }
      pi++;
    }
  return 0;
}

}  // extern "C"
