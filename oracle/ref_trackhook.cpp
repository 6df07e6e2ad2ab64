// ORACLE-SIDE TEST INFRASTRUCTURE (part of oracle/_ref/libref.so).  FullSystem.cpp — and only it — is compiled with -DtrackNewestCoarse=trackNewestCoarse_pinhook
// (oracle/Makefile.ref), so the try loop of FullSystem::trackNewCoarse (FullSystem.cpp:419-489), unmodified, calls the member defined here instead of
// CoarseTracker::trackNewestCoarse.  By default the call is handed straight on to the real member (compiled, unrenamed, in CoarseTracker.o).  In ENUMERATE mode it records the
// initial guess of every try and reports "tracking failed", which makes the reference walk through its whole hypothesis list (lastF_2_fh_tries, :364-402): the list the
// oracle's and the HIP library's make_track_hypotheses must reproduce bit for bit (tests/test_ref_pin_cpu.py).
#define trackNewestCoarse trackNewestCoarse_pinhook
#include "FullSystem/CoarseTracker.h"
#undef trackNewestCoarse
#include <vector>

namespace dso { bool ref_real_trackNewestCoarse(CoarseTracker* ct, FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, Vec5 minResForAbort,
                                                IOWrap::Output3DWrapper* wrap); }

static bool g_enumerate = false;
static std::vector<double> g_tries;   // 7 doubles per try: t(3), q xyzw

namespace dso
{
bool CoarseTracker::trackNewestCoarse_pinhook(FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, Vec5 minResForAbort,
                                              IOWrap::Output3DWrapper* wrap)
{
	if(!g_enumerate) return ref_real_trackNewestCoarse(this, newFrameHessian, lastToNew_out, aff_g2l_out, coarsestLvl, minResForAbort, wrap);
	for(int i = 0; i < 3; i++) g_tries.push_back(lastToNew_out.translation()[i]);
	const Eigen::Quaterniond& q = lastToNew_out.unit_quaternion();
	g_tries.push_back(q.x()); g_tries.push_back(q.y()); g_tries.push_back(q.z()); g_tries.push_back(q.w());
	lastResiduals.setConstant(NAN);
	return false;
}
}

extern "C" {
void ref_enumerate_tries(int on) { g_enumerate = on != 0; g_tries.clear(); }
int ref_enumerated_tries(double* pose7_out, int max_tries)
{
	const int n = (int)(g_tries.size() / 7);
	for(int i = 0; i < n && i < max_tries; i++) for(int k = 0; k < 7; k++) pose7_out[7 * i + k] = g_tries[7 * i + k];
	return n;
}
}
