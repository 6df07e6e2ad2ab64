// The adapter of INTEGRATION.md, compiled for real against the reference's own headers: replacement definitions of the five members through which
// FullSystem reaches the photometric hot path —
//     FrameHessian::makeImages              src/dso/FullSystem/HessianBlocks.cpp:128-191
//     CoarseTracker::setCoarseTrackingRef   src/dso/FullSystem/CoarseTracker.cpp:524-538 (+ makeCoarseDepthL0 :138-295)
//     CoarseTracker::trackNewestCoarse      src/dso/FullSystem/CoarseTracker.cpp:539-770
//     FullSystem::traceNewCoarse            src/dso/FullSystem/FullSystem.cpp:541-584
//     FullSystem::optimize                  src/dso/FullSystem/FullSystemOptimize.cpp:417-647
//     FullSystem::activatePointsMT_Reductor src/dso/FullSystem/FullSystem.cpp:589-601 (-> FullSystem::optimizeImmaturePoint, FullSystemOptPoint.cpp:51-205, for a range of candidates)
//     CoarseInitializer::calcResAndGS       src/dso/FullSystem/CoarseInitializer.cpp:331-624   (optional: dropin_set_initializer)
//     EnergyFunctional::insertFrame / insertPoint / insertResidual / dropResidual / removePoint / marginalizeFrame
//                                           src/dso/OptimizationBackend/EnergyFunctional.cpp:435-518, 641-646, 766-782 (unchanged + one forwarded call each: the window graph stays resident)
//     EnergyFunctional::marginalizePointsF  src/dso/OptimizationBackend/EnergyFunctional.cpp:678-742 (its accumulation on the device, from the window optimize left there)
// — each forwarding to the C ABI of include/dmvio_hip.h (libdmvio_hip.so) and writing the results back into the reference's pointer graph, so that the rest of
// FullSystem (initialiser, pixel selector, activation, marginalisation policy, keyframe management: all unmodified reference code) runs on unchanged.
//
// HOW IT GETS INTO THE REFERENCE WITHOUT EDITING IT (test infrastructure; a maintainer would simply replace the five definitions): this file is built into
// oracle/_ref/libdropin_hip.so, which is loaded BEFORE oracle/_ref/libref.so (the reference's sources compiled unmodified by oracle/Makefile.ref).  libref.so calls
// these members through its PLT (default visibility, -fPIC, not -Bsymbolic — checked by tests/test_dropin_cpu.py), so the dynamic linker binds the calls inside the
// reference's FullSystem to the definitions below: ELF symbol interposition, the reference's object files stay byte for byte what Makefile.ref produced.  The
// originals remain reachable through dlsym(RTLD_NEXT, mangled name); with the adapter switched off (dropin_enable(0, ...)) every member forwards to its original,
// which is how the all-CPU run of the comparison is made with the same binary.
//
// What stays on the CPU even with the adapter on: FrameHessian::makeImages ALSO runs the reference's own pyramid build, because the initialiser, the pixel selector,
// ImmaturePoint's constructor, optimizeImmaturePoint and flagPointsForRemoval's relinearisation — all left to the reference here — read FrameHessian::dIp.
// Only tests/ and bench.py's drop_in leg use this file; the product (libdmvio_hip.so) never links the reference.
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <complex>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <atomic>
#include <map>
#include <unordered_map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

// like a maintainer's edit, the replacement members live inside the classes: private members are theirs to use
#define private public
#define protected public
#include "util/NumType.h"
#include "util/settings.h"
#include "util/globalCalib.h"
#include "util/FrameShell.h"
#include "OptimizationBackend/EnergyFunctional.h"
#include "OptimizationBackend/EnergyFunctionalStructs.h"
#include "FullSystem/FullSystem.h"
#include "FullSystem/HessianBlocks.h"
#include "FullSystem/Residuals.h"
#include "FullSystem/ImmaturePoint.h"
#include "FullSystem/CoarseTracker.h"
#include "FullSystem/CoarseInitializer.h"
#include "util/TimeMeasurement.h"
#undef private
#undef protected

#include "dmvio_hip.h"

using namespace dso;

namespace
{
enum { N_STATS = 7 };
struct Stats { double seconds[N_STATS] = {0, 0, 0, 0, 0, 0, 0}; long calls[N_STATS] = {0, 0, 0, 0, 0, 0, 0}; };   // makeImages, setCoarseTrackingRef, trackNewestCoarse, traceNewCoarse, optimize, activatePointsMT_Reductor, calcResAndGS
struct Timer
{
	Stats& s; int k; std::chrono::steady_clock::time_point t0;
	Timer(Stats& s_, int k_) : s(s_), k(k_), t0(std::chrono::steady_clock::now()) {}
	~Timer();
};

struct Backend
{
	bool on = false;
	int accumulators = 0;          // 0 = the library's default
	dmvio_hip_ctx* ctx = nullptr;
	dmvio_hip_ba* ba = nullptr;
	dmvio_hip_immature* imm = nullptr;
	int n_slots = 0;
	std::recursive_mutex mu;       // the adapter's own containers (real-time mode: makeImages / trackNewestCoarse on the tracking thread, everything else on the mapping thread)
	std::map<const FrameHessian*, int> slotOf;
	std::map<int, long> slotAge;   // slot -> value of slotClock when it was last handed out
	std::map<int, int> slotShellId;   // slot -> FrameShell::id of the frame it was handed to
	std::atomic<int> lastMappedId{-1};   // FrameShell::id of the last frame makeKeyFrame / makeNonKeyFrame finished: younger frames are queued, in flight, or not yet seen by the mapper
	long slotClock = 0;
	std::atomic<const FrameHessian*> mappingFrame{nullptr};   // the frame makeKeyFrame / makeNonKeyFrame is working on: taken from the mapper's queue, not (yet) in the window
	std::map<const CoarseTracker*, dmvio_hip_tracker*> trackerOf;
	FullSystem* fs = nullptr;      // learnt from the first FullSystem member that comes by
	std::deque<std::pair<const FrameHessian*, std::vector<float>>> pendingImages;   // frames whose host pyramids were not built (see FrameHessian::makeImages below)
	bool imm_valid = false;        // the immature handle holds exactly imm_pts (of the keyframes imm_hostIDs) with the state the objects have
	std::vector<ImmaturePoint*> imm_pts;
	std::vector<int> imm_hostIDs;
	double opt_split[3] = {0, 0, 0};   // FullSystem::optimize member: flatten + upload, dmvio_hip_ba_optimize, write-back (seconds)
	// the window graph kept resident across keyframes (dmvio_hip_graph, include/dmvio_hip.h): the EnergyFunctional mutators below forward every change with the indices the
	// reference holds itself, FullSystem::optimize hands the mirror over instead of flattening the pointer graph.  resident: 0 = flatten every keyframe (rounds 1-3),
	// 1 = resident, 2 = resident AND flattened, the two compared element for element (tests)
	dmvio_hip_graph* graph = nullptr;
	const EnergyFunctional* graph_ef = nullptr;   // the EnergyFunctional the mirror follows
	bool graph_valid = false;
	int resident = 1;
	int real_marg = 1;                  // EnergyFunctional::marginalizePointsF: 1 = its accumulation on the device from the window optimize left there (resident mode, outside shadow mode), 0 = the reference's own
	bool window_current = false;        // the BA handle holds the window of THIS keyframe's optimize, untouched since
	long n_real_marg = 0, n_real_marg_pts = 0, n_real_marg_disagree = 0;
	long graph_ops = 0, graph_resyncs = 0, graph_verified = 0, graph_mismatch = 0;
	double wb_split[5] = {0, 0, 0, 0, 0};   // write-back of optimize: calibration + keyframe states + adjoints / precalc, the downloads, the per-point pass, the removals, the tail
	double up_split[6] = {0, 0, 0, 0, 0, 0};   // uploadWindow: frame tables, graph walk (index / flatten / verify), set_window, set_graph(_from), frame states + thresholds + calibration, marginalisation prior
	std::unordered_map<const PointHessian*, int> windowPoint;   // point -> index in the window the BA handle holds (set by the last optimize; shadow mode and resident mode 0)
	std::vector<PointHessian*> windowPoints;   // resident mode: the points of that window by index; the index itself is kept in PointHessian::idx (a member the reference declares, HessianBlocks.h:
	                                           // PointHessian, and never uses), written by optimize's write-back, which touches every point anyway
	std::unordered_map<const PointHessian*, float> deviceHessian;   // shadow mode: idepth_hessian the DEVICE's optimize of this keyframe left behind
	// CoarseInitializer::calcResAndGS: 0 = the reference's own (its point loop always runs on NUM_THREADS workers that take 50-point chunks as they come,
	// CoarseInitializer.cpp:507 / util/IndexThreadReduce.h:83-87 — the sums, and with them everything downstream, vary in the last bits from run to run);
	// 1 = the oracle's single-threaded restatement (oracle/init_oracle.cpp, per-point bit-identical to the reference: the sums one worker taking every chunk would form)
	//     — a DETERMINISTIC all-CPU run to compare against; 2 = libdmvio_hip.so (dmvio_hip_initializer_calc_res_and_gs)
	int init_mode = 0;
	// shadow: the reference's own members stay in charge of the pipeline (so the two sides never drift apart through a flipped discrete decision); every call is ALSO run on
	// libdmvio_hip.so from the same inputs and the two answers are compared — per-call parity on the live windows / frames of a run of the reference's FullSystem
	bool shadow = false;
	struct Shadow {
		long n_opt = 0, n_track = 0, n_trace_pts = 0, n_trace_diff = 0, n_track_good_diff = 0, n_resInA_diff = 0, n_opt_iter_diff = 0, n_act_pts = 0, n_act_diff = 0;
		// marginalisation: n_marg_decision_diff = marginalise-or-drop decisions (flagPointsForRemoval, FullSystem.cpp:846: idepth_hessian of the LAST solveSystemF of this
		// keyframe's optimize > setting_minIdepthH_marg) that the device's own optimize would have taken differently; n_marg_reacc_diff = the same rule on a Hessian
		// RE-accumulated at the post-optimisation state (what dmvio_hip_ba_marginalize_points sees in this shadow set-up: another linearisation point, informational)
		long n_marg = 0, n_marg_pts = 0, n_marg_decision_diff = 0, n_marg_reacc_diff = 0, n_marg_res_diff = 0, n_marg_unknown = 0;
		double marg_H_rel = 0, marg_b_rel = 0, opt_hessian_rel = 0;
		long solve_calls = 0;   // EnergyFunctional::solveSystemF calls of the reference's own optimize = its Gauss-Newton iterations
		double opt_rmse_rel = 0, opt_energy_rel = 0, opt_pose = 0, opt_aff = 0, opt_idepth_med = 0, track_pose = 0, track_aff_a = 0, track_aff_b = 0, track_res_rel = 0;
	} sh;
	dmvio_hip_initializer* ini = nullptr;
	void* oracle_lib = nullptr;
	// FullSystem::trackNewCoarse: 1 = the member below hands the whole try loop (FullSystem.cpp:364-489) to dmvio_hip_tracker_track_new_coarse — try 0 alone, the remaining
	// motion hypotheses as ONE device batch, the sequential abort / winner rule replayed on their results; 0 = the reference's own loop, one trackNewestCoarse per try
	int batch_tries = 1;
	long n_tnc = 0, n_tnc_batched = 0, n_tnc_past_try0 = 0, n_tnc_tries = 0;
	long tnc_inner_calls = 0;   // trackNewestCoarse calls made by the reference's own loop during the current trackNewCoarse (shadow: the number of tries it walked)
	struct ShadowTnc { long n = 0, n_past_try0 = 0, n_tries_diff = 0, n_good_diff = 0; double pose = 0, aff_a = 0, res_rel = 0; long max_tries = 0; } tnc;
	// the reference's DEFAULT configuration (setting_useIMU / setting_useGTSAMIntegration): calls that went through dmvio_hip_tracker_track_vio with computeCoarseUpdate
	// behind it, through it with the visual-only step (IMU not yet coarse-initialised), through dmvio_hip_ba_optimize_vio; hook calls made from the callbacks
	long n_vio_track = 0, n_vio_track_visual = 0, n_vio_opt = 0, n_vio_hook_calls = 0;
	// shadow mode and a stateful IMU / GTSAM facade: both sides of a shadowed call go through the same facade object, whose state is saved before the device's call and put
	// back before the reference's own (the test's stand-in registers how: dropin_set_vio_state_hooks)
	void* vio_sys = nullptr;
	void* (*vio_save)(void*) = nullptr;
	void (*vio_restore)(void*, void*, int) = nullptr;
	Stats stats;
	char error[512] = "";
	long failures = 0;
};
Backend g;
Timer::~Timer() { std::lock_guard<std::recursive_mutex> lk(g.mu); s.seconds[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); s.calls[k]++; }

void fail(const char* what)
{
	snprintf(g.error, sizeof(g.error), "%s: %s", what, dmvio_hip_last_error());
	g.failures++;
	fprintf(stderr, "[dropin] %s\n", g.error);
}
#define HIP_OK(call) ((call) >= 0 ? true : (fail(#call), false))

template <class Fn> Fn original(const char* mangled)
{
	void* p = dlsym(RTLD_NEXT, mangled);
	if (!p) { fprintf(stderr, "[dropin] the reference's own %s is not loaded behind the adapter\n", mangled); abort(); }
	return (Fn)p;
}

void toPose7(const SE3& T, double* p)
{
	p[0] = T.translation()[0]; p[1] = T.translation()[1]; p[2] = T.translation()[2];
	const Eigen::Quaterniond& q = T.unit_quaternion();
	p[3] = q.x(); p[4] = q.y(); p[5] = q.z(); p[6] = q.w();
}
SE3 fromPose7(const double* p)
{
	Eigen::Quaterniond q(p[6], p[3], p[4], p[5]);
	return SE3(q, Vec3(p[0], p[1], p[2]));
}

// frame slots of the HIP context: a frame keeps its slot while the reference can still refer to it (window keyframes, the trackers' reference frames, the frame being
// added); slots of frames the reference has deleted are handed out again
int acquireSlot(const FrameHessian* fh)
{
	for (int attempt = 0;; attempt++)
	{
		bool needGC = false;
		{
			std::lock_guard<std::recursive_mutex> lk(g.mu);
			auto it = g.slotOf.find(fh);
			if (it != g.slotOf.end()) { g.slotAge[it->second] = ++g.slotClock; g.slotShellId[it->second] = fh->shell ? fh->shell->id : -1; return it->second; }      // a new frame at the address of a deleted one: its slot is rebuilt by the upload that follows
			needGC = ((int)g.slotOf.size() >= g.n_slots / 2 || attempt > 0) && g.fs;
		}
		// the frames the reference can still refer to, read under ITS locks (the mapping thread changes these containers in real-time mode) and before the adapter's own lock is
		// taken again: the mapping thread calls slotFor() while it holds mapMutex
		std::set<const FrameHessian*> live;
		if (needGC)
		{
			{
				boost::unique_lock<boost::mutex> lock(g.fs->mapMutex);
				live.insert(g.fs->frameHessians.begin(), g.fs->frameHessians.end());
				if (g.fs->coarseTracker) live.insert(g.fs->coarseTracker->lastRef);
				if (g.fs->coarseTracker_forNewKF) live.insert(g.fs->coarseTracker_forNewKF->lastRef);
				if (g.fs->coarseInitializer) { live.insert(g.fs->coarseInitializer->firstFrame); live.insert(g.fs->coarseInitializer->newFrame); }
			}
			{
				boost::unique_lock<boost::mutex> lock(g.fs->trackMapSyncMutex);
				live.insert(g.fs->unmappedTrackedFrames.begin(), g.fs->unmappedTrackedFrames.end());
			}
		}
		{
			std::lock_guard<std::recursive_mutex> lk(g.mu);
			if (needGC)
			{
				// Reclaimed: slots of frames the reference no longer holds.  A frame on its way through the mapper (queued, or taken from the queue and inside makeKeyFrame /
				// makeNonKeyFrame, or between the two) is in none of the containers for a moment: while the system is mapping, only frames the mapper is DONE with
				// (FrameShell::id <= that of the last frame it finished) are candidates
				live.insert(g.mappingFrame.load());
				const int done = g.lastMappedId.load();
				const bool mapping = g.fs->initialized;
				for (auto i = g.slotOf.begin(); i != g.slotOf.end();)
				{
					const bool dead = !live.count(i->first) && g.slotClock - g.slotAge[i->second] > 4 && (!mapping || g.slotShellId[i->second] <= done);
					if (dead) i = g.slotOf.erase(i); else ++i;
				}
			}
			std::set<int> used;
			for (auto& kv : g.slotOf) used.insert(kv.second);
			for (int s = 0; s < g.n_slots; s++) if (!used.count(s)) { g.slotOf[fh] = s; g.slotAge[s] = ++g.slotClock; g.slotShellId[s] = fh->shell ? fh->shell->id : -1; return s; }
		}
		// every slot belongs to a frame the mapper has not finished: the tracking thread is that many frames ahead of it.  A live system is paced by its camera; here the frame
		// buffer is simply full, and the new frame waits for the mapper
		if (attempt > 20000) { fprintf(stderr, "[dropin] out of frame slots\n"); abort(); }
		std::this_thread::sleep_for(std::chrono::microseconds(500));
	}
}
#define slotFor(fh) slotForAt(fh, __LINE__)
int slotForAt(const FrameHessian* fh, int line)
{
	std::lock_guard<std::recursive_mutex> lk(g.mu);
	auto it = g.slotOf.find(fh);
	if (it == g.slotOf.end())
	{
		fprintf(stderr, "[dropin] frame without a slot (makeImages did not come through the adapter): adapter line %d, frame %p shell id %d frameID %d, %zu slots in use, mapping frame %p\n", line,
		        (const void*)fh, fh && fh->shell ? fh->shell->id : -1, fh ? fh->frameID : -1, g.slotOf.size(), (const void*)g.mappingFrame.load());
		abort();
	}
	return it->second;
}
dmvio_hip_tracker* trackerFor(const CoarseTracker* ct)
{
	std::lock_guard<std::recursive_mutex> lk(g.mu);
	auto it = g.trackerOf.find(ct);
	if (it != g.trackerOf.end()) return it->second;
	dmvio_hip_tracker* t = dmvio_hip_tracker_create(g.ctx);
	if (!t) { fail("dmvio_hip_tracker_create"); abort(); }
	dmvio_hip_tracker_settings st;
	st.huberTH = setting_huberTH; st.coarseCutoffTH = setting_coarseCutoffTH; st.affineOptModeA = setting_affineOptModeA; st.affineOptModeB = setting_affineOptModeB;
	dmvio_hip_tracker_set_settings(t, &st);
	g.trackerOf[ct] = t;
	return t;
}

// ---- the reference's DEFAULT branch, tracking side: the three IMUIntegration members CoarseTracker::trackNewestCoarse calls (CoarseTracker.cpp:620, 709, 766) as the callbacks
// of dmvio_hip_tracker_track_vio.  `user` is the CoarseTracker, whose imuIntegration reference is the FullSystem's own object — exactly what the replaced loop talks to.
int coarseUpdateThunk(void* user, const double H[64], const double b[8], float extrapFac, float lambda, const double* /*pose7_cur*/, const double* /*aff_cur*/, double pose7_new[7],
                      double* incA, double* incB, double* incNorm)
{
	CoarseTracker* t = (CoarseTracker*)user;
	Mat88 Hm; Vec8 bv;
	for (int r = 0; r < 8; r++) { for (int c = 0; c < 8; c++) Hm(r, c) = H[8 * r + c]; bv[r] = b[r]; }
	g.n_vio_hook_calls++;
	const SE3 refToNew_new = t->imuIntegration.computeCoarseUpdate(Hm, bv, extrapFac, lambda, *incA, *incB, *incNorm);   // note: H, not Hl — the damping happens inside (:618-620)
	toPose7(refToNew_new, pose7_new);
	return 0;
}
void coarseAcceptThunk(void* user) { g.n_vio_hook_calls++; ((CoarseTracker*)user)->imuIntegration.acceptCoarseUpdate(); }
void coarseVisualThunk(void* user, const double H[64], const double b[8], int trackingGood)
{
	Mat88 Hm; Vec8 bv;
	for (int r = 0; r < 8; r++) { for (int c = 0; c < 8; c++) Hm(r, c) = H[8 * r + c]; bv[r] = b[r]; }
	g.n_vio_hook_calls++;
	((CoarseTracker*)user)->imuIntegration.addVisualToCoarseGraph(Hm, bv, trackingGood != 0);
}
// what trackNewestCoarse does when setting_useIMU is set: the LM loop of dmvio_hip_tracker_track_vio with computeCoarseUpdate in the place of the LDL^T step once the IMU is
// coarse-initialised (:612), acceptCoarseUpdate after every accepted step and addVisualToCoarseGraph at the end whether it is or not (:708, :765: both only test setting_useIMU)
bool trackVio(CoarseTracker* t, dmvio_hip_tracker* trk, int slot, float exposure, double pose7[7], double aff[2], int coarsestLvl, const double minRes[5], double lastRes[5], double flow[3],
              double H[64], double b[8], int* good)
{
	dmvio_hip_coarse_callbacks cb;
	memset(&cb, 0, sizeof(cb));
	cb.user = t;
	const bool initialised = t->imuIntegration.isCoarseInitialized();
	cb.update = initialised ? coarseUpdateThunk : nullptr;   // NULL: the library's visual-only step = the reference's else-branch (:639-682)
	cb.accept = coarseAcceptThunk;
	cb.visual = coarseVisualThunk;
	int n_evals = 0;
	if (initialised) g.n_vio_track++; else g.n_vio_track_visual++;
	return HIP_OK(dmvio_hip_tracker_track_vio(trk, slot, exposure, pose7, aff, coarsestLvl, minRes, &cb, lastRes, flow, H, b, good, &n_evals));
}

// ---- the reference's DEFAULT branch, mapping side: the seven BAGTSAMIntegration members EnergyFunctional::solveSystemF / calcMEnergyF and FullSystem::optimize call
// (EnergyFunctional.cpp:335-341, 958-969; FullSystemOptimize.cpp:491-503, 523, 534-538, 569-572, 594, 641) as the callbacks of dmvio_hip_ba_optimize_vio.  The real class reads the
// keyframes through std::vector<EFFrame*> (EFFrame::data->get_state(), PRE_worldToCam, shell->id; BAGTSAMIntegration.cpp:97-120) and the calibration through CalibHessian: the
// frame views of a callback are written into those objects first, so the hook sees what it would see inside the reference's own loop at the same point.
void writeFrameViews(FullSystem* fs, int F, const dmvio_hip_ba_frame_view* fr, const double calib_value[4])
{
	VecC v; for (int i = 0; i < 4; i++) v[i] = calib_value[i];
	fs->Hcalib.setValue(v);
	for (int f = 0; f < F; f++)
	{
		Vec10 st; for (int i = 0; i < 10; i++) st[i] = fr[f].state10[i];
		fs->frameHessians[fr[f].index]->setState(st);
	}
}
int baComputeThunk(void* user, int n, const double* HPassed, const double* b, double lambda, const double* HNoLambda, int F, const dmvio_hip_ba_frame_view* frames, const double calib_value[4],
                   double* x_out)
{
	FullSystem* fs = (FullSystem*)user;
	writeFrameViews(fs, F, frames, calib_value);
	MatXX H(n, n), HN(n, n); VecX bv(n);
	for (int r = 0; r < n; r++) { bv[r] = b[r]; for (int c = 0; c < n; c++) { H(r, c) = HPassed[(size_t)r * n + c]; HN(r, c) = HNoLambda[(size_t)r * n + c]; } }
	g.n_vio_hook_calls++;
	const VecX x = fs->baIntegration->computeBAUpdate(H, bv, lambda, fs->ef->frames, HN);
	if (x.size() != n) return -1;
	for (int i = 0; i < n; i++) x_out[i] = x[i];
	return 0;
}
void baAcceptThunk(void* user, double energy) { g.n_vio_hook_calls++; ((FullSystem*)user)->baIntegration->acceptBAUpdate(energy); }
double baEnergyThunk(void* user, int useNewValues) { g.n_vio_hook_calls++; return ((FullSystem*)user)->baIntegration->getBAEnergy(useNewValues != 0); }
void baValuesThunk(void* user, int F, const dmvio_hip_ba_frame_view* frames, const double calib_value[4])
{
	FullSystem* fs = (FullSystem*)user;
	writeFrameViews(fs, F, frames, calib_value);
	g.n_vio_hook_calls++;
	fs->baIntegration->updateBAValues(fs->ef->frames);
}
double baWeightThunk(void* user, double energy, double rmse, int good) { g.n_vio_hook_calls++; return ((FullSystem*)user)->baIntegration->updateDynamicWeight(energy, rmse, good != 0); }
int baBreakThunk(void* user) { g.n_vio_hook_calls++; return ((FullSystem*)user)->baIntegration->canBreak() ? 1 : 0; }
void baPostThunk(void* user, int F, const dmvio_hip_ba_frame_view* frames, const double calib_value[4])
{
	FullSystem* fs = (FullSystem*)user;
	writeFrameViews(fs, F, frames, calib_value);
	g.n_vio_hook_calls++;
	fs->baIntegration->postOptimization(fs->ef->frames);
}
// FullSystem::optimize's loop on its default branch
bool optimizeVio(FullSystem* fs, dmvio_hip_ba* ba, int mnumOptIts, float* rmse, double* finalEnergy, int* iterations)
{
	dmvio_hip_ba_callbacks cb;
	memset(&cb, 0, sizeof(cb));
	cb.user = fs;
	cb.computeBAUpdate = baComputeThunk; cb.acceptBAUpdate = baAcceptThunk; cb.getBAEnergy = baEnergyThunk; cb.updateBAValues = baValuesThunk;
	cb.updateDynamicWeight = baWeightThunk; cb.canBreak = baBreakThunk; cb.postOptimization = baPostThunk;
	EnergyFunctional* ef = fs->ef;
	const int n = CPARS + 8 * ef->nFrames;
	std::vector<double> HMg((size_t)n * n, 0.0), bMg(n, 0.0);
	const bool havePrior = ef->HMForGTSAM.rows() == n && ef->bMForGTSAM.size() == n;
	if (havePrior) for (int r = 0; r < n; r++) { bMg[r] = ef->bMForGTSAM[r]; for (int c = 0; c < n; c++) HMg[(size_t)r * n + c] = ef->HMForGTSAM(r, c); }
	dmvio_hip_ba_vio_options opt;
	memset(&opt, 0, sizeof(opt));
	opt.coarseTrackingWasGood = fs->frameHessians.back()->shell->trackingWasGood ? 1 : 0;
	opt.updateDynamicWeightDuringOptimization = fs->imuIntegration.getImuSettings().updateDynamicWeightDuringOptimization ? 1 : 0;
	opt.minOptIterations = setting_minOptIterations;
	opt.resInA_at_entry = ef->resInA;
	opt.HMForGTSAM = havePrior ? HMg.data() : nullptr; opt.bMForGTSAM = havePrior ? bMg.data() : nullptr;
	g.n_vio_opt++;
	return HIP_OK(dmvio_hip_ba_optimize_vio(ba, mnumOptIts, &cb, &opt, rmse, finalEnergy, iterations, nullptr));
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------------------------
extern "C" {
// on != 0: forward the five members to libdmvio_hip.so (context of w x h frames on `device`; accumulators: dmvio_hip_ba_set_accumulators, 0 = default);
// on == 0: every member calls the reference's own definition.  Returns 0, or -1 when the HIP context cannot be created.
int dropin_enable(int on, int device, int w, int h, int accumulators)
{
	if (g.ctx)
	{
		for (auto& kv : g.trackerOf) dmvio_hip_tracker_destroy(kv.second);
		g.trackerOf.clear();
		if (g.ba) dmvio_hip_ba_destroy(g.ba);
		if (g.imm) dmvio_hip_immature_destroy(g.imm);
		if (g.ini) dmvio_hip_initializer_destroy(g.ini);
		dmvio_hip_destroy(g.ctx);
		g.ba = nullptr; g.imm = nullptr; g.ini = nullptr; g.ctx = nullptr; g.init_mode = 0;
	}
	g.slotOf.clear(); g.fs = nullptr; g.on = false; g.stats = Stats(); g.failures = 0; g.error[0] = 0;
	g.n_vio_track = g.n_vio_track_visual = g.n_vio_opt = g.n_vio_hook_calls = 0;
	g.n_tnc = g.n_tnc_batched = g.n_tnc_past_try0 = g.n_tnc_tries = 0; g.tnc = Backend::ShadowTnc();
	if (g.graph) { dmvio_hip_graph_destroy(g.graph); g.graph = nullptr; }
	g.graph_ef = nullptr; g.graph_valid = false; g.graph_ops = g.graph_resyncs = g.graph_verified = g.graph_mismatch = 0;
	if (!on) return 0;
	g.graph = dmvio_hip_graph_create();
	g.n_slots = 96;
	g.ctx = dmvio_hip_create(device, w, h, g.n_slots);
	if (!g.ctx) { fail("dmvio_hip_create"); return -1; }
	g.ba = dmvio_hip_ba_create(g.ctx);
	g.imm = dmvio_hip_immature_create(g.ctx, 1 << 16);
	if (!g.ba || !g.imm) { fail("dmvio_hip_ba_create / dmvio_hip_immature_create"); return -1; }
	g.accumulators = accumulators;
	if (accumulators > 0 && !HIP_OK(dmvio_hip_ba_set_accumulators(g.ba, accumulators))) return -1;
	g.on = true;
	return 0;
}
// CoarseInitializer::calcResAndGS: 0 the reference's own (multi-threaded, run-to-run noise), 1 oracle/_build/liboracle.so's single-threaded restatement (path given),
// 2 libdmvio_hip.so (needs dropin_enable(1, ...) first)
int dropin_set_initializer(int mode, const char* liboracle_path)
{
	if (mode == 1)
	{
		if (!g.oracle_lib) g.oracle_lib = dlopen(liboracle_path, RTLD_NOW | RTLD_LOCAL);
		if (!g.oracle_lib || !dlsym(g.oracle_lib, "orc_init_calc_res_and_gs")) { fprintf(stderr, "[dropin] cannot load %s\n", liboracle_path ? liboracle_path : "(null)"); return -1; }
	}
	if (mode == 2)
	{
		if (!g.on) return -1;
		if (!g.ini) g.ini = dmvio_hip_initializer_create(g.ctx, 1 << 17);
		if (!g.ini) { fail("dmvio_hip_initializer_create"); return -1; }
	}
	g.init_mode = mode;
	return 0;
}
// shadow mode (needs dropin_enable(1, ...)): the reference's own members run the pipeline, the HIP library runs every call beside them, deviations are recorded
void dropin_set_shadow(int on) { g.shadow = on != 0; g.sh = Backend::Shadow(); }
// out[16]: n_opt, n_track, n_trace_pts, n_trace_diff, n_track_good_diff, n_resInA_diff (windows whose EnergyFunctional::resInA differs), max over calls of: optimize rmse (relative), final energy (relative), keyframe
// translation (m), affine a|b (scaled units), median relative idepth difference; trackNewestCoarse translation (m), affine a, affine b, lastResiduals[0] (relative)
void dropin_get_shadow(double* out)
{
	const Backend::Shadow& h = g.sh;
	const double v[15] = {(double)h.n_opt, (double)h.n_track, (double)h.n_trace_pts, (double)h.n_trace_diff, (double)h.n_track_good_diff, (double)h.n_resInA_diff, h.opt_rmse_rel,
	                      h.opt_energy_rel, h.opt_pose, h.opt_aff, h.opt_idepth_med, h.track_pose, h.track_aff_a, h.track_aff_b, h.track_res_rel};
	for (int i = 0; i < 15; i++) out[i] = v[i];
	out[15] = (double)h.n_opt_iter_diff;   // windows in which the device ran another number of Gauss-Newton iterations than the reference's own optimize
}
// shadowed EnergyFunctional::marginalizePointsF calls: n_calls, n_points, points the device's optimize would have DROPPED instead (its own idepth_hessian against
// setting_minIdepthH_marg), calls whose residual count differs; max relative deviation of the increment of HM / of bM
void dropin_get_shadow_marginalization(double* out6)
{
	out6[0] = (double)g.sh.n_marg; out6[1] = (double)g.sh.n_marg_pts; out6[2] = (double)g.sh.n_marg_decision_diff; out6[3] = (double)g.sh.n_marg_res_diff;
	out6[4] = g.sh.marg_H_rel; out6[5] = g.sh.marg_b_rel;
}
// more of the same: decisions that differ on the RE-accumulated Hessian (informational), marginalised points the device's optimize never saw, max relative difference of
// PointHessian::idepth_hessian after optimize (device vs reference) over all windows
void dropin_get_shadow_marginalization2(double* out3)
{
	out3[0] = (double)g.sh.n_marg_reacc_diff; out3[1] = (double)g.sh.n_marg_unknown; out3[2] = g.sh.opt_hessian_rel;
}
// n_candidates, n_differing of the shadowed FullSystem::optimizeImmaturePoint calls (result class, idepth bits, the targets of the residuals created)
void dropin_get_shadow_activation(long* out2) { out2[0] = g.sh.n_act_pts; out2[1] = g.sh.n_act_diff;
}
// how FullSystem::optimize hands the window over: 0 = the pointer graph flattened every keyframe, 1 (default) = the resident window graph, 2 = both, compared
void dropin_set_resident(int mode) { g.resident = mode; }
// forwarded EnergyFunctional mutations, keyframes at which the mirror had to be rebuilt from the pointer graph (0 in a run that went through the adapter from its first
// frame), keyframes verified against the flattened pointer graph (mode 2), keyframes at which the two differed
void dropin_get_writeback_split(double* out5) { for (int k = 0; k < 5; k++) out5[k] = g.wb_split[k]; }
void dropin_set_real_marginalization(int on) { g.real_marg = on; }
void dropin_get_real_marginalization(long* out2) { out2[0] = g.n_real_marg; out2[1] = g.n_real_marg_pts; }
void dropin_get_upload_split(double* out6) { for (int k = 0; k < 6; k++) out6[k] = g.up_split[k]; }
void dropin_get_resident(long* out4) { out4[0] = g.graph_ops; out4[1] = g.graph_resyncs; out4[2] = g.graph_verified; out4[3] = g.graph_mismatch; }
int dropin_is_on() { return g.on ? 1 : 0; }
// FullSystem::trackNewCoarse: hand the try loop to dmvio_hip_tracker_track_new_coarse (1, default) or leave the reference's own loop in place (0)
void dropin_set_batch_tries(int on) { g.batch_tries = on; }
// trackNewCoarse calls seen, calls served by the batched try loop, calls among those whose walk went past try 0, tries walked in total;
// shadow mode: calls compared, calls past try 0 (reference), calls whose number of tries differs, whose good-verdict differs, most tries walked in one call;
// max deviation of the winning pose (translation, m), of aff a, of the achieved level-0 residual (relative)
void dropin_get_track_new_coarse(double* out12)
{
	const double v[12] = {(double)g.n_tnc, (double)g.n_tnc_batched, (double)g.n_tnc_past_try0, (double)g.n_tnc_tries, (double)g.tnc.n, (double)g.tnc.n_past_try0,
	                      (double)g.tnc.n_tries_diff, (double)g.tnc.n_good_diff, (double)g.tnc.max_tries, g.tnc.pose, g.tnc.aff_a, g.tnc.res_rel};
	for (int i = 0; i < 12; i++) out12[i] = v[i];
}
// the default (VIO) configuration: how the state of a stateful IMU / GTSAM facade is saved and restored around a shadowed call (sys = the argument of both functions)
void dropin_set_vio_state_hooks(void* sys, void* (*save)(void*), void (*restore)(void*, void*, int)) { g.vio_sys = sys; g.vio_save = save; g.vio_restore = restore; }
// trackNewestCoarse calls served by dmvio_hip_tracker_track_vio with computeCoarseUpdate as the step / with the visual-only step (IMU not coarse-initialised yet), optimize
// calls served by dmvio_hip_ba_optimize_vio, IMUIntegration / BAGTSAMIntegration members called from the library's callbacks
void dropin_get_vio(long* out4) { out4[0] = g.n_vio_track; out4[1] = g.n_vio_track_visual; out4[2] = g.n_vio_opt; out4[3] = g.n_vio_hook_calls; }
// the FullSystem whose frames the slots belong to (lets the adapter see which frames are still alive before the first optimize / traceNewCoarse call comes by)
void dropin_attach(void* fullSystem) { g.fs = (FullSystem*)fullSystem; }
// seconds[6], calls[6] in the order makeImages, setCoarseTrackingRef, trackNewestCoarse, traceNewCoarse, optimize, activatePointsMT_Reductor — counted in both modes
void dropin_get_optimize_split(double* out3) { for (int k = 0; k < 3; k++) out3[k] = g.opt_split[k]; }
void dropin_get_stats(double* seconds7, long* calls7) { for (int k = 0; k < N_STATS; k++) { seconds7[k] = g.stats.seconds[k]; calls7[k] = g.stats.calls[k]; } }
void dropin_reset_stats() { g.stats = Stats(); }
long dropin_failures(char* msg, int cap) { if (msg && cap > 0) { strncpy(msg, g.error, cap - 1); msg[cap - 1] = 0; } return g.failures; }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
namespace dso
{

// ---- FrameHessian::makeImages (HessianBlocks.cpp:128-191)
void FrameHessian::makeImages(float* color, CalibHessian* HCalib)
{
	typedef void (*Fn)(FrameHessian*, float*, CalibHessian*);
	static Fn orig = original<Fn>("_ZN3dso12FrameHessian10makeImagesEPfPNS_12CalibHessianE");
	Timer tm(g.stats, 0);
	// The reference's own pyramids (dIp, absSquaredGrad) are read by the parts of the pipeline that stay on the CPU: the initialiser's driver, and — for KEYFRAMES only — the pixel
	// selector, the ImmaturePoint constructor, the relinearisation before marginalisation.  While the system is tracking, a new frame therefore only goes to the device here;
	// its host pyramids are built when (and if) FullSystem::makeKeyFrame is entered for it (below), from a copy of the image kept until then.  Shadow mode builds them always
	// (the reference's own members read them for every frame).
	const bool lazy = g.on && !g.shadow && g.fs && g.fs->initialized && !getenv("DROPIN_EAGER_PYRAMIDS");
	if (!lazy) orig(this, color, HCalib);
	if (!g.on) return;
	if (lazy)
	{
		for (int i = 0; i < PYR_LEVELS; i++) { dIp[i] = nullptr; absSquaredGrad[i] = nullptr; }   // the constructor leaves them unset; the destructor delete[]s them
		dI = nullptr;
		std::lock_guard<std::recursive_mutex> lk(g.mu);
		if (g.pendingImages.size() >= 16) g.pendingImages.pop_front();
		g.pendingImages.emplace_back(this, std::vector<float>(color, color + (size_t)wG[0] * hG[0]));
	}
	const int slot = acquireSlot(this);
	HIP_OK(dmvio_hip_frame_upload(g.ctx, slot, color));
}

// ---- FullSystem::makeKeyFrame (FullSystem.cpp:1228-1390): unchanged, but the frame's host pyramids are built first if makeImages skipped them (see there)
void FullSystem::makeKeyFrame(FrameHessian* fh)
{
	typedef void (*Fn)(FullSystem*, FrameHessian*);
	static Fn orig = original<Fn>("_ZN3dso10FullSystem12makeKeyFrameEPNS_12FrameHessianE");
	struct InMapping { int id; InMapping(const FrameHessian* f) : id(f->shell ? f->shell->id : -1) { g.mappingFrame = f; } ~InMapping() { g.lastMappedId = std::max(g.lastMappedId.load(), id); g.mappingFrame = nullptr; } } inMapping(fh);
	if (g.on && fh->dI == nullptr)
	{
		typedef void (*MI)(FrameHessian*, float*, CalibHessian*);
		static MI makeImagesOrig = original<MI>("_ZN3dso12FrameHessian10makeImagesEPfPNS_12CalibHessianE");
		std::vector<float> image;
		{
			std::lock_guard<std::recursive_mutex> lk(g.mu);
			for (auto it = g.pendingImages.rbegin(); it != g.pendingImages.rend(); ++it)
				if (it->first == fh) { image.swap(it->second); break; }
		}
		if (image.empty()) { fprintf(stderr, "[dropin] makeKeyFrame for a frame whose image is gone\n"); abort(); }
		Timer tm(g.stats, 0);
		makeImagesOrig(fh, image.data(), &Hcalib);
	}
	{
		std::lock_guard<std::recursive_mutex> lk(g.mu);
		for (auto it = g.pendingImages.begin(); it != g.pendingImages.end();) { if (it->first == fh) it = g.pendingImages.erase(it); else ++it; }
	}
	orig(this, fh);
}

// ---- FullSystem::makeNonKeyFrame (FullSystem.cpp:1322-1336): unchanged; the frame is deleted at its end, so its device slot and the copy of its image are given back
void FullSystem::makeNonKeyFrame(FrameHessian* fh)
{
	typedef void (*Fn)(FullSystem*, FrameHessian*);
	static Fn orig = original<Fn>("_ZN3dso10FullSystem15makeNonKeyFrameEPNS_12FrameHessianE");
	g.mappingFrame = fh;
	const int shellId = fh->shell ? fh->shell->id : -1;
	long stamp = -1;
	{
		std::lock_guard<std::recursive_mutex> lk(g.mu);
		auto it = g.slotOf.find(fh);
		if (it != g.slotOf.end()) stamp = g.slotAge[it->second];
		for (auto pi = g.pendingImages.begin(); pi != g.pendingImages.end();) { if (pi->first == fh) pi = g.pendingImages.erase(pi); else ++pi; }   // it will not become a keyframe
	}
	orig(this, fh);      // ... traceNewCoarse(fh); delete fh;
	{
		// the tracking thread may already have built a NEW frame at the address of the deleted one (its upload restamps the slot): only an untouched entry is the dead frame's
		std::lock_guard<std::recursive_mutex> lk(g.mu);
		auto it = g.slotOf.find(fh);
		if (it != g.slotOf.end() && g.slotAge[it->second] == stamp) g.slotOf.erase(it);
	}
	g.lastMappedId = std::max(g.lastMappedId.load(), shellId);
	g.mappingFrame = nullptr;
}

// ---- CoarseInitializer::calcResAndGS (CoarseInitializer.cpp:331-624): the point arrays of the level flattened (struct Pnt, CoarseInitializer.h:44-83), the evaluation on
// the device (or by the oracle's sequential restatement), the per-point results written back into Pnt / JbBuffer_new
Vec3f CoarseInitializer::calcResAndGS(int lvl, Mat88f& H_out, Vec8f& b_out, Mat88f& H_out_sc, Vec8f& b_out_sc, const SE3& refToNew, AffLight refToNew_aff, bool plot)
{
	typedef Vec3f (*Fn)(CoarseInitializer*, int, Mat88f&, Vec8f&, Mat88f&, Vec8f&, const SE3&, AffLight, bool);
	static Fn orig = original<Fn>("_ZN3dso17CoarseInitializer12calcResAndGSEiRN5Eigen6MatrixIfLi8ELi8ELi0ELi8ELi8EEERNS2_IfLi8ELi1ELi0ELi8ELi1EEES4_S6_RKN6Sophus8SE3GroupIdLi0EEENS_8AffLightEb");
	Timer tm(g.stats, 6);
	if (g.init_mode == 0) return orig(this, lvl, H_out, b_out, H_out_sc, b_out_sc, refToNew, refToNew_aff, plot);
	const int n = numPoints[lvl];
	Pnt* pts = points[lvl];
	// scratch that stays with the thread: the initialiser's LM loop calls this a few hundred times per frame
	static thread_local std::vector<float> u, v, iR, idn, en, oth, en_new, maxstep, lastH, Jb;
	static thread_local std::vector<unsigned char> good, good_new;
	u.resize(n); v.resize(n); iR.resize(n); idn.resize(n); en.resize(2 * (size_t)n); oth.resize(n); en_new.resize(2 * (size_t)n); maxstep.resize(n); lastH.resize(n);
	Jb.resize(10 * (size_t)n); good.resize(n); good_new.resize(n);
	for (int i = 0; i < n; i++)
	{
		const Pnt& q = pts[i];
		u[i] = q.u; v[i] = q.v; iR[i] = q.iR; idn[i] = q.idepth_new; en[2 * i] = q.energy[0]; en[2 * i + 1] = q.energy[1]; oth[i] = q.outlierTH; good[i] = q.isGood ? 1 : 0;
		lastH[i] = q.lastHessian_new;
	}
	double Ki9[9], pose7[7], aff[2] = {refToNew_aff.a, refToNew_aff.b};
	for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ki9[3 * r + c] = Ki[lvl](r, c);
	toPose7(refToNew, pose7);
	const float k4[4] = {(float)fx[lvl], (float)fy[lvl], (float)cx[lvl], (float)cy[lvl]};
	float H64[64], b8[8], Hsc64[64], bsc8[8], res3[3];
	if (g.init_mode == 2)
	{
		bool ok = HIP_OK(dmvio_hip_initializer_set_points(g.ini, n, u.data(), v.data(), iR.data(), good.data(), en.data(), oth.data()));
		ok = ok && HIP_OK(dmvio_hip_initializer_calc_res_and_gs(g.ini, lvl, slotFor(firstFrame), slotFor(newFrame), Ki9, k4, pose7, aff, idn.data(), alphaW, alphaK, couplingWeight,
		                                                        setting_weightZeroPriorDSOInitY, setting_weightZeroPriorDSOInitX, H64, b8, Hsc64, bsc8, res3, en_new.data(),
		                                                        good_new.data(), maxstep.data(), lastH.data(), Jb.data()));
		if (!ok) return orig(this, lvl, H_out, b_out, H_out_sc, b_out_sc, refToNew, refToNew_aff, plot);
	}
	else
	{
		typedef void (*Orc)(const float*, const float*, int, int, const double*, float, float, float, float, const double*, double, double, int, const float*, const float*, const float*,
		                    const float*, const unsigned char*, const float*, const float*, float, float, float, double, double, float*, float*, float*, float*, float*, float*,
		                    unsigned char*, float*, float*, float*);
		static Orc orc = (Orc)dlsym(g.oracle_lib, "orc_init_calc_res_and_gs");
		orc((const float*)firstFrame->dIp[lvl], (const float*)newFrame->dIp[lvl], w[lvl], h[lvl], Ki9, k4[0], k4[1], k4[2], k4[3], pose7, aff[0], aff[1], n, u.data(), v.data(), idn.data(),
		    iR.data(), good.data(), en.data(), oth.data(), alphaW, alphaK, couplingWeight, setting_weightZeroPriorDSOInitY, setting_weightZeroPriorDSOInitX, H64, b8, Hsc64, bsc8, res3,
		    en_new.data(), good_new.data(), maxstep.data(), lastH.data(), Jb.data());
	}
	for (int i = 0; i < n; i++)
	{
		Pnt& q = pts[i];
		q.maxstep = maxstep[i]; q.energy_new = Eigen::Vector2f(en_new[2 * i], en_new[2 * i + 1]); q.isGood_new = good_new[i] != 0;
		if (q.isGood_new) q.lastHessian_new = lastH[i];
		if (good[i]) for (int k = 0; k < 10; k++) JbBuffer_new[i][k] = Jb[10 * (size_t)i + k];
	}
	for (int r = 0; r < 8; r++) { for (int c = 0; c < 8; c++) { H_out(r, c) = H64[8 * r + c]; H_out_sc(r, c) = Hsc64[8 * r + c]; } b_out[r] = b8[r]; b_out_sc[r] = bsc8[r]; }
	return Vec3f(res3[0], res3[1], res3[2]);
}

// ---- CoarseTracker::setCoarseTrackingRef (CoarseTracker.cpp:524-538) with makeCoarseDepthL0 (:138-295) on the device
void CoarseTracker::setCoarseTrackingRef(std::vector<FrameHessian*> frameHessians)
{
	typedef void (*Fn)(CoarseTracker*, std::vector<FrameHessian*>);
	static Fn orig = original<Fn>("_ZN3dso13CoarseTracker20setCoarseTrackingRefESt6vectorIPNS_12FrameHessianESaIS3_EE");
	Timer tm(g.stats, 1);
	if (!g.on) { orig(this, frameHessians); return; }
	if (g.shadow) orig(this, frameHessians);   // the CPU template for the reference's own trackNewestCoarse; the device template below for the shadow call
	assert(frameHessians.size() > 0);
	lastRef = frameHessians.back();
	// the points makeCoarseDepthL0 scatters (:144-161): active points whose newest residual is IN, at the pixel it projects to in lastRef, weighted by HdiF
	std::vector<float> u, v, id, hdi;
	for (FrameHessian* fh : frameHessians)
		for (PointHessian* ph : fh->pointHessians)
			if (ph->lastResiduals[0].first != 0 && ph->lastResiduals[0].second == ResState::IN)
			{
				PointFrameResidual* r = ph->lastResiduals[0].first;
				assert(r->efResidual->isActive() && r->target == lastRef);
				u.push_back(r->centerProjectedTo[0]); v.push_back(r->centerProjectedTo[1]); id.push_back(r->centerProjectedTo[2]); hdi.push_back(ph->efPoint->HdiF);
			}
	dmvio_hip_tracker* trk = trackerFor(this);
	const float k4[4] = {fx[0], fy[0], cx[0], cy[0]};   // makeK(&Hcalib) ran just before (FullSystem.cpp:1459): level-0 intrinsics as the tracker holds them
	lastRef_aff_g2l = lastRef->aff_g2l();
	HIP_OK(dmvio_hip_tracker_make_k(trk, k4));
	HIP_OK(dmvio_hip_tracker_set_ref(trk, slotFor(lastRef), lastRef->ab_exposure, lastRef_aff_g2l.a, lastRef_aff_g2l.b, (int)u.size(), u.data(), v.data(), id.data(), hdi.data()));
	refFrameID = lastRef->shell->id;
	firstCoarseRMSE = -1;
}

// ---- CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:539-770): the visual-only branch through dmvio_hip_tracker_track, the default branch (setting_useIMU) through
// dmvio_hip_tracker_track_vio with the three IMUIntegration members as callbacks (trackVio above)
bool CoarseTracker::trackNewestCoarse(FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl, Vec5 minResForAbort, IOWrap::Output3DWrapper* wrap)
{
	typedef bool (*Fn)(CoarseTracker*, FrameHessian*, SE3&, AffLight&, int, Vec5, IOWrap::Output3DWrapper*);
	static Fn orig = original<Fn>("_ZN3dso13CoarseTracker17trackNewestCoarseEPNS_12FrameHessianERN6Sophus8SE3GroupIdLi0EEERNS_8AffLightEiN5Eigen6MatrixIdLi5ELi1ELi0ELi5ELi1EEEPNS_6IOWrap15Output3DWrapperE");
	Timer tm(g.stats, 2);
	g.tnc_inner_calls++;
	if (!g.on) return orig(this, newFrameHessian, lastToNew_out, aff_g2l_out, coarsestLvl, minResForAbort, wrap);
	if (g.shadow)
	{
		double pose7[7], aff[2] = {aff_g2l_out.a, aff_g2l_out.b}, minRes[5], lastRes[5], flow[3], H[64], b[8];
		toPose7(lastToNew_out, pose7);
		for (int i = 0; i < 5; i++) minRes[i] = minResForAbort[i];
		int good = 0;
		bool ok;
		if (setting_useIMU)
		{
			// both sides talk to the same stateful facade: the device's call first, on a state that is put back for the reference's own
			void* saved = g.vio_save ? g.vio_save(g.vio_sys) : nullptr;
			ok = trackVio(this, trackerFor(this), slotFor(newFrameHessian), newFrameHessian->ab_exposure, pose7, aff, coarsestLvl, minRes, lastRes, flow, H, b, &good);
			if (g.vio_restore) g.vio_restore(g.vio_sys, saved, 0);
		}
		else ok = HIP_OK(dmvio_hip_tracker_track(trackerFor(this), slotFor(newFrameHessian), newFrameHessian->ab_exposure, pose7, aff, coarsestLvl, minRes, lastRes, flow, H, b, &good));
		const bool ret = orig(this, newFrameHessian, lastToNew_out, aff_g2l_out, coarsestLvl, minResForAbort, wrap);
		if (ok)
		{
			g.sh.n_track++;
			bool finished = true, finishedRef = true;
			for (int l = 0; l <= coarsestLvl; l++) { if (!std::isfinite(lastRes[l])) finished = false; if (!std::isfinite(lastResiduals[l])) finishedRef = false; }
			if (finished != finishedRef || (finished && (good != 0) != ret)) g.sh.n_track_good_diff++;
			else if (finished)
			{
				double p7[7]; toPose7(lastToNew_out, p7);
				for (int i = 0; i < 3; i++) g.sh.track_pose = std::max(g.sh.track_pose, std::fabs(p7[i] - pose7[i]));
				g.sh.track_aff_a = std::max(g.sh.track_aff_a, std::fabs(aff[0] - aff_g2l_out.a)); g.sh.track_aff_b = std::max(g.sh.track_aff_b, std::fabs(aff[1] - aff_g2l_out.b));
				g.sh.track_res_rel = std::max(g.sh.track_res_rel, std::fabs(lastRes[0] - lastResiduals[0]) / lastResiduals[0]);
			}
		}
		return ret;
	}
	assert(coarsestLvl < 5 && coarsestLvl < pyrLevelsUsed);
	lastResiduals.setConstant(NAN);
	lastFlowIndicators.setConstant(1000);
	newFrame = newFrameHessian;
	double pose7[7], aff[2] = {aff_g2l_out.a, aff_g2l_out.b}, minRes[5], lastRes[5], flow[3], H[64], b[8];
	toPose7(lastToNew_out, pose7);
	for (int i = 0; i < 5; i++) minRes[i] = minResForAbort[i];
	int good = 0;
	// setting_useIMU (the reference's default): the LM step belongs to IMUIntegration::computeCoarseUpdate once the IMU is coarse-initialised (:612-637), acceptCoarseUpdate and
	// addVisualToCoarseGraph are told about accepted steps and the final system either way (:708, :765)
	const bool ok = setting_useIMU ? trackVio(this, trackerFor(this), slotFor(newFrameHessian), newFrameHessian->ab_exposure, pose7, aff, coarsestLvl, minRes, lastRes, flow, H, b, &good)
	                               : HIP_OK(dmvio_hip_tracker_track(trackerFor(this), slotFor(newFrameHessian), newFrameHessian->ab_exposure, pose7, aff, coarsestLvl, minRes, lastRes, flow, H, b, &good));
	if (!ok) return false;   // a device error reads as "tracking failed" 
	for (int i = 0; i < 5; i++) lastResiduals[i] = lastRes[i];
	// an aborted level leaves the outputs untouched and the flow indicators of the levels it finished (:729-733); otherwise they are the finest level's
	bool finished = true;
	for (int l = 0; l <= coarsestLvl; l++) if (!std::isfinite(lastRes[l])) finished = false;
	lastFlowIndicators = Vec3(flow[0], flow[1], flow[2]);   // the library mirrors the member: 1000 until a level finished, then that level's indicators
	if (!finished) return false;
	lastToNew_out = fromPose7(pose7);
	aff_g2l_out = AffLight(aff[0], aff[1]);
	return good != 0;
}

// ---- FullSystem::trackNewCoarse (FullSystem.cpp:300-539), visual-only branch without a pose hint: the motion hypotheses (:364-402) from dmvio_hip_make_track_hypotheses, the
// try loop (:419-489) as dmvio_hip_tracker_track_new_coarse — try 0 alone (it usually wins), the remaining tries as ONE batch of alignment problems, then the reference's
// sequential rule (abort against achievedRes, winner, re-track threshold) replayed on their results —, the bookkeeping around it (:304-345, 491-537) as in the reference.
// With a hint (the IMU's, one try) or setting_useIMU (a try that is not good still counts, the facade is told about every accepted step: per-try state) the reference's own
// loop stays and calls trackNewestCoarse above per try.
std::pair<Vec4, bool> FullSystem::trackNewCoarse(FrameHessian* fh, Sophus::SE3* referenceToFrameHint)
{
	typedef std::pair<Vec4, bool> (*Fn)(FullSystem*, FrameHessian*, Sophus::SE3*);
	static Fn orig = original<Fn>("_ZN3dso10FullSystem14trackNewCoarseEPNS_12FrameHessianEPN6Sophus8SE3GroupIdLi0EEE");
	g.fs = this;
	g.n_tnc++;
	const bool batchable = g.on && g.batch_tries && !referenceToFrameHint && !setting_useIMU && allFrameHistory.size() > 2;
	if (!batchable) return orig(this, fh, referenceToFrameHint);
	FrameHessian* lastF = coarseTracker->lastRef;
	FrameShell* slast = allFrameHistory[allFrameHistory.size() - 2];
	FrameShell* sprelast = allFrameHistory[allFrameHistory.size() - 3];
	double s7[7], p7[7], l7[7], aff_last[2];
	bool posesValid;
	{	// lock on global pose consistency (:354-360)
		boost::unique_lock<boost::mutex> crlock(shellPoseMutex);
		toPose7(slast->camToWorld, s7); toPose7(sprelast->camToWorld, p7); toPose7(lastF->shell->camToWorld, l7);
		aff_last[0] = slast->aff_g2l.a; aff_last[1] = slast->aff_g2l.b;
		posesValid = slast->poseValid && sprelast->poseValid && lastF->shell->poseValid;
	}
	std::vector<double> tries(7 * 64);
	int n_tries = dmvio_hip_make_track_hypotheses(s7, p7, l7, tries.data(), 64);
	if (n_tries < 1) { fail("dmvio_hip_make_track_hypotheses"); return orig(this, fh, referenceToFrameHint); }
	if (!posesValid) { n_tries = 1; const double ident[7] = {0, 0, 0, 0, 0, 0, 1}; memcpy(tries.data(), ident, sizeof(ident)); }   // :397-401
	double rmse[5], pose7[7], aff_out[2], flow[3];
	for (int i = 0; i < 5; i++) rmse[i] = lastCoarseRMSE[i];
	int winner = -1, used = 0, good = 0;
	const bool shadow = g.shadow;
	std::unique_ptr<dmvio::TimeMeasurement> timeMeasurement;
	if (!shadow) timeMeasurement.reset(new dmvio::TimeMeasurement("FullSystem::trackNewCoarseNoIMU"));
	if (!shadow) for (IOWrap::Output3DWrapper* ow : outputWrapper) ow->pushLiveFrame(fh);
	dmvio_hip_tracker* trk = trackerFor(coarseTracker);
	std::unique_ptr<Timer> tm;
	if (!shadow) tm.reset(new Timer(g.stats, 2));   // counted where the per-try trackNewestCoarse calls it replaces were counted
	const bool ok = HIP_OK(dmvio_hip_tracker_track_new_coarse(trk, slotFor(fh), fh->ab_exposure, n_tries, tries.data(), aff_last, rmse, setting_reTrackThreshold, pose7, aff_out, flow,
	                                                         &winner, &used, &good));
	if (shadow)
	{
		// the reference's own loop (every try through the shadowed trackNewestCoarse above), then the two walks side by side
		g.tnc_inner_calls = 0;
		const std::pair<Vec4, bool> ret = orig(this, fh, referenceToFrameHint);
		if (ok)
		{
			g.tnc.n++;
			if (g.tnc_inner_calls > 1) g.tnc.n_past_try0++;
			g.tnc.max_tries = std::max(g.tnc.max_tries, g.tnc_inner_calls);
			if (g.tnc_inner_calls != used) g.tnc.n_tries_diff++;
			if ((good != 0) != ret.second) g.tnc.n_good_diff++;
			double r7[7]; toPose7(fh->shell->camToTrackingRef.inverse(), r7);
			for (int i = 0; i < 3; i++) g.tnc.pose = std::max(g.tnc.pose, std::fabs(r7[i] - pose7[i]));
			g.tnc.aff_a = std::max(g.tnc.aff_a, std::fabs(fh->shell->aff_g2l.a - aff_out[0]));
			if (std::isfinite(rmse[0]) && std::isfinite(lastCoarseRMSE[0])) g.tnc.res_rel = std::max(g.tnc.res_rel, std::fabs(rmse[0] - lastCoarseRMSE[0]) / lastCoarseRMSE[0]);
		}
		return ret;
	}
	tm.reset();
	if (!ok) return orig(this, fh, referenceToFrameHint);
	g.n_tnc_batched++; g.n_tnc_tries += used;
	if (used > 1) g.n_tnc_past_try0++;
	if (winner < 0)
	{
		// :491-505: no try was good — the predicted pose is taken (the library returned tries[0], the last affine parameters and zero flow)
		printf("BIG ERROR! tracking failed entirely. Take predicted pose and hope we may somehow recover.\n");
		const SE3 lastF_2_fh = fromPose7(pose7);
		if (lastF_2_fh.translation().norm() > 100000 || lastF_2_fh.matrix().hasNaN()) { std::cerr << "TRACKING FAILED ENTIRELY, NO HOPE TO RECOVER" << std::endl; exit(1); }
	}
	for (int i = 0; i < 5; i++) lastCoarseRMSE[i] = rmse[i];   // = achievedRes (:507)
	// the members of CoarseTracker the rest of FullSystem reads after the loop: what the LAST try it walked left there is not reproduced (nothing reads it), the winner's is
	for (int i = 0; i < 5; i++) coarseTracker->lastResiduals[i] = rmse[i];
	coarseTracker->lastFlowIndicators = Vec3(flow[0], flow[1], flow[2]);
	// no lock required, as fh is not used anywhere yet (:509-514)
	fh->shell->camToTrackingRef = fromPose7(pose7).inverse();
	fh->shell->trackingRef = lastF->shell;
	fh->shell->aff_g2l = AffLight(aff_out[0], aff_out[1]);
	fh->shell->camToWorld = fh->shell->trackingRef->camToWorld * fh->shell->camToTrackingRef;
	fh->shell->trackingWasGood = good != 0;
	if (coarseTracker->firstCoarseRMSE < 0) coarseTracker->firstCoarseRMSE = rmse[0];
	if (!setting_debugout_runquiet) printf("Coarse Tracker tracked ab = %f %f (exp %f). Res %f!\n", aff_out[0], aff_out[1], fh->ab_exposure, rmse[0]);
	if (setting_logStuff)
		(*coarseTrackingLog) << std::setprecision(16) << fh->shell->id << " " << fh->shell->timestamp << " " << fh->ab_exposure << " " << fh->shell->camToWorld.log().transpose() << " "
		                     << aff_out[0] << " " << aff_out[1] << " " << rmse[0] << " " << used << "\n";
	return std::make_pair(Vec4(rmse[0], flow[0], flow[1], flow[2]), good != 0);
}

// ---- FullSystem::traceNewCoarse (FullSystem.cpp:541-584): the per-host tables exactly as the reference forms them, ImmaturePoint::traceOn of every point on the device
void FullSystem::traceNewCoarse(FrameHessian* fh)
{
	typedef void (*Fn)(FullSystem*, FrameHessian*);
	static Fn orig = original<Fn>("_ZN3dso10FullSystem14traceNewCoarseEPNS_12FrameHessianE");
	Timer tm(g.stats, 3);
	g.fs = this;
	if (!g.on) { orig(this, fh); return; }
	std::unique_ptr<dmvio::TimeMeasurement> timeMeasurement;   // the profiler scopes the reference opens (:543, FullSystemOptimize.cpp:419) stay where they were
	if (!g.shadow) timeMeasurement.reset(new dmvio::TimeMeasurement("traceNewCoarse"));
	std::unique_ptr<boost::unique_lock<boost::mutex>> lock(new boost::unique_lock<boost::mutex>(mapMutex));
	Mat33f K = Mat33f::Identity();
	K(0, 0) = Hcalib.fxl(); K(1, 1) = Hcalib.fyl(); K(0, 2) = Hcalib.cxl(); K(1, 2) = Hcalib.cyl();
	const int nH = (int)frameHessians.size();
	std::vector<float> KRKi9(9 * nH), Kt3(3 * nH), aff2(2 * nH);
	std::vector<ImmaturePoint*> pts;
	std::vector<int> hostIDs(nH);
	for (int hI = 0; hI < nH; hI++)
	{
		FrameHessian* host = frameHessians[hI];
		hostIDs[hI] = host->frameID;
		SE3 hostToNew = fh->PRE_worldToCam * host->PRE_camToWorld;
		Mat33f KRKi = K * hostToNew.rotationMatrix().cast<float>() * K.inverse();
		Vec3f Kt = K * hostToNew.translation().cast<float>();
		Vec2f aff = AffLight::fromToVecExposure(host->ab_exposure, fh->ab_exposure, host->aff_g2l(), fh->aff_g2l()).cast<float>();
		for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) KRKi9[9 * hI + 3 * r + c] = KRKi(r, c); Kt3[3 * hI + r] = Kt[r]; }
		aff2[2 * hI] = aff[0]; aff2[2 * hI + 1] = aff[1];
		for (ImmaturePoint* ip : host->immaturePoints) pts.push_back(ip);
	}
	const int n = (int)pts.size();
	if (n == 0) { if (g.shadow) { lock.reset(); orig(this, fh); } return; }
	std::vector<float> imin(n), imax(n), qual(n), uv(2 * n), interval(n);
	std::vector<int> status(n);
	std::vector<unsigned char> wasOOB(n);   // traceOn returns at once for a point that is OOB already (:79): nothing of it changes
	for (int i = 0; i < n; i++) wasOOB[i] = pts[i]->lastTraceStatus == IPS_OOB;
	// Between two keyframes nothing but traceOn touches the immature points, and the device holds what the last call left: the set stays resident.  It is rebuilt when the
	// window or a point list changed (every keyframe: activation and optimize invalidate it) and always in shadow mode, where the reference's own traceOn writes the objects.
	const bool resident = !g.shadow && g.imm_valid && pts == g.imm_pts && hostIDs == g.imm_hostIDs;
	if (!resident)
	{
		g.imm_valid = false;
		if (!HIP_OK(dmvio_hip_immature_clear(g.imm))) return;
		for (int hI = 0; hI < nH; hI++)
		{
			// the points of this host: constructed on the device from its image (bit-identical to ImmaturePoint::ImmaturePoint, ImmaturePoint.cpp:34-62), state from the objects
			FrameHessian* host = frameHessians[hI];
			std::vector<int> ui, vi;
			for (ImmaturePoint* ip : host->immaturePoints) { ui.push_back((int)ip->u); vi.push_back((int)ip->v); }
			if (!ui.empty() && !HIP_OK(dmvio_hip_immature_add_points(g.imm, hI, slotFor(host), (int)ui.size(), ui.data(), vi.data()))) return;
		}
		for (int i = 0; i < n; i++) { imin[i] = pts[i]->idepth_min; imax[i] = pts[i]->idepth_max; qual[i] = pts[i]->quality; status[i] = (int)pts[i]->lastTraceStatus; }
		if (!HIP_OK(dmvio_hip_immature_set_state(g.imm, imin.data(), imax.data(), qual.data(), status.data()))) return;
		g.imm_pts = pts; g.imm_hostIDs = hostIDs; g.imm_valid = true;
	}
	if (!HIP_OK(dmvio_hip_immature_trace(g.imm, slotFor(fh), nH, KRKi9.data(), Kt3.data(), aff2.data()))) return;
	if (!HIP_OK(dmvio_hip_immature_get_state(g.imm, imin.data(), imax.data(), qual.data(), uv.data(), interval.data(), status.data()))) return;
	if (g.shadow)
	{
		lock.reset();
		orig(this, fh);   // the reference's own traceOn of every point; then field by field, bit by bit
		for (int i = 0; i < n; i++)
		{
			ImmaturePoint* ip = pts[i];
			if (wasOOB[i]) continue;
			const float ruv[2] = {ip->lastTraceUV[0], ip->lastTraceUV[1]};
			const bool same = (int)ip->lastTraceStatus == status[i] && memcmp(&ip->idepth_min, &imin[i], 4) == 0 && memcmp(&ip->idepth_max, &imax[i], 4) == 0 &&
			                  memcmp(&ip->quality, &qual[i], 4) == 0 && memcmp(ruv, &uv[2 * i], 8) == 0 && memcmp(&ip->lastTracePixelInterval, &interval[i], 4) == 0;
			g.sh.n_trace_pts++;
			if (!same)
			{
				if (g.sh.n_trace_diff < 8 && getenv("DROPIN_DEBUG"))
					fprintf(stderr, "[dropin] trace diff: status %d / %d, idepth [%g %g] / [%g %g], quality %g / %g, uv (%g %g) / (%g %g), interval %g / %g\n", (int)ip->lastTraceStatus, status[i],
					        ip->idepth_min, ip->idepth_max, imin[i], imax[i], ip->quality, qual[i], ruv[0], ruv[1], uv[2 * i], uv[2 * i + 1], ip->lastTracePixelInterval, interval[i]);
				g.sh.n_trace_diff++;
			}
		}
		return;
	}
	for (int i = 0; i < n; i++)
	{
		ImmaturePoint* ip = pts[i];
		if (wasOOB[i]) continue;
		ip->idepth_min = imin[i]; ip->idepth_max = imax[i]; ip->quality = qual[i];
		ip->lastTraceUV = Vec2f(uv[2 * i], uv[2 * i + 1]); ip->lastTracePixelInterval = interval[i];
		ip->lastTraceStatus = (ImmaturePointStatus)status[i];
	}
}

// ---- FullSystem::activatePointsMT_Reductor (FullSystem.cpp:589-601): FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:51-205) for the candidates [min, max) — the Gauss-Newton
// on the inverse depth of every candidate on the device in ONE call, then the tail of optimizeImmaturePoint (:165-203: the PointHessian and its residuals) for those it activates
namespace {
struct Activated { int result; float idepth; std::vector<int> targets; };   // targets: frame indices of the residuals that are IN
bool hipActivate(FullSystem* fs, const std::vector<ImmaturePoint*>& cand, std::vector<Activated>& out)
{
	const int F = (int)fs->frameHessians.size(), n = (int)cand.size();
	out.assign(n, Activated{0, 0.f, {}});
	if (n == 0) return true;
	g.imm_valid = false;
	if (!HIP_OK(dmvio_hip_immature_clear(g.imm))) return false;
	// the handle keeps its points grouped by host keyframe: candidates of host h, in candidate order
	std::vector<int> order;
	for (int h = 0; h < F; h++)
	{
		std::vector<int> ui, vi;
		for (int k = 0; k < n; k++) if (cand[k]->host == fs->frameHessians[h]) { ui.push_back((int)cand[k]->u); vi.push_back((int)cand[k]->v); order.push_back(k); }
		if (!ui.empty() && !HIP_OK(dmvio_hip_immature_add_points(g.imm, h, slotFor(fs->frameHessians[h]), (int)ui.size(), ui.data(), vi.data()))) return false;
	}
	if ((int)order.size() != n) { fprintf(stderr, "[dropin] activation candidate without a keyframe in the window\n"); abort(); }
	std::vector<float> imin(n), imax(n), qual(n), idepth(n), expo(F);
	std::vector<int> status(n), slots(F), result(n), rstate((size_t)n * F);
	std::vector<double> w2c(7 * (size_t)F), aff(2 * (size_t)F);
	for (int j = 0; j < n; j++) { const ImmaturePoint* ip = cand[order[j]]; imin[j] = ip->idepth_min; imax[j] = ip->idepth_max; qual[j] = ip->quality; status[j] = (int)ip->lastTraceStatus; }
	for (int f = 0; f < F; f++)
	{
		FrameHessian* fh = fs->frameHessians[f];
		slots[f] = slotFor(fh); expo[f] = fh->ab_exposure; toPose7(fh->PRE_worldToCam, &w2c[7 * f]); aff[2 * f] = fh->aff_g2l().a; aff[2 * f + 1] = fh->aff_g2l().b;
	}
	if (!HIP_OK(dmvio_hip_immature_set_state(g.imm, imin.data(), imax.data(), qual.data(), status.data()))) return false;
	if (!HIP_OK(dmvio_hip_immature_optimize(g.imm, F, slots.data(), w2c.data(), aff.data(), expo.data(), fs->Hcalib.value_scaled.data(), nullptr, 1, result.data(), idepth.data(), rstate.data())))
		return false;
	for (int j = 0; j < n; j++)
	{
		Activated& a = out[order[j]];
		a.result = result[j]; a.idepth = idepth[j];
		if (result[j] == 1) for (int t = 0; t < F; t++) if (rstate[(size_t)j * F + t] == 0) a.targets.push_back(t);
	}
	return true;
}
}  // namespace
void FullSystem::activatePointsMT_Reductor(std::vector<PointHessian*>* optimized, std::vector<ImmaturePoint*>* toOptimize, int min, int max, Vec10* stats, int tid)
{
	typedef void (*Fn)(FullSystem*, std::vector<PointHessian*>*, std::vector<ImmaturePoint*>*, int, int, Vec10*, int);
	static Fn orig = original<Fn>("_ZN3dso10FullSystem25activatePointsMT_ReductorEPSt6vectorIPNS_12PointHessianESaIS3_EEPS1_IPNS_13ImmaturePointESaIS8_EEiiPN5Eigen6MatrixIdLi10ELi1ELi0ELi10ELi1EEEi");
	Timer tm(g.stats, 5);
	g.fs = this;
	if (!g.on) { orig(this, optimized, toOptimize, min, max, stats, tid); return; }
	std::vector<ImmaturePoint*> cand(toOptimize->begin() + min, toOptimize->begin() + max);
	std::vector<Activated> act;
	bool ok;
	{
		// multiThreading = true: the reference's worker pool enters this member from six threads at once, 50 candidates each (FullSystem.cpp:744); the immature handle is one
		static std::mutex activateMu;
		std::lock_guard<std::mutex> lk(activateMu);
		ok = hipActivate(this, cand, act);
	}
	if (!ok || g.shadow)
	{
		orig(this, optimized, toOptimize, min, max, stats, tid);
		if (!ok) return;
		for (int k = 0; k < (int)cand.size(); k++)
		{
			PointHessian* p = (*optimized)[min + k];
			const int cls = p == 0 ? 0 : (p == (PointHessian*)((long)(-1)) ? -1 : 1);
			bool same = cls == act[k].result;
			if (same && cls == 1)
			{
				same = memcmp(&p->idepth, &act[k].idepth, 4) == 0 && p->residuals.size() == act[k].targets.size();
				for (size_t i = 0; same && i < p->residuals.size(); i++) same = p->residuals[i]->target->idx == act[k].targets[i];
			}
			g.sh.n_act_pts++;
			if (!same)
			{
				if (g.sh.n_act_diff < 8 && getenv("DROPIN_DEBUG"))
					fprintf(stderr, "[dropin] activation diff: class %d / %d, idepth %g / %g, residuals %zu / %zu\n", cls, act[k].result, cls == 1 ? p->idepth : 0.f, act[k].idepth,
					        cls == 1 ? p->residuals.size() : (size_t)0, act[k].targets.size());
				g.sh.n_act_diff++;
			}
		}
		return;
	}
	for (int k = 0; k < (int)cand.size(); k++)
	{
		const Activated& a = act[k];
		if (a.result == 0) { (*optimized)[min + k] = 0; continue; }
		if (a.result < 0 || !std::isfinite(a.idepth)) { (*optimized)[min + k] = (PointHessian*)((long)(-1)); continue; }
		// FullSystemOptPoint.cpp:165-203
		PointHessian* p = new PointHessian(cand[k], &Hcalib);
		if (!std::isfinite(p->energyTH)) { delete p; (*optimized)[min + k] = (PointHessian*)((long)(-1)); continue; }
		p->lastResiduals[0].first = 0; p->lastResiduals[0].second = ResState::OOB;
		p->lastResiduals[1].first = 0; p->lastResiduals[1].second = ResState::OOB;
		p->setIdepthZero(a.idepth);
		p->setIdepth(a.idepth);
		p->setPointStatus(PointHessian::ACTIVE);
		for (int t : a.targets)
		{
			PointFrameResidual* r = new PointFrameResidual(p, p->host, frameHessians[t]);
			r->state_NewEnergy = r->state_energy = 0;
			r->state_NewState = ResState::OUTLIER;
			r->setState(ResState::IN);
			p->residuals.push_back(r);
			if (r->target == frameHessians.back()) { p->lastResiduals[0].first = r; p->lastResiduals[0].second = ResState::IN; }
			else if (r->target == (frameHessians.size() < 2 ? 0 : frameHessians[frameHessians.size() - 2])) { p->lastResiduals[1].first = r; p->lastResiduals[1].second = ResState::IN; }
		}
		statistics_numActivatedPoints++;
		(*optimized)[min + k] = p;
	}
}

// ---- EnergyFunctional's mutators (EnergyFunctional.cpp:435-518, 641-646, 766-782), each forwarded to the resident window graph with the indices the reference keeps in
// its own objects (EFFrame::idx, EFPoint::idxInPoints, EFResidual::idxInAll) — what a maintainer adds as ONE line at the end of each member.  The reference's definition
// runs unchanged; the mirror answers with the index the reference just assigned, which is checked (a disagreement invalidates the mirror: the next optimize rebuilds it
// from the pointer graph and counts a resync).
namespace {
bool graphFollows(const EnergyFunctional* ef) { return g.graph && g.graph_valid && g.graph_ef == ef; }
void graphCheck(bool ok, const char* what)
{
	g.graph_ops++;
	if (!ok) { g.graph_valid = false; fprintf(stderr, "[dropin] window graph out of step at %s: %s\n", what, dmvio_hip_last_error()); }
}
}  // namespace
}  // namespace dso
namespace dso
{
EFFrame* EnergyFunctional::insertFrame(FrameHessian* fh, CalibHessian* Hcalib)
{
	typedef EFFrame* (*Fn)(EnergyFunctional*, FrameHessian*, CalibHessian*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional11insertFrameEPNS_12FrameHessianEPNS_12CalibHessianE");
	// a fresh EnergyFunctional (a new FullSystem): the mirror starts over with it
	if (g.graph && frames.empty() && nPoints == 0) { dmvio_hip_graph_clear(g.graph); g.graph_ef = this; g.graph_valid = true; }
	EFFrame* eff = orig(this, fh, Hcalib);
	if (graphFollows(this)) graphCheck(dmvio_hip_graph_insert_frame(g.graph) == eff->idx, "insertFrame");
	return eff;
}
EFPoint* EnergyFunctional::insertPoint(PointHessian* ph)
{
	typedef EFPoint* (*Fn)(EnergyFunctional*, PointHessian*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional11insertPointEPNS_12PointHessianE");
	EFPoint* efp = orig(this, ph);
	if (graphFollows(this))
		graphCheck(dmvio_hip_graph_insert_point(g.graph, efp->host->idx, ph->u, ph->v, ph->idepth, ph->color, ph->weights, ph->hasDepthPrior ? 1 : 0) == efp->idxInPoints, "insertPoint");
	return efp;
}
EFResidual* EnergyFunctional::insertResidual(PointFrameResidual* r)
{
	typedef EFResidual* (*Fn)(EnergyFunctional*, PointFrameResidual*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional14insertResidualEPNS_18PointFrameResidualE");
	EFResidual* efr = orig(this, r);
	if (graphFollows(this))
		graphCheck(dmvio_hip_graph_insert_residual(g.graph, efr->point->host->idx, efr->point->idxInPoints, efr->target->idx) == efr->idxInAll, "insertResidual");
	return efr;
}
void EnergyFunctional::dropResidual(EFResidual* r)
{
	typedef void (*Fn)(EnergyFunctional*, EFResidual*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional12dropResidualEPNS_10EFResidualE");
	const int h = r->point->host->idx, i = r->point->idxInPoints, k = r->idxInAll;   // r is deleted by the call
	orig(this, r);
	if (graphFollows(this)) graphCheck(dmvio_hip_graph_drop_residual(g.graph, h, i, k) >= 0, "dropResidual");
}
void EnergyFunctional::removePoint(EFPoint* p)
{
	typedef void (*Fn)(EnergyFunctional*, EFPoint*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional11removePointEPNS_7EFPointE");
	const int h = p->host->idx, i = p->idxInPoints;
	orig(this, p);   // drops the point's residuals through dropResidual above, one by one, then moves the frame's last point into its place
	if (graphFollows(this)) graphCheck(dmvio_hip_graph_remove_point(g.graph, h, i) >= 0, "removePoint");
}
void EnergyFunctional::marginalizeFrame(EFFrame* fh)
{
	typedef void (*Fn)(EnergyFunctional*, EFFrame*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional16marginalizeFrameEPNS_7EFFrameE");
	const int idx = fh->idx;
	orig(this, fh);
	if (graphFollows(this)) graphCheck(dmvio_hip_graph_remove_frame(g.graph, idx) >= 0, "marginalizeFrame");   // the residuals that target it are dropped next (FullSystemMarginalize.cpp:169-196)
}

// ---- the window of the reference's FullSystem handed to the BA handle: keyframes in frameHessians order, points in EnergyFunctional::allPoints order (makeIDX,
// EnergyFunctional.cpp:997-1017), residuals in EFPoint::residualsAll order — the orders the reference's accumulators add in — with the states, FEJ points, thresholds,
// calibration and marginalisation prior the reference holds at this moment
namespace {
struct FlatWindow
{
	int F = 0, N = 0, R = 0, n = 0;
	std::vector<PointHessian*> points;
	// flat position of a residual without a look-up table: the EnergyFunctional keeps the indices itself (EFFrame::idx, EFPoint::idxInPoints, EFResidual::idxInAll,
	// EnergyFunctional.cpp:438, 488, 503-506, 772-773)
	std::vector<int> frameFirstPoint, pointFirstRes;
	int resIndex(const PointFrameResidual* r) const
	{
		const EFPoint* efp = r->point->efPoint;
		return pointFirstRes[frameFirstPoint[efp->host->idx] + efp->idxInPoints] + r->efResidual->idxInAll;
	}
};
// the pointer graph flattened into the arrays of dmvio_hip_ba_set_graph (what every keyframe cost before the window graph was kept resident; still the path of shadow
// mode, of resident mode 0, of the verification in mode 2 and of a resync)
struct FlatArrays
{
	std::vector<int> host, resPoint, resTarget;
	std::vector<float> pu, pv, pid, color, weights;
	std::vector<unsigned char> prior, linearised;
};
void flattenGraph(FullSystem* fs, FlatWindow& W, FlatArrays& A)
{
	W.points.clear(); W.frameFirstPoint.assign(fs->ef->frames.size() + 1, 0); W.pointFirstRes.clear();
	{
		size_t np = 0, nr = 0;
		for (EFFrame* eff : fs->ef->frames) { np += eff->points.size(); for (EFPoint* efp : eff->points) nr += efp->residualsAll.size(); }
		W.points.reserve(np); W.pointFirstRes.reserve(np + 1); A.host.reserve(np); A.pu.reserve(np); A.pv.reserve(np); A.pid.reserve(np); A.prior.reserve(np); A.color.reserve(8 * np); A.weights.reserve(8 * np);
		A.resPoint.reserve(nr); A.resTarget.reserve(nr); A.linearised.reserve(nr);
	}
	for (EFFrame* eff : fs->ef->frames)
	{
		assert(fs->ef->frames[eff->idx] == eff);
		W.frameFirstPoint[eff->idx] = (int)W.points.size();
		for (EFPoint* efp : eff->points)
		{
			PointHessian* ph = efp->data;
			const int pi = (int)W.points.size();
			assert(eff->points[efp->idxInPoints] == efp);
			W.points.push_back(ph);
			W.pointFirstRes.push_back((int)A.resPoint.size());
			A.host.push_back(ph->host->idx); A.pu.push_back(ph->u); A.pv.push_back(ph->v); A.pid.push_back(ph->idepth); A.prior.push_back(ph->hasDepthPrior ? 1 : 0);
			for (int k = 0; k < 8; k++) { A.color.push_back(ph->color[k]); A.weights.push_back(ph->weights[k]); }
			for (EFResidual* er : efp->residualsAll)
			{
				assert((int)A.resPoint.size() - W.pointFirstRes.back() == er->idxInAll);
				A.resPoint.push_back(pi); A.resTarget.push_back(er->data->target->idx); A.linearised.push_back(er->isLinearized ? 1 : 0);
			}
		}
	}
	W.pointFirstRes.push_back((int)A.resPoint.size());
	W.N = (int)W.points.size(); W.R = (int)A.resPoint.size();
}
// resident mode: only the index tables of the write-back are formed here (one pass over the EFPoints: their PointHessian and the size of residualsAll) — no per-residual walk
void indexGraph(FullSystem* fs, FlatWindow& W)
{
	W.points.clear(); W.frameFirstPoint.assign(fs->ef->frames.size() + 1, 0); W.pointFirstRes.clear();
	W.points.reserve(fs->ef->nPoints); W.pointFirstRes.reserve(fs->ef->nPoints + 1);
	int nr = 0;
	for (EFFrame* eff : fs->ef->frames)
	{
		W.frameFirstPoint[eff->idx] = (int)W.points.size();
		for (EFPoint* efp : eff->points) { W.points.push_back(efp->data); W.pointFirstRes.push_back(nr); nr += (int)efp->residualsAll.size(); }
	}
	W.pointFirstRes.push_back(nr);
	W.N = (int)W.points.size(); W.R = nr;
}
// the mirror rebuilt from the pointer graph (the adapter was switched on in the middle of a run, or a forwarded call disagreed)
bool resyncGraph(FullSystem* fs, const FlatWindow& W, const FlatArrays& A)
{
	bool ok = HIP_OK(dmvio_hip_graph_clear(g.graph));
	for (size_t f = 0; ok && f < fs->ef->frames.size(); f++) ok = HIP_OK(dmvio_hip_graph_insert_frame(g.graph));
	std::vector<int> idxIn(W.N);
	for (int pi = 0; ok && pi < W.N; pi++)
	{
		idxIn[pi] = dmvio_hip_graph_insert_point(g.graph, A.host[pi], A.pu[pi], A.pv[pi], A.pid[pi], &A.color[8 * (size_t)pi], &A.weights[8 * (size_t)pi], A.prior[pi]);
		ok = idxIn[pi] >= 0;
	}
	for (int ri = 0; ok && ri < W.R; ri++) ok = dmvio_hip_graph_insert_residual(g.graph, A.host[A.resPoint[ri]], idxIn[A.resPoint[ri]], A.resTarget[ri]) >= 0;
	g.graph_ef = fs->ef; g.graph_valid = ok; g.graph_resyncs++;
	return ok;
}
bool graphEqualsFlat(const FlatWindow& W, const FlatArrays& A)
{
	int F = 0, N = 0, R = 0;
	dmvio_hip_graph_counts(g.graph, &F, &N, &R);
	if (N != W.N || R != W.R) return false;
	FlatArrays B;
	B.host.resize(N); B.pu.resize(N); B.pv.resize(N); B.pid.resize(N); B.color.resize(8 * (size_t)N); B.weights.resize(8 * (size_t)N); B.prior.resize(N); B.resPoint.resize(R); B.resTarget.resize(R);
	dmvio_hip_graph_export(g.graph, B.host.data(), B.pu.data(), B.pv.data(), B.pid.data(), B.color.data(), B.weights.data(), B.prior.data(), B.resPoint.data(), B.resTarget.data());
	return B.host == A.host && B.resPoint == A.resPoint && B.resTarget == A.resTarget && B.prior == A.prior && !memcmp(B.pu.data(), A.pu.data(), 4 * (size_t)N) && !memcmp(B.pv.data(), A.pv.data(), 4 * (size_t)N) &&
	       !memcmp(B.pid.data(), A.pid.data(), 4 * (size_t)N) && !memcmp(B.color.data(), A.color.data(), 32 * (size_t)N) && !memcmp(B.weights.data(), A.weights.data(), 32 * (size_t)N);
}
// checkLinearised: hand EFResidual::isLinearized to the library, which refuses a window with linearised residuals outside a marginalisation (dmvio_hip_ba_set_residual_flags).
// resident: take the graph from the mirror (FullSystem::optimize outside shadow mode); the per-residual flags are then only checked when the pointer graph is walked anyway
// (mode 2, resync) — the reference sets EFResidual::isLinearized in one place, right before it removes the point (FullSystem.cpp:836-849).
bool uploadWindow(FullSystem* fs, FlatWindow& W, bool checkLinearised = false, bool resident = false)
{
	auto tprev = std::chrono::steady_clock::now();
	auto lap = [&](int k) { const auto t = std::chrono::steady_clock::now(); g.up_split[k] += std::chrono::duration<double>(t - tprev).count(); tprev = t; };
	const int F = (int)fs->frameHessians.size();
	std::vector<int> slots(F), frameIDs(F);
	std::vector<double> evalPT7(7 * F), affZero(2 * F);
	std::vector<float> expo(F), th(F);
	for (int f = 0; f < F; f++)
	{
		FrameHessian* fh = fs->frameHessians[f];
		assert(fh->idx == f);
		slots[f] = slotFor(fh); frameIDs[f] = fh->frameID; expo[f] = fh->ab_exposure; th[f] = fh->frameEnergyTH;
		toPose7(fh->get_worldToCam_evalPT(), &evalPT7[7 * f]);
		affZero[2 * f] = fh->get_state_zero()[6] * SCALE_A; affZero[2 * f + 1] = fh->get_state_zero()[7] * SCALE_B;
	}
	lap(0);
	resident = resident && g.graph && g.resident != 0;
	FlatArrays A;
	bool walked = false;
	if (resident)
	{
		indexGraph(fs, W);
		int gF = 0, gN = 0, gR = 0;
		dmvio_hip_graph_counts(g.graph, &gF, &gN, &gR);
		const bool inStep = g.graph_valid && g.graph_ef == fs->ef && gF == F && gN == W.N && gR == W.R;
		if (!inStep || g.resident == 2)
		{
			flattenGraph(fs, W, A); walked = true;
			if (!inStep) { if (!resyncGraph(fs, W, A)) return false; }
			else { g.graph_verified++; if (!graphEqualsFlat(W, A)) { g.graph_mismatch++; fprintf(stderr, "[dropin] the resident window graph differs from the flattened pointer graph\n"); if (!resyncGraph(fs, W, A)) return false; } }
		}
	}
	else { flattenGraph(fs, W, A); walked = true; }
	W.F = F; W.n = CPARS + 8 * F;
	if (W.N < 1 || W.R < 1) return false;
	if (!resident)
	{
		g.windowPoint.clear(); g.windowPoint.reserve(2 * W.points.size());
		for (size_t pi = 0; pi < W.points.size(); pi++) g.windowPoint[W.points[pi]] = (int)pi;
	}
	dmvio_hip_ba* ba = g.ba;
	lap(1);
	bool ok = HIP_OK(dmvio_hip_ba_set_window(ba, F, slots.data(), evalPT7.data(), affZero.data(), expo.data(), frameIDs.data(), fs->Hcalib.value_scaled.data()));
	lap(2);
	if (resident) ok = ok && HIP_OK(dmvio_hip_ba_set_graph_from(ba, g.graph));
	else ok = ok && HIP_OK(dmvio_hip_ba_set_graph(ba, W.N, A.host.data(), A.pu.data(), A.pv.data(), A.pid.data(), A.color.data(), A.weights.data(), A.prior.data(), W.R, A.resPoint.data(), A.resTarget.data()));
	if (checkLinearised && walked) ok = ok && HIP_OK(dmvio_hip_ba_set_residual_flags(ba, W.R, A.linearised.data()));
	lap(3);
	{
		std::vector<double> sz(10 * (size_t)F), st(10 * (size_t)F);
		for (int f = 0; f < F; f++)
		{
			const Vec10 z = fs->frameHessians[f]->get_state_zero(), c = fs->frameHessians[f]->get_state();
			for (int i = 0; i < 10; i++) { sz[10 * f + i] = z[i]; st[10 * f + i] = c[i]; }
		}
		ok = ok && HIP_OK(dmvio_hip_ba_set_frame_states(ba, sz.data(), st.data()));
	}
	ok = ok && HIP_OK(dmvio_hip_ba_set_frame_energy_th(ba, th.data())) && HIP_OK(dmvio_hip_ba_set_calib_values(ba, fs->Hcalib.value.data(), fs->Hcalib.value_zero.data()));
	lap(4);
	{
		const int n = W.n;
		std::vector<double> HM((size_t)n * n), bM(n);
		for (int r = 0; r < n; r++) { bM[r] = fs->ef->bM[r]; for (int c = 0; c < n; c++) HM[(size_t)r * n + c] = fs->ef->HM(r, c); }
		ok = ok && HIP_OK(dmvio_hip_ba_set_marg_prior(ba, HM.data(), bM.data()));
	}
	lap(5);
	return ok;
}
}  // namespace

// ---- EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:678-742) with the relinearisation FullSystem::flagPointsForRemoval ran just before it (FullSystem.cpp:836-849).
// The reference's own code stays in charge (the point lists, removePoint, connectivity bookkeeping are host structure); in shadow mode the increment of the marginalisation
// prior is ALSO formed on the device, from the window the BA handle still holds since the optimize of this keyframe: masked relinearisation of the points the reference flagged,
// fixLinearizationF, addPoint<2> + the Schur side, stitched — and compared
}  // namespace dso
namespace dso
{
void EnergyFunctional::marginalizePointsF()
{
	typedef void (*Fn)(EnergyFunctional*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional18marginalizePointsFEv");
	if (g.on && !g.shadow && g.fs && g.real_marg && g.window_current && graphFollows(this))
	{
		// ---- the real member: the window optimize left on the device IS the window the reference holds now (same states, FEJ points, residual states; points dropped since
		// are simply no candidates).  The relinearisation of the flagged points (FullSystem.cpp:836-849), fixLinearizationF, addPoint<2> + the Schur side and the stitch run
		// there; what stays here is the reference's bookkeeping around them (EnergyFunctional.cpp:686-700, 707, 715, 731-742).
		assert(EFDeltaValid); assert(EFAdjointsValid); assert(EFIndicesValid);
		g.window_current = false;
		std::vector<unsigned char> cand;
		cand.assign(g.windowPoints.size(), 0);   // the candidates are addressed in the order of the OPTIMIZE-TIME window (PointHessian::idx, checked against windowPoints)
		allPointsToMarg.clear();
		bool known = true;
		for (EFFrame* f : frames)
			for (EFPoint* p : f->points)
				if (p->stateFlag == EFPointStatus::PS_MARGINALIZE)
				{
					const int i = p->data->idx;
					if (i < 0 || i >= (int)g.windowPoints.size() || g.windowPoints[i] != p->data) { known = false; break; }
					cand[i] = 1;
					allPointsToMarg.push_back(p);
				}
		const int n = CPARS + 8 * nFrames;
		std::vector<double> Hadd((size_t)n * n), badd(n);
		std::vector<unsigned char> decision(cand.size(), 0);
		int resInMdev = 0;
		bool done = known && !allPointsToMarg.empty() && HIP_OK(dmvio_hip_ba_marginalize_points(g.ba, cand.data(), decision.data(), Hadd.data(), badd.data(), &resInMdev, 0));
		// the device classifies every candidate itself (marginalise = 1 / drop = 2, from the idepth_hessian ITS optimize left) and leaves a point it would drop out of Hadd / badd:
		// if the two sides ever disagree about a point, the reference's own member runs instead — nothing has been mutated yet
		if (done) for (size_t i = 0; i < cand.size(); i++) if (cand[i] && decision[i] != 1) { done = false; g.n_real_marg_disagree++; break; }
		if (done)
		{
			for (EFPoint* p : allPointsToMarg)
			{
				p->priorF *= setting_idepthFixPriorMargFac;
				for (EFResidual* r : p->residualsAll)
					if (r->isActive()) connectivityMap[(((uint64_t)r->host->frameID) << 32) + ((uint64_t)r->target->frameID)][1]++;
			}
			for (EFPoint* p : allPointsToMarg) removePoint(p);
			resInM += resInMdev;
			// (setting_solverMode = SOLVER_FIX_LAMBDA | SOLVER_ORTHOGONALIZE_X_LATER, settings.cpp:81: neither of the two orthogonalisations of :717-736 is selected)
			for (int r = 0; r < n; r++)
			{
				for (int c = 0; c < n; c++) { HM(r, c) += Hadd[(size_t)r * n + c]; HMForGTSAM(r, c) += Hadd[(size_t)r * n + c]; }
				bM[r] += badd[r]; bMForGTSAM[r] += badd[r];
			}
			EFIndicesValid = false;
			makeIDX();
			g.n_real_marg++; g.n_real_marg_pts += (long)allPointsToMarg.size();
			return;
		}
		allPointsToMarg.clear();
		orig(this); return;   // nothing flagged, or a point the window does not know: the reference's own
	}
	if (!g.on || !g.shadow || !g.fs) { orig(this); return; }
	// the window as the reference holds it NOW (after optimize, removeOutliers, flagPointsForRemoval's relinearisation, dropPointsF): identical states on both sides; one
	// linearisation + accumulation gives the device the per-point Hessians its marginalise-or-drop rule reads
	FlatWindow FW;
	double e0 = 0;
	if (!uploadWindow(g.fs, FW) || !HIP_OK(dmvio_hip_ba_activate_all(g.ba)) || !HIP_OK(dmvio_hip_ba_linearize_local(g.ba, 0, &e0, nullptr, nullptr)) /* thresholds stay the reference's */ || !HIP_OK(dmvio_hip_ba_apply(g.ba)) ||
	    !HIP_OK(dmvio_hip_ba_accumulate(g.ba, nullptr, nullptr, nullptr, nullptr, nullptr))) { orig(this); return; }
	const int N = FW.N, n = CPARS + 8 * nFrames;
	std::vector<unsigned char> cand(N, 0), decision(N, 0);
	int nc = 0;
	bool known = true;
	for (EFFrame* f : frames)
		for (EFPoint* p : f->points)
			if (p->stateFlag == EFPointStatus::PS_MARGINALIZE)
			{
				auto it = g.windowPoint.find(p->data);
				if (it == g.windowPoint.end()) { known = false; break; }
				cand[it->second] = 1; nc++;
			}
	if (!known || nc == 0) { orig(this); return; }
	std::vector<double> Hadd((size_t)n * n), badd(n);
	int resInMdev = 0;
	const bool ok = HIP_OK(dmvio_hip_ba_marginalize_points(g.ba, cand.data(), decision.data(), Hadd.data(), badd.data(), &resInMdev, 0));
	const MatXX HM0 = HM; const VecX bM0 = bM; const int resInM0 = resInM;
	orig(this);
	if (!ok || HM.rows() != n) return;
	g.sh.n_marg++; g.sh.n_marg_pts += nc;
	for (int i = 0; i < N; i++) if (cand[i] && decision[i] != 1) g.sh.n_marg_reacc_diff++;
	for (int i = 0; i < N; i++)
		if (cand[i])
		{
			auto it = g.deviceHessian.find(FW.points[i]);
			if (it == g.deviceHessian.end()) g.sh.n_marg_unknown++;
			else if (!(it->second > setting_minIdepthH_marg)) g.sh.n_marg_decision_diff++;
		}
	if (resInM - resInM0 != resInMdev) g.sh.n_marg_res_diff++;
	if (getenv("DROPIN_DEBUG")) fprintf(stderr, "[dropin] marginalizePointsF: %d points, residuals reference %d device %d\n", nc, resInM - resInM0, resInMdev);
	double dH = 0, sH = 0, db = 0, sb = 0;
	for (int r = 0; r < n; r++)
	{
		for (int c = 0; c < n; c++) { const double v = HM(r, c) - HM0(r, c); dH = std::max(dH, std::fabs(v - Hadd[(size_t)r * n + c])); sH = std::max(sH, std::fabs(v)); }
		const double v = bM[r] - bM0[r]; db = std::max(db, std::fabs(v - badd[r])); sb = std::max(sb, std::fabs(v));
	}
	if (sH > 0) g.sh.marg_H_rel = std::max(g.sh.marg_H_rel, dH / sH);
	if (sb > 0) g.sh.marg_b_rel = std::max(g.sh.marg_b_rel, db / sb);
}

// ---- EnergyFunctional::solveSystemF (EnergyFunctional.cpp:841-996): forwarded unchanged; counted, so that shadow mode can compare the number of Gauss-Newton iterations the
// reference's own optimize ran (one solveSystemF per iteration, FullSystemOptimize.cpp:485-486 -> :661) with the device's
void EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian* HCalib)
{
	typedef void (*Fn)(EnergyFunctional*, int, double, CalibHessian*);
	static Fn orig = original<Fn>("_ZN3dso16EnergyFunctional12solveSystemFEidPNS_12CalibHessianE");
	g.sh.solve_calls++;
	orig(this, iteration, lambda, HCalib);
}

// ---- FullSystem::optimize (FullSystemOptimize.cpp:417-647): the window flattened once, the whole Gauss-Newton loop and the final fix-linearisation on the device,
// the results written back into FrameHessian / PointHessian / PointFrameResidual with the bookkeeping linearizeAll(true) does (:150-218, :51-85)
float FullSystem::optimize(int mnumOptIts)
{
	typedef float (*Fn)(FullSystem*, int);
	static Fn orig = original<Fn>("_ZN3dso10FullSystem8optimizeEi");
	Timer tm(g.stats, 4);
	g.fs = this;
	g.imm_valid = false;   // a keyframe: immature points are about to be activated, dropped and created
	g.window_current = false;
	if (!g.on) return orig(this, mnumOptIts);
	std::unique_ptr<dmvio::TimeMeasurement> timeMeasurement;
	if (!g.shadow) timeMeasurement.reset(new dmvio::TimeMeasurement("FullSystemOptimize"));
	const int F = (int)frameHessians.size();
	if (F < 2) return 0;
	// ---- statistics and active residuals (:429-448).  With the window graph resident nothing below needs the list (the write-back goes point by point), and resetOOB's
	// effects are all overwritten by it: the 12.8k-object walk is left out
	const bool resident = !g.shadow && g.graph && g.resident != 0;
	activeResiduals.clear();
	if (!resident)
		for (FrameHessian* fh : frameHessians)
			for (PointHessian* ph : fh->pointHessians)
				for (PointFrameResidual* r : ph->residuals)
				{
					activeResiduals.push_back(r);
					r->resetOOB();
				}
	FlatWindow FW;
	const auto tq0 = std::chrono::steady_clock::now();
	bool ok = uploadWindow(this, FW, true, resident);   // linearised residuals only exist between flagPointsForRemoval and marginalizePointsF: the library refuses a window that holds one
	const auto tq1 = std::chrono::steady_clock::now();
	const int N = FW.N, R = FW.R;
	std::vector<PointHessian*>& points = FW.points;
	if (ok && !resident && (size_t)R != activeResiduals.size()) { fprintf(stderr, "[dropin] residual lists disagree\n"); abort(); }
	if (ok && resident && R != ef->nResiduals) { fprintf(stderr, "[dropin] residual counts disagree\n"); abort(); }
	dmvio_hip_ba* ba = g.ba;
	std::vector<float> th(F);
	// ---- the Gauss-Newton loop + final fix-linearisation (:450-609)
	float rmse = 0; double finalEnergy = 0; int iterations = 0;
	if (setting_useGTSAMIntegration)
	{
		// the reference's default solver branch: every solve, the marginalisation / factor energies and the loop's break / accept notifications go through BAGTSAMIntegration
		// (optimizeVio above).  The hooks write the iteration's keyframe states into the FrameHessians (as the reference's own loop would have); in shadow mode they are put
		// back, with the facade's state, for the reference's own optimize that follows
		std::vector<Vec10> st0; VecC c0 = Hcalib.value;
		void* saved = nullptr;
		if (g.shadow) { for (FrameHessian* fh : frameHessians) st0.push_back(fh->get_state()); saved = g.vio_save ? g.vio_save(g.vio_sys) : nullptr; }
		ok = ok && optimizeVio(this, ba, mnumOptIts, &rmse, &finalEnergy, &iterations);
		if (g.shadow)
		{
			for (size_t f = 0; f < st0.size(); f++) frameHessians[f]->setState(st0[f]);
			Hcalib.setValue(c0);
			if (g.vio_restore) g.vio_restore(g.vio_sys, saved, 0);
		}
	}
	else ok = ok && HIP_OK(dmvio_hip_ba_optimize(ba, mnumOptIts, &rmse, &finalEnergy, &iterations, nullptr));
	const auto tq2 = std::chrono::steady_clock::now();
	if (!ok && g.shadow) return orig(this, mnumOptIts);
	if (!ok) { isLost = true; return 0; }   // a failure inside optimize reads as isLost (:613-617)
	if (g.shadow)
	{
		// the reference's own optimize on the same window; then its results against the device's
		std::vector<double> hp(7 * (size_t)F), ha(2 * (size_t)F);
		std::vector<float> hid(N), hstep(N);
		for (int f = 0; f < F; f++) { double st10[10]; HIP_OK(dmvio_hip_ba_get_frame(ba, f, &hp[7 * f], &ha[2 * f], st10)); }
		HIP_OK(dmvio_hip_ba_get_points(ba, hid.data(), hstep.data()));
		int resInA_hip = 0; HIP_OK(dmvio_hip_ba_get_res_in_a(ba, &resInA_hip));
		std::vector<float> hhess(N); HIP_OK(dmvio_hip_ba_get_point_hessian(ba, hhess.data()));
		g.sh.solve_calls = 0;
		const float r0 = orig(this, mnumOptIts);
		g.sh.n_opt++;
		if (g.sh.solve_calls != iterations) { g.sh.n_opt_iter_diff++; if (getenv("DROPIN_DEBUG")) fprintf(stderr, "[dropin] optimize: reference %ld iterations, device %d\n", g.sh.solve_calls, iterations); }
		g.deviceHessian.clear();
		for (int pi = 0; pi < N; pi++)
		{
			g.deviceHessian[points[pi]] = hhess[pi];
			const double href = points[pi]->idepth_hessian;
			g.sh.opt_hessian_rel = std::max(g.sh.opt_hessian_rel, std::fabs(href - hhess[pi]) / std::max(1.0, std::fabs(href)));
		}
		g.sh.opt_rmse_rel = std::max(g.sh.opt_rmse_rel, (double)std::fabs(rmse - r0) / r0);
		const double Eref = (double)r0 * r0 * patternNum * ef->resInA;
		g.sh.opt_energy_rel = std::max(g.sh.opt_energy_rel, std::fabs(finalEnergy - Eref) / Eref);
		if (resInA_hip != ef->resInA) g.sh.n_resInA_diff++;
		for (int f = 0; f < F; f++)
		{
			double p7[7]; toPose7(frameHessians[f]->PRE_worldToCam, p7);
			for (int i = 0; i < 3; i++) g.sh.opt_pose = std::max(g.sh.opt_pose, std::fabs(p7[i] - hp[7 * f + i]));
			g.sh.opt_aff = std::max(g.sh.opt_aff, std::max(std::fabs(frameHessians[f]->aff_g2l().a - ha[2 * f]), std::fabs(frameHessians[f]->aff_g2l().b - ha[2 * f + 1]) / 100.0));
		}
		std::vector<double> rel(N);
		for (int pi = 0; pi < N; pi++) rel[pi] = std::fabs(points[pi]->idepth - hid[pi]) / std::max(1e-3f, std::fabs(points[pi]->idepth));
		std::nth_element(rel.begin(), rel.begin() + N / 2, rel.end());
		g.sh.opt_idepth_med = std::max(g.sh.opt_idepth_med, rel[N / 2]);
		return r0;
	}
	auto wprev = std::chrono::steady_clock::now();
	auto wlap = [&](int k) { const auto t = std::chrono::steady_clock::now(); g.wb_split[k] += std::chrono::duration<double>(t - wprev).count(); wprev = t; };
	// ---- write-back: calibration, keyframe states (the newest one re-anchored like :596-603), thresholds
	{
		double value[4], value_zero[4];
		HIP_OK(dmvio_hip_ba_get_calib_values(ba, value, value_zero));
		VecC v; for (int i = 0; i < 4; i++) v[i] = value[i];
		Hcalib.setValue(v);
	}
	for (int f = 0; f < F; f++)
	{
		double pose7[7], aff[2], st10[10];
		HIP_OK(dmvio_hip_ba_get_frame(ba, f, pose7, aff, st10));
		Vec10 st; for (int i = 0; i < 10; i++) st[i] = st10[i];
		if (f < F - 1) frameHessians[f]->setState(st);
		else frameHessians[f]->setEvalPT(fromPose7(pose7), st);   // newStateZero = (0, .., a, b, 0, 0) at the optimised pose
	}
	HIP_OK(dmvio_hip_ba_get_frame_energy_th(ba, th.data()));
	for (int f = 0; f < F; f++) frameHessians[f]->frameEnergyTH = th[f];
	EFDeltaValid = false; EFAdjointsValid = false;
	ef->setAdjointsF(&Hcalib);
	setPrecalcValues();
	// ---- points: idepth (= idepth_zero, doStepFromBackup :283-291), and what the LAST solveSystemF's accumulation left in PointHessian / EFPoint
	// (idepth_hessian, HdiF: AccumulatedSCHessian.cpp:42-54; read by flagPointsForRemoval and makeCoarseDepthL0);
	// ---- residuals: what linearize + applyRes(true) of the final linearizeAll(true) leave behind (Residuals.cpp:78-328), then linearizeAll's own tail (:176-212).
	// One pass, point by point: a point's residuals are visited in PointHessian::residuals order (the order activeResiduals lists them in), so the removals leave
	// EFPoint::residualsAll in the order the reference's loop leaves it; across points the order is immaterial (dropResidual touches the point's own list and counters).
	{
		std::vector<float> idepth(N), step(N), hess(N), Hdd(N), bd(N), Hcd(4 * (size_t)N), HdiF(N), bdSumF(N);
		HIP_OK(dmvio_hip_ba_get_points(ba, idepth.data(), step.data()));
		HIP_OK(dmvio_hip_ba_get_point_hessian(ba, hess.data()));
		HIP_OK(dmvio_hip_ba_get_point_acc(ba, Hdd.data(), bd.data(), Hcd.data(), HdiF.data(), bdSumF.data()));
		wlap(0);
		if (resident) HIP_OK(dmvio_hip_graph_set_idepths(g.graph, N, idepth.data()));   // the mirror's values follow the optimisation; its structure follows the dropResidual calls below
		std::vector<unsigned char> newState(R), active(R);
		std::vector<float> newEnergy(R), newEnergyWO(R), center(3 * (size_t)R);
		HIP_OK(dmvio_hip_ba_get_res_state(ba, newState.data(), newEnergy.data(), newEnergyWO.data(), active.data(), center.data()));
		wlap(1);
		// the per-point part (21 ns per residual on one thread; dealt out over the reference's worker pool, FullSystem::treadReduce, it took LONGER — 0.32 vs 0.27 ms for 2000
		// points: waking six workers costs more than the loop); the removals (they go through EnergyFunctional::dropResidual) follow, points ascending
		std::vector<int> pending[1];
		auto perPoint = [&](int first, int end, Vec10* /*stats*/, int tid)
		{
			for (int pi = first; pi < end; pi++)
			{
				PointHessian* ph = points[pi];
				ph->idx = pi;   // where marginalizePointsF finds the point in the window the device holds
				ph->setIdepth(idepth[pi]); ph->setIdepthZero(idepth[pi]); ph->step = step[pi];
				ph->idepth_hessian = hess[pi];
				EFPoint* efp = ph->efPoint;
				efp->HdiF = HdiF[pi]; efp->bdSumF = bdSumF[pi]; efp->Hdd_accAF = Hdd[pi]; efp->bd_accAF = bd[pi];
				for (int k = 0; k < 4; k++) efp->Hcd_accAF[k] = Hcd[4 * pi + k];
				const int base = FW.pointFirstRes[pi];
				bool removes = false;
				for (PointFrameResidual* r : ph->residuals)
				{
					const int ri = base + r->efResidual->idxInAll;
					const ResState ns = newState[ri] == 0 ? ResState::IN : (newState[ri] == 1 ? ResState::OOB : ResState::OUTLIER);
					r->state_NewState = ns; r->state_NewEnergy = newEnergy[ri]; r->state_NewEnergyWithOutlier = newEnergyWO[ri];
					if (ns != ResState::OOB) r->centerProjectedTo = Vec3f(center[3 * ri], center[3 * ri + 1], center[3 * ri + 2]);
					r->efResidual->isActiveAndIsGoodNEW = active[ri] != 0;   // applyRes(true); the Jacobians stay on the device (marginalisation relinearises its points itself, FullSystem.cpp:836-849)
					r->setState(ns);
					r->state_energy = newEnergy[ri];
					if (r->efResidual->isActive())
					{
						if (r->isNew)
						{
							Vec3f ptp_inf = r->host->targetPrecalc[r->target->idx].PRE_KRKiTll * Vec3f(ph->u, ph->v, 1);
							Vec3f ptp = ptp_inf + r->host->targetPrecalc[r->target->idx].PRE_KtTll * ph->idepth_scaled;
							float relBS = 0.01 * ((ptp_inf.head<2>() / ptp_inf[2]) - (ptp.head<2>() / ptp[2])).norm();
							if (relBS > ph->maxRelBaseline) ph->maxRelBaseline = relBS;
							ph->numGoodResiduals++;
						}
					}
					else removes = true;
					if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].second = r->state_state;
					else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].second = r->state_state;
				}
				if (removes) pending[tid].push_back(pi);
			}
		};
		perPoint(0, N, nullptr, 0);
		wlap(2);
		std::vector<int> withRemovals;
		withRemovals = pending[0];
		std::vector<PointFrameResidual*> toRemove;
		for (int pi : withRemovals)
		{
			PointHessian* ph = points[pi];
			toRemove.clear();
			for (PointFrameResidual* r : ph->residuals) if (!r->efResidual->isActive()) toRemove.push_back(r);
			for (PointFrameResidual* r : toRemove)
			{
				if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = 0;
				else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = 0;
				for (unsigned int k = 0; k < ph->residuals.size(); k++)
					if (ph->residuals[k] == r)
					{
						ef->dropResidual(r->efResidual);
						deleteOut<PointFrameResidual>(ph->residuals, k);
						break;
					}
			}
		}
	}
	wlap(3);
	// ---- tail of optimize (:611-645)
	HIP_OK(dmvio_hip_ba_get_res_in_a(ba, &ef->resInA));   // the count the last solveSystemF's accumulation left behind (EnergyFunctional.cpp:209)
	if (!std::isfinite(finalEnergy)) { std::cout << "Tracking lost after bundle adjustment!" << std::endl; isLost = true; }
	statistics_lastFineTrackRMSE = rmse;
	{
		boost::unique_lock<boost::mutex> crlock(shellPoseMutex);
		for (FrameHessian* fh : frameHessians)
		{
			fh->shell->camToWorld = fh->PRE_camToWorld;
			fh->shell->aff_g2l = fh->aff_g2l();
		}
	}
	wlap(4);
	g.window_current = resident;   // marginalizePointsF of this keyframe finds the optimised window on the device
	if (resident) g.windowPoints = points;
	const auto tq3 = std::chrono::steady_clock::now();
	g.opt_split[0] += std::chrono::duration<double>(tq1 - tq0).count(); g.opt_split[1] += std::chrono::duration<double>(tq2 - tq1).count();
	g.opt_split[2] += std::chrono::duration<double>(tq3 - tq2).count();
	return statistics_lastFineTrackRMSE;
}

}  // namespace dso
