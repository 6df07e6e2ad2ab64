// ORACLE-SIDE TEST INFRASTRUCTURE — stub of the reference's IMU / GTSAM facade (src/IMU/IMUIntegration.hpp:64-213,
// src/GTSAMIntegration/BAGTSAMIntegration.h), which is OUT OF SCOPE and needs GTSAM 4.2a6 (absent).
// Only the members the compiled hot-path sources call exist.  The two hand-off points — computeCoarseUpdate(H,b,...)
// (CoarseTracker.cpp:620) and computeBAUpdate(H,b,...) (EnergyFunctional.cpp:967) — forward to std::function hooks so a test
// can drive the reference's VIO branch with any solver it likes and observe the dense systems the reference hands over.
#pragma once
#include <deque>
#include <functional>
#include <memory>
#include <vector>
#include "sophus/se3.hpp"
#include "util/NumType.h"
#include "util/GTData.hpp"
#include "OptimizationBackend/EnergyFunctional.h"  // as the real header does (CoarseInitializer.h relies on it for IndexThreadReduce)
#include "util/FrameShell.h"

namespace dso { class CalibHessian; class EFFrame; class FrameShell; }

namespace dmvio
{
struct IMUCalibration {};
struct IMUData {};
struct IMUSettings
{
	bool updateDynamicWeightDuringOptimization = false;
	int numMeasurementsGravityInit = 40;
	double maxTimeBetweenInitFrames = 100000.0;
};
// FullSystem::printResult (FullSystem.cpp:285) only touches this when an IMU scale is used
struct TransformDSOToIMU
{
	Eigen::Matrix<double, 4, 4> transformPose(const Eigen::Matrix<double, 4, 4>& m) const { return m; }
};

class BAGTSAMIntegration
{
public:
	// hooks: every member of the real class that EnergyFunctional / FullSystem::optimize call on the default (setting_useGTSAMIntegration) branch can be observed / driven
	// by a test (src/GTSAMIntegration/BAGTSAMIntegration.cpp:97-120 updateBAValues, :124-251 computeBAUpdate, :358 acceptBAUpdate, :442 getBAEnergy, :453 canBreak,
	// :508 postOptimization, :526 updateDynamicWeight).  Unset hooks behave like a graph without factors.
	std::function<dso::VecX(const dso::MatXX&, const dso::VecX&, double, const dso::MatXX&)> computeBAUpdateHook;
	std::function<dso::VecX(const dso::MatXX&, const dso::VecX&, double, std::vector<dso::EFFrame*>&, const dso::MatXX&)> computeBAUpdateFramesHook;
	std::function<double(double, double, bool)> updateDynamicWeightHook;
	std::function<void(double)> acceptBAUpdateHook;
	std::function<void(std::vector<dso::EFFrame*>&)> updateBAValuesHook, postOptimizationHook;
	std::function<double(bool)> getBAEnergyHook;
	std::function<bool()> canBreakHook;
	bool canBreakValue = false;  // BAGTSAMIntegration.h:225: canBreakOptimization starts false and only computeBAUpdate sets it

	double updateDynamicWeight(double e, double rmse, bool good) { return updateDynamicWeightHook ? updateDynamicWeightHook(e, rmse, good) : 1.0; }
	bool canBreak() { return canBreakHook ? canBreakHook() : canBreakValue; }
	void acceptBAUpdate(double e) { if (acceptBAUpdateHook) acceptBAUpdateHook(e); }
	void postOptimization(std::vector<dso::EFFrame*>& f) { if (postOptimizationHook) postOptimizationHook(f); }
	void updateBAValues(std::vector<dso::EFFrame*>& f) { if (updateBAValuesHook) updateBAValuesHook(f); }
	double getBAEnergy(bool useNew) { return getBAEnergyHook ? getBAEnergyHook(useNew) : 0.0; }
	// graph maintenance around the optimisation (FullSystem.cpp:1402-1406, EnergyFunctional.cpp:538-566): observable, no-ops without a hook
	std::function<void(const dso::MatXX&, const dso::VecX&, std::vector<dso::EFFrame*>&)> addMarginalizedPointsBAHook;
	std::function<void(dso::EFFrame*)> marginalizeBAFrameHook;
	std::function<void(int, std::vector<dso::EFFrame*>&)> addKeyframeToBAHook;
	void addMarginalizedPointsBA(const dso::MatXX& H, const dso::VecX& b, std::vector<dso::EFFrame*>& f) { if (addMarginalizedPointsBAHook) addMarginalizedPointsBAHook(H, b, f); }
	void addPriorBA(dso::EFFrame*, const dso::MatXX&, const dso::VecX&) {}
	template<typename A, typename B> void addPriorBA(dso::EFFrame*, const A&, const B&) {}
	void marginalizeBAFrame(dso::EFFrame* f) { if (marginalizeBAFrameHook) marginalizeBAFrameHook(f); }
	void addKeyframeToBA(int id, const Sophus::SE3d&, std::vector<dso::EFFrame*>& f) { if (addKeyframeToBAHook) addKeyframeToBAHook(id, f); }
	void updateBAOrdering(std::vector<dso::EFFrame*>&) {}
	void addFirstBAFrame(int) {}
	dso::VecX computeBAUpdate(const dso::MatXX& H, const dso::VecX& b, double lambda, std::vector<dso::EFFrame*>& frames, const dso::MatXX& HNoLambda)
	{
		if (computeBAUpdateFramesHook) return computeBAUpdateFramesHook(H, b, lambda, frames, HNoLambda);
		return computeBAUpdateHook(H, b, lambda, HNoLambda);
	}
};

class IMUIntegration
{
public:
	IMUIntegration() {}
	IMUIntegration(dso::CalibHessian*, const IMUCalibration&, IMUSettings& s, bool) : settings_(&s) {}

	// hooks
	bool coarseInitialized = false;
	std::function<Sophus::SE3d(const dso::Mat88&, const dso::Vec8&, float, float, double&, double&, double&)> computeCoarseUpdateHook;
	std::function<void()> acceptCoarseUpdateHook;
	std::function<void(const dso::Mat88&, const dso::Vec8&, bool)> addVisualToCoarseGraphHook;
	float energyThCap = -1.f;

	bool isCoarseInitialized() { return coarseInitialized; }
	Sophus::SE3d computeCoarseUpdate(const dso::Mat88& H, const dso::Vec8& b, float extrapFac, float lambda, double& incA, double& incB, double& incNorm)
	{
		return computeCoarseUpdateHook(H, b, extrapFac, lambda, incA, incB, incNorm);
	}
	void acceptCoarseUpdate() { if (acceptCoarseUpdateHook) acceptCoarseUpdateHook(); }
	void addVisualToCoarseGraph(const dso::Mat88& H, const dso::Vec8& b, bool good) { if (addVisualToCoarseGraphHook) addVisualToCoarseGraphHook(H, b, good); }
	IMUSettings& getImuSettings() const { return settings_ ? *settings_ : const_cast<IMUSettings&>(own_); }
	// IMUIntegration.cpp: caps the threshold once the IMU is initialised; without IMU data it leaves it untouched
	void newFrameEnergyTH(float& th) { if (energyThCap > 0 && th > energyThCap) th = energyThCap; }
	Sophus::SE3 TS_cam_imu;

	// the rest of the facade FullSystem.cpp talks to.  The keyframe bookkeeping is the real class's (src/IMU/IMUIntegration.cpp:107-125 initCoarseGraph, :228-247
	// prepareKeyframe, :282-308 postOptimization / finishKeyframeOptimization, :320-341): a LIVE run of FullSystem with setting_useIMU walks through it
	// (FullSystem.cpp:986-1016, 1130-1200, 1442-1455).  What stands in for the IMU itself: `initAfterKeyframes` >= 0 declares the (absent) IMU initialiser finished after that
	// many keyframe optimisations — from then on finishKeyframeOptimization hands "information BA -> coarse" over, the next initCoarseGraph sets coarseInitialized, and
	// trackNewestCoarse takes its computeCoarseUpdate branch (CoarseTracker.cpp:612); addIMUData's pose hint and finishCoarseTracking are hooks.
	int initAfterKeyframes = -1;          // < 0: never (the visual-only runs: no IMU data ever arrives)
	int keyframesOptimized = 0;
	bool baInitialized = false, initializedBeforePostOptimization = false, haveInformationBAToCoarse = false;
	int preparedKeyframe = -1;
	bool preparedKFCreated = false;
	std::function<Sophus::SE3(int, double, bool, int)> addIMUDataHook;
	std::function<void(const dso::FrameShell&, bool)> finishCoarseTrackingHook;
	std::function<void(int)> initCoarseGraphHook;
	const std::unique_ptr<BAGTSAMIntegration>& getBAGTSAMIntegration() const { return ba_; }
	TransformDSOToIMU& getTransformDSOToIMU() { return transform_; }
	double getCoarseScale() { return 1.0; }
	void addIMUDataToBA(const IMUData&) {}
	void setGTData(GTData*, int) {}
	int getPreparedKeyframe() const { return preparedKeyframe; }
	bool isPreparedKFCreated() const { return preparedKFCreated; }
	Sophus::SE3d initCoarseGraph()
	{
		const int keyframeId = preparedKeyframe;
		preparedKeyframe = -1;
		if (!haveInformationBAToCoarse) return Sophus::SE3d();
		haveInformationBAToCoarse = false;
		coarseInitialized = true;
		if (initCoarseGraphHook) initCoarseGraphHook(keyframeId);
		return Sophus::SE3d();
	}
	Sophus::SE3 addIMUData(const IMUData&, int frameId, double timestamp, bool trackingRefChanged, int lastFrameId, bool = false)
	{
		return addIMUDataHook ? addIMUDataHook(frameId, timestamp, trackingRefChanged, lastFrameId) : Sophus::SE3();
	}
	void finishCoarseTracking(const dso::FrameShell& shell, bool willBecomeKeyframe) { if (finishCoarseTrackingHook) finishCoarseTrackingHook(shell, willBecomeKeyframe); }
	void prepareKeyframe(int frameId) { preparedKeyframe = frameId; preparedKFCreated = false; }
	void keyframeCreated(int) { preparedKFCreated = true; }
	void skipPreparedKeyframe() { preparedKeyframe = -1; }
	void postOptimization(int)
	{
		initializedBeforePostOptimization = baInitialized;
		keyframesOptimized++;
		if (initAfterKeyframes >= 0 && keyframesOptimized >= initAfterKeyframes) baInitialized = true;
	}
	bool finishKeyframeOptimization(int)
	{
		if (!initializedBeforePostOptimization) return false;
		haveInformationBAToCoarse = true;
		return true;
	}
	void finishKeyframeOperations(int) {}
	void resetBAPreintegration() {}
private:
	std::unique_ptr<BAGTSAMIntegration> ba_{new BAGTSAMIntegration()};
	TransformDSOToIMU transform_;
	IMUSettings* settings_ = nullptr;
	IMUSettings own_;
};
}
