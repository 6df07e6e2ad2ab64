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
	void addMarginalizedPointsBA(const dso::MatXX&, const dso::VecX&, std::vector<dso::EFFrame*>&) {}
	void addPriorBA(dso::EFFrame*, const dso::MatXX&, const dso::VecX&) {}
	template<typename A, typename B> void addPriorBA(dso::EFFrame*, const A&, const B&) {}
	void marginalizeBAFrame(dso::EFFrame*) {}
	void addKeyframeToBA(int, const Sophus::SE3d&, std::vector<dso::EFFrame*>&) {}
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

	// the rest of the facade FullSystem.cpp talks to: no IMU data ever arrives in the visual-only runs the tests make
	const std::unique_ptr<BAGTSAMIntegration>& getBAGTSAMIntegration() const { return ba_; }
	TransformDSOToIMU& getTransformDSOToIMU() { return transform_; }
	double getCoarseScale() { return 1.0; }
	void addIMUDataToBA(const IMUData&) {}
	void setGTData(GTData*, int) {}
	int getPreparedKeyframe() const { return -1; }
	bool isPreparedKFCreated() const { return false; }
	Sophus::SE3d initCoarseGraph() { return Sophus::SE3d(); }
	Sophus::SE3 addIMUData(const IMUData&, int, double, bool, int, bool = false) { return Sophus::SE3(); }
	void finishCoarseTracking(const dso::FrameShell&, bool) {}
	void prepareKeyframe(int) {}
	void keyframeCreated(int) {}
	void skipPreparedKeyframe() {}
	void postOptimization(int) {}
	bool finishKeyframeOptimization(int) { return false; }
	void finishKeyframeOperations(int) {}
	void resetBAPreintegration() {}
private:
	std::unique_ptr<BAGTSAMIntegration> ba_{new BAGTSAMIntegration()};
	TransformDSOToIMU transform_;
	IMUSettings* settings_ = nullptr;
	IMUSettings own_;
};
}
