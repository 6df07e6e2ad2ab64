// ORACLE-SIDE TEST INFRASTRUCTURE — stub (out-of-scope IMU initialisation; only has to exist as a member type of FullSystem).
#pragma once
#include "IMU/IMUIntegration.hpp"
namespace dmvio
{
class GravityInitializer
{
public:
	GravityInitializer() {}
	GravityInitializer(int, const IMUCalibration&) {}
	Sophus::SE3d addMeasure(const IMUData&, const Sophus::SE3d&) { return Sophus::SE3d(); }
};
}
