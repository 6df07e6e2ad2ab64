// ORACLE-SIDE TEST INFRASTRUCTURE — stand-in (see se3.hpp).  Sim3 only has to exist as a type on the hot path
// (util/NumType.h:45 typedefs it); none of the compiled reference functions does arithmetic with it.
#pragma once
#include "se3.hpp"
namespace Sophus
{
class Sim3d
{
public:
	EIGEN_MAKE_ALIGNED_OPERATOR_NEW;
	Sim3d() : s_(1.0) {}
	double scale() const { return s_; }
	const SE3d& se3() const { return T_; }
private:
	SE3d T_;
	double s_;
};
typedef Sim3d Sim3;
}
