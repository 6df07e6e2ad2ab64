// ORACLE-SIDE TEST INFRASTRUCTURE.  "sophus/se3.hpp" of the reference's sources resolves here (ref_shim is first on the include path) and hands on to the
// reference's VENDORED Sophus v0.9a — thirdparty/Sophus/sophus/se3.hpp (+ so3.hpp, sophus.hpp), included unmodified from where it lies (REF_SOPHUS_DIR is set
// by oracle/Makefile.ref) — so SE3 / SO3 exp, log, products, inverse, Adj, constructors in oracle/_ref/libref.so are Sophus' own code.  Only Eigen underneath is
// a stand-in (Eigen/Core, Eigen/src/QuaternionStandin.h).  oracle/lie.h, the restatement the oracles use, is held against this code bit for bit by
// tests/test_ref_pin_cpu.py::test_lie_algebra_against_the_vendored_sophus.
#pragma once
#include "Eigen/Core"
#include "Eigen/Geometry"
#define DMV_STR2(x) #x
#define DMV_STR(x) DMV_STR2(x)
#include DMV_STR(REF_SOPHUS_DIR/se3.hpp)
