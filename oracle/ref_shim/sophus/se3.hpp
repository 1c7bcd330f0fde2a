// oracle/ref_shim/sophus/se3.hpp -- TEST INFRASTRUCTURE, see ../Eigen/Core.  src/util/NumType.h only names these types in typedefs
// (typedef Sophus::SE3d SE3; ...); none of the code compiled for oracle/_ref uses them.
#pragma once
namespace Sophus {
struct SE3d {};
struct SO3d {};
struct Sim3d {};
}
