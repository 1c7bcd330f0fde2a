// oracle/ref_shim/ros/ros.h -- TEST INFRASTRUCTURE.  ROS is not installed here.  FullSystem/FullSystem.h declares two kinds of ROS members
// (a NodeHandle and three Subscribers); none of the code compiled into oracle/_ref/libref.so touches them.
#pragma once
#include <iomanip>
#include <sstream>
#include <string>
namespace ros {
struct NodeHandle { NodeHandle() {} explicit NodeHandle(const std::string&) {} };
struct Subscriber {};
struct Publisher {};
struct Time { static Time now() { return Time(); } double toSec() const { return 0; } };
}
