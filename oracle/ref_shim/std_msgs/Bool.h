// oracle/ref_shim/std_msgs/Bool.h -- TEST INFRASTRUCTURE.  ROS is not installed here; FullSystem/FullSystem.h includes this header and the code compiled
// into oracle/_ref/libref.so uses nothing of it.
#pragma once
