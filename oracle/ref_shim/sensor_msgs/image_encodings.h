// oracle/ref_shim/sensor_msgs/image_encodings.h -- TEST INFRASTRUCTURE.  ROS is not installed here; FullSystem/FullSystem.h includes this header and the code compiled
// into oracle/_ref/libref.so uses nothing of it.
#pragma once
