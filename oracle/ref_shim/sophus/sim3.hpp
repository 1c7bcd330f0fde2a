// oracle/ref_shim/sophus/sim3.hpp -- see se3.hpp
#pragma once
#include "se3.hpp"
