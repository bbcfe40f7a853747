// include/compat/include/depth_estimate.h — see ../depth_estimate.h (the runner spells the path "include/depth_estimate.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
