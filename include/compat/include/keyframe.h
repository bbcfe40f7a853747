// include/compat/include/keyframe.h — see ../keyframe.h (the runner spells the path "include/keyframe.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
