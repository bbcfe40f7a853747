// include/compat/include/lm_optimizer.h — see ../lm_optimizer.h (the runner spells the path "include/lm_optimizer.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
