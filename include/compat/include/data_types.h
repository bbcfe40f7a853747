// include/compat/include/data_types.h — see ../data_types.h (the runner spells the path "include/data_types.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
