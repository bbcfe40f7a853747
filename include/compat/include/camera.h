// include/compat/include/camera.h — see ../camera.h (the runner spells the path "include/camera.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
