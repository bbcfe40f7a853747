// include/compat/include/image_pyramid.h — see ../image_pyramid.h (the runner spells the path "include/image_pyramid.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
