// include/compat/include/image_processing_global.h — see ../image_processing_global.h (the runner spells the path "include/image_processing_global.h", ref: run_odometry_kitti_offline.cpp:14-19).
#pragma once
#include "odometry_shim.hpp"
