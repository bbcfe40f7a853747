// include/compat/image_pyramid.h — forwarding header of the drop-in build: the reference's callers include "image_pyramid.h" / "include/image_pyramid.h"
// (ref: run_odometry_kitti_offline.cpp:14-19, test_disparity.cpp:12-13, include/image_pyramid.h); with -I<repo>/include/compat -I<repo>/include
// in front of the reference's own include directories they get the MI355X classes instead, without an edit.
#pragma once
#include "odometry_shim.hpp"
