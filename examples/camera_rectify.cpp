// examples/camera_rectify.cpp — the CameraPyramid half of the reference's stereo set-up, against the shim:
// read the calibration file (ref: src/camera.cpp:170-352), construct the two cameras (ref: :130-133), configure each
// with a rectifying rotation and a new 3x4 projection (ref: :147-148 — cv::stereoRectify, the one-off OpenCV calibration
// step that produces them, is the caller's: here they come from the command line), then undistort + rectify a raw
// 480x640 frame per camera (ref: :71-82) and print the per-level rectified intrinsics.
//
//   camera_rectify <camchain.yaml> <RP.txt: 2 x (9 + 12) doubles> <frames.bin: 2 x 480*640 float32> <out.bin>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "odometry_io.hpp"
#include "odometry_shim.hpp"

int main(int argc, char** argv) {
  if (argc < 5) {
    std::cout << "usage: camera_rectify <camchain.yaml> <RP.txt> <frames.bin> <out.bin>" << std::endl;
    return 2;
  }
  odometry::io::StereoCalibration cal;
  if (!odometry::io::read_stereo_calibration_file(argv[1], cal)) {
    std::cout << "open calibration file failed!" << std::endl;
    return 1;
  }
  std::ifstream rp(argv[2]);
  std::ifstream frames(argv[3], std::ios::binary);
  std::ofstream out(argv[4], std::ios::binary);
  const int levels = 4;
  for (int cam = 0; cam < 2; cam++) {
    const double* k = cal.intrinsics[cam];
    const double* d = cal.distortion[cam];
    auto cam_ptr = std::make_shared<odometry::CameraPyramid>(levels, k[0], k[1], 0.0, k[2], k[3], d[0], d[1], d[2], d[3],
                                                             cal.sensor_size[cam][0], cal.sensor_size[cam][1],
                                                             cal.resolution[0], cal.resolution[1]);
    odometry::Mat R(3, 3, odometry::CV_64F, 0.0), P(3, 4, odometry::CV_64F, 0.0);
    for (int i = 0; i < 9; i++) rp >> R.at<double>(i / 3, i % 3);
    for (int i = 0; i < 12; i++) rp >> P.at<double>(i / 4, i % 4);
    cam_ptr->ConfigureCamera(R, P, odometry::Size(640, 480));
    odometry::Mat raw(480, 640, PixelType), rect;
    frames.read(reinterpret_cast<char*>(raw.ptr<float>()), sizeof(float) * 480 * 640);
    if (cam_ptr->UndistortRectify(raw, rect) != 0) return 1;
    out.write(reinterpret_cast<const char*>(rect.ptr<float>()), sizeof(float) * rect.rows * rect.cols);
    for (int l = 0; l < levels; l++)
      std::printf("cam%d level %d fx %.9f fy %.9f cx %.9f cy %.9f f_m %.9f\n", cam, l, cam_ptr->fx_double(l), cam_ptr->fy_double(l),
                  cam_ptr->cx_double(l), cam_ptr->cy_double(l), cam_ptr->f_meters_double(l));
    odometry::Mat small(100, 100, PixelType), dummy;
    if (cam == 0 && cam_ptr->UndistortRectify(small, dummy) != -1) return 3;  // the reference's size check
  }
  std::cout << "stereo configuration done!" << std::endl;  // ref: src/camera.cpp:162
  return 0;
}
