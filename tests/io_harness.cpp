// tests/io_harness.cpp — exposes include/odometry_io.hpp to the Python tests (compiled with g++ on the fly).
#include "../include/odometry_io.hpp"

using namespace odometry::io;

extern "C" {
int io_read_png(const char* path, unsigned char* out, int cap, int* w, int* h) {
  std::vector<uint8_t> px;
  if (!read_png_gray8(path, px, *w, *h)) return -1;
  if ((int)px.size() > cap) return -2;
  std::memcpy(out, px.data(), px.size());
  return 0;
}
int io_pose_roundtrip(const char* in_path, const char* out_path, int max_frames, float* first_last /*24*/, float* mean_err) {
  std::vector<Pose34> p;
  if (!load_gt_poses(in_path, p, (size_t)max_frames)) return -1;
  if (!save_poses_kitti(out_path, p)) return -2;
  std::memcpy(first_last, p.front().m, sizeof(float) * 12);
  std::memcpy(first_last + 12, p.back().m, sizeof(float) * 12);
  std::vector<Pose34> shifted = p;
  for (size_t i = 0; i < shifted.size(); i++) { shifted[i].m[3] += 3.0f; shifted[i].m[11] -= 4.0f; }
  *mean_err = eval_translation_error(p, shifted, p.size());
  return (int)p.size();
}
void io_image_path(const char* root, const char* seq, int cam, int frame, char* out, int cap) {
  std::snprintf(out, cap, "%s", kitti_image_path(root, seq, cam, frame).c_str());
}
}

extern "C" int io_read_calibration(const char* path, double* out /* 8 + 8 + 4 + 9 + 3 + 2 = 34 */) {
  StereoCalibration c;
  if (!read_stereo_calibration_file(path, c)) return -1;
  int k = 0;
  for (int cam = 0; cam < 2; cam++) for (int i = 0; i < 4; i++) out[k++] = c.intrinsics[cam][i];
  for (int cam = 0; cam < 2; cam++) for (int i = 0; i < 4; i++) out[k++] = c.distortion[cam][i];
  for (int cam = 0; cam < 2; cam++) for (int i = 0; i < 2; i++) out[k++] = c.sensor_size[cam][i];
  for (int i = 0; i < 9; i++) out[k++] = c.rotate_left_right[i];
  for (int i = 0; i < 3; i++) out[k++] = c.translate_left_right[i];
  out[k++] = c.resolution[0];
  out[k++] = c.resolution[1];
  return 0;
}
