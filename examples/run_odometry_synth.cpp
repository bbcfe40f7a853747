// run_odometry_synth.cpp — the reference runner's frame loop (run_odometry_kitti_offline.cpp:58-145, 198-271) written
// against include/odometry_shim.hpp: the same classes, constructor arguments and call order, with synthetic frames
// read from a raw file instead of KITTI PNGs (this image has neither OpenCV nor the dataset).
//
//   python examples/make_frames.py frames.bin 12          # left/right fp32 frames, 1241x376
//   g++ -O2 -std=c++17 -Iinclude examples/run_odometry_synth.cpp -o run_odometry_synth
//       -Lodometry_amd/lib -lodometry_hip -Wl,-rpath,$PWD/odometry_amd/lib      (one command line)
//   ./run_odometry_synth frames.bin
#include <cmath>
#include <cstdio>
#include <tuple>
#include <vector>

#include "odometry_shim.hpp"

using namespace odometry;

static bool read_frames(const char* path, std::vector<Mat>& left, std::vector<Mat>& right) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  int hdr[3];
  if (std::fread(hdr, sizeof(int), 3, f) != 3) { std::fclose(f); return false; }
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  for (int i = 0; i < n; i++) {
    Mat l(rows, cols, PixelType), r(rows, cols, PixelType);
    if (std::fread(l.ptr<float>(), sizeof(float), (size_t)rows * cols, f) != (size_t)rows * cols) break;
    if (std::fread(r.ptr<float>(), sizeof(float), (size_t)rows * cols, f) != (size_t)rows * cols) break;
    left.push_back(l);
    right.push_back(r);
  }
  std::fclose(f);
  return (int)left.size() == n;
}

// 4x4 rigid inverse / product on the column-major Affine4f (the runner uses Eigen for these, :218).
static Affine4f rigid_inverse(const Affine4f& T) {
  Affine4f o = Affine4f::Identity();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o(i, j) = T(j, i);
  for (int i = 0; i < 3; i++) o(i, 3) = -(o(i, 0) * T(0, 3) + o(i, 1) * T(1, 3) + o(i, 2) * T(2, 3));
  return o;
}
static Affine4f mul(const Affine4f& A, const Affine4f& B) {
  Affine4f C;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j) + A(i, 3) * B(3, j);
  return C;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: %s frames.bin\n", argv[0]); return 2; }
  std::vector<Mat> left, right;
  if (!read_frames(argv[1], left, right) || left.empty()) { std::printf("cannot read %s\n", argv[1]); return 2; }
  const unsigned num_frames = (unsigned)left.size();
  const unsigned num_pyramid = 4;
  const float baseline = 386.1448f / 718.856f;                          // :41
  std::shared_ptr<CameraPyramid> left_cam_ptr = nullptr, right_cam_ptr = nullptr;  // :51-52

  DepthEstimator depth_estimator(8.0f, 900.0f, 15.0f, 0.1f, 30.0f, 0.01f, 28.0f, 0.995f, 50, 4, left_cam_ptr, right_cam_ptr,
                                 baseline, 80000);                      // :58-70
  std::vector<int> pose_max_iters = {10, 20, 30, 30};                   // :76
  Affine4f init_relative_affine = Affine4f::Identity();
  LevenbergMarquardtOptimizer pose_estimator(0.01f, 0.995f, pose_max_iters, init_relative_affine, left_cam_ptr, 1, 28.0f);  // :88

  Affine4f cur_pose = Affine4f::Identity(), pose_to_keyframe = cur_pose;
  Mat pre_left_val(left[0].rows, left[0].cols, CV_8U, 0), pre_left_disp(left[0].rows, left[0].cols, PixelType, 0),
      pre_left_dep(left[0].rows, left[0].cols, PixelType, 0);
  if (depth_estimator.ComputeDepth(left[0], right[0], pre_left_val, pre_left_disp, pre_left_dep) == -1) {  // :102
    std::cout << "Init 0-th frame failed!" << std::endl;
    return 1;
  }
  depth_estimator.ReportStatus();
  auto* pre_img_pyramid_ptr = new ImagePyramid(num_pyramid, left[0], true);       // :130
  auto* pre_dep_pyramid_ptr = new DepthPyramid(num_pyramid, pre_left_dep, false); // :131

  std::vector<std::tuple<ImagePyramid, DepthPyramid, Mat>> keyframes;   // :138-141 (the runner "simulates" keyframes)
  std::vector<Affine4f> keyframe_poses_abs;
  unsigned current_kf = 0;
  keyframes.emplace_back(std::make_tuple(*pre_img_pyramid_ptr, *pre_dep_pyramid_ptr, pre_left_val));
  keyframe_poses_abs.emplace_back(cur_pose);
  const float w[6] = {0.1f / 3.3f, 1.0f / 3.3f, 0.1f / 3.3f, 1.0f / 3.3f, 0.1f / 3.3f, 1.0f / 3.3f};  // :144-145

  for (unsigned frame_id = 1; frame_id < num_frames; frame_id++) {     // :198
    ImagePyramid cur_img_pyramid(num_pyramid, left[frame_id], true);   // :205
    pose_to_keyframe = pose_estimator.Solve(std::get<0>(keyframes[current_kf]), std::get<1>(keyframes[current_kf]),
                                            cur_img_pyramid);          // :215
    cur_pose = mul(keyframe_poses_abs[current_kf], rigid_inverse(pose_to_keyframe));  // :218

    Mat cur_left_val(left[0].rows, left[0].cols, CV_8U, 0), cur_left_disp(left[0].rows, left[0].cols, PixelType),
        cur_left_dep(left[0].rows, left[0].cols, PixelType);
    if (depth_estimator.ComputeDepth(left[frame_id], right[frame_id], cur_left_val, cur_left_disp, cur_left_dep) == -1) {
      std::cout << "    depth failed!" << std::endl;                   // :230-232
      break;
    }
    delete pre_img_pyramid_ptr;
    delete pre_dep_pyramid_ptr;
    pre_img_pyramid_ptr = new ImagePyramid(num_pyramid, left[frame_id], true);       // :251
    pre_dep_pyramid_ptr = new DepthPyramid(num_pyramid, cur_left_dep, false);        // :252

    // :253-257 weighted motion; angles as Sophus SO3::angleX/Y/Z reduce to for a rotation matrix
    const Affine4f& T = pose_to_keyframe;
    const float ax = std::atan2(T(2, 1) - T(1, 2), T(1, 1) + T(2, 2));
    const float ay = std::atan2(T(0, 2) - T(2, 0), T(0, 0) + T(2, 2));
    const float az = std::atan2(T(1, 0) - T(0, 1), T(0, 0) + T(1, 1));
    const float mot[6] = {std::fabs(ax), std::fabs(ay), std::fabs(az), std::fabs(T(0, 3)), std::fabs(T(1, 3)), std::fabs(T(2, 3))};
    float motion_mag = 0.0f;
    for (int i = 0; i < 6; i++) motion_mag += mot[i] * w[i];
    if (motion_mag > 1.1f) {                                          // :258
      keyframes.emplace_back(*pre_img_pyramid_ptr, *pre_dep_pyramid_ptr, cur_left_val);
      keyframe_poses_abs.emplace_back(cur_pose);
      current_kf++;
    }
    pose_estimator.Reset(pose_to_keyframe, 0.01f);                    // :261 / :268
    std::printf("frame %u kf %u motion %.4f  t = [% .5f % .5f % .5f]\n", frame_id, current_kf, motion_mag, cur_pose(0, 3),
                cur_pose(1, 3), cur_pose(2, 3));
  }
  delete pre_img_pyramid_ptr;
  delete pre_dep_pyramid_ptr;
  std::cout << "Total keyframes: " << current_kf << std::endl;
  return 0;
}
