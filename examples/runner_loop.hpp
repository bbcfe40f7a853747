// runner_loop.hpp — the reference runner's frame loop (run_odometry_kitti_offline.cpp:58-145, 198-271) written against
// include/odometry_shim.hpp: same classes, constructor arguments and call order. Shared by the synthetic-data example
// and the KITTI example. Fills `pred` with the absolute 3x4 pose of every frame (row-major, frame 0 = `pose0`).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <tuple>
#include <vector>

#include "odometry_io.hpp"
#include "odometry_shim.hpp"

namespace odometry {

// 4x4 rigid inverse / product on the column-major Affine4f (the runner uses Eigen for these, :218).
inline Affine4f rigid_inverse(const Affine4f& T) {
  Affine4f o = Affine4f::Identity();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o(i, j) = T(j, i);
  for (int i = 0; i < 3; i++) o(i, 3) = -(o(i, 0) * T(0, 3) + o(i, 1) * T(1, 3) + o(i, 2) * T(2, 3));
  return o;
}
inline Affine4f mul(const Affine4f& A, const Affine4f& B) {
  Affine4f C;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j) + A(i, 3) * B(3, j);
  return C;
}


inline io::Pose34 to_pose34(const Affine4f& T) {
  io::Pose34 p;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) p.m[i * 4 + j] = T(i, j);
  return p;
}

// Where the frames come from — the runner's load_data(data_path, gray, frame_id) (:200,334-359), which refills the SAME two Mats
// every frame: imread, then gray_8u.convertTo(gray[i], PixelType).
//   PreloadedFrames   every frame already sits in its own PixelType Mat; load() hands out headers (no pixels move)
//   LoadPerFrame      the frames are kept as 8-bit images (imread's output); load() converts the pair into gray[0] / gray[1] INSIDE the
//                     frame loop, exactly as load_data does after decoding — no frame is known to the classes before its turn
struct PreloadedFrames {
  const std::vector<Mat>&left, &right;
  unsigned size() const { return (unsigned)left.size(); }
  void load(std::vector<Mat>& gray, unsigned id) const { gray[0] = left[id]; gray[1] = right[id]; }
};
struct LoadPerFrame {
  const std::vector<Mat>&left_8u, &right_8u;
  unsigned size() const { return (unsigned)left_8u.size(); }
  void load(std::vector<Mat>& gray, unsigned id) const { left_8u[id].convertTo(gray[0], PixelType); right_8u[id].convertTo(gray[1], PixelType); }
};

// Returns the number of keyframes created after the first one, or -1 if frame 0 could not be initialised.
// rel (optional): pose_to_keyframe of every frame (what Solve returned), for bit-exact comparisons.
template <class FrameSource>
inline int track_sequence(const FrameSource& frames, const Affine4f& pose0, std::vector<io::Pose34>& pred, bool verbose = true,
                          std::vector<Affine4f>* rel = nullptr) {
  const unsigned num_frames = frames.size();
  std::vector<Mat> pre_gray(2), cur_gray(2);                            // :93-94
  frames.load(pre_gray, 0);                                             // :95
  const std::vector<Mat>&left = pre_gray;                               // (frame 0 below)
  const Mat& right0 = pre_gray[1];
  const unsigned num_pyramid = 4;
  const auto t_setup = std::chrono::steady_clock::now();
  const float baseline = 386.1448f / 718.856f;                          // :41
  std::shared_ptr<CameraPyramid> left_cam_ptr = nullptr, right_cam_ptr = nullptr;  // :51-52

  DepthEstimator depth_estimator(8.0f, 900.0f, 15.0f, 0.1f, 30.0f, 0.01f, 28.0f, 0.995f, 50, 4, left_cam_ptr, right_cam_ptr,
                                 baseline, 80000);                      // :58-70
  std::vector<int> pose_max_iters = {10, 20, 30, 30};                   // :76
  Affine4f init_relative_affine = Affine4f::Identity();
  LevenbergMarquardtOptimizer pose_estimator(0.01f, 0.995f, pose_max_iters, init_relative_affine, left_cam_ptr, 1, 28.0f);  // :88

  Affine4f cur_pose = pose0, pose_to_keyframe = cur_pose;  // :96-98
  pred.assign(num_frames, to_pose34(cur_pose));
  const Scalar init_val(0);                                             // :99-101
  Mat pre_left_val(left[0].rows, left[0].cols, CV_8U, init_val), pre_left_disp(left[0].rows, left[0].cols, PixelType, init_val),
      pre_left_dep(left[0].rows, left[0].cols, PixelType, init_val);
  if (depth_estimator.ComputeDepth(left[0], right0, pre_left_val, pre_left_disp, pre_left_dep) == -1) {  // :102
    std::cout << "Init 0-th frame failed!" << std::endl;
    return -1;
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

  // ODO_RUNNER_PHASES=1: host time per phase of the frame loop, mean over the frames, on stderr (diagnostic)
  const bool phases = std::getenv("ODO_RUNNER_PHASES") != nullptr;
  const double setup_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_setup).count();
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](std::chrono::steady_clock::time_point& t0, int i) {
    if (!phases) return;
    const auto t1 = now();
    ph[i] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    t0 = t1;
  };
  for (unsigned frame_id = 1; frame_id < num_frames; frame_id++) {     // :198
    auto tp = now();
    {
    frames.load(cur_gray, frame_id);                                   // :200
    lap(tp, 7);
    ImagePyramid cur_img_pyramid(num_pyramid, cur_gray[0], true);      // :205
    lap(tp, 0);
    pose_to_keyframe = pose_estimator.Solve(std::get<0>(keyframes[current_kf]), std::get<1>(keyframes[current_kf]),
                                            cur_img_pyramid);          // :215
    lap(tp, 1);
    cur_pose = mul(keyframe_poses_abs[current_kf], rigid_inverse(pose_to_keyframe));  // :218

    Mat cur_left_val(left[0].rows, left[0].cols, CV_8U, init_val), cur_left_disp(left[0].rows, left[0].cols, PixelType),
        cur_left_dep(left[0].rows, left[0].cols, PixelType);
    lap(tp, 2);
    if (depth_estimator.ComputeDepth(cur_gray[0], cur_gray[1], cur_left_val, cur_left_disp, cur_left_dep) == -1) {
      std::cout << "    depth failed!" << std::endl;                   // :230-232
      break;
    }
    lap(tp, 3);
    delete pre_img_pyramid_ptr;
    delete pre_dep_pyramid_ptr;
    pre_img_pyramid_ptr = new ImagePyramid(num_pyramid, cur_gray[0], true);          // :251
    pre_dep_pyramid_ptr = new DepthPyramid(num_pyramid, cur_left_dep, false);        // :252
    lap(tp, 4);

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
    pred[frame_id] = to_pose34(cur_pose);                             // :222
    if (rel) rel->push_back(pose_to_keyframe);
    if (verbose)
      std::printf("frame %u kf %u motion %.4f  t = [% .5f % .5f % .5f]\n", frame_id, current_kf, motion_mag, cur_pose(0, 3),
                  cur_pose(1, 3), cur_pose(2, 3));
    lap(tp, 5);
    }
    lap(tp, 6);
  }
  if (phases && num_frames > 1) {
    const double n = num_frames - 1;
    std::fprintf(stderr, "[runner phases] load_data :200 %.1f  ", ph[7] / n);
    std::fprintf(stderr, "[runner phases] ImagePyramid :205 %.1f  Solve :215 %.1f  pose + output Mats %.1f  ComputeDepth :229 %.1f  "
                 "pyramids :251-252 %.1f  keyframe test + Reset %.1f  end of the frame's scope %.1f us per frame; constructors + frame 0: %.0f us once\n", ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, ph[6] / n, setup_us);
  }
  delete pre_img_pyramid_ptr;
  delete pre_dep_pyramid_ptr;
  return (int)current_kf;
}
inline int track_sequence(const std::vector<Mat>& left, const std::vector<Mat>& right, const Affine4f& pose0,
                          std::vector<io::Pose34>& pred, bool verbose = true, std::vector<Affine4f>* rel = nullptr) {
  return track_sequence(PreloadedFrames{left, right}, pose0, pred, verbose, rel);
}

}  // namespace odometry
