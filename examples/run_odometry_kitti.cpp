// run_odometry_kitti.cpp — the reference's offline KITTI runner (run_odometry_kitti_offline.cpp) on the HIP hot path:
// reads <root>/sequences/<seq>/image_{0,1}/%06d.png and <root>/poses/<seq>.txt with include/odometry_io.hpp (std-only
// PNG decoder), tracks with examples/runner_loop.hpp, prints the translation error (ref: :361-372) and writes the
// predicted poses in KITTI format (ref: :374-430).
//
//   g++ -O2 -std=c++17 -Iinclude examples/run_odometry_kitti.cpp -o run_odometry_kitti -Lodometry_amd/lib -lodometry_hip -Wl,-rpath,$PWD/odometry_amd/lib
//   ./run_odometry_kitti /data/kitti/dataset 00 130 pred_00.txt
#include "runner_loop.hpp"

using namespace odometry;

int main(int argc, char** argv) {
  if (argc < 5) { std::printf("usage: %s <kitti dataset root> <sequence> <num_frames> <out poses.txt>\n", argv[0]); return 2; }
  const std::string root = argv[1], seq = argv[2];
  const int num_frames = std::atoi(argv[3]);
  std::vector<io::Pose34> gt;
  if (!io::load_gt_poses(root + "/poses/" + seq + ".txt", gt, (size_t)num_frames)) {
    std::cout << "open gt pose file failed: " << root + "/poses/" + seq + ".txt" << std::endl;  // ref: :302-305
    return 1;
  }
  std::vector<Mat> left, right;
  for (int i = 0; i < num_frames; i++) {
    for (int cam = 0; cam < 2; cam++) {
      std::vector<float> px;
      int w = 0, h = 0;
      const std::string p = io::kitti_image_path(root, seq, cam, i);
      std::cout << "reading frame: " << p << std::endl;                                      // ref: :340
      if (!io::read_png_gray_f32(p, px, w, h)) { std::cout << "read img failed." << std::endl; return 1; }  // ref: :343-346
      Mat m(h, w, PixelType);
      std::memcpy(m.ptr<float>(), px.data(), sizeof(float) * px.size());
      (cam == 0 ? left : right).push_back(m);
    }
  }
  Affine4f pose0 = Affine4f::Identity();                                                      // ref: :96-97 pred_poses[0] = gt_poses[0]
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) pose0(i, j) = gt[0].m[i * 4 + j];
  std::vector<io::Pose34> pred;
  const int kf = track_sequence(left, right, pose0, pred, false);
  if (kf < 0) return 1;
  std::vector<float> err;
  const float mean = io::eval_translation_error(gt, pred, pred.size(), &err);
  for (size_t i = 0; i < err.size(); i++) std::cout << "frame " << i << ": " << err[i] << std::endl;
  std::cout << "avg error over " << err.size() << " frames: " << mean << std::endl;
  std::cout << "Total keyframes: " << kf << std::endl;
  if (!io::save_poses_kitti(argv[4], pred)) { std::cout << "open pred write file failed: " << argv[4] << std::endl; return 1; }
  std::cout << "save completed." << std::endl;
  return 0;
}
