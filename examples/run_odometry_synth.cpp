// run_odometry_synth.cpp — the reference runner's frame loop (examples/runner_loop.hpp) on synthetic frames read from a
// raw file (this image has neither OpenCV nor the KITTI dataset).
//
//   python examples/make_frames.py frames.bin 12          # left/right fp32 frames, 1241x376
//   g++ -O2 -std=c++17 -Iinclude examples/run_odometry_synth.cpp -o run_odometry_synth -Lodometry_amd/lib -lodometry_hip -Wl,-rpath,$PWD/odometry_amd/lib
//   ./run_odometry_synth frames.bin
#include "runner_loop.hpp"

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

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: %s frames.bin\n", argv[0]); return 2; }
  std::vector<Mat> left, right;
  if (!read_frames(argv[1], left, right) || left.empty()) { std::printf("cannot read %s\n", argv[1]); return 2; }
  std::vector<io::Pose34> pred;
  const int kf = track_sequence(left, right, Affine4f::Identity(), pred);
  if (kf < 0) return 1;
  std::cout << "Total keyframes: " << kf << std::endl;
  return 0;
}
