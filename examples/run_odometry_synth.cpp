// run_odometry_synth.cpp — the reference runner's frame loop (examples/runner_loop.hpp) on synthetic frames read from a
// raw file (this image has neither OpenCV nor the KITTI dataset).
//
//   python examples/make_frames.py frames.bin 12          # left/right fp32 frames, 1241x376
//   g++ -O2 -std=c++17 -Iinclude examples/run_odometry_synth.cpp -o run_odometry_synth -Lodometry_amd/lib -lodometry_hip -Wl,-rpath,$PWD/odometry_amd/lib
//   ./run_odometry_synth frames.bin
#include <cstring>

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

// The frames as 8-bit images, what cv::imread hands the runner's load_data (the synthetic frames are integer-valued; anything else is refused).
static bool to_8u(const std::vector<Mat>& in, std::vector<Mat>& out) {
  for (const Mat& m : in) {
    Mat u(m.rows, m.cols, CV_8U);
    for (int y = 0; y < m.rows; y++) {
      const float* sp = m.ptr<float>(y);
      unsigned char* dp = u.ptr<unsigned char>(y);
      for (int x = 0; x < m.cols; x++) {
        if (!(sp[x] >= 0.0f && sp[x] <= 255.0f && sp[x] == (float)(int)sp[x])) return false;
        dp[x] = (unsigned char)sp[x];
      }
    }
    out.push_back(u);
  }
  return true;
}

// --load-per-frame: the runner's own frame source — the two Mats of a frame are (re)filled INSIDE the frame loop from 8-bit images
// (load_data, ref: run_odometry_kitti_offline.cpp:200,334-359, minus the PNG decoding), instead of every frame waiting in its own Mat.
// --time [passes]: the loop again, silently, `passes` times over the same frames (a fresh run of the sequence each pass,
// like the runner starts one), and the rate at which the drop-in classes track host-resident frames on stderr:
//   SHIM_FPS <frames/s> FRAMES <tracked frames> PASSES <n>
// Set-up (estimator construction) and the PCIe transfers of every frame are inside the clock; reading the file is not.
#include <chrono>
int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: %s frames.bin [--load-per-frame] [--time [passes]] [--poses out.txt]\n", argv[0]); return 2; }
  std::vector<Mat> left, right;
  if (!read_frames(argv[1], left, right) || left.empty()) { std::printf("cannot read %s\n", argv[1]); return 2; }
  int passes = 0;
  const char* poses_out = nullptr;
  const char* rel_out = nullptr;   // --rel-bin: pose_to_keyframe of every tracked frame, 16 raw floats each (column-major)
  bool per_frame = false;
  for (int i = 2; i < argc; i++) {
    if (!std::strcmp(argv[i], "--load-per-frame")) per_frame = true;
    else if (!std::strcmp(argv[i], "--time")) { passes = (i + 1 < argc && argv[i + 1][0] != '-') ? std::atoi(argv[++i]) : 3; }
    else if (!std::strcmp(argv[i], "--poses") && i + 1 < argc) poses_out = argv[++i];
    else if (!std::strcmp(argv[i], "--rel-bin") && i + 1 < argc) rel_out = argv[++i];
  }
  std::vector<io::Pose34> pred;
  std::vector<Affine4f> rel;
  std::vector<Mat> left_8u, right_8u;
  if (per_frame && (!to_8u(left, left_8u) || !to_8u(right, right_8u))) { std::printf("--load-per-frame needs 8-bit-valued frames\n"); return 2; }
  auto track = [&](std::vector<io::Pose34>& out, bool verbose, std::vector<Affine4f>* r) {
    return per_frame ? track_sequence(LoadPerFrame{left_8u, right_8u}, Affine4f::Identity(), out, verbose, r)
                     : track_sequence(PreloadedFrames{left, right}, Affine4f::Identity(), out, verbose, r);
  };
  const int kf = track(pred, passes == 0, &rel);
  if (kf < 0) return 1;
  if (rel_out) {
    FILE* f = std::fopen(rel_out, "wb");
    if (!f) return 1;
    for (const Affine4f& T : rel) std::fwrite(affine_data(T), sizeof(float), 16, f);
    std::fclose(f);
  }
  std::cout << "Total keyframes: " << kf << std::endl;
  if (poses_out && !io::save_poses_kitti(poses_out, pred)) return 1;
  if (passes > 0) {
    std::streambuf* keep = std::cout.rdbuf(nullptr);  // the classes print the reference's messages: not part of the timing
    const size_t n = left.size();
    // the frames are re-read into fresh Mats before every pass, outside the clock (a new sequence from disk): each frame
    // crosses PCIe once per pass
    double secs = 0.0;
    for (int p = 0; p < passes; p++) {
      if (!per_frame) {
        left.clear(); right.clear();
        if (!read_frames(argv[1], left, right)) return 2;
      }
      std::vector<io::Pose34> pr;
      const auto t0 = std::chrono::steady_clock::now();
      if (track(pr, false, nullptr) < 0) return 1;
      secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 12; k++)
          if (pr[i].m[k] != pred[i].m[k]) { std::fprintf(stderr, "SHIM_MISMATCH frame %zu\n", i); return 3; }
    }
    std::fprintf(stderr, "SHIM_FPS %.1f FRAMES %zu PASSES %d\n", (double)(n - 1) * passes / secs, (n - 1) * (size_t)passes, passes);
    const ShimStats& st = shim_stats();
    std::fprintf(stderr, "SHIM_STATS uploads %lu fingerprints %lu unchanged %lu changed %lu early_adopted %lu early_dropped %lu delivered %lu outputs_prepared %lu verify_failures %lu\n",
                 st.uploads, st.fingerprints, st.unchanged, st.changed, st.early_adopted, st.early_dropped, st.delivered, st.outputs_prepared,
                 st.verify_failures);
    std::cout.rdbuf(keep);
  }
  return 0;
}
