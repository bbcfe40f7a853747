// tests/shim_lookahead_harness.cpp — what the drop-in classes start ahead of the call that needs it (include/odometry_shim.hpp,
// struct Lookahead) under the usage patterns that try to break it. Usage: shim_lookahead_harness frames.bin <mode>
//   vector   every frame in its own Mat, created pair by pair (the guesses come true)
//   refill   two Mats refilled per frame through ptr<float>() — what the reference's load_data does (run_odometry_kitti_offline.cpp
//            :334-359): partner known from the last ComputeDepth, no next frame to send ahead
//   swap     like refill, but the roles of the two Mats alternate every frame (the partner guess names the wrong image)
//   poke     like vector, and the right image is modified between Solve and ComputeDepth (a job started ahead must be dropped)
//   poke_left   like refill, and the LEFT image is modified in place between Solve (:215) and ComputeDepth (:229): the upload made for
//            ImagePyramid (:205) no longer is the image — ComputeDepth and the keyframe pyramid must see the new pixels
//   shared_outputs   like refill, but ComputeDepth's output Mats are not the caller's alone: left_disp has a second header, left_dep (cv::Mat
//            build) lies in user memory — the cv::Mat build must then write them IN PLACE (it otherwise hands over images it built while
//            Solve waited, by header assignment); the checksums are taken through the OTHER header / the user memory
//   raw_pointers   like refill, and the caller keeps the raw data pointers of its three output Mats from BEFORE ComputeDepth and reads the
//            results through them — valid against the reference, which writes in place (ref: src/depth_estimate.cpp:176-191,388-397);
//            exit code 4 if an output Mat came back with another buffer (what ODOMETRY_SHIM_SWAP_OUTPUTS=1 does, hence opt-in)
// Built twice by the test: with the stand-in Mat (writes are seen through ptr<T>() / at<T>()) and with -DODOMETRY_SHIM_WITH_OPENCV against
// tests/stubs (a cv::Mat reports nothing: every use fingerprints the pixels). The last line on stderr: SHIM_STATS (ShimStats).
// Prints one line per frame: pose bits and checksums of the three depth outputs. The test runs every mode with and without
// ODOMETRY_SHIM_NO_LOOKAHEAD=1: same lines (poke: against its own look-ahead-off run).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../include/odometry_shim.hpp"
using namespace odometry;

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[2];
  FILE* f = std::fopen(argv[1], "rb");
  int hdr[3];
  if (!f || std::fread(hdr, sizeof(int), 3, f) != 3) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  const size_t px = (size_t)rows * cols;
  std::vector<std::vector<float>> raw(2 * (size_t)n, std::vector<float>(px));
  for (auto& v : raw) if (std::fread(v.data(), sizeof(float), px, f) != px) return 2;
  std::fclose(f);
  std::streambuf* keep = std::cout.rdbuf(nullptr);   // the classes print the reference's messages
  const bool per_frame_mats = (mode == "vector" || mode == "poke");
  std::vector<Mat> vl, vr;
  if (per_frame_mats)
    for (int k = 0; k < n; k++) {
      Mat l(rows, cols, PixelType), r(rows, cols, PixelType);
      std::memcpy(l.ptr<float>(), raw[2 * k].data(), px * 4);
      std::memcpy(r.ptr<float>(), raw[2 * k + 1].data(), px * 4);
      vl.push_back(l); vr.push_back(r);
    }
  Mat g[2] = {Mat(rows, cols, PixelType), Mat(rows, cols, PixelType)};
  std::shared_ptr<CameraPyramid> cam = nullptr;
  DepthEstimator de(8.0f, 900.0f, 15.0f, 0.1f, 30.0f, 0.01f, 28.0f, 0.995f, 50, 4, cam, cam, 386.1448f / 718.856f, 80000);
  LevenbergMarquardtOptimizer lm(0.01f, 0.995f, std::vector<int>{10, 20, 30, 30}, Affine4f::Identity(), cam, 1, 28.0f);
  std::unique_ptr<ImagePyramid> kf_img;
  std::unique_ptr<DepthPyramid> kf_dep;
  for (int k = 0; k < n; k++) {
    const int li = (mode == "swap") ? (k & 1) : 0;
    if (!per_frame_mats) {
      std::memcpy(g[li].ptr<float>(), raw[2 * k].data(), px * 4);
      std::memcpy(g[1 - li].ptr<float>(), raw[2 * k + 1].data(), px * 4);
    }
    Mat& L = per_frame_mats ? vl[k] : g[li];
    Mat& R = per_frame_mats ? vr[k] : g[1 - li];
    Affine4f T = Affine4f::Identity();
    if (k > 0) {
      ImagePyramid cur(4, L, true);
      T = lm.Solve(*kf_img, *kf_dep, cur);
    }
    if (mode == "poke" && (k % 3) == 1) R.at<float>(rows / 2, cols / 2) += 1.0f;   // after anything was started ahead for R
    if (mode == "poke_left" && (k % 3) == 1)                                        // a patch no sparse sample would notice
      for (int y = 100; y < 108; y++) for (int x = 301; x < 309; x++) L.at<float>(y, x) = 255.0f - L.at<float>(y, x);
    Mat val(rows, cols, CV_8U, 0.0), disp(rows, cols, PixelType), dep(rows, cols, PixelType);
    Mat disp_other;
    static std::vector<float> user_dep;
    if (mode == "shared_outputs") {
      disp_other = disp;   // a second header: whoever holds it must see the results
#ifdef ODOMETRY_SHIM_WITH_OPENCV
      user_dep.assign(px, -1.0f);
      dep = Mat(rows, cols, PixelType, user_dep.data());   // user memory: nothing but these bytes may receive the image
#endif
    }
#ifdef ODOMETRY_SHIM_WITH_OPENCV
    const unsigned char* raw_before[3] = {val.data, disp.data, dep.data};
#endif
    const int st = de.ComputeDepth(L, R, val, disp, dep);
#ifdef ODOMETRY_SHIM_WITH_OPENCV
    if (mode == "raw_pointers" && (val.data != raw_before[0] || disp.data != raw_before[1] || dep.data != raw_before[2])) return 4;
#endif
#ifdef ODOMETRY_SHIM_WITH_OPENCV
    if (mode == "shared_outputs" && (disp_other.data != disp.data || dep.data != reinterpret_cast<unsigned char*>(user_dep.data()))) return 3;
#endif
    if (mode == "shared_outputs") disp = disp_other;
    if (k == 0 || (k % 4) == 0) {   // a new keyframe now and then
      kf_img.reset(new ImagePyramid(4, L, true));
      kf_dep.reset(new DepthPyramid(4, dep, false));
      lm.Reset(Affine4f::Identity(), 0.01f);
    } else {
      lm.Reset(T, 0.01f);
    }
    Download(disp); Download(dep);   // (ODOMETRY_SHIM_LAZY_OUTPUTS=1 leaves them on the device until asked for; otherwise nothing to do)
    const Mat &cv = val, &cd = disp, &cp = dep;
    std::printf("%d %d %016llx %016llx %016llx %016llx\n", k, st, (unsigned long long)fnv(affine_data(T), 64),
                (unsigned long long)fnv(cv.ptr<uint8_t>(), px), (unsigned long long)fnv(cd.ptr<float>(), px * 4),
                (unsigned long long)fnv(cp.ptr<float>(), px * 4));
  }
  std::cout.rdbuf(keep);
  const ShimStats& st = shim_stats();
  std::fprintf(stderr, "SHIM_STATS uploads %lu fingerprints %lu unchanged %lu changed %lu early_adopted %lu early_dropped %lu delivered %lu outputs_prepared %lu verify_failures %lu\n",
               st.uploads, st.fingerprints, st.unchanged, st.changed, st.early_adopted, st.early_dropped, st.delivered, st.outputs_prepared,
               st.verify_failures);
  return 0;
}
