// tests/shim_fuzz_harness.cpp — a random walk over what a caller can do to the drop-in classes' images (include/odometry_shim.hpp):
// refill a Mat in place, modify a small patch of an image or of a ComputeDepth output, build pyramids of Mats / header copies / views
// (cv::Mat build), run ComputeDepth into fresh or reused output Mats, hand an output back in, Solve against the last keyframe — in any
// order. Every operation prints a checksum of what it produced. The test runs the same walk (same seed) on the stand-in Mat with the
// look-ahead off — writes are seen through ptr<T>() there, nothing is started ahead — and on the cv::Mat build (tests/stubs) with
// the look-ahead on, where every decision rests on fingerprints: the lines must be identical.
//   shim_fuzz_harness frames.bin <seed> <operations>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#include "../include/odometry_shim.hpp"
using namespace odometry;

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
static uint64_t sum_mat(const Mat& m, size_t elem) {
  uint64_t h = 1469598103934665603ull;
  for (int y = 0; y < m.rows; y++) h = fnv(m.ptr<uint8_t>(y), (size_t)m.cols * elem, h);
  return h;
}
struct Rng {
  uint64_t s;
  unsigned next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 33); }
  int below(int n) { return (int)(next() % (unsigned)n); }
};

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  int hdr[3];
  if (!f || std::fread(hdr, sizeof(int), 3, f) != 3) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  const size_t px = (size_t)rows * cols;
  std::vector<std::vector<float>> raw(2 * (size_t)n, std::vector<float>(px));
  for (auto& v : raw) if (std::fread(v.data(), sizeof(float), px, f) != px) return 2;
  std::fclose(f);
  Rng rng{(uint64_t)std::atoll(argv[2]) * 2654435761ull + 88172645463325252ull};
  const int n_ops = std::atoi(argv[3]);
  std::streambuf* keep = std::cout.rdbuf(nullptr);
  std::shared_ptr<CameraPyramid> cam = nullptr;
  DepthEstimator de(8.0f, 900.0f, 15.0f, 0.1f, 30.0f, 0.01f, 28.0f, 0.995f, 50, 4, cam, cam, 386.1448f / 718.856f, 80000);
  LevenbergMarquardtOptimizer lm(0.01f, 0.995f, std::vector<int>{10, 20, 30, 30}, Affine4f::Identity(), cam, 1, 28.0f);
  Mat g[4] = {Mat(rows, cols, PixelType), Mat(rows, cols, PixelType), Mat(rows, cols, PixelType), Mat(rows, cols, PixelType)};   // two stereo pairs
  auto refill = [&](int pair, int k) {
    for (int y = 0; y < rows; y++) {
      std::memcpy(g[2 * pair].ptr<float>(y), raw[2 * (size_t)k].data() + (size_t)y * cols, (size_t)cols * 4);
      std::memcpy(g[2 * pair + 1].ptr<float>(y), raw[2 * (size_t)k + 1].data() + (size_t)y * cols, (size_t)cols * 4);
    }
  };
  refill(0, 0); refill(1, 1 % n);
  Mat val(rows, cols, CV_8U, 0.0), disp(rows, cols, PixelType), dep(rows, cols, PixelType);   // reused outputs
  std::unique_ptr<ImagePyramid> kf_img;
  std::unique_ptr<DepthPyramid> kf_dep;
  bool have_dep = false;
  for (int op = 0; op < n_ops; op++) {
    const int what = (op < 2) ? 3 : rng.below(8);
    const int pair = rng.below(2), img = rng.below(4);
    switch (what) {
      case 0: {   // the runner's load_data: both Mats of a pair refilled in place
        const int k = rng.below(n);
        refill(pair, k);
        std::printf("%d refill pair %d <- frame %d\n", op, pair, k);
        break;
      }
      case 1: {   // a small in-place write: an image, or the inverse-depth output
        const int y0 = 8 + rng.below(rows - 24), x0 = 8 + rng.below(cols - 24), sz = 1 + rng.below(6);
        Mat& m = (have_dep && rng.below(4) == 0) ? dep : g[img];
        for (int y = y0; y < y0 + sz; y++) for (int x = x0; x < x0 + sz; x++) m.at<float>(y, x) = m.at<float>(y, x) * 0.5f + 3.0f;
        std::printf("%d poke %s at %d,%d size %d\n", op, (&m == &dep) ? "dep" : "image", y0, x0, sz);
        break;
      }
      case 2: {   // ImagePyramid of a Mat, or of a header copy of it
        Mat h = g[img];
        ImagePyramid p(4, rng.below(2) ? h : g[img], rng.below(2) != 0);
        std::printf("%d image pyramid of %d: %016llx %016llx\n", op, img, (unsigned long long)sum_mat(p.GetPyramidImage(0), 4),
                    (unsigned long long)sum_mat(p.GetPyramidImage(3), 4));
        break;
      }
      case 3: case 4: {   // ComputeDepth of a pair (sometimes with left and right exchanged) into the reused or into fresh output Mats
        const bool flip = (what == 4) && rng.below(3) == 0, fresh = rng.below(2) != 0;
        Mat v2, d2, p2;
        if (fresh) { v2 = Mat(rows, cols, CV_8U, 0.0); d2 = Mat(rows, cols, PixelType); p2 = Mat(rows, cols, PixelType); }
        Mat &ov = fresh ? v2 : val, &od = fresh ? d2 : disp, &op_ = fresh ? p2 : dep;
        const Mat &L = g[2 * pair + (flip ? 1 : 0)], &R = g[2 * pair + (flip ? 0 : 1)];
        const int st = de.ComputeDepth(L, R, ov, od, op_);
        Download(od); Download(op_);
        const Mat &cv_ = ov, &cd = od, &cp = op_;
        std::printf("%d depth pair %d flip %d fresh %d: %d %016llx %016llx %016llx\n", op, pair, (int)flip, (int)fresh, st,
                    (unsigned long long)sum_mat(cv_, 1), (unsigned long long)sum_mat(cd, 4), (unsigned long long)sum_mat(cp, 4));
        if (fresh) { val = v2; disp = d2; dep = p2; }   // (header assignment: the reused outputs now ARE the fresh ones)
        have_dep = true;
        if (st == 0 && (!kf_img || rng.below(3) == 0)) {   // a new keyframe: this left image and its inverse depth
          kf_img.reset(new ImagePyramid(4, L, true));
          kf_dep.reset(new DepthPyramid(4, dep, false));
          lm.Reset(Affine4f::Identity(), 0.01f);
          std::printf("%d keyframe\n", op);
        }
        break;
      }
      case 5: {   // DepthPyramid of the inverse-depth output handed back in
        if (!have_dep) break;
        DepthPyramid p(4, dep, false);
        std::printf("%d depth pyramid: %016llx %016llx\n", op, (unsigned long long)sum_mat(p.GetPyramidDepth(0), 4),
                    (unsigned long long)sum_mat(p.GetPyramidDepth(2), 4));
        break;
      }
      case 6: case 7: {   // the runner's :205 + :215 (+ :229 right behind it half of the time: what the look-ahead is for)
        if (!kf_img) break;
        ImagePyramid cur(4, g[2 * pair], true);
        const Affine4f T = lm.Solve(*kf_img, *kf_dep, cur);
        std::printf("%d solve left of pair %d: %016llx\n", op, pair, (unsigned long long)fnv(affine_data(T), 64));
        lm.Reset(Affine4f::Identity(), 0.01f);
        if (what == 7) {
          if (rng.below(4) == 0) { const int y = 20 + rng.below(rows - 40), x = 20 + rng.below(cols - 40); g[2 * pair + rng.below(2)].at<float>(y, x) += 2.0f; }
          Mat v2(rows, cols, CV_8U, 0.0), d2(rows, cols, PixelType), p2(rows, cols, PixelType);
          const int st = de.ComputeDepth(g[2 * pair], g[2 * pair + 1], v2, d2, p2);
          Download(d2); Download(p2);
          const Mat &cv_ = v2, &cd = d2, &cp = p2;
          std::printf("%d depth behind the solve: %d %016llx %016llx %016llx\n", op, st, (unsigned long long)sum_mat(cv_, 1),
                      (unsigned long long)sum_mat(cd, 4), (unsigned long long)sum_mat(cp, 4));
          ImagePyramid again(4, g[2 * pair], true);   // :251
          DepthPyramid dp(4, p2, false);               // :252
          std::printf("%d pyramids behind it: %016llx %016llx\n", op, (unsigned long long)sum_mat(again.GetPyramidImage(1), 4),
                      (unsigned long long)sum_mat(dp.GetPyramidDepth(1), 4));
        }
        break;
      }
    }
  }
  std::cout.rdbuf(keep);
  const ShimStats& st = shim_stats();
  std::fprintf(stderr, "SHIM_STATS uploads %lu fingerprints %lu unchanged %lu changed %lu early_adopted %lu early_dropped %lu delivered %lu verify_failures %lu\n",
               st.uploads, st.fingerprints, st.unchanged, st.changed, st.early_adopted, st.early_dropped, st.delivered, st.verify_failures);
  return 0;
}
