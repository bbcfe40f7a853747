// tests/host_fp_harness.cpp — odometry_amd/csrc/host_fp.h (the image fingerprint + fused copy of the cv::Mat branch) compiled on its own
// with -fsanitize=address,undefined: every size from 0 to 5 000 bytes and a few image shapes, source and destination allocated
// EXACTLY (heap blocks of the image's size: a read or write one byte past the end is reported), dense and pitched; the AVX2 and the
// scalar form must agree, the copy must be exact, a flipped bit must change the value. Prints OK.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../odometry_amd/csrc/host_fp.h"

int main() {
  unsigned seed = 12345u;
  auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return (unsigned char)(seed >> 24); };
  for (size_t n = 0; n <= 5000; n++) {
    unsigned char* src = (unsigned char*)std::malloc(n ? n : 1);
    unsigned char* dst = (unsigned char*)std::malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++) src[i] = rnd();
    const uint64_t a = hostfp::run_scalar(src, n, nullptr);
    const uint64_t b = hostfp::run(src, n, nullptr);
    const uint64_t c = hostfp::run(src, n, dst);
    if (a != b || a != c || std::memcmp(src, dst, n) != 0) { std::printf("FAILED at %zu bytes\n", n); return 1; }
    if (n) { src[n - 1] ^= 1; if (hostfp::run(src, n, nullptr) == a) { std::printf("FAILED: last byte of %zu not seen\n", n); return 1; } }
    std::free(src); std::free(dst);
  }
  const int shapes[][3] = {{376, 1241, 4}, {376, 1241, 1}, {47, 155, 4}, {3, 5, 4}, {1, 1, 1}, {9, 263, 4}};
  for (auto& sh : shapes) {
    const int rows = sh[0], cols = sh[1], es = sh[2];
    const size_t row_bytes = (size_t)cols * es;
    for (size_t pad : {(size_t)0, (size_t)12}) {
      const size_t pitch = row_bytes + pad, total = pitch * (rows - 1) + row_bytes;   // the last row has no padding behind it
      unsigned char* src = (unsigned char*)std::malloc(total);
      for (size_t i = 0; i < total; i++) src[i] = rnd();
      unsigned char* dense = (unsigned char*)std::malloc(row_bytes * rows);
      const uint64_t f = hostfp::image(src, pitch, row_bytes, rows, nullptr, 0);
      const uint64_t g = hostfp::image(src, pitch, row_bytes, rows, dense, row_bytes);
      if (f != g) { std::printf("FAILED: copy variant differs (%d x %d)\n", rows, cols); return 1; }
      for (int y = 0; y < rows; y++)
        if (std::memcmp(dense + (size_t)y * row_bytes, src + (size_t)y * pitch, row_bytes) != 0) { std::printf("FAILED: copy\n"); return 1; }
      // and out again into a pitched destination
      unsigned char* back = (unsigned char*)std::malloc(total);
      std::memset(back, 0xAB, total);
      (void)hostfp::image(dense, row_bytes, row_bytes, rows, back, pitch);
      for (int y = 0; y < rows; y++) {
        if (std::memcmp(back + (size_t)y * pitch, dense + (size_t)y * row_bytes, row_bytes) != 0) { std::printf("FAILED: copy back\n"); return 1; }
        if (pad && y + 1 < rows && back[(size_t)y * pitch + row_bytes] != 0xAB) { std::printf("FAILED: wrote into the padding\n"); return 1; }
      }
      std::free(src); std::free(dense); std::free(back);
    }
  }
  std::printf("OK\n");
  return 0;
}
