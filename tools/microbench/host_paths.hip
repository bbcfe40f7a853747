// What does the HOST side of a frame cost when the images live in plain (pageable) cv::Mat memory? Prices the pieces the cv::Mat
// branch of include/odometry_shim.hpp is built from, for one 1241x376 fp32 image (1.87 MB) unless stated:
//   * a full-image fingerprint pass (read only), single thread — the check that a Mat seen before still holds the bytes that were uploaded;
//   * the copy into page-locked staging memory, plain and fused with the fingerprint, normal and non-temporal stores;
//   * the copy out of staging memory into a pageable Mat (ComputeDepth's three outputs: 0.47 + 1.87 + 1.87 MB);
//   * hipHostRegister / hipHostUnregister of such a buffer, and a DMA from a registered buffer against one from staging memory;
//   * the DMAs themselves.
//   hipcc --offload-arch=gfx950 -O2 -mavx2 host_paths.hip -o host_paths && ./host_paths
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint64_t g_key[128 * 4];
static void make_keys() { uint64_t s = 0x9E3779B97F4A7C15ull; for (auto& k : g_key) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; k = s; } }

// NH-style: per 64-bit word (lo + k_lo) * (hi + k_hi), summed per lane; lanes scrambled every 1 KB.
__attribute__((target("avx2"))) static uint64_t fp_avx2(const void* p, size_t bytes, void* copy_to, bool nt) {
  const __m256i* s = (const __m256i*)p;
  __m256i* d = (__m256i*)copy_to;
  __m256i a0 = _mm256_set1_epi64x(1), a1 = _mm256_set1_epi64x(2), a2 = _mm256_set1_epi64x(3), a3 = _mm256_set1_epi64x(4);
  const __m256i* k = (const __m256i*)g_key;
  const __m256i prime = _mm256_set1_epi64x(0x9E3779B1);
  size_t n = bytes / 32, i = 0;
  while (i + 32 <= n) {
    for (int j = 0; j < 32; j += 4) {
      __m256i v0 = _mm256_loadu_si256(s + i + j), v1 = _mm256_loadu_si256(s + i + j + 1), v2 = _mm256_loadu_si256(s + i + j + 2), v3 = _mm256_loadu_si256(s + i + j + 3);
      if (d) {
        if (nt) { _mm256_stream_si256(d + i + j, v0); _mm256_stream_si256(d + i + j + 1, v1); _mm256_stream_si256(d + i + j + 2, v2); _mm256_stream_si256(d + i + j + 3, v3); }
        else { _mm256_storeu_si256(d + i + j, v0); _mm256_storeu_si256(d + i + j + 1, v1); _mm256_storeu_si256(d + i + j + 2, v2); _mm256_storeu_si256(d + i + j + 3, v3); }
      }
      v0 = _mm256_add_epi32(v0, k[j]); v1 = _mm256_add_epi32(v1, k[j + 1]); v2 = _mm256_add_epi32(v2, k[j + 2]); v3 = _mm256_add_epi32(v3, k[j + 3]);
      a0 = _mm256_add_epi64(a0, _mm256_mul_epu32(v0, _mm256_shuffle_epi32(v0, 0xB1)));
      a1 = _mm256_add_epi64(a1, _mm256_mul_epu32(v1, _mm256_shuffle_epi32(v1, 0xB1)));
      a2 = _mm256_add_epi64(a2, _mm256_mul_epu32(v2, _mm256_shuffle_epi32(v2, 0xB1)));
      a3 = _mm256_add_epi64(a3, _mm256_mul_epu32(v3, _mm256_shuffle_epi32(v3, 0xB1)));
    }
    i += 32;
    // scramble: a ^= a >> 29; a = lo(a) * prime + (hi(a) * prime << 32)
#define SCR(a) { a = _mm256_xor_si256(a, _mm256_srli_epi64(a, 29)); __m256i lo = _mm256_mul_epu32(a, prime); __m256i hi = _mm256_mul_epu32(_mm256_srli_epi64(a, 32), prime); a = _mm256_add_epi64(lo, _mm256_slli_epi64(hi, 32)); }
    SCR(a0) SCR(a1) SCR(a2) SCR(a3)
  }
  uint64_t acc[16];
  _mm256_storeu_si256((__m256i*)acc, a0); _mm256_storeu_si256((__m256i*)(acc + 4), a1); _mm256_storeu_si256((__m256i*)(acc + 8), a2); _mm256_storeu_si256((__m256i*)(acc + 12), a3);
  uint64_t h = bytes * 0x9E3779B97F4A7C15ull;
  // tail (< 1 KB): scalar
  const unsigned char* t = (const unsigned char*)p + i * 32;
  size_t rest = bytes - i * 32;
  if (d && rest) memcpy((char*)copy_to + i * 32, t, rest);
  for (size_t q = 0; q < rest; q++) h = (h ^ t[q]) * 0x100000001B3ull;
  for (int q = 0; q < 16; q++) { h ^= acc[q]; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; }
  return h;
}

template <class F> static double med(F f, int reps = 15) {
  std::vector<double> t;
  for (int r = 0; r < reps; r++) { double t0 = now_us(); f(); t.push_back(now_us() - t0); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
static void flush_cache() {   // walk 256 MB
  static std::vector<char> junk(256u << 20, 1);
  volatile long s = 0;
  for (size_t i = 0; i < junk.size(); i += 64) s += junk[i];
}

int main() {
  make_keys();
  const size_t N = (size_t)1241 * 376 * 4, NV = (size_t)1241 * 376;
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  void *dev, *dev2; hipMalloc(&dev, N); hipMalloc(&dev2, N);
  void* pin; hipHostMalloc(&pin, N, hipHostMallocDefault);
  void* pin2; hipHostMalloc(&pin2, N, hipHostMallocDefault);
  float* img = (float*)malloc(N);
  float* img2 = (float*)malloc(N);
  for (size_t i = 0; i < N / 4; i++) img[i] = (float)(i * 2654435761u >> 24);
  memcpy(img2, img, N);
  volatile uint64_t sink = 0;
  printf("image %.2f MB\n", N / 1e6);
  // 1. fingerprint, warm (just written) and cold
  printf("fingerprint warm       %7.1f us\n", med([&] { sink = fp_avx2(img, N, nullptr, false); }));
  { std::vector<double> t; for (int r = 0; r < 5; r++) { flush_cache(); double t0 = now_us(); sink = fp_avx2(img, N, nullptr, false); t.push_back(now_us() - t0); } std::sort(t.begin(), t.end()); printf("fingerprint cold       %7.1f us\n", t[2]); }
  printf("memcpy -> pinned warm  %7.1f us\n", med([&] { memcpy(pin, img, N); }));
  printf("copy+fp -> pinned      %7.1f us\n", med([&] { sink = fp_avx2(img, N, pin, false); }));
  printf("copy+fp -> pinned (nt) %7.1f us\n", med([&] { sink = fp_avx2(img, N, pin, true); }));
  { std::vector<double> t; for (int r = 0; r < 5; r++) { flush_cache(); double t0 = now_us(); sink = fp_avx2(img, N, pin, true); t.push_back(now_us() - t0); } std::sort(t.begin(), t.end()); printf("copy+fp(nt) cold src   %7.1f us\n", t[2]); }
  // after a load that just wrote the image (u8 -> f32 convert, like cv::Mat::convertTo)
  std::vector<unsigned char> u8(NV);
  for (size_t i = 0; i < NV; i++) u8[i] = (unsigned char)(i * 2654435761u >> 24);
  printf("u8->f32 convert (load) %7.1f us\n", med([&] { for (size_t i = 0; i < NV; i++) img[i] = (float)u8[i]; }));
  printf("memcpy f32 frame (load)%7.1f us\n", med([&] { memcpy(img, img2, N); }));
  // 2. DMA
  auto dma = [&](void* src, size_t n) { hipMemcpyAsync(dev, src, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); };
  printf("H2D from pinned, sync  %7.1f us\n", med([&] { dma(pin, N); }));
  printf("H2D from pageable,sync %7.1f us (hipMemcpyAsync does the staging)\n", med([&] { dma(img, N); }));
  // fingerprint of the Mat while the DMA from a REGISTERED Mat runs
  double t0 = now_us();
  hipError_t e = hipHostRegister(img, N, hipHostRegisterDefault);
  printf("hipHostRegister        %7.1f us (%s)\n", now_us() - t0, hipGetErrorString(e));
  if (e == hipSuccess) {
    printf("H2D from registered    %7.1f us\n", med([&] { dma(img, N); }));
    printf("H2D registered || fp   %7.1f us\n", med([&] { hipMemcpyAsync(dev, img, N, hipMemcpyHostToDevice, st); sink = fp_avx2(img, N, nullptr, false); hipStreamSynchronize(st); }));
    t0 = now_us(); hipHostUnregister(img); printf("hipHostUnregister      %7.1f us\n", now_us() - t0);
    printf("register+unregister    %7.1f us (median of repeats)\n", med([&] { hipHostRegister(img, N, hipHostRegisterDefault); hipHostUnregister(img); }, 7));
  }
  // 3. outputs: D2H into staging, then staging -> pageable Mats
  printf("D2H 1.87 MB to pinned  %7.1f us\n", med([&] { hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }));
  printf("D2H 1.87 MB to pageable%7.1f us\n", med([&] { hipMemcpyAsync(img2, dev, N, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }));
  printf("memcpy pinned->Mat     %7.1f us (dest reused)\n", med([&] { memcpy(img2, pin, N); }));
  printf("copy+fp pinned->Mat    %7.1f us\n", med([&] { sink = fp_avx2(pin, N, img2, false); }));
  printf("fresh malloc+memcpy    %7.1f us\n", med([&] { void* m = malloc(N); memcpy(m, pin, N); sink = *(volatile char*)m; free(m); }));
  printf("memset 1.87 MB Mat     %7.1f us\n", med([&] { memset(img2, 0, N); }));
  // sparse delivery: memset + scatter of 20 000 points
  std::vector<unsigned> idx(20000); for (size_t i = 0; i < idx.size(); i++) idx[i] = (unsigned)((i * 2654435761u) % NV);
  std::sort(idx.begin(), idx.end());
  printf("memset + scatter 20k   %7.1f us\n", med([&] { memset(img2, 0, N); for (unsigned q : idx) img2[q] = 1.5f; }));
  // zero-copy kernel write into mapped pinned memory is not measured here
  printf("(sink %llu)\n", (unsigned long long)sink);
  return 0;
}
