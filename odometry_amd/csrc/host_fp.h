// host_fp.h — host-side passes over an image in plain (pageable) memory, for callers whose images the library cannot watch
// (cv::Mat in include/odometry_shim.hpp): a 64-bit content fingerprint, and the same fused with a copy (into the pinned staging
// ring on the way up, out of pinned staging memory on the way down), so that "is this the image I uploaded?" costs one read of the
// image (17 us per 1241x376 fp32 frame on an EPYC 9575F core, cache-warm; 42 us cold) and a staged upload costs no more than its copy.
//
// The fingerprint is an NH-style multiply-accumulate hash (per 64-bit word (lo + k_lo) * (hi + k_hi), 16 independent 64-bit lanes,
// lanes scrambled every 1 KB so that the position of a block matters), not a cryptographic one: it guards against a caller rewriting
// a buffer, not against an adversary. Scalar and AVX2 forms give the same value. An image whose rows are `pitch` bytes apart is hashed
// row by row (a view and its dense copy hash differently; a record is only ever compared with the same geometry).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hostfp {

struct Keys {
  uint64_t k[32 * 4];
  Keys() { uint64_t s = 0x9E3779B97F4A7C15ull; for (auto& v : k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; } }
};
inline const Keys& keys() { static const Keys K; return K; }
constexpr uint64_t kPrime32 = 0x9E3779B1ull;

struct State { uint64_t acc[16]; };   // lane q of vector j & 3 is acc[(j & 3) * 4 + q]
inline void init(State& s) { for (int i = 0; i < 16; i++) s.acc[i] = (uint64_t)(i / 4 + 1); }
inline uint64_t scramble1(uint64_t a) { a ^= a >> 29; return (a & 0xFFFFFFFFull) * kPrime32 + (((a >> 32) * kPrime32) << 32); }
inline void accumulate_vec_scalar(State& s, const uint8_t* p, int j) {   // one 32-byte vector, key row j (0..31)
  const uint64_t* key = keys().k + (size_t)j * 4;
  for (int q = 0; q < 4; q++) {
    uint64_t w;
    std::memcpy(&w, p + q * 8, 8);
    const uint32_t lo = (uint32_t)w + (uint32_t)key[q], hi = (uint32_t)(w >> 32) + (uint32_t)(key[q] >> 32);
    s.acc[(j & 3) * 4 + q] += (uint64_t)lo * (uint64_t)hi;
  }
}
inline uint64_t finish(const State& s, size_t bytes) {
  uint64_t h = (uint64_t)bytes * 0x9E3779B97F4A7C15ull;
  for (int q = 0; q < 16; q++) { h ^= s.acc[q]; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; }
  return h;
}

// One dense run of `bytes` bytes; copy_to (optional) receives the bytes.
inline uint64_t run_scalar(const void* src, size_t bytes, void* copy_to) {
  State s;
  init(s);
  const uint8_t* p = static_cast<const uint8_t*>(src);
  if (copy_to) std::memcpy(copy_to, src, bytes);
  const size_t nvec = bytes / 32;
  for (size_t i = 0; i < nvec; i++) {
    accumulate_vec_scalar(s, p + i * 32, (int)(i & 31));
    if ((i & 31) == 31) for (auto& a : s.acc) a = scramble1(a);
  }
  const size_t rest = bytes - nvec * 32;
  if (rest) {
    uint8_t last[32] = {0};
    std::memcpy(last, p + nvec * 32, rest);
    accumulate_vec_scalar(s, last, (int)(nvec & 31));
  }
  return finish(s, bytes);
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline uint64_t run_avx2(const void* src, size_t bytes, void* copy_to) {
  const __m256i* sp = static_cast<const __m256i*>(src);
  __m256i* dp = static_cast<__m256i*>(copy_to);
  const __m256i* k = reinterpret_cast<const __m256i*>(keys().k);
  __m256i a0 = _mm256_set1_epi64x(1), a1 = _mm256_set1_epi64x(2), a2 = _mm256_set1_epi64x(3), a3 = _mm256_set1_epi64x(4);
  const __m256i prime = _mm256_set1_epi64x((long long)kPrime32);
  const size_t nvec = bytes / 32;
  size_t i = 0;
#define HOSTFP_ACC(a, v, kk) { __m256i t = _mm256_add_epi32(v, _mm256_loadu_si256(kk)); a = _mm256_add_epi64(a, _mm256_mul_epu32(t, _mm256_shuffle_epi32(t, 0xB1))); }
#define HOSTFP_SCR(a) { a = _mm256_xor_si256(a, _mm256_srli_epi64(a, 29)); const __m256i lo = _mm256_mul_epu32(a, prime), hi = _mm256_mul_epu32(_mm256_srli_epi64(a, 32), prime); a = _mm256_add_epi64(lo, _mm256_slli_epi64(hi, 32)); }
  for (; i + 32 <= nvec; i += 32) {
    for (int j = 0; j < 32; j += 4) {
      const __m256i v0 = _mm256_loadu_si256(sp + i + j), v1 = _mm256_loadu_si256(sp + i + j + 1), v2 = _mm256_loadu_si256(sp + i + j + 2),
                    v3 = _mm256_loadu_si256(sp + i + j + 3);
      if (dp) { _mm256_storeu_si256(dp + i + j, v0); _mm256_storeu_si256(dp + i + j + 1, v1); _mm256_storeu_si256(dp + i + j + 2, v2); _mm256_storeu_si256(dp + i + j + 3, v3); }
      HOSTFP_ACC(a0, v0, k + j) HOSTFP_ACC(a1, v1, k + j + 1) HOSTFP_ACC(a2, v2, k + j + 2) HOSTFP_ACC(a3, v3, k + j + 3)
    }
    HOSTFP_SCR(a0) HOSTFP_SCR(a1) HOSTFP_SCR(a2) HOSTFP_SCR(a3)
  }
  State s;
  _mm256_storeu_si256(reinterpret_cast<__m256i*>(s.acc), a0); _mm256_storeu_si256(reinterpret_cast<__m256i*>(s.acc + 4), a1);
  _mm256_storeu_si256(reinterpret_cast<__m256i*>(s.acc + 8), a2); _mm256_storeu_si256(reinterpret_cast<__m256i*>(s.acc + 12), a3);
#undef HOSTFP_ACC
#undef HOSTFP_SCR
  const uint8_t* p = static_cast<const uint8_t*>(src);
  const size_t done = i * 32;
  if (copy_to && bytes > done) std::memcpy(static_cast<uint8_t*>(copy_to) + done, p + done, bytes - done);
  for (; i < nvec; i++) accumulate_vec_scalar(s, p + i * 32, (int)(i & 31));   // (< 32 vectors: no scramble falls due)
  const size_t rest = bytes - nvec * 32;
  if (rest) {
    uint8_t last[32] = {0};
    std::memcpy(last, p + nvec * 32, rest);
    accumulate_vec_scalar(s, last, (int)(nvec & 31));
  }
  return finish(s, bytes);
}
#endif

inline bool use_avx2() {
#if defined(__x86_64__)
  static const bool on = __builtin_cpu_supports("avx2") && std::getenv("ODO_HOST_FP_SCALAR") == nullptr;
  return on;
#else
  return false;
#endif
}
inline uint64_t run(const void* src, size_t bytes, void* copy_to) {
#if defined(__x86_64__)
  if (use_avx2()) return run_avx2(src, bytes, copy_to);
#endif
  return run_scalar(src, bytes, copy_to);
}
// (Round 5 tried helper threads for these passes — the image cut into 128 KB pieces hashed by the caller and three spinning helpers.
// Measured on the EPYC 9575F box, and dropped: the passes got ~15 % faster, but the next refill of the image by the caller's thread —
// the runner's convertTo, ref: run_odometry_kitti_offline.cpp:348 — went from 44 to 420-690 us, because its cache lines were by then
// shared by four cores; cv::Mat frame rate 1 230-1 400 -> 780-940. One thread, one cache.)
inline uint64_t fold(uint64_t h, uint64_t piece) {
  h = (h ^ piece) * 0xD6E8FEB86659FD93ull;
  return h ^ (h >> 29);
}
// rows of row_bytes bytes, src_pitch apart; copy_to (optional): rows dst_pitch apart.
inline uint64_t image(const void* src, size_t src_pitch, size_t row_bytes, int rows, void* copy_to, size_t dst_pitch) {
  if (rows <= 0 || row_bytes == 0) return 0;
  if (src_pitch == row_bytes && (!copy_to || dst_pitch == row_bytes)) return run(src, row_bytes * (size_t)rows, copy_to);
  uint64_t h = 0x243F6A8885A308D3ull;   // a view: row by row
  for (int y = 0; y < rows; y++)
    h = fold(h, run(static_cast<const uint8_t*>(src) + (size_t)y * src_pitch, row_bytes,
                    copy_to ? static_cast<uint8_t*>(copy_to) + (size_t)y * dst_pitch : nullptr));
  return h;
}

}  // namespace hostfp
