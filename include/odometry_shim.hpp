// odometry_shim.hpp — the reference's public C++ surface over the C ABI in odometry_hip.h.
//
// Header-only drop-in for the four headers the runner includes
//   include/image_pyramid.h   (ref: :14-69)    ImagePyramid, DepthPyramid
//   include/lm_optimizer.h    (ref: :24-115)   LevenbergMarquardtOptimizer
//   include/depth_estimate.h  (ref: :24-121)   DepthEstimator
//   include/keyframe.h        (ref: :17-60)    KeyFrame
//   include/data_types.h      (ref: :10-28)    Affine4f, OptimizerStatus, GlobalStatus, PixelType
//   include/camera.h          (ref: :16-119)   CameraPyramid (the cv::stereoRectify set-up step stays with the caller)
// Same namespace, class names, constructor / method signatures, copy semantics and error behaviour (status ints,
// messages on std::cout, no exceptions, Solve returns the pseudo-identity on failure), so
// run_odometry_kitti_offline.cpp compiles against it unchanged once its includes point here.
//
// Images: with -DODOMETRY_SHIM_WITH_OPENCV the classes take cv::Mat exactly like the reference. Without OpenCV (this
// repository's build image has none) a minimal odometry::Mat with the members the runner uses stands in.
// Poses: with -DODOMETRY_SHIM_WITH_EIGEN Affine4f is Eigen::Matrix<float,4,4> (column-major, as in the reference);
// otherwise a 16-float column-major struct with operator()(row, col).
#ifndef ODOMETRY_SHIM_HPP
#define ODOMETRY_SHIM_HPP

#include <atomic>
#ifdef ODOMETRY_SHIM_PHASES
#include <chrono>
#include <cstdio>
#endif
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <vector>

#include "odometry_hip.h"

#ifdef ODOMETRY_SHIM_WITH_OPENCV
#include <opencv2/core.hpp>
#endif
#ifdef ODOMETRY_SHIM_WITH_EIGEN
#include <Eigen/Core>
#endif
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(ODOMETRY_SHIM_WITH_OPENCV) && defined(__GLIBC__)
#include <malloc.h>
#endif

#ifndef PixelType
#ifdef ODOMETRY_SHIM_WITH_OPENCV
#define PixelType CV_32F
#else
#define PixelType 5 /* CV_32F */
#endif
#endif

namespace odometry {

typedef int OptimizerStatus;  // ref: include/data_types.h:27
typedef int GlobalStatus;     // ref: include/data_types.h:28

// ------------------------------------------------------------------------------------------------
#ifdef ODOMETRY_SHIM_WITH_EIGEN
typedef Eigen::Matrix<float, 4, 4> Affine4f;  // ref: include/data_types.h:24
inline const float* affine_data(const Affine4f& a) { return a.data(); }
inline float* affine_data(Affine4f& a) { return a.data(); }
#else
struct Affine4f {  // 16 fp32, column-major like Eigen's default
  float m[16];
  float& operator()(int r, int c) { return m[c * 4 + r]; }
  float operator()(int r, int c) const { return m[c * 4 + r]; }
  static Affine4f Identity() {
    Affine4f a;
    std::memset(a.m, 0, sizeof(a.m));
    a.m[0] = a.m[5] = a.m[10] = a.m[15] = 1.0f;
    return a;
  }
};
inline const float* affine_data(const Affine4f& a) { return a.m; }
inline float* affine_data(Affine4f& a) { return a.m; }
#endif

// ------------------------------------------------------------------------------------------------
// What the classes did on this thread (diagnostics and tests; the cv::Mat build counts all of it, the stand-in build the look-ahead).
struct ShimStats {
  unsigned long uploads = 0;          // images staged and sent to the device
  unsigned long fingerprints = 0;     // full-image fingerprint passes
  unsigned long unchanged = 0;        // uses of a Mat whose mirror was found current
  unsigned long changed = 0;          // uses of a known Mat whose content had changed since its upload
  unsigned long early_adopted = 0;    // ComputeDepth calls that found their job started ahead
  unsigned long early_dropped = 0;    // jobs started ahead for images that then did not come (or had changed)
  unsigned long delivered = 0;        // output images copied into host memory
  unsigned long outputs_prepared = 0; // ComputeDepth calls whose output images had been built while Solve waited (cv::Mat build)
  unsigned long verify_failures = 0;  // ODOMETRY_SHIM_VERIFY_MIRRORS: mirrors that differed from an "unchanged" Mat
};
namespace detail { inline ShimStats& stats() { static thread_local ShimStats s; return s; } }
inline const ShimStats& shim_stats() { return detail::stats(); }
// ------------------------------------------------------------------------------------------------
#ifdef ODOMETRY_SHIM_WITH_OPENCV
typedef cv::Mat Mat;
typedef cv::Size Size;
typedef cv::Scalar Scalar;
enum { kInterLinear = cv::INTER_LINEAR, kBorderConstant = cv::BORDER_CONSTANT, kMap32FC1 = CV_32FC1 };
inline double scalar0(const Scalar& s) { return s[0]; }
// A cv::Mat knows nothing about the device and reports no writes. What the stand-in Mat below carries inside — a device mirror, a
// content stamp — is kept here in a per-thread table of RECORDS keyed by the pixels' address and geometry, made safe by two rules:
//   * the records of a STEREO PAIR (ComputeDepth's two inputs) hold a header copy of the Mat (cv::Mat headers share their buffer by
//     reference count), so the buffer can neither be released nor its address be handed to another image while the record lives —
//     the classes read such a buffer again on their own initiative (the stereo partner's upload is started from Solve, before
//     ComputeDepth names it). All other records hold nothing: their pixels are only ever looked at while the caller hands the Mat in,
//     and holding the runner's per-frame output Mats (:226-228) would keep the allocator from recycling their buffers (fresh pages
//     every frame: page faults worth 200 us);
//   * a record remembers the 64-bit FINGERPRINT of the bytes that were uploaded (odo_dev_upload_fp_async: fused with the staging
//     copy). Every later use of the Mat in a call into a shim class fingerprints all its pixels again (odo_host_fingerprint: one read
//     of the image, ~20-40 us per KITTI frame) before the mirror — or anything derived from it: the :251 pyramid, a ComputeDepth started
//     ahead — is used; a caller who rewrote the buffer in between gets a fresh upload and fresh results. A cheap sample of 48 words
//     goes first: a refilled frame is recognised as changed without the full pass.
// ODOMETRY_SHIM_VERIFY_MIRRORS=1 (environment) additionally downloads the mirror and compares it byte for byte whenever the
// fingerprint says "unchanged", and counts disagreements (ShimStats::verify_failures; must stay 0).
// Outputs (ComputeDepth's left_val / left_disp / left_dep): by default all three are in host memory when ComputeDepth returns, as in
// the reference (src/depth_estimate.cpp:33-78). With ODOMETRY_SHIM_LAZY_OUTPUTS=1 left_disp and left_dep stay on the device — the
// runner never reads them on the host: left_dep only goes to DepthPyramid (run_odometry_kitti_offline.cpp:252) — until
// odometry::Download(mat) is called for them; left_val (summed by the runner, :236) is always delivered.
namespace detail {
inline odo_ctx* context();
inline odo_ctx* side_context();
inline unsigned long long next_stamp() { static std::atomic<unsigned long long> s{0}; return ++s; }
// Every public method of a shim class is one "call": a Mat is checked at most once per call (the caller cannot write while we run).
inline unsigned long long& call_epoch() { static thread_local unsigned long long e = 1; return e; }
struct CallScope { CallScope() { ++call_epoch(); } };
constexpr int kSampleWords = 48;
struct MatBuf : std::enable_shared_from_this<MatBuf> {
  Mat keep;                  // header copy: pins the pixel buffer
  bool keepalive = false;    // false for a header over user data (cv::Mat::u == nullptr): nothing pins those bytes — never read ahead
  uint8_t* host = nullptr;
  size_t bytes = 0, pitch = 0, row_bytes = 0;   // bytes = rows * row_bytes (the dense image on the device)
  int rows = 0, cols = 0, type = 0;
  void* dev = nullptr;
  int dev_async = 0;
  odo_ctx *ctx = nullptr, *side = nullptr;
  bool host_valid = true, dev_valid = false, side_pending = false;
  bool pinned = false;
  unsigned long long fp = 0;
  bool fp_known = false;
  unsigned long long sample[kSampleWords];
  unsigned long long stamp = 0, checked = 0;
  MatBuf* next = nullptr;    // (a cv::Mat has no creation order to guess a partner from)
  std::shared_ptr<MatBuf> successor() { return nullptr; }
  void bind() { if (ctx != context()) { if (ctx) free_mirror(); ctx = context(); side = side_context(); } }
  void free_mirror() {
    if (!dev) return;
    if (side_pending) odo_ctx_stream_wait(ctx, side);
    odo_dev_free_async(ctx, dev, bytes, dev_async);
    dev = nullptr; dev_valid = false; side_pending = false;
  }
  ~MatBuf() { free_mirror(); }
  const uint8_t* sample_at(int i) const {
    const int y = (int)(((long long)i * rows) / kSampleWords);
    const size_t span = row_bytes >= 8 ? row_bytes - 8 : 0;
    const size_t x = span ? (((size_t)i * 2654435761u) % span) & ~(size_t)7 : 0;
    return host + (size_t)y * pitch + x;
  }
  void take_sample() { if (row_bytes >= 8) for (int i = 0; i < kSampleWords; i++) std::memcpy(&sample[i], sample_at(i), 8); }
  bool sample_same() const {
    if (row_bytes < 8) return true;
    for (int i = 0; i < kSampleWords; i++) { unsigned long long w; std::memcpy(&w, sample_at(i), 8); if (w != sample[i]) return false; }
    return true;
  }
  void content_changed() { stamp = next_stamp(); dev_valid = false; side_pending = false; fp_known = false; }
  // Is the mirror (and everything keyed by `stamp`) still the image in host memory? Once per call into a shim class.
  void validate() {
    if (checked == call_epoch()) return;
    checked = call_epoch();
    if (!host_valid) return;              // (lazy outputs: the device copy is the image; the host bytes are stale by contract)
    if (!fp_known) { content_changed(); return; }
    bool same = sample_same();
    if (same) { stats().fingerprints++; same = odo_host_fingerprint(host, pitch, row_bytes, rows) == fp; }
    if (!same) { stats().changed++; content_changed(); return; }
    stats().unchanged++;
    static const bool verify = std::getenv("ODOMETRY_SHIM_VERIFY_MIRRORS") != nullptr;
    if (verify && dev && dev_valid) {
      std::vector<uint8_t> tmp(bytes);
      if (side_pending) { odo_ctx_stream_wait(ctx, side); side_pending = false; }
      odo_dev_download(ctx, tmp.data(), dev, bytes);
      for (int y = 0; y < rows; y++)
        if (std::memcmp(tmp.data() + (size_t)y * row_bytes, host + (size_t)y * pitch, row_bytes) != 0) {
          stats().verify_failures++;
          std::cout << "odometry_hip: a device mirror differs from a cv::Mat whose fingerprint had not changed" << std::endl;
          content_changed();
          return;
        }
    }
  }
  // The image goes to the device through the pinned staging ring of `to` (the main or the side context), fingerprinted on the way.
  bool upload(odo_ctx* to) {
    if (odo_dev_upload_fp_async(to, dev, host, pitch, row_bytes, rows, &fp) != 0) return false;
    stats().uploads++;
    take_sample();
    fp_known = true; dev_valid = true; host_valid = true;
    side_pending = (to == side);
    return true;
  }
  // The device copy (a kernel's output) comes to the host now: synchronous.
  void deliver_now() {
    if (!dev) return;
    if (side_pending) { odo_ctx_stream_wait(ctx, side); side_pending = false; }
    if (pitch == row_bytes) odo_dev_download(ctx, host, dev, bytes);
    else {   // (a view: through a dense temporary)
      std::vector<uint8_t> tmp(bytes);
      odo_dev_download(ctx, tmp.data(), dev, bytes);
      for (int y = 0; y < rows; y++) std::memcpy(host + (size_t)y * pitch, tmp.data() + (size_t)y * row_bytes, row_bytes);
    }
    stats().fingerprints++;
    fp = odo_host_fingerprint(host, pitch, row_bytes, rows);
    finish_delivery();
  }
  // ... or was staged in page-locked memory beside the pose LM and is copied out (fingerprinted on the way).
  void deliver_from(const void* staged) {
    fp = odo_host_copy_fingerprint(host, pitch, staged, row_bytes, row_bytes, rows);
    finish_delivery();
  }
  void finish_delivery() { stats().delivered++; take_sample(); fp_known = true; host_valid = true; checked = call_epoch(); }
};
// least recently used first; the oldest record (and its mirror, and its hold on the caller's buffer) goes when the table is full
struct MatTable {
  std::vector<std::shared_ptr<MatBuf>> recs;
  // hold: the Mat comes in as an input (see above)
  std::shared_ptr<MatBuf> find(const Mat& m, bool create, bool hold = false) {
    if (m.empty()) return nullptr;
    for (size_t i = recs.size(); i-- > 0;) {
      MatBuf& r = *recs[i];
      if (r.host == m.data && r.rows == m.rows && r.cols == m.cols && r.pitch == (size_t)m.step && r.type == m.type()) {
        std::shared_ptr<MatBuf> hit = recs[i];
        if (i + 1 != recs.size()) { recs.erase(recs.begin() + (long)i); recs.push_back(hit); }
        if (hold && !hit->keepalive && m.u != nullptr) { hit->keep = m; hit->keepalive = true; }
        return hit;
      }
    }
    if (!create) return nullptr;
    auto b = std::make_shared<MatBuf>();
    if (hold && m.u != nullptr) { b->keep = m; b->keepalive = true; }
    b->host = m.data; b->rows = m.rows; b->cols = m.cols; b->type = m.type();
    b->pitch = (size_t)m.step; b->row_bytes = (size_t)m.cols * m.elemSize(); b->bytes = b->row_bytes * (size_t)m.rows;
    b->stamp = next_stamp();
    static const size_t cap = std::getenv("ODOMETRY_SHIM_MAT_RECORDS") ? (size_t)std::atoi(std::getenv("ODOMETRY_SHIM_MAT_RECORDS")) : 24;
    while (recs.size() >= (cap < 4 ? 4 : cap)) recs.erase(recs.begin());
    recs.push_back(b);
    return b;
  }
};
inline MatTable& mat_table() { static thread_local MatTable t; return t; }
inline bool lazy_outputs() { static const bool on = std::getenv("ODOMETRY_SHIM_LAZY_OUTPUTS") != nullptr; return on; }
// the stand-in's helpers of the same names: the partner's mirror block first (before the main stream is marked), its upload later
inline void prefetch_reserve(const std::shared_ptr<MatBuf>& sp) {
  MatBuf& b = *sp;
  if (!b.keepalive) return;
  b.bind();
  if (!b.dev && odo_dev_alloc_async(b.ctx, b.bytes, &b.dev, &b.dev_async) != 0) b.dev = nullptr;
}
inline void prefetch_to_device(const std::shared_ptr<MatBuf>& sp) {
  MatBuf& b = *sp;
  if (!b.keepalive || !b.dev || b.ctx != context()) return;
  b.validate();                      // (the caller has usually refilled it since the last frame: the sample says so at once)
  if (!b.dev_valid && b.host_valid) (void)b.upload(b.side);
}
}  // namespace detail
// ODOMETRY_SHIM_LAZY_OUTPUTS=1: brings a ComputeDepth output that was left on the device into the Mat's host memory.
inline void Download(Mat& m) {
  auto b = detail::mat_table().find(m, false);
  if (b && !b->host_valid && b->dev && b->dev_valid) b->deliver_now();
}
#else
enum { CV_8U = 0, CV_32F = 5, CV_64F = 6, CV_32FC1 = 5 };
enum { INTER_LINEAR = 1, BORDER_CONSTANT = 0 };
enum { kInterLinear = INTER_LINEAR, kBorderConstant = BORDER_CONSTANT, kMap32FC1 = CV_32FC1 };
struct Size {  // cv::Size: (width, height)
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {  // cv::Scalar: only the first channel matters for single-channel images
  double val[4] = {0, 0, 0, 0};
  Scalar() {}
  Scalar(double v0) { val[0] = v0; }
};
inline double scalar0(const Scalar& s) { return s.val[0]; }
// The subset of cv::Mat the runner and the estimators touch: rows, cols, type(), at<T>, ptr<T>, isContinuous,
// copyTo / clone, ref-counted header copies.
//
// Unlike cv::Mat this stand-in knows about the device. Its pixel buffer is page-locked host memory (uploads from it are
// plain asynchronous DMAs) and carries an optional DEVICE MIRROR with two validity flags:
//   * the shim's classes read inputs through the mirror — a frame that goes to ImagePyramid (:205), ComputeDepth (:229)
//     and ImagePyramid again (:251) crosses PCIe once, not three times;
//   * ComputeDepth leaves val / disp / dep on the device and only marks the host side stale: the bytes come back when (if)
//     host code looks at them. The runner only hands left_dep to DepthPyramid (:252), which takes the mirror.
// Correctness does not rest on trust: every non-const access (ptr<T>() / at<T>() on a non-const Mat) waits for a pending
// upload out of the buffer, brings the host copy up to date and invalidates the mirror; every const access brings the host
// copy up to date. Header copies share buffer and mirror, as cv::Mat headers share data.
// ONE RULE for callers: a pointer obtained from non-const ptr<T>() must not be kept across a call into a shim class and
// written through afterwards — the shim cannot see such a write (the mirror would go stale). Fetch the pointer again after
// the call, as the reference's own code does (it calls ptr<T>() / at<T>() per row / per pixel).
namespace detail {
inline void convert_8u_32f(const uint8_t* s, float* d, size_t n) {   // (whatever optimisation level the includer compiles with)
  size_t i = 0;
#if defined(__SSE2__)
  const __m128i z = _mm_setzero_si128();
  for (; i + 16 <= n; i += 16) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
    const __m128i lo = _mm_unpacklo_epi8(v, z), hi = _mm_unpackhi_epi8(v, z);
    _mm_storeu_ps(d + i, _mm_cvtepi32_ps(_mm_unpacklo_epi16(lo, z)));
    _mm_storeu_ps(d + i + 4, _mm_cvtepi32_ps(_mm_unpackhi_epi16(lo, z)));
    _mm_storeu_ps(d + i + 8, _mm_cvtepi32_ps(_mm_unpacklo_epi16(hi, z)));
    _mm_storeu_ps(d + i + 12, _mm_cvtepi32_ps(_mm_unpackhi_epi16(hi, z)));
  }
#endif
  for (; i < n; i++) d[i] = (float)s[i];
}
inline odo_ctx* context();
inline odo_ctx* side_context();
// Content stamps are process-wide (a Mat may be created on one thread and used on another: equal stamps must mean equal content).
inline unsigned long long next_stamp() { static std::atomic<unsigned long long> s{0}; return ++s; }
struct MatBuf;
// The live buffers of a thread in creation order (the stereo-partner guess). A buffer may be destroyed on another thread than the
// one that created it: the list is its own heap object (kept alive by its members) and every link is touched under its lock.
struct BufList { std::mutex mu; MatBuf* tail = nullptr; };
inline const std::shared_ptr<BufList>& buf_list() { static thread_local std::shared_ptr<BufList> l = std::make_shared<BufList>(); return l; }
struct MatBuf : std::enable_shared_from_this<MatBuf> {
  uint8_t* host = nullptr;
  size_t bytes = 0;
  bool pinned = false;
  void* dev = nullptr;        // device mirror (allocated on first use by an estimator)
  int dev_async = 0;
  odo_ctx *ctx = nullptr, *side = nullptr;   // the contexts of the thread the mirror belongs to (its free list, its two streams)
  bool host_valid = true, dev_valid = false;
  unsigned long upload_ticket = 0;  // != 0: an asynchronous DMA issued with this ticket may still be reading `host`
  bool upload_on_side = false;      // ... on the side context's stream (a stereo partner uploaded ahead of its first use)
  // (a buffer that wanders to another thread's classes gives up the mirror it had on the first thread's streams)
  void bind() { if (ctx != context()) { if (ctx) { wait_upload(); free_mirror(); } ctx = context(); side = side_context(); } }
  void free_mirror() {              // stream-ordered release to the owning context's free list: behind what either stream has queued on the block
    if (!dev) return;
    if (side_pending) odo_ctx_stream_wait(ctx, side);
    odo_dev_free_async(ctx, dev, bytes, dev_async);
    dev = nullptr; dev_valid = false; side_pending = false;
  }
  bool side_pending = false;        // the main stream has not been ordered behind that upload yet
  unsigned long long stamp;         // names the CONTENT: renewed whenever host code may have written or a kernel has (keys the
                                    // pyramid cache and the prepared front half of ComputeDepth)
  std::shared_ptr<BufList> list;             // the creating thread's list
  MatBuf *prev = nullptr, *next = nullptr;   // ... in creation order (stereo-partner guess); guarded by list->mu
  std::shared_ptr<MatBuf> successor() {      // the buffer created right after this one, if it is still alive
    std::lock_guard<std::mutex> lk(list->mu);
    return next ? next->weak_from_this().lock() : nullptr;
  }
  bool fill_pending = false;                 // Mat(rows, cols, type, value): the fill is done on the first host access
  double fill_value = 0.0;
  int fill_type = 0;
  void materialize_fill() {
    if (!fill_pending) return;
    fill_pending = false;
    if (fill_type == 5 /* CV_32F */) for (size_t i = 0; i < bytes / 4; i++) reinterpret_cast<float*>(host)[i] = (float)fill_value;
    else if (fill_type == 6 /* CV_64F */) for (size_t i = 0; i < bytes / 8; i++) reinterpret_cast<double*>(host)[i] = fill_value;
    else std::memset(host, (int)fill_value, bytes);
  }
  explicit MatBuf(size_t n);
  ~MatBuf();
  MatBuf(const MatBuf&) = delete;
  MatBuf& operator=(const MatBuf&) = delete;
  void wait_upload() {  // the DMA out of `host` (if any) has finished: the block may be rewritten / recycled
    if (upload_ticket) { odo_ctx_upload_wait(upload_on_side ? side : ctx, upload_ticket); upload_ticket = 0; }
  }
  void sync_host() {  // host copy current (lazy download; a pending fill is void once a kernel has overwritten the image)
    if (!host_valid && dev) { odo_dev_download(ctx, host, dev, bytes); upload_ticket = 0; fill_pending = false; }
    else materialize_fill();
    host_valid = true;
  }
  // Host code may write: the pending upload out of the block is waited for first (cv::Mat semantics let the caller refill a
  // Mat the moment a constructor it was passed to has returned), the host copy is brought up to date, the mirror is void.
  void touch() { wait_upload(); sync_host(); dev_valid = false; side_pending = false; stamp = next_stamp(); }
};
// Page-locked blocks are expensive to create (hipHostMalloc): per-frame Mats recycle them through a small free list.
struct PinnedPool {
  std::vector<std::pair<size_t, void*>> free_;
  ~PinnedPool() { for (auto& e : free_) odo_host_free(e.second); }
  void* get(size_t n) {
    for (size_t i = 0; i < free_.size(); i++)
      if (free_[i].first == n) { void* p = free_[i].second; free_.erase(free_.begin() + (long)i); return p; }
    return odo_host_alloc(n);
  }
  void put(size_t n, void* p) { if (free_.size() < 32) free_.emplace_back(n, p); else odo_host_free(p); }
};
inline PinnedPool& pinned_pool() { static thread_local PinnedPool pool; return pool; }
// Mirrors that are pure caches (host copy current as well) are bounded: the runner keeps every frame of a sequence in
// host memory (ref: run_odometry_kitti_offline.cpp:334-359 fills two vectors), and only the few it touched last need to
// stay on the device. Least recently used first out; a mirror that holds the only current copy is never dropped.
struct MirrorLru {
  std::vector<std::weak_ptr<MatBuf>> order;  // most recent last
  void use(const std::shared_ptr<MatBuf>& b, size_t keep = 16) {
    for (size_t i = 0; i < order.size(); i++) {
      auto q = order[i].lock();
      if (!q || q.get() == b.get()) { order.erase(order.begin() + (long)i); i--; }
    }
    order.push_back(b);
    size_t caches = 0;
    for (auto& w : order) { auto q = w.lock(); if (q && q->dev && q->host_valid) caches++; }
    for (size_t i = 0; i < order.size() && caches > keep; i++) {
      auto q = order[i].lock();
      if (q && q->dev && q->host_valid && q.get() != b.get()) {
        q->free_mirror();
        caches--;
      }
    }
  }
};
inline MirrorLru& mirror_lru() { static thread_local MirrorLru lru; return lru; }
// The upload of a buffer started EARLY, on the side context's stream, for an image the next call is expected to ask for (the right
// image of the pair whose left image just went to ImagePyramid): it then crosses PCIe while the pose LM runs instead of in front
// of ComputeDepth. Only page-locked, current, not yet mirrored buffers; a wrong guess costs one idle-time copy. Mat::device_in
// orders the main stream behind the copy when the image is used; a host write in between waits for it and voids it (touch()).
inline bool prefetch_eligible(const MatBuf& b) { return b.pinned && b.host_valid && !b.dev_valid && !b.upload_ticket && !b.fill_pending; }
// First half, BEFORE the main stream's fill level is marked: the mirror block (a recycled block's earlier use is queued on the main
// stream in front of that mark, and the side stream goes behind the mark).
inline void prefetch_reserve(const std::shared_ptr<MatBuf>& sp) {
  MatBuf& b = *sp;
  if (!prefetch_eligible(b)) return;
  b.bind();
  if (!b.dev && odo_dev_alloc_async(b.ctx, b.bytes, &b.dev, &b.dev_async) != 0) { b.dev = nullptr; return; }
  mirror_lru().use(sp);
}
// Second half: the copy, on the side stream (already ordered behind the mark).
inline void prefetch_to_device(const std::shared_ptr<MatBuf>& sp) {
  MatBuf& b = *sp;
  if (!prefetch_eligible(b) || !b.dev) return;
  if (odo_dev_upload_async(b.side, b.dev, b.host, b.bytes) != 0) return;
  b.upload_ticket = odo_ctx_upload_ticket(b.side);
  b.upload_on_side = true;
  b.side_pending = true;
  b.dev_valid = true;
}
inline MatBuf::MatBuf(size_t n) : bytes(n), stamp(next_stamp()), list(buf_list()) {
  if (n >= (64u << 10)) { host = static_cast<uint8_t*>(pinned_pool().get(n)); pinned = host != nullptr; }
  if (!host) host = static_cast<uint8_t*>(std::malloc(n ? n : 1));
  std::lock_guard<std::mutex> lk(list->mu);
  prev = list->tail;
  if (prev) prev->next = this;
  list->tail = this;
}
inline MatBuf::~MatBuf() {
  {
    std::lock_guard<std::mutex> lk(list->mu);
    if (prev) prev->next = next;
    if (next) next->prev = prev; else list->tail = prev;
  }
  free_mirror();
  if (pinned) {
    wait_upload();  // a DMA may still be reading the block (a retired ticket costs nothing)
    pinned_pool().put(bytes, host);
  } else {
    std::free(host);
  }
}
}  // namespace detail

class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) : Mat(r, c, type, s.val[0]) {}   // cv::Mat(rows, cols, type, cv::Scalar)
  Mat(int r, int c, int type, double fill) {
    create(r, c, type);
    // filled when (if) somebody looks: the runner's per-frame output Mats (ref: run_odometry_kitti_offline.cpp:226) go straight to
    // ComputeDepth, which overwrites them on the device — 466 KB of memset per frame that nobody reads
    buf_->fill_pending = true; buf_->fill_value = fill; buf_->fill_type = type_;
  }
  void create(int r, int c, int type) {
    if (buf_ && rows == r && cols == c && type_ == type) return;   // cv::Mat::create keeps a fitting buffer
    rows = r; cols = c; type_ = type;
    buf_ = std::make_shared<detail::MatBuf>((size_t)r * c * elemSize());
  }
  int type() const { return type_; }
  int channels() const { return 1; }
  size_t elemSize() const { return type_ == CV_64F ? 8 : type_ == CV_32F ? 4 : 1; }
  bool isContinuous() const { return true; }
  bool empty() const { return !buf_ || rows == 0 || cols == 0; }
  template <class T> T* ptr(int y = 0) { buf_->touch(); return reinterpret_cast<T*>(buf_->host + (size_t)y * cols * elemSize()); }
  template <class T> const T* ptr(int y = 0) const {
    buf_->sync_host();
    return reinterpret_cast<const T*>(buf_->host + (size_t)y * cols * elemSize());
  }
  template <class T> T& at(int y, int x) { return ptr<T>(y)[x]; }
  template <class T> const T& at(int y, int x) const { return ptr<T>(y)[x]; }
  void copyTo(Mat& dst) const {
    dst.create(rows, cols, type_);
    if (buf_ && dst.buf_ != buf_) { buf_->sync_host(); dst.buf_->touch(); std::memcpy(dst.buf_->host, buf_->host, buf_->bytes); }
  }
  // the one conversion the runner uses: imread's 8-bit image to PixelType (ref: run_odometry_kitti_offline.cpp:348,358); dst keeps its
  // buffer when it fits (the runner refills the same two Mats every frame)
  void convertTo(Mat& dst, int rtype) const {
    if (rtype == type_) { copyTo(dst); return; }
    if (empty() || type_ != CV_8U || rtype != CV_32F) { std::cout << "odometry_hip: Mat::convertTo converts CV_8U to CV_32F only" << std::endl; return; }
    dst.create(rows, cols, rtype);
    detail::convert_8u_32f(ptr<uint8_t>(), dst.ptr<float>(), (size_t)rows * cols);
    // (Round 5, measured and dropped: sending dst up from HERE, on the side stream, so that the image has crossed PCIe by the time
    //  ImagePyramid asks for it — the Solve behind it starts 8 us sooner and the three runtime calls per image cost load_data 16 us
    //  each: 2 085-2 304 against 2 287-2 324 frames/s in the runner's load-per-frame shape.)
  }
  Mat clone() const { Mat m; copyTo(m); return m; }
  // -- device side, for the shim's classes only --
  // Device copy of the pixels for reading (uploads when the mirror is missing or stale). Null on failure.
  const void* device_in() const {
    detail::MatBuf& b = *buf_;
    b.bind();
    if (!b.dev && odo_dev_alloc_async(b.ctx, b.bytes, &b.dev, &b.dev_async) != 0) return nullptr;
    detail::mirror_lru().use(buf_);
    if (!b.dev_valid) {
      b.materialize_fill();
      if (odo_dev_upload_async(b.ctx, b.dev, b.host, b.bytes) != 0) return nullptr;
      b.upload_ticket = b.pinned ? odo_ctx_upload_ticket(b.ctx) : 0;
      b.upload_on_side = false;
      b.dev_valid = true;
    } else if (b.side_pending) {   // uploaded ahead on the side stream: the main stream's work goes behind that copy
      if (odo_ctx_stream_wait(b.ctx, b.side) != 0) return nullptr;
      b.side_pending = false;
    }
    return b.dev;
  }
  unsigned long long content_stamp() const { return buf_ ? buf_->stamp : 0; }
  // Device buffer a kernel is about to overwrite completely: afterwards the device holds the truth, the host copy is stale.
  void* device_out() {
    detail::MatBuf& b = *buf_;
    b.bind();
    b.wait_upload();
    if (b.side_pending) { odo_ctx_stream_wait(b.ctx, b.side); b.side_pending = false; }
    if (!b.dev && odo_dev_alloc_async(b.ctx, b.bytes, &b.dev, &b.dev_async) != 0) return nullptr;
    b.dev_valid = true;
    b.host_valid = false;
    b.fill_pending = false;
    b.stamp = detail::next_stamp();
    return b.dev;
  }
  // A device block a kernel has (or will have, in stream order) written completely becomes this Mat's mirror: as device_out(),
  // without allocating (ComputeDepth started ahead writes into blocks reserved for its outputs before their Mats exist).
  void adopt_device(void* block, int async_flag) {
    detail::MatBuf& b = *buf_;
    b.bind();
    b.wait_upload();
    b.free_mirror();
    b.dev = block; b.dev_async = async_flag;
    b.dev_valid = true; b.host_valid = false; b.fill_pending = false; b.side_pending = false;
    b.stamp = detail::next_stamp();
  }
  size_t device_bytes() const { return buf_ ? buf_->bytes : 0; }
  const std::shared_ptr<detail::MatBuf>& buffer() const { return buf_; }
 private:
  int type_ = CV_32F;
  std::shared_ptr<detail::MatBuf> buf_;
};
namespace detail { inline bool lazy_outputs() { return false; } }
inline void Download(Mat& m) { const Mat& c = m; if (!c.empty()) (void)c.ptr<uint8_t>(); }   // (any const access brings the host copy up to date)
#endif

// ------------------------------------------------------------------------------------------------
namespace detail { inline odo_ctx* context(); inline odo_ctx* side_context(); }

// ref: include/camera.h:16-119. Raw calibration, the rectified intrinsic pyramid and the undistort + rectify remap, on
// the device. The runner passes nullptr for its cameras (ref: run_odometry_kitti_offline.cpp:51-52) and the
// optimizer / estimator never dereference them (the reference hard-codes the KITTI-00 constants, "only for debug now").
class CameraPyramid {
 public:
  CameraPyramid() = delete;
  CameraPyramid(int levels, double fx, double fy, double f_theta, double cx, double cy, double k1, double k2, double r1,
                double r2, double sensor_width, double sensor_height, int resolution_width, int resolution_height)
      : resolution_width_(resolution_width), resolution_height_(resolution_height), sensor_width_(sensor_width),
        sensor_height_(sensor_height), pixels_per_mm_x_(resolution_width / sensor_width),      // ref: src/camera.cpp:36
        pixels_per_mm_y_(resolution_height / sensor_height) {                                  // ref: :37
    intrinsic_raw_ = Mat(3, 3, CV_64F, 0.0);
    intrinsic_raw_.at<double>(0, 0) = fx; intrinsic_raw_.at<double>(1, 1) = fy; intrinsic_raw_.at<double>(0, 2) = cx;
    intrinsic_raw_.at<double>(1, 2) = cy; intrinsic_raw_.at<double>(2, 2) = 1; intrinsic_raw_.at<double>(0, 1) = f_theta;
    distortion_param_ = Mat(1, 4, CV_64F, 0.0);
    distortion_param_.at<double>(0, 0) = k1; distortion_param_.at<double>(0, 1) = k2;
    distortion_param_.at<double>(0, 2) = r1; distortion_param_.at<double>(0, 3) = r2;
    levels_ = levels;
    if (odo_camera_create(detail::context(), levels, fx, fy, f_theta, cx, cy, k1, k2, r1, r2, sensor_width, sensor_height,
                          resolution_width, resolution_height, &cam_) != 0)
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
  }
  ~CameraPyramid() { odo_camera_destroy(cam_); }
  CameraPyramid(const CameraPyramid&) = delete;
  CameraPyramid& operator=(const CameraPyramid&) = delete;

  // ref: include/camera.h:56-57, src/camera.cpp:40-69. rectify_rotation 3x3 and new_intrinsic 3x4, CV_64F.
  void ConfigureCamera(const Mat& rectify_rotation, const Mat& new_intrinsic, const Size& new_size, int map_type = kMap32FC1,
                       bool use_int_map = false) {
    if (map_type != kMap32FC1 || use_int_map) {
      std::cout << "odometry_hip: only CV_32FC1 floating-point remaps are implemented" << std::endl;
      return;
    }
    double R[9], P[12];
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R[i * 3 + j] = rectify_rotation.at<double>(i, j);
      for (int j = 0; j < 4; j++) P[i * 4 + j] = new_intrinsic.at<double>(i, j);
    }
    if (odo_camera_configure(cam_, R, P, new_size.width, new_size.height) != 0) {
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
      return;
    }
    intrinsic_.clear();
    for (int l = 0; l < levels_; l++) {
      double k[5];
      odo_camera_intrinsics(cam_, l, k);
      Mat m(3, 3, CV_64F, 0.0);
      m.at<double>(0, 0) = k[0]; m.at<double>(1, 1) = k[1]; m.at<double>(0, 1) = k[2];
      m.at<double>(0, 2) = k[3]; m.at<double>(1, 2) = k[4]; m.at<double>(2, 2) = 1;
      intrinsic_.push_back(m);
    }
    new_size_ = new_size;
  }

  // ref: include/camera.h:68, src/camera.cpp:71-82. dst is (re)allocated to the configured size.
  GlobalStatus UndistortRectify(const Mat& src_raw, Mat& dst, int interpolation = kInterLinear,
                                int borderMode = kBorderConstant, const Scalar& borderValue = Scalar()) {
    if (src_raw.rows != 480 || src_raw.cols != 640) {  // ref: src/camera.cpp:74-77
      std::cout << "camera raw image is not 480x640!" << std::endl;
      return -1;
    }
    if (interpolation != kInterLinear || borderMode != kBorderConstant || src_raw.type() != PixelType ||
        !src_raw.isContinuous()) {
      std::cout << "odometry_hip: UndistortRectify takes continuous CV_32F, INTER_LINEAR, BORDER_CONSTANT" << std::endl;
      return -1;
    }
    dst.create(new_size_.height, new_size_.width, PixelType);
    if (odo_camera_undistort_rectify(cam_, src_raw.ptr<float>(), src_raw.rows, src_raw.cols, dst.ptr<float>(),
                                     (float)scalar0(borderValue)) != 0) {
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
      return -1;
    }
    return 0;
  }

  // accessors, ref: include/camera.h:73-105
  float fx_float(int level) { return float(intrinsic_[level].at<double>(0, 0)); }
  float fy_float(int level) { return float(intrinsic_[level].at<double>(1, 1)); }
  float f_theta_float(int level) { return float(intrinsic_[level].at<double>(0, 1)); }
  float cx_float(int level) { return float(intrinsic_[level].at<double>(0, 2)); }
  float cy_float(int level) { return float(intrinsic_[level].at<double>(1, 2)); }
  float f_meters_float(int level) { return float(intrinsic_[level].at<double>(0, 0) / pixels_per_mm_x_); }
  double fx_double(int level) { return intrinsic_[level].at<double>(0, 0); }
  double fy_double(int level) { return intrinsic_[level].at<double>(1, 1); }
  double f_theta_double(int level) { return intrinsic_[level].at<double>(0, 1); }
  double cx_double(int level) { return intrinsic_[level].at<double>(0, 2); }
  double cy_double(int level) { return intrinsic_[level].at<double>(1, 2); }
  double f_meters_double(int level) { return intrinsic_[level].at<double>(0, 0) / pixels_per_mm_x_; }
  const Mat& get_intrinsic_rectified(int level) { return intrinsic_[level]; }
  double sensor_w_double() { return sensor_width_; }
  double sensor_h_double() { return sensor_height_; }
  double pixels_per_mm_x_double() { return pixels_per_mm_x_; }
  double pixels_per_mm_y_double() { return pixels_per_mm_y_; }
  int resolution_raw_w() { return resolution_width_; }
  int resolution_raw_h() { return resolution_height_; }
  float sensor_w_float() { return float(sensor_width_); }
  float sensor_h_float() { return float(sensor_height_); }
  float pixels_per_mm_x_float() { return float(pixels_per_mm_x_); }
  float pixels_per_mm_y_float() { return float(pixels_per_mm_y_); }
  const Mat& get_intrinsic_raw() { return intrinsic_raw_; }
  const Mat& get_distortion_coeff() { return distortion_param_; }
  odo_camera* handle() { return cam_; }

 private:
  Mat intrinsic_raw_, distortion_param_;
  int resolution_width_, resolution_height_;
  double sensor_width_, sensor_height_, pixels_per_mm_x_, pixels_per_mm_y_;
  int levels_ = 0;
  std::vector<Mat> intrinsic_;
  Size new_size_;
  odo_camera* cam_ = nullptr;
};

namespace detail {
// One HIP context (stream) per host thread, created on first use — the reference is single-threaded
// (ref: run_odometry_kitti_offline.cpp:3).
inline odo_ctx* context() {
  static thread_local odo_ctx* ctx = nullptr;
  if (!ctx) {
#if defined(ODOMETRY_SHIM_WITH_OPENCV) && defined(__GLIBC__)
    // The runner allocates three fresh 0.5-1.9 MB cv::Mats per frame (ref: run_odometry_kitti_offline.cpp:226-228) and frees them at the
    // end of the frame; with glibc's default trim / mmap thresholds every one of those buffers goes back to the kernel and comes back
    // as fresh pages: ~1 000 page faults per frame (400 us at KITTI size), in ComputeDepth's copy-out and in the Mats' destructors.
    // Telling the allocator to keep such blocks is worth +6 % (runner shape) to +17 % (preloaded frames), but it is a PROCESS-WIDE policy
    // (up to 256 MB never trimmed, 64 MB top padding) and a header has no business changing it silently: it is OPT-IN —
    // ODOMETRY_SHIM_TUNE_MALLOC=1 makes the three mallopt() calls below, once per process, and says so on stderr. The same effect
    // without the header's help: MALLOC_MMAP_THRESHOLD_=33554432 MALLOC_TRIM_THRESHOLD_=268435456 MALLOC_TOP_PAD_=67108864 in the
    // runner's environment.
    static const bool tuned = [] {
      if (!std::getenv("ODOMETRY_SHIM_TUNE_MALLOC")) return false;
      mallopt(M_MMAP_THRESHOLD, 32 << 20);
      mallopt(M_TRIM_THRESHOLD, 256 << 20);
      mallopt(M_TOP_PAD, 64 << 20);
      std::fprintf(stderr, "odometry_shim: ODOMETRY_SHIM_TUNE_MALLOC set: mallopt(M_MMAP_THRESHOLD, 32 MB), mallopt(M_TRIM_THRESHOLD, "
                           "256 MB), mallopt(M_TOP_PAD, 64 MB) for this process\n");
      return true;
    }();
    (void)tuned;
#endif
    const char* dev = std::getenv("ODOMETRY_HIP_DEVICE");
    if (odo_ctx_create(dev ? std::atoi(dev) : 0, &ctx) != 0) {
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
      std::exit(1);  // no CPU fallback
    }
  }
  return ctx;
}
// A second context (stream) of the same thread for work that need not queue behind the pose LM: the upload of the stereo
// partner and the left half of ComputeDepth's front end, both started from ImagePyramid's constructor.
inline odo_ctx* side_context() {
  static thread_local odo_ctx* ctx = nullptr;
  if (!ctx) {
    const char* dev = std::getenv("ODOMETRY_HIP_DEVICE");
    if (odo_ctx_create(dev ? std::atoi(dev) : 0, &ctx) != 0) {
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
      std::exit(1);
    }
  }
  return ctx;
}
// Device view of an input / output image for both Mat flavours: the Mat's mirror (no copy when it is current). The stand-in Mat
// carries its mirror itself and voids it on every non-const access; a cv::Mat's mirror lives in the record table above and is
// checked against the Mat's pixels by fingerprint, once per call into a shim class.
#ifdef ODOMETRY_SHIM_WITH_OPENCV
inline std::shared_ptr<MatBuf> buffer_of(const Mat& m) { return mat_table().find(m, true, false); }
inline std::shared_ptr<MatBuf> held_buffer_of(const Mat& m) { return mat_table().find(m, true, true); }   // ComputeDepth's stereo pair
// (an output left on the device — ODOMETRY_SHIM_LAZY_OUTPUTS — IS held: its record says "the host bytes are stale, do not look", and
//  a recycled address with the same geometry would inherit that)
inline std::shared_ptr<MatBuf> output_buffer_of(const Mat& m, bool hold = false) { return mat_table().find(m, true, hold); }
inline unsigned long long content_stamp(const Mat& m) {
  auto b = buffer_of(m);
  if (!b) return 0;
  b->validate();
  return b->stamp;
}
inline unsigned long long output_stamp(const Mat& m) { auto b = output_buffer_of(m); return b ? b->stamp : 0; }   // (just written by us)
inline size_t device_bytes(const Mat& m) { return m.total() * m.elemSize(); }
// (a cv::Mat may be a view: its rows lie m.step bytes apart. Inputs are staged row by row into a dense device image; outputs must be
// continuous — the callers check, with the reference's own message, ref: src/depth_estimate.cpp:259-263 — and a view handed in as an
// output yields no device buffer instead of bytes in the wrong place.)
inline const void* device_in(const Mat& m) {
  auto sp = buffer_of(m);
  if (!sp) return nullptr;
  MatBuf& b = *sp;
  b.bind();
  b.validate();
  if (!b.dev && odo_dev_alloc_async(b.ctx, b.bytes, &b.dev, &b.dev_async) != 0) { b.dev = nullptr; return nullptr; }
  if (!b.dev_valid) {
    if (!b.upload(b.ctx)) return nullptr;
  } else if (b.side_pending) {   // uploaded ahead on the side stream: the main stream's work goes behind that copy
    if (odo_ctx_stream_wait(b.ctx, b.side) != 0) return nullptr;
    b.side_pending = false;
  }
  return b.dev;
}
inline void* device_out(Mat& m, bool hold = false) {
  if (m.empty() || !m.isContinuous()) return nullptr;
  auto sp = output_buffer_of(m, hold);
  MatBuf& b = *sp;
  b.bind();
  if (b.side_pending) { odo_ctx_stream_wait(b.ctx, b.side); b.side_pending = false; }
  if (!b.dev && odo_dev_alloc_async(b.ctx, b.bytes, &b.dev, &b.dev_async) != 0) { b.dev = nullptr; return nullptr; }
  b.content_changed();
  b.dev_valid = true; b.host_valid = false; b.checked = call_epoch();
  return b.dev;
}
inline void adopt_device(Mat& m, void* block, int async_flag, bool hold = false) {
  auto sp = output_buffer_of(m, hold);
  MatBuf& b = *sp;
  b.bind();
  b.free_mirror();
  b.content_changed();
  b.dev = block; b.dev_async = async_flag;
  b.dev_valid = true; b.host_valid = false; b.checked = call_epoch();
}
struct DevIn {
  const void* dev;
  explicit DevIn(const Mat& m) : dev(device_in(m)) {}
  const void* get() const { return dev; }
};
// An output image: on the device while the call runs, in the Mat's host memory when the call returns — unless `lazy` (see above).
struct DevOut {
  std::shared_ptr<MatBuf> b; void* dev; bool lazy;
  explicit DevOut(Mat& m, bool lazy_ = false) : dev(device_out(m, lazy_)), lazy(lazy_) { if (dev) b = output_buffer_of(m); }
  ~DevOut() { if (b && !lazy && !b->host_valid) b->deliver_now(); }
  DevOut(const DevOut&) = delete;
  DevOut& operator=(const DevOut&) = delete;
  void* get() { return dev; }
};
#else
inline std::shared_ptr<MatBuf> buffer_of(const Mat& m) { return m.buffer(); }
inline std::shared_ptr<MatBuf> held_buffer_of(const Mat& m) { return m.buffer(); }
inline unsigned long long content_stamp(const Mat& m) { return m.content_stamp(); }
inline unsigned long long output_stamp(const Mat& m) { return m.content_stamp(); }
inline size_t device_bytes(const Mat& m) { return m.device_bytes(); }
inline void adopt_device(Mat& m, void* block, int async_flag, bool = false) { m.adopt_device(block, async_flag); }
struct CallScope { CallScope() {} };
struct DevIn {
  const void* dev;
  explicit DevIn(const Mat& m) : dev(m.device_in()) {}
  const void* get() const { return dev; }
};
struct DevOut {
  void* dev;
  explicit DevOut(Mat& m, bool = false) : dev(m.device_out()) {}
  void* get() { return dev; }
};
#endif

struct PyrHandle {
  odo_pyr* p = nullptr;
  std::vector<Mat> host;       // lazily downloaded levels for GetPyramidImage / GetPyramidDepth
  std::vector<char> have;
  ~PyrHandle() { odo_pyramid_destroy(p); }
};
inline const Mat& pyr_level(const std::shared_ptr<PyrHandle>& h, int level) {
  if (!h->have[level]) {
    int r = 0, c = 0;
    odo_pyramid_level_dims(h->p, level, &r, &c);
    h->host[level].create(r, c, PixelType);
    odo_pyramid_download(h->p, level, h->host[level].ptr<float>());
    h->have[level] = 1;
  }
  return h->host[level];
}
// What the drop-in classes do AHEAD of the call that needs it (same launches and copies, earlier; ODOMETRY_SHIM_NO_LOOKAHEAD=1 in
// the environment switches all of it off). The runner's frame is strictly serial — ImagePyramid(left) :205, Solve :215,
// ComputeDepth(left, right) :229, ImagePyramid(left) again :251 — so whatever does not depend on the Solve's result is started
// from ImagePyramid's constructor and runs beside the Solve:
//   * the right image's upload (the partner guess: the Mat ComputeDepth was given with this left Mat last time — the runner
//     refills the same two Mats every frame, ref: :200,334-359 — else the same-sized Mat created right after it);
//   * ComputeDepth(left, right) itself — it does not depend on the pose — into three blocks reserved for its outputs
//     (odo_depth_compute_begin_dev, for the estimator of this thread); the ComputeDepth call of :229 then finds the job under the
//     content stamps of its two images, makes the blocks the mirrors of its output Mats and only waits (a wrong guess, a refilled
//     Mat, another size: the job is dropped and ComputeDepth computes as if nothing had been started);
//     where the whole job cannot be started (no partner known, the depth LM on host-paced step launches) its left half is:
//     blur + point selection of the left image (odo_depth_prepare_left_dev);
//   * and the :251 pyramid is the :205 pyramid (same content stamp, levels and smoothing: the device pyramid is shared).
struct Lookahead {
  struct Early {                                 // ComputeDepth started ahead
    void* blk[3] = {nullptr, nullptr, nullptr};  // val (u8), disp, dep (f32): reserved from the main context's free list
    int async_[3] = {0, 0, 0};
    size_t bytes[3] = {0, 0, 0};
    bool reserved = false, started = false;
    odo_depth* est = nullptr;
    const void *left_dev = nullptr, *right_dev = nullptr;
    unsigned long long left_stamp = 0, right_stamp = 0;
    int rows = 0, cols = 0;
  } early;
  bool on = std::getenv("ODOMETRY_SHIM_NO_LOOKAHEAD") == nullptr;
  std::weak_ptr<MatBuf> last_left, last_right;   // the pair of the last ComputeDepth
  odo_depth* estimator = nullptr;                // the DepthEstimator of this thread (the last one constructed)
  int est_rows = 0, est_cols = 0;                // the frame size it was last used with
  struct Entry { unsigned long long stamp; int levels, smooth, kind; std::weak_ptr<struct PyrHandle> h; bool on_side; unsigned long side_mark; };
  Entry cache[8] = {};
  std::shared_ptr<struct PyrHandle> early_dep_pyr;   // the depth pyramid of the ComputeDepth started ahead (:252, built beside the Solve)
  int dp_levels = 0, dp_smooth = 0;                  // what the last DepthPyramid was constructed with
  unsigned long early_dep_mark = 0;                  // the side stream's position right behind that pyramid's launch
  std::shared_ptr<struct PyrHandle> ahead_pyr;   // the next frame's image pyramid, built ahead on the side stream
  unsigned long long recorded_stamp = 0;         // the image content the pending lookahead was recorded for
  int pending_levels = 0, pending_smooth = 0;
  int cache_next = 0;
  std::weak_ptr<MatBuf> pending_left;            // the left image ImagePyramid was just built from: its lookahead is still to be issued
  std::weak_ptr<MatBuf> pending_partner;         // ... and the guessed right image (mirror block reserved)
  std::weak_ptr<MatBuf> pending_next;            // ... and the guessed NEXT left image (the same-sized Mat created right after the partner)
  std::weak_ptr<MatBuf> pending_next_r;          // ... and ITS partner (the one after that): the next frame's ComputeDepth can then start at once
  const MatBuf* next_guess_was = nullptr;        // what the previous frame's guess named (compared, never dereferenced)
  int pending_rows = 0, pending_cols = 0;
  unsigned long pending_mark = 0;                // the main stream's fill level then (odo_ctx_mark): the side stream goes behind THAT
#ifdef ODOMETRY_SHIM_WITH_OPENCV
  // cv::Mat outputs are host memory. The three images of a ComputeDepth started ahead are zero but at the selected points: the points
  // {pixel, val, disp, dep} are gathered and copied into ONE page-locked block behind the job, still beside the Solve (532 KB instead
  // of three dense images, 4.2 MB: the copy no longer outlasts the Solve), and ComputeDepth (:229) rebuilds the images in its output
  // Mats from it (odo_host_scatter_outputs). ODOMETRY_SHIM_LAZY_OUTPUTS: only left_val, as a dense copy.
  void* out_stage[3] = {nullptr, nullptr, nullptr};   // [0]: the compact block, or left_val's dense copy (lazy outputs)
  size_t out_stage_bytes[3] = {0, 0, 0};
  unsigned long out_mark = 0;                    // the side stream's position behind the copy; 0: nothing staged
  bool out_compact = false;
  // ... and they are rebuilt AHEAD as well: the thread inside Solve spins for ~0.25 ms (the reference's Solve keeps the CPU busy all
  // that time, ref: src/lm_optimizer.cpp:73-160); from that wait loop (odo_lm_set_idle_callback, a few us per call) the three images are
  // zero-filled, and — once the compact block has arrived — the points written, in Mats of the shim's own. ComputeDepth (:229) hands
  // them over by header assignment where that cannot be observed: a caller's output Mat that is empty, or whose buffer nobody else
  // refers to (cv::Mat::u->refcount == 1 — the runner's, ref: run_odometry_kitti_offline.cpp:226-228), ends up with a buffer of the
  // right size, type and content either way; any other output Mat (a header some other Mat shares, user memory, a view) is written
  // in place as before. ONE RULE would follow, the one every OpenCV function with an output Mat has: a raw pointer taken from an output
  // Mat before ComputeDepth is not that Mat's data pointer afterwards. The reference writes its outputs in place
  // (ref: src/depth_estimate.cpp:176-191,388-397), so a caller that is valid against it may hold such a pointer: the hand-over is
  // therefore OPT-IN (ODOMETRY_SHIM_SWAP_OUTPUTS=1); by default every output is written into the caller's own buffer.
  struct Prepared {
    Mat img[3];                 // val (CV_8U), disp, dep (PixelType)
    int phase = 0;              // 0 none; 1 zero fill under way; 2 zero-filled, the compact block still to come; 3 complete; -1 failed
    size_t zeroed = 0;          // phase 1: bytes of the three images (one after the other) done so far
    unsigned long long dep_fp = 0;
    int rows = 0, cols = 0;
    unsigned long mark = 0;     // the out_mark this belongs to
  } prep;
  std::vector<Mat> pool;        // output Mats handed out before: reused once the caller has let go of them (refcount back to 1)
  ~Lookahead() { for (void* p : out_stage) if (p) odo_host_free(p); }
#endif
};
inline Lookahead& lookahead() { static thread_local Lookahead l; return l; }
#ifdef ODOMETRY_SHIM_WITH_OPENCV
inline bool keep_output_buffers() {   // (ODOMETRY_SHIM_KEEP_OUTPUT_BUFFERS: the round-5 opt-out, still honoured)
  static const bool on = std::getenv("ODOMETRY_SHIM_SWAP_OUTPUTS") == nullptr || std::getenv("ODOMETRY_SHIM_KEEP_OUTPUT_BUFFERS") != nullptr;
  return on;
}
inline void prep_drop(Lookahead& la) {
  for (Mat& m : la.prep.img) m.release();
  la.prep.phase = 0; la.prep.mark = 0;
}
inline Mat prep_take(Lookahead& la, int rows, int cols, int type) {
  for (Mat& m : la.pool)
    if (m.rows == rows && m.cols == cols && m.type() == type && m.u && m.u->refcount == 1) {
      if (auto r = mat_table().find(m, false)) r->content_changed();   // (what the table knew about these bytes is void)
      return m;
    }
  Mat m(rows, cols, type);
  if (la.pool.size() >= 12) la.pool.erase(la.pool.begin());
  la.pool.push_back(m);
  return m;
}
// Called where the compact copy has just been queued (run_lookahead).
inline void prep_begin(Lookahead& la, int rows, int cols) {
  prep_drop(la);
  if (keep_output_buffers() || !la.out_mark || !la.out_compact) return;
  la.prep.img[0] = prep_take(la, rows, cols, CV_8U);
  la.prep.img[1] = prep_take(la, rows, cols, PixelType);
  la.prep.img[2] = prep_take(la, rows, cols, PixelType);
  la.prep.rows = rows; la.prep.cols = cols; la.prep.zeroed = 0; la.prep.mark = la.out_mark; la.prep.phase = 1;
}
// One short piece of the work (<= 256 KB of zero fill, or the scatter once the block is there). block: wait for the block.
inline void prep_step(Lookahead& la, bool block) {
  Lookahead::Prepared& q = la.prep;
  if (q.phase == 1) {
    size_t off = q.zeroed, left = 256u << 10;
    for (int i = 0; i < 3 && left; i++) {
      const size_t n = q.img[i].total() * q.img[i].elemSize();
      if (off >= n) { off -= n; continue; }
      const size_t take = (n - off < left) ? n - off : left;
      std::memset(q.img[i].data + off, 0, take);
      q.zeroed += take; left -= take; off = 0;
    }
    if (left) q.phase = 2;   // (ran out of image before running out of budget)
    return;
  }
  if (q.phase == 2) {
    if (block) { if (odo_ctx_wait_mark(side_context(), q.mark) != 0) { q.phase = -1; return; } }
    else {
      const int r = odo_ctx_mark_reached(side_context(), q.mark);
      if (r == 0) return;
      if (r < 0) { q.phase = -1; return; }
    }
    q.phase = odo_host_scatter_outputs_prezeroed(la.out_stage[0], q.rows, q.cols, q.img[0].data, (size_t)q.img[0].step, q.img[1].ptr<float>(),
                                                 (size_t)q.img[1].step, q.img[2].ptr<float>(), (size_t)q.img[2].step, &q.dep_fp) == 0 ? 3 : -1;
  }
}
inline void solve_idle(void*) { Lookahead& la = lookahead(); if (la.prep.phase == 1 || la.prep.phase == 2) prep_step(la, false); }
inline bool prep_finish(Lookahead& la) {
  while (la.prep.phase == 1 || la.prep.phase == 2) prep_step(la, true);
  return la.prep.phase == 3;
}
// May this output Mat get a new buffer without anyone being able to tell?
inline bool prep_swappable(const Mat& m, int rows, int cols, int type) {
  if (m.empty()) return true;
  return m.u != nullptr && m.u->refcount == 1 && m.rows == rows && m.cols == cols && m.type() == type && m.isContinuous();
}
// ComputeDepth, before its outputs' records are touched: the prepared Mats take the place of the caller's (or nothing happens).
inline bool prep_swap_in(Lookahead& la, Mat& val, Mat& disp, Mat& dep) {
  Lookahead::Prepared& q = la.prep;
  if (q.phase == 0 || q.phase == -1 || q.mark == 0 || q.mark != la.out_mark) return false;
  if (!prep_swappable(val, q.rows, q.cols, CV_8U) || !prep_swappable(disp, q.rows, q.cols, PixelType) ||
      !prep_swappable(dep, q.rows, q.cols, PixelType))
    return false;
  val = q.img[0]; disp = q.img[1]; dep = q.img[2];   // (q.img keeps its headers until prep_drop: prep_finish may still have to write)
  return true;
}
#endif
// The three output blocks of a ComputeDepth started ahead go back to the free list: behind the job (side stream) when it was started.
inline void early_release() {
  Lookahead::Early& e = lookahead().early;
  if (!e.reserved) return;
  if (e.started) odo_ctx_stream_wait(context(), side_context());
  for (int i = 0; i < 3; i++)
    if (e.blk[i]) { odo_dev_free_async(context(), e.blk[i], e.bytes[i], e.async_[i]); e.blk[i] = nullptr; }
  e.reserved = e.started = false;
}
inline void early_reserve(int rows, int cols) {
  early_release();
  Lookahead::Early& e = lookahead().early;
  const size_t n = (size_t)rows * cols;
  e.bytes[0] = n; e.bytes[1] = e.bytes[2] = n * sizeof(float);
  for (int i = 0; i < 3; i++)
    if (odo_dev_alloc_async(context(), e.bytes[i], &e.blk[i], &e.async_[i]) != 0) {
      e.blk[i] = nullptr;
      for (int k = 0; k < i; k++) { odo_dev_free_async(context(), e.blk[k], e.bytes[k], e.async_[k]); e.blk[k] = nullptr; }
      return;
    }
  e.reserved = true; e.started = false; e.rows = rows; e.cols = cols;
}
// Issues the recorded lookahead. Called by Solve right AFTER its own launches have gone out (the Solve is the critical path: ~15 us
// of host work must not sit in front of it), else by the next ComputeDepth.
inline void run_lookahead(odo_lm* lm = nullptr, const odo_pyr* cur_img = nullptr) {
  Lookahead& la = lookahead();
  std::shared_ptr<MatBuf> lb = la.pending_left.lock(), guess = la.pending_partner.lock(), nxt = la.pending_next.lock(),
                          nxt_r = la.pending_next_r.lock();
  la.pending_left.reset(); la.pending_partner.reset(); la.pending_next.reset(); la.pending_next_r.reset();
  if (!la.on || !lb || !la.pending_mark) return;
  if (odo_ctx_stream_wait_mark(side_context(), context(), la.pending_mark) != 0) return;
  // the stereo partner's upload, on the side stream
  if (guess && guess.get() != lb.get()) prefetch_to_device(guess);
  // ComputeDepth on the side stream (frames of the size the estimator has seen), behind the left image's upload — the mark — not
  // behind the Solve queued since: all of it when the partner is on the device (or on its way there on the side stream), else
  // its left half
  if (la.estimator && la.pending_rows == la.est_rows && la.pending_cols == la.est_cols && lb->dev && lb->dev_valid &&
      !lb->side_pending) {
    Lookahead::Early& e = la.early;
    if (e.reserved && !e.started && e.rows == la.pending_rows && e.cols == la.pending_cols && guess && guess.get() != lb.get() &&
        guess->dev && guess->dev_valid && guess->bytes == lb->bytes &&
        odo_depth_compute_begin_dev(la.estimator, side_context(), static_cast<const float*>(lb->dev),
                                    static_cast<const float*>(guess->dev), e.rows, e.cols, static_cast<uint8_t*>(e.blk[0]),
                                    static_cast<float*>(e.blk[1]), static_cast<float*>(e.blk[2]), lb->stamp, guess->stamp,
                                    la.pending_mark) == 0) {
      e.started = true; e.est = la.estimator;
      e.left_dev = lb->dev; e.right_dev = guess->dev; e.left_stamp = lb->stamp; e.right_stamp = guess->stamp;
#ifdef ODOMETRY_SHIM_WITH_OPENCV
      // (first behind the job: the sooner the compact block is on the host, the more of the output images is built while Solve waits;
      //  behind the pyramid and the lists instead: 1 525-1 538 against 1 565-1 590 frames/s)
      la.out_mark = 0;
      la.out_compact = !lazy_outputs();
      const size_t need = la.out_compact ? odo_depth_compact_bytes() : e.bytes[0];
      if (la.out_stage_bytes[0] != need) {
        if (la.out_stage[0]) odo_host_free(la.out_stage[0]);
        la.out_stage[0] = odo_host_alloc(need);
        la.out_stage_bytes[0] = la.out_stage[0] ? need : 0;
      }
      const bool staged = la.out_stage[0] &&
          (la.out_compact ? odo_depth_compact_outputs_async(la.estimator, side_context(), static_cast<const uint8_t*>(e.blk[0]),
                                                            static_cast<const float*>(e.blk[1]), static_cast<const float*>(e.blk[2]), e.cols,
                                                            la.out_stage[0])
                          : odo_dev_download_async(side_context(), la.out_stage[0], e.blk[0], e.bytes[0])) == 0;
      if (staged) la.out_mark = odo_ctx_mark(side_context());
      prep_begin(la, e.rows, e.cols);
#endif
      // ... and behind it, still beside the Solve: the frame's depth pyramid (:252, with what the last DepthPyramid was built with)
      // and the keyframe-candidate point lists of (this image pyramid, that depth pyramid) — if the runner promotes this frame
      // (:258-260), the next Solve finds its lists built (odo_lm_candidate_begin) instead of building them in front of its launches
      la.early_dep_pyr.reset();
      if (la.dp_levels > 0) {
        auto hp = std::make_shared<PyrHandle>();
        hp->host.resize(la.dp_levels);
        hp->have.assign(la.dp_levels, 0);
        if (odo_pyramid_create_dev(side_context(), static_cast<const float*>(e.blk[2]), e.rows, e.cols, la.dp_levels, la.dp_smooth,
                                   ODO_PYR_DEPTH, &hp->p) == 0) {
          la.early_dep_pyr = hp;
          la.early_dep_mark = odo_ctx_mark(side_context());
          if (lm && cur_img) (void)odo_lm_candidate_begin(lm, side_context(), cur_img, hp->p, la.pending_mark);
        }
      }
    } else {
      (void)odo_depth_prepare_left_dev_marked(la.estimator, side_context(), static_cast<const float*>(lb->dev), la.pending_rows,
                                              la.pending_cols, lb->stamp, la.pending_mark);
    }
  }
  // the NEXT frame's left image follows on the side stream, behind this frame's ComputeDepth: it crosses PCIe while the Solve still
  // runs instead of in front of the next ImagePyramid — and its pyramid is built there too (the next ImagePyramid of the same
  // content, levels and smoothing finds it in the cache)
  if (nxt && nxt.get() != lb.get() && nxt.get() != guess.get()) {
    prefetch_to_device(nxt);
    static const bool ahead_pyr_on = std::getenv("ODOMETRY_SHIM_NO_AHEAD_PYRAMID") == nullptr;
    if (ahead_pyr_on && nxt->dev && nxt->dev_valid && nxt->side_pending && la.pending_levels > 0) {
      auto h = std::make_shared<PyrHandle>();
      h->host.resize(la.pending_levels);
      h->have.assign(la.pending_levels, 0);
      if (odo_pyramid_create_dev(side_context(), static_cast<const float*>(nxt->dev), la.pending_rows, la.pending_cols, la.pending_levels,
                                 la.pending_smooth, ODO_PYR_IMAGE, &h->p) == 0) {
        Lookahead::Entry& ce = la.cache[la.cache_next++ % 8];
        ce.stamp = nxt->stamp; ce.levels = la.pending_levels; ce.smooth = la.pending_smooth; ce.kind = ODO_PYR_IMAGE; ce.h = h;
        ce.on_side = true;
        ce.side_mark = odo_ctx_mark(side_context());   // (the consumer waits for THIS point of the side stream, not for what follows)
        la.ahead_pyr = h;   // (kept alive until the next ImagePyramid has had its chance)
      }
    }
    // ... and the next frame's RIGHT image behind that: the next frame's ComputeDepth then starts the moment its Solve has been
    // queued instead of behind a 39-us upload, and this stream's chain ends inside the Solve
    if (nxt_r && nxt_r.get() != lb.get() && nxt_r.get() != guess.get() && nxt_r.get() != nxt.get()) prefetch_to_device(nxt_r);
  }
}
// What ImagePyramid's constructor records for the Solve that follows (run_lookahead issues it): once per image content.
inline void record_lookahead(const Mat& in, int num_levels, bool smooth) {
  Lookahead& la = lookahead();
  const std::shared_ptr<MatBuf> lb = buffer_of(in);
  if (!lb || la.recorded_stamp == lb->stamp) return;   // (:251 builds the pyramid of the :205 image again)
  la.recorded_stamp = lb->stamp;
  la.pending_left = lb; la.pending_rows = in.rows; la.pending_cols = in.cols;
  la.pending_levels = num_levels; la.pending_smooth = smooth ? 1 : 0;
  // the partner guess: the Mat ComputeDepth was given with this one last time, else the same-sized Mat created right after it
  std::shared_ptr<MatBuf> guess;
  if (la.last_left.lock().get() == lb.get()) guess = la.last_right.lock();
  else if (auto nx = lb->successor()) { if (nx->bytes == lb->bytes) guess = nx; }
  if (guess && guess.get() != lb.get()) prefetch_reserve(guess);
  la.pending_partner = guess;
  // the next frame's left image: a sequence read into Mats pair by pair (ref: :334-359) has it right behind the partner
  std::shared_ptr<MatBuf> nxt;
  if (guess && guess.get() != lb.get())
    if (auto nx = guess->successor()) { if (nx.get() != lb.get() && nx->bytes == lb->bytes) nxt = nx; }
  // ... but only once the guess has proved right: this left image IS what the previous frame's guess named (a runner that
  // refills two Mats per frame, ref: :200, never gets there — the same-sized Mat behind its partner is something else)
  const bool proven = la.next_guess_was == lb.get();
  la.next_guess_was = nxt.get();
  if (nxt && proven) prefetch_reserve(nxt); else nxt.reset();
  la.pending_next = nxt;
  std::shared_ptr<MatBuf> nxt_r;
  if (nxt)
    if (auto nx = nxt->successor()) { if (nx.get() != lb.get() && nx.get() != guess.get() && nx->bytes == lb->bytes) nxt_r = nx; }
  if (nxt_r) prefetch_reserve(nxt_r);
  la.pending_next_r = nxt_r;
  // ... and the three output blocks of a ComputeDepth started ahead (recycled blocks: their earlier use is in front of the mark)
  if (la.estimator && in.rows == la.est_rows && in.cols == la.est_cols && guess && guess.get() != lb.get()) early_reserve(in.rows, in.cols);
  else early_release();
  la.pending_mark = odo_ctx_mark(context());
  // (Round 5, cv::Mat build, measured and dropped: the partner's staging copy + upload from HERE instead of from Solve — the side
  //  stream's chain then ends ~45 us sooner, inside the Solve, but the copy sits in front of the Solve's launches: 1 686-1 710 against
  //  1 672-1 703 frames/s.)
}
inline std::shared_ptr<PyrHandle> make_pyr(int num_levels, const Mat& in, bool smooth, int kind, const char* what) {
  CallScope call;
  Lookahead& la = lookahead();
  const unsigned long long stamp = (in.type() == PixelType && !in.empty()) ? content_stamp(in) : 0;
  if (la.on && stamp)
    for (auto& e : la.cache)
      if (e.stamp == stamp && e.levels == num_levels && e.smooth == (smooth ? 1 : 0) && e.kind == kind)
        if (auto hit = e.h.lock()) {
          // :251 after :205: the same image, the same arithmetic — the same device pyramid; or the pyramid built ahead on the side
          // stream during the previous frame's Solve (the main stream goes behind it: the image's upload and the pyramid kernel)
          if (e.on_side) {
            odo_ctx_stream_wait_mark(context(), side_context(), e.side_mark);   // (an overwritten mark: behind everything queued there)
            e.on_side = false;
            if (auto ib = buffer_of(in)) ib->side_pending = false;
          }
          if (la.ahead_pyr.get() == hit.get()) la.ahead_pyr.reset();
          if (la.early_dep_pyr.get() == hit.get()) la.early_dep_pyr.reset();
          if (kind == ODO_PYR_IMAGE) record_lookahead(in, num_levels, smooth);
          else { la.dp_levels = num_levels; la.dp_smooth = smooth ? 1 : 0; }
          return hit;
        }
  if (kind == ODO_PYR_IMAGE) la.ahead_pyr.reset();   // (a pyramid built ahead for an image that did not come)
  auto h = std::make_shared<PyrHandle>();
  h->host.resize(num_levels > 0 ? num_levels : 0);
  h->have.assign(num_levels > 0 ? num_levels : 0, 0);
  bool ok = in.type() == PixelType && !in.empty();   // (a cv::Mat view is fine: DevIn uploads it row by row, like cv::GaussianBlur reads it)
  if (ok) {
    DevIn src(in);  // the frame on the device (already there when the same Mat was used a moment ago)
    ok = src.get() != nullptr &&
         odo_pyramid_create_dev(context(), static_cast<const float*>(src.get()), in.rows, in.cols, num_levels, smooth ? 1 : 0, kind,
                                &h->p) == 0;
  }
  if (!ok) {
    std::cout << what << std::endl;  // ref: src/image_pyramid.cpp:16-18,34-36 (prints, object stays unusable)
    h->p = nullptr;
  }
  if (ok && la.on) {
    Lookahead::Entry& e = la.cache[la.cache_next++ % 8];
    e.stamp = content_stamp(in); e.levels = num_levels; e.smooth = smooth ? 1 : 0; e.kind = kind; e.h = h; e.on_side = false; e.side_mark = 0;
    if (kind == ODO_PYR_IMAGE) record_lookahead(in, num_levels, smooth);
    else { la.dp_levels = num_levels; la.dp_smooth = smooth ? 1 : 0; }
  }
  return h;
}
}  // namespace detail

// ------------------------------------------------------------------------------------------------
class ImagePyramid {  // ref: include/image_pyramid.h:14-41
 public:
  ImagePyramid() = delete;
  ImagePyramid(int num_levels, const Mat& in_img, bool smooth = true)
      : num_levels_(num_levels),
        h_(detail::make_pyr(num_levels, in_img, smooth, ODO_PYR_IMAGE, "Compute Gaussian Image Pyramid failed!")) {}
  ImagePyramid& operator=(const ImagePyramid&) = delete;
  int GetNumberLevels() const { return num_levels_; }
  const Mat& GetPyramidImage(int level_idx) const {
    if (level_idx >= num_levels_) {  // ref: src/image_pyramid.cpp:22-25
      std::cout << "Requested image pyramid does not exist! Max pyramid id: " << num_levels_ - 1 << std::endl;
      std::exit(1);
    }
    return detail::pyr_level(h_, level_idx);
  }
  const odo_pyr* handle() const { return h_->p; }
 private:
  int num_levels_;
  std::shared_ptr<detail::PyrHandle> h_;  // copies share the device pyramid, like cv::Mat headers in the reference
};

class DepthPyramid {  // ref: include/image_pyramid.h:43-69
 public:
  DepthPyramid() = delete;
  DepthPyramid(int num_levels, const Mat& in_depth, bool smooth = true)
      : num_levels_(num_levels),
        h_(detail::make_pyr(num_levels, in_depth, smooth, ODO_PYR_DEPTH, "Compute Gaussian Depth Pyramid failed!")) {}
  DepthPyramid& operator=(const DepthPyramid&) = delete;
  int GetNumberLevels() const { return num_levels_; }
  const Mat& GetPyramidDepth(int level_idx) const { return detail::pyr_level(h_, level_idx); }
  const odo_pyr* handle() const { return h_->p; }
 private:
  int num_levels_;
  std::shared_ptr<detail::PyrHandle> h_;
};

// ------------------------------------------------------------------------------------------------
class LevenbergMarquardtOptimizer {  // ref: include/lm_optimizer.h:24-115
 public:
  LevenbergMarquardtOptimizer() = delete;
  LevenbergMarquardtOptimizer(float lambda, float precision, const std::vector<int> kMaxIterations,
                              const Affine4f& kRelativeInit, const std::shared_ptr<CameraPyramid>& kCameraPtr,
                              const int robust_est, const float huber_delta = 4.0f / 255.0f) {
    if (kCameraPtr == nullptr) std::cout << "LM Optimizer failed! Invalid camera pointer!" << std::endl;  // ref: :35-36
    // The reference stores the camera but evaluates with hard-coded KITTI-00 constants (include/image_processing_global.h:
    // 33-36,48-51: the camera_ptr lines are commented out, "only for debug now"), so parity mode passes K = NULL.
    // -DODOMETRY_SHIM_USE_CAMERA_INTRINSICS switches to the camera's rectified level-0 intrinsics instead.
    const odo_intrinsics* Kp = nullptr;
#ifdef ODOMETRY_SHIM_USE_CAMERA_INTRINSICS
    odo_intrinsics K;
    if (kCameraPtr != nullptr) {
      K.f0 = kCameraPtr->fx_float(0); K.cx0 = kCameraPtr->cx_float(0); K.cy0 = kCameraPtr->cy_float(0);
      Kp = &K;
    }
#endif
    if (odo_lm_create(detail::context(), lambda, precision, kMaxIterations.data(), (int)kMaxIterations.size(),
                      affine_data(kRelativeInit), robust_est, huber_delta, Kp, &lm_) != 0)
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
#ifndef ODOMETRY_SHIM_REAL_REPORT
    // ShowReport prints the reference's never-written statistics (zeros, see there) and nothing else of this class reads the
    // per-evaluation trace: the Solves run the lean LM kernels (odo_lm_set_record)
    else (void)odo_lm_set_record(lm_, 0);
#endif
#ifdef ODOMETRY_SHIM_WITH_OPENCV
    if (lm_) (void)odo_lm_set_idle_callback(lm_, &detail::solve_idle, nullptr);   // ComputeDepth's output images are built while Solve waits
#endif
  }
  ~LevenbergMarquardtOptimizer() { odo_lm_destroy(lm_); }
  LevenbergMarquardtOptimizer(const LevenbergMarquardtOptimizer&) = delete;
  LevenbergMarquardtOptimizer& operator=(const LevenbergMarquardtOptimizer&) = delete;

  Affine4f Solve(const ImagePyramid& kImagePyr1, const DepthPyramid& kDepthPyr1, const ImagePyramid& kImagePyr2) {
    Affine4f out;
    detail::CallScope call;
    // the Solve's launches first (odo_lm_solve_begin returns once they are queued), then what can run beside them
    if (kImagePyr1.handle() && kDepthPyr1.handle() && kImagePyr2.handle())
      (void)odo_lm_solve_begin(lm_, kImagePyr1.handle(), kDepthPyr1.handle(), kImagePyr2.handle());
    detail::run_lookahead(lm_, kImagePyr2.handle());
    if (odo_lm_solve(lm_, kImagePyr1.handle(), kDepthPyr1.handle(), kImagePyr2.handle(), affine_data(out)) != 0)
      std::cout << "Optimize failed! " << std::endl;  // ref: src/lm_optimizer.cpp:60-65 (out = pseudo-identity)
    return out;
  }
  // ref: src/lm_optimizer.cpp:364-371. The reference never writes iters_stat_ / cost_stat_ (no writer anywhere in
  // lm_optimizer.cpp; ResetStatistics zeroes them), so its report is all zeros: that is what this prints. Compile with
  // -DODOMETRY_SHIM_REAL_REPORT to print what the device actually counted (evaluations and first / last mean weighted
  // error per level) — odo_lm_report() returns those either way.
  void ShowReport() {
    int it[4] = {0, 0, 0, 0};
    float cost[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#ifdef ODOMETRY_SHIM_REAL_REPORT
    odo_lm_report(lm_, it, cost);
#endif
    std::cout << "Number of iterations performed per level: ";
    std::cout << it[0] << ", " << it[1] << ", " << it[2] << ", " << it[3] << std::endl;
    std::cout << "Costs before/after per level: " << std::endl;
    for (int i = 0; i < 4; i++) std::cout << cost[i][0] << ", " << cost[i][1] << std::endl;
  }
  OptimizerStatus Reset(const Affine4f& kRelativeInit, const float lambda) {  // ref: :373-382
    if (odo_lm_reset(lm_, affine_data(kRelativeInit), lambda) != 0) {
      std::cout << "Reset optimizer failed!" << std::endl;
      return -1;
    }
    return 0;
  }
  odo_lm* handle() { return lm_; }
 private:
  odo_lm* lm_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
class DepthEstimator {  // ref: include/depth_estimate.h:24-121
 public:
  DepthEstimator() = delete;
  DepthEstimator(float grad_th, float ssd_th, float photo_th, float min_depth, float max_depth, float lambda,
                 float huber_delta, float precision, int max_iters, int boundary,
                 const std::shared_ptr<CameraPyramid>& left_cam_ptr, const std::shared_ptr<CameraPyramid>& /*right_cam_ptr*/,
                 float baseline, int max_residuals = 5000)
      : max_iters_(max_iters) {
    const odo_intrinsics* Kp = nullptr;  // parity mode: the reference's hard-coded focal length (src/depth_estimate.cpp:213-214,272-273)
#ifdef ODOMETRY_SHIM_USE_CAMERA_INTRINSICS
    odo_intrinsics K;
    if (left_cam_ptr != nullptr) {
      K.f0 = left_cam_ptr->fx_float(0); K.cx0 = left_cam_ptr->cx_float(0); K.cy0 = left_cam_ptr->cy_float(0);
      Kp = &K;
    }
#else
    (void)left_cam_ptr;
#endif
    if (odo_depth_create(detail::context(), grad_th, ssd_th, photo_th, min_depth, max_depth, lambda, huber_delta, precision,
                         max_iters, boundary, Kp, baseline, max_residuals, 0, 0, &d_) != 0)
      std::cout << "odometry_hip: " << odo_last_error() << std::endl;
    else detail::lookahead().estimator = d_;
  }
  ~DepthEstimator() {
    if (detail::lookahead().estimator == d_) { detail::early_release(); detail::lookahead().estimator = nullptr; }
    odo_depth_destroy(d_);
  }
  DepthEstimator(const DepthEstimator&) = delete;
  DepthEstimator& operator=(const DepthEstimator&) = delete;

  GlobalStatus ComputeDepth(const Mat& left_img, const Mat& right_img, Mat& left_val, Mat& left_disp, Mat& left_dep) {
    if (left_img.rows != right_img.rows || left_img.cols != right_img.cols) {  // ref: src/depth_estimate.cpp:37-40
      std::cout << "Number of rows/cols do not match for left/right images." << std::endl;
      return -1;
    }
    if (left_img.type() != PixelType || right_img.type() != PixelType) {       // ref: :41-44
      std::cout << "Pixel type of left/right images not 32-bit float." << std::endl;
      return -1;
    }
    if (!left_img.isContinuous() || !right_img.isContinuous() || !left_disp.isContinuous() || !left_dep.isContinuous() ||
        !left_val.isContinuous()) {                                              // ref: :259-263
      std::cout << "The cv::Mat matrix is not continuous in disparity search!" << std::endl;
      return -1;
    }
    std::cout << "computing disparity ..." << std::endl;
    int st = -1;
    bool collected = false;
    detail::CallScope call;
    {
      detail::Lookahead& la = detail::lookahead();
      la.pending_left.reset();   // (a lookahead nobody issued: too late for it now)
      // the job started ahead from ImagePyramid's constructor / Solve, if it was started for exactly these two images (a cv::Mat's
      // pixels are fingerprinted here: a caller who rewrote one of them since the upload gets a fresh job)
      detail::Lookahead::Early& e = la.early;
      const std::shared_ptr<detail::MatBuf> lsp = detail::held_buffer_of(left_img), rsp = detail::held_buffer_of(right_img);
      const detail::MatBuf *lb = lsp.get(), *rb = rsp.get();
#ifdef ODOMETRY_SHIM_PHASES
      static double ph_[6] = {0, 0, 0, 0, 0, 0}; static long phn_ = 0;
      auto t_ = std::chrono::steady_clock::now();
      auto lap_ = [&](int i) { auto n_ = std::chrono::steady_clock::now(); ph_[i] += std::chrono::duration<double, std::micro>(n_ - t_).count(); t_ = n_; };
#else
      auto lap_ = [](int) {};
#endif
#ifdef ODOMETRY_SHIM_WITH_OPENCV
      if (e.started && lsp && rsp) { lsp->validate(); rsp->validate(); }
#endif
      lap_(0);
      if (e.started && e.est == d_ && lb && rb && lb->dev == e.left_dev && lb->dev_valid && lb->stamp == e.left_stamp &&
          rb->dev == e.right_dev && rb->dev_valid && rb->stamp == e.right_stamp && left_img.rows == e.rows && left_img.cols == e.cols &&
          detail::device_bytes(left_val) == e.bytes[0] && detail::device_bytes(left_disp) == e.bytes[1] &&
          detail::device_bytes(left_dep) == e.bytes[2]) {
#ifdef ODOMETRY_SHIM_WITH_OPENCV
        const bool swapped = detail::prep_swap_in(la, left_val, left_disp, left_dep);   // (see Lookahead::Prepared)
#endif
        detail::adopt_device(left_val, e.blk[0], e.async_[0]);
        detail::adopt_device(left_disp, e.blk[1], e.async_[1], detail::lazy_outputs());
        detail::adopt_device(left_dep, e.blk[2], e.async_[2], detail::lazy_outputs());
        int persist_on = 0, redone_before = 0, redone_after = 0;
        odo_depth_persistent_stats(d_, &persist_on, &redone_before);
        st = odo_depth_compute_end_dev(d_, static_cast<const float*>(e.left_dev), static_cast<const float*>(e.right_dev), e.rows, e.cols,
                                       static_cast<uint8_t*>(e.blk[0]), static_cast<float*>(e.blk[1]), static_cast<float*>(e.blk[2]),
                                       e.left_stamp, e.right_stamp);
        lap_(1);
        e.blk[0] = e.blk[1] = e.blk[2] = nullptr;   // (the Mats own them now)
        e.reserved = e.started = false;
        collected = true;
        detail::stats().early_adopted++;
        odo_depth_persistent_stats(d_, &persist_on, &redone_after);
#ifdef ODOMETRY_SHIM_WITH_OPENCV
        {  // the outputs reach the caller's host memory: out of the staging blocks filled beside the Solve, or (the job had to be run
           // again, nothing was staged) straight from the device
          Mat* outs[3] = {&left_val, &left_disp, &left_dep};
          const bool staged = la.out_mark != 0 && redone_after == redone_before && st == 0;
          const bool prepared = staged && swapped && detail::prep_finish(la);   // (built while Solve waited: usually nothing is left to do)
          if (staged && !prepared) odo_ctx_wait_mark(detail::side_context(), la.out_mark);
          lap_(2);
          if (staged && la.out_compact && !detail::lazy_outputs()) {
            unsigned long long dep_fp = la.prep.dep_fp;
            if (prepared) detail::stats().outputs_prepared++;
            if (prepared ||
                odo_host_scatter_outputs(la.out_stage[0], e.rows, e.cols, left_val.data, (size_t)left_val.step, left_disp.ptr<float>(),
                                         (size_t)left_disp.step, left_dep.ptr<float>(), (size_t)left_dep.step, &dep_fp) == 0) {
              for (int i = 0; i < 3; i++) {
                auto ob = detail::output_buffer_of(*outs[i]);
                ob->fp = dep_fp;
                ob->finish_delivery();
                if (i < 2) ob->fp_known = false;   // (left_val / left_disp never come back as inputs: no fingerprint pass spent on them —
              }                                    //  if one ever does, it is simply uploaded again)
            } else {
              for (int i = 0; i < 3; i++) detail::output_buffer_of(*outs[i])->deliver_now();
            }
          } else {
            for (int i = 0; i < 3; i++) {
              if (i > 0 && detail::lazy_outputs()) break;
              auto ob = detail::output_buffer_of(*outs[i]);
              if (staged && !la.out_compact && i == 0) ob->deliver_from(la.out_stage[0]); else ob->deliver_now();
            }
          }
          la.out_mark = 0;
          detail::prep_drop(la);
          lap_(3);
        }
#endif
#ifdef ODOMETRY_SHIM_PHASES
        if (++phn_ % 199 == 0) std::fprintf(stderr, "[ComputeDepth phases] validate L,R %.1f  adopt + wait for the job %.1f  wait for the staged copies %.1f  copy out %.1f us\n", ph_[0] / phn_, ph_[1] / phn_, ph_[2] / phn_, ph_[3] / phn_);
#endif
        // the :252 pyramid was built from this block beside the Solve: DepthPyramid finds it — unless the job had to be run again
        // (its persistent launch gave up: the block was rewritten after the pyramid had been built from it)
        if (la.early_dep_pyr && st == 0 && redone_after == redone_before) {
          detail::Lookahead::Entry& ce = la.cache[la.cache_next++ % 8];
          ce.stamp = detail::output_stamp(left_dep); ce.levels = la.dp_levels; ce.smooth = la.dp_smooth; ce.kind = ODO_PYR_DEPTH;
          ce.h = la.early_dep_pyr; ce.on_side = true; ce.side_mark = la.early_dep_mark;
        } else {
          la.early_dep_pyr.reset();
        }
        la.last_left = lsp; la.last_right = rsp;
      } else {
        if (e.started) detail::stats().early_dropped++;
        detail::early_release();   // (the library drops the job itself at its next call)
        la.early_dep_pyr.reset();
#ifdef ODOMETRY_SHIM_WITH_OPENCV
        detail::prep_drop(la);
#endif
      }
    }
    if (!collected) {
      detail::DevIn l(left_img), r(right_img);
      detail::DevOut v(left_val), ds(left_disp, detail::lazy_outputs()), dp(left_dep, detail::lazy_outputs());
      if (l.get() && r.get() && v.get() && ds.get() && dp.get()) {
        // (the left image's content stamp lets the estimator pick up the half prepared from ImagePyramid's constructor)
        st = odo_depth_compute_dev_stamped(d_, static_cast<const float*>(l.get()), static_cast<const float*>(r.get()), left_img.rows,
                                           left_img.cols, static_cast<uint8_t*>(v.get()), static_cast<float*>(ds.get()),
                                           static_cast<float*>(dp.get()), detail::content_stamp(left_img));
        detail::Lookahead& la = detail::lookahead();
        la.last_left = detail::buffer_of(left_img); la.last_right = detail::buffer_of(right_img);
        la.est_rows = left_img.rows; la.est_cols = left_img.cols;
      }
    }
    int iters = 0, nsel = 0, nmatch = 0, nvalid = 0;
    float cost = 0;
    odo_depth_report(d_, &iters, &cost, &nsel, &nmatch, &nvalid);
    if (st != 0) {
      std::cout << odo_last_error() << std::endl;
      std::cout << "Depth optimization failed!" << std::endl;                    // ref: :70-72
      return -1;
    }
    std::cout << "valid disparities: " << nsel << std::endl;                     // ref: :62
    std::cout << "optimizing depth ..." << std::endl;
    std::cout << "valid depth: " << nvalid << std::endl;                         // ref: :74
    return 0;
  }
  void ReportStatus() {  // ref: src/depth_estimate.cpp:465-468
    int iters = 0;
    float cost = 0;
    odo_depth_report(d_, &iters, &cost, nullptr, nullptr, nullptr);
    std::cout << "    Number of iters performed: " << iters << "(max allowed: " << max_iters_ << ")" << std::endl;
    std::cout << "    Final cost: " << cost << std::endl;
  }
 private:
  odo_depth* d_ = nullptr;
  int max_iters_;
};

// ------------------------------------------------------------------------------------------------
class KeyFrame {  // ref: include/keyframe.h:17-60, src/keyframe.cpp
 public:
  KeyFrame() = delete;
  KeyFrame(const std::shared_ptr<Mat>& kLeftImg, const std::shared_ptr<Mat>& kRightImg, const std::shared_ptr<Mat>& kLeftDep,
           const std::shared_ptr<Mat>& kLeftVal, const Affine4f kAbsoPose)
      : left_img_ptr_(kLeftImg), right_img_ptr_(kRightImg), left_dep_ptr_(kLeftDep), left_val_ptr_(kLeftVal),
        abso_pose_(kAbsoPose) {}
  KeyFrame(const KeyFrame&) = delete;
  KeyFrame& operator=(const KeyFrame&) = delete;
  const Mat& GetLeftImg() { return *left_img_ptr_; }
  const Mat& GetRightImg() { return *right_img_ptr_; }
  const Mat& GetLeftDep() { return *left_dep_ptr_; }
  const Mat& GetLeftVal() { return *left_val_ptr_; }
  const Affine4f GetAbsoPose() { return abso_pose_; }
  Mat& ModifyLeftDep() { return *left_dep_ptr_; }
  Mat& ModifyLeftVal() { return *left_val_ptr_; }
  Affine4f& ModifyAbsoPose() { return abso_pose_; }
 private:
  std::shared_ptr<Mat> left_img_ptr_, right_img_ptr_, left_dep_ptr_, left_val_ptr_;
  Affine4f abso_pose_;
};

}  // namespace odometry
#endif  // ODOMETRY_SHIM_HPP
