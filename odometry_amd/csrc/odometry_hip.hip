// odometry_hip.hip — C ABI (include/odometry_hip.h) over the gfx950 kernels in kernels.hip.h.
// Host side only sequences launches on the context's HIP stream; all arithmetic of the hot path runs on
// the device. There is no CPU fallback: every entry point fails with -1 if HIP does.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#include <atomic>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/odometry_hip.h"
#include "kernels.hip.h"

using namespace odo;

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
#define HIP_OK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" const char* odo_last_error(void) { return g_err; }
extern "C" int odo_version(void) { return 100; }

// Home XCDs of the persistent launches (kernels.hip.h, fine_read_classes): every user of one — a pose optimiser, a depth estimator —
// takes the next XCD of its device when it is created, so that the launches that run side by side (a tracker's pose LM and its depth
// LM; up to four trackers of one process) never sit on the same XCD.
// The XCC ids themselves are read from the device once (eight blocks of one launch, one per XCD); if they are not eight distinct
// ids (a partitioned device), homes are off (-1) and the launches use their block class 0 wherever it lands, as before round 4.
static std::atomic<unsigned> g_next_home_xcd[16];
static std::mutex g_xcc_mu;
static int g_xcc_state[16];      // 0 unknown, 1 eight distinct ids, -1 not usable
static XccIds g_xcc_ids[16];
static const XccIds& device_xcc_ids(int device) {
  std::lock_guard<std::mutex> lk(g_xcc_mu);
  const int dv = device & 15;
  if (g_xcc_state[dv] == 0) {
    g_xcc_state[dv] = -1;
    for (int& v : g_xcc_ids[dv].id) v = -1;
    int* d = nullptr;
    int h[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    if (!getenv("ODO_NO_HOME_XCD") && hipMalloc((void**)&d, sizeof(h)) == hipSuccess) {
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(8), dim3(64), 0, 0, d);
      if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
        unsigned seen = 0u;
        for (int v : h) if (v >= 0 && v < 16) seen |= 1u << v;
        if (__builtin_popcount(seen) == 8) { for (int i = 0; i < 8; i++) g_xcc_ids[dv].id[i] = h[i]; g_xcc_state[dv] = 1; }
      }
      (void)hipFree(d);
    }
    (void)hipGetLastError();
  }
  return g_xcc_ids[dv];
}
// Measured (round 4, tools/persist_conflict_probe.py): with homes 3 288 frames/s, with the launches left where the dispatcher puts them
// 3 341 — and the occasional give-up of a depth launch did not go away (it is not the two launches meeting on one XCD). Homes are
// therefore OFF unless ODO_HOME_XCD=1 asks for them (several trackers in one process).
static int next_home_xcd(int device) {
  static const bool on = getenv("ODO_HOME_XCD") != nullptr;
  if (!on) return -1;
  const XccIds& x = device_xcc_ids(device);
  const unsigned k = g_next_home_xcd[device & 15].fetch_add(1u);
  return x.id[0] >= 0 ? x.id[k & 7u] : -1;
}

// ------------------------------------------------------------------------------------------------
constexpr int kStageSlots = 4;
#include <mutex>
struct odo_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  // pinned staging ring for uploads from pageable host memory: a slot is reused once the copy that read it has finished
  void* stage[kStageSlots];
  size_t stage_cap[kStageSlots];
  hipEvent_t stage_ev[kStageSlots];
  int stage_busy[kStageSlots];
  int stage_next;
  std::vector<struct PoolBlock>* pool;  // recycled device blocks (see dev_alloc_any)
  std::vector<hipEvent_t>* pool_events; // release events not in use (creating / destroying one per block and frame costs ~3 us of host time each)
  std::mutex* mu;                       // guards pool + staging ring + upload tickets (a context may be shared by host threads)
  // upload tickets (odo_ctx_upload_ticket / odo_ctx_upload_wait): ticket t is retired once up_ev[t % kUpRing] — recorded behind
  // the t-th asynchronous upload, or behind a later one that reused the ring entry — has passed
  hipEvent_t up_ev[16];
  unsigned long up_issued, up_retired;
  hipEvent_t sw_ev[32];  // odo_ctx_stream_wait / odo_ctx_mark: events recorded on this stream for other streams to wait on (ring)
  unsigned long sw_next;
  void* bounce;          // pinned: downloads into caller memory that is not page-locked go through it (copy_to_user_host)
  size_t bounce_cap;
  // per-sequence argument table of the batched Solves issued on this stream (odo_lm_solve_batch): per context, because the
  // launches a finished Solve still has queued read it, and only the stream orders the next upload behind them
  void* lm_batch_h;   // pinned
  void* lm_batch_d;
  int lm_batch_cap;   // entries
  // tables of the batched UNFUSED pipeline (dense levels of a batched Solve): per pyramid level a row of n entries
  void *dense_h, *dense_d, *upd_h, *upd_d;   // DenseBatchItem / UpdItem [ODO_MAX_LEVELS][dense_cap]
  int dense_cap;
  struct LmBatchJob* lm_batch_job;   // the batched Solve in flight on this stream, if any
  int batch_fine_bails;              // batched Solves whose persistent launch gave up and were redone on the step launches (lifetime)
  int batch_fine_strikes, batch_fine_clean;   // as odo_lm::fine_strikes / fine_clean, for the batched launch of this context
  int batch_fine_offs;                        // as odo_lm::fine_offs
};

// ---- recycled device memory -------------------------------------------------------------------------------------------
// Per-frame objects of the drop-in path (a pyramid per constructor call, ref: run_odometry_kitti_offline.cpp:205,251-252) take
// their device blocks from a free list kept by the context and give them back to it: after the first frames an allocation
// is a list hit and a release neither synchronises the stream nor calls the driver. A recycled block may be handed out
// while work that used it is still queued — on the SAME stream, so the new owner's work is ordered behind it; that is why
// the list belongs to the context (one stream) and why objects that move between streams (the tracker's pyramids) do not
// use it. (hipMallocAsync / hipFreeAsync, the runtime's own stream-ordered allocator, was tried first: with blocks of
// several sizes cycling through it the LM read stale images on ROCm 7.2 — tests/test_gpu_dense_1080p.py run in one process
// — so the reuse rule is spelled out here instead.) ODO_NO_POOL=1 turns recycling off.
struct PoolBlock { size_t bytes; void* p; hipEvent_t ev; };   // ev: recorded on the stream when the block was released
constexpr size_t kPoolMaxBlocks = 48;
static bool pool_enabled() { static const bool on = getenv("ODO_NO_POOL") == nullptr; return on; }
// *pooled tells dev_free_any how the block has to be returned.
static int dev_alloc_any(odo_ctx* c, size_t bytes, void** out, bool* pooled) {
  *pooled = pool_enabled();
  if (*pooled && c->pool) {
    std::lock_guard<std::mutex> lk(*c->mu);
    auto& v = *c->pool;
    for (size_t i = v.size(); i-- > 0;)   // most recently released first
      if (v[i].bytes == bytes) { *out = v[i].p; if (v[i].ev) c->pool_events->push_back(v[i].ev); v.erase(v.begin() + (long)i); return 0; }
  }
  HIP_OK(hipMalloc(out, bytes));
  return 0;
}
static void dev_free_any(odo_ctx* c, void* p, size_t bytes, bool pooled) {
  if (!p) return;
  if (pooled && c->pool) {
    // The list never refuses a block: when it is full, the OLDEST entry (a size nobody has asked for since it was released —
    // hits are taken from the young end) is evicted. Its release event has long passed in a running frame loop, so the
    // eviction waits for nothing; the caller's stream is never synchronised.
    PoolBlock old{0, nullptr, nullptr};
    {
      std::lock_guard<std::mutex> lk(*c->mu);
      auto& v = *c->pool;
      if (v.size() >= kPoolMaxBlocks) { old = v.front(); v.erase(v.begin()); }
      PoolBlock nb{bytes, p, nullptr};
      if (!c->pool_events->empty()) { nb.ev = c->pool_events->back(); c->pool_events->pop_back(); }
      else if (hipEventCreateWithFlags(&nb.ev, hipEventDisableTiming) != hipSuccess) nb.ev = nullptr;
      if (nb.ev) (void)hipEventRecord(nb.ev, c->stream);
      v.push_back(nb);
    }
    if (old.p) {
      if (old.ev) { (void)hipEventSynchronize(old.ev); (void)hipEventDestroy(old.ev); }
      else (void)hipStreamSynchronize(c->stream);
      (void)hipFree(old.p);
    }
    return;
  }
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(p);
}

static int ctx_create(int device, int high_priority, odo_ctx** out);
extern "C" int odo_ctx_create(int device, odo_ctx** out) { return ctx_create(device, 0, out); }
extern "C" int odo_ctx_create_high_priority(int device, odo_ctx** out) { return ctx_create(device, 1, out); }
static int ctx_create(int device, int high_priority, odo_ctx** out) {
  if (!out) return fail("odo_ctx_create: out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("odo_ctx_create: no HIP device (this library has no CPU path)");
  if (device < 0 || device >= n) return fail("odo_ctx_create: device %d out of range (have %d)", device, n);
  HIP_OK(hipSetDevice(device));
  odo_ctx* c = new (std::nothrow) odo_ctx();
  if (!c) return fail("odo_ctx_create: out of memory");
  memset(c, 0, sizeof(*c));
  c->device = device;
  if (high_priority) {
    int lo = 0, hi = 0;  // numerically lower = higher priority
    HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_OK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
  } else {
    HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  }
  HIP_OK(hipEventCreate(&c->ev0));
  HIP_OK(hipEventCreate(&c->ev1));
  for (int i = 0; i < kStageSlots; i++) HIP_OK(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
  c->pool = new std::vector<PoolBlock>();
  c->pool_events = new std::vector<hipEvent_t>();
  c->mu = new std::mutex();
  for (auto& e : c->up_ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : c->sw_ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *out = c;
  return 0;
}
static void lm_batch_job_free(odo_ctx* c);
extern "C" int odo_ctx_destroy(odo_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  for (int i = 0; i < kStageSlots; i++) {
    (void)hipEventDestroy(c->stage_ev[i]);
    if (c->stage[i]) (void)hipHostFree(c->stage[i]);
  }
  if (c->bounce) (void)hipHostFree(c->bounce);
  if (c->pool) { for (auto& b : *c->pool) { if (b.ev) (void)hipEventDestroy(b.ev); (void)hipFree(b.p); } delete c->pool; }
  if (c->pool_events) { for (auto& e : *c->pool_events) (void)hipEventDestroy(e); delete c->pool_events; }
  for (auto& e : c->up_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->sw_ev) if (e) (void)hipEventDestroy(e);
  delete c->mu;
  lm_batch_job_free(c);
  if (c->lm_batch_h) (void)hipHostFree(c->lm_batch_h);
  if (c->lm_batch_d) (void)hipFree(c->lm_batch_d);
  if (c->dense_h) (void)hipHostFree(c->dense_h);
  if (c->upd_h) (void)hipHostFree(c->upd_h);
  if (c->dense_d) (void)hipFree(c->dense_d);
  if (c->upd_d) (void)hipFree(c->upd_d);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}
extern "C" int odo_ctx_synchronize(odo_ctx* c) {
  if (!c) return fail("odo_ctx_synchronize: NULL ctx");
  HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int odo_ctx_timer_start(odo_ctx* c) {
  if (!c) return fail("NULL ctx");
  HIP_OK(hipEventRecord(c->ev0, c->stream));
  return 0;
}
extern "C" int odo_ctx_timer_stop(odo_ctx* c, float* ms) {
  if (!c || !ms) return fail("NULL arg");
  HIP_OK(hipEventRecord(c->ev1, c->stream));
  HIP_OK(hipEventSynchronize(c->ev1));
  HIP_OK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}
extern "C" int odo_dev_alloc(odo_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return fail("NULL arg");
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipMalloc(out, bytes));
  return 0;
}
extern "C" int odo_dev_free(odo_ctx* c, void* p) {
  if (!c) return fail("NULL ctx");
  HIP_OK(hipStreamSynchronize(c->stream));
  HIP_OK(hipFree(p));
  return 0;
}

// ---- pinned host memory + uploads that do not stall -----------------------------------------------------------------
// Host blocks handed out by odo_host_alloc are page-locked: an upload from one of them is a plain asynchronous DMA.
// Uploads from any other host memory go through the context's pinned staging ring (one CPU copy, then the same DMA).
// Either way the call returns as soon as the caller may reuse / release its buffer, without waiting for the device.
#include <map>
#include "host_fp.h"
static std::mutex g_pin_mu;
static std::map<const char*, size_t> g_pinned;  // start -> bytes of every live odo_host_alloc block
static bool host_is_pinned(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pinned.upper_bound((const char*)p);
  if (it == g_pinned.begin()) return false;
  --it;
  return (const char*)p >= it->first && (const char*)p + bytes <= it->first + it->second;
}
extern "C" void* odo_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pinned[(const char*)p] = bytes;
  return p;
}
extern "C" void odo_host_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.erase((const char*)p);
  }
  (void)hipHostFree(p);
}
// Device -> caller's host memory, synchronous. The runtime is never handed a pageable pointer: for a large copy it page-locks the
// caller's pages on the fly and keeps them registered, and when the caller later frees that memory (a per-frame cv::Mat) the unmap
// notifier evicts every queue of the process for milliseconds (measured round 5: a Solve in flight went from 0.3 to 23 ms).
// Page-locked destinations (odo_host_alloc) are written in place; anything else through the context's pinned bounce block.
// Several downloads, queued back to back and waited for ONCE (odo_depth_compute's three outputs). The context's mutex is held only
// to take the bounce block out of the context and to put it back (holding it across hipStreamSynchronize stalled every
// other thread that needs the context — upload tickets, odo_ctx_mark_reached from Solve's idle callback — for the whole device
// wait); a second thread downloading at the same time finds no block and brings its own.
static int copy_many_to_user_host(odo_ctx* c, int n, void* const* dst, const void* const* src, const size_t* bytes) {
  size_t need = 0;
  bool all_pinned = true;
  for (int i = 0; i < n; i++) {
    if (!bytes[i]) continue;
    if (!host_is_pinned(dst[i], bytes[i])) { all_pinned = false; need += (bytes[i] + 255) & ~(size_t)255; }
  }
  void* blk = nullptr;
  size_t cap = 0;
  if (!all_pinned) {
    {
      std::lock_guard<std::mutex> lk(*c->mu);
      blk = c->bounce; cap = c->bounce_cap;
      c->bounce = nullptr; c->bounce_cap = 0;
    }
    if (cap < need) {
      if (blk) HIP_OK(hipHostFree(blk));
      blk = nullptr; cap = 0;
      HIP_OK(hipHostMalloc(&blk, need, hipHostMallocDefault));
      cap = need;
    }
  }
  int rc = 0;
  size_t off = 0;
  for (int i = 0; i < n && !rc; i++) {
    if (!bytes[i]) continue;
    const bool pinned = all_pinned || host_is_pinned(dst[i], bytes[i]);
    void* to = pinned ? dst[i] : (void*)((char*)blk + off);
    if (hipMemcpyAsync(to, src[i], bytes[i], hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail("copy_to_user_host: hipMemcpyAsync failed");
    if (!pinned) off += (bytes[i] + 255) & ~(size_t)255;
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail("copy_to_user_host: hipStreamSynchronize failed");
  off = 0;
  for (int i = 0; i < n; i++) {
    if (!bytes[i] || all_pinned || host_is_pinned(dst[i], bytes[i])) continue;
    if (!rc) memcpy(dst[i], (const char*)blk + off, bytes[i]);
    off += (bytes[i] + 255) & ~(size_t)255;
  }
  if (blk) {
    std::lock_guard<std::mutex> lk(*c->mu);
    if (c->bounce_cap < cap) { std::swap(c->bounce, blk); std::swap(c->bounce_cap, cap); }
  }
  if (blk) (void)hipHostFree(blk);   // (the context already holds a block at least as large: another thread put it back first)
  return rc;
}
static int copy_to_user_host(odo_ctx* c, void* dst, const void* src, size_t bytes) {
  return copy_many_to_user_host(c, 1, &dst, &src, &bytes);
}
// dst (device, dense rows of row_bytes) <- src (host, pitch src_pitch), `rows` rows; asynchronous on the context's stream.
static int upload_rows_async(odo_ctx* c, void* dst, const void* src, size_t src_pitch, size_t row_bytes, int rows) {
  const size_t total = row_bytes * (size_t)rows;
  const size_t span = src_pitch * (size_t)(rows - 1) + row_bytes;
  std::lock_guard<std::mutex> lk(*c->mu);
  if (host_is_pinned(src, span)) {
    HIP_OK(hipMemcpy2DAsync(dst, row_bytes, src, src_pitch, row_bytes, rows, hipMemcpyHostToDevice, c->stream));
    // the DMA reads the caller's block in place: it may be rewritten / released once this upload's ticket has retired
    c->up_issued++;
    HIP_OK(hipEventRecord(c->up_ev[c->up_issued % 16], c->stream));
    return 0;
  }
  const int slot = c->stage_next;
  c->stage_next = (slot + 1) % kStageSlots;
  if (c->stage_busy[slot]) { HIP_OK(hipEventSynchronize(c->stage_ev[slot])); c->stage_busy[slot] = 0; }
  if (c->stage_cap[slot] < total) {
    if (c->stage[slot]) HIP_OK(hipHostFree(c->stage[slot]));
    c->stage[slot] = nullptr; c->stage_cap[slot] = 0;
    HIP_OK(hipHostMalloc(&c->stage[slot], total, hipHostMallocDefault));
    c->stage_cap[slot] = total;
  }
  if (src_pitch == row_bytes) memcpy(c->stage[slot], src, total);
  else for (int y = 0; y < rows; y++) memcpy((char*)c->stage[slot] + (size_t)y * row_bytes, (const char*)src + (size_t)y * src_pitch, row_bytes);
  HIP_OK(hipMemcpyAsync(dst, c->stage[slot], total, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipEventRecord(c->stage_ev[slot], c->stream));
  c->stage_busy[slot] = 1;
  return 0;
}
// Ticket of the most recent in-place (page-locked source) asynchronous upload of this context; 0 = none yet.
extern "C" unsigned long odo_ctx_upload_ticket(odo_ctx* c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(*c->mu);
  return c->up_issued;
}
// Returns once the upload with that ticket — and every earlier one — no longer reads its host block.
extern "C" int odo_ctx_upload_wait(odo_ctx* c, unsigned long ticket) {
  if (!c) return fail("NULL ctx");
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lk(*c->mu);
    if (ticket == 0 || ticket <= c->up_retired) return 0;
    if (ticket > c->up_issued) return fail("odo_ctx_upload_wait: ticket %lu was never issued", ticket);
    ev = c->up_ev[ticket % 16];   // this upload's event, or that of a LATER upload that reused the entry: both imply it
  }
  HIP_OK(hipEventSynchronize(ev));
  std::lock_guard<std::mutex> lk(*c->mu);
  if (ticket > c->up_retired) c->up_retired = ticket;
  return 0;
}
extern "C" unsigned long odo_ctx_mark(odo_ctx* c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(*c->mu);
  const unsigned long mark = ++c->sw_next;             // marks count from 1; ring entry = mark % 32
  if (hipEventRecord(c->sw_ev[mark % 32], c->stream) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return mark;
}
extern "C" int odo_ctx_stream_wait_mark(odo_ctx* waiter, odo_ctx* signaller, unsigned long mark) {
  if (!waiter || !signaller) return fail("odo_ctx_stream_wait_mark: NULL ctx");
  if (waiter == signaller) return 0;
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lk(*signaller->mu);
    if (mark != 0 && mark <= signaller->sw_next && signaller->sw_next - mark < 32) ev = signaller->sw_ev[mark % 32];
  }
  if (!ev) return odo_ctx_stream_wait(waiter, signaller);   // unknown or overwritten mark: behind everything queued so far
  HIP_OK(hipStreamWaitEvent(waiter->stream, ev, 0));
  return 0;
}
extern "C" int odo_ctx_stream_wait(odo_ctx* waiter, odo_ctx* signaller) {
  if (!waiter || !signaller) return fail("odo_ctx_stream_wait: NULL ctx");
  if (waiter == signaller) return 0;
  const unsigned long mark = odo_ctx_mark(signaller);
  if (mark == 0) return fail("odo_ctx_stream_wait: hipEventRecord failed");
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lk(*signaller->mu);
    ev = signaller->sw_ev[mark % 32];  // (a ring entry is re-recorded 32 marks later: a stream that was told to wait for it has
  }                                    //  captured the earlier record by then — hipStreamWaitEvent snapshots the event's state)
  HIP_OK(hipStreamWaitEvent(waiter->stream, ev, 0));
  return 0;
}
extern "C" int odo_dev_upload(odo_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c) return fail("NULL ctx");
  if (upload_rows_async(c, dst, src, bytes, bytes, 1)) return -1;   // (staged: no pageable pointer reaches the runtime)
  HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int odo_dev_upload_async(odo_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || !dst || !src) return fail("NULL arg");
  return upload_rows_async(c, dst, src, bytes, bytes, 1);
}
extern "C" int odo_dev_upload_2d_async(odo_ctx* c, void* dst, const void* src, size_t src_pitch, size_t row_bytes, int rows) {
  if (!c || !dst || !src || rows < 1 || src_pitch < row_bytes) return fail("odo_dev_upload_2d_async: bad arg");
  return upload_rows_async(c, dst, src, src_pitch, row_bytes, rows);
}
// ---- images in memory the library cannot watch (cv::Mat): fingerprints, staged uploads that fingerprint what they stage,
// downloads into page-locked memory that are copied out (and fingerprinted) later — see host_fp.h
extern "C" unsigned long long odo_host_fingerprint(const void* src, size_t src_pitch, size_t row_bytes, int rows) {
  if (!src || rows < 1 || src_pitch < row_bytes) return 0;
  return hostfp::image(src, src_pitch, row_bytes, rows, nullptr, 0);
}
extern "C" unsigned long long odo_host_copy_fingerprint(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes,
                                                        int rows) {
  if (!dst || !src || rows < 1 || src_pitch < row_bytes || dst_pitch < row_bytes) return 0;
  return hostfp::image(src, src_pitch, row_bytes, rows, dst, dst_pitch);
}
extern "C" int odo_dev_upload_fp_async(odo_ctx* c, void* dst, const void* src, size_t src_pitch, size_t row_bytes, int rows,
                                       unsigned long long* fp) {
  if (!c || !dst || !src || !fp || rows < 1 || src_pitch < row_bytes) return fail("odo_dev_upload_fp_async: bad arg");
  const size_t total = row_bytes * (size_t)rows;
  std::lock_guard<std::mutex> lk(*c->mu);
  const int slot = c->stage_next;
  c->stage_next = (slot + 1) % kStageSlots;
  if (c->stage_busy[slot]) { HIP_OK(hipEventSynchronize(c->stage_ev[slot])); c->stage_busy[slot] = 0; }
  if (c->stage_cap[slot] < total) {
    if (c->stage[slot]) HIP_OK(hipHostFree(c->stage[slot]));
    c->stage[slot] = nullptr; c->stage_cap[slot] = 0;
    HIP_OK(hipHostMalloc(&c->stage[slot], total, hipHostMallocDefault));
    c->stage_cap[slot] = total;
  }
  *fp = hostfp::image(src, src_pitch, row_bytes, rows, c->stage[slot], row_bytes);   // the fingerprint of exactly the bytes that go up
  // (Round 5, measured and dropped: the image sent in 256 KB - 1 MB pieces, each piece's DMA queued from inside the copy loop while the
  //  next piece is copied and hashed — the frame got SLOWER with every extra piece, cv::Mat runner 1 568 -> 1 558 / 1 535 / 1 438
  //  frames/s for 2 / 4 / 8 pieces: a hipMemcpyAsync call costs the host more than the overlap gives back. The other direction —
  //  ONE launch queued first whose blocks fetch the staging block over PCIe piece by piece as the copy loop announces the pieces with
  //  plain stores into page-locked memory — was built as well: bit-identical, the Solve behind the upload starts 8 us sooner, and the
  //  copy loop the device reads behind gets 18 us slower: 1 555-1 730 against 1 579-1 711 frames/s, nothing.)
  HIP_OK(hipMemcpyAsync(dst, c->stage[slot], total, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipEventRecord(c->stage_ev[slot], c->stream));
  c->stage_busy[slot] = 1;
  return 0;
}
extern "C" int odo_dev_download_async(odo_ctx* c, void* dst_pinned, const void* src, size_t bytes) {
  if (!c || !dst_pinned || !src) return fail("odo_dev_download_async: NULL arg");
  if (!host_is_pinned(dst_pinned, bytes)) return fail("odo_dev_download_async: the destination is not an odo_host_alloc block");
  HIP_OK(hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return 0;
}
extern "C" int odo_ctx_wait_mark(odo_ctx* c, unsigned long mark) {
  if (!c) return fail("odo_ctx_wait_mark: NULL ctx");
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lk(*c->mu);
    if (mark != 0 && mark <= c->sw_next && c->sw_next - mark < 32) ev = c->sw_ev[mark % 32];
  }
  if (ev) HIP_OK(hipEventSynchronize(ev));     // (a ring entry re-recorded since: a later point of the same stream — implies the mark)
  else HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int odo_ctx_mark_reached(odo_ctx* c, unsigned long mark) {
  if (!c) return fail("odo_ctx_mark_reached: NULL ctx");
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lk(*c->mu);
    if (mark != 0 && mark <= c->sw_next && c->sw_next - mark < 32) ev = c->sw_ev[mark % 32];
  }
  const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(c->stream);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return fail("odo_ctx_mark_reached: %s", hipGetErrorString(e));
}
extern "C" int odo_dev_download(odo_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || !dst || !src) return fail("odo_dev_download: NULL arg");
  return copy_to_user_host(c, dst, src, bytes);
}
// Recycled scratch for per-frame images of the drop-in path (see dev_alloc_any): the release does not synchronise.
extern "C" int odo_dev_alloc_async(odo_ctx* c, size_t bytes, void** out, int* is_async) {
  if (!c || !out || !is_async) return fail("NULL arg");
  HIP_OK(hipSetDevice(c->device));
  bool a = false;
  if (dev_alloc_any(c, bytes, out, &a)) return -1;
  *is_async = a ? 1 : 0;
  return 0;
}
extern "C" int odo_dev_free_async(odo_ctx* c, void* p, size_t bytes, int is_async) {
  if (!c) return fail("NULL ctx");
  dev_free_any(c, p, bytes, is_async != 0);
  return 0;
}

static inline dim3 grid2d(int cols, int rows, int z = 1) { return dim3((cols + 63) / 64, (rows + 3) / 4, z); }

// ------------------------------------------------------------------------------------------------
// Pyramids
// ------------------------------------------------------------------------------------------------
struct odo_pyr {
  odo_ctx* ctx;
  int kind, levels, rows, cols;
  float* dev;      // all levels back to back
  float* staging;  // level-0-sized device copy of a host input (IMAGE kind: pyrDown reads the unsmoothed input)
  bool dev_async, staging_async;  // how the two blocks were allocated (dev_alloc_any)
  size_t dev_bytes, staging_bytes;
  size_t off[ODO_MAX_LEVELS];
  int r[ODO_MAX_LEVELS], c[ODO_MAX_LEVELS];
  unsigned long long version;  // bumped by every (re)build: keys the LM's keyframe point-list cache
};
static std::atomic<unsigned long long> g_pyr_version{0};

static int pyr_build(odo_pyr* p, const float* img_dev, int smooth) {
  hipStream_t s = p->ctx->stream;
  p->version = ++g_pyr_version;
  const int rows = p->rows, cols = p->cols;
  if (p->kind == ODO_PYR_DEPTH && smooth) {
    // cv::medianBlur(in, L0, 3) (ref: src/image_processing_global.cpp:76-80; no caller of the reference passes smooth = true): the
    // median goes into level 0 and the levels below are decimated from it (the fused kernel's level-0 copy is then in place)
    hipLaunchKernelGGL(median3x3_kernel, grid2d(cols, rows), dim3(256), 0, s, img_dev, p->dev, rows, cols);
    img_dev = p->dev;
    smooth = 0;
  }
  static const bool unfused = getenv("ODO_PYR_UNFUSED") != nullptr;
  if (p->levels <= 4 && !unfused) {  // whole pyramid in one launch
    PyrOut o;
    memset(&o, 0, sizeof(o));
    o.n_levels = p->levels; o.smooth = smooth ? 1 : 0;
    for (int l = 0; l < p->levels; l++) { o.lvl[l] = p->dev + p->off[l]; o.rows[l] = p->r[l]; o.cols[l] = p->c[l]; }
    if (p->kind == ODO_PYR_IMAGE) {
      const dim3 tiles((cols + kPT - 1) / kPT, (rows + kPT - 1) / kPT);
      static const int wide_from = getenv("ODO_PYR_WIDE_FROM") ? atoi(getenv("ODO_PYR_WIDE_FROM")) : 1024;   // tiles
      if ((int)(tiles.x * tiles.y) >= wide_from)
        hipLaunchKernelGGL(image_pyramid_fused_wide_kernel, tiles, dim3(kPyrThreadsWide), 0, s, img_dev, o);
      else
        hipLaunchKernelGGL(image_pyramid_fused_kernel, tiles, dim3(kPyrThreads), 0, s, img_dev, o);
    }
    else
      hipLaunchKernelGGL(depth_pyramid_fused_kernel, grid2d(cols, rows), dim3(256), 0, s, img_dev, o);
    HIP_OK(hipGetLastError());
    return 0;
  }
  if (p->kind == ODO_PYR_IMAGE) {
    if (smooth) {
      hipLaunchKernelGGL(blur3x3_kernel, grid2d(cols, rows), dim3(256), 0, s, img_dev, p->dev, img_dev, p->dev, rows, cols,
                         (uint8_t*)nullptr, (float*)nullptr, (float*)nullptr);
    } else {
      HIP_OK(hipMemcpyAsync(p->dev, img_dev, sizeof(float) * (size_t)rows * cols, hipMemcpyDeviceToDevice, s));
    }
    const float* prev = img_dev;  // L1 comes from the UNSMOOTHED input (ref: src/image_processing_global.cpp:38)
    for (int l = 1; l < p->levels; l++) {
      hipLaunchKernelGGL(pyrdown_kernel, grid2d(p->c[l], p->r[l]), dim3(256), 0, s, prev, p->r[l - 1], p->c[l - 1],
                         p->dev + p->off[l]);
      prev = p->dev + p->off[l];
    }
  } else {
    if (img_dev != p->dev) HIP_OK(hipMemcpyAsync(p->dev, img_dev, sizeof(float) * (size_t)rows * cols, hipMemcpyDeviceToDevice, s));
    for (int l = 1; l < p->levels; l++)
      hipLaunchKernelGGL(decimate_odd_kernel, grid2d(p->c[l], p->r[l]), dim3(256), 0, s, p->dev + p->off[l - 1],
                         p->c[l - 1], p->dev + p->off[l], p->r[l], p->c[l]);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// pooled: stream-ordered allocation on ctx's stream (per-frame pyramids of the drop-in path, used on that stream only);
// false: a plain allocation (the tracker's long-lived pyramids move between its two streams).
static int pyr_alloc(odo_ctx* ctx, int rows, int cols, int levels, int kind, odo_pyr** out, bool pooled = true) {
  if (!ctx || !out) return fail("pyramid: NULL arg");
  *out = nullptr;
  if (levels < 1 || levels > ODO_MAX_LEVELS) return fail("pyramid: levels %d out of range", levels);
  if (rows < 1 || cols < 1) return fail("pyramid: bad size %dx%d", rows, cols);
  if (kind != ODO_PYR_IMAGE && kind != ODO_PYR_DEPTH) return fail("pyramid: bad kind %d", kind);
  odo_pyr* p = new (std::nothrow) odo_pyr();
  if (!p) return fail("out of memory");
  p->ctx = ctx; p->kind = kind; p->levels = levels; p->rows = rows; p->cols = cols;
  p->dev = nullptr; p->staging = nullptr; p->dev_async = p->staging_async = false; p->dev_bytes = p->staging_bytes = 0;
  size_t tot = 0;
  int r = rows, c = cols;
  for (int l = 0; l < levels; l++) {
    p->off[l] = tot; p->r[l] = r; p->c[l] = c;
    tot += (size_t)r * c;
    r /= 2; c /= 2;
    if ((r < 1 || c < 1) && l + 1 < levels) { delete p; return fail("pyramid: image too small for %d levels", levels); }
  }
  HIP_OK(hipSetDevice(ctx->device));
  p->dev_bytes = sizeof(float) * tot;
  if (pooled) {
    if (dev_alloc_any(ctx, sizeof(float) * tot, (void**)&p->dev, &p->dev_async)) { delete p; return -1; }
  } else {
    p->dev_async = false;
    if (hipMalloc((void**)&p->dev, sizeof(float) * tot) != hipSuccess) { delete p; return fail("pyramid: hipMalloc failed"); }
  }
  *out = p;
  return 0;
}

extern "C" int odo_pyramid_create(odo_ctx* ctx, const float* img, int rows, int cols, size_t stride_bytes, int levels,
                                  int smooth, int kind, odo_pyr** out) {
  if (!img) return fail("odo_pyramid_create: NULL image");
  if (stride_bytes == 0) stride_bytes = sizeof(float) * (size_t)cols;
  if (stride_bytes < sizeof(float) * (size_t)cols) return fail("odo_pyramid_create: stride smaller than a row");
  odo_pyr* p = nullptr;
  if (pyr_alloc(ctx, rows, cols, levels, kind, &p)) return -1;
  p->staging_bytes = sizeof(float) * (size_t)rows * cols;
  if (dev_alloc_any(ctx, p->staging_bytes, (void**)&p->staging, &p->staging_async)) {
    odo_pyramid_destroy(p);
    return fail("odo_pyramid_create: device allocation failed");
  }
  // The caller may release `img` as soon as this returns (the reference's constructor copies synchronously): the image is
  // copied into pinned staging memory on the spot (or, when it lives in an odo_host_alloc block, read by the DMA later —
  // such a block must outlive the stream's pending work). Nothing here waits for the device.
  if (upload_rows_async(ctx, p->staging, img, stride_bytes, sizeof(float) * (size_t)cols, rows)) { odo_pyramid_destroy(p); return -1; }
  if (pyr_build(p, p->staging, smooth)) { odo_pyramid_destroy(p); return -1; }
  *out = p;
  return 0;
}

extern "C" int odo_pyramid_create_dev(odo_ctx* ctx, const float* img_dev, int rows, int cols, int levels, int smooth,
                                      int kind, odo_pyr** out) {
  if (!img_dev) return fail("odo_pyramid_create_dev: NULL image");
  odo_pyr* p = nullptr;
  if (pyr_alloc(ctx, rows, cols, levels, kind, &p)) return -1;
  if (pyr_build(p, img_dev, smooth)) { odo_pyramid_destroy(p); return -1; }
  *out = p;
  return 0;
}

extern "C" int odo_pyramid_rebuild_dev(odo_pyr* p, const float* img_dev, int smooth) {
  if (!p || !img_dev) return fail("odo_pyramid_rebuild_dev: NULL arg");
  return pyr_build(p, img_dev, smooth);
}

extern "C" int odo_pyramid_levels(const odo_pyr* p) { return p ? p->levels : -1; }
extern "C" int odo_pyramid_level_dims(const odo_pyr* p, int level, int* rows, int* cols) {
  if (!p || level < 0 || level >= p->levels) return fail("odo_pyramid_level_dims: bad level");
  if (rows) *rows = p->r[level];
  if (cols) *cols = p->c[level];
  return 0;
}
extern "C" int odo_pyramid_download(const odo_pyr* p, int level, float* dst) {
  if (!p || !dst) return fail("odo_pyramid_download: NULL arg");
  if (level < 0 || level >= p->levels)  // ref: src/image_pyramid.cpp:22-25 exits the process; here: -1
    return fail("Requested image pyramid does not exist! Max pyramid id: %d", p->levels - 1);
  return copy_to_user_host(p->ctx, dst, p->dev + p->off[level], sizeof(float) * (size_t)p->r[level] * p->c[level]);
}
extern "C" const float* odo_pyramid_level_dev(const odo_pyr* p, int level) {
  if (!p || level < 0 || level >= p->levels) return nullptr;
  return p->dev + p->off[level];
}
extern "C" int odo_pyramid_destroy(odo_pyr* p) {
  if (!p) return 0;
  // stream-ordered blocks go back to the pool behind whatever the stream still has queued on them; plain ones need the
  // stream drained first
  dev_free_any(p->ctx, p->dev, p->dev_bytes, p->dev_async);
  dev_free_any(p->ctx, p->staging, p->staging_bytes, p->staging_async);
  delete p;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Pose LM
// ------------------------------------------------------------------------------------------------
static inline LevelK lm_level_k(const struct odo_lm* m, int level);
static const odo_intrinsics kKitti00 = {718.856f, (float)607.1928, (float)185.2157};
constexpr int kLmMaxBlocks = 1280;       // partial rows per buffer: dense scan = 5 blocks per CU

// One set of keyframe-candidate point lists (see odo_lm::cand).
struct LmCandSet {
  PointList pl[ODO_MAX_LEVELS];
  size_t pl_cap[ODO_MAX_LEVELS];
  int* d_rowcnt; int* d_npts; int* h_npts; int* hm_npts; int rows_cap;
  long tag;   // caller's tag of the candidate (frame id), -1 = none
};

// A fused Solve in flight (see lm_fused_begin / odo_lm_solve_begin).
struct LmJob {
  int active;
  const odo_pyr *kf_img, *kf_dep, *cur_img;
  unsigned long long kf_img_ver, kf_dep_ver, cur_ver;
  StepArgs a;
  int seq, launches, it, budget, grid, token, min_level, stop_level;
  int slot;   // which of the two host-mapped result blocks its finishing launch writes (a chained Solve takes the other one)
  bool poll, result_by_launch, issued_all;
  double bytes_per_level[ODO_MAX_LEVELS];
  int fine_lo;          // lm_plan_levels' answer at begin (an armed Solve's persistent launch is issued later: lm_arm_go)
};

struct odo_lm {
  odo_ctx* ctx;
  float lambda, precision, huber_delta;
  int n_levels, robust;
  int max_iters[ODO_MAX_LEVELS];
  odo_intrinsics K;
  float init[16];
  // device
  LmState* d_state;   // [0], [1]: the fused pipeline's double buffer ([0]: a whole unfused Solve); [2]: unfused continuation of a fused Solve
  LmState* ust;       // the state the unfused kernels work on (d_state, or d_state + 2 during a continuation)
  int upo;            // offset of the unfused pipeline's host-mapped progress words (0, or 16 during a continuation)
  int fuse_dense_max; // dense levels with at most this many points run on their point list through the fused pipeline (0: never)
  double* d_partials;
  float* d_init;
  float* d_out;  // 26 floats
  LmTraceRow* d_trace;
  float* d_cost;  // 16 floats
  float* d_res;   // t-dist residual buffer (lazy)
  size_t res_cap;
  float* d_scale;
  unsigned long long* d_ts_xbuf;   // lm_tdist_scale_multi_kernel: exchange buffer (kTsXbufWords), launch epoch, give-up flag
  int* d_ts_gave_up;
  unsigned ts_epoch;
  int ts_multi;                    // 0: ODO_TDIST_SINGLE=1 (the single-workgroup scale kernel for every level)
  int fine_passes;                 // passes per evaluation a level may take inside the persistent launch (lm_plan_levels; ODO_LM_FINE_PASSES)
  int ts_fault;                    // test hook (ODO_TDIST_MULTI_FAULT): a workgroup never publishes
  long ts_multi_launches;          // lm_tdist_scale_multi_kernel launches so far (odo_lm_tdist_stats)
  unsigned ts_wait;                // wait bound in ticks of the 100 MHz clock (ODO_LM_FINE_WAIT_US), 0: 4 ms
  // host-mapped progress words the update kernel writes (early-exit polling)
  int* h_prog;
  int* d_prog;
  int poll;  // 0 = enqueue every launch blindly
  int run_ahead;
  // keyframe point lists (semi-dense levels), cached per (kf_img, kf_dep) build
  PointList pl[ODO_MAX_LEVELS];
  size_t pl_cap[ODO_MAX_LEVELS];
  int npts[ODO_MAX_LEVELS];
  int use_list[ODO_MAX_LEVELS];
  unsigned long long kf_img_ver, kf_dep_ver;
  int* d_rowcnt; int* d_npts; int* h_npts; int* hm_npts /* device alias of h_npts */; int rows_cap;
  // A second, identical set of list buffers for the keyframe CANDIDATE of the frame being tracked: the tracker fills it on
  // its depth stream every frame (lm_build_candidate), off the Solve's critical path; when the candidate becomes the
  // keyframe the two sets trade places (lm_adopt_candidate) instead of list-building launches + a read-back in front of the Solve.
  LmCandSet cand[2];   // two of them: the tracker's depth stream may run a frame ahead of the pose LM (slot = job parity)
  // odo_lm_candidate_begin (an optimiser that is NOT a tracker's): the lists of a frame that may become the keyframe, built ahead
  // on another stream into cand[1]; lm_prepare_keyframe adopts them when a Solve names exactly those pyramids
  struct Ahead { int active; unsigned long long img_ver, dep_ver; hipEvent_t ev; } ahead;
  // optional per-launch HIP-event timing of the evaluation kernels (bench.py roofline leg)
  int ev_on;            // 0 off; N >= 1: every N-th launch of a Solve carries start / stop events (1 = every launch)
  int ev_phase;         // which residue of the launch index is sampled in the Solve in flight (rotates Solve by Solve)
  long ev_solves;
  std::vector<hipEvent_t>* ev_pool;
  double ev_total_us, ev_bytes, ev_coarse_us;
  long ev_launches, ev_active, ev_coarse_launches;   // launches issued / evaluations / SAMPLED coarse launches
  long ev_sampled, ev_coarse_all;                    // sampled launches (step + coarse) / all coarse launches
  // execution spans of the sampled launches (StepArgs::span): a ring of {min start, max end} device wall-clock pairs, one slot
  // per sampled launch, collected (stream sync + one copy) when the statistics are read
  unsigned long long* d_span;
  int span_used;
  struct SpanTag { unsigned char coarse; long solve; int idx; };
  std::vector<SpanTag>* span_kind;                   // per assigned slot: which launch (coarse / step, Solve number, launch index)
  double ev_period_us, ev_cperiod_us;                // start-to-start distance of two CONSECUTIVE sampled launches (step / coarse first)
  long ev_period_n, ev_cperiod_n;
  int last_coarse;  // 1 if the last Solve started with the single-workgroup coarse kernel
  int last_coarse_batch;  // (lms[0] of a batched Solve) 1 if the batch began with a coarse launch
  int trace_stale;
  int record;      // 1: per-evaluation trace rows and per-level cost statistics are written (odo_lm_trace / odo_lm_report);
                   // the trackers switch it off for their own optimisers (ODO_LM_TRACE=1 keeps it)
  unsigned reset_gen;   // bumped by odo_lm_reset: a batched Solve started early is tied to it
  float* h_res; float* d_res_map; int* h_done; int* d_done; int token;  // host-mapped result + completion word
  int coarse;  // 1 = levels with <= kCoarseMaxPoints points run inside one workgroup (fused pipeline)
  // Fine point-list levels in ONE persistent launch (lm_fine_kernel) instead of a step launch per evaluation: fine_k workgroups
  // on one XCD exchange their partial rows through d_xbuf. 0 = off (ODO_LM_NO_FINE: step launches; also the fall-back). The batched Solve has its own launch (lm_fine_kernel_batch).
  int fine_k;
  int fine_bails;   // Solves whose persistent launch gave up and that were redone on the step launches (lifetime count)
  int fine_k_cfg;   // the configured number of workgroups (fine_k is 0 while the launch is switched off after three strikes)
  int fine_k_last;  // what the last persistent launch was issued with (lm_fine_launch_k)
  int fine_strikes; // give-ups that count towards switching off; a long run of clean Solves forgives them
  int fine_clean;   // Solves since the last give-up (persistent launch on) / since it was switched off (off)
  int fine_offs;    // times the launch has been switched off since the last forgiveness: the retry interval doubles with each (fine_retry_after)
  unsigned fine_wait;   // bound of one wait inside the launch, wall-clock ticks (ODO_LM_FINE_WAIT_US; 0: the kernel's default, 4 ms)
  unsigned fine_epoch;  // tag epoch of the exchange buffer (lm_fine_next_epoch)
  int fine_fault;   // test hook (ODO_LM_FINE_FAULT): the first partial row of the persistent launch is never published
  int fine_home;    // the XCD its persistent launch runs on (next_home_xcd)
  unsigned long long* d_xbuf;
  int fused;  // 1 = one launch per evaluation (robust 0/1); 0 = separate residual / update kernels
  int mode;  // 0 auto (list when <= half of the interior has depth), 1 always dense scan, 2 always list
  int bilinear;  // odo_lm_set_sampling: 1 = bilinear sampling of the current image (non-parity option)
  unsigned* dispatch_words;   // device address of g_lm_fine_dispatch on this optimiser's device (looked up once, at creation)
  int dense_plain_div;  // 1 = dense levels use the plain IEEE divisions only (ODO_DENSE_PLAIN_DIV: A/B of the shared-reciprocal path)
  void (*idle_pump)(void*);  // called while the host waits for the device (the tracker feeds its depth stream here)
  void* idle_arg;
  // pinned host mirrors
  float* h_out;
  LmTraceRow* h_trace;
  float* h_cost;
  // stats of the last solve
  int last_evals, last_launches;
  double last_bytes;
  int iters[ODO_MAX_LEVELS];
  int launches_level[ODO_MAX_LEVELS];
  LmJob job;
  // A Solve queued BEHIND the one in flight before its result exists (lm_chain_begin: the trackers' next Solve): it starts from the
  // pose the finishing launch of `job` leaves in d_chain_pose, if the guard in d_chain_guard lets it (success, keyframe kept).
  LmJob chained;
  // Armed Solves (the single tracker; lm_arm_begin / lm_arm_go / lm_arm_abort): the NEXT Solve's coarse launch is queued behind the
  // Solve in flight before that Solve's result exists, and reads the host's word (StepArgs::arm) when it starts.
  LmJob armed;
  int arm_on;                                                 // lm_enable_arming was called (the tracker's optimiser)
  unsigned long long* h_arm; unsigned long long* d_arm_map;   // host-mapped: 17 tagged granules (pose + verdict), one armed Solve at a time
  long arm_used, arm_aborted;
  long arm_fault_at; unsigned arm_wait;   // test hook (ODO_ARM_FAULT) / the armed launch's wait bound in ticks (ODO_ARM_WAIT_US; 0: 2 s), read by lm_enable_arming
  float* d_chain_pose;     // [16]
  int* d_chain_guard;
  float kf_rule[7];        // the runner's keyframe test (six weights, threshold): lm_set_chain_rule
  int kf_on;
  int slot_next;
  int last_token, last_slot;   // of the Solve odo_lm_solve collected last (lm_chain_verdict)
};

static inline LevelK lm_level_k(const odo_lm* m, int level) {
  LevelK k = make_level_k(m->K.f0, m->K.cx0, m->K.cy0, level);
  k.bilinear = m->bilinear;
  return k;
}

extern "C" int odo_lm_create(odo_ctx* ctx, float lambda, float precision, const int* max_iters, int n_levels,
                             const float init_colmajor[16], int robust, float huber_delta, const odo_intrinsics* K,
                             odo_lm** out) {
  if (!ctx || !out || !max_iters || !init_colmajor) return fail("odo_lm_create: NULL arg");
  *out = nullptr;
  if (n_levels < 1 || n_levels > ODO_MAX_LEVELS) return fail("odo_lm_create: n_levels %d out of range", n_levels);
  if (robust < 0 || robust > 2) return fail("odo_lm_create: robust must be 0, 1 or 2");
  odo_lm* m = new (std::nothrow) odo_lm();
  if (!m) return fail("out of memory");
  memset(m, 0, sizeof(*m));
  m->ctx = ctx; m->lambda = lambda; m->precision = precision; m->huber_delta = huber_delta;
  m->n_levels = n_levels; m->robust = robust;
  for (int i = 0; i < n_levels; i++) m->max_iters[i] = max_iters[i];
  m->K = K ? *K : kKitti00;  // null camera: the reference only warns (ref: src/lm_optimizer.cpp:35-38)
  memcpy(m->init, init_colmajor, sizeof(m->init));
  HIP_OK(hipSetDevice(ctx->device));
  // the single-workgroup coarse kernel reduces through 118 KB of LDS (gfx950: up to 160 KB per workgroup)
  HIP_OK(lm_chain_setup());   // (the chain kernels live in lm_chain_kernels.hip: their dynamic LDS limit)
  HIP_OK(hipGetSymbolAddress((void**)&m->dispatch_words, HIP_SYMBOL(g_lm_fine_dispatch)));
  HIP_OK(hipMalloc((void**)&m->d_state, sizeof(LmState) * 3));                          // double-buffered (fused pipeline) + continuation
  m->ust = m->d_state; m->upo = 0;
  m->fuse_dense_max = getenv("ODO_FUSE_DENSE_MAX") ? atoi(getenv("ODO_FUSE_DENSE_MAX")) : 131072;
  HIP_OK(hipMalloc((void**)&m->d_partials, sizeof(double) * 2 * kLmMaxBlocks * ODO_NACC)); // idem
  HIP_OK(hipMalloc((void**)&m->d_init, sizeof(float) * 16));
  HIP_OK(hipMalloc((void**)&m->d_out, sizeof(float) * 48));  // 26 result floats + 16 cost statistics: one read-back
  HIP_OK(hipMalloc((void**)&m->d_trace, sizeof(LmTraceRow) * kTraceCap));
  m->d_cost = m->d_out + 26;
  HIP_OK(hipMalloc((void**)&m->d_scale, sizeof(float)));
  HIP_OK(hipMalloc((void**)&m->d_ts_xbuf, sizeof(unsigned long long) * kTsXbufWords));
  HIP_OK(hipMemset(m->d_ts_xbuf, 0, sizeof(unsigned long long) * kTsXbufWords));
  HIP_OK(hipMalloc((void**)&m->d_ts_gave_up, 2 * sizeof(int)));   // {the launch in flight gave up, scale passes redone by the fall-back so far}
  HIP_OK(hipMemset(m->d_ts_gave_up, 0, 2 * sizeof(int)));
  m->ts_epoch = 0;
  m->ts_multi = getenv("ODO_TDIST_SINGLE") ? 0 : 1;
  m->fine_passes = getenv("ODO_LM_FINE_PASSES") ? atoi(getenv("ODO_LM_FINE_PASSES")) : 2;
  m->ts_fault = getenv("ODO_TDIST_MULTI_FAULT") ? 1 : 0;
  m->ts_wait = getenv("ODO_LM_FINE_WAIT_US") ? (unsigned)(100L * atol(getenv("ODO_LM_FINE_WAIT_US"))) : 0u;
  HIP_OK(hipHostMalloc((void**)&m->h_out, sizeof(float) * 48, hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&m->h_trace, sizeof(LmTraceRow) * kTraceCap, hipHostMallocDefault));
  m->h_cost = m->h_out + 26;
  HIP_OK(hipHostMalloc((void**)&m->h_res, sizeof(float) * 48 * 2, hipHostMallocMapped | hipHostMallocCoherent));   // two result blocks (LmJob::slot)
  memset(m->h_res, 0, sizeof(float) * 48 * 2);
  HIP_OK(hipMalloc((void**)&m->d_chain_pose, sizeof(float) * 32));   // pose [0 .. 15] and guard [16] in one cache line
  HIP_OK(hipMemset(m->d_chain_pose, 0, sizeof(float) * 32));
  m->d_chain_guard = (int*)(m->d_chain_pose + 16);
  HIP_OK(hipHostGetDevicePointer((void**)&m->d_res_map, m->h_res, 0));
  HIP_OK(hipHostMalloc((void**)&m->h_done, sizeof(int) * 4, hipHostMallocMapped | hipHostMallocCoherent));
  HIP_OK(hipHostGetDevicePointer((void**)&m->d_done, m->h_done, 0));
  m->h_done[0] = 0; m->h_done[1] = 0;
  HIP_OK(hipHostMalloc((void**)&m->h_prog, sizeof(int) * 32, hipHostMallocMapped | hipHostMallocCoherent));
  HIP_OK(hipHostGetDevicePointer((void**)&m->d_prog, m->h_prog, 0));
  memset(m->h_prog, 0, sizeof(int) * 32);
  // per-level point counts: written by kf_fill_kernel straight into host-mapped memory (no copy operation on the stream);
  // d_npts / cand_d_npts are device arrays for the batched tracker, which forwards them from its last launch
  HIP_OK(hipMalloc((void**)&m->d_npts, sizeof(int) * ODO_MAX_LEVELS));
  HIP_OK(hipHostMalloc((void**)&m->h_npts, sizeof(int) * ODO_MAX_LEVELS, hipHostMallocMapped | hipHostMallocCoherent));
  HIP_OK(hipHostGetDevicePointer((void**)&m->hm_npts, m->h_npts, 0));
  for (LmCandSet& cs : m->cand) {
    HIP_OK(hipMalloc((void**)&cs.d_npts, sizeof(int) * ODO_MAX_LEVELS));
    HIP_OK(hipHostMalloc((void**)&cs.h_npts, sizeof(int) * ODO_MAX_LEVELS, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_OK(hipHostGetDevicePointer((void**)&cs.hm_npts, cs.h_npts, 0));
    cs.tag = -1;
  }
  m->ahead.active = 0;
  HIP_OK(hipEventCreateWithFlags(&m->ahead.ev, hipEventDisableTiming));
  m->record = 1;
  m->mode = getenv("ODO_LM_MODE") ? atoi(getenv("ODO_LM_MODE")) : 0;
  m->dense_plain_div = getenv("ODO_DENSE_PLAIN_DIV") ? 1 : 0;
  m->fused = getenv("ODO_LM_UNFUSED") ? 0 : 1;
  m->coarse = getenv("ODO_LM_NO_COARSE") ? 0 : 1;
  m->fine_k = getenv("ODO_LM_NO_FINE") ? 0 : (getenv("ODO_LM_FINE_K") ? atoi(getenv("ODO_LM_FINE_K")) : 32);
  if (m->fine_k < 0 || m->fine_k > kFineKMax) m->fine_k = 32;
  m->fine_k_cfg = m->fine_k;
  m->fine_fault = getenv("ODO_LM_FINE_FAULT") ? 1 : 0;
  m->fine_home = next_home_xcd(ctx->device);
  m->fine_wait = getenv("ODO_LM_FINE_WAIT_US") ? (unsigned)(100L * atol(getenv("ODO_LM_FINE_WAIT_US"))) : 0u;
  HIP_OK(hipMalloc((void**)&m->d_xbuf, sizeof(unsigned long long) * kFineXbufWords));
  HIP_OK(hipMemset(m->d_xbuf, 0, sizeof(unsigned long long) * kFineXbufWords));   // tag 0: no Solve has token 0
  m->poll = getenv("ODO_NO_POLL") ? 0 : 1;
  m->run_ahead = getenv("ODO_RUN_AHEAD") ? atoi(getenv("ODO_RUN_AHEAD")) : 2;
  HIP_OK(hipMemsetAsync(m->d_trace, 0, sizeof(LmTraceRow) * kTraceCap, ctx->stream));
  memset(m->h_trace, 0, sizeof(LmTraceRow) * kTraceCap);
  memset(m->h_out, 0, sizeof(float) * 48);
  *out = m;
  return 0;
}

static void lm_arm_abort(odo_lm* m);
extern "C" int odo_lm_destroy(odo_lm* m) {
  if (!m) return 0;
  lm_arm_abort(m);   // (an armed launch would hold the stream until its word came)
  if (m->ahead.ev) { (void)hipEventSynchronize(m->ahead.ev); (void)hipEventDestroy(m->ahead.ev); }   // lists being built ahead on another stream
  (void)hipStreamSynchronize(m->ctx->stream);
  void* dv[] = {m->d_state, m->d_partials, m->d_xbuf, m->d_init, m->d_out, m->d_trace, m->d_scale, m->d_res, m->d_chain_pose, m->d_ts_xbuf, m->d_ts_gave_up,
                m->d_rowcnt, m->d_npts, m->cand[0].d_rowcnt, m->cand[0].d_npts, m->cand[1].d_rowcnt, m->cand[1].d_npts};
  for (void* q : dv) if (q) (void)hipFree(q);
  for (int l = 0; l < ODO_MAX_LEVELS; l++) {
    void* pv[] = {m->pl[l].a, m->pl[l].b, m->pl[l].c, m->pl[l].d, m->cand[0].pl[l].a, m->cand[0].pl[l].b, m->cand[0].pl[l].c, m->cand[0].pl[l].d,
                  m->cand[1].pl[l].a, m->cand[1].pl[l].b, m->cand[1].pl[l].c, m->cand[1].pl[l].d};
    for (void* q : pv) if (q) (void)hipFree(q);
  }
  (void)hipHostFree(m->h_npts);
  (void)hipHostFree(m->cand[0].h_npts);
  (void)hipHostFree(m->cand[1].h_npts);
  if (m->ev_pool) { for (auto& e : *m->ev_pool) (void)hipEventDestroy(e); delete m->ev_pool; }
  if (m->d_span) (void)hipFree(m->d_span);
  delete m->span_kind;
  (void)hipHostFree(m->h_out); (void)hipHostFree(m->h_trace); (void)hipHostFree(m->h_prog); (void)hipHostFree(m->h_res); (void)hipHostFree(m->h_done);
  if (m->h_arm) (void)hipHostFree(m->h_arm);
  delete m;
  return 0;
}

extern "C" int odo_lm_reset(odo_lm* m, const float init_colmajor[16], float lambda) {
  if (!m || !init_colmajor) { fail("Reset optimizer failed!"); return -1; }
  m->job.active = 0;                                // a Solve started early with the old initial pose is abandoned
  m->chained.active = 0;                            // (lm_chain_adopt re-activates a chained Solve that started from exactly this pose)
  m->reset_gen++;
  memcpy(m->init, init_colmajor, sizeof(m->init));  // SetInitialAffine, ref: src/lm_optimizer.cpp:385-389
  m->lambda = lambda;                               // SetLambda, ref: :391-395
  for (int i = 0; i < ODO_MAX_LEVELS; i++) m->iters[i] = 0;  // ResetStatistics, ref: :397-405
  memset(m->h_cost, 0, sizeof(float) * 16);
  return 0;
}

static int lm_check_pyrs(const odo_lm* m, const odo_pyr* a, const odo_pyr* d, const odo_pyr* b) {
  if (!m || !a || !d || !b) return fail("LM: NULL pyramid");
  if (a->kind != ODO_PYR_IMAGE || b->kind != ODO_PYR_IMAGE || d->kind != ODO_PYR_DEPTH)
    return fail("Image types don't match in LevenbergMarquardtOptimizer::OptimizeCameraPose().");
  if (a->levels < m->n_levels || b->levels < m->n_levels || d->levels < m->n_levels)
    return fail("LM: pyramids have fewer levels than the optimiser (%d)", m->n_levels);
  for (int l = 0; l < m->n_levels; l++) {
    if (a->r[l] != b->r[l] || a->r[l] != d->r[l])  // ref: src/lm_optimizer.cpp:98-101
      return fail("Image rows don't match in LevenbergMarquardtOptimizer::OptimizeCameraPose().");
    if (a->c[l] != b->c[l] || a->c[l] != d->c[l])  // ref: :102-105
      return fail("Image cols don't match in LevenbergMarquardtOptimizer::OptimizeCameraPose().");
  }
  return 0;
}

// Dense levels: 256-thread blocks, register allocation held to four waves per SIMD (no spills), at most four blocks per CU
// (kDenseBlock / kDenseWaves / kDenseGridCap, dense.hip.h).
static inline DenseLevel lm_dense_level(const LevelView& v, const LevelK& k, int max_iters) {
  DenseLevel L;
  memset(&L, 0, sizeof(L));
  L.I1 = v.I1; L.I2 = v.I2; L.D1 = v.D1; L.rows = v.rows; L.cols = v.cols; L.k = k;
  L.fast_ok = dense_fast_ok(k.fl, k.cx, k.cy, v.rows, v.cols);
  L.max_iters = max_iters;
  dense_level_geometry(&L, kDenseBlock, kDenseGridCap);
  return L;
}
static inline int lm_grid(int rows, int cols) {
  DenseLevel L;
  memset(&L, 0, sizeof(L));
  L.rows = rows; L.cols = cols;
  dense_level_geometry(&L, kDenseBlock, kDenseGridCap);
  return L.nblk;
}

static int lm_ensure_res(odo_lm* m, size_t n) {
  if (m->res_cap >= n) return 0;
  if (m->d_res) { HIP_OK(hipStreamSynchronize(m->ctx->stream)); HIP_OK(hipFree(m->d_res)); m->d_res = nullptr; }
  HIP_OK(hipMalloc((void**)&m->d_res, sizeof(float) * n));
  m->res_cap = n;
  return 0;
}

// Enqueues the two launches that compact the keyframe's valid-depth pixels of every level into point lists (and the
// 32-byte read-back of the per-level counts) on `s`. img supplies I1, dep the inverse depths; both pyramids have the same
// geometry. Buffers grow on demand (only the first keyframes of a run allocate).
// Level geometry of a list build + the (grow-only) allocations it needs; no launches.
static int lm_lists_layout(odo_lm* m, PointList* pl, size_t* pl_cap, int*& d_rowcnt, int& rows_cap, const odo_pyr* img,
                           const odo_pyr* dep, hipStream_t s, KfLevels* kl_out, int* rows_total_out) {
  KfLevels& kl = *kl_out;
  memset(&kl, 0, sizeof(kl));
  kl.n_levels = m->n_levels;
  int rows_total = 0;
  for (int l = 0; l < m->n_levels; l++) {
    kl.I1[l] = img->dev + img->off[l];
    kl.D1[l] = dep->dev + dep->off[l];
    kl.rows[l] = img->r[l]; kl.cols[l] = img->c[l];
    kl.row_base[l] = rows_total;
    const int ir = kl.rows[l] - 8, ic = kl.cols[l] - 8;
    rows_total += (ir > 0 && ic > 0) ? ir : 0;
    const size_t cap = (ir > 0 && ic > 0) ? (size_t)ir * ic : 0;
    if (cap > pl_cap[l]) {
      HIP_OK(hipStreamSynchronize(s));
      void* pv[] = {pl[l].a, pl[l].b, pl[l].c, pl[l].d};
      for (void* q : pv) if (q) HIP_OK(hipFree(q));
      HIP_OK(hipMalloc((void**)&pl[l].a, sizeof(float4) * cap));
      HIP_OK(hipMalloc((void**)&pl[l].b, sizeof(float4) * cap));
      HIP_OK(hipMalloc((void**)&pl[l].c, sizeof(float4) * cap));
      HIP_OK(hipMalloc((void**)&pl[l].d, sizeof(float) * cap));
      pl_cap[l] = cap;
    }
  }
  kl.row_base[m->n_levels] = rows_total;
  for (int l = m->n_levels + 1; l <= ODO_MAX_LEVELS; l++) kl.row_base[l] = rows_total;
  if (rows_total > rows_cap) {
    HIP_OK(hipStreamSynchronize(s));
    if (d_rowcnt) HIP_OK(hipFree(d_rowcnt));
    d_rowcnt = nullptr;
    HIP_OK(hipMalloc((void**)&d_rowcnt, sizeof(int) * rows_total));
    rows_cap = rows_total;
  }
  *rows_total_out = rows_total;
  return 0;
}

// hm_npts: device alias of the host-mapped per-level counts (the host reads them once the stream has passed the launches).
static int lm_enqueue_lists(odo_lm* m, PointList* pl, size_t* pl_cap, int*& d_rowcnt, int& rows_cap, int* hm_npts,
                            const odo_pyr* img, const odo_pyr* dep, hipStream_t s, int* rows_total_out) {
  KfLevels kl;
  int rows_total = 0;
  if (lm_lists_layout(m, pl, pl_cap, d_rowcnt, rows_cap, img, dep, s, &kl, &rows_total)) return -1;
  *rows_total_out = rows_total;
  if (rows_total > 0) {
    hipLaunchKernelGGL(kf_count_kernel, dim3(rows_total), dim3(256), 0, s, kl, d_rowcnt, hm_npts);
    hipLaunchKernelGGL(kf_fill_kernel, dim3(rows_total), dim3(256), 0, s, kl, m->K.f0, m->K.cx0, m->K.cy0,
                       (const int*)d_rowcnt, hm_npts, pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], pl[6], pl[7],
                       m->mode == 0 ? m->fuse_dense_max : 0);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

// Per-level counts (host copy) -> which levels run on their list (auto: when at most half of the interior has depth).
static void lm_take_counts(odo_lm* m, const int* h_npts, const odo_pyr* img, int rows_total) {
  for (int l = 0; l < ODO_MAX_LEVELS; l++) { m->npts[l] = 0; m->use_list[l] = 0; }
  if (rows_total <= 0) return;
  for (int l = 0; l < m->n_levels; l++) {
    m->npts[l] = h_npts[l];
    const long interior = (long)(img->r[l] - 8) * (img->c[l] - 8);
    // sparse levels always; small dense levels too (a few launches of the fused pipeline instead of an evaluation + an update
    // launch per LM iteration: the hand-over to the dense pipeline happens at the first level that is neither)
    m->use_list[l] = (m->mode == 2) || (interior > 0 && 2L * m->npts[l] <= interior) ||
                     (m->mode == 0 && m->npts[l] > 0 && m->npts[l] <= m->fuse_dense_max);
  }
}

// Builds (or reuses) the keyframe point lists for the pyramids of this Solve. Two launches over all levels +
// a 32-byte read-back of the per-level counts; done once per keyframe (the cache is keyed on the pyramids' build
// versions). Levels where more than half of the interior carries depth keep the dense scan.
static int lm_adopt_candidate(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, long tag, int slot);
static int lm_prepare_keyframe(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep) {
  if (m->mode == 1) { for (int l = 0; l < m->n_levels; l++) m->use_list[l] = 0; return 0; }
  if (m->kf_img_ver == kf_img->version && m->kf_dep_ver == kf_dep->version) return 0;
  if (m->ahead.active && m->ahead.img_ver == kf_img->version && m->ahead.dep_ver == kf_dep->version) {
    // built ahead for exactly these pyramids (odo_lm_candidate_begin): the launches have long finished — the event makes sure —
    // and the two list sets trade places: no launch and no read-back in front of this Solve
    m->ahead.active = 0;
    HIP_OK(hipEventSynchronize(m->ahead.ev));
    if (lm_adopt_candidate(m, kf_img, kf_dep, 1, 1) == 0) return 0;
  }
  hipStream_t s = m->ctx->stream;
  int rows_total = 0;
  if (lm_enqueue_lists(m, m->pl, m->pl_cap, m->d_rowcnt, m->rows_cap, m->hm_npts, kf_img, kf_dep, s,
                       &rows_total)) return -1;
  if (rows_total > 0) HIP_OK(hipStreamSynchronize(s));
  lm_take_counts(m, m->h_npts, kf_img, rows_total);
  m->kf_img_ver = kf_img->version;
  m->kf_dep_ver = kf_dep->version;
  return 0;
}

// Candidate lists (see odo_lm::cand_pl): enqueue on stream `s` (the tracker's depth stream); `img` holds the frame's image
// pyramid, `dep` its freshly estimated depth pyramid. The counts are valid once `s` has drained past this point.
static int lm_build_candidate(odo_lm* m, const odo_pyr* img, const odo_pyr* dep, hipStream_t s, long tag, int slot = 0) {
  LmCandSet& cs = m->cand[slot];
  cs.tag = -1;
  if (m->mode == 1) return 0;
  int rows_total = 0;
  if (lm_enqueue_lists(m, cs.pl, cs.pl_cap, cs.d_rowcnt, cs.rows_cap, cs.hm_npts, img, dep, s, &rows_total)) return -1;
  if (rows_total > 0) cs.tag = tag;
  return 0;
}

// The candidate tagged `tag` (in set `slot`) has become the keyframe (kf_img / kf_dep are its pyramids, the stream that built
// the lists has drained): trade the two list sets. Returns 0 when adopted, 1 when there was no such candidate (the next
// Solve then builds the lists itself).
static int lm_adopt_candidate(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, long tag, int slot) {
  LmCandSet& cs = m->cand[slot];
  if (m->mode == 1 || cs.tag < 0 || cs.tag != tag) return 1;
  for (int l = 0; l < ODO_MAX_LEVELS; l++) { std::swap(m->pl[l], cs.pl[l]); std::swap(m->pl_cap[l], cs.pl_cap[l]); }
  std::swap(m->d_rowcnt, cs.d_rowcnt); std::swap(m->rows_cap, cs.rows_cap);
  std::swap(m->d_npts, cs.d_npts); std::swap(m->h_npts, cs.h_npts); std::swap(m->hm_npts, cs.hm_npts);
  int rows_total = 0;
  for (int l = 0; l < m->n_levels; l++) {
    const int ir = kf_img->r[l] - 8, ic = kf_img->c[l] - 8;
    rows_total += (ir > 0 && ic > 0) ? ir : 0;
  }
  lm_take_counts(m, m->h_npts, kf_img, rows_total);
  m->kf_img_ver = kf_img->version;
  m->kf_dep_ver = kf_dep->version;
  cs.tag = -1;
  return 0;
}

static inline int lm_list_blocks(int npts);
static inline int lm_grid_for(const odo_lm* m, int level, int rows, int cols) {
  if (m->use_list[level]) return lm_list_blocks(m->npts[level]);
  return lm_grid(rows, cols);
}

// ComputeScaleNaive over the n residuals in m->d_res (ref: src/lm_optimizer.cpp:338-358). Levels a fused kernel could also take
// (<= 2 * kFineKMax virtual blocks = 16 384 points): one workgroup in the fused kernels' summation order — the same sigma bit for
// bit on every pipeline. Larger levels only ever run here (27 k-point lists of a dense pyramid's coarse levels, dense levels of up
// to 2 M residuals): lm_tdist_scale_multi_kernel on up to 128 workgroups, with the single-workgroup kernel queued behind it as its
// fall-back (a no-op unless the launch gave up). The fall-back adds in another order (strided per-thread sums and a tree) than the
// multi-workgroup launch (per thread ascending, eight waves, G workgroups): sigma after a give-up may differ in the last bits from a
// clean run — results stay within every tolerance, but "bit-identical on every pipeline" is a statement about runs without give-ups.
// odo_lm_tdist_stats counts them.
static void lm_launch_scale(odo_lm* m, int n, int level) {
  hipStream_t s = m->ctx->stream;
  if (n <= 2 * kFineKMax * kLmBlock) {
    hipLaunchKernelGGL(lm_tdist_scale_kernel, dim3(1), dim3(1024), 0, s, m->d_res, n, m->ust, level, m->d_scale, 1, (int*)nullptr);
    return;
  }
  const long per_wg = (long)kTsThreads * 8;   // eight residuals per thread until the grid is full
  long G = (n + per_wg - 1) / per_wg;
  if (G > kTsMaxWg) G = kTsMaxWg;
  if (G < 2) G = 2;
  const bool fits = (long)n <= G * kTsThreads * (long)kTsPerThread;
  const int single_order = (n <= 64 * kTdistChunksMax) ? 1 : 0;   // (what the single-workgroup kernel can hold in registers)
  if (!m->ts_multi || !fits) {
    hipLaunchKernelGGL(lm_tdist_scale_kernel, dim3(1), dim3(1024), 0, s, m->d_res, n, m->ust, level, m->d_scale, single_order, (int*)nullptr);
    return;
  }
  m->ts_multi_launches++;
  m->ts_epoch = (m->ts_epoch + 1) & 0x3fffffu;   // 22 bits beside the 10-bit pass number: cleared when it starts over
  if (m->ts_epoch == 0) {
    (void)hipMemsetAsync(m->d_ts_xbuf, 0, sizeof(unsigned long long) * kTsXbufWords, s);
    m->ts_epoch = 1;
  }
  hipLaunchKernelGGL(lm_tdist_scale_multi_kernel, dim3((unsigned)G), dim3(kTsThreads), 0, s, m->d_res, n, m->ust, level, m->d_scale,
                     m->d_ts_xbuf, m->ts_epoch, m->ts_wait, m->d_ts_gave_up, m->ts_fault);
  hipLaunchKernelGGL(lm_tdist_scale_kernel, dim3(1), dim3(1024), 0, s, m->d_res, n, m->ust, level, m->d_scale, single_order, m->d_ts_gave_up);
}

// One evaluation of the hot loop at the pose held in the device state.
static void lm_launch_eval(odo_lm* m, const LevelView& v, const LevelK& k, int level, int nblk, hipEvent_t e0 = nullptr,
                           hipEvent_t e1 = nullptr) {
  hipStream_t s = m->ctx->stream;
  if (e0 && e1 && m->robust != 2) {  // timing legs: dispatch-bound start / stop events
    if (m->use_list[level])
      hipExtLaunchKernelGGL(lm_residual_list_kernel, dim3(nblk), dim3(kLmBlock), 0, s, e0, e1, 0, m->pl[level], m->npts[level],
                            v.I2, v.rows, v.cols, k, (const LmState*)m->ust, level, m->robust, m->huber_delta,
                            (const float*)m->d_scale, m->d_partials);
    else
    launch_dense_eval(lm_dense_level(v, k, 0), m->ust, level, m->robust, m->huber_delta, m->d_scale, m->d_partials, s, e0, e1,
                      m->dense_plain_div);
    return;
  }
  if (m->use_list[level]) {
    const int n = m->npts[level];
    if (m->robust == 2) {
      hipLaunchKernelGGL(lm_residual_only_list_kernel, dim3(nblk), dim3(kLmBlock), 0, s, m->pl[level], n, v.I2, v.rows,
                         v.cols, k, m->ust, level, m->d_res);
      lm_launch_scale(m, n, level);
    }
    hipLaunchKernelGGL(lm_residual_list_kernel, dim3(nblk), dim3(kLmBlock), 0, s, m->pl[level], n, v.I2, v.rows, v.cols, k,
                       m->ust, level, m->robust, m->huber_delta, m->d_scale, m->d_partials);
    return;
  }
  if (m->robust == 2) {
    const int n = (v.rows - 8) * (v.cols - 8);
    hipLaunchKernelGGL(lm_residual_only_kernel, dim3(nblk), dim3(kLmBlock), 0, s, v, k, m->ust, level, m->d_res);
    lm_launch_scale(m, n > 0 ? n : 0, level);
  }
  launch_dense_eval(lm_dense_level(v, k, 0), m->ust, level, m->robust, m->huber_delta, m->d_scale, m->d_partials, s, nullptr,
                    nullptr, m->dense_plain_div);
}

// Algorithmic bytes of one evaluation on `level`: dense scan 12 B per interior pixel,
// point list 32 B per point (packed coordinates + inverse depth, I1, five I2 taps); + the fp64 partials written.
static inline double lm_level_bytes(const odo_lm* m, int level, int rows, int cols, int nblk) {
  const long interior = (rows > 8 && cols > 8) ? (long)(rows - 8) * (cols - 8) : 0;
  const double in = m->use_list[level] ? 32.0 * (double)m->npts[level] : 12.0 * (double)interior;
  return in + 8.0 * ODO_NACC * nblk;
}

// ---------------------------------------------------------------------------------------------------------------
// A fused Solve as a resumable job: lm_fused_begin fills the launch arguments and enqueues the coarse launch,
// lm_fused_pump issues the identical step launches (it never runs more than run_ahead launches ahead of the device) and
// odo_lm_solve collects the result. odo_lm_solve_begin lets a caller that already knows the next Solve's inputs (the
// tracker: next frame's pyramid, initial pose = this frame's result) start it before it is asked for: same launches,
// earlier. A job is tied to its pyramids' build versions and to the optimiser's initial pose: Reset or a Solve on other
// pyramids abandons it (its launches drain on the stream like the stragglers of any finished Solve: the next Solve starts
// with first_of_solve = 1 and a new token).
// ---------------------------------------------------------------------------------------------------------------
// Which residue of the launch index is sampled in the next Solve: a hash of the Solve counter, NOT the counter modulo N — the
// tracker promotes a keyframe every ~8 frames on bench.py's drive and the Solve after a switch is the long one, so "every 8th
// Solve" aliased with the keyframe rhythm (coarse launch sampled at 193 or 220 us depending on the warm-up length, population
// mean 203).
static inline int lm_ev_next_phase(odo_lm* m) {
  const unsigned long long h = (unsigned long long)(m->ev_solves++) * 0x9E3779B97F4A7C15ull;
  return (int)((h >> 40) % (unsigned long long)m->ev_on);
}
// Launch timing: is launch `i` of the Solve in flight one of the sampled ones? If so it gets the next free span slot.
constexpr int kSpanSlots = 16384;
static inline unsigned long long* lm_span_slot(odo_lm* m, int i, bool coarse) {
  // launch i and its successor i + 1 are sampled together: the distance of their starts is the launch PERIOD — execution plus
  // the dependent-kernel boundary, what an evaluation costs the serial chain — next to the execution span of each
  const bool hit = m->ev_on > 0 && m->d_span && (((i + m->ev_phase) % m->ev_on) == 0 || (i > 0 && ((i - 1 + m->ev_phase) % m->ev_on) == 0));
  if (!hit || m->span_used >= kSpanSlots) return nullptr;
  m->span_kind->push_back(odo_lm::SpanTag{(unsigned char)(coarse ? 1 : 0), m->ev_solves, i});
  return m->d_span + 2 * (size_t)(m->span_used++);
}
// Reads the spans of every sampled launch so far into the accumulators and frees the slots (drains the stream first).
static int lm_span_collect(odo_lm* m) {
  if (!m->d_span || m->span_used == 0) return 0;
  HIP_OK(hipSetDevice(m->ctx->device));
  HIP_OK(hipStreamSynchronize(m->ctx->stream));
  std::vector<unsigned long long> h(2 * (size_t)m->span_used);
  HIP_OK(hipMemcpy(h.data(), m->d_span, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  for (int i = 0; i < m->span_used; i++) {
    const unsigned long long t0 = h[2 * i], t1 = h[2 * i + 1];
    if (t0 == ~0ull || t1 == 0 || t1 < t0) continue;   // a launch that never ran (dropped with its Solve)
    const double us = (double)(t1 - t0) * 0.01;          // wall_clock64: 100 MHz
    const odo_lm::SpanTag& tg = (*m->span_kind)[i];
    m->ev_total_us += us;
    m->ev_sampled++;
    if (tg.coarse) { m->ev_coarse_us += us; m->ev_coarse_launches++; }
    if (i + 1 < m->span_used) {   // its successor was sampled too: start-to-start period
      const odo_lm::SpanTag& nx = (*m->span_kind)[i + 1];
      const unsigned long long n0 = h[2 * (i + 1)];
      if (nx.solve == tg.solve && nx.idx == tg.idx + 1 && n0 != ~0ull && n0 > t0) {
        const double per = (double)(n0 - t0) * 0.01;
        if (tg.coarse) { m->ev_cperiod_us += per; m->ev_cperiod_n++; } else { m->ev_period_us += per; m->ev_period_n++; }
      }
    }
  }
  for (int i = 0; i < m->span_used; i++) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
  HIP_OK(hipMemcpy(m->d_span, h.data(), sizeof(unsigned long long) * h.size(), hipMemcpyHostToDevice));
  m->span_used = 0;
  m->span_kind->clear();
  return 0;
}
// Diagnostic build (-DODO_DIAG) + ODO_TRACK_DEBUG: host-side laps of the relaunch path (result seen -> next Solve's launches issued),
// printed by odo_tracker_track. Compiled out of the product.
#ifdef ODO_DIAG
static double g_lap_us[12]; static long g_lap_n; static std::chrono::steady_clock::time_point g_lap_t;
static inline void lap_start() { g_lap_t = std::chrono::steady_clock::now(); }
static inline void lap(int i) { const auto n = std::chrono::steady_clock::now(); g_lap_us[i] += std::chrono::duration<double, std::micro>(n - g_lap_t).count(); g_lap_t = n; }
#else
static inline void lap_start() {}
static inline void lap(int) {}
#endif
static inline int lm_job_progress(const odo_lm* m) {
  const int v = ((volatile int*)m->h_prog)[0];
  return ((v >> kProgSeqBits) == m->job.token) ? (v & ((1 << kProgSeqBits) - 1)) : 0;
}
static inline bool lm_job_finished(const odo_lm* m) { return ((volatile int*)m->h_prog)[1] == m->job.token; }

// Every pose-LM persistent launch of the process has its own number (g_lm_fine_dispatch in kernels.hip.h: "half dispatched" = the first
// and the last block of a grid have written different numbers; the per-optimiser epochs below may coincide across optimisers).
static unsigned lm_fine_next_dispatch() { static std::atomic<unsigned> n{0}; return ++n; }
// The exchange buffer's tags are (epoch << 8) + evaluation. An epoch is used for one launch; when the 24 bits are exhausted the
// buffer is cleared on the stream (tag 0 = never valid) before they start over, so a granule left by an old launch — a row beyond
// the levels of every Solve since — can never carry the tag a new launch waits for.
static unsigned lm_fine_next_epoch(odo_lm* m) {
  if (++m->fine_epoch >= (1u << 24)) {
    (void)hipMemsetAsync(m->d_xbuf, 0, sizeof(unsigned long long) * kFineXbufWords, m->ctx->stream);
    m->fine_epoch = 1;
  }
  // the scale area's tags (t-distribution passes) carry the low 12 bits of the epoch next to the evaluation and pass numbers:
  // cleared whenever those start over
  if ((m->fine_epoch & 0xfffu) == 0u)
    (void)hipMemsetAsync(m->d_xbuf + kFineScaleOff, 0, sizeof(unsigned long long) * kScaleWords, m->ctx->stream);
  return m->fine_epoch;
}

constexpr int kLmListMaxBlocks = 160;
static inline int lm_list_blocks(int npts) {
  long g = ((long)npts + kLmBlock - 1) / kLmBlock;
  if (g > kLmListMaxBlocks) g = kLmListMaxBlocks;
  if (g < 1) g = 1;
  return (int)g;
}
// Which kernel takes which point-list level of a fused Solve (levels stop .. n_levels - 1, coarse to fine):
//   levels >= *min_level          the coarse launch (one workgroup: few points),
//   levels [*fine_lo, *min_level) the persistent launch (fine_k workgroups; 0: there is none),
//   levels [stop, *fine_lo)       a step launch per evaluation.
// With the persistent launch behind it the coarse launch keeps only the levels that fit ONE round of its workgroup (512 points):
// a level of two rounds costs less as four virtual blocks on four CUs. Every kernel sums a level in the same order, so where a
// level runs does not show in the result. The persistent launch pays when its workgroups can hold a level's points in registers
// (<= 2 virtual blocks each); a level of more blocks (the point-list levels of a 1080p stream: 115 and 160) is better spread over
// 160 CUs by the step launches. t-distribution weights (robust == 2): the coarse launch takes every level it can hold (1 024 points).
static void lm_plan_levels(const odo_lm* m, int stop, int fine_k, int* min_level_out, int* fine_lo_out) {
  static const int coarse_env = getenv("ODO_COARSE_MAX") ? atoi(getenv("ODO_COARSE_MAX")) : -1;
  // A level of up to 2 K virtual blocks stays in the workgroups' registers (one pass per evaluation). Levels of up to 4 K — a
  // keyframe near the reference's selection cap: 28 k points = 110 virtual blocks on level 0 — take TWO passes per evaluation with
  // their points re-read: 3 930 frames/s on bench.py's saturated drive against 3 860 with that level on step launches behind the
  // launch (round 3 measured the opposite on another level shape; ODO_LM_FINE_PASSES=1 restores it, 3 allows three passes).
  // t-distribution weights: one pass only (the scale iteration needs every residual of the level in registers).
  const int fine_passes = (m->robust == 2 || m->fine_passes < 1) ? 1 : m->fine_passes;
  auto fits = [&](int l) {
    const int nblk = lm_list_blocks(m->npts[l]);
    return nblk <= 2 * fine_k * fine_passes && nblk <= kFineRowsMax && m->npts[l] <= nblk * kLmBlock;
  };
  int min_level = m->n_levels, fine_lo = m->n_levels;
  for (int pass = 0; pass < 2; pass++) {
    const bool want_fine = fine_k > 0 && pass == 0;
    int coarse_max = coarse_env >= 0 ? (coarse_env < kCoarseMaxPoints ? coarse_env : kCoarseMaxPoints)
                                     : (want_fine ? kCoarseBlock : kCoarseMaxPoints);
    // (t-distribution weights: every scale pass is a reduction over the level — one barrier inside the coarse launch's workgroup,
    //  a trip through L2 between the persistent launch's: the coarse launch takes what it can hold, both of its rounds)
    if (m->robust == 2 && coarse_env < 0) coarse_max = kCoarseMaxPoints;
    min_level = m->n_levels;
    while (min_level > stop && m->npts[min_level - 1] <= coarse_max) min_level--;
    if (!m->coarse) min_level = m->n_levels;
    fine_lo = min_level;
    if (!want_fine) break;
    while (fine_lo > stop && fits(fine_lo - 1)) fine_lo--;
    if (fine_lo < min_level) break;   // it takes at least the level under the coarse launch's; else the coarse launch's full reach
  }
  *min_level_out = min_level;
  *fine_lo_out = fine_lo;
}

// Workgroups the persistent launch of levels [fine_lo, min_level) is issued with. The plan above is made for fine_k (32: every CU of
// an XCD); the launch leaves TWO of those CUs free when that changes nothing about the plan — every level needs as many passes with 30
// workgroups as with 32 (levels of 61-64 and 121-128 virtual blocks are the exceptions). A block of another persistent launch that is
// dealt to this XCD only to return at once then finds a CU instead of waiting for the whole Solve: the depth launch's give-ups halve
// (14 -> 6-7 in 40 000 frames at the same frame rate). Sums are the same bit for bit: the fold order is
// per virtual block, not per workgroup. ODO_LM_FINE_K pins the number.
static int lm_fine_launch_k(const odo_lm* m, int fine_lo, int min_level) {
  static const bool pinned = getenv("ODO_LM_FINE_K") != nullptr;
  const int k = m->fine_k, spare = 2;
  if (pinned || k <= 16) return k;
  for (int l = fine_lo; l < min_level; l++) {
    const int nblk = lm_list_blocks(m->npts[l]);
    if ((nblk + 2 * k - 1) / (2 * k) != (nblk + 2 * (k - spare) - 1) / (2 * (k - spare))) return k;
  }
  return k - spare;
}

// Lowest level of the coarse-to-fine run of point-list levels the fused pipeline can take (it starts at the coarsest level):
// 0 = the whole Solve, n_levels = nothing (the Solve runs on the unfused pipeline from the start). fine_k: workgroups of the
// persistent launch this Solve may use.
// t-distribution weights: every evaluation needs the scale of ALL its residuals before any weight — a fixed-point iteration, one
// reduction over the level per pass (ref: src/lm_optimizer.cpp:257-261,338-358). The coarse launch (LDS) and the persistent launch
// (L2 exchange) iterate it in place; a step launch per evaluation cannot (no grid-wide reduction inside a launch), so the fused part
// of such a Solve ends above the first level neither of the two takes and the unfused pipeline carries on from there.
static int lm_fused_stop_level(const odo_lm* m, int fine_k) {
  if (!m->fused) return m->n_levels;
  int stop = m->n_levels;
  while (stop > 0 && m->use_list[stop - 1]) stop--;
  if (m->robust == 2) {
    int min_level = 0, fine_lo = 0;
    lm_plan_levels(m, stop, fine_k, &min_level, &fine_lo);
    stop = fine_lo;
    long budget = 0;
    for (int l = stop; l < m->n_levels; l++) budget += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
    if (budget > kTdistMaxEvals) stop = m->n_levels;   // (the scale tags count evaluations in 10 bits)
  }
  return stop;
}
static int lm_fused_stop_level(const odo_lm* m) { return lm_fused_stop_level(m, m->fine_k); }
static bool lm_fused_eligible(const odo_lm* m) { return lm_fused_stop_level(m) < m->n_levels; }

// Give-up policy of the persistent launches (single and batched). A give-up costs one bounded wait (4 ms) + a redo of the Solve on
// the step launches. Three give-ups switch the launch off; it is tried again after kFineRetryFirst Solves, and every further
// switch-off doubles that interval (up to kFineRetryCap): on a GPU that is shared for good the hiccup becomes rarer and rarer
// instead of recurring every second. kFineForgive clean Solves with the launch on forget everything.
constexpr int kFineRetryFirst = 4096, kFineRetryCap = 1 << 20, kFineForgive = 1024;
static inline int fine_retry_after(int offs) {
  long v = kFineRetryFirst;
  for (int i = 1; i < offs && v < kFineRetryCap; i++) v *= 2;
  return (int)(v < kFineRetryCap ? v : kFineRetryCap);
}
// One give-up: returns true when the launch is (now) switched off.
static inline bool fine_note_giveup(int* strikes, int* clean, int* offs) {
  (*strikes)++;
  *clean = 0;
  if (*strikes >= 3) { (*offs)++; return true; }
  return false;
}
// One Solve that did not give up. on: the launch was used. Returns true when a switched-off launch is due for another try
// (the caller switches it on; one more give-up switches it off again with a doubled interval).
static inline bool fine_note_clean(bool on, int* strikes, int* clean, int* offs) {
  (*clean)++;
  if (on && *strikes < 3 && *clean >= kFineForgive) { *strikes = 0; *clean = 0; *offs = 0; }
  if (*strikes >= 3 && *clean >= fine_retry_after(*offs)) { *strikes = 2; *clean = 0; return true; }
  return false;
}

// Keyframe lists must be current (lm_prepare_keyframe) and the Solve fused-eligible.
// chain_in_token != 0: a chained Solve (lm_chain_begin) behind the Solve with that token: only if the coarse + persistent launches
// cover all of it (returns 1 without launching anything otherwise), lambda0 = chain_lambda (what Reset will set).
// The persistent launch of a job whose coarse launch is out: levels [jb.fine_lo, jb.min_level) in ONE launch behind it; if they are the
// rest of the Solve it reports the result itself.
static void lm_job_launch_fine(odo_lm* m, LmJob& jb) {
  StepArgs& a = jb.a;
  const int fine_lo = jb.fine_lo, min_level = jb.min_level, stop = jb.stop_level;
  LmState* st[2] = {m->d_state, m->d_state + 1};
  double* part[2] = {m->d_partials, m->d_partials + (size_t)kLmMaxBlocks * ODO_NACC};
  int fine_budget = 0;
  for (int l = fine_lo; l < min_level; l++) fine_budget += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
  if (!(fine_lo < min_level && fine_budget > 0)) return;
  jb.budget -= fine_budget;
  a.st_in = st[jb.seq & 1]; a.st_out = st[(jb.seq + 1) & 1];
  a.part_in = part[jb.seq & 1]; a.part_out = part[(jb.seq + 1) & 1];
  a.seq = jb.seq; a.first_of_solve = (jb.seq == 0) ? 1 : 0;
  a.arm = nullptr;
  a.span = lm_span_slot(m, jb.launches, false);
  a.fine_epoch = lm_fine_next_epoch(m);
  a.fine_dispatch = lm_fine_next_dispatch();
  a.fine_wait = m->fine_wait;
  a.fine_home = m->fine_home;
  const int k_use = lm_fine_launch_k(m, fine_lo, min_level);
  m->fine_k_last = k_use;
  unsigned* const dispatch_words = m->dispatch_words;   // this unit's g_lm_fine_dispatch: what the depth launches read
  // the lean build has neither the trace writes nor the bilinear sampling path
  launch_lm_fine(m->robust == 2 ? 2 : (a.trace || m->bilinear) ? 1 : 0, 8 * k_use, m->ctx->stream, a, k_use, m->d_xbuf, m->fine_fault, fine_lo, dispatch_words);
  jb.seq++;
  jb.launches++;
  if (fine_lo <= stop) {   // nothing left for step launches
    jb.issued_all = true;
    jb.result_by_launch = true;
  }
}
static int lm_fused_begin_job(odo_lm* m, LmJob& jb, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img,
                              int chain_in_token, float chain_lambda, bool arm = false);
static void lm_arm_abort(odo_lm* m);
static int lm_fused_begin(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img) {
  m->chained.active = 0;   // (a chained Solve belongs to the job it was queued behind)
  if (m->armed.active) lm_arm_abort(m);   // (an armed Solve the caller never resolved — a redo of the Solve in flight —: it returns at once)
  return lm_fused_begin_job(m, m->job, kf_img, kf_dep, cur_img, 0, 0.0f);
}
static int lm_fused_begin_job(odo_lm* m, LmJob& jb, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img,
                              int chain_in_token, float chain_lambda, bool arm) {
  hipStream_t s = m->ctx->stream;
  if (chain_in_token || arm) {   // feasibility first: nothing is launched for a Solve that would need step launches or the unfused pipeline
    const int stop0 = lm_fused_stop_level(m);
    int ml = 0, fl = 0;
    lm_plan_levels(m, stop0, m->fine_k, &ml, &fl);
    if (stop0 != 0 || !((fl <= stop0 && fl < ml) || (ml <= stop0 && ml < m->n_levels))) return 1;
    // an armed Solve: a coarse launch that reads the host's word + the persistent launch for the rest, lean kernels only
    if (arm && !(ml < m->n_levels && fl <= stop0 && fl < ml && m->robust != 2 && !m->record && !m->bilinear)) return 1;
  }
  jb.kf_img = kf_img; jb.kf_dep = kf_dep; jb.cur_img = cur_img;
  jb.kf_img_ver = kf_img->version; jb.kf_dep_ver = kf_dep->version; jb.cur_ver = cur_img->version;
  jb.seq = 0; jb.launches = 0; jb.it = 0; jb.issued_all = false; jb.result_by_launch = false;
  jb.poll = m->poll != 0;
  for (int l = 0; l < ODO_MAX_LEVELS; l++) jb.bytes_per_level[l] = 0.0;
  // Every Solve has a token. The fused launches tag their progress words with it — launches of the previous Solve may
  // still be draining on the stream when this one starts (they are no-ops, but they do report) — and the launch that
  // finishes the Solve writes the result and the token into host-mapped memory itself.
  m->token = (m->token % 0x3ffff) + 1;
  jb.token = m->token;
  jb.slot = (m->slot_next++) & 1;
  if (m->ev_on > 0 && !chain_in_token) m->ev_phase = lm_ev_next_phase(m);
  StepArgs& a = jb.a;
  memset(&a, 0, sizeof(a));
  a.n_levels = m->n_levels;
  const int stop = lm_fused_stop_level(m);   // levels below it are handed over to the unfused pipeline
  jb.stop_level = stop;
  int grid = 1, budget = 0;
  for (int l = stop; l < m->n_levels; l++) {
    StepLevel& L = a.lv[l];
    L.pl = m->pl[l]; L.n = m->npts[l];
    L.rows = kf_img->r[l]; L.cols = kf_img->c[l];
    L.nblk = lm_grid_for(m, l, L.rows, L.cols);
    L.I2 = cur_img->dev + cur_img->off[l];
    L.k = lm_level_k(m, l);
    L.max_iters = m->max_iters[l];
    if (L.nblk + 1 > grid) grid = L.nblk + 1;  // + the publisher block
    budget += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
    jb.bytes_per_level[l] = lm_level_bytes(m, l, L.rows, L.cols, L.nblk);
  }
  a.lambda0 = (chain_in_token || arm) ? chain_lambda : m->lambda; a.precision = m->precision; a.robust = m->robust; a.huber_delta = m->huber_delta;
  a.trace = m->record ? m->d_trace : nullptr; a.cost_stat = m->record ? m->d_cost : nullptr; a.host_prog = m->d_prog;
  a.out = m->d_res_map + 48 * jb.slot; a.done_flag = m->d_done + jb.slot; a.token = jb.token;
  a.stop_level = stop;
  a.final_state = stop > 0 ? m->d_state + 2 : nullptr;
  memcpy(a.init, m->init, sizeof(a.init));
  a.chain_pose = m->d_chain_pose; a.chain_guard = m->d_chain_guard;
  a.chain_in_token = chain_in_token;
  a.chain_out = m->kf_on;
  memcpy(a.kf_rule, m->kf_rule, sizeof(a.kf_rule));
    static unsigned long long* dbg_buf = [] {
      unsigned long long* p = nullptr;
      if (getenv("ODO_COARSE_STAMPS")) {
        if (!ODO_PHASE_STAMPS) fprintf(stderr, "odometry_hip: ODO_COARSE_STAMPS needs the diagnostic build (python -m odometry_amd.build --stamps; "
                                               "ODOMETRY_HIP_LIB=.../libodometry_hip_stamps.so): this library has its phase stamps compiled out\n");
        else if (hipHostMalloc((void**)&p, 2048, hipHostMallocMapped) == hipSuccess) memset(p, 0, 2048);
      }
      return p;
    }();
    a.dbg = dbg_buf;
    if (dbg_buf) {  // timeline of the previous Solve: device wall clock (100 MHz) at the start / end of every launch
      static double gap_us = 0.0, dur_us = 0.0, span_us = 0.0; static long n_gap = 0, n_solve = 0;
      int n = 0;
      while (n < 56 && dbg_buf[16 + 2 * n] && dbg_buf[16 + 2 * n + 1]) n++;
      if (n > 1) {
        for (int i = 0; i + 1 < n; i++) { gap_us += (double)(long long)(dbg_buf[16 + 2 * (i + 1)] - dbg_buf[16 + 2 * i + 1]) * 0.01; n_gap++; }
        for (int i = 1; i < n; i++) dur_us += (double)(long long)(dbg_buf[16 + 2 * i + 1] - dbg_buf[16 + 2 * i]) * 0.01;
        span_us += (double)(long long)(dbg_buf[16 + 2 * (n - 1) + 1] - dbg_buf[16]) * 0.01;
        if (++n_solve % 100 == 0)
          fprintf(stderr, "[lm timeline] launches/Solve %.1f, publisher-block start->end %.2f us per step launch, end->next start "
                  "%.2f us, first start -> last end %.1f us per Solve\n", (double)(n_gap + n_solve) / n_solve,
                  dur_us / (n_gap ? n_gap : 1), gap_us / (n_gap ? n_gap : 1), span_us / n_solve);
      }
      memset(dbg_buf + 16, 0, sizeof(unsigned long long) * 112);
    }
    if (dbg_buf && dbg_buf[5] > 0 && dbg_buf[5] % 100 == 0)
      fprintf(stderr, "[coarse stamps] per iteration: eval %.0f reduce %.0f state-machine %.0f cycles; iterations/launch %.2f, "
              "cycles/launch %.0f\n", (double)dbg_buf[0] / dbg_buf[3], (double)dbg_buf[1] / dbg_buf[3],
              (double)dbg_buf[2] / dbg_buf[3], (double)dbg_buf[3] / dbg_buf[5], (double)dbg_buf[4] / dbg_buf[5]);
    if (dbg_buf && dbg_buf[132] > 0 && dbg_buf[132] % 100 == 0)
      fprintf(stderr, "[fine stamps] per evaluation: eval + publish %.0f gather + fold %.0f state-machine %.0f cycles; evaluations/launch "
              "%.2f, same-XCD launches %.0f %%\n", (double)dbg_buf[128] / dbg_buf[131], (double)dbg_buf[129] / dbg_buf[131],
              (double)dbg_buf[130] / dbg_buf[131], (double)dbg_buf[131] / dbg_buf[132], 100.0 * (double)dbg_buf[133] / dbg_buf[132]);
#if ODO_PHASE_STAMPS
    if (dbg_buf && dbg_buf[5] > 0 && dbg_buf[5] % 500 == 0) {
      unsigned long long g[24];
      lm_chain_diag_read(g);
      if (g[8] && g[13])
        fprintf(stderr, "[wall clock outside the loops] fine loop end -> next coarse entry %.2f us; coarse: previous fine exit -> entry %.2f us, prologue %.2f, loop %.2f, epilogue %.2f; fine: coarse exit -> "
                "entry %.2f us, prologue %.2f, loop %.2f, epilogue %.2f\n", 0.01 * g[3] / (g[14] ? g[14] : 1), 0.01 * g[7] / (g[14] ? g[14] : 1), 0.01 * g[4] / g[8], 0.01 * g[5] / g[8], 0.01 * g[6] / g[8],
                0.01 * g[12] / (g[15] ? g[15] : 1), 0.01 * g[9] / g[13], 0.01 * g[10] / g[13], 0.01 * g[11] / g[13]);
      if (g[20]) fprintf(stderr, "[fine launch] eval + publish of a level's FIRST evaluation: %.0f cycles (%.1f per launch)\n", (double)g[19] / g[20], (double)g[20] / g[13]);
      if (g[8]) fprintf(stderr, "[coarse prologue] entry -> level table in LDS %.2f us, lm_fused_prologue %.2f, hot state %.2f\n", 0.01 * g[16] / g[8], 0.01 * g[17] / g[8], 0.01 * g[18] / g[8]);
    }
#endif
    if (dbg_buf && dbg_buf[5] > 0 && dbg_buf[5] % 100 == 0 && dbg_buf[6] > 0 && dbg_buf[135] > 0)
      fprintf(stderr, "[shader clock] inside the coarse launch %.0f MHz, inside the fine launch %.0f MHz (cycle counter / 100 MHz wall clock)\n",
              (double)dbg_buf[4] / (double)dbg_buf[6] * 100.0, (double)dbg_buf[134] / (double)dbg_buf[135] * 100.0);
    if (dbg_buf && dbg_buf[5] > 0 && dbg_buf[5] % 100 == 0 && dbg_buf[11] > 0)
      fprintf(stderr, "[state machine] coarse launch: decide %.0f solve %.0f exp/compose %.0f cycles per evaluation\n",
              (double)dbg_buf[8] / dbg_buf[11], (double)dbg_buf[9] / dbg_buf[11], (double)dbg_buf[10] / dbg_buf[11]);
    if (dbg_buf && dbg_buf[132] > 0 && dbg_buf[132] % 100 == 0 && dbg_buf[139] > 0)
      fprintf(stderr, "[state machine] fine launch (publisher workgroup): decide %.0f solve %.0f exp/compose %.0f cycles per evaluation\n",
              (double)dbg_buf[136] / dbg_buf[139], (double)dbg_buf[137] / dbg_buf[139], (double)dbg_buf[138] / dbg_buf[139]);
  LmState* st[2] = {m->d_state, m->d_state + 1};
  double* part[2] = {m->d_partials, m->d_partials + (size_t)kLmMaxBlocks * ODO_NACC};
  // which kernel takes which level (lm_plan_levels): the coarse launch levels >= min_level, the persistent launch
  // [fine_lo, min_level), step launches what is left above `stop` (t-distribution weights: nothing — stop == fine_lo)
  int min_level = m->n_levels, fine_lo = m->n_levels;
  lm_plan_levels(m, stop, m->fine_k, &min_level, &fine_lo);
  m->last_coarse = (min_level < m->n_levels) ? 1 : 0;
  // The host-mapped progress words are what the step launches' pump reads. A Solve that is all coarse + persistent launch has no
  // pump: its launches skip the two system-scope stores (one a release) at their exits — the launch behind them starts that much earlier.
  {
    int fb = 0;
    for (int l = fine_lo; l < min_level; l++) fb += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
    const bool fine_covers = fine_lo < min_level && fb > 0 && fine_lo <= stop;
    const bool coarse_covers = min_level <= stop && min_level < m->n_levels;
    if (fine_covers || coarse_covers) a.host_prog = nullptr;
  }
  if (min_level < m->n_levels) {
    int coarse_budget = 0;
    for (int l = min_level; l < m->n_levels; l++) coarse_budget += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
    budget -= coarse_budget;
    a.st_in = st[jb.seq & 1]; a.st_out = st[(jb.seq + 1) & 1];
    a.part_in = part[jb.seq & 1]; a.part_out = part[(jb.seq + 1) & 1];
    a.seq = jb.seq; a.first_of_solve = 1;
    a.span = lm_span_slot(m, 0, true);
    // the trackers' optimisers (Huber / L2, nothing recorded) run the build without the scale passes and the trace writes
    lap(6);
    if (arm) {
      // (ODO_ARM_WAIT_US: the bound of the armed launch's wait for its word, default 2 s — StepArgs::fine_wait carries it, in ticks)
      a.arm = m->d_arm_map; a.fine_wait = m->arm_wait;
      launch_lm_coarse_armed(s, a, min_level);
    }
    else launch_lm_coarse(m->robust != 2 && !a.trace && !m->bilinear, s, a, min_level);
    lap(7);
    jb.seq++;
    jb.launches++;
  }
  jb.grid = grid; jb.budget = budget; jb.min_level = min_level; jb.fine_lo = fine_lo;
  if (!arm) lm_job_launch_fine(m, jb);   // (an armed Solve's persistent launch follows the host's word: lm_arm_go)
  if (min_level <= stop && min_level < m->n_levels) {   // the coarse launch covers the whole fused part: it reports the result itself
    jb.issued_all = true;
    jb.result_by_launch = true;
  }
  jb.active = 1;
  return 0;
}

// Issues step launches. block = false: returns as soon as the next launch would have to wait for the device.
// `budget` launches evaluate at most `budget` times; while the host is polling, one more launch consumes the last
// evaluation and reports the result, so no separate finalize launch is needed.
static void lm_fused_pump(odo_lm* m, bool block) {
  LmJob& jb = m->job;
  if (!jb.active || jb.issued_all) return;
  hipStream_t s = m->ctx->stream;
  StepArgs& a = jb.a;
  LmState* st[2] = {m->d_state, m->d_state + 1};
  double* part[2] = {m->d_partials, m->d_partials + (size_t)kLmMaxBlocks * ODO_NACC};
  while (jb.it < jb.budget + (jb.poll ? 1 : 0)) {
    if (jb.poll) {
      const auto t0 = std::chrono::steady_clock::now();
      while (jb.seq - lm_job_progress(m) > m->run_ahead && !lm_job_finished(m)) {
        if (!block) return;
        if (m->idle_pump) m->idle_pump(m->idle_arg);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { jb.poll = false; break; }  // never hang
      }
      if (lm_job_finished(m)) { jb.result_by_launch = true; break; }  // every level has finished on the device
      if (!jb.poll && jb.it >= jb.budget) break;
    }
    a.st_in = st[jb.seq & 1]; a.st_out = st[(jb.seq + 1) & 1];
    a.part_in = part[jb.seq & 1]; a.part_out = part[(jb.seq + 1) & 1];
    a.seq = jb.seq; a.first_of_solve = (jb.seq == 0) ? 1 : 0;
    a.span = lm_span_slot(m, jb.launches, false);   // sampled launches record their own execution span (bench.py roofline)
    hipLaunchKernelGGL(lm_step_kernel, dim3(jb.grid), dim3(kLmBlock), 0, s, a);
    jb.seq++;
    jb.launches++;
    if (jb.it == jb.budget) jb.result_by_launch = true;  // the extra launch consumes the last evaluation and reports
    jb.it++;
  }
  jb.issued_all = true;
}

// Levels l_hi .. 0 of the unfused pipeline (one evaluation launch + one lm_update_kernel launch per LM iteration, the host at
// most run_ahead iterations ahead, early exit through the host-mapped progress words) on the state m->ust; the progress words
// are h_prog[m->upo ...]. Used for a whole Solve (t-distribution weights, dense levels from the top) and for the fine dense
// levels below the fused pipeline's hand-over.
static int lm_unfused_levels(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, int l_hi,
                             int* launches_io, double bytes_per_level[ODO_MAX_LEVELS]) {
  hipStream_t s = m->ctx->stream;
  volatile int* prog = m->h_prog + m->upo;
  for (int i = 0; i < 16; i++) m->h_prog[m->upo + i] = 0;  // nothing of the unfused pipeline is draining: its Solves end with a stream sync
  bool poll = m->poll != 0;
  int seq = 0;
  for (int l = l_hi; l >= 0; l--) {  // ref: src/lm_optimizer.cpp:92
    LevelView v;
    v.I1 = kf_img->dev + kf_img->off[l];
    v.I2 = cur_img->dev + cur_img->off[l];
    v.D1 = kf_dep->dev + kf_dep->off[l];
    v.rows = kf_img->r[l]; v.cols = kf_img->c[l];
    const LevelK k = lm_level_k(m, l);
    const int nblk = lm_grid_for(m, l, v.rows, v.cols);
    if (m->robust == 2 && lm_ensure_res(m, (size_t)v.rows * v.cols)) return -1;
    bytes_per_level[l] = lm_level_bytes(m, l, v.rows, v.cols, nblk);
    hipLaunchKernelGGL(lm_begin_level_kernel, dim3(1), dim3(64), 0, s, m->ust, l, m->lambda, m->max_iters[l]);
    for (int it = 0; it < m->max_iters[l]; it++) {  // ref: :117
      if (poll) {
        const auto t0 = std::chrono::steady_clock::now();
        while (seq - prog[0] > m->run_ahead && !prog[2 + l]) {
          if (m->idle_pump) m->idle_pump(m->idle_arg);
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { poll = false; break; }  // never hang
        }
        if (prog[2 + l]) break;  // the level's loop has stopped on the device
      }
      seq++;
      lm_launch_eval(m, v, k, l, nblk);
      hipLaunchKernelGGL(lm_update_kernel<false>, dim3(1), dim3(kUpdThreads), 0, s, m->ust, m->d_partials, nblk, l,
                         m->precision, m->max_iters[l], m->record ? m->d_trace : nullptr, m->d_cost, m->d_prog + m->upo, seq,
                         (unsigned long long*)nullptr);
      (*launches_io)++;
    }
  }
  return 0;
}

static bool lm_job_matches(const odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img) {
  const LmJob& jb = m->job;
  return jb.active && jb.kf_img == kf_img && jb.kf_dep == kf_dep && jb.cur_img == cur_img && jb.kf_img_ver == kf_img->version &&
         jb.kf_dep_ver == kf_dep->version && jb.cur_ver == cur_img->version;
}

extern "C" int odo_lm_set_idle_callback(odo_lm* m, void (*fn)(void*), void* arg) {
  if (!m) return fail("odo_lm_set_idle_callback: NULL optimiser");
  m->idle_pump = fn; m->idle_arg = arg;
  return 0;
}

// Starts the Solve that a following odo_lm_solve(lm, kf_img, kf_dep, cur_img) will collect. Returns 0 when started (or already
// running), 1 when this Solve does not run on the fused pipeline (nothing started: odo_lm_solve does all of it), -1 on error.
// set_device = false: the caller has made the optimiser's device current on this thread already (odo_tracker_track: the call sits between
// one Solve's result and the next Solve's first launch, where 0.3-0.5 us of hipSetDevice is 0.3-0.5 us of every frame)
static int lm_solve_begin(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, bool set_device);
extern "C" int odo_lm_solve_begin(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img) {
  return lm_solve_begin(m, kf_img, kf_dep, cur_img, true);
}
static int lm_solve_begin(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, bool set_device) {
  if (lm_check_pyrs(m, kf_img, kf_dep, cur_img)) return -1;
  if (set_device) HIP_OK(hipSetDevice(m->ctx->device));
  lap(4);
  if (lm_job_matches(m, kf_img, kf_dep, cur_img)) return 0;
  m->job.active = 0;
  if (lm_prepare_keyframe(m, kf_img, kf_dep)) return -1;
  if (!lm_fused_eligible(m)) return 1;
  lap(5);
  if (lm_fused_begin(m, kf_img, kf_dep, cur_img)) return -1;
  lap(8);
  lm_fused_pump(m, false);
  HIP_OK(hipGetLastError());
  lap(9);
  return 0;
}

// Keyframe-candidate point lists built AHEAD (the lists are what ComputeResidualJacobianNaive's validity test selects, ref:
// src/lm_optimizer.cpp:190-198): `img` / `dep` are the pyramids of a frame that may become the keyframe of later Solves. The two
// list-building launches go to `side`'s stream, behind `mark` of the optimiser's stream (0: behind everything queued there), and
// the call returns at once. A later Solve whose keyframe pyramids are exactly these (same builds) adopts the lists by a buffer swap
// instead of building them, with a read-back of the counts, in front of its first launch; any other Solve ignores them. One
// candidate at a time (a new call replaces the previous one); `side` must be the same context for every call on this optimiser.
// Not for an optimiser owned by odo_tracker / odo_tracker_batch (they keep their own candidates).
extern "C" int odo_lm_candidate_begin(odo_lm* m, odo_ctx* side, const odo_pyr* img, const odo_pyr* dep, unsigned long mark) {
  if (!m || !side || !img || !dep) return fail("odo_lm_candidate_begin: NULL arg");
  if (side->device != m->ctx->device) return fail("odo_lm_candidate_begin: the two contexts must be on one device");
  if (img->levels != dep->levels || img->levels < m->n_levels) return fail("odo_lm_candidate_begin: pyramids do not fit the optimiser");
  HIP_OK(hipSetDevice(m->ctx->device));
  m->ahead.active = 0;
  if (mark ? odo_ctx_stream_wait_mark(side, m->ctx, mark) : odo_ctx_stream_wait(side, m->ctx)) return -1;
  if (lm_build_candidate(m, img, dep, side->stream, 1, 1)) return -1;
  if (m->cand[1].tag != 1) return 0;   // (dense-scan mode, or nothing to list)
  HIP_OK(hipEventRecord(m->ahead.ev, side->stream));
  m->ahead.img_ver = img->version; m->ahead.dep_ver = dep->version;
  m->ahead.active = 1;
  return 0;
}

// ---- chained Solves (used by the trackers) -----------------------------------------------------------------------------
// The keyframe test the guard of a chained Solve applies (ref: run_odometry_kitti_offline.cpp:144-145,257-258).
static void lm_set_chain_rule(odo_lm* m, const float w[6], float th) {
  for (int i = 0; i < 6; i++) m->kf_rule[i] = w[i];
  m->kf_rule[6] = th;
  m->kf_on = 1;
}
// Queues the Solve of `next_img` against the SAME keyframe behind the Solve in flight (m->job, all of whose launches must be out),
// starting from the pose that Solve will leave — what the runner's Reset(pose_to_keyframe, lambda) hands the next Solve (ref: :261,
// :268) — before that pose exists: the GPU goes from one Solve into the next without the host in between (15-18 us per frame on
// bench.py's drive). The caller orders the stream behind next_img's construction first. Returns 0 queued, 1 not possible.
static int lm_chain_begin(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* next_img, float lambda) {
  m->chained.active = 0;
  if (!m->kf_on || !m->fused || !m->job.active || !m->job.issued_all || !m->job.result_by_launch) return 1;
  if (m->job.kf_img != kf_img || m->job.kf_dep != kf_dep || m->job.kf_img_ver != kf_img->version || m->job.kf_dep_ver != kf_dep->version) return 1;
  if (lm_check_pyrs(m, kf_img, kf_dep, next_img)) return 1;
  if (!lm_fused_eligible(m)) return 1;
  const int rc = lm_fused_begin_job(m, m->chained, kf_img, kf_dep, next_img, m->job.token, lambda);
  if (rc != 0) { m->chained.active = 0; return 1; }
  return 0;
}
// After the Solve in flight has been collected: what its guard told the chained Solve — 1 it runs, 2 its launches return at once,
// 0 there is none / unknown (its launches then saw a guard without their token and return at once: the caller treats 0 as 2, no
// stream sync needed — the chained launches are harmless either way and the next Solve is issued behind them).
static int lm_chain_verdict(odo_lm* m, int token_of_collected, int slot_of_collected) {
  if (!m->chained.active || m->chained.a.chain_in_token != token_of_collected) return 0;
  volatile int* w = (volatile int*)(m->h_res + 48 * slot_of_collected) + 43;
  const auto t0 = std::chrono::steady_clock::now();
  while ((w[0] >> 2) != token_of_collected) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return 0;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return w[0] & 3;
}
// The chained Solve becomes the job in flight (call after odo_lm_reset with the pose it started from).
static void lm_chain_adopt(odo_lm* m) {
  m->job = m->chained;
  m->job.active = 1;
  m->chained.active = 0;
}

// ---- armed Solves (the single tracker) ------------------------------------------------------------------------------------
// Between the last evaluation of one Solve and the first of the next lie the result's trip to the host, the runner's keyframe test, a
// launch call (3 us) and the dispatch. An armed Solve has its coarse launch queued behind the Solve in flight BEFORE that one has
// returned; it starts the moment the launch in front of it retires and reads the host's word: lm_arm_go writes the pose Reset would set
// (ref: run_odometry_kitti_offline.cpp:268) and issues the persistent launch behind it; lm_arm_abort tells it to return (new keyframe,
// failure). The host's own keyframe test decides, nothing is guessed on the device (unlike the chained Solves above), and the
// arithmetic is the unarmed Solve's. (The same on a second stream, resident while it waits: tools/experiments/armed_solve — slower.)
static int lm_enable_arming(odo_lm* m) {
  if (m->arm_on) return 0;
  HIP_OK(hipHostMalloc((void**)&m->h_arm, sizeof(unsigned long long) * 32, hipHostMallocMapped | hipHostMallocCoherent));
  memset(m->h_arm, 0, sizeof(unsigned long long) * 32);
  HIP_OK(hipHostGetDevicePointer((void**)&m->d_arm_map, m->h_arm, 0));
  m->arm_fault_at = getenv("ODO_ARM_FAULT") ? atol(getenv("ODO_ARM_FAULT")) : 0;
  m->arm_wait = getenv("ODO_ARM_WAIT_US") ? (unsigned)(100L * atol(getenv("ODO_ARM_WAIT_US"))) : 0u;
  m->arm_on = 1;
  return 0;
}
// Queues the coarse launch of the Solve of `next_img` against the keyframe of the Solve in flight (m->job, all of whose launches must
// be out: a coarse + a persistent launch) behind it; next_img must be complete (the caller has seen its event). lambda: what Reset
// will set. Returns 0 armed, 1 not possible (nothing was launched).
static int lm_arm_begin(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* next_img, float lambda) {
  if (m->armed.active) lm_arm_abort(m);
  // (m->job describes the Solve in flight also while odo_lm_solve is collecting it — it clears `active` first —, which is when the
  //  tracker calls this, from that Solve's wait loop)
  if (!m->arm_on || !m->fused || !m->job.issued_all || !m->job.result_by_launch || m->job.launches != 2) return 1;
  if (m->job.kf_img != kf_img || m->job.kf_dep != kf_dep || m->job.kf_img_ver != kf_img->version || m->job.kf_dep_ver != kf_dep->version) return 1;
  if (lm_check_pyrs(m, kf_img, kf_dep, next_img)) return 1;
  if (!lm_fused_eligible(m)) return 1;
  const int rc = lm_fused_begin_job(m, m->armed, kf_img, kf_dep, next_img, 0, lambda, true);
  if (rc != 0) { m->armed.active = 0; return 1; }
  return 0;
}
static void lm_arm_write(odo_lm* m, const float pose[16], int verdict) {
  volatile unsigned long long* w = m->h_arm;
  const unsigned long long tag = (unsigned long long)(unsigned)m->armed.token << 32;
  for (int i = 0; i < 16; i++) { unsigned bits = 0; if (pose) memcpy(&bits, &pose[i], 4); w[i] = tag | bits; }
  std::atomic_thread_fence(std::memory_order_release);
  w[16] = tag | (unsigned)verdict;
  std::atomic_thread_fence(std::memory_order_seq_cst);   // out of the store buffer now
}
// The Solve in flight has returned `pose` and the keyframe stays: the armed Solve starts from it (call after odo_lm_reset(pose, lambda))
// and becomes the job in flight. Returns 1 when there is no armed Solve any more (told to return by a redo of the Solve in flight).
static int lm_arm_go(odo_lm* m, const float pose[16]) {
  if (!m->armed.active) return 1;
  // test hook (ODO_ARM_FAULT=n): the n-th word is never written — the armed launch runs into its bound and reports a give-up, the
  // Solve is redone the ordinary way
  if (!(m->arm_fault_at > 0 && m->arm_used + 1 == m->arm_fault_at)) lm_arm_write(m, pose, 1);
  lm_job_launch_fine(m, m->armed);
  m->job = m->armed;
  m->job.active = 1;
  m->armed.active = 0;
  m->arm_used++;
  return 0;
}
static void lm_arm_abort(odo_lm* m) {
  if (!m->armed.active) return;
  lm_arm_write(m, nullptr, 2);
  m->armed.active = 0;
  m->arm_aborted++;
}

extern "C" int odo_lm_solve(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img,
                            float out_colmajor[16]) {
  if (!out_colmajor) return fail("odo_lm_solve: NULL out");
  // failure value first: pseudo-identity whose (3,3) is 0 (ref: src/lm_optimizer.cpp:48-52,60-65)
  for (int i = 0; i < 16; i++) out_colmajor[i] = 0.0f;
  out_colmajor[0] = out_colmajor[5] = out_colmajor[10] = 1.0f;
  if (lm_check_pyrs(m, kf_img, kf_dep, cur_img)) return -1;
  hipStream_t s = m->ctx->stream;
  HIP_OK(hipSetDevice(m->ctx->device));
  const bool resumed = lm_job_matches(m, kf_img, kf_dep, cur_img);   // started earlier by odo_lm_solve_begin
  if (!resumed) {
    m->job.active = 0;
    m->chained.active = 0;
    if (lm_prepare_keyframe(m, kf_img, kf_dep)) return -1;
  }
  int launches = 0;
  double bytes_per_level[ODO_MAX_LEVELS] = {0};
  // Early exit without a host sync: the device publishes its progress in host-mapped memory; the host stays at
  // most `run_ahead` launches ahead of the device and stops issuing launches once the device reports that the Solve
  // (fused pipeline) or the level (unfused pipeline) has ended. Stale launches are no-ops on the device either way.
  bool fused = resumed || lm_fused_eligible(m);
  int seq = 0;
  bool started = resumed;
  const int fine_k_asked = m->fine_k;
fused_again:
  if (fused) {
    // ---- fused pipeline: the coarse launch + one persistent launch (or identical generic step launches); the device walks
    // the pyramid itself ----
    if (!started && lm_fused_begin(m, kf_img, kf_dep, cur_img)) return -1;
    started = true;
    lm_fused_pump(m, true);
    launches = m->job.launches;
    seq = m->job.seq;
    for (int l = 0; l < ODO_MAX_LEVELS; l++) bytes_per_level[l] = m->job.bytes_per_level[l];
  } else {
    // ---- unfused pipeline (t-distribution mode, dense levels): residual kernel(s) + update kernel per evaluation ----
    HIP_OK(hipMemcpyAsync(m->d_init, m->init, sizeof(float) * 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lm_begin_solve_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_init, m->d_cost);
    m->ust = m->d_state; m->upo = 0;
    if (lm_unfused_levels(m, kf_img, kf_dep, cur_img, m->n_levels - 1, &launches, bytes_per_level)) return -1;
  }
  if (fused) {
    const int token = m->job.token, slot = m->job.slot;
    const bool result_by_launch = m->job.result_by_launch;
    LmState* st[2] = {m->d_state, m->d_state + 1};
    double* part[2] = {m->d_partials, m->d_partials + (size_t)kLmMaxBlocks * ODO_NACC};
    m->job.active = 0;
    // The result comes back through host-mapped memory: no copy operation and no stream-sync call on the critical
    // path; the host spins on the completion word (bounded; falls back to a stream sync).
    auto launch_finalize = [&]() {
      FinalizeArgs fa;
      fa.st_in = st[seq & 1]; fa.part_in = part[seq & 1]; fa.precision = m->precision; fa.trace = m->record ? m->d_trace : nullptr;
      fa.cost_stat = m->record ? m->d_cost : nullptr; fa.st_out = st[0]; fa.out = m->d_res_map + 48 * slot; fa.done_flag = m->d_done + slot;
      fa.token = token; fa.first_of_solve = (seq == 0) ? 1 : 0;
      memcpy(fa.init, m->init, sizeof(fa.init));
      hipLaunchKernelGGL(lm_fused_finalize_kernel, dim3(1), dim3(kLmBlock), 0, s, fa);
    };
    // Not polling (or nothing launched): a finalize launch consumes the last evaluation and reports.
    if (!result_by_launch) launch_finalize();
    HIP_OK(hipGetLastError());
    volatile int* done = m->h_done + slot;
    const auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    while (done[0] != token) {
      if (m->idle_pump) m->idle_pump(m->idle_arg);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { ok = false; break; }
    }
    if (!ok) {  // never hang: drain the stream; if no launch reported (it should have), finalize explicitly
      lm_arm_abort(m);   // (an armed launch queued behind this Solve would hold the stream until its word came)
      HIP_OK(hipStreamSynchronize(s));
      if (done[0] != token) {
        launch_finalize();
        HIP_OK(hipStreamSynchronize(s));
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    lap_start();
    memcpy(m->h_out, m->h_res + 48 * slot, sizeof(float) * 42);
    if (!m->record) memset(&m->h_out[26], 0, sizeof(float) * 16);   // (cost statistics nobody asked for: lm_write_result leaves them out)
    if (m->h_out[16] == -2.0f && m->fine_k > 0) {
      // The persistent launch gave up: one of its workgroups never showed up within the wait bound (they wait for each other,
      // so all of them must be resident at once: another client of this GPU can hold the CUs they need). Nothing is lost but
      // time: the same Solve again on the step launches, which need no co-residency; after three such Solves the optimiser
      // stays on them (fine_note_giveup: for a while that doubles each time).
      m->fine_bails++;
      (void)fine_note_giveup(&m->fine_strikes, &m->fine_clean, &m->fine_offs);
      m->fine_k = 0;
      started = false;
      launches = 0;
      m->chained.active = 0;   // (a Solve chained behind this one sees a guard without its token and returns at once)
      lm_arm_abort(m);         // (... and an armed one is told to return: it sits in this stream)
      HIP_OK(hipStreamSynchronize(s));
      fused = lm_fused_eligible(m);   // (t-distribution weights: without the persistent launch the fused part may be empty)
      goto fused_again;
    }
    if (fine_k_asked > 0 && m->fine_k == 0 && m->fine_strikes < 3) m->fine_k = fine_k_asked;
    // A disturbance that has passed must not cost the persistent launch for good: 1 024 clean Solves forgive the strikes, and an
    // optimiser that was switched off tries again after 4 096 Solves on the step launches — 8 192 after the next switch-off, and
    // so on (one more give-up switches it off again).
    if (m->fine_k_cfg > 0 && fine_note_clean(m->fine_k > 0, &m->fine_strikes, &m->fine_clean, &m->fine_offs)) m->fine_k = m->fine_k_cfg;
    const int stop = m->job.stop_level;
    if (stop > 0 && m->h_out[16] == 0.0f) {
      // ---- hand-over: the fine levels are dense. The finishing launch left the state in d_state[2] (ordered before anything
      // enqueued from here on; the launches the host had queued ahead are no-ops that touch d_state[0 / 1] only); the unfused
      // pipeline carries on from it with its own progress words, then reports like an unfused Solve.
      if (!result_by_launch || !ok)   // the explicit finalize wrote st[0]: bring it over
        HIP_OK(hipMemcpyAsync(m->d_state + 2, st[0], sizeof(LmState), hipMemcpyDeviceToDevice, s));
      m->ust = m->d_state + 2; m->upo = 16;
      const int rc = lm_unfused_levels(m, kf_img, kf_dep, cur_img, stop - 1, &launches, bytes_per_level);
      if (rc == 0) {
        hipLaunchKernelGGL(lm_finalize_kernel, dim3(1), dim3(64), 0, s, (const LmState*)m->ust, m->d_out);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(m->h_out, m->d_out, sizeof(float) * 42, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
      }
      m->ust = m->d_state; m->upo = 0;
      if (rc) return -1;
    }
  } else {
    hipLaunchKernelGGL(lm_finalize_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(m->h_out, m->d_out, sizeof(float) * 42, hipMemcpyDeviceToHost, s));  // pose, status, counters, costs
    HIP_OK(hipStreamSynchronize(s));
  }
  m->trace_stale = 1;  // the per-evaluation trace stays on the device until odo_lm_trace() asks for it
  memcpy(out_colmajor, m->h_out, sizeof(float) * 16);
  m->last_token = fused ? m->job.token : 0;
  m->last_slot = fused ? m->job.slot : 0;
  m->last_evals = (int)m->h_out[17];
  m->last_launches = launches;
  m->last_bytes = 0.0;
  for (int l = 0; l < ODO_MAX_LEVELS; l++) {
    m->iters[l] = (int)m->h_out[18 + l];
    m->last_bytes += bytes_per_level[l] * m->iters[l];
  }
  if (m->ev_on && fused) {
    m->ev_launches += launches;
    m->ev_coarse_all += m->last_coarse;
    m->ev_active += m->last_evals;
    m->ev_bytes += m->last_bytes;
  }
  if (m->h_out[16] != 0.0f) return fail("Optimize failed! ");  // ref: src/lm_optimizer.cpp:60-61
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Batched Solve: S independent optimisers (sequences) advance in the SAME launches (blockIdx.y = sequence). A single
// sequence is a serial chain of ~8 us launches that leaves the chip idle; S chains side by side cost the time of the
// longest one. Every optimiser keeps its own state, partial sums, point lists, trace, progress and result words — the
// per-sequence arithmetic and launch order are exactly those of odo_lm_solve, so the results are bit-identical to S
// separate Solves (tests/test_gpu_batch.py). All optimisers must share one context (stream). Falls back to one Solve after
// the other when a sequence cannot take the fused point-list pipeline (dense levels, t-distribution weights).
// ---------------------------------------------------------------------------------------------------------------
// Fills `a` for a fused Solve of `m` (everything that stays constant over the Solve's launches). Returns the grid the
// step launches need, the launch budget of the step kernel and the first level the coarse kernel does not take.
static void lm_fill_step_args(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* cur_img, int token, StepArgs* ap, int* grid_out,
                              int* step_budget, int* min_level_out, double bytes_per_level[ODO_MAX_LEVELS], int fine_k = 0,
                              int* fine_lo_out = nullptr) {
  StepArgs& a = *ap;
  memset(&a, 0, sizeof(a));
  a.n_levels = m->n_levels;
  const int stop = lm_fused_stop_level(m);   // > 0: the levels below are dense and handed over to the unfused pipeline
  int grid = 1, budget = 0;
  for (int l = stop; l < m->n_levels; l++) {
    StepLevel& L = a.lv[l];
    L.pl = m->pl[l]; L.n = m->npts[l];
    L.rows = kf_img->r[l]; L.cols = kf_img->c[l];
    L.nblk = lm_grid_for(m, l, L.rows, L.cols);
    L.I2 = cur_img->dev + cur_img->off[l];
    L.k = lm_level_k(m, l);
    L.max_iters = m->max_iters[l];
    if (L.nblk + 1 > grid) grid = L.nblk + 1;  // + the publisher block
    budget += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
    bytes_per_level[l] = lm_level_bytes(m, l, L.rows, L.cols, L.nblk);
  }
  a.lambda0 = m->lambda; a.precision = m->precision; a.robust = m->robust; a.huber_delta = m->huber_delta;
  a.trace = m->record ? m->d_trace : nullptr; a.cost_stat = m->record ? m->d_cost : nullptr; a.host_prog = m->d_prog;
  a.out = m->d_res_map; a.done_flag = m->d_done; a.token = token;
  a.stop_level = stop;
  a.final_state = stop > 0 ? m->d_state + 2 : nullptr;
  memcpy(a.init, m->init, sizeof(a.init));
  a.st2[0] = m->d_state; a.st2[1] = m->d_state + 1;
  a.part2[0] = m->d_partials; a.part2[1] = m->d_partials + (size_t)kLmMaxBlocks * ODO_NACC;
  // which kernel takes which level: as in lm_fused_begin (fine_k workgroups per sequence in the batched persistent launch; 0: none)
  int min_level = m->n_levels, fine_lo = m->n_levels;
  lm_plan_levels(m, stop, fine_k, &min_level, &fine_lo);
  a.min_level = min_level;
  a.fine_lo = fine_lo;
  a.xbuf = m->d_xbuf;
  a.fine_epoch = (fine_lo < min_level) ? lm_fine_next_epoch(m) : 0u;   // (a sequence that only carries its state exchanges nothing)
  a.fine_dispatch = lm_fine_next_dispatch();
  a.fine_wait = m->fine_wait;
  a.fine_home = m->fine_home;
  int above = 0;   // evaluations the coarse and the persistent launch can take
  for (int l = fine_lo; l < m->n_levels; l++) above += m->max_iters[l] > 0 ? m->max_iters[l] : 0;
  *grid_out = grid;
  *step_budget = budget - above;
  *min_level_out = min_level;
  if (fine_lo_out) *fine_lo_out = fine_lo;
}

// A batched Solve as a resumable job (one per context): lm_batch_begin fills and uploads the argument table and enqueues the
// coarse launch and as many step launches as the run-ahead allows; lm_solve_batch collects a job that matches its arguments
// (same optimisers, same pyramids by build version, no Reset since) or starts one. The batched tracker begins the NEXT lock
// step's Solve as soon as this step's decisions are taken.
struct LmBatchJob {
  int active, n;
  std::vector<odo_lm*> lms;
  std::vector<const odo_pyr*> kf_img, kf_dep, cur_img;
  std::vector<unsigned long long> kf_img_ver, kf_dep_ver, cur_ver;
  std::vector<unsigned> reset_gen;
  std::vector<int> tokens;
  std::vector<std::vector<double>> bytes;
  int grid, budget, seq, launches, it;
  bool poll_ok, issued_all;
  bool fine_used;   // this job has a batched persistent launch (lm_fine_kernel_batch)
  bool fine_off_once = false;   // the next lm_batch_begin must not use it (redo after a launch that gave up)
};
static void lm_batch_job_free(odo_ctx* c) { delete c->lm_batch_job; c->lm_batch_job = nullptr; }

static bool lm_batch_all_finished(const LmBatchJob& jb) {
  for (int i = 0; i < jb.n; i++) if (((volatile int*)jb.lms[i]->h_prog)[1] != jb.tokens[i]) return false;
  return true;
}
static int lm_batch_min_progress(const LmBatchJob& jb) {
  int p = 1 << 30;
  for (int i = 0; i < jb.n; i++) {
    if (((volatile int*)jb.lms[i]->h_prog)[1] == jb.tokens[i]) continue;
    const int v = ((volatile int*)jb.lms[i]->h_prog)[0];
    const int q = ((v >> kProgSeqBits) == jb.tokens[i]) ? (v & ((1 << kProgSeqBits) - 1)) : 0;
    if (q < p) p = q;
  }
  return p;
}
// Issues step launches; block = false: returns as soon as the next launch would have to wait for the device.
static void lm_batch_pump(odo_ctx* cx, bool block, void (*idle)(void*), void* idle_arg) {
  LmBatchJob& jb = *cx->lm_batch_job;
  if (!jb.active || jb.issued_all) return;
  hipStream_t s = cx->stream;
  const int run_ahead = jb.lms[0]->run_ahead;
  while (jb.it < jb.budget + 1) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!lm_batch_all_finished(jb) && jb.seq - lm_batch_min_progress(jb) > run_ahead) {
      if (!block) return;
      if (idle) idle(idle_arg);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { jb.poll_ok = false; break; }  // never hang
    }
    if (lm_batch_all_finished(jb)) break;
    hipLaunchKernelGGL(lm_step_kernel_batch, dim3(jb.grid, jb.n), dim3(kLmBlock), 0, s, (const StepArgs*)cx->lm_batch_d, jb.seq,
                       (jb.seq == 0) ? 1 : 0, lm_span_slot(jb.lms[0], jb.launches, false));
    jb.seq++; jb.launches++;
    if (!jb.poll_ok && jb.it >= jb.budget) break;
    jb.it++;
  }
  jb.issued_all = true;
}

// Returns 0 started, 1 not batchable (nothing started), -1 error. The keyframe lists must be current for every optimiser.
static int lm_batch_begin(int n, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                          const odo_pyr* const* cur_img) {
  odo_ctx* cx = lms[0]->ctx;
  hipStream_t s = cx->stream;
  if (!cx->lm_batch_job) cx->lm_batch_job = new LmBatchJob();
  LmBatchJob& jb = *cx->lm_batch_job;
  jb.active = 0;
  bool batchable = n <= 64;
  for (int i = 0; i < n; i++) {
    odo_lm* m = lms[i];
    m->job.active = 0;
    if (lm_prepare_keyframe(m, kf_img[i], kf_dep[i])) return -1;
    // the fused point-list pipeline must cover the top of the pyramid; what lies below its hand-over level (dense levels) is
    // batched too (lm_unfused_levels_batch) provided every stream hands over at the same level and those levels are all dense
    bool ok = m->fused && m->robust != 2 && m->poll && lm_fused_eligible(m) && lm_fused_stop_level(m) == lm_fused_stop_level(lms[0]) &&
              m->n_levels == lms[0]->n_levels;
    for (int l = 0; l < lm_fused_stop_level(m); l++)
      if (m->use_list[l] || kf_img[i]->r[l] != kf_img[0]->r[l] || kf_img[i]->c[l] != kf_img[0]->c[l]) ok = false;
    batchable = batchable && ok;
  }
  if (!batchable) return 1;
  if (cx->lm_batch_cap < n) {
    if (cx->lm_batch_h) { HIP_OK(hipStreamSynchronize(s)); (void)hipHostFree(cx->lm_batch_h); (void)hipFree(cx->lm_batch_d); }
    cx->lm_batch_cap = 0; cx->lm_batch_h = cx->lm_batch_d = nullptr;
    HIP_OK(hipHostMalloc(&cx->lm_batch_h, sizeof(StepArgs) * (size_t)n, hipHostMallocDefault));
    HIP_OK(hipMalloc(&cx->lm_batch_d, sizeof(StepArgs) * (size_t)n));
    cx->lm_batch_cap = n;
  }
  StepArgs* const h_table = (StepArgs*)cx->lm_batch_h;
  StepArgs* const d_table = (StepArgs*)cx->lm_batch_d;
  jb.n = n;
  jb.lms.assign(lms, lms + n); jb.kf_img.assign(kf_img, kf_img + n); jb.kf_dep.assign(kf_dep, kf_dep + n); jb.cur_img.assign(cur_img, cur_img + n);
  jb.kf_img_ver.resize(n); jb.kf_dep_ver.resize(n); jb.cur_ver.resize(n); jb.reset_gen.resize(n); jb.tokens.resize(n);
  jb.bytes.assign(n, std::vector<double>(ODO_MAX_LEVELS, 0.0));
  // The batched persistent launch: every sequence on its own XCD (beyond eight, several per XCD with fewer workgroups each).
  // ODO_LM_BATCH_FINE_K: workgroups per sequence (0 = off); default 32 (an XCD's CUs) up to four sequences, 16 up to eight, 8
  // beyond — the depth front end of that many frames needs most of the chip while the workgroups wait for each other
  // (measured, frames/s at S = 4 / 8 / 11: 32 workgroups 8 850 / 12 130 / —, 16: 8 100 / 12 580 / 10 100, 8: 8 140 / 12 780 / 13 230,
  // none: 7 930 / 12 320 / 12 830).
  static const int fine_env = getenv("ODO_LM_BATCH_FINE_K") ? atoi(getenv("ODO_LM_BATCH_FINE_K")) : -1;
  const int per_xcd = (n + 7) / 8;
  int fine_k = (fine_env >= 0) ? fine_env : ((n <= 4) ? kFineKMax : (n <= 8) ? kFineKMax / 2 : kFineKMax / 4);
  if (fine_k * per_xcd > kFineKMax) fine_k = kFineKMax / per_xcd;
  if (cx->batch_fine_strikes >= 3 || lms[0]->fine_k_cfg <= 0 || getenv("ODO_LM_NO_FINE")) fine_k = 0;
  if (jb.fine_off_once) { fine_k = 0; jb.fine_off_once = false; }
  int grid = 1, budget = 0, any_coarse = 0, any_fine = 0;
  bool lean = true;   // every sequence a trackers' optimiser (Huber / L2, floor sampling, nothing recorded): the lean kernel builds
  for (int i = 0; i < n; i++) {
    odo_lm* m = lms[i];
    if (m->robust == 2 || m->record || m->bilinear) lean = false;
    jb.kf_img_ver[i] = kf_img[i]->version; jb.kf_dep_ver[i] = kf_dep[i]->version; jb.cur_ver[i] = cur_img[i]->version;
    jb.reset_gen[i] = m->reset_gen;
    m->token = (m->token % 0x3ffff) + 1;
    jb.tokens[i] = m->token;
    int g = 1, b = 0, ml = 0, fl = 0;
    lm_fill_step_args(m, kf_img[i], cur_img[i], jb.tokens[i], &h_table[i], &g, &b, &ml, jb.bytes[i].data(), fine_k, &fl);
    if (g > grid) grid = g;
    if (b > budget) budget = b;
    if (ml < m->n_levels) any_coarse = 1;
    if (fl < ml) any_fine = 1;
    m->last_coarse = (ml < m->n_levels) ? 1 : 0;
  }
  // the table of the previous batched Solve may still be read by its draining launches: order the upload behind them
  HIP_OK(hipMemcpyAsync(d_table, h_table, sizeof(StepArgs) * (size_t)n, hipMemcpyHostToDevice, s));
  jb.grid = grid; jb.budget = budget; jb.seq = 0; jb.launches = 0; jb.it = 0; jb.poll_ok = true; jb.issued_all = false;
  if (lms[0]->ev_on > 0) lms[0]->ev_phase = lm_ev_next_phase(lms[0]);
  lms[0]->last_coarse_batch = any_coarse;
  if (any_coarse) {
    if (lean) hipLaunchKernelGGL(lm_coarse_kernel_batch, dim3(1, n), dim3(kCoarseBlock), kCoarseLdsBytes, s, (const StepArgs*)d_table, jb.seq, 1,
                                 lm_span_slot(lms[0], 0, true));
    else hipLaunchKernelGGL(lm_coarse_full_kernel_batch, dim3(1, n), dim3(kCoarseBlock), kCoarseLdsBytes, s, (const StepArgs*)d_table, jb.seq, 1,
                            lm_span_slot(lms[0], 0, true));
    jb.seq++; jb.launches++;
  }
  jb.fine_used = any_fine != 0;
  if (any_fine) {
    unsigned* const dispatch_words = lms[0]->dispatch_words;   // g_lm_fine_dispatch on this device
    const XccIds xcc_ids = lms[0]->fine_home >= 0 ? device_xcc_ids(cx->device) : XccIds{{-1, -1, -1, -1, -1, -1, -1, -1}};
    if (lean) hipLaunchKernelGGL(lm_fine_kernel_batch, dim3(8 * fine_k * per_xcd), dim3(kFineThreads), 0, s, (const StepArgs*)d_table, n, fine_k, jb.seq,
                                 (jb.seq == 0) ? 1 : 0, lm_span_slot(lms[0], jb.launches, false), lms[0]->fine_fault, xcc_ids, dispatch_words);
    else hipLaunchKernelGGL(lm_fine_trace_kernel_batch, dim3(8 * fine_k * per_xcd), dim3(kFineThreads), 0, s, (const StepArgs*)d_table, n, fine_k, jb.seq,
                            (jb.seq == 0) ? 1 : 0, lm_span_slot(lms[0], jb.launches, false), lms[0]->fine_fault, xcc_ids, dispatch_words);
    jb.seq++; jb.launches++;
  }
  jb.active = 1;
  lm_batch_pump(cx, false, nullptr, nullptr);
  HIP_OK(hipGetLastError());
  return 0;
}

static bool lm_batch_matches(const odo_ctx* cx, int n, odo_lm* const* lms, const odo_pyr* const* kf_img,
                             const odo_pyr* const* kf_dep, const odo_pyr* const* cur_img) {
  const LmBatchJob* jb = cx->lm_batch_job;
  if (!jb || !jb->active || jb->n != n) return false;
  for (int i = 0; i < n; i++)
    if (jb->lms[i] != lms[i] || jb->kf_img[i] != kf_img[i] || jb->kf_dep[i] != kf_dep[i] || jb->cur_img[i] != cur_img[i] ||
        jb->kf_img_ver[i] != kf_img[i]->version || jb->kf_dep_ver[i] != kf_dep[i]->version || jb->cur_ver[i] != cur_img[i]->version ||
        jb->reset_gen[i] != lms[i]->reset_gen) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// The unfused pipeline for several streams in the SAME launches: levels l_hi .. 0 (all dense) of the optimisers lms[0 .. nc),
// each on its own state m->ust with its own progress words h_prog[m->upo ...]. Per LM iteration ONE evaluation launch
// (lm_dense_eval_batch_kernel, blockIdx.y = stream) and ONE update launch (lm_update_batch_kernel, blockIdx.x = stream): every
// stream's blocks do exactly what its own launches would do, so the results are bit-identical to lm_unfused_levels stream by
// stream. A level is issued until EVERY stream's loop has stopped on it (a stopped stream's blocks return at once).
// ---------------------------------------------------------------------------------------------------------------
static int lm_unfused_levels_batch(int nc, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                                   const odo_pyr* const* cur_img, int l_hi, int* launches_io, std::vector<std::vector<double>>& bytes,
                                   void (*idle)(void*), void* idle_arg) {
  odo_ctx* cx = lms[0]->ctx;
  hipStream_t s = cx->stream;
  if (cx->dense_cap < nc) {
    if (cx->dense_h) {
      HIP_OK(hipStreamSynchronize(s));
      (void)hipHostFree(cx->dense_h); (void)hipHostFree(cx->upd_h); (void)hipFree(cx->dense_d); (void)hipFree(cx->upd_d);
    }
    cx->dense_h = cx->dense_d = cx->upd_h = cx->upd_d = nullptr; cx->dense_cap = 0;
    HIP_OK(hipHostMalloc(&cx->dense_h, sizeof(DenseBatchItem) * ODO_MAX_LEVELS * (size_t)nc, hipHostMallocDefault));
    HIP_OK(hipHostMalloc(&cx->upd_h, sizeof(UpdItem) * ODO_MAX_LEVELS * (size_t)nc, hipHostMallocDefault));
    HIP_OK(hipMalloc(&cx->dense_d, sizeof(DenseBatchItem) * ODO_MAX_LEVELS * (size_t)nc));
    HIP_OK(hipMalloc(&cx->upd_d, sizeof(UpdItem) * ODO_MAX_LEVELS * (size_t)nc));
    cx->dense_cap = nc;
  }
  const int cap = cx->dense_cap;
  for (int i = 0; i < nc; i++)
    for (int k = 0; k < 16; k++) lms[i]->h_prog[lms[i]->upo + k] = 0;   // nothing of the unfused pipeline is draining (see lm_unfused_levels)
  bool poll = true;
  int seq = 0;
  const int run_ahead = lms[0]->run_ahead;
  for (int l = l_hi; l >= 0; l--) {  // ref: src/lm_optimizer.cpp:92
    // one table row per level: a row is rewritten only by the next Solve, long after the launches that read it
    DenseBatchItem* dh = (DenseBatchItem*)cx->dense_h + (size_t)l * cap;
    DenseBatchItem* dd = (DenseBatchItem*)cx->dense_d + (size_t)l * cap;
    UpdItem* uh = (UpdItem*)cx->upd_h + (size_t)l * cap;
    UpdItem* ud = (UpdItem*)cx->upd_d + (size_t)l * cap;
    int max_nblk = 1, max_it = 0;
    for (int i = 0; i < nc; i++) {
      odo_lm* m = lms[i];
      LevelView v;
      v.I1 = kf_img[i]->dev + kf_img[i]->off[l];
      v.I2 = cur_img[i]->dev + cur_img[i]->off[l];
      v.D1 = kf_dep[i]->dev + kf_dep[i]->off[l];
      v.rows = kf_img[i]->r[l]; v.cols = kf_img[i]->c[l];
      const DenseLevel L = lm_dense_level(v, lm_level_k(m, l), m->max_iters[l]);
      memset(&dh[i], 0, sizeof(dh[i]));
      dh[i].L = L; dh[i].st = m->ust; dh[i].scale_sqr = m->d_scale; dh[i].partials = m->d_partials;
      dh[i].expect_level = l; dh[i].robust = m->robust; dh[i].huber_delta = m->huber_delta;
      memset(&uh[i], 0, sizeof(uh[i]));
      uh[i].st = m->ust; uh[i].partials = m->d_partials; uh[i].nblk = L.nblk; uh[i].expect_level = l; uh[i].precision = m->precision;
      uh[i].max_iters = m->max_iters[l]; uh[i].trace = m->record ? m->d_trace : nullptr; uh[i].cost_stat = m->d_cost;
      uh[i].host_prog = m->d_prog + m->upo; uh[i].lambda0 = m->lambda; uh[i].init = m->d_init; uh[i].out = m->d_out;
      bytes[i][l] = lm_level_bytes(m, l, v.rows, v.cols, L.nblk);
      if (L.nblk > max_nblk) max_nblk = L.nblk;
      if (m->max_iters[l] > max_it) max_it = m->max_iters[l];
    }
    HIP_OK(hipMemcpyAsync(dd, dh, sizeof(DenseBatchItem) * (size_t)nc, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(ud, uh, sizeof(UpdItem) * (size_t)nc, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lm_begin_level_batch_kernel, dim3(nc), dim3(64), 0, s, (const UpdItem*)ud);
    for (int it = 0; it < max_it; it++) {  // ref: :117
      if (poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
          bool all_stopped = true;
          int min_prog = 1 << 30;
          for (int i = 0; i < nc; i++) {
            volatile int* prog = lms[i]->h_prog + lms[i]->upo;
            if (!prog[2 + l]) all_stopped = false;
            if (prog[0] < min_prog) min_prog = prog[0];
          }
          if (all_stopped) { it = max_it; break; }
          if (seq - min_prog <= run_ahead) break;
          if (idle) idle(idle_arg);
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { poll = false; break; }  // never hang
        }
        if (it >= max_it) break;  // every stream's loop on this level has stopped on the device
      }
      seq++;
      launch_dense_eval_batch(dd, nc, max_nblk, s, nullptr, nullptr, lms[0]->dense_plain_div);
      hipLaunchKernelGGL(lm_update_batch_kernel, dim3(nc), dim3(kUpdThreads), 0, s, (const UpdItem*)ud, seq);
      (*launches_io)++;
    }
  }
  // affine_ = current_estimate.matrix() of every stream, then one read-back each behind the same sync
  hipLaunchKernelGGL(lm_finalize_batch_kernel, dim3(nc), dim3(64), 0, s, (const UpdItem*)((UpdItem*)cx->upd_d));   // row of level 0
  HIP_OK(hipGetLastError());
  for (int i = 0; i < nc; i++) HIP_OK(hipMemcpyAsync(lms[i]->h_out, lms[i]->d_out, sizeof(float) * 42, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}

// idle / idle_arg: called from the wait loops, may be NULL.
static int lm_solve_batch(int n, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                          const odo_pyr* const* cur_img, float* out_colmajor /* n x 16 */, int* status /* n */,
                          void (*idle)(void*), void* idle_arg) {
  if (n < 1 || !lms || !kf_img || !kf_dep || !cur_img || !out_colmajor || !status) return fail("odo_lm_solve_batch: bad arg");
  for (int i = 0; i < n; i++) {
    if (!lms[i] || lms[i]->ctx != lms[0]->ctx) return fail("odo_lm_solve_batch: the optimisers must share one context");
    for (int j = 0; j < i; j++) if (lms[j] == lms[i]) return fail("odo_lm_solve_batch: the same optimiser twice");
    if (lm_check_pyrs(lms[i], kf_img[i], kf_dep[i], cur_img[i])) return -1;
  }
  odo_ctx* cx = lms[0]->ctx;
  hipStream_t s = cx->stream;
  HIP_OK(hipSetDevice(cx->device));
  int launches = 0;
  std::vector<std::vector<double>> bytes(n, std::vector<double>(ODO_MAX_LEVELS, 0.0));
  int any_fail = 0;
  // per-stream bookkeeping once m->h_out holds the stream's final result
  auto take_result = [&](int i) {
    odo_lm* m = lms[i];
    m->trace_stale = 1;
    memcpy(out_colmajor + 16 * i, m->h_out, sizeof(float) * 16);
    m->last_evals = (int)m->h_out[17];
    m->last_launches = launches;
    m->last_bytes = 0.0;
    for (int l = 0; l < ODO_MAX_LEVELS; l++) { m->iters[l] = (int)m->h_out[18 + l]; m->last_bytes += bytes[i][l] * m->iters[l]; }
    status[i] = (m->h_out[16] != 0.0f) ? -1 : 0;
  };
  if (!lm_batch_matches(cx, n, lms, kf_img, kf_dep, cur_img)) {
    if (cx->lm_batch_job) cx->lm_batch_job->active = 0;   // a job started on other inputs is abandoned: its launches drain
    const int rc = lm_batch_begin(n, lms, kf_img, kf_dep, cur_img);
    if (rc < 0) return -1;
    if (rc == 1) {
      // No fused part to batch. Pyramids that are dense on EVERY level (no point-list level at the top) still batch: the whole
      // Solve is the unfused pipeline, all streams in the same launches.
      bool dense_only = n <= 64;
      for (int i = 0; i < n && dense_only; i++) {
        odo_lm* m = lms[i];
        dense_only = m->fused && m->robust != 2 && m->poll && m->n_levels == lms[0]->n_levels && lm_fused_stop_level(m) == m->n_levels;
        for (int l = 0; l < m->n_levels && dense_only; l++)
          if (m->use_list[l] || kf_img[i]->r[l] != kf_img[0]->r[l] || kf_img[i]->c[l] != kf_img[0]->c[l]) dense_only = false;
      }
      if (dense_only) {
        for (int i = 0; i < n; i++) {
          odo_lm* m = lms[i];
          HIP_OK(hipMemcpyAsync(m->d_init, m->init, sizeof(float) * 16, hipMemcpyHostToDevice, s));
          hipLaunchKernelGGL(lm_begin_solve_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_init, m->d_cost);
          m->ust = m->d_state; m->upo = 0;
        }
        if (lm_unfused_levels_batch(n, lms, kf_img, kf_dep, cur_img, lms[0]->n_levels - 1, &launches, bytes, idle, idle_arg)) return -1;
        for (int i = 0; i < n; i++) { take_result(i); if (status[i]) any_fail = 1; }
        if (any_fail) fail("Optimize failed! ");
        return 0;
      }
      // one after the other: same results, no batching (t-distribution weights, mixed list / dense structures)
      for (int i = 0; i < n; i++) { status[i] = odo_lm_solve(lms[i], kf_img[i], kf_dep[i], cur_img[i], out_colmajor + 16 * i); }
      return 0;
    }
  }
  int redone = 0;
collect_again:
  LmBatchJob& jb = *cx->lm_batch_job;
  lm_batch_pump(cx, true, idle, idle_arg);
  HIP_OK(hipGetLastError());
  jb.active = 0;
  const StepArgs* h_table = (const StepArgs*)cx->lm_batch_h;
  const int seq = jb.seq;
  launches = jb.launches;
  for (int i = 0; i < n; i++) bytes[i] = jb.bytes[i];
  const int stop = lm_fused_stop_level(lms[0]);   // the same for every stream of a batch
  // results: each sequence's finishing launch wrote its own host-mapped block
  for (int i = 0; i < n; i++) {
    odo_lm* m = lms[i];
    volatile int* done = m->h_done;
    const auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    while (done[0] != jb.tokens[i]) {
      if (idle) idle(idle_arg);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { ok = false; break; }
    }
    if (!ok) {  // drain, then consume this sequence's last evaluation explicitly
      HIP_OK(hipStreamSynchronize(s));
      if (done[0] != jb.tokens[i]) {
        FinalizeArgs fa;
        const StepArgs& a = h_table[i];
        fa.st_in = a.st2[seq & 1]; fa.part_in = a.part2[seq & 1]; fa.precision = m->precision; fa.trace = m->record ? m->d_trace : nullptr;
        fa.cost_stat = m->record ? m->d_cost : nullptr; fa.st_out = a.st2[0]; fa.out = m->d_res_map; fa.done_flag = m->d_done;
        fa.token = jb.tokens[i]; fa.first_of_solve = (seq == 0) ? 1 : 0;
        memcpy(fa.init, m->init, sizeof(fa.init));
        hipLaunchKernelGGL(lm_fused_finalize_kernel, dim3(1), dim3(kLmBlock), 0, s, fa);
        if (stop > 0) HIP_OK(hipMemcpyAsync(m->d_state + 2, a.st2[0], sizeof(LmState), hipMemcpyDeviceToDevice, s));   // the hand-over state
        HIP_OK(hipStreamSynchronize(s));
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    memcpy(m->h_out, m->h_res, sizeof(float) * 42);
    if (!m->record) memset(&m->h_out[26], 0, sizeof(float) * 16);
  }
  if (!redone && jb.fine_used) {
    // a sequence's persistent workgroups gave up waiting for each other (status -2: they could not all be resident): the whole
    // batched Solve again on the step launches — same results; after three such Solves this context stays on them
    bool gave_up = false;
    for (int i = 0; i < n; i++) gave_up = gave_up || lms[i]->h_out[16] == -2.0f;
    if (gave_up) {
      cx->batch_fine_bails++;
      (void)fine_note_giveup(&cx->batch_fine_strikes, &cx->batch_fine_clean, &cx->batch_fine_offs);
      HIP_OK(hipStreamSynchronize(s));
      jb.fine_off_once = true;
      if (lm_batch_begin(n, lms, kf_img, kf_dep, cur_img) != 0) return fail("odo_lm_solve_batch: redo after a persistent launch gave up failed");
      redone = 1;
      goto collect_again;
    }
  }
  if (stop > 0) {
    // ---- hand-over: the levels below `stop` are dense. Every stream whose fused part succeeded left its state in d_state[2];
    // the unfused pipeline carries them on together (one evaluation + one update launch per iteration for all of them).
    std::vector<int> c;
    for (int i = 0; i < n; i++) if (lms[i]->h_out[16] == 0.0f) c.push_back(i);
    if (!c.empty()) {
      const int nc = (int)c.size();
      std::vector<odo_lm*> l2(nc);
      std::vector<const odo_pyr*> ki(nc), kd(nc), cu(nc);
      std::vector<std::vector<double>> b2(nc, std::vector<double>(ODO_MAX_LEVELS, 0.0));
      for (int e = 0; e < nc; e++) {
        const int i = c[e];
        l2[e] = lms[i]; ki[e] = kf_img[i]; kd[e] = kf_dep[i]; cu[e] = cur_img[i];
        lms[i]->ust = lms[i]->d_state + 2; lms[i]->upo = 16;
      }
      const int rc = lm_unfused_levels_batch(nc, l2.data(), ki.data(), kd.data(), cu.data(), stop - 1, &launches, b2, idle, idle_arg);
      for (int e = 0; e < nc; e++) {
        const int i = c[e];
        lms[i]->ust = lms[i]->d_state; lms[i]->upo = 0;
        for (int l = 0; l < stop; l++) bytes[i][l] = b2[e][l];
      }
      if (rc) return -1;
    }
  }
  for (int i = 0; i < n; i++) { take_result(i); if (status[i]) any_fail = 1; }
  // (forgiveness and retry as in odo_lm_solve: lm_batch_begin uses the launch whenever batch_fine_strikes < 3)
  (void)fine_note_clean(jb.fine_used && !redone, &cx->batch_fine_strikes, &cx->batch_fine_clean, &cx->batch_fine_offs);
  if (lms[0]->ev_on) {   // launch statistics of the batched Solve, kept with the first optimiser (odo_lm_event_stats_ex)
    odo_lm* m0 = lms[0];
    m0->ev_launches += launches;
    m0->ev_coarse_all += m0->last_coarse_batch;
    for (int i = 0; i < n; i++) { m0->ev_active += lms[i]->last_evals; m0->ev_bytes += lms[i]->last_bytes; }
  }
  if (any_fail) fail("Optimize failed! ");  // ref: src/lm_optimizer.cpp:60-61 (per-sequence status in `status`)
  return 0;
}

extern "C" int odo_lm_solve_batch(int n, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                                  const odo_pyr* const* cur_img, float* out_colmajor /* n x 16 */, int* status /* n */) {
  return lm_solve_batch(n, lms, kf_img, kf_dep, cur_img, out_colmajor, status, nullptr, nullptr);
}

// HIP-event timing of the evaluation kernels (fused pipeline) on the stream they are launched on: start / stop events bound
// to the dispatch. on = 0 stops; on = 1 brackets every launch; on = N > 1 brackets every N-th launch of a Solve (the sampled
// residue rotates from Solve to Solve, so the coarse launch — index 0 — is sampled every N-th Solve): cheap enough to leave on
// inside a timed region. Turning it on clears the accumulators.
extern "C" int odo_lm_event_timing(odo_lm* m, int on) {
  if (!m) return fail("NULL lm");
  if (on < 0) return fail("odo_lm_event_timing: on must be >= 0");
  if (on && !m->d_span) {
    HIP_OK(hipSetDevice(m->ctx->device));
    HIP_OK(hipMalloc((void**)&m->d_span, sizeof(unsigned long long) * 2 * kSpanSlots));
    std::vector<unsigned long long> h(2 * (size_t)kSpanSlots);
    for (int i = 0; i < kSpanSlots; i++) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
    HIP_OK(hipMemcpy(m->d_span, h.data(), sizeof(unsigned long long) * h.size(), hipMemcpyHostToDevice));
    m->span_kind = new std::vector<odo_lm::SpanTag>();
    m->span_used = 0;
  }
  if (on) {
    if (lm_span_collect(m)) return -1;   // slots of an earlier period
    m->ev_total_us = m->ev_bytes = m->ev_coarse_us = 0.0;
    m->ev_launches = m->ev_active = m->ev_coarse_launches = m->ev_sampled = m->ev_coarse_all = 0;
    m->ev_period_us = m->ev_cperiod_us = 0.0; m->ev_period_n = m->ev_cperiod_n = 0;
    m->ev_solves = 0; m->ev_phase = 0;
  } else if (m->ev_on) {
    if (lm_span_collect(m)) return -1;
  }
  m->ev_on = on;
  return 0;
}
// out[0] sampled step-kernel time (us), out[1] sampled step launches, out[2] sampled coarse-kernel time (us), out[3] sampled
// coarse launches, out[4] all launches issued, out[5] all coarse launches, out[6] evaluations, out[7] algorithmic bytes,
// out[8] / out[9] summed start-to-start periods of consecutive sampled step launches (us) and their number, out[10] / out[11]
// the same from a coarse launch to the step launch behind it.
extern "C" int odo_lm_event_stats_ex(odo_lm* m, double out[12]) {
  if (!m || !out) return fail("NULL arg");
  if (lm_span_collect(m)) return -1;
  out[8] = m->ev_period_us; out[9] = (double)m->ev_period_n; out[10] = m->ev_cperiod_us; out[11] = (double)m->ev_cperiod_n;
  out[0] = m->ev_total_us - m->ev_coarse_us; out[1] = (double)(m->ev_sampled - m->ev_coarse_launches);
  out[2] = m->ev_coarse_us; out[3] = (double)m->ev_coarse_launches;
  out[4] = (double)m->ev_launches; out[5] = (double)m->ev_coarse_all; out[6] = (double)m->ev_active; out[7] = m->ev_bytes;
  return 0;
}
extern "C" int odo_lm_event_stats2(odo_lm* m, double* coarse_us, long* coarse_launches) {
  if (!m) return fail("NULL lm");
  if (lm_span_collect(m)) return -1;
  if (coarse_us) *coarse_us = m->ev_coarse_us;
  if (coarse_launches) *coarse_launches = m->ev_coarse_launches;
  return 0;
}
extern "C" int odo_lm_event_stats(odo_lm* m, double* total_us, long* launches, long* active_launches,
                                  double* algorithmic_bytes) {
  if (!m) return fail("NULL lm");
  if (lm_span_collect(m)) return -1;
  if (total_us) *total_us = m->ev_total_us;
  if (launches) *launches = m->ev_launches;
  if (active_launches) *active_launches = m->ev_active;
  if (algorithmic_bytes) *algorithmic_bytes = m->ev_bytes;
  return 0;
}

extern "C" int odo_lm_set_sampling(odo_lm* m, int sampling) {
  if (!m || (sampling != ODO_SAMPLE_FLOOR && sampling != ODO_SAMPLE_BILINEAR)) return fail("odo_lm_set_sampling: bad arg");
  m->bilinear = sampling == ODO_SAMPLE_BILINEAR ? 1 : 0;
  return 0;
}
extern "C" int odo_lm_set_mode(odo_lm* m, int mode) {
  if (!m || mode < 0 || mode > 2) return fail("odo_lm_set_mode: bad arg");
  m->mode = mode;
  m->kf_img_ver = m->kf_dep_ver = 0;  // rebuild the per-level choice on the next Solve
  return 0;
}
extern "C" int odo_lm_points(const odo_lm* m, int npts[ODO_MAX_LEVELS], int use_list[ODO_MAX_LEVELS]) {
  if (!m) return fail("NULL lm");
  for (int l = 0; l < ODO_MAX_LEVELS; l++) {
    if (npts) npts[l] = m->npts[l];
    if (use_list) use_list[l] = m->use_list[l];
  }
  return 0;
}

extern "C" int odo_lm_report(const odo_lm* m, int iters[4], float cost[4][2]) {
  if (!m) return fail("NULL lm");
  for (int i = 0; i < 4; i++) {
    if (iters) iters[i] = m->iters[i];
    if (cost) { cost[i][0] = m->h_cost[i * 2]; cost[i][1] = m->h_cost[i * 2 + 1]; }
  }
  return 0;
}

extern "C" int odo_lm_set_record(odo_lm* m, int on) {
  if (!m) return fail("odo_lm_set_record: NULL arg");
  m->record = on ? 1 : 0;
  return 0;
}
extern "C" int odo_lm_trace(const odo_lm* mc, odo_lm_trace_row* rows, int cap, int* n_rows) {
  odo_lm* m = const_cast<odo_lm*>(mc);
  if (!m || !n_rows) return fail("NULL arg");
  if (!m->record) return fail("odo_lm_trace: this optimiser does not record its trace (odo_lm_set_record(lm, 0), or a tracker's own: ODO_LM_TRACE=1 keeps it)");
  if (m->trace_stale) {
    HIP_OK(hipMemcpyAsync(m->h_trace, m->d_trace, sizeof(LmTraceRow) * kTraceCap, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_OK(hipStreamSynchronize(m->ctx->stream));
    m->trace_stale = 0;
  }
  int n = m->last_evals < kTraceCap ? m->last_evals : kTraceCap;
  if (n > cap) n = cap;
  static_assert(sizeof(odo_lm_trace_row) == sizeof(LmTraceRow), "trace row layout");
  if (rows && n > 0) memcpy(rows, m->h_trace, sizeof(LmTraceRow) * n);
  *n_rows = n;
  return 0;
}

extern "C" int odo_lm_persistent_stats(const odo_lm* m, int* workgroups, int* fallbacks) {
  if (!m) return fail("odo_lm_persistent_stats: NULL lm");
  if (workgroups) *workgroups = (m->fine_k > 0 && m->fine_k_last > 0) ? m->fine_k_last : m->fine_k;
  if (fallbacks) *fallbacks = m->fine_bails + m->ctx->batch_fine_bails;   // its own Solves + its context's batched Solves
  return 0;
}
// Back-off state of the persistent launch: give-ups that count (3 = switched off), the interval after which a switched-off launch
// is tried again (doubles with every switch-off), and the Solves left until then (0 while the launch is on).
extern "C" int odo_lm_tdist_stats(odo_lm* m, long* multi_launches, int* fallbacks) {
  if (!m) return fail("odo_lm_tdist_stats: NULL lm");
  HIP_OK(hipSetDevice(m->ctx->device));
  HIP_OK(hipStreamSynchronize(m->ctx->stream));
  int w[2] = {0, 0};
  HIP_OK(hipMemcpy(w, m->d_ts_gave_up, sizeof(w), hipMemcpyDeviceToHost));
  if (multi_launches) *multi_launches = m->ts_multi_launches;
  if (fallbacks) *fallbacks = w[1];
  return 0;
}
extern "C" int odo_lm_persistent_backoff(const odo_lm* m, int* strikes, int* retry_after, int* solves_until_retry) {
  if (!m) return fail("odo_lm_persistent_backoff: NULL lm");
  const int ra = fine_retry_after(m->fine_offs);
  if (strikes) *strikes = m->fine_strikes;
  if (retry_after) *retry_after = ra;
  if (solves_until_retry) *solves_until_retry = (m->fine_k_cfg > 0 && m->fine_k == 0 && m->fine_strikes >= 3) ? (ra - m->fine_clean > 0 ? ra - m->fine_clean : 0) : 0;
  return 0;
}
extern "C" int odo_lm_launch_stats(const odo_lm* m, int* n_active, int* n_total, double* bytes) {
  if (!m) return fail("NULL lm");
  if (n_active) *n_active = m->last_evals;
  if (n_total) *n_total = m->last_launches;
  if (bytes) *bytes = m->last_bytes;
  return 0;
}

// Test entry: a single evaluation at an explicit pose.
__global__ void lm_force_state_kernel(LmState* st, const float* T, int level) {
  if (threadIdx.x == 0) {
    LmState s;
    float m[16];
    for (int i = 0; i < 16; i++) m[i] = T[i];
    lm_begin_solve(&s, m);
    s.level = level; s.iter = 0; s.lambda = 0.0f; s.err_last = 1e+10f; s.active = 1;
    for (int i = 0; i < 16; i++) s.T[i] = m[i];  // exactly the caller's matrix (no R->q->R round trip)
    *st = s;
  }
}
__global__ void lm_sum_partials_kernel(const double* __restrict__ partials, int nblk, double* __restrict__ out) {
  __shared__ double sh[8][32];
  const int t = threadIdx.x, q = t & 31, seg = t >> 5;
  double v = 0.0;
  if (q < ODO_NACC)
    for (int b = seg; b < nblk; b += 8) v += partials[(size_t)b * ODO_NACC + q];
  sh[seg][q] = v;
  __syncthreads();
  if (t < ODO_NACC)
    out[t] = ((((((sh[0][t] + sh[1][t]) + sh[2][t]) + sh[3][t]) + sh[4][t]) + sh[5][t]) + sh[6][t]) + sh[7][t];
}

extern "C" int odo_lm_accumulate(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img,
                                 int level, const float T_colmajor[16], double acc[ODO_NACC]) {
  if (!T_colmajor || !acc) return fail("odo_lm_accumulate: NULL arg");
  if (lm_check_pyrs(m, kf_img, kf_dep, cur_img)) return -1;
  if (level < 0 || level >= m->n_levels) return fail("odo_lm_accumulate: bad level");
  hipStream_t s = m->ctx->stream;
  HIP_OK(hipSetDevice(m->ctx->device));
  LevelView v;
  v.I1 = kf_img->dev + kf_img->off[level];
  v.I2 = cur_img->dev + cur_img->off[level];
  v.D1 = kf_dep->dev + kf_dep->off[level];
  v.rows = kf_img->r[level]; v.cols = kf_img->c[level];
  const LevelK k = lm_level_k(m, level);
  if (lm_prepare_keyframe(m, kf_img, kf_dep)) return -1;
  const int nblk = lm_grid_for(m, level, v.rows, v.cols);
  if (m->robust == 2 && lm_ensure_res(m, (size_t)v.rows * v.cols)) return -1;
  double* d_acc = nullptr;
  HIP_OK(hipMalloc((void**)&d_acc, sizeof(double) * ODO_NACC));
  HIP_OK(hipMemcpyAsync(m->d_init, T_colmajor, sizeof(float) * 16, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(lm_force_state_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_init, level);
  lm_launch_eval(m, v, k, level, nblk);
  hipLaunchKernelGGL(lm_sum_partials_kernel, dim3(1), dim3(256), 0, s, m->d_partials, nblk, d_acc);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(acc, d_acc, sizeof(double) * ODO_NACC, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipFree(d_acc));
  return (acc[28] > 0.0) ? 0 : -1;  // N == 0 fails (ref: src/lm_optimizer.cpp:244-248)
}

// Diagnostic entry: one evaluation + one STAMPed update at pose T on `level`; returns 8 cycle-counter stamps of the
// update kernel (phase boundaries: entry, fold, state copy, decide, solve, apply, trace, exit).
extern "C" int odo_debug_update_stamps(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img,
                                       int level, const float T_colmajor[16], unsigned long long stamps[8]) {
  if (!T_colmajor || !stamps) return fail("odo_debug_update_stamps: NULL arg");
  if (lm_check_pyrs(m, kf_img, kf_dep, cur_img)) return -1;
  hipStream_t s = m->ctx->stream;
  HIP_OK(hipSetDevice(m->ctx->device));
  LevelView v;
  v.I1 = kf_img->dev + kf_img->off[level];
  v.I2 = cur_img->dev + cur_img->off[level];
  v.D1 = kf_dep->dev + kf_dep->off[level];
  v.rows = kf_img->r[level]; v.cols = kf_img->c[level];
  const LevelK k = lm_level_k(m, level);
  if (lm_prepare_keyframe(m, kf_img, kf_dep)) return -1;
  const int nblk = lm_grid_for(m, level, v.rows, v.cols);
  unsigned long long* d_st = nullptr;
  HIP_OK(hipMalloc((void**)&d_st, sizeof(unsigned long long) * 8));
  HIP_OK(hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * 8, s));
  HIP_OK(hipMemcpyAsync(m->d_init, T_colmajor, sizeof(float) * 16, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(lm_force_state_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_init, level);
  lm_launch_eval(m, v, k, level, nblk);
  hipLaunchKernelGGL(lm_update_kernel<true>, dim3(1), dim3(kUpdThreads), 0, s, m->d_state, m->d_partials, nblk, level, m->precision,
                     100, m->d_trace, m->d_cost, m->d_prog, 1, d_st);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(stamps, d_st, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipFree(d_st));
  return 0;
}

// Test entry: the wave-parallel damped 6x6 solve the update kernel uses, on caller-supplied accumulators.
extern "C" int odo_debug_solve(odo_ctx* ctx, const double acc[ODO_NACC], float lambda, float delta[6]) {
  if (!ctx || !acc || !delta) return fail("odo_debug_solve: NULL arg");
  HIP_OK(hipSetDevice(ctx->device));
  double* d_acc = nullptr;
  float* d_out = nullptr;
  HIP_OK(hipMalloc((void**)&d_acc, sizeof(double) * ODO_NACC));
  HIP_OK(hipMalloc((void**)&d_out, sizeof(float) * 6));
  HIP_OK(hipMemcpyAsync(d_acc, acc, sizeof(double) * ODO_NACC, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(solve_damped_wave_test_kernel, dim3(1), dim3(64), 0, ctx->stream, d_acc, lambda, d_out);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(delta, d_out, sizeof(float) * 6, hipMemcpyDeviceToHost, ctx->stream));
  HIP_OK(hipStreamSynchronize(ctx->stream));
  HIP_OK(hipFree(d_acc));
  HIP_OK(hipFree(d_out));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Depth estimator
// ------------------------------------------------------------------------------------------------
struct odo_depth {
  odo_ctx* ctx;
  float grad_th, ssd_th, photo_th, min_depth, max_depth, lambda, huber_delta, precision, baseline;
  int max_iters, boundary, max_residuals, max_disparity, any_size;
  odo_intrinsics K;
  int rows, cols;  // size the device buffers were made for
  float *d_left, *d_right, *d_bl, *d_br, *d_disp, *d_dep, *d_d0, *d_scratch;
  uint8_t* d_val;
  uint32_t* d_pts;
  int* d_cnt;
  uint8_t* d_matched;
  void* d_compact;                // odo_depth_compact_outputs_async: {idx u32, disp f32, dep f32, val u8} x point slots
  DepthLmState* d_lmstate;
  double* d_part_e;
  int* d_part_n;
  int* d_counts;
  DepthLmStats* d_stats;
  DepthLmStats* h_stats;
  int* h_prog;
  int* d_prog;
  int poll, run_ahead;
  DepthLmStats* d_stats_map;  // device alias of the host-mapped h_stats
  int token;
  DepthLmStats last;
  // DepthOptimization in ONE persistent launch (depth_lm_persistent_kernel) instead of a launch per iteration; the step launches
  // are its fall-back (give-up policy as the pose LM's persistent launch: fine_note_giveup / fine_note_clean)
  int persist, persist_cfg;       // 1: on (ODO_DEPTH_NO_PERSIST=1 / three give-ups: off)
  int persist_off_once;           // the job being run again after a give-up goes to the step launches
  int persist_bails, persist_strikes, persist_clean, persist_offs;
  int persist_fault;              // test hook (ODO_DEPTH_PERSIST_FAULT)
  unsigned persist_epoch, persist_wait;
  int persist_home;               // the XCD its persistent launch runs on (next_home_xcd)
  unsigned long long* d_xbuf;     // kDpXbufWords
  int* d_gave_up;
  // odo_depth_prepare_left_dev: blur(left) + selection enqueued ahead of the call on another stream
  const float* prep_left;         // NULL: nothing prepared
  unsigned long long prep_stamp;
  int prep_rows, prep_cols;
  hipEvent_t prep_ev;             // recorded behind the prepared launches
  // odo_depth_compute_begin_dev: a WHOLE ComputeDepth enqueued ahead of the call on another stream (same event)
  hipStream_t job_stream;         // stream the depth_job_* functions enqueue on; NULL: the estimator's own (ctx->stream)
  struct Early {
    int active;
    const float *left, *right;
    unsigned long long left_stamp, right_stamp;
    int rows, cols;
    uint8_t* val; float *disp, *dep;
  } early;
};
static inline hipStream_t depth_stream(const odo_depth* d) { return d->job_stream ? d->job_stream : d->ctx->stream; }

extern "C" int odo_depth_create(odo_ctx* ctx, float grad_th, float ssd_th, float photo_th, float min_depth, float max_depth,
                                float lambda, float huber_delta, float precision, int max_iters, int boundary,
                                const odo_intrinsics* K, float baseline, int max_residuals, int max_disparity, int any_size,
                                odo_depth** out) {
  if (!ctx || !out) return fail("odo_depth_create: NULL arg");
  *out = nullptr;
  if (boundary < 2) return fail("odo_depth_create: boundary must be >= 2 (8-tap pattern reaches +-2)");
  odo_depth* d = new (std::nothrow) odo_depth();
  if (!d) return fail("out of memory");
  memset(d, 0, sizeof(*d));
  d->ctx = ctx; d->grad_th = grad_th; d->ssd_th = ssd_th; d->photo_th = photo_th; d->min_depth = min_depth;
  d->max_depth = max_depth; d->lambda = lambda; d->huber_delta = huber_delta; d->precision = precision;
  d->max_iters = max_iters; d->boundary = boundary; d->baseline = baseline; d->max_residuals = max_residuals;
  d->max_disparity = max_disparity; d->any_size = any_size;
  d->K = K ? *K : kKitti00;
  HIP_OK(hipSetDevice(ctx->device));
  HIP_OK(hipMalloc((void**)&d->d_pts, sizeof(uint32_t) * kSelBlocks * kSelCap));
  HIP_OK(hipMalloc((void**)&d->d_cnt, sizeof(int) * kSelBlocks));
  HIP_OK(hipMalloc((void**)&d->d_d0, sizeof(float) * kSelBlocks * kSelCap));
  HIP_OK(hipMalloc((void**)&d->d_scratch, sizeof(float) * 6 * kSelBlocks * kSelCap));
  HIP_OK(hipMalloc((void**)&d->d_matched, kSelBlocks * kSelCap));
  HIP_OK(hipMalloc((void**)&d->d_lmstate, sizeof(DepthLmState) * 2));
  HIP_OK(hipMalloc((void**)&d->d_part_e, sizeof(double) * 2 * kDlmBlocks));
  HIP_OK(hipMalloc((void**)&d->d_part_n, sizeof(int) * 2 * kDlmBlocks));
  HIP_OK(hipMalloc((void**)&d->d_counts, sizeof(int) * 3 * kDlmBlocks));
  HIP_OK(hipHostMalloc((void**)&d->h_stats, sizeof(DepthLmStats), hipHostMallocMapped | hipHostMallocCoherent));
  HIP_OK(hipHostGetDevicePointer((void**)&d->d_stats_map, d->h_stats, 0));
  HIP_OK(hipHostMalloc((void**)&d->h_prog, sizeof(int) * 8, hipHostMallocMapped | hipHostMallocCoherent));
  HIP_OK(hipHostGetDevicePointer((void**)&d->d_prog, d->h_prog, 0));
  memset(d->h_prog, 0, sizeof(int) * 8);
  d->poll = getenv("ODO_NO_POLL") ? 0 : 1;
  d->run_ahead = getenv("ODO_RUN_AHEAD") ? atoi(getenv("ODO_RUN_AHEAD")) : 3;
  d->persist = d->persist_cfg = getenv("ODO_DEPTH_NO_PERSIST") ? 0 : 1;
  d->persist_fault = getenv("ODO_DEPTH_PERSIST_FAULT") ? 1 : 0;
  d->persist_home = next_home_xcd(ctx->device);
  // Wait bound of the persistent depth launch: 0.5 ms of the 100 MHz device clock (a whole depth LM is 70-100 us). Deliberately
  // SHORTER than the pose LM's (4 ms): the two persistent kernels cannot share a CU (416 + 160 VGPRs per SIMD), so when both are
  // partially resident on one XCD each holds CUs the other is waiting for — seen ~ once in 4 000 frames of a 100 000-frame soak —
  // and the one that is NOT the frame's critical chain must be the one that yields, quickly.
  d->persist_wait = getenv("ODO_DEPTH_WAIT_US") ? (unsigned)(100L * atol(getenv("ODO_DEPTH_WAIT_US"))) : 50000u;
  HIP_OK(hipMalloc((void**)&d->d_xbuf, sizeof(unsigned long long) * kDpXbufWords));
  HIP_OK(hipMemset(d->d_xbuf, 0, sizeof(unsigned long long) * kDpXbufWords));
  HIP_OK(hipMalloc((void**)&d->d_gave_up, sizeof(int) * 512));   // [0] the flag, then diagnostics of the last give-up
  HIP_OK(hipMemset(d->d_gave_up, 0, sizeof(int) * 512));
  HIP_OK(hipEventCreateWithFlags(&d->prep_ev, hipEventDisableTiming));
  *out = d;
  return 0;
}

static void depth_free_images(odo_depth* d) {
  float** ps[] = {&d->d_left, &d->d_right, &d->d_bl, &d->d_br, &d->d_disp, &d->d_dep};
  for (auto p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
  if (d->d_val) (void)hipFree(d->d_val);
  d->d_val = nullptr;
  d->rows = d->cols = 0;
}

extern "C" int odo_depth_destroy(odo_depth* d) {
  if (!d) return 0;
  if (d->early.active) { d->early.active = 0; (void)hipEventSynchronize(d->prep_ev); }
  (void)hipStreamSynchronize(d->ctx->stream);
  if (d->prep_ev) (void)hipEventSynchronize(d->prep_ev);   // a half / a whole job started ahead on another stream writes these buffers
  depth_free_images(d);
  void* dv[] = {d->d_pts, d->d_cnt, d->d_d0, d->d_scratch, d->d_matched, d->d_lmstate, d->d_part_e, d->d_part_n, d->d_counts, d->d_xbuf,
                d->d_gave_up, d->d_compact};
  for (void* q : dv) if (q) (void)hipFree(q);
  (void)hipHostFree(d->h_stats); (void)hipHostFree(d->h_prog);
  if (d->prep_ev) (void)hipEventDestroy(d->prep_ev);
  delete d;
  return 0;
}

static int depth_ensure(odo_depth* d, int rows, int cols) {
  if (d->rows == rows && d->cols == cols) return 0;
  if (d->prep_left) { HIP_OK(hipEventSynchronize(d->prep_ev)); d->prep_left = nullptr; }   // a prepared half on the old buffers is dropped
  // (a whole job started ahead has been settled by every caller before it gets here: depth_early_settle)
  HIP_OK(hipStreamSynchronize(d->ctx->stream));
  depth_free_images(d);
  const size_t n = (size_t)rows * cols;
  HIP_OK(hipMalloc((void**)&d->d_left, sizeof(float) * n));
  HIP_OK(hipMalloc((void**)&d->d_right, sizeof(float) * n));
  HIP_OK(hipMalloc((void**)&d->d_bl, sizeof(float) * n));
  HIP_OK(hipMalloc((void**)&d->d_br, sizeof(float) * n));
  HIP_OK(hipMalloc((void**)&d->d_disp, sizeof(float) * n));
  HIP_OK(hipMalloc((void**)&d->d_dep, sizeof(float) * n));
  HIP_OK(hipMalloc((void**)&d->d_val, n));
  d->rows = rows; d->cols = cols;
  return 0;
}

static int depth_check_size(const odo_depth* d, int rows, int cols) {
  if (!d->any_size && (rows != 376 || cols != 1241))  // ref: src/depth_estimate.cpp:46-49
    return fail("rows != 480 or cols != 640.");
  const int bw = (cols - 2 * d->boundary) / 32, bh = (rows - 2 * d->boundary) / 16;
  if (bw < 1 || bh < 1) return fail("depth: image too small for the 16x32 selection grid");
  if (bw * bh > kSelMaxElems) return fail("depth: selection block %dx%d exceeds %d pixels", bw, bh, kSelMaxElems);
  if (cols > 65535 || rows > 65535) return fail("depth: image too large");
  return 0;
}

// ComputeDepth as a resumable launch sequence on the estimator's stream, so a caller that is also feeding another
// stream (the tracker's pose LM) can interleave the two from ONE host thread:
//   depth_job_begin  : zero-fill, blur, point selection, disparity scan          (6 launches)
//   depth_job_pump   : issues at most one depth-LM launch per call, honouring the early-exit progress words and
//                      the run-ahead bound; returns true once the tail (filters, counts, stats copy) is enqueued.
struct DepthJob {
  const float *left, *right;
  uint8_t* val;
  float *disp, *dep;
  int rows, cols, stage;
  int k;          // next depth-LM launch index
  int n_launches;
  bool poll, tail_done;
  bool persistent;   // the depth LM + write-back of this job went out as ONE persistent launch
  std::chrono::steady_clock::time_point wait_since;
  bool waiting;
};

// The epipolar scan (depth_disparity_kernel: a wave per point slot). e0 / e1 (optional): dispatch-bound start / stop events
// (odo_depth_time_stages).
static void depth_launch_scan(const odo_depth* d, hipStream_t s, int rows, int cols, float* disp, float* dep, hipEvent_t e0 = nullptr,
                              hipEvent_t e1 = nullptr) {
  hipExtLaunchKernelGGL(depth_disparity_kernel, dim3(kSelBlocks * kSelCap / 4), dim3(256), 0, s, e0, e1, 0, (const float*)d->d_bl,
                        (const float*)d->d_br, rows, cols, d->boundary, d->max_disparity, d->ssd_th, d->K.f0, d->baseline,
                        (const uint32_t*)d->d_pts, (const int*)d->d_cnt, disp, dep, d->d_d0, d->d_matched);
}

static int depth_job_begin(odo_depth* d, DepthJob* j, const float* left, const float* right, int rows, int cols,
                           uint8_t* val, float* disp, float* dep, int stage, unsigned long long left_stamp = 0) {
  hipStream_t s = depth_stream(d);
  // a front half prepared ahead (odo_depth_prepare_left_dev): either it is this call's — the blurred left image and the selected
  // points are (or will be) there, behind prep_ev — or it is dropped; in both cases this stream goes on behind it (it writes the
  // estimator's buffers)
  bool prepared = false;
  if (d->prep_left) {
    HIP_OK(hipStreamWaitEvent(s, d->prep_ev, 0));
    prepared = stage == 2 && left_stamp != 0 && d->prep_left == left && d->prep_stamp == left_stamp && d->prep_rows == rows && d->prep_cols == cols;
    d->prep_left = nullptr;
  }
  const size_t n = (size_t)rows * cols;
  j->left = left; j->right = right; j->val = val; j->disp = disp; j->dep = dep;
  j->rows = rows; j->cols = cols; j->stage = stage;
  j->k = 0; j->n_launches = 0; j->poll = d->poll != 0; j->tail_done = false; j->waiting = false; j->persistent = false;
  // progress words may only be reset while this stream is idle: every ComputeDepth ends with depth_finish()'s sync
  d->h_prog[0] = 0; d->h_prog[1] = 0;
  (void)n;
  if (prepared) {
    // the right image's blur alone (as slice 0 of the launch: it carries the zero-fill of val / disp / dep); the selection has run
    // (without marking val: DepthOptimization's write-back sets every selected pixel's flag anyway, ref: :176-191)
    hipLaunchKernelGGL(blur3x3_kernel, grid2d(cols, rows, 1), dim3(256), 0, s, right, d->d_br, right, d->d_br, rows, cols, val, disp, dep);
  } else {
    // blur both images; the same launch zero-fills val / disp / dep
    hipLaunchKernelGGL(blur3x3_kernel, grid2d(cols, rows, 2), dim3(256), 0, s, left, d->d_bl, right, d->d_br, rows, cols, val,
                       disp, dep);
    hipLaunchKernelGGL(depth_select_kernel, dim3(kSelBlocks), dim3(kSelThreads), 0, s, d->d_bl, rows, cols, d->boundary,
                       d->grad_th, val, d->d_pts, d->d_cnt);
  }
  depth_launch_scan(d, s, rows, cols, disp, dep);
  HIP_OK(hipGetLastError());
  return 0;
}

// Tail, part 1: write-back + filters. Part 2 (depth_job_stats) reduces the counts into host-mapped memory and sets
// the completion word; a caller may enqueue more work of its own between the two (the tracker's pyramids), so that
// "completion word set" implies that work is done too.
static int depth_job_tail(odo_depth* d, DepthJob* j) {
  hipStream_t s = depth_stream(d);
  const int run_lm = j->stage != 1 ? 1 : 0;
  hipLaunchKernelGGL(depth_finalize_kernel, dim3(kDlmBlocks), dim3(kDlmBlock), 0, s, run_lm, j->cols, d->d_pts, d->d_cnt,
                     d->d_matched, d->d_scratch, d->photo_th, d->min_depth, d->max_depth, j->val, j->dep, d->d_counts);
  HIP_OK(hipGetLastError());
  j->tail_done = true;
  return 0;
}
static int depth_job_stats(odo_depth* d, DepthJob* j) {
  const int run_lm = j->stage != 1 ? 1 : 0;
  d->token++;
  hipLaunchKernelGGL(depth_stats_kernel, dim3(1), dim3(kDlmBlock), 0, depth_stream(d), run_lm, j->n_launches, d->d_counts,
                     d->d_lmstate, d->d_stats_map, d->d_prog + 4, d->token, j->persistent ? d->d_gave_up : (int*)nullptr);
  HIP_OK(hipGetLastError());
  return 0;
}

// Returns 1 when the whole job has been enqueued, 0 when there is more to do (call again), -1 on error.
// DepthOptimization + write-back in one persistent launch (depth_lm_persistent_kernel); depth_job_stats follows as for the step launches.
static int depth_job_persistent(odo_depth* d, DepthJob* j) {
  hipStream_t s = depth_stream(d);
  if ((++d->persist_epoch & 0xffu) == 0u)   // the pairs' tags carry the low byte of the epoch: cleared whenever it starts over
    HIP_OK(hipMemsetAsync(d->d_xbuf, 0, sizeof(unsigned long long) * 2 * kDlmBlocks * 2, s));
  DepthPersistArgs a;
  memset(&a, 0, sizeof(a));
  a.left = j->left; a.right = j->right; a.cols = j->cols; a.pts = d->d_pts; a.cnt = d->d_cnt; a.d0 = d->d_d0; a.matched = d->d_matched;
  a.state_out = d->d_lmstate;   // [0]: depth_job_stats reads state[n_launches & 1] with n_launches = 0
  a.tx = d->baseline; a.fx = d->K.f0; a.huber_delta = d->huber_delta; a.lambda0 = d->lambda; a.precision = d->precision;
  a.max_iters = d->max_iters; a.photo_th = d->photo_th; a.min_depth = d->min_depth; a.max_depth = d->max_depth;
  a.val = j->val; a.dep = j->dep; a.counts = d->d_counts; a.xbuf = d->d_xbuf; a.epoch = d->persist_epoch; a.wait_ticks = d->persist_wait;
  a.gave_up = d->d_gave_up; a.fault = d->persist_fault; a.home = d->persist_home;
  static const int depth_cls = getenv("ODO_DEPTH_CLASS") ? (atoi(getenv("ODO_DEPTH_CLASS")) & 7) : 4;   // (the pose LM's launch: class 0)
  a.cls = depth_cls;
  static unsigned long long* dbg_buf = [] {
    unsigned long long* p = nullptr;
    if (getenv("ODO_DEPTH_STAMPS")) {
      if (!ODO_PHASE_STAMPS) fprintf(stderr, "odometry_hip: ODO_DEPTH_STAMPS needs the diagnostic build (python -m odometry_amd.build --stamps)\n");
      else if (hipHostMalloc((void**)&p, 256, hipHostMallocMapped) == hipSuccess) memset(p, 0, 256);
    }
    return p;
  }();
  a.dbg = dbg_buf;
  if (dbg_buf && dbg_buf[5] > 0 && dbg_buf[5] % 100 == 0)
    fprintf(stderr, "[depth stamps] per iteration: gather %.0f decide + update %.0f evaluate %.0f sums + publish %.0f cycles; iterations/launch "
            "%.2f, loop cycles/launch %.0f, same-XCD launches %.0f %%\n", (double)dbg_buf[0] / dbg_buf[4], (double)dbg_buf[1] / dbg_buf[4],
            (double)dbg_buf[2] / dbg_buf[4], (double)dbg_buf[3] / dbg_buf[4], (double)dbg_buf[4] / dbg_buf[5], (double)dbg_buf[7] / dbg_buf[5],
            100.0 * (double)dbg_buf[6] / dbg_buf[5]);
  hipLaunchKernelGGL(depth_lm_persistent_kernel, dim3(8 * kDpK), dim3(kDpThreads), 0, s, a);
  HIP_OK(hipGetLastError());
  j->persistent = true;
  j->n_launches = 0;
  j->tail_done = true;
  return 0;
}

static int depth_job_pump(odo_depth* d, DepthJob* j) {
  if (j->tail_done) return 1;
  if (j->stage != 1 && j->k == 0 && d->persist && !d->persist_off_once && d->max_iters <= kDpMaxIters)
    return depth_job_persistent(d, j) ? -1 : 1;
  volatile int* prog = d->h_prog;
  // launch k decides on evaluation k-1 and runs evaluation k: max_iters evaluations need max_iters + 1 launches
  bool lm_over = (j->stage == 1) || (j->k > d->max_iters) || (j->poll && prog[1]);
  if (!lm_over) {
    if (j->poll && j->k - prog[0] > d->run_ahead) {  // device is behind: do not queue more yet
      const auto now = std::chrono::steady_clock::now();
      if (!j->waiting) { j->waiting = true; j->wait_since = now; }
      else if (now - j->wait_since > std::chrono::seconds(2)) j->poll = false;  // never hang on a lost progress word
      return 0;
    }
    j->waiting = false;
    hipLaunchKernelGGL(depth_lm_step_kernel, dim3(kDlmBlocks), dim3(kDlmBlock), 0, depth_stream(d), j->k, j->left, j->right,
                       j->cols, d->d_pts, d->d_cnt, d->d_d0, d->d_scratch, d->d_lmstate, d->d_part_e, d->d_part_n,
                       d->baseline, d->K.f0, d->huber_delta, d->lambda, d->precision, d->max_iters, d->d_prog);
    j->k++;
    j->n_launches++;
    return 0;
  }
  return depth_job_tail(d, j) ? -1 : 1;
}

// Enqueues the whole ComputeDepth (stage 2) or only the disparity stage (stage 1) on device pointers.
static int depth_run(odo_depth* d, const float* left, const float* right, int rows, int cols, uint8_t* val, float* disp,
                     float* dep, int stage, unsigned long long left_stamp = 0) {
  DepthJob j;
  if (depth_job_begin(d, &j, left, right, rows, cols, val, disp, dep, stage, left_stamp)) return -1;
  for (;;) {
    const int r = depth_job_pump(d, &j);
    if (r < 0) return -1;
    if (r > 0) return depth_job_stats(d, &j);
  }
}

// Waits for the job: spins on the host-mapped completion word (bounded), or synchronises the stream when the caller
// needs every queued operation retired (host-buffer entry points copy results back afterwards).
static int depth_finish(odo_depth* d, bool full_sync = true) {
  if (full_sync) {
    HIP_OK(hipStreamSynchronize(d->ctx->stream));
  } else {
    volatile int* done = d->h_prog + 4;
    const auto t0 = std::chrono::steady_clock::now();
    while (done[0] != d->token) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIP_OK(hipStreamSynchronize(d->ctx->stream)); break; }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  d->last = *d->h_stats;
  if (d->last.status == -2) {
    // the persistent depth-LM launch gave up (its workgroups could not all be resident within the wait bound): the caller runs the
    // job again — on the step launches, which need no co-residency (depth_run_checked / tracker_job_run)
    d->persist_bails++;
    if (getenv("ODO_LOG_GIVEUPS")) {
      int dg[512];
      (void)hipMemcpy(dg, d->d_gave_up, sizeof(dg), hipMemcpyDeviceToHost);
      (void)hipMemset(d->d_gave_up + 1, 0, sizeof(int) * 511);
      {   // when each workgroup entered and left (us after the first entry) and the iteration it was in (-1: never placed)
        unsigned t0 = 0xffffffffu;
        for (int i = 0; i < 80; i++) if ((unsigned)dg[16 + 4 * i + 1] < t0 && dg[16 + 4 * i + 2] != 0) t0 = (unsigned)dg[16 + 4 * i + 1];
        fprintf(stderr, "[odometry_hip]    workgroup: iteration, entry us, exit us, XCC:");
        for (int i = 0; i < 80; i += 1)
          if (i < 6 || i >= 74 || dg[16 + 4 * i] != dg[16 + 4 * 40]) fprintf(stderr, " %d: %d %.0f %.0f %d;", i, dg[16 + 4 * i], ((unsigned)dg[16 + 4 * i + 1] - t0) * 0.01, ((unsigned)dg[16 + 4 * i + 2] - t0) * 0.01, dg[16 + 4 * i + 3]);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[odometry_hip] depth-LM persistent launch gave up (launch epoch %u, %d clean jobs before): %d of 80 workgroups, e.g. workgroup %d "
              "on XCC %d, %s\n", d->persist_epoch, d->persist_clean, dg[4], dg[1], dg[2], dg[3] ? "waiting for an iteration's sums" : "waiting for the others to be placed");
    }
    if (fine_note_giveup(&d->persist_strikes, &d->persist_clean, &d->persist_offs)) d->persist = 0;
    d->persist_off_once = 1;
    return 2;
  }
  if (d->persist_cfg) {
    const bool used = d->persist && !d->persist_off_once;
    if (getenv("ODO_LOG_GIVEUPS")) {   // (diagnostic) how many jobs took which path, at process exit
      static long n_persist = 0, n_step = 0;
      static bool hooked = false;
      static long* counts[2] = {&n_persist, &n_step};
      (used ? n_persist : n_step)++;
      if (!hooked) { hooked = true; atexit([] { fprintf(stderr, "[odometry_hip] depth jobs: %ld on the persistent launch, %ld on step launches\n", *counts[0], *counts[1]); }); }
    }
    d->persist_off_once = 0;
    if (fine_note_clean(used, &d->persist_strikes, &d->persist_clean, &d->persist_offs)) d->persist = 1;
  }
  if (d->last.status != 0) return fail("number of valid after optimization is too small: %d", d->last.n_valid);
  return 0;
}

// A whole job started ahead (odo_depth_compute_begin_dev) that nobody picked up: it is waited for — it owns the estimator's buffers
// and its progress words — and its result dropped.
static int depth_early_settle(odo_depth* d) {
  if (!d->early.active) return 0;
  d->early.active = 0;
  HIP_OK(hipStreamWaitEvent(d->ctx->stream, d->prep_ev, 0));
  (void)depth_finish(d, false);
  return 0;
}

static int depth_host(odo_depth* d, const float* left, const float* right, int rows, int cols, uint8_t* val, float* disp,
                      float* dep, int stage) {
  if (!d || !left || !right || !val || !disp || !dep) return fail("depth: NULL arg");
  if (depth_check_size(d, rows, cols)) return -1;
  HIP_OK(hipSetDevice(d->ctx->device));
  if (depth_early_settle(d)) return -1;
  if (depth_ensure(d, rows, cols)) return -1;
  hipStream_t s = d->ctx->stream;
  const size_t n = (size_t)rows * cols;
  if (upload_rows_async(d->ctx, d->d_left, left, sizeof(float) * (size_t)cols, sizeof(float) * (size_t)cols, rows)) return -1;
  if (upload_rows_async(d->ctx, d->d_right, right, sizeof(float) * (size_t)cols, sizeof(float) * (size_t)cols, rows)) return -1;
  if (depth_run(d, d->d_left, d->d_right, rows, cols, d->d_val, d->d_disp, d->d_dep, stage)) return -1;
  int rc = depth_finish(d);
  if (rc == 2) {   // the persistent launch gave up: the same job again on the step launches
    if (depth_run(d, d->d_left, d->d_right, rows, cols, d->d_val, d->d_disp, d->d_dep, stage)) return -1;
    rc = depth_finish(d);
  }
  (void)s;
  // the outputs reach the caller whatever the status: the reference has written left_val / left_dep back (ref: src/depth_estimate.cpp:
  // 176-191) before it returns -1 for "number of valid after optimization is too small" (:192-194)
  {   // the three outputs: queued back to back, one wait
    void* const dsts[3] = {val, disp, dep};
    const void* const srcs[3] = {d->d_val, d->d_disp, d->d_dep};
    const size_t sizes[3] = {n, sizeof(float) * n, sizeof(float) * n};
    if (copy_many_to_user_host(d->ctx, 3, dsts, srcs, sizes)) return -1;
  }
  return rc == 0 ? 0 : -1;
}

extern "C" int odo_depth_compute(odo_depth* d, const float* left, const float* right, int rows, int cols, uint8_t* val,
                                 float* disp, float* dep) {
  return depth_host(d, left, right, rows, cols, val, disp, dep, 2);
}
extern "C" int odo_depth_disparity(odo_depth* d, const float* left, const float* right, int rows, int cols, uint8_t* val,
                                   float* disp, float* dep) {
  return depth_host(d, left, right, rows, cols, val, disp, dep, 1);
}
extern "C" int odo_depth_prepare_left_dev(odo_depth* d, odo_ctx* side, const float* left_dev, int rows, int cols, unsigned long long stamp) {
  return odo_depth_prepare_left_dev_marked(d, side, left_dev, rows, cols, stamp, 0);
}
extern "C" int odo_depth_prepare_left_dev_marked(odo_depth* d, odo_ctx* side, const float* left_dev, int rows, int cols,
                                                 unsigned long long stamp, unsigned long mark) {
  if (!d || !side || !left_dev || stamp == 0) return fail("odo_depth_prepare_left_dev: bad arg");
  if (side->device != d->ctx->device) return fail("odo_depth_prepare_left_dev: the two contexts must be on one device");
  if (depth_check_size(d, rows, cols)) return -1;
  HIP_OK(hipSetDevice(d->ctx->device));
  if (depth_early_settle(d)) return -1;
  if (d->prep_left) HIP_OK(hipStreamWaitEvent(side->stream, d->prep_ev, 0));   // (a prepared half nobody picked up: one at a time)
  if (depth_ensure(d, rows, cols)) return -1;
  // what produced left_dev (an upload), and the estimator's previous call, come first
  if (mark ? odo_ctx_stream_wait_mark(side, d->ctx, mark) : odo_ctx_stream_wait(side, d->ctx)) return -1;
  hipLaunchKernelGGL(blur3x3_kernel, grid2d(cols, rows, 1), dim3(256), 0, side->stream, left_dev, d->d_bl, left_dev, d->d_bl, rows, cols,
                     (uint8_t*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(depth_select_kernel, dim3(kSelBlocks), dim3(kSelThreads), 0, side->stream, (const float*)d->d_bl, rows, cols, d->boundary,
                     d->grad_th, (uint8_t*)nullptr, d->d_pts, d->d_cnt);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(d->prep_ev, side->stream));
  d->prep_left = left_dev; d->prep_stamp = stamp; d->prep_rows = rows; d->prep_cols = cols;
  return 0;
}
extern "C" int odo_depth_compute_dev_stamped(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols,
                                             uint8_t* val_dev, float* disp_dev, float* dep_dev, unsigned long long left_stamp) {
  if (!d || !left_dev || !right_dev || !val_dev || !disp_dev || !dep_dev) return fail("depth: NULL arg");
  if (depth_check_size(d, rows, cols)) return -1;
  HIP_OK(hipSetDevice(d->ctx->device));
  if (depth_early_settle(d)) return -1;
  if (depth_ensure(d, rows, cols)) return -1;
  if (depth_run(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, 2, left_stamp)) return -1;
  int rc = depth_finish(d, false);
  if (rc == 2) {   // the persistent launch gave up: the same job again on the step launches
    if (depth_run(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, 2)) return -1;
    rc = depth_finish(d, false);
  }
  return rc == 0 ? 0 : -1;
}
// The whole of ComputeDepth(left, right) enqueued on `side`'s stream, without waiting: the call returns once the launches are out.
// Returns 0: started; 1: not started (the depth LM would need host-paced step launches: persistent launch off / switched off);
// -1: error. odo_depth_compute_end_dev with the same arguments collects it.
// ---- sparse hand-over of the three output images (callers that own host memory: cv::Mat) ----
constexpr size_t kCompactSlots = (size_t)kSelBlocks * kSelCap;
extern "C" size_t odo_depth_compact_bytes(void) { return kCompactSlots * 13; }
extern "C" int odo_depth_compact_outputs_async(odo_depth* d, odo_ctx* on, const uint8_t* val_dev, const float* disp_dev,
                                               const float* dep_dev, int cols, void* dst_pinned) {
  if (!d || !on || !val_dev || !disp_dev || !dep_dev || !dst_pinned || cols < 1) return fail("odo_depth_compact_outputs_async: bad arg");
  if (!host_is_pinned(dst_pinned, kCompactSlots * 13)) return fail("odo_depth_compact_outputs_async: dst is not an odo_host_alloc block of odo_depth_compact_bytes()");
  HIP_OK(hipSetDevice(d->ctx->device));
  if (!d->d_compact) HIP_OK(hipMalloc(&d->d_compact, kCompactSlots * 13));
  char* c = (char*)d->d_compact;
  hipLaunchKernelGGL(depth_compact_outputs_kernel, dim3((unsigned)((kCompactSlots + 255) / 256)), dim3(256), 0, on->stream,
                     (const uint32_t*)d->d_pts, (const int*)d->d_cnt, cols, val_dev, disp_dev, dep_dev, (uint32_t*)c,
                     (float*)(c + kCompactSlots * 4), (float*)(c + kCompactSlots * 8), (uint8_t*)(c + kCompactSlots * 12));
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(dst_pinned, d->d_compact, kCompactSlots * 13, hipMemcpyDeviceToHost, on->stream));
  return 0;
}
static int host_scatter_outputs(const void* compact, int rows, int cols, uint8_t* val, size_t val_pitch, float* disp, size_t disp_pitch,
                                float* dep, size_t dep_pitch, unsigned long long* dep_fingerprint, bool zero_first) {
  if (!compact || !val || !disp || !dep || rows < 1 || cols < 1 || val_pitch < (size_t)cols || disp_pitch < sizeof(float) * (size_t)cols ||
      dep_pitch < sizeof(float) * (size_t)cols)
    return fail("odo_host_scatter_outputs: bad arg");
  const char* c = (const char*)compact;
  const uint32_t* idx = (const uint32_t*)c;
  const float* cd = (const float*)(c + kCompactSlots * 4);
  const float* cp = (const float*)(c + kCompactSlots * 8);
  const uint8_t* cv = (const uint8_t*)(c + kCompactSlots * 12);
  if (zero_first)
    for (int y = 0; y < rows; y++) {
      memset(val + (size_t)y * val_pitch, 0, (size_t)cols);
      memset((char*)disp + (size_t)y * disp_pitch, 0, sizeof(float) * (size_t)cols);
      memset((char*)dep + (size_t)y * dep_pitch, 0, sizeof(float) * (size_t)cols);
    }
  const uint32_t npx = (uint32_t)rows * (uint32_t)cols;
  for (size_t s = 0; s < kCompactSlots; s++) {
    const uint32_t i = idx[s];
    if (i >= npx) continue;
    const uint32_t y = i / (uint32_t)cols, x = i - y * (uint32_t)cols;
    val[(size_t)y * val_pitch + x] = cv[s];
    *(float*)((char*)disp + (size_t)y * disp_pitch + (size_t)x * 4) = cd[s];
    *(float*)((char*)dep + (size_t)y * dep_pitch + (size_t)x * 4) = cp[s];
  }
  if (dep_fingerprint) *dep_fingerprint = hostfp::image(dep, dep_pitch, sizeof(float) * (size_t)cols, rows, nullptr, 0);
  return 0;
}
extern "C" int odo_host_scatter_outputs(const void* compact, int rows, int cols, uint8_t* val, size_t val_pitch, float* disp,
                                        size_t disp_pitch, float* dep, size_t dep_pitch, unsigned long long* dep_fingerprint) {
  return host_scatter_outputs(compact, rows, cols, val, val_pitch, disp, disp_pitch, dep, dep_pitch, dep_fingerprint, true);
}
extern "C" int odo_host_scatter_outputs_prezeroed(const void* compact, int rows, int cols, uint8_t* val, size_t val_pitch, float* disp,
                                                  size_t disp_pitch, float* dep, size_t dep_pitch, unsigned long long* dep_fingerprint) {
  return host_scatter_outputs(compact, rows, cols, val, val_pitch, disp, disp_pitch, dep, dep_pitch, dep_fingerprint, false);
}

extern "C" int odo_depth_compute_begin_dev(odo_depth* d, odo_ctx* side, const float* left_dev, const float* right_dev, int rows, int cols,
                                           uint8_t* val_dev, float* disp_dev, float* dep_dev, unsigned long long left_stamp,
                                           unsigned long long right_stamp, unsigned long mark) {
  if (!d || !side || !left_dev || !right_dev || !val_dev || !disp_dev || !dep_dev || !left_stamp || !right_stamp)
    return fail("odo_depth_compute_begin_dev: bad arg");
  if (side->device != d->ctx->device) return fail("odo_depth_compute_begin_dev: the two contexts must be on one device");
  if (depth_check_size(d, rows, cols)) return -1;
  HIP_OK(hipSetDevice(d->ctx->device));
  if (depth_early_settle(d)) return -1;
  if (!(d->persist && !d->persist_off_once && d->max_iters <= kDpMaxIters)) return 1;
  if (d->prep_left) { HIP_OK(hipStreamWaitEvent(side->stream, d->prep_ev, 0)); d->prep_left = nullptr; }
  if (depth_ensure(d, rows, cols)) return -1;
  // what produced the two images and the three blocks, and the estimator's previous call, come first
  if (mark ? odo_ctx_stream_wait_mark(side, d->ctx, mark) : odo_ctx_stream_wait(side, d->ctx)) return -1;
  d->job_stream = side->stream;
  const int rc = depth_run(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, 2);
  d->job_stream = nullptr;
  if (rc) return -1;
  HIP_OK(hipEventRecord(d->prep_ev, side->stream));
  d->early.active = 1;
  d->early.left = left_dev; d->early.right = right_dev; d->early.left_stamp = left_stamp; d->early.right_stamp = right_stamp;
  d->early.rows = rows; d->early.cols = cols; d->early.val = val_dev; d->early.disp = disp_dev; d->early.dep = dep_dev;
  return 0;
}
// ComputeDepth on device pointers with both images' content stamps: the job started ahead with exactly these arguments is waited for
// (the estimator's stream is ordered behind it); anything else — nothing started, other images, other blocks — is computed now.
// Same results either way (same launches, earlier).
extern "C" int odo_depth_compute_end_dev(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols,
                                         uint8_t* val_dev, float* disp_dev, float* dep_dev, unsigned long long left_stamp,
                                         unsigned long long right_stamp) {
  if (!d || !left_dev || !right_dev || !val_dev || !disp_dev || !dep_dev) return fail("depth: NULL arg");
  const odo_depth::Early& e = d->early;
  if (!(e.active && left_stamp && right_stamp && e.left == left_dev && e.right == right_dev && e.left_stamp == left_stamp &&
        e.right_stamp == right_stamp && e.rows == rows && e.cols == cols && e.val == val_dev && e.disp == disp_dev && e.dep == dep_dev))
    return odo_depth_compute_dev_stamped(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, left_stamp);
  HIP_OK(hipSetDevice(d->ctx->device));
  d->early.active = 0;
  HIP_OK(hipStreamWaitEvent(d->ctx->stream, d->prep_ev, 0));
  int rc = depth_finish(d, false);
  if (rc == 2) {   // the persistent launch gave up: the same job again on the step launches, on the estimator's own stream
    if (depth_run(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, 2)) return -1;
    rc = depth_finish(d, false);
  }
  return rc == 0 ? 0 : -1;
}
// 1 while a job started by odo_depth_compute_begin_dev has not been collected.
extern "C" int odo_depth_early_pending(const odo_depth* d) { return d && d->early.active ? 1 : 0; }
extern "C" int odo_depth_compute_dev(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols,
                                     uint8_t* val_dev, float* disp_dev, float* dep_dev) {
  if (!d || !left_dev || !right_dev || !val_dev || !disp_dev || !dep_dev) return fail("depth: NULL arg");
  if (depth_check_size(d, rows, cols)) return -1;
  HIP_OK(hipSetDevice(d->ctx->device));
  if (depth_early_settle(d)) return -1;
  if (depth_ensure(d, rows, cols)) return -1;
  if (depth_run(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, 2)) return -1;
  // outputs stay on the device: only the statistics are needed back, and they arrive through host-mapped memory behind the
  // last launch of the job (no stream synchronisation)
  int rc = depth_finish(d, false);
  if (rc == 2) {   // the persistent launch gave up: the same job again on the step launches
    if (depth_run(d, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, 2)) return -1;
    rc = depth_finish(d, false);
  }
  return rc == 0 ? 0 : -1;
}

extern "C" int odo_depth_persistent_stats(const odo_depth* d, int* on, int* fallbacks) {
  if (!d) return fail("NULL depth estimator");
  if (on) *on = (d->persist && d->max_iters <= kDpMaxIters) ? 1 : 0;
  if (fallbacks) *fallbacks = d->persist_bails;
  return 0;
}
extern "C" int odo_depth_report(const odo_depth* d, int* iters, float* cost, int* n_selected, int* n_matched, int* n_valid) {
  if (!d) return fail("NULL depth estimator");
  if (iters) *iters = d->last.iters;
  if (cost) *cost = d->last.cost;
  if (n_selected) *n_selected = d->last.n_selected;
  if (n_matched) *n_matched = d->last.n_matched;
  if (n_valid) *n_valid = d->last.n_valid;
  return 0;
}

#include "tracker.hip.h"
#include "batch.hip.h"
#include "gather.hip.h"
#include "camera.hip.h"
