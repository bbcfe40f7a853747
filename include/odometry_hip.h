/*
 * odometry_hip.h — C ABI of the MI355X-native photometric-LM tracking hot path.
 *
 * The reference (WangYuTum/odometry) has no FFI layer; its boundary is the public C++ surface the
 * runner uses (run_odometry_kitti_offline.cpp:68-70,88,102,130-131,205,215,229,251-252,261,268).
 * Each entry point below names the reference interface it replaces. The C++ shim classes in
 * include/odometry_shim.hpp keep the reference's class names and signatures over this ABI.
 *
 * Conventions: plain C, no exceptions; every call returns 0 on success and -1 on failure
 * (OptimizerStatus / GlobalStatus, ref: include/data_types.h:27-28) and odo_last_error() describes
 * the last failure of the calling thread. Poses are 16 fp32 COLUMN-major (Eigen Affine4f,
 * ref: include/data_types.h:24). Images are single-channel fp32 row-major (CV_32F, ref: data_types.h:10-12).
 * The caller owns all host buffers; the library owns device memory behind the opaque handles.
 * One odo_ctx = one HIP stream on one device; handles are not thread-safe (the reference is strictly
 * single-threaded, ref: run_odometry_kitti_offline.cpp:3).
 * There is NO CPU fallback: without a HIP device odo_ctx_create fails.
 */
#ifndef ODOMETRY_HIP_H
#define ODOMETRY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct odo_ctx odo_ctx;
typedef struct odo_pyr odo_pyr;
typedef struct odo_lm odo_lm;
typedef struct odo_depth odo_depth;
typedef struct odo_tracker odo_tracker;
typedef struct odo_camera odo_camera;

/* Level-0 pinhole intrinsics (fy = fx). NULL wherever accepted = the KITTI-00 constants the reference
 * hard-codes (ref: include/image_processing_global.h:35-36: 718.856f, 607.1928, 185.2157). */
typedef struct {
  float f0, cx0, cy0;
} odo_intrinsics;

enum { ODO_PYR_IMAGE = 0, ODO_PYR_DEPTH = 1 };
enum { ODO_MAX_LEVELS = 8, ODO_NACC = 29 };

const char* odo_last_error(void);
int odo_version(void);

/* ---- context ------------------------------------------------------------------------------
 * One HIP stream + its bookkeeping. A context is meant for one host thread (the reference is single-threaded, ref:
 * run_odometry_kitti_offline.cpp:3): launches of two threads on the same context would interleave on its stream. Its device-block
 * free list, pinned staging ring and upload tickets are mutex-guarded, so allocation / upload / release from a second thread
 * (e.g. a Mat destroyed elsewhere) is safe. */
int odo_ctx_create(int device, odo_ctx** out);
/* Same, on a high-priority HIP stream (its hardware queue comes from a pool of its own): for a latency-critical chain of
 * dependent launches that must not queue behind other streams' work. Used by odo_tracker for the LM stream. */
int odo_ctx_create_high_priority(int device, odo_ctx** out);
int odo_ctx_destroy(odo_ctx* ctx);
int odo_ctx_synchronize(odo_ctx* ctx);
/* HIP-event timing on the context's own stream (bench.py): record start / stop around a region,
 * then read the elapsed milliseconds (synchronises on the stop event). */
int odo_ctx_timer_start(odo_ctx* ctx);
int odo_ctx_timer_stop(odo_ctx* ctx, float* elapsed_ms);
/* Device scratch for callers that keep inputs resident in HBM (bench.py, tracker). */
int odo_dev_alloc(odo_ctx* ctx, size_t bytes, void** out_dev);
int odo_dev_free(odo_ctx* ctx, void* dev);
int odo_dev_upload(odo_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int odo_dev_download(odo_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* The drop-in path (include/odometry_shim.hpp) without stalls: page-locked host blocks (an upload from one is a plain
 * asynchronous DMA; uploads from other host memory are copied once into the context's pinned staging ring), uploads that
 * return as soon as the caller may reuse its buffer instead of waiting for the device, and device blocks recycled through a
 * free list of the context, whose release neither synchronises the stream nor calls the driver (is_async, as returned by
 * the allocation, and the size go back into the matching free). Recycled blocks are for work on the context's own stream
 * only: a block may be handed out again while work that used it is still queued on that stream. A block from
 * odo_host_alloc must not be freed while uploads from it may be pending (odo_ctx_synchronize first). */
void* odo_host_alloc(size_t bytes);
void odo_host_free(void* host);
int odo_dev_upload_async(odo_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
/* The same for an image whose rows lie src_pitch bytes apart on the host (a cv::Mat view: step > cols * elemSize): `rows` rows of
 * row_bytes bytes land densely packed at dst_dev. */
int odo_dev_upload_2d_async(odo_ctx* ctx, void* dst_dev, const void* src_host, size_t src_pitch, size_t row_bytes, int rows);
/* Upload tickets: odo_dev_upload_async from a page-locked block (odo_host_alloc) is a DMA that reads the block in place after
 * the call has returned. odo_ctx_upload_ticket returns the ticket of the most recent such upload (monotonic, 0 = none);
 * odo_ctx_upload_wait(ticket) returns once that upload and all earlier ones no longer read host memory — the moment the block
 * may be rewritten or released. Waiting for a retired ticket costs nothing. */
unsigned long odo_ctx_upload_ticket(odo_ctx* ctx);
int odo_ctx_upload_wait(odo_ctx* ctx, unsigned long ticket);
/* Orders `waiter`'s stream behind everything queued on `signaller`'s stream so far (an event, no host wait): two contexts of one
 * host thread — the drop-in classes keep a second one for work that need not wait for the pose LM (an upload started early). */
int odo_ctx_stream_wait(odo_ctx* waiter, odo_ctx* signaller);
/* The same in two halves: odo_ctx_mark names the point `ctx`'s stream has been filled up to (returns a non-zero mark), and
 * odo_ctx_stream_wait_mark orders `waiter` behind that point — not behind what was queued on `signaller` since (a mark older
 * than the last 32 falls back to "everything queued so far"). */
unsigned long odo_ctx_mark(odo_ctx* ctx);
int odo_ctx_stream_wait_mark(odo_ctx* waiter, odo_ctx* signaller, unsigned long mark);
/* Host waits until `ctx`'s stream has passed `mark` (odo_ctx_mark). */
int odo_ctx_wait_mark(odo_ctx* ctx, unsigned long mark);
/* Has the stream passed that mark? 1 yes, 0 not yet (never blocks), -1 on error. */
int odo_ctx_mark_reached(odo_ctx* ctx, unsigned long mark);
/* Images in host memory the library cannot watch — a cv::Mat (ref: run_odometry_kitti_offline.cpp:200,334-359 refills the same two
 * Mats every frame; the classes of include/odometry_shim.hpp see them three times per frame, :205 / :229 / :251). A device mirror
 * of such an image may be reused only if the bytes are still the ones that were uploaded:
 *   odo_host_fingerprint       64-bit fingerprint of all pixels (rows of row_bytes bytes, src_pitch apart; one read of the image);
 *   odo_dev_upload_fp_async    odo_dev_upload_2d_async through the pinned staging ring, *fp = the fingerprint of exactly the bytes
 *                              that were staged (fused with the copy; the caller may rewrite the image once the call has returned);
 *   odo_dev_download_async     device -> a block from odo_host_alloc, asynchronous on ctx's stream (ComputeDepth's outputs are staged
 *                              on the host beside the pose LM, ref: src/depth_estimate.cpp:33-78 fills left_val / left_disp / left_dep);
 *   odo_host_copy_fingerprint  host -> host copy (staging block -> the caller's cv::Mat), returns the fingerprint of what was written.
 * Not cryptographic: it guards against a caller rewriting its buffer, not against an adversary. No device needed for the two
 * host-only calls. */
unsigned long long odo_host_fingerprint(const void* src_host, size_t src_pitch, size_t row_bytes, int rows);
unsigned long long odo_host_copy_fingerprint(void* dst_host, size_t dst_pitch, const void* src_host, size_t src_pitch, size_t row_bytes,
                                             int rows);
int odo_dev_upload_fp_async(odo_ctx* ctx, void* dst_dev, const void* src_host, size_t src_pitch, size_t row_bytes, int rows,
                            unsigned long long* fp);
int odo_dev_download_async(odo_ctx* ctx, void* dst_pinned_host, const void* src_dev, size_t bytes);
int odo_dev_alloc_async(odo_ctx* ctx, size_t bytes, void** out_dev, int* is_async);
int odo_dev_free_async(odo_ctx* ctx, void* dev, size_t bytes, int is_async);

/* ---- pyramids ------------------------------------------------------------------------------
 * Replaces ImagePyramid::ImagePyramid / DepthPyramid::DepthPyramid
 * (ref: include/image_pyramid.h:24,51; src/image_pyramid.cpp:13-19,30-37;
 *  src/image_processing_global.cpp:12-56,58-113).
 * kind IMAGE: L0 = 3x3 Gaussian blur if smooth else copy; L1 = pyrDown(input); Lk = pyrDown(L(k-1)).
 * kind DEPTH: L0 = copy, or cv::medianBlur 3x3 (replicated border) if smooth (ref: src/image_processing_global.cpp:76-80; no
 * reference caller passes 1); Lk(y,x) = L(k-1)(2y+1,2x+1).
 * `img` is a host pointer with row pitch stride_bytes (0 = cols*4). */
int odo_pyramid_create(odo_ctx* ctx, const float* img, int rows, int cols, size_t stride_bytes, int levels,
                       int smooth, int kind, odo_pyr** out);
/* Same, input already resident in device memory (dense, row pitch = cols*4). */
int odo_pyramid_create_dev(odo_ctx* ctx, const float* img_dev, int rows, int cols, int levels, int smooth,
                           int kind, odo_pyr** out);
/* Rebuild an existing pyramid in place from a new device-resident image of the same size. */
int odo_pyramid_rebuild_dev(odo_pyr* pyr, const float* img_dev, int smooth);
/* GetNumberLevels / GetPyramidImage / GetPyramidDepth (ref: include/image_pyramid.h:33,36,60,63). */
int odo_pyramid_levels(const odo_pyr* pyr);
int odo_pyramid_level_dims(const odo_pyr* pyr, int level, int* rows, int* cols);
int odo_pyramid_download(const odo_pyr* pyr, int level, float* dst_host);
const float* odo_pyramid_level_dev(const odo_pyr* pyr, int level);
int odo_pyramid_destroy(odo_pyr* pyr);

/* ---- pose optimiser --------------------------------------------------------------------------
 * Replaces LevenbergMarquardtOptimizer (ref: include/lm_optimizer.h:32,44,47,54;
 * src/lm_optimizer.cpp:19-41,54-69,73-160,163-264,364-405).
 * max_iters is indexed by pyramid level (ref: src/lm_optimizer.cpp:117). robust: 0 none, 1 Huber, 2 t-dist. */
int odo_lm_create(odo_ctx* ctx, float lambda, float precision, const int* max_iters, int n_levels,
                  const float init_colmajor[16], int robust, float huber_delta, const odo_intrinsics* K,
                  odo_lm** out);
/* Solve (ref: src/lm_optimizer.cpp:54-69): returns 0 and the keyframe->current pose, or -1 and the
 * pseudo-identity whose (3,3) element is 0 (ref: src/lm_optimizer.cpp:48-52,60-65). */
int odo_lm_solve(odo_lm* lm, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img,
                 float out_colmajor[16]);
/* Optional: starts the Solve that the next odo_lm_solve(lm, kf_img, kf_dep, cur_img) with the SAME pyramids will collect, and
 * returns without waiting (a caller that knows the inputs early — the next frame's pyramid, the initial pose set by Reset —
 * overlaps the head of the Solve with its own bookkeeping). Same launches, earlier: results are unchanged. Reset, or a Solve on
 * other pyramids, abandons it. Returns 0 started, 1 nothing started (this Solve does not use the fused pipeline), -1 error.
 * While a started Solve is in flight the optimiser's trace / report of the previous Solve are being overwritten. */
int odo_lm_solve_begin(odo_lm* lm, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img);
/* Host work for the time a Solve leaves the calling thread idle (ref: src/lm_optimizer.cpp:73-160 keeps the CPU busy for the whole
 * Solve; here the thread inside odo_lm_solve spins on the device's completion word for ~0.25 ms): `fn(arg)` is called over and over
 * from that wait loop, on the caller's thread, until the result is there — keep each call short (~10 us: the result is noticed between
 * calls). NULL removes it. Not for an optimiser owned by odo_tracker / odo_tracker_batch (they feed their depth stream from there). */
int odo_lm_set_idle_callback(odo_lm* lm, void (*fn)(void*), void* arg);
/* Keyframe-candidate point lists built ahead of the Solve that may need them (the point lists are what the validity test of
 * ComputeResidualJacobianNaive selects, ref: src/lm_optimizer.cpp:190-198; the runner promotes a frame to keyframe AFTER its Solve,
 * run_odometry_kitti_offline.cpp:258-260, so the first Solve against a new keyframe would build them in front of its first launch):
 * img / dep are the pyramids of a frame that may become the keyframe. The launches go to `side`'s stream behind `mark` of the
 * optimiser's stream (odo_ctx_mark; 0: behind everything queued there); the call returns at once. A later Solve on exactly these
 * pyramids adopts the lists by a buffer swap; any other Solve ignores them. One candidate at a time. Not for an optimiser owned
 * by a tracker. */
int odo_lm_candidate_begin(odo_lm* lm, odo_ctx* side, const odo_pyr* img, const odo_pyr* dep, unsigned long mark);
/* n independent Solves (n sequences, each with its own optimiser and pyramids) in the SAME launches: a single Solve is a
 * serial chain of short launches that leaves most of the chip idle, n chains side by side take the time of the longest.
 * Per-sequence arithmetic and launch order are those of odo_lm_solve: results are bit-identical to n separate calls. The
 * optimisers must share one context. out_colmajor: n x 16 floats; status[i] = 0 / -1 per sequence (a failed sequence gets
 * the pseudo-identity, like odo_lm_solve). Sequences that cannot take the fused point-list pipeline make the call fall back
 * to one Solve after the other. */
int odo_lm_solve_batch(int n, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                       const odo_pyr* const* cur_img, float* out_colmajor, int* status);
/* Reset (ref: src/lm_optimizer.cpp:373-382): new initial pose and lambda, statistics cleared. */
int odo_lm_reset(odo_lm* lm, const float init_colmajor[16], float lambda);
/* ShowReport data (ref: src/lm_optimizer.cpp:364-371). The reference never writes its statistics, so
 * `iters`/`cost` are always 0 there; here iters[l] = evaluations spent on level l in the last Solve and
 * cost[l][0..1] = mean weighted error at the first / last evaluation of level l. */
int odo_lm_report(const odo_lm* lm, int iters[4], float cost[4][2]);
int odo_lm_destroy(odo_lm* lm);

/* One ComputeResidualJacobianNaive + normal-equation pass (ref: src/lm_optimizer.cpp:163-264,129,145-149)
 * at pose T on `level`: acc[0..20] upper triangle of JtWJ (row-major), acc[21..26] JtWr, acc[27] sum w r^2,
 * acc[28] N. Parity-test entry for the dominant kernel. */
int odo_lm_accumulate(odo_lm* lm, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, int level,
                      const float T_colmajor[16], double acc[ODO_NACC]);
/* Per-evaluation trace of the last Solve (level, iteration, residual count, accept / stop decision, error, lambda, step; see
 * DESIGN.md). Optimisers created with odo_lm_create record it (and the per-level cost statistics of odo_lm_report); the
 * optimisers inside odo_tracker / odo_tracker_batch do not — nobody reads them, and the row costs ~600 cycles per evaluation on
 * the critical wave — unless ODO_LM_TRACE=1 is set when the tracker is created; odo_lm_trace then returns -1. */
typedef struct {
  int level, iter, n_res, accepted, stop;
  float err, lambda_after;
  float delta[6];
} odo_lm_trace_row;
int odo_lm_trace(const odo_lm* lm, odo_lm_trace_row* rows, int cap, int* n_rows);
/* Recording on (1, the default of odo_lm_create) / off (0). Off: the optimiser's Solves run the lean builds of the LM kernels —
 * no per-evaluation trace rows, no per-level cost statistics (odo_lm_trace returns -1, odo_lm_report's costs read 0; its
 * evaluation counts stay) —, ~2 % less time per Solve. The drop-in LevenbergMarquardtOptimizer switches it off: the reference's
 * ShowReport prints statistics nobody ever wrote (ref: src/lm_optimizer.cpp:364-371). Not while a Solve is in flight. */
int odo_lm_set_record(odo_lm* lm, int on);
/* Roofline leg of bench.py (any size / intrinsics, e.g. the dense 1920x1080 config): `reps` event-bracketed launches of
 * the evaluation kernel on `level` at pose T; mean / min launch time, algorithmic bytes of one launch, residual count. */
int odo_lm_time_eval(odo_lm* lm, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, int level,
                     const float T_colmajor[16], int reps, float* mean_us, float* min_us, double* algorithmic_bytes,
                     int* n_points);
/* The same for n optimisers of one context in ONE launch (blockIdx.y = stream): the batched dense evaluation of odo_lm_solve_batch.
 * `level` must be dense (not a point-list level) for every optimiser; algorithmic_bytes / n_points_total are totals over the streams. */
int odo_lm_time_eval_batch(int n, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                           const odo_pyr* const* cur_img, int level, const float T_colmajor[16], int reps, float* mean_us,
                           float* min_us, double* algorithmic_bytes, int* n_points_total);
/* Roofline leg of bench.py: execution spans of the LM kernels' launches (lm_coarse_kernel, lm_step_kernel and their batched
 * twins). A sampled launch records the device wall clock (100 MHz) at the entry of its earliest block and at the exit of its latest
 * one — the kernel's own execution time, free of queueing and dispatch effects — into a slot of device memory; the statistics
 * calls drain the stream and read the slots. on = 0: off; 1: every launch; N > 1: every N-th launch of a Solve (rotating residue:
 * cheap enough to stay on inside a timed region). For a batched Solve the statistics live with the first optimiser. */
int odo_lm_event_timing(odo_lm* lm, int on);
int odo_lm_event_stats(odo_lm* lm, double* total_us, long* launches, long* active_launches,
                       double* algorithmic_bytes);
/* Share of the above spent in the single-workgroup coarse-level kernel (one launch per Solve). */
int odo_lm_event_stats2(odo_lm* lm, double* coarse_us, long* coarse_launches);
/* out[0] sampled step-kernel time (us), [1] sampled step launches, [2] sampled coarse-kernel time (us), [3] sampled coarse
 * launches, [4] all launches issued, [5] all coarse launches, [6] evaluations, [7] algorithmic bytes, [8] / [9] summed
 * start-to-start periods of consecutive sampled step launches (us) and their number — execution plus the dependent-kernel
 * boundary: what an evaluation costs the serial chain —, [10] / [11] the same from a coarse launch to the step launch behind it. */
int odo_lm_event_stats_ex(odo_lm* lm, double out[12]);
/* Sampling of the current image at the warped point. ODO_SAMPLE_FLOOR (default, parity mode) is what the reference does:
 * I2 at floor(u), floor(v), central-difference gradient at that pixel (ref: src/lm_optimizer.cpp:208-217,
 * include/image_processing_global.h:62-69). ODO_SAMPLE_BILINEAR is a NON-PARITY option (BASELINE.json north_star: "bilinear
 * sample"): I2 interpolated in the 2x2 cell around (u, v), gradient = derivative of that interpolant, points whose cell
 * leaves the image skipped; everything else (geometric Jacobian at the un-warped point, weights, LM schedule) unchanged. It
 * has its own oracle mode (orc_set_sampling) and its own parity tests. */
enum { ODO_SAMPLE_FLOOR = 0, ODO_SAMPLE_BILINEAR = 1 };
int odo_lm_set_sampling(odo_lm* lm, int sampling);
/* Iteration space of the residual kernel: 0 = automatic (per keyframe and level: a compacted point list when at most
 * half of the interior pixels carry depth, the dense scan of the reference otherwise), 1 = always the dense scan,
 * 2 = always the point list. All three evaluate the same per-point arithmetic. */
int odo_lm_set_mode(odo_lm* lm, int mode);
/* Points per level of the cached keyframe lists and which levels use them (after a Solve / accumulate). */
int odo_lm_points(const odo_lm* lm, int npts[ODO_MAX_LEVELS], int use_list[ODO_MAX_LEVELS]);
/* Launch statistics of the last Solve (bench.py roofline): number of residual-kernel launches that did
 * work, and the algorithmic bytes they touched (dense scan 12 B per interior
 * pixel, point list 32 B per point, plus the fp64 partials written). */
int odo_lm_launch_stats(const odo_lm* lm, int* n_active_launches, int* n_total_launches, double* algorithmic_bytes);
/* The persistent launch of the fine levels (lm_fine_kernel: every evaluation the coarse launch leaves in ONE launch whose
 * workgroups exchange partial sums through L2): *workgroups = how many cooperate (0: off — this optimiser
 * issues a step launch per evaluation, by choice (ODO_LM_NO_FINE) or after three fall-backs), *fallbacks = Solves whose persistent
 * launch gave up waiting for one of its workgroups and that were redone on the step launches (results unaffected): this
 * optimiser's own Solves plus the batched Solves (odo_lm_solve_batch, odo_tracker_batch) of its context. */
int odo_lm_persistent_stats(const odo_lm* lm, int* workgroups, int* fallbacks);
/* Its give-up policy. A wait inside the launch is bounded by the device wall clock (4 ms; ODO_LM_FINE_WAIT_US) — stretched while the
 * shader clock runs below nominal (process start, power cap), but never beyond 4 x the bound (16 ms) —, so a give-up costs that + one
 * redo of the Solve on the step launches. Three give-ups switch the launch off; it is tried again after *retry_after
 * Solves — 4 096, doubling with every further switch-off up to 2^20 — and 1 024 clean Solves with the launch on forget all of it.
 * *strikes = give-ups that count at the moment (3: switched off), *solves_until_retry = Solves left on the step launches before the
 * next try (0 while the launch is on). */
int odo_lm_persistent_backoff(const odo_lm* lm, int* strikes, int* retry_after, int* solves_until_retry);
/* ComputeScaleNaive over levels too large for one workgroup (ref: src/lm_optimizer.cpp:338-358; dense levels, robust mode 2):
 * *multi_launches = scale iterations issued on the multi-workgroup kernel, *fallbacks = those redone by the single-workgroup kernel
 * queued behind it because the launch gave up waiting for a workgroup. The two kernels add in different orders: a Solve with a
 * fall-back may differ from one without in the last bits of sigma (and so of the pose). Synchronises the optimiser's stream. */
int odo_lm_tdist_stats(odo_lm* lm, long* multi_launches, int* fallbacks);

/* Diagnostic: cycle-counter stamps at the phase boundaries of one LM update launch ("update kernel"). */
int odo_debug_update_stamps(odo_lm* lm, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, int level,
                            const float T_colmajor[16], unsigned long long stamps[8]);
/* Test entry: the wave-parallel damped 6x6 solve used by the LM update kernel (ref: src/lm_optimizer.cpp:145-151)
 * on caller-supplied accumulators. */
int odo_debug_solve(odo_ctx* ctx, const double acc[ODO_NACC], float lambda, float delta[6]);

/* ---- depth estimator ---------------------------------------------------------------------------
 * Replaces DepthEstimator (ref: include/depth_estimate.h:31-33,51,54; src/depth_estimate.cpp:9-26,33-78,
 * 80-198,200-242,244-401,435-453,465-468). Camera pointers are replaced by K (NULL = KITTI-00).
 * max_disparity: 0 = the reference search range [boundary, x) (ref: src/depth_estimate.cpp:382), else
 * [max(boundary, x - max_disparity), x). any_size: 0 keeps the 376x1241 guard (ref: :46-49). */
int odo_depth_create(odo_ctx* ctx, float grad_th, float ssd_th, float photo_th, float min_depth, float max_depth,
                     float lambda, float huber_delta, float precision, int max_iters, int boundary,
                     const odo_intrinsics* K, float baseline, int max_residuals, int max_disparity, int any_size,
                     odo_depth** out);
/* ComputeDepth (ref: src/depth_estimate.cpp:33-78). Host buffers; val/disp/dep are overwritten
 * (zero-filled first: SURVEY appendix B #14). Returns -1 when fewer than 500 points survive (ref: :192-197). */
int odo_depth_compute(odo_depth* d, const float* left, const float* right, int rows, int cols, uint8_t* val,
                      float* disp, float* dep);
/* Same with device-resident inputs and outputs (no PCIe in the timed region). */
int odo_depth_compute_dev(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols,
                          uint8_t* val_dev, float* disp_dev, float* dep_dev);
/* The front half of ComputeDepth that needs the LEFT image only — its 3x3 blur and the block-median point selection (ref:
 * src/depth_estimate.cpp:255-256,300-342) — enqueued ahead of the call, on `side`'s stream, so that it runs beside whatever the
 * estimator's own stream is busy with (the drop-in classes issue it from ImagePyramid's constructor, ref:
 * run_odometry_kitti_offline.cpp:205: the pose LM's Solve of :215 then hides it). `stamp` (non-zero) names the image's content:
 * odo_depth_compute_dev_stamped(.., the same left_dev, the same stamp) picks the prepared half up (same launches, earlier: results
 * identical); any other call drops it. Work queued on the estimator's stream before this call is ordered in front of it. */
int odo_depth_prepare_left_dev(odo_depth* d, odo_ctx* side, const float* left_dev, int rows, int cols, unsigned long long stamp);
/* ... ordered behind `mark` of the estimator's stream (odo_ctx_mark, taken once left_dev was complete) instead of behind everything
 * queued there by now — for a caller that has meanwhile queued work the prepared half is meant to run BESIDE (the pose LM's Solve).
 * The estimator's previous ComputeDepth must have returned before the mark was taken. */
int odo_depth_prepare_left_dev_marked(odo_depth* d, odo_ctx* side, const float* left_dev, int rows, int cols, unsigned long long stamp,
                                      unsigned long mark);
int odo_depth_compute_dev_stamped(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols,
                                  uint8_t* val_dev, float* disp_dev, float* dep_dev, unsigned long long left_stamp);
/* The WHOLE of ComputeDepth(left, right) (ref: src/depth_estimate.cpp:31-78) started ahead of the call: every launch of it goes to
 * `side`'s stream — behind `mark` of the estimator's stream (0: behind everything queued there), i.e. behind whatever produced the
 * two images and recycled the three output blocks — and the function returns without waiting. ComputeDepth does not depend on the
 * pose: the drop-in classes start it once the Solve of :215 has been queued, and the runner's ComputeDepth of :229 only collects it.
 * Returns 0: started; 1: not started (the inverse-depth LM would need host-paced step launches: its persistent launch is off or
 * switched off) — nothing was queued; -1: error. Stamps (non-zero) name the two images' contents.
 * odo_depth_compute_end_dev: ComputeDepth proper with both stamps. The job started ahead with exactly these arguments is waited for
 * (bounded spin on its completion word; the estimator's stream is ordered behind it), a job started ahead with other arguments is
 * waited for and dropped, and without a matching job everything is computed now: the same launches either way, results identical.
 * Every other entry point of the estimator drops a job started ahead the same way. odo_depth_early_pending: 1 while one is out. */
int odo_depth_compute_begin_dev(odo_depth* d, odo_ctx* side, const float* left_dev, const float* right_dev, int rows, int cols,
                                uint8_t* val_dev, float* disp_dev, float* dep_dev, unsigned long long left_stamp,
                                unsigned long long right_stamp, unsigned long mark);
int odo_depth_compute_end_dev(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols, uint8_t* val_dev,
                              float* disp_dev, float* dep_dev, unsigned long long left_stamp, unsigned long long right_stamp);
int odo_depth_early_pending(const odo_depth* d);
/* The three output images handed over SPARSELY, for callers whose images live in host memory they own (cv::Mat): the images are zero
 * everywhere but at the selected points (ref: src/depth_estimate.cpp:388-397,176-191 write at the points only; the zero fill is forced
 * deviation #14), so {pixel index, val, disp, dep} per point slot — odo_depth_compact_bytes() = 532 KB instead of 4.2 MB at KITTI size —
 * crosses PCIe, and the host rebuilds the images: odo_depth_compact_outputs_async queues the gather and the copy into `dst_pinned`
 * (an odo_host_alloc block) on `on`'s stream, behind the job that wrote val_dev / disp_dev / dep_dev there (the estimator's point list of
 * that job must still be current: call it before the estimator's next ComputeDepth); odo_host_scatter_outputs, once the copy has
 * completed (odo_ctx_mark / odo_ctx_wait_mark), zero-fills the caller's three images and writes the points, and returns the
 * fingerprint (odo_host_fingerprint) of the inverse-depth image it wrote — the one output the runner hands back in (DepthPyramid,
 * ref: run_odometry_kitti_offline.cpp:252). Host-only, no device needed for the second call. */
size_t odo_depth_compact_bytes(void);
int odo_depth_compact_outputs_async(odo_depth* d, odo_ctx* on, const uint8_t* val_dev, const float* disp_dev, const float* dep_dev, int cols,
                                    void* dst_pinned);
int odo_host_scatter_outputs(const void* compact, int rows, int cols, uint8_t* val, size_t val_pitch, float* disp, size_t disp_pitch,
                             float* dep, size_t dep_pitch, unsigned long long* dep_fingerprint);
/* The same into images the caller has ALREADY zero-filled (the fill needs nothing from the device: a caller with idle time before the
 * compact block arrives does it then, odo_lm_set_idle_callback). */
int odo_host_scatter_outputs_prezeroed(const void* compact, int rows, int cols, uint8_t* val, size_t val_pitch, float* disp,
                                       size_t disp_pitch, float* dep, size_t dep_pitch, unsigned long long* dep_fingerprint);
/* Disparity stage only (DisparityDepthEstimate, ref: src/depth_estimate.cpp:244-401). */
int odo_depth_disparity(odo_depth* d, const float* left, const float* right, int rows, int cols, uint8_t* val,
                        float* disp, float* dep);
/* bench.py config-5 leg: event-timed blur / point selection / epipolar SSD scan on device-resident images (mean of
 * `reps`, microseconds), the number of SSD candidates one scan evaluates and the number of selected points. */
int odo_depth_time_stages(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols, int reps,
                          float us[3], double* candidates, int* n_selected);
/* ReportStatus data (ref: src/depth_estimate.cpp:465-468) + counts printed by ComputeDepth (:62,74). */
int odo_depth_report(const odo_depth* d, int* iters, float* cost, int* n_selected, int* n_matched, int* n_valid);
/* DepthOptimization (ref: src/depth_estimate.cpp:141-191) runs as ONE persistent launch (depth_lm_persistent_kernel: 80 workgroups of
 * 512 threads on one XCD, one point slot per thread with its state in registers, one tagged 16-byte pair {error sum, count} per
 * workgroup and iteration through L2) instead of a launch
 * per iteration; the step launches are its fall-back, bit-identical. *on = 1 while the persistent launch is in use (0: off, by
 * choice — ODO_DEPTH_NO_PERSIST — or after three give-ups, with the pose LM's back-off: odo_lm_persistent_backoff), *fallbacks =
 * ComputeDepth calls whose launch gave up waiting for one of its workgroups and that were run again on the step launches. */
int odo_depth_persistent_stats(const odo_depth* d, int* on, int* fallbacks);
int odo_depth_destroy(odo_depth* d);

/* ---- tracker: the runner's frame loop ---------------------------------------------------------------
 * Replaces the body of main() in run_odometry_kitti_offline.cpp:58-145 (set-up, frame 0) and :198-271 (per
 * frame): ImagePyramid(cur) -> Solve against the current keyframe -> cur_pose = KF * T^-1 -> ComputeDepth ->
 * rebuild the frame's image / depth pyramids -> keyframe test on the weighted motion -> Reset(T, 0.01).
 * Inputs are device-resident fp32 images (rows x cols, dense). ComputeDepth runs on a second HIP stream
 * concurrently with Solve when overlap_depth != 0 (the two are independent in the reference's loop):
 * 1 = both streams fed by the calling thread, 2 = stream B fed by a helper host thread (default). */
typedef struct {
  int rows, cols, levels;
  float lm_lambda, lm_precision;          /* ref: run_odometry_kitti_offline.cpp:88 (0.01f, 0.995f) */
  int lm_max_iters[ODO_MAX_LEVELS];       /* ref: :76 {10,20,30,30} */
  int lm_robust;                          /* ref: :86 (1 = Huber) */
  float lm_huber_delta;                   /* ref: :87 (28) */
  float grad_th, ssd_th, photo_th;        /* ref: :62-64 (8, 900, 15) */
  float min_depth, max_depth;             /* ref: :59-60 (0.1, 30) */
  float depth_lambda, depth_huber_delta, depth_precision; /* ref: :65-67 (0.01, 28, 0.995) */
  int depth_max_iters, boundary, max_residuals;            /* ref: :68,:69 (50, 4), :61 (80000) */
  int max_disparity, any_size;            /* deviations from the reference, both 0 in parity mode */
  odo_intrinsics K;
  float baseline;                         /* ref: :41 */
  float keyframe_weight[6];               /* ref: :144-145 */
  float keyframe_motion_th;               /* ref: :258 (1.1) */
  int smooth_image;                       /* ref: :130,:205,:251 (true) */
  int overlap_depth;
} odo_tracker_params;

int odo_tracker_default_params(odo_tracker_params* p); /* the runner's constants for KITTI 1241x376 */
int odo_tracker_create(int device, const odo_tracker_params* p, odo_tracker** out);
/* Frame 0 (ref: :95-145): ComputeDepth, pyramids, first keyframe with absolute pose abs_pose0. May be called again at
 * any time to start a new sequence on the same tracker: it drains both streams first and the tracker then behaves
 * exactly like a freshly created one. */
int odo_tracker_init(odo_tracker* t, const float* left_dev, const float* right_dev, const float abs_pose0_colmajor[16]);
/* One iteration of the frame loop (ref: :198-271). Returns 0, or -1 when ComputeDepth failed (the runner
 * breaks out of its loop there, ref: :230-232); like the runner, which stores the frame's pose before it computes the depth
 * (ref: :215-232), pose_to_keyframe / abs_pose / solve_status of that last frame are still written. A failed Solve is NOT an
 * error (the runner carries on with the pseudo-identity); solve_status reports it, and abs_pose is then NaN (the inverse of
 * the singular pseudo-identity, ref: :218), not zeros. */
int odo_tracker_track(odo_tracker* t, const float* left_dev, const float* right_dev, float pose_to_keyframe[16],
                      float abs_pose[16], int* is_new_keyframe, float* motion_mag, int* solve_status);
/* Optional pipelining for callers that already hold the next frame (offline runs): announce its left image before
 * tracking the current frame; its image pyramid (ref: :205 of the NEXT iteration) is then built during this call on a stream
 * of its own, and the next frame's Solve is started (odo_lm_solve_begin: initial pose = this frame's result, ref: :261 / :268)
 * as soon as this frame's Solve has returned, while the depth stream finishes this frame. Same work, earlier; results are
 * unchanged (the LM's per-evaluation trace of the frame just tracked may already be overwritten when the call returns;
 * ODO_NO_EARLY_SOLVE=1 keeps the pyramid prefetch only). The hinted buffer is identified by its device
 * address: its contents must not change between the hint and the odo_tracker_track call that consumes it (a caller that
 * recycles one buffer for every frame must not hint). odo_tracker_init drops a pending hint / prefetched pyramid. */
int odo_tracker_hint_next(odo_tracker* t, const float* next_left_dev);
/* The same with the next frame's right image as well: its ComputeDepth + candidate pyramids (ref: :226-252 of the NEXT
 * iteration; they depend on the images only) are then enqueued on the depth stream a frame early too, behind this frame's, so
 * the depth stream works a frame ahead of the pose LM and a short Solve no longer waits for it (overlap_depth == 2;
 * ODO_NO_DEPTH_AHEAD=1 turns it off). Both buffers must stay unchanged until the odo_tracker_track call that consumes them.
 * Results are unchanged. odo_tracker_outputs stays valid until the next odo_tracker_track call, as before. */
int odo_tracker_hint_next_pair(odo_tracker* t, const float* next_left_dev, const float* next_right_dev);
/* Runs to completion (or drops) everything still in flight on behalf of frames the caller handed over — the stream-B job of an
 * announced pair, the prefetched pyramid of an announced image, an early-started Solve — and leaves all streams idle. After it
 * returns nothing, queued or yet to be issued by the helper thread, reads a caller-owned frame buffer: call it before freeing or
 * overwriting frames that were announced but never tracked (odo_tracker_destroy / odo_tracker_init do it themselves). */
int odo_tracker_quiesce(odo_tracker* t);
/* Counters of the last tracked frame: LM evaluations, depth-LM iterations, valid depth points, keyframes so far. */
int odo_tracker_stats(const odo_tracker* t, int* lm_evals, int* depth_iters, int* n_valid_depth, int* n_keyframes);
/* Device pointers to the last frame's outputs (rows x cols): validity mask (u8), disparity, inverse depth. */
int odo_tracker_outputs(const odo_tracker* t, const uint8_t** val_dev, const float** disp_dev, const float** dep_dev);
/* Roofline leg of bench.py: `reps` event-bracketed launches of the dominant kernel (residual / normal-equation
 * pass) on `level` of the tracker's current pyramids; mean / min launch time, algorithmic bytes of one launch
 * (12 B per interior pixel + the fp64 partials written) and the number of residuals it produced. */
int odo_tracker_time_residual(odo_tracker* t, int level, int reps, float* mean_us, float* min_us,
                              double* algorithmic_bytes, int* n_points);
/* Diagnostics: host-clock averages per tracked frame since the last call, microseconds:
 * {track() call, Solve on stream A, stream-B job on the helper thread, wait for the helper}. */
int odo_tracker_timing(odo_tracker* t, double out[4]);
/* Chained Solves: with the next frame announced, its Solve is queued BEHIND this frame's before this frame's result exists —
 * same keyframe, initial pose = this Solve's result taken on the device (what Reset hands it, ref: run_odometry_kitti_offline.cpp:
 * 261,268), guarded on the device by the runner's keyframe test (ref: :253-258: a promoted or failed frame makes every launch of
 * the chained Solve return at once). The host's own test decides what counts; results are those of the unchained order.
 * Opt-in (ODO_CHAIN_SOLVE=1): it closes the GPU's idle gap between two Solves and does not change the frame rate. *adopted = chained Solves that became the next Solve, *wasted = chained Solves that ran for
 * nothing because the two keyframe tests disagreed (an ulp of atan2f: expected never). */
int odo_tracker_chain_stats(const odo_tracker* t, long* adopted, long* wasted);
/* Armed Solves (round 6; the default with the next pair announced, ODO_NO_ARM=1 turns them off): the next frame's Solve has its
 * coarse launch queued behind this frame's before this frame's result exists and starts on a word the host writes after the runner's
 * keyframe test (ref: run_odometry_kitti_offline.cpp:253-268) — same launches, same arithmetic, no launch call and no dispatch between
 * two Solves. started: Solves that began that way; returned: armed launches told to return (new keyframe, failed Solve). */
int odo_tracker_arm_stats(const odo_tracker* t, long* started, long* returned);
odo_lm* odo_tracker_lm(odo_tracker* t);
odo_depth* odo_tracker_depth(odo_tracker* t);   /* its depth estimator (odo_depth_persistent_stats, odo_depth_report) */
odo_ctx* odo_tracker_ctx(odo_tracker* t);

/* ---- S sequences in lock step on one GPU (the data-parallel axis of SURVEY section 8e inside one device) -----------------
 * One odo_tracker tracks one sequence as a serial chain of ~70 short launches per frame, which leaves most of the chip
 * idle. odo_tracker_batch runs the same frame loop (ref: run_odometry_kitti_offline.cpp:198-271) for n_sequences independent
 * sequences with every launch carrying all of them (blockIdx.y / .z = sequence). Per-sequence arithmetic, reduction order and
 * launch order are those of odo_tracker: poses, depth maps and keyframe decisions are bit-identical to n separate trackers.
 * All sequences share rows / cols / parameters. Arrays are indexed by sequence; poses are n x 16 floats, column-major. */
typedef struct odo_tracker_batch odo_tracker_batch;
int odo_tracker_batch_create(int device, const odo_tracker_params* p, int n_sequences, odo_tracker_batch** out);
int odo_tracker_batch_destroy(odo_tracker_batch* b);
/* The inverse-depth LM of a lock step (ref: src/depth_estimate.cpp:141-168,200-242) in ONE persistent launch for all sequences — each on
 * an XCD of its own, beside the batched pose LM's — instead of a launch per iteration: up to four sequences (more share their XCDs with
 * the pose LM, where the 80 workgroups of a sequence do not fit). *on = 1 while it is in use, *chains_redone = lock steps whose launch
 * gave up and whose depth jobs were run again on the launches per iteration (results unaffected; three switch it off). */
int odo_tracker_batch_depth_persistent_stats(const odo_tracker_batch* b, int* on, int* chains_redone);
int odo_tracker_batch_size(const odo_tracker_batch* b);
/* The pose optimiser of one slot (launch statistics of the batched Solves live with the first slot's: odo_lm_event_timing). */
odo_lm* odo_tracker_batch_lm(odo_tracker_batch* b, int slot);
odo_ctx* odo_tracker_batch_ctx(odo_tracker_batch* b); /* for odo_dev_alloc / upload / download of its frames */
/* Frame 0 (ref: :95-145) of every slot that is given a pair; a slot whose left_dev[i] / right_dev[i] are NULL stays empty.
 * abs_pose0: n x 16, or NULL for identity. Returns -1 if any sequence's ComputeDepth fails ("Init 0-th frame failed!",
 * ref: :103-106; the other slots are initialised all the same). May be called again to start new sequences. */
int odo_tracker_batch_init(odo_tracker_batch* b, const float* const* left_dev, const float* const* right_dev,
                           const float* abs_pose0_colmajor);
/* Starts a new sequence in ONE slot while the other slots keep theirs (sequences of different lengths: when one ends its
 * slot takes the next sequence). Drains both streams first. abs_pose0: 16 floats or NULL for identity. */
int odo_tracker_batch_init_one(odo_tracker_batch* b, int slot, const float* left_dev, const float* right_dev,
                               const float abs_pose0_colmajor[16]);
/* One iteration of the frame loop for every slot that is given a frame. status[i]: 0 tracked; 1 the Solve failed
 * (pseudo-identity, the runner carries on, ref: src/lm_optimizer.cpp:60-65); -1 ComputeDepth failed on this frame (its pose is
 * still written and the sequence stops, ref: :230-232); -2 the slot holds no running sequence (empty, or stopped earlier: its
 * outputs are untouched); -3 no frame was given for the slot (left_dev[i] == NULL): it sits this step out, its state is kept,
 * it costs nothing. Returns -1 only for argument / device errors. pose_to_keyframe, abs_pose, is_new_keyframe, motion_mag may
 * be NULL. */
int odo_tracker_batch_track(odo_tracker_batch* b, const float* const* left_dev, const float* const* right_dev,
                            float* pose_to_keyframe, float* abs_pose, int* is_new_keyframe, float* motion_mag, int* status);
/* Optional pipelining, as odo_tracker_hint_next: the left images of the NEXT step (n pointers, NULL entries allowed), announced
 * before odo_tracker_batch_track of the current one; their pyramids are built behind this step's Solve. Same work, earlier. */
int odo_tracker_batch_hint_next(odo_tracker_batch* b, const float* const* next_left_dev);
/* The same with the right images (odo_tracker_hint_next_pair for every slot): the next step's ComputeDepth, depth pyramids and
 * candidate lists are enqueued a step early, so the depth stream works a step ahead of the pose LM. NULL entries allowed. */
int odo_tracker_batch_hint_next_pair(odo_tracker_batch* b, const float* const* next_left_dev, const float* const* next_right_dev);
/* Counters of the last tracked frame, one entry per sequence (any pointer may be NULL). */
int odo_tracker_batch_stats(const odo_tracker_batch* b, int* lm_evals, int* depth_iters, int* n_valid_depth, int* n_keyframes);
/* The batched twin of odo_tracker_quiesce: posted chains run to completion, the early-started next Solve is dropped,
 * announcements are void, all streams idle; afterwards nothing reads a caller-owned frame buffer. */
int odo_tracker_batch_quiesce(odo_tracker_batch* b);
/* Diagnostics: host-clock averages per lock step since the last call, microseconds: {whole call, table + pyramid launches,
 * batched Solve, wait for the depth chain after the Solve}. */
int odo_tracker_batch_timing(odo_tracker_batch* b, double out[4]);
/* Device pointers to sequence `seq`'s last outputs (rows x cols): validity mask (u8), disparity, inverse depth. */
int odo_tracker_batch_outputs(const odo_tracker_batch* b, int seq, const uint8_t** val_dev, const float** disp_dev,
                              const float** dep_dev);
int odo_tracker_destroy(odo_tracker* t);

/* ---- camera model: rectified intrinsics per level, undistort + rectify ---------------------------------
 * Replaces odometry::CameraPyramid (ref: include/camera.h:16-119).
 * odo_camera_create      = CameraPyramid(levels, fx, fy, f_theta, cx, cy, k1, k2, r1, r2, sensor_w, sensor_h,
 *                          resolution_w, resolution_h) (ref: include/camera.h:34-35; src/camera.cpp:12-38).
 * odo_camera_configure   = ConfigureCamera(rectify_rotation 3x3, new_intrinsic 3x4, new_size, CV_32FC1, false)
 *                          (ref: include/camera.h:56-57; src/camera.cpp:40-69): the intrinsic pyramid
 *                          (f / 2, c <- (c + 0.5) / 2 + 0.5 per level, double) and the cv::initUndistortRectifyMap
 *                          lookup maps, built on the device. R and P are row-major doubles as cv::stereoRectify
 *                          returns them (cv::stereoRectify itself, a one-off host-side calibration step inside
 *                          SetUpStereoCameraSystem, ref: src/camera.cpp:138-141, stays with the caller).
 * odo_camera_undistort_rectify = UndistortRectify(src_raw, dst, INTER_LINEAR, BORDER_CONSTANT, borderValue)
 *                          (ref: include/camera.h:68; src/camera.cpp:71-82): cv::remap through the maps. The
 *                          reference's hard 480x640 check lives in the C++ shim; this entry takes any size.
 *                          dst is map_rows x map_cols. The _dev variant works on device-resident images and
 *                          is asynchronous on the context's stream.
 * odo_camera_intrinsics  = fx/fy/f_theta/cx/cy_double(level) (ref: include/camera.h:80-85), out5 in that order.
 * odo_camera_raw         = the raw-parameter accessors (ref: include/camera.h:91-105). */
int odo_camera_create(odo_ctx* ctx, int levels, double fx, double fy, double f_theta, double cx, double cy, double k1,
                      double k2, double r1, double r2, double sensor_width, double sensor_height, int resolution_width,
                      int resolution_height, odo_camera** out);
int odo_camera_configure(odo_camera* cam, const double R_rowmajor[9], const double P_rowmajor[12], int new_width,
                         int new_height);
int odo_camera_levels(const odo_camera* cam);
int odo_camera_intrinsics(const odo_camera* cam, int level, double out5[5]);
int odo_camera_raw(const odo_camera* cam, double raw5[5], double dist4[4], double sensor2[2], int resolution2[2]);
int odo_camera_map_size(const odo_camera* cam, int* rows, int* cols);
int odo_camera_download_maps(const odo_camera* cam, float* mapx_host, float* mapy_host);
int odo_camera_undistort_rectify(odo_camera* cam, const float* src, int src_rows, int src_cols, float* dst,
                                 float border_value);
int odo_camera_undistort_rectify_dev(odo_camera* cam, const float* src_dev, int src_rows, int src_cols, float* dst_dev,
                                     float border_value);
int odo_camera_destroy(odo_camera* cam);

/* ---- the pose exchange of the multi-GPU path for hosts that are not Python ---------------------------
 * Tracking shards by sequence (rank r owns sequences r, r + N, ...; ref: run_odometry_kitti_offline.cpp:198-271 is sequential
 * inside a sequence and independent across sequences): there is no data-path collective. The one exchange is an all-gather of
 * the results over RCCL: per tracked frame a row of ODO_GATHER_ROW four-byte words (int32 sequence id, int32 frame id — bit
 * patterns in the float row, exact for any id — then the 3x4 absolute pose as floats, row-major), `every` rows per ncclAllGather, on a stream of its own, never waited for while tracking. The schedule is fixed up
 * front: EVERY rank issues ceil(n_max_frames / every) collectives of a fixed block and pads with rows whose sequence id is -1
 * (uneven shards — 11 sequences over 8 GPUs — cannot leave ranks with different numbers of collectives). RCCL is dlopen'ed
 * (librccl.so.1, or ODO_RCCL_SO): no link-time dependency. bench.py uses the same schedule through torch.distributed
 * (odometry_amd/dist.py); on a one-GPU box this entry runs at world size 1 only (RCCL refuses two ranks on one device),
 * tests/test_gpu_gather.py::test_two_ranks_* runs it at world size 2 wherever two devices are visible. */
#define ODO_GATHER_ID_BYTES 128
#define ODO_GATHER_ROW 14
typedef struct odo_gather odo_gather;
/* Rank 0: an ncclUniqueId; hand its bytes to every rank by the host's own means (MPI, a file, a socket). */
int odo_gather_unique_id(unsigned char id[ODO_GATHER_ID_BYTES]);
/* Collective over all ranks (ncclCommInitRank). n_local_frames: rows THIS rank will push; n_max_frames: the largest such number
 * over the ranks (every rank derives both from the same sharding rule). */
int odo_gather_create(int device, int world, int rank, const unsigned char id[ODO_GATHER_ID_BYTES], int every,
                      int n_local_frames, int n_max_frames, odo_gather** out);
/* One tracked frame's result; every `every`-th push issues a collective and returns at once. */
int odo_gather_push(odo_gather* g, int seq_id, int frame_id, const float abs_pose_colmajor[16]);
/* Issues what is left of the schedule (padded) and waits for every collective. */
int odo_gather_flush(odo_gather* g);
/* After flush: the valid rows received from `rank` (ODO_GATHER_ROW words each, in push order; words 0 and 1 are int32 bit
 * patterns: memcpy them into ints). */
int odo_gather_rows(odo_gather* g, int rank, const float** rows, int* n_rows);
int odo_gather_issued(const odo_gather* g);   /* collectives issued so far */
int odo_gather_ranks(const odo_gather* g);    /* ranks in the communicator, from ncclCommCount; -1 on error */
int odo_gather_destroy(odo_gather* g);

#ifdef __cplusplus
}
#endif
#endif /* ODOMETRY_HIP_H */
