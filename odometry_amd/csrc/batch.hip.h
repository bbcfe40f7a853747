// batch.hip.h — S sequences tracked in lock step on one GPU: every launch of the frame loop carries all S sequences
// (blockIdx.y / blockIdx.z = sequence), so the serial launch chains that bound a single sequence (~46 pose-LM launches +
// ~25 depth-LM launches of 5-10 us each, the chip mostly idle) are paid once per S frames instead of once per frame.
//
// Same arithmetic as S separate odo_tracker objects: the batched kernels are the single-sequence kernel BODIES
// (kernels.hip.h *_kernel_body) called with one sequence's pointers out of a device table; grids, block sizes, launch
// order per sequence and every reduction order are unchanged, so poses / depth maps / keyframe decisions are bit-identical
// (tests/test_gpu_batch.py). The runner's loop being restated is run_odometry_kitti_offline.cpp:198-271, once per sequence.
//
// Included by odometry_hip.hip after tracker.hip.h (uses its pose helpers).
#pragma once

namespace odo {

// One sequence's device pointers for one tracked frame. The table (S entries) is uploaded once per frame.
struct BatchSeq {
  const float* left;
  const float* right;
  PyrOut cur_img;  // image pyramid of the incoming frame (:205; also the keyframe image pyramid if the frame is promoted, :251)
  PyrOut pre_dep;  // depth pyramid of the frame's inverse depth (:252)
  // ComputeDepth (src/depth_estimate.cpp:40-207)
  float *bl, *br, *disp, *dep, *d0, *scratch;
  uint8_t *val, *matched;
  uint32_t* pts;
  int* cnt;
  DepthLmState* dstate;
  double* part_e;
  int* part_n;
  int* counts;
  DepthLmStats* stats;  // host-mapped
  int* dprog;           // host-mapped progress words: [0] launches consumed, [1] LM stopped, [4] job complete (token)
  int dtoken;
  // keyframe-candidate point lists
  KfLevels kl;
  int* rowcnt;
  int* npts;            // device
  int* npts_host;       // host-mapped copy, written by the job's last launch
  PointList pl[ODO_MAX_LEVELS_K];
};

__global__ void __launch_bounds__(kPyrThreads) image_pyramid_batch_kernel(const BatchSeq* __restrict__ tab) {
  const BatchSeq& q = tab[blockIdx.z];
  image_pyramid_fused_kernel_body(q.left, q.cur_img);
}
__global__ void __launch_bounds__(256) depth_pyramid_batch_kernel(const BatchSeq* __restrict__ tab) {
  const BatchSeq& q = tab[blockIdx.z];
  depth_pyramid_fused_kernel_body(q.dep, q.pre_dep);
}
__global__ void __launch_bounds__(256) blur3x3_batch_kernel(const BatchSeq* __restrict__ tab, int rows, int cols) {
  const BatchSeq& q = tab[blockIdx.z >> 1];
  blur3x3_kernel_body((int)(blockIdx.z & 1), q.left, q.bl, q.right, q.br, rows, cols, q.val, q.disp, q.dep);
}
__global__ void __launch_bounds__(kSelThreads) depth_select_batch_kernel(const BatchSeq* __restrict__ tab, int rows, int cols,
                                                                        int bnd, float grad_th) {
  const BatchSeq& q = tab[blockIdx.y];
  depth_select_kernel_body(q.bl, rows, cols, bnd, grad_th, q.val, q.pts, q.cnt);
}
__global__ void __launch_bounds__(256) depth_disparity_batch_kernel(const BatchSeq* __restrict__ tab, int rows, int cols, int bnd,
                                                                    int max_disp, float ssd_th, float f0, float baseline) {
  const BatchSeq& q = tab[blockIdx.y];
  depth_disparity_kernel_body(q.bl, q.br, rows, cols, bnd, max_disp, ssd_th, f0, baseline, q.pts, q.cnt, q.disp, q.dep, q.d0,
                              q.matched);
}
__global__ void __launch_bounds__(kDlmBlock) depth_lm_step_batch_kernel(const BatchSeq* __restrict__ tab, int k, int cols,
                                                                        float tx, float fx, float huber_delta, float lambda0,
                                                                        float precision, int max_iters) {
  const BatchSeq& q = tab[blockIdx.y];
  depth_lm_step_kernel_body(k, q.left, q.right, cols, q.pts, q.cnt, q.d0, q.scratch, q.dstate, q.part_e, q.part_n, tx, fx,
                            huber_delta, lambda0, precision, max_iters, q.dprog);
}
__global__ void __launch_bounds__(kDlmBlock) depth_finalize_batch_kernel(const BatchSeq* __restrict__ tab, int run_lm, int cols,
                                                                         float photo_th, float min_depth, float max_depth) {
  const BatchSeq& q = tab[blockIdx.y];
  depth_finalize_kernel_body(run_lm, cols, q.pts, q.cnt, q.matched, q.scratch, photo_th, min_depth, max_depth, q.val, q.dep,
                             q.counts);
}
__global__ void __launch_bounds__(256) kf_count_batch_kernel(const BatchSeq* __restrict__ tab) {
  const BatchSeq& q = tab[blockIdx.y];
  if (blockIdx.x == 0 && threadIdx.x < ODO_MAX_LEVELS_K) q.npts[threadIdx.x] = 0;  // levels without interior rows stay at 0
  kf_count_kernel_body<const KfLevels&>(q.kl, q.rowcnt);
}
__global__ void __launch_bounds__(256) kf_fill_batch_kernel(const BatchSeq* __restrict__ tab, float f0, float cx0, float cy0) {
  const BatchSeq& q = tab[blockIdx.y];
  const int l = kf_find_level(q.kl, blockIdx.x);
  kf_fill_kernel_body<const KfLevels&>(q.kl, l, q.pl[l], f0, cx0, cy0, q.rowcnt, q.npts);
}
// Last launch of a frame: per-sequence statistics, the candidate lists' per-level counts and the completion word, all
// through host-mapped memory (the release store of the completion word orders them).
__global__ void __launch_bounds__(kDlmBlock) depth_stats_batch_kernel(const BatchSeq* __restrict__ tab, int run_lm, int n_launches,
                                                                      int with_lists) {
  const BatchSeq& q = tab[blockIdx.y];
  if (with_lists && threadIdx.x < ODO_MAX_LEVELS_K) q.npts_host[threadIdx.x] = q.npts[threadIdx.x];
  depth_stats_kernel_body(run_lm, n_launches, q.counts, q.dstate, q.stats, q.dprog + 4, q.dtoken);
}

}  // namespace odo

struct odo_tracker_batch {
  odo_tracker_params p;
  int S;           // slots
  odo_ctx* ctx_a;  // image pyramids + pose LM
  odo_ctx* ctx_b;  // ComputeDepth, depth pyramids, candidate lists
  std::vector<odo_lm*> lm;
  std::vector<odo_depth*> depth;
  std::vector<odo_pyr*> kf_img, kf_dep, cur_img, pre_dep, next_img;
  std::vector<uint8_t*> d_val;
  std::vector<float*> d_disp, d_dep;
  std::vector<float> kf_abs, pose_to_kf;  // S x 16
  std::vector<int> n_keyframes, alive, last_evals, last_depth_iters, last_valid;
  std::vector<long> frame_id;             // per slot: frames tracked since its init (tags the candidate lists)
  std::vector<const float*> hint_next, prefetched;  // per slot: announced next left image / image whose pyramid sits in next_img
  BatchSeq* h_tab;  // pinned, 2 S entries: [0, S) the lock step in flight, [S, 2S) the prefetch launch's pyramid-only entries
  BatchSeq* d_tab;
  int* h_cand_npts;  // host-mapped, S x ODO_MAX_LEVELS
  int* d_cand_npts;  // its device alias
  hipEvent_t ev_cur_img, ev_tab;
  int overlap;       // 1: the depth chain runs on stream B beside the pose LM on stream A (one host thread feeds both)
  // the set of slots of the step in flight (table entry e <-> slot ids[e]) and its depth chain
  std::vector<int> ids;
  int dk, dn_launches, dstage;  // dstage: 0 idle, 1 LM launches, 2 tail enqueued
  bool dpoll, dwaiting, with_lists;
  int derr;
  std::chrono::steady_clock::time_point dwait_since;
  double tm_frame_us, tm_head_us, tm_solve_us, tm_depth_wait_us; long tm_frames;  // host-clock averages (diagnostics)
};

static PyrOut batch_pyr_out(const odo_pyr* p, int smooth) {
  PyrOut o;
  memset(&o, 0, sizeof(o));
  o.n_levels = p->levels; o.smooth = smooth ? 1 : 0;
  for (int l = 0; l < p->levels; l++) { o.lvl[l] = p->dev + p->off[l]; o.rows[l] = p->r[l]; o.cols[l] = p->c[l]; }
  return o;
}

extern "C" int odo_tracker_batch_destroy(odo_tracker_batch* b) {
  if (!b) return 0;
  if (b->ctx_a) (void)hipStreamSynchronize(b->ctx_a->stream);
  if (b->ctx_b) (void)hipStreamSynchronize(b->ctx_b->stream);
  for (odo_lm* m : b->lm) odo_lm_destroy(m);
  for (odo_depth* d : b->depth) odo_depth_destroy(d);
  for (auto* v : {&b->kf_img, &b->kf_dep, &b->cur_img, &b->pre_dep, &b->next_img})
    for (odo_pyr* q : *v) odo_pyramid_destroy(q);
  for (uint8_t* q : b->d_val) if (q) (void)hipFree(q);
  for (float* q : b->d_disp) if (q) (void)hipFree(q);
  for (float* q : b->d_dep) if (q) (void)hipFree(q);
  if (b->h_tab) (void)hipHostFree(b->h_tab);
  if (b->d_tab) (void)hipFree(b->d_tab);
  if (b->h_cand_npts) (void)hipHostFree(b->h_cand_npts);
  if (b->ev_cur_img) (void)hipEventDestroy(b->ev_cur_img);
  if (b->ev_tab) (void)hipEventDestroy(b->ev_tab);
  odo_ctx_destroy(b->ctx_b);
  odo_ctx_destroy(b->ctx_a);
  delete b;
  return 0;
}

extern "C" int odo_tracker_batch_create(int device, const odo_tracker_params* p, int n_sequences, odo_tracker_batch** out) {
  if (!p || !out) return fail("odo_tracker_batch_create: NULL arg");
  *out = nullptr;
  if (n_sequences < 1 || n_sequences > 64) return fail("odo_tracker_batch_create: 1..64 sequences, got %d", n_sequences);
  if (p->levels > 4) return fail("odo_tracker_batch_create: at most 4 pyramid levels (the one-launch pyramid kernels)");
  odo_tracker_batch* b = new (std::nothrow) odo_tracker_batch();
  if (!b) return fail("out of memory");
  const int S = n_sequences;
  b->p = *p; b->S = S;
  b->ctx_a = b->ctx_b = nullptr; b->h_tab = nullptr; b->d_tab = nullptr; b->h_cand_npts = b->d_cand_npts = nullptr;
  b->ev_cur_img = b->ev_tab = nullptr;
  b->dstage = 0; b->derr = 0;
  b->tm_frame_us = b->tm_head_us = b->tm_solve_us = b->tm_depth_wait_us = 0.0; b->tm_frames = 0;
  b->overlap = (p->overlap_depth != 0) && !getenv("ODO_BATCH_NO_OVERLAP");
  b->lm.assign(S, nullptr); b->depth.assign(S, nullptr);
  b->kf_img.assign(S, nullptr); b->kf_dep.assign(S, nullptr); b->cur_img.assign(S, nullptr); b->pre_dep.assign(S, nullptr);
  b->next_img.assign(S, nullptr);
  b->d_val.assign(S, nullptr); b->d_disp.assign(S, nullptr); b->d_dep.assign(S, nullptr);
  b->kf_abs.assign((size_t)S * 16, 0.0f); b->pose_to_kf.assign((size_t)S * 16, 0.0f);
  b->n_keyframes.assign(S, 0); b->alive.assign(S, 0); b->last_evals.assign(S, 0); b->last_depth_iters.assign(S, 0);
  b->last_valid.assign(S, 0); b->frame_id.assign(S, 0);
  b->hint_next.assign(S, nullptr); b->prefetched.assign(S, nullptr);
  const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const size_t n = (size_t)p->rows * p->cols;
  const bool lm_prio = !(getenv("ODO_LM_PRIORITY") && atoi(getenv("ODO_LM_PRIORITY")) == 0);
  bool ok = (lm_prio ? odo_ctx_create_high_priority(device, &b->ctx_a) : odo_ctx_create(device, &b->ctx_a)) == 0 &&
            odo_ctx_create(device, &b->ctx_b) == 0;
  for (int i = 0; i < S && ok; i++) {
    ok = ok && odo_lm_create(b->ctx_a, p->lm_lambda, p->lm_precision, p->lm_max_iters, p->levels, eye, p->lm_robust,
                             p->lm_huber_delta, &p->K, &b->lm[i]) == 0;
    ok = ok && odo_depth_create(b->ctx_b, p->grad_th, p->ssd_th, p->photo_th, p->min_depth, p->max_depth, p->depth_lambda,
                                p->depth_huber_delta, p->depth_precision, p->depth_max_iters, p->boundary, &p->K, p->baseline,
                                p->max_residuals, p->max_disparity, p->any_size, &b->depth[i]) == 0;
    ok = ok && depth_check_size(b->depth[i], p->rows, p->cols) == 0 && depth_ensure(b->depth[i], p->rows, p->cols) == 0;
    ok = ok && pyr_alloc(b->ctx_a, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &b->cur_img[i], false) == 0;
    ok = ok && pyr_alloc(b->ctx_a, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &b->next_img[i], false) == 0;
    ok = ok && pyr_alloc(b->ctx_a, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &b->kf_img[i], false) == 0;
    ok = ok && pyr_alloc(b->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_DEPTH, &b->kf_dep[i], false) == 0;
    ok = ok && pyr_alloc(b->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_DEPTH, &b->pre_dep[i], false) == 0;
    ok = ok && hipMalloc((void**)&b->d_val[i], n) == hipSuccess && hipMalloc((void**)&b->d_disp[i], sizeof(float) * n) == hipSuccess &&
         hipMalloc((void**)&b->d_dep[i], sizeof(float) * n) == hipSuccess;
  }
  ok = ok && hipHostMalloc((void**)&b->h_tab, sizeof(BatchSeq) * 2 * (size_t)S, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&b->d_tab, sizeof(BatchSeq) * 2 * (size_t)S) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&b->h_cand_npts, sizeof(int) * ODO_MAX_LEVELS * (size_t)S,
                           hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void**)&b->d_cand_npts, b->h_cand_npts, 0) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&b->ev_cur_img, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&b->ev_tab, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", g_err);
    odo_tracker_batch_destroy(b);
    return fail("odo_tracker_batch_create failed: %s", keep);
  }
  *out = b;
  return 0;
}

// Fills and uploads the step's table: entry e <-> slot b->ids[e] (stream A; stream B waits for the upload through ev_tab).
// first: frame 0 of the listed slots, whose pyramids go straight into the keyframe buffers.
static int batch_upload_table(odo_tracker_batch* b, const float* const* left, const float* const* right, bool first) {
  const odo_tracker_params& p = b->p;
  const int n = (int)b->ids.size();
  for (int e = 0; e < n; e++) {
    const int i = b->ids[e];
    BatchSeq& q = b->h_tab[e];
    odo_depth* d = b->depth[i];
    odo_lm* m = b->lm[i];
    odo_pyr* img = first ? b->kf_img[i] : b->cur_img[i];
    odo_pyr* dep = first ? b->kf_dep[i] : b->pre_dep[i];
    memset(&q, 0, sizeof(q));
    q.left = left[i]; q.right = right[i];
    q.cur_img = batch_pyr_out(img, p.smooth_image);
    q.pre_dep = batch_pyr_out(dep, 0);
    dep->version = ++g_pyr_version;
    q.bl = d->d_bl; q.br = d->d_br; q.disp = b->d_disp[i]; q.dep = b->d_dep[i]; q.d0 = d->d_d0; q.scratch = d->d_scratch;
    q.val = b->d_val[i]; q.matched = d->d_matched; q.pts = d->d_pts; q.cnt = d->d_cnt;
    q.dstate = d->d_lmstate; q.part_e = d->d_part_e; q.part_n = d->d_part_n; q.counts = d->d_counts;
    q.stats = d->d_stats_map; q.dprog = d->d_prog;
    d->token++;
    q.dtoken = d->token;
    d->h_prog[0] = 0; d->h_prog[1] = 0;  // both streams are idle here (every step ends on the completion words)
    int rows_total = 0;
    m->cand[0].tag = -1;
    if (b->with_lists) {
      if (lm_lists_layout(m, m->cand[0].pl, m->cand[0].pl_cap, m->cand[0].d_rowcnt, m->cand[0].rows_cap, img, dep, b->ctx_b->stream, &q.kl,
                          &rows_total)) return -1;
      if (rows_total > 0) m->cand[0].tag = b->frame_id[i];
    }
    q.rowcnt = m->cand[0].d_rowcnt; q.npts = m->cand[0].d_npts; q.npts_host = b->d_cand_npts + (size_t)i * ODO_MAX_LEVELS;
    for (int l = 0; l < ODO_MAX_LEVELS; l++) q.pl[l] = m->cand[0].pl[l];
  }
  hipStream_t sa = b->ctx_a->stream;
  HIP_OK(hipMemcpyAsync(b->d_tab, b->h_tab, sizeof(BatchSeq) * (size_t)n, hipMemcpyHostToDevice, sa));
  HIP_OK(hipEventRecord(b->ev_tab, sa));
  return 0;
}

static int batch_rows_total(const odo_tracker_batch* b) {
  int rows_total = 0, r = b->p.rows, c = b->p.cols;
  for (int l = 0; l < b->p.levels; l++) {
    if (r - 8 > 0 && c - 8 > 0) rows_total += r - 8;
    r /= 2; c /= 2;
  }
  return rows_total;
}

static inline dim3 batch_pyr_grid(const odo_tracker_batch* b, int n) {
  return dim3((b->p.cols + kPT - 1) / kPT, (b->p.rows + kPT - 1) / kPT, n);
}

// Front of the depth chain: blur, point selection, disparity scan for the step's slots (3 launches).
static int batch_depth_begin(odo_tracker_batch* b, hipStream_t s) {
  const odo_tracker_params& p = b->p;
  const odo_depth* d = b->depth[0];
  const int n = (int)b->ids.size();
  b->dk = 0; b->dn_launches = 0; b->dstage = 1; b->derr = 0; b->dpoll = d->poll != 0; b->dwaiting = false;
  hipLaunchKernelGGL(blur3x3_batch_kernel, grid2d(p.cols, p.rows, 2 * n), dim3(256), 0, s, (const BatchSeq*)b->d_tab, p.rows, p.cols);
  hipLaunchKernelGGL(depth_select_batch_kernel, dim3(kSelBlocks, n), dim3(kSelThreads), 0, s, (const BatchSeq*)b->d_tab, p.rows,
                     p.cols, d->boundary, d->grad_th);
  hipLaunchKernelGGL(depth_disparity_batch_kernel, dim3(kSelBlocks * kSelCap / 4, n), dim3(256), 0, s, (const BatchSeq*)b->d_tab,
                     p.rows, p.cols, d->boundary, d->max_disparity, d->ssd_th, d->K.f0, d->baseline);
  HIP_OK(hipGetLastError());
  return 0;
}

// Tail of the step on the depth stream: filters, depth pyramids, candidate lists, statistics + completion words.
static int batch_depth_tail(odo_tracker_batch* b, hipStream_t s) {
  const odo_tracker_params& p = b->p;
  const odo_depth* d = b->depth[0];
  const int n = (int)b->ids.size();
  const BatchSeq* tab = b->d_tab;
  hipLaunchKernelGGL(depth_finalize_batch_kernel, dim3(kDlmBlocks, n), dim3(kDlmBlock), 0, s, tab, 1, p.cols, d->photo_th,
                     d->min_depth, d->max_depth);
  hipLaunchKernelGGL(depth_pyramid_batch_kernel, grid2d(p.cols, p.rows, n), dim3(256), 0, s, tab);  // :252
  const int rows_total = batch_rows_total(b);
  const int with_lists = (b->with_lists && rows_total > 0) ? 1 : 0;
  if (with_lists) {
    if (s != b->ctx_a->stream) HIP_OK(hipStreamWaitEvent(s, b->ev_cur_img, 0));
    hipLaunchKernelGGL(kf_count_batch_kernel, dim3(rows_total, n), dim3(256), 0, s, tab);
    hipLaunchKernelGGL(kf_fill_batch_kernel, dim3(rows_total, n), dim3(256), 0, s, tab, p.K.f0, p.K.cx0, p.K.cy0);
  }
  hipLaunchKernelGGL(depth_stats_batch_kernel, dim3(1, n), dim3(kDlmBlock), 0, s, tab, 1, b->dn_launches, with_lists);
  HIP_OK(hipGetLastError());
  b->dstage = 2;
  return 0;
}

// Issues at most one depth-LM launch (all slots of the step) per call; enqueues the tail once every slot's LM has stopped.
static void batch_depth_pump(void* arg) {
  odo_tracker_batch* b = (odo_tracker_batch*)arg;
  if (b->dstage != 1) return;
  const odo_depth* d0 = b->depth[0];
  hipStream_t s = b->overlap ? b->ctx_b->stream : b->ctx_a->stream;
  bool all_stopped = true;
  int min_prog = 1 << 30;
  if (b->dpoll) {
    for (int i : b->ids) {
      volatile int* prog = b->depth[i]->h_prog;
      if (!prog[1]) { all_stopped = false; if (prog[0] < min_prog) min_prog = prog[0]; }
    }
  } else {
    all_stopped = false;
  }
  // launch k decides on evaluation k-1 and runs evaluation k: max_iters evaluations need max_iters + 1 launches
  const bool lm_over = (b->dk > d0->max_iters) || (b->dpoll && all_stopped);
  if (!lm_over) {
    if (b->dpoll && b->dk - min_prog > d0->run_ahead) {
      const auto now = std::chrono::steady_clock::now();
      if (!b->dwaiting) { b->dwaiting = true; b->dwait_since = now; }
      else if (now - b->dwait_since > std::chrono::seconds(2)) b->dpoll = false;  // never hang on a lost progress word
      return;
    }
    b->dwaiting = false;
    hipLaunchKernelGGL(depth_lm_step_batch_kernel, dim3(kDlmBlocks, (int)b->ids.size()), dim3(kDlmBlock), 0, s,
                       (const BatchSeq*)b->d_tab, b->dk, b->p.cols, d0->baseline, d0->K.f0, d0->huber_delta, d0->lambda,
                       d0->precision, d0->max_iters);
    b->dk++;
    b->dn_launches++;
    return;
  }
  if (batch_depth_tail(b, s)) b->derr = 1;
}

// Waits for every slot's completion word (bounded), takes the statistics. ok[e] = 0 when ComputeDepth failed for ids[e].
static int batch_depth_finish(odo_tracker_batch* b, int* ok) {
  hipStream_t s = b->overlap ? b->ctx_b->stream : b->ctx_a->stream;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i : b->ids) {
    odo_depth* d = b->depth[i];
    volatile int* done = d->h_prog + 4;
    while (done[0] != d->token) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { HIP_OK(hipStreamSynchronize(s)); break; }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (size_t e = 0; e < b->ids.size(); e++) {
    const int i = b->ids[e];
    odo_depth* d = b->depth[i];
    d->last = *d->h_stats;
    ok[e] = d->last.status == 0;
    b->last_valid[i] = d->last.n_valid;
    b->last_depth_iters[i] = d->last.iters;
  }
  return 0;
}

// The candidate lists built for slot i this step become its keyframe lists.
static void batch_adopt(odo_tracker_batch* b, int i) {
  odo_lm* m = b->lm[i];
  memcpy(m->cand[0].h_npts, b->h_cand_npts + (size_t)i * ODO_MAX_LEVELS, sizeof(int) * ODO_MAX_LEVELS);
  (void)lm_adopt_candidate(m, b->kf_img[i], b->kf_dep[i], b->frame_id[i]);  // 1 = no candidate: the Solve builds the lists
}

// Frame 0 of the slots in b->ids (ref: :95-145).
static int batch_init_set(odo_tracker_batch* b, const float* const* left_dev, const float* const* right_dev,
                          const float* abs_pose0) {
  HIP_OK(hipSetDevice(b->ctx_a->device));
  // stragglers of the previous step (LM launches queued ahead) may still run on stream A; the keyframe buffers are rewritten now
  HIP_OK(hipStreamSynchronize(b->ctx_a->stream));
  HIP_OK(hipStreamSynchronize(b->ctx_b->stream));
  const odo_tracker_params& p = b->p;
  const int n = (int)b->ids.size();
  if (n == 0) return 0;
  b->with_lists = !getenv("ODO_NO_CAND_LISTS");
  for (int i : b->ids) {
    b->frame_id[i] = 0; b->alive[i] = 0; b->hint_next[i] = b->prefetched[i] = nullptr;
    b->kf_img[i]->version = ++g_pyr_version;
  }
  if (batch_upload_table(b, left_dev, right_dev, true)) return -1;
  hipStream_t sa = b->ctx_a->stream;
  hipStream_t sb = b->overlap ? b->ctx_b->stream : sa;
  hipLaunchKernelGGL(image_pyramid_batch_kernel, batch_pyr_grid(b, n), dim3(kPyrThreads), 0, sa, (const BatchSeq*)b->d_tab);  // :130
  HIP_OK(hipEventRecord(b->ev_cur_img, sa));
  if (sb != sa) HIP_OK(hipStreamWaitEvent(sb, b->ev_tab, 0));
  if (batch_depth_begin(b, sb)) return -1;                                                // :102
  while (b->dstage == 1) batch_depth_pump(b);
  if (b->derr) { b->dstage = 0; return -1; }
  std::vector<int> ok(n, 0);
  if (batch_depth_finish(b, ok.data())) return -1;
  HIP_OK(hipStreamSynchronize(sa));
  b->dstage = 0;
  const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int bad = -1;
  for (int e = 0; e < n; e++) {
    const int i = b->ids[e];
    const float* a0 = abs_pose0 ? abs_pose0 + 16 * (size_t)i : eye;
    memcpy(&b->kf_abs[16 * (size_t)i], a0, sizeof(float) * 16);        // :143
    memcpy(&b->pose_to_kf[16 * (size_t)i], a0, sizeof(float) * 16);    // :98
    odo_lm_reset(b->lm[i], eye, p.lm_lambda);                          // :77-81
    b->lm[i]->kf_img_ver = b->lm[i]->kf_dep_ver = 0;
    b->n_keyframes[i] = 1;
    b->alive[i] = ok[e];
    if (!ok[e] && bad < 0) bad = i;
    if (ok[e] && b->with_lists) batch_adopt(b, i);
  }
  if (bad >= 0) return fail("Init 0-th frame failed! (sequence %d: number of valid after optimization is too small: %d)", bad,
                            b->last_valid[bad]);                       // :103-106
  return 0;
}

extern "C" int odo_tracker_batch_init(odo_tracker_batch* b, const float* const* left_dev, const float* const* right_dev,
                                      const float* abs_pose0 /* S x 16 column-major, or NULL for identity */) {
  if (!b || !left_dev || !right_dev) return fail("odo_tracker_batch_init: NULL arg");
  b->ids.clear();
  for (int i = 0; i < b->S; i++) {
    if ((left_dev[i] == nullptr) != (right_dev[i] == nullptr)) return fail("odo_tracker_batch_init: slot %d has one image only", i);
    if (left_dev[i]) b->ids.push_back(i);
    else b->alive[i] = 0;   // an empty slot
  }
  return batch_init_set(b, left_dev, right_dev, abs_pose0);
}

extern "C" int odo_tracker_batch_init_one(odo_tracker_batch* b, int slot, const float* left_dev, const float* right_dev,
                                          const float abs_pose0[16]) {
  if (!b || !left_dev || !right_dev || slot < 0 || slot >= b->S) return fail("odo_tracker_batch_init_one: bad arg");
  std::vector<const float*> l(b->S, nullptr), r(b->S, nullptr);
  std::vector<float> a0;
  l[slot] = left_dev; r[slot] = right_dev;
  if (abs_pose0) { a0.assign((size_t)b->S * 16, 0.0f); memcpy(&a0[16 * (size_t)slot], abs_pose0, sizeof(float) * 16); }
  b->ids.assign(1, slot);
  return batch_init_set(b, l.data(), r.data(), abs_pose0 ? a0.data() : nullptr);
}

// Optional: the left images of the NEXT step (NULL entries allowed). Their pyramids are built at the end of the current step
// on the pose-LM stream, which is idle while the depth stream finishes, instead of at the head of the next step.
extern "C" int odo_tracker_batch_hint_next(odo_tracker_batch* b, const float* const* next_left_dev) {
  if (!b) return fail("NULL batch tracker");
  for (int i = 0; i < b->S; i++) b->hint_next[i] = next_left_dev ? next_left_dev[i] : nullptr;
  return 0;
}

// One iteration of the frame loop for every slot that is given a frame. status[i]: 0 tracked; 1 the Solve failed (the runner
// carries on with the pseudo-identity, ref: src/lm_optimizer.cpp:60-61); -1 ComputeDepth failed on this frame (pose still
// written, the sequence stops: ref :230-232 breaks out of the loop); -2 the slot holds no running sequence (never initialised,
// or stopped earlier: outputs untouched); -3 no frame given for the slot this step (left_dev[i] == NULL: it sits the step out).
extern "C" int odo_tracker_batch_track(odo_tracker_batch* b, const float* const* left_dev, const float* const* right_dev,
                                       float* pose_to_keyframe /* S x 16 */, float* abs_pose /* S x 16 */,
                                       int* is_new_keyframe /* S */, float* motion_mag /* S */, int* status /* S */) {
  if (!b || !left_dev || !right_dev || !status) return fail("odo_tracker_batch_track: NULL arg");
  const int S = b->S;
  const odo_tracker_params& p = b->p;
  const auto f0 = std::chrono::steady_clock::now();
  HIP_OK(hipSetDevice(b->ctx_a->device));
  b->ids.clear();
  for (int i = 0; i < S; i++) {
    if (is_new_keyframe) is_new_keyframe[i] = 0;
    if (motion_mag) motion_mag[i] = 0.0f;
    if ((left_dev[i] == nullptr) != (right_dev[i] == nullptr)) return fail("odo_tracker_batch_track: slot %d has one image only", i);
    if (!left_dev[i]) { status[i] = -3; continue; }
    if (!b->alive[i]) { status[i] = -2; continue; }
    b->ids.push_back(i);
  }
  const int n = (int)b->ids.size();
  if (n == 0) return 0;
  b->with_lists = !getenv("ODO_NO_CAND_LISTS");
  // slots whose pyramid was prefetched at the end of the last step take it; the others get theirs built now
  std::vector<int> build;
  for (int i : b->ids) {
    b->frame_id[i]++;
    if (b->prefetched[i] && b->prefetched[i] == left_dev[i]) std::swap(b->cur_img[i], b->next_img[i]);
    else { build.push_back(i); b->cur_img[i]->version = ++g_pyr_version; }
    b->prefetched[i] = nullptr;
  }
  if (batch_upload_table(b, left_dev, right_dev, false)) return -1;
  hipStream_t sa = b->ctx_a->stream;
  hipStream_t sb = b->overlap ? b->ctx_b->stream : sa;
  if ((int)build.size() == n) {
    hipLaunchKernelGGL(image_pyramid_batch_kernel, batch_pyr_grid(b, n), dim3(kPyrThreads), 0, sa, (const BatchSeq*)b->d_tab);  // :205
  } else if (!build.empty()) {
    // a mix (some slots were hinted, some not): the stragglers one launch each, through the step's own table entries
    for (size_t e = 0; e < b->ids.size(); e++)
      for (int i : build)
        if (b->ids[e] == i)
          hipLaunchKernelGGL(image_pyramid_batch_kernel, batch_pyr_grid(b, 1), dim3(kPyrThreads), 0, sa, (const BatchSeq*)b->d_tab + e);
  }
  HIP_OK(hipEventRecord(b->ev_cur_img, sa));
  if (b->overlap) {
    HIP_OK(hipStreamWaitEvent(sb, b->ev_tab, 0));
    if (batch_depth_begin(b, sb)) return -1;                                              // :226 beside the Solve
  }
  std::vector<float> T((size_t)n * 16);
  std::vector<int> st(n, 0);
  std::vector<odo_lm*> lms(n);
  std::vector<const odo_pyr*> kfi(n), kfd(n), cur(n);
  for (int e = 0; e < n; e++) { const int i = b->ids[e]; lms[e] = b->lm[i]; kfi[e] = b->kf_img[i]; kfd[e] = b->kf_dep[i]; cur[e] = b->cur_img[i]; }
  const auto f1 = std::chrono::steady_clock::now();
  if (lm_solve_batch(n, lms.data(), kfi.data(), kfd.data(), cur.data(), T.data(), st.data(),
                     b->overlap ? batch_depth_pump : nullptr, b) < 0) {                    // :215
    while (b->dstage == 1) batch_depth_pump(b);
    std::vector<int> okd(n, 0);
    if (b->dstage == 2) (void)batch_depth_finish(b, okd.data());
    b->dstage = 0;
    return -1;
  }
  const auto f2 = std::chrono::steady_clock::now();
  // the next step's pyramids, on stream A behind the Solve (it is idle until the next step; the depth stream is still busy)
  {
    int m = 0;
    for (int i = 0; i < S; i++) {
      if (!b->hint_next[i]) continue;
      BatchSeq& q = b->h_tab[S + m];
      memset(&q, 0, sizeof(q));
      q.left = b->hint_next[i];
      q.cur_img = batch_pyr_out(b->next_img[i], p.smooth_image);
      b->next_img[i]->version = ++g_pyr_version;
      b->prefetched[i] = b->hint_next[i];
      b->hint_next[i] = nullptr;
      m++;
    }
    if (m > 0) {
      HIP_OK(hipMemcpyAsync(b->d_tab + S, b->h_tab + S, sizeof(BatchSeq) * (size_t)m, hipMemcpyHostToDevice, sa));
      hipLaunchKernelGGL(image_pyramid_batch_kernel, batch_pyr_grid(b, m), dim3(kPyrThreads), 0, sa, (const BatchSeq*)(b->d_tab + S));
    }
  }
  if (!b->overlap && batch_depth_begin(b, sb)) return -1;
  while (b->dstage == 1) batch_depth_pump(b);
  // :218 — poses are stored before the depth result is known
  std::vector<float> curp((size_t)n * 16);
  for (int e = 0; e < n; e++) {
    const int i = b->ids[e];
    const float* Ti = &T[16 * (size_t)e];
    float inv[16];
    if (!invert4(Ti, inv)) for (int k = 0; k < 16; k++) inv[k] = __builtin_nanf("");
    matmul4(&b->kf_abs[16 * (size_t)i], inv, &curp[16 * (size_t)e]);
    memcpy(&b->pose_to_kf[16 * (size_t)i], Ti, sizeof(float) * 16);
    if (pose_to_keyframe) memcpy(pose_to_keyframe + 16 * (size_t)i, Ti, sizeof(float) * 16);
    if (abs_pose) memcpy(abs_pose + 16 * (size_t)i, &curp[16 * (size_t)e], sizeof(float) * 16);
  }
  if (b->derr) { b->dstage = 0; return -1; }
  std::vector<int> okd(n, 0);
  if (batch_depth_finish(b, okd.data())) return -1;
  const auto f3 = std::chrono::steady_clock::now();
  b->dstage = 0;
  int any_depth_fail = 0;
  for (int e = 0; e < n; e++) {
    const int i = b->ids[e];
    b->last_evals[i] = b->lm[i]->last_evals;
    if (!okd[e]) { status[i] = -1; b->alive[i] = 0; any_depth_fail = 1; continue; }        // :230-232
    const float* Ti = &T[16 * (size_t)e];
    float ang[3];
    motion_angles(Ti, ang);                                                                // :253
    const float mot[6] = {fabsf(ang[0]), fabsf(ang[1]), fabsf(ang[2]), fabsf(Ti[12]), fabsf(Ti[13]), fabsf(Ti[14])};
    float mag = 0.0f;
    for (int k = 0; k < 6; k++) mag += mot[k] * p.keyframe_weight[k];                       // :257
    if (mag > p.keyframe_motion_th) {                                                      // :258
      std::swap(b->kf_img[i], b->cur_img[i]);                                              // :259 (the :251 rebuild has the same content)
      std::swap(b->kf_dep[i], b->pre_dep[i]);
      memcpy(&b->kf_abs[16 * (size_t)i], &curp[16 * (size_t)e], sizeof(float) * 16);       // :260
      b->n_keyframes[i]++;
      if (is_new_keyframe) is_new_keyframe[i] = 1;
      if (b->with_lists) batch_adopt(b, i);
    }
    odo_lm_reset(b->lm[i], Ti, 0.01f);                                                     // :261 / :268
    if (motion_mag) motion_mag[i] = mag;
    status[i] = st[e] ? 1 : 0;
  }
  if (any_depth_fail) fail("    depth failed!");
  b->tm_frame_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f0).count();
  b->tm_head_us += std::chrono::duration<double, std::micro>(f1 - f0).count();
  b->tm_solve_us += std::chrono::duration<double, std::micro>(f2 - f1).count();
  b->tm_depth_wait_us += std::chrono::duration<double, std::micro>(f3 - f2).count();
  b->tm_frames++;
  return 0;
}

// Host-clock averages per lock step since the last call (microseconds): whole call, head (table + pyramid launches), the
// batched Solve (depth launches are pumped from its wait loops), the wait for the depth chain after the Solve.
extern "C" int odo_tracker_batch_timing(odo_tracker_batch* b, double out[4]) {
  if (!b || !out) return fail("NULL arg");
  const double n = b->tm_frames > 0 ? (double)b->tm_frames : 1.0;
  out[0] = b->tm_frame_us / n; out[1] = b->tm_head_us / n; out[2] = b->tm_solve_us / n; out[3] = b->tm_depth_wait_us / n;
  b->tm_frame_us = b->tm_head_us = b->tm_solve_us = b->tm_depth_wait_us = 0.0; b->tm_frames = 0;
  return 0;
}

extern "C" int odo_tracker_batch_stats(const odo_tracker_batch* b, int* lm_evals, int* depth_iters, int* n_valid, int* n_kf) {
  if (!b) return fail("NULL batch tracker");
  for (int i = 0; i < b->S; i++) {
    if (lm_evals) lm_evals[i] = b->last_evals[i];
    if (depth_iters) depth_iters[i] = b->last_depth_iters[i];
    if (n_valid) n_valid[i] = b->last_valid[i];
    if (n_kf) n_kf[i] = b->n_keyframes[i];
  }
  return 0;
}
extern "C" int odo_tracker_batch_outputs(const odo_tracker_batch* b, int seq, const uint8_t** val, const float** disp,
                                         const float** dep) {
  if (!b || seq < 0 || seq >= b->S) return fail("odo_tracker_batch_outputs: bad arg");
  if (val) *val = b->d_val[seq];
  if (disp) *disp = b->d_disp[seq];
  if (dep) *dep = b->d_dep[seq];
  return 0;
}
extern "C" int odo_tracker_batch_size(const odo_tracker_batch* b) { return b ? b->S : 0; }
extern "C" odo_ctx* odo_tracker_batch_ctx(odo_tracker_batch* b) { return b ? b->ctx_a : nullptr; }
