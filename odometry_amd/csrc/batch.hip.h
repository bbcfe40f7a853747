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
  int* gave_up;         // the estimator's give-up word when the depth LM ran in the persistent launch (else null)
  // keyframe-candidate point lists
  KfLevels kl;
  int* rowcnt;
  int* npts;            // device
  int* npts_host;       // host-mapped copy, written by the job's last launch
  PointList pl[ODO_MAX_LEVELS_K];
};

#ifndef ODO_BATCH_PYR_THREADS
#define ODO_BATCH_PYR_THREADS 256
#endif
constexpr int kBatchPyrThreads = ODO_BATCH_PYR_THREADS;
__global__ void __launch_bounds__(kBatchPyrThreads) image_pyramid_batch_kernel(const BatchSeq* __restrict__ tab) {
  const BatchSeq& q = tab[blockIdx.z];
  image_pyramid_fused_kernel_body<kBatchPyrThreads>(q.left, q.cur_img);
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
  depth_stats_kernel_body(run_lm, n_launches, q.counts, q.dstate, q.stats, q.dprog + 4, q.dtoken, q.gave_up);
}

}  // namespace odo

// One batched stream-B job: ComputeDepth, depth pyramid and candidate lists of a set of slots (table entry e <-> slot ids[e]).
// Two of them exist: `late` (slots whose frame of THIS step has no finished job yet) and `ahead` (the frames announced for the
// NEXT step, odo_tracker_batch_hint_next_pair: the depth stream then works a step ahead of the pose LM).
struct BatchChain {
  std::vector<int> ids;
  std::vector<int> par;      // output parity per entry (each slot has two output sets)
  std::vector<long> tag;     // frame id per entry (tags the candidate lists)
  std::vector<const float*> left, right;   // per SLOT: the pairs the job works on (copies: the job may outlive the caller's arrays)
  std::vector<DepthLmStats> stats;         // per entry: the job's results, taken over by the calling thread (batch_chain_absorb)
  bool img_is_next;          // the candidate lists read next_img (the announced frame's pyramid) instead of cur_img / kf_img
  bool first;                // frame 0: the image pyramids are the keyframe buffers
  int tag_add;               // 0 = the frame being tracked, 1 = the next one
  BatchSeq* h_tab;           // pinned, S entries
  BatchSeq* d_tab;
  hipEvent_t img_ready;      // the image pyramids the candidate lists read are complete after this event
  int k, n_launches, stage;  // stage: 0 idle, 1 depth-LM launches, 2 tail enqueued, 3 complete (results not absorbed yet)
  std::atomic<int> complete; // 1: the job has been run (set by the thread that ran it, read by the tracker's owner)
  bool poll, waiting, with_lists;
  int err;
  std::chrono::steady_clock::time_point wait_since;
  // DepthOptimization of every entry in ONE persistent launch (depth_lm_persistent_batch_kernel) instead of a launch per iteration
  bool persistent;           // this run of the chain used it
  bool use_persistent;       // decided ONCE per run, in batch_chain_begin (the tables' gave_up pointers and the launch choice must agree)
  bool no_persist_once;      // the chain is being run again after a give-up: step launches
  DepthPersistArgs* h_ptab;  // pinned, S entries
  DepthPersistArgs* d_ptab;
};

struct odo_tracker_batch {
  odo_tracker_params p;
  int S;           // slots
  odo_ctx* ctx_a;  // image pyramids + pose LM
  odo_ctx* ctx_b;  // ComputeDepth, depth pyramids, candidate lists
  odo_ctx* ctx_c;  // the next step's image pyramids
  std::vector<odo_lm*> lm;
  std::vector<odo_depth*> depth;
  std::vector<odo_pyr*> kf_img, kf_dep, cur_img, next_img;
  std::vector<odo_pyr*> pre_dep[2];        // per output parity
  std::vector<uint8_t*> d_val[2];
  std::vector<float*> d_disp[2], d_dep[2];
  std::vector<int> next_par;               // parity the slot's next stream-B job writes
  std::vector<int> res_par;                // parity that holds the slot's latest collected job
  std::vector<int> res_ok;                 // that job's ComputeDepth status (1 = ok)
  std::vector<int> out_par;                // parity that holds the outputs of the slot's last tracked frame
  std::vector<const float*> res_left, res_right;  // the pair it was computed from (NULL: none)
  std::vector<long> res_tag;               // frame id it was tagged with
  std::vector<float> kf_abs, pose_to_kf;   // S x 16
  std::vector<int> n_keyframes, alive, last_evals, last_depth_iters, last_valid;
  std::vector<long> frame_id;              // per slot: frames tracked since its init (tags the candidate lists)
  std::vector<const float*> hint_next, hint_next_right, prefetched;  // announced next pair / image whose pyramid sits in next_img
  BatchSeq* h_pyr;  // pinned, 2 S entries: pyramid-only launches ([0, S) this step's on stream A, [S, 2S) the next step's on stream C)
  BatchSeq* d_pyr;
  int* h_cand_npts;  // host-mapped, S x 2 x ODO_MAX_LEVELS
  int* d_cand_npts;  // its device alias
  hipEvent_t ev_cur_img, ev_next;
  int overlap;       // 1: the depth chain runs on stream B beside the pose LM on stream A (one host thread feeds both)
  int depth_ahead;   // 1: announced pairs get their stream-B job a step early (ODO_NO_DEPTH_AHEAD=1: off)
  int early_solve;   // 1: the next lock step's Solve is started as soon as this step's decisions are taken (ODO_NO_EARLY_SOLVE=1: off)
  BatchChain late, ahead;
  // overlap: a helper host thread feeds stream B — it runs the posted chains one after the other, each up to its completion
  // words (a slot's depth object serves one job at a time) — while the calling thread feeds the pose LM on stream A.
  std::thread worker;
  BatchChain* w_ring[2];
  std::atomic<long> w_posted, w_done;
  std::atomic<int> w_quit;
  std::vector<int> ids;    // slots of the lock step in flight (pose LM)
  bool with_lists;
  hipEvent_t ev_a;         // a point on stream A (the rebuild of next_img on stream C goes behind an abandoned early Solve)
  int depth_persist_bails, depth_persist_strikes, depth_persist_clean;   // chains run again after a give-up (lifetime) / strikes that count / clean chains since
  int dead;                // 1: a wait timed out with work in flight — every later call fails fast
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
  if (b->worker.joinable()) {
    b->w_quit.store(1, std::memory_order_release);  // chains already posted are run first
    b->worker.join();
  }
  for (odo_ctx* c : {b->ctx_a, b->ctx_b, b->ctx_c}) if (c) (void)hipStreamSynchronize(c->stream);
  for (odo_lm* m : b->lm) odo_lm_destroy(m);
  for (odo_depth* d : b->depth) odo_depth_destroy(d);
  for (auto* v : {&b->kf_img, &b->kf_dep, &b->cur_img, &b->next_img, &b->pre_dep[0], &b->pre_dep[1]})
    for (odo_pyr* q : *v) odo_pyramid_destroy(q);
  for (int k = 0; k < 2; k++) {
    for (uint8_t* q : b->d_val[k]) if (q) (void)hipFree(q);
    for (float* q : b->d_disp[k]) if (q) (void)hipFree(q);
    for (float* q : b->d_dep[k]) if (q) (void)hipFree(q);
  }
  for (BatchChain* c : {&b->late, &b->ahead}) {
    if (c->h_tab) (void)hipHostFree(c->h_tab);
    if (c->d_tab) (void)hipFree(c->d_tab);
    if (c->h_ptab) (void)hipHostFree(c->h_ptab);
    if (c->d_ptab) (void)hipFree(c->d_ptab);
  }
  if (b->h_pyr) (void)hipHostFree(b->h_pyr);
  if (b->d_pyr) (void)hipFree(b->d_pyr);
  if (b->h_cand_npts) (void)hipHostFree(b->h_cand_npts);
  if (b->ev_cur_img) (void)hipEventDestroy(b->ev_cur_img);
  if (b->ev_next) (void)hipEventDestroy(b->ev_next);
  if (b->ev_a) (void)hipEventDestroy(b->ev_a);
  odo_ctx_destroy(b->ctx_c);
  odo_ctx_destroy(b->ctx_b);
  odo_ctx_destroy(b->ctx_a);
  delete b;
  return 0;
}

// Largest lock step whose inverse-depth LMs run in ONE persistent launch (ODO_BATCH_DEPTH_PERSIST_MAX; the launch places at most
// eight sequences). Round 6: 8 — see batch_depth_persist_ok.
static int batch_depth_persist_max() {
  static const int n = getenv("ODO_BATCH_DEPTH_PERSIST_MAX") ? atoi(getenv("ODO_BATCH_DEPTH_PERSIST_MAX")) : 8;
  return n < 8 ? n : 8;
}
extern "C" int odo_tracker_batch_depth_persistent_stats(const odo_tracker_batch* b, int* on, int* chains_redone) {
  if (!b) return fail("odo_tracker_batch_depth_persistent_stats: NULL tracker");
  if (on) *on = (b->S <= batch_depth_persist_max() && b->depth_persist_strikes < 3 && b->depth[0] && b->depth[0]->persist_cfg &&
                 (!getenv("ODO_BATCH_DEPTH_PERSIST") || atoi(getenv("ODO_BATCH_DEPTH_PERSIST")) != 0)) ? 1 : 0;
  if (chains_redone) *chains_redone = b->depth_persist_bails;
  return 0;
}

static void batch_worker_main(odo_tracker_batch* b);

extern "C" int odo_tracker_batch_create(int device, const odo_tracker_params* p, int n_sequences, odo_tracker_batch** out) {
  if (!p || !out) return fail("odo_tracker_batch_create: NULL arg");
  *out = nullptr;
  if (n_sequences < 1 || n_sequences > 64) return fail("odo_tracker_batch_create: 1..64 sequences, got %d", n_sequences);
  if (p->levels > 4) return fail("odo_tracker_batch_create: at most 4 pyramid levels (the one-launch pyramid kernels)");
  odo_tracker_batch* b = new (std::nothrow) odo_tracker_batch();
  if (!b) return fail("out of memory");
  const int S = n_sequences;
  b->p = *p; b->S = S;
  b->ctx_a = b->ctx_b = b->ctx_c = nullptr; b->h_pyr = b->d_pyr = nullptr; b->h_cand_npts = b->d_cand_npts = nullptr;
  b->ev_cur_img = b->ev_next = b->ev_a = nullptr; b->dead = 0;
  for (BatchChain* c : {&b->late, &b->ahead}) {
    c->h_tab = c->d_tab = nullptr; c->stage = 0; c->err = 0; c->img_ready = nullptr; c->complete.store(0);
    c->h_ptab = c->d_ptab = nullptr; c->persistent = c->no_persist_once = c->use_persistent = false;
  }
  b->depth_persist_bails = b->depth_persist_strikes = b->depth_persist_clean = 0;
  b->w_ring[0] = b->w_ring[1] = nullptr; b->w_posted.store(0); b->w_done.store(0); b->w_quit.store(0);
  b->tm_frame_us = b->tm_head_us = b->tm_solve_us = b->tm_depth_wait_us = 0.0; b->tm_frames = 0;
  b->overlap = (p->overlap_depth != 0) && !getenv("ODO_BATCH_NO_OVERLAP");
  b->depth_ahead = b->overlap && !getenv("ODO_NO_DEPTH_AHEAD");
  b->early_solve = b->overlap && !getenv("ODO_NO_EARLY_SOLVE");
  b->lm.assign(S, nullptr); b->depth.assign(S, nullptr);
  b->kf_img.assign(S, nullptr); b->kf_dep.assign(S, nullptr); b->cur_img.assign(S, nullptr); b->next_img.assign(S, nullptr);
  for (int k = 0; k < 2; k++) {
    b->pre_dep[k].assign(S, nullptr); b->d_val[k].assign(S, nullptr); b->d_disp[k].assign(S, nullptr); b->d_dep[k].assign(S, nullptr);
  }
  b->next_par.assign(S, 0); b->res_par.assign(S, 0); b->res_ok.assign(S, 0); b->out_par.assign(S, 0);
  b->res_left.assign(S, nullptr); b->res_right.assign(S, nullptr); b->res_tag.assign(S, -1);
  b->kf_abs.assign((size_t)S * 16, 0.0f); b->pose_to_kf.assign((size_t)S * 16, 0.0f);
  b->n_keyframes.assign(S, 0); b->alive.assign(S, 0); b->last_evals.assign(S, 0); b->last_depth_iters.assign(S, 0);
  b->last_valid.assign(S, 0); b->frame_id.assign(S, 0);
  b->hint_next.assign(S, nullptr); b->hint_next_right.assign(S, nullptr); b->prefetched.assign(S, nullptr);
  const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const size_t n = (size_t)p->rows * p->cols;
  const bool lm_prio = !(getenv("ODO_LM_PRIORITY") && atoi(getenv("ODO_LM_PRIORITY")) == 0);
  bool ok = (lm_prio ? odo_ctx_create_high_priority(device, &b->ctx_a) : odo_ctx_create(device, &b->ctx_a)) == 0 &&
            odo_ctx_create(device, &b->ctx_b) == 0 && odo_ctx_create(device, &b->ctx_c) == 0;
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
    for (int k = 0; k < 2 && ok; k++) {
      ok = ok && pyr_alloc(b->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_DEPTH, &b->pre_dep[k][i], false) == 0;
      ok = ok && hipMalloc((void**)&b->d_val[k][i], n) == hipSuccess && hipMalloc((void**)&b->d_disp[k][i], sizeof(float) * n) == hipSuccess &&
           hipMalloc((void**)&b->d_dep[k][i], sizeof(float) * n) == hipSuccess;
    }
  }
  for (BatchChain* c : {&b->late, &b->ahead}) {
    ok = ok && hipHostMalloc((void**)&c->h_tab, sizeof(BatchSeq) * (size_t)S, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_tab, sizeof(BatchSeq) * (size_t)S) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_ptab, sizeof(DepthPersistArgs) * (size_t)S, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_ptab, sizeof(DepthPersistArgs) * (size_t)S) == hipSuccess;
  }
  ok = ok && hipHostMalloc((void**)&b->h_pyr, sizeof(BatchSeq) * 2 * (size_t)S, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&b->d_pyr, sizeof(BatchSeq) * 2 * (size_t)S) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&b->h_cand_npts, sizeof(int) * 2 * ODO_MAX_LEVELS * (size_t)S,
                           hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void**)&b->d_cand_npts, b->h_cand_npts, 0) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&b->ev_cur_img, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&b->ev_next, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&b->ev_a, hipEventDisableTiming) == hipSuccess;
  if (ok && !getenv("ODO_LM_TRACE")) for (odo_lm* m : b->lm) m->record = 0;
  if (!ok) {
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", g_err);
    odo_tracker_batch_destroy(b);
    return fail("odo_tracker_batch_create failed: %s", keep);
  }
  if (b->overlap) b->worker = std::thread(batch_worker_main, b);
  *out = b;
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

// Keyframe-candidate point lists of every frame built AHEAD on the depth stream (13 us of chip time per frame; a frame that is then
// promoted, one in eight, hands its lists over by a buffer swap), or built by the first Solve against a new keyframe (~60 us in front
// of that lock step's launches)? Ahead while the lock step is latency-bound — up to four sequences: S = 4 9 470-9 520 frames/s
// against 9 330-9 390 —, lazily once the chip is throughput-bound on the front end of that many frames: S = 8 13 290-13 420 ->
// 14 000-14 180 (round 5). ODO_NO_CAND_LISTS=1 / ODO_CAND_LISTS=1 force one or the other. Results are the same either way.
static bool batch_lists_ahead(int n_in_step) {
  if (getenv("ODO_NO_CAND_LISTS")) return false;
  if (getenv("ODO_CAND_LISTS")) return true;
  return n_in_step <= 4;
}

// One pyramid-only launch: the image pyramids `dst[i]` of `src[i]` for the listed slots, on stream s, through table rows
// [row0, row0 + n) of the pyramid table.
static int batch_build_pyramids(odo_tracker_batch* b, const std::vector<int>& slots, const std::vector<const float*>& src,
                                const std::vector<odo_pyr*>& dst, int row0, hipStream_t s) {
  const int n = (int)slots.size();
  if (n == 0) return 0;
  for (int e = 0; e < n; e++) {
    const int i = slots[e];
    BatchSeq& q = b->h_pyr[row0 + e];
    memset(&q, 0, sizeof(q));
    q.left = src[i];
    q.cur_img = batch_pyr_out(dst[i], b->p.smooth_image);
    dst[i]->version = ++g_pyr_version;
  }
  HIP_OK(hipMemcpyAsync(b->d_pyr + row0, b->h_pyr + row0, sizeof(BatchSeq) * (size_t)n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(image_pyramid_batch_kernel, batch_pyr_grid(b, n), dim3(kBatchPyrThreads), 0, s, (const BatchSeq*)(b->d_pyr + row0));
  HIP_OK(hipGetLastError());
  return 0;
}

static bool batch_depth_persist_ok(const odo_tracker_batch* b, const BatchChain* c);
static int batch_chain_run(odo_tracker_batch* b, BatchChain* c);
// Fills and uploads a chain's table and enqueues the front of its depth job (blur, point selection, disparity scan) on stream s.
// The chain describes the job: ids, the pairs (left / right per slot), which image pyramid the candidate lists read, the
// frame tag. Entry e writes the slot's next output parity.
static int batch_chain_begin(odo_tracker_batch* b, BatchChain* c, hipStream_t s) {
  const odo_tracker_params& p = b->p;
  const int n = (int)c->ids.size();
  c->par.assign(n, 0); c->tag.assign(n, -1); c->stats.assign(n, DepthLmStats());
  c->stage = 0; c->err = 0; c->k = 0; c->n_launches = 0; c->waiting = false; c->persistent = false;
  // batch_depth_persist_ok reads state another chain's give-up can change (depth_persist_strikes): asked once, here
  c->use_persistent = batch_depth_persist_ok(b, c);
  if (n == 0) { c->stage = 3; return 0; }
  for (int e = 0; e < n; e++) {
    const int i = c->ids[e];
    const int par = b->next_par[i];
    b->next_par[i] ^= 1;
    c->par[e] = par;
    c->tag[e] = b->frame_id[i] + c->tag_add;
    BatchSeq& q = c->h_tab[e];
    odo_depth* d = b->depth[i];
    odo_lm* m = b->lm[i];
    odo_pyr* img = c->first ? b->kf_img[i] : c->img_is_next ? b->next_img[i] : b->cur_img[i];
    odo_pyr* dep = b->pre_dep[par][i];
    memset(&q, 0, sizeof(q));
    q.left = c->left[i]; q.right = c->right[i];
    q.cur_img = batch_pyr_out(img, p.smooth_image);
    q.pre_dep = batch_pyr_out(dep, 0);
    dep->version = ++g_pyr_version;
    q.bl = d->d_bl; q.br = d->d_br; q.disp = b->d_disp[par][i]; q.dep = b->d_dep[par][i]; q.d0 = d->d_d0; q.scratch = d->d_scratch;
    q.val = b->d_val[par][i]; q.matched = d->d_matched; q.pts = d->d_pts; q.cnt = d->d_cnt;
    q.dstate = d->d_lmstate; q.part_e = d->d_part_e; q.part_n = d->d_part_n; q.counts = d->d_counts;
    q.stats = d->d_stats_map; q.dprog = d->d_prog;
    q.gave_up = c->use_persistent ? d->d_gave_up : nullptr;
    d->token++;
    q.dtoken = d->token;
    d->h_prog[0] = 0; d->h_prog[1] = 0;  // the slot's previous job is complete: nobody is polling these
    int rows_total = 0;
    LmCandSet& cs = m->cand[par];
    cs.tag = -1;
    if (c->with_lists) {
      if (lm_lists_layout(m, cs.pl, cs.pl_cap, cs.d_rowcnt, cs.rows_cap, img, dep, s, &q.kl, &rows_total)) return -1;
      if (rows_total > 0) cs.tag = c->tag[e];
    }
    q.rowcnt = cs.d_rowcnt; q.npts = cs.d_npts; q.npts_host = b->d_cand_npts + ((size_t)i * 2 + par) * ODO_MAX_LEVELS;
    for (int l = 0; l < ODO_MAX_LEVELS; l++) q.pl[l] = cs.pl[l];
  }
  HIP_OK(hipMemcpyAsync(c->d_tab, c->h_tab, sizeof(BatchSeq) * (size_t)n, hipMemcpyHostToDevice, s));
  const odo_depth* d = b->depth[0];
  c->poll = d->poll != 0;
  const BatchSeq* tab = c->d_tab;
  hipLaunchKernelGGL(blur3x3_batch_kernel, grid2d(p.cols, p.rows, 2 * n), dim3(256), 0, s, tab, p.rows, p.cols);
  hipLaunchKernelGGL(depth_select_batch_kernel, dim3(kSelBlocks, n), dim3(kSelThreads), 0, s, tab, p.rows, p.cols, d->boundary,
                     d->grad_th);
  hipLaunchKernelGGL(depth_disparity_batch_kernel, dim3(kSelBlocks * kSelCap / 4, n), dim3(256), 0, s, tab, p.rows, p.cols,
                     d->boundary, d->max_disparity, d->ssd_th, d->K.f0, d->baseline);
  HIP_OK(hipGetLastError());
  c->stage = 1;
  return 0;
}

// Tail of a chain: filters, depth pyramids, candidate lists, statistics + completion words.
static int batch_chain_tail(odo_tracker_batch* b, BatchChain* c, hipStream_t s) {
  const odo_tracker_params& p = b->p;
  const odo_depth* d = b->depth[0];
  const int n = (int)c->ids.size();
  const BatchSeq* tab = c->d_tab;
  if (!c->persistent)   // (the persistent launch writes back and counts itself)
    hipLaunchKernelGGL(depth_finalize_batch_kernel, dim3(kDlmBlocks, n), dim3(kDlmBlock), 0, s, tab, 1, p.cols, d->photo_th,
                       d->min_depth, d->max_depth);
  hipLaunchKernelGGL(depth_pyramid_batch_kernel, grid2d(p.cols, p.rows, n), dim3(256), 0, s, tab);  // :252
  const int rows_total = batch_rows_total(b);
  const int with_lists = (c->with_lists && rows_total > 0) ? 1 : 0;
  if (with_lists) {
    if (c->img_ready) HIP_OK(hipStreamWaitEvent(s, c->img_ready, 0));
    hipLaunchKernelGGL(kf_count_batch_kernel, dim3(rows_total, n), dim3(256), 0, s, tab);
    hipLaunchKernelGGL(kf_fill_batch_kernel, dim3(rows_total, n), dim3(256), 0, s, tab, p.K.f0, p.K.cx0, p.K.cy0);
  }
  hipLaunchKernelGGL(depth_stats_batch_kernel, dim3(1, n), dim3(kDlmBlock), 0, s, tab, 1, c->n_launches, with_lists);
  HIP_OK(hipGetLastError());
  c->stage = 2;
  return 0;
}

// Issues at most one depth-LM launch (all slots of the chain) per call; enqueues the tail once every slot's LM has stopped.
// The inverse-depth LM of every entry in one persistent launch: up to eight sequences. Up to four each have an XCD of their own (block
// classes 4 .. 7; the batched pose LM's sequences sit on 0 .. 3 with all 32 CUs of their XCDs); more than that share XCDs with the
// pose LM (16 of 32 CUs each at S = 8). Round 5 kept S > 4 on the launch per iteration: the persistent launch was no faster there
// (its 80 workgroups of 512 threads held the CUs the front end of eight frames was short of). With round 6's builds — the depth
// kernels under the occupancy-first scheduler: depth_lm_persistent_batch_kernel 62 VGPRs, eight waves per SIMD; the batched pose
// launch 203 — it is: S = 8 14 230-14 450 -> 16 600-17 050 frames/s, no chain redone in 20 000 lock steps (tools/batch_soak.py 8),
// every pass bit-identical; S = 5 / 6: 12 370 / 14 010. S > 8 keeps the launch per iteration.
// ODO_BATCH_DEPTH_PERSIST=0: off. A sequence whose launch gives up (a wait ran out: its workgroups were not all resident) makes the
// whole chain run again on the step launches (batch_chain_run); three such chains switch the launch off for this tracker.
static bool batch_depth_persist_ok(const odo_tracker_batch* b, const BatchChain* c) {
  static const bool on = !getenv("ODO_BATCH_DEPTH_PERSIST") || atoi(getenv("ODO_BATCH_DEPTH_PERSIST")) != 0;
  const odo_depth* d0 = b->depth[0];
  return on && !c->no_persist_once && b->depth_persist_strikes < 3 && (int)c->ids.size() <= batch_depth_persist_max() && d0->persist_cfg && d0->max_iters <= kDpMaxIters &&
         c->h_ptab && c->d_ptab;
}
static int batch_chain_persistent(odo_tracker_batch* b, BatchChain* c, hipStream_t s) {
  const int n = (int)c->ids.size();
  for (int e = 0; e < n; e++) {
    odo_depth* d = b->depth[c->ids[e]];
    const BatchSeq& q = c->h_tab[e];
    if ((++d->persist_epoch & 0xffu) == 0u)
      HIP_OK(hipMemsetAsync(d->d_xbuf, 0, sizeof(unsigned long long) * 2 * kDlmBlocks * 2, s));
    DepthPersistArgs& a = c->h_ptab[e];
    memset(&a, 0, sizeof(a));
    a.left = q.left; a.right = q.right; a.cols = b->p.cols; a.pts = d->d_pts; a.cnt = d->d_cnt; a.d0 = d->d_d0; a.matched = d->d_matched;
    a.state_out = d->d_lmstate;
    a.tx = d->baseline; a.fx = d->K.f0; a.huber_delta = d->huber_delta; a.lambda0 = d->lambda; a.precision = d->precision;
    a.max_iters = d->max_iters; a.photo_th = d->photo_th; a.min_depth = d->min_depth; a.max_depth = d->max_depth;
    a.val = q.val; a.dep = q.dep; a.counts = d->d_counts; a.xbuf = d->d_xbuf; a.epoch = d->persist_epoch; a.wait_ticks = d->persist_wait;
    a.gave_up = d->d_gave_up; a.fault = d->persist_fault; a.home = -1; a.cls = 4;
  }
  HIP_OK(hipMemcpyAsync(c->d_ptab, c->h_ptab, sizeof(DepthPersistArgs) * (size_t)n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(depth_lm_persistent_batch_kernel, dim3(8 * kDpK), dim3(kDpThreads), 0, s, (const DepthPersistArgs*)c->d_ptab, n,
                     device_xcc_ids(b->ctx_b->device));
  HIP_OK(hipGetLastError());
  c->persistent = true;
  c->n_launches = 0;
  return 0;
}

static void batch_chain_pump(odo_tracker_batch* b, BatchChain* c, hipStream_t s) {
  if (c->stage != 1) return;
  if (c->k == 0 && c->use_persistent) {
    if (batch_chain_persistent(b, c, s) || batch_chain_tail(b, c, s)) { c->err = 1; c->stage = 0; }
    return;
  }
  const odo_depth* d0 = b->depth[0];
  bool all_stopped = true;
  int min_prog = 1 << 30;
  if (c->poll) {
    for (int i : c->ids) {
      volatile int* prog = b->depth[i]->h_prog;
      if (!prog[1]) { all_stopped = false; if (prog[0] < min_prog) min_prog = prog[0]; }
    }
  } else {
    all_stopped = false;
  }
  // launch k decides on evaluation k-1 and runs evaluation k: max_iters evaluations need max_iters + 1 launches
  const bool lm_over = (c->k > d0->max_iters) || (c->poll && all_stopped);
  if (!lm_over) {
    if (c->poll && c->k - min_prog > d0->run_ahead) {
      const auto now = std::chrono::steady_clock::now();
      if (!c->waiting) { c->waiting = true; c->wait_since = now; }
      else if (now - c->wait_since > std::chrono::seconds(2)) c->poll = false;  // never hang on a lost progress word
      return;
    }
    c->waiting = false;
    hipLaunchKernelGGL(depth_lm_step_batch_kernel, dim3(kDlmBlocks, (int)c->ids.size()), dim3(kDlmBlock), 0, s,
                       (const BatchSeq*)c->d_tab, c->k, b->p.cols, d0->baseline, d0->K.f0, d0->huber_delta, d0->lambda,
                       d0->precision, d0->max_iters);
    c->k++;
    c->n_launches++;
    return;
  }
  if (batch_chain_tail(b, c, s)) { c->err = 1; c->stage = 0; }
}

// A whole chain on the calling thread: front, depth-LM launches, tail, then the wait for every slot's completion word (bounded)
// and a copy of the statistics. The results stay in the chain until batch_chain_absorb.
static int batch_chain_run(odo_tracker_batch* b, BatchChain* c) {
  hipStream_t s = b->overlap ? b->ctx_b->stream : b->ctx_a->stream;
  if (batch_chain_begin(b, c, s)) { c->err = 1; c->stage = 3; c->complete.store(1, std::memory_order_release); return -1; }
  while (c->stage == 1) batch_chain_pump(b, c, s);
  if (c->stage != 2) { c->stage = 3; c->complete.store(1, std::memory_order_release); return c->err ? -1 : 0; }
  const auto t0 = std::chrono::steady_clock::now();
  for (int i : c->ids) {
    odo_depth* d = b->depth[i];
    volatile int* done = d->h_prog + 4;
    while (done[0] != d->token) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { (void)hipStreamSynchronize(s); break; }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (size_t e = 0; e < c->ids.size(); e++) c->stats[e] = *b->depth[c->ids[e]]->h_stats;
  if (c->persistent) {
    bool gave_up = false;
    for (const DepthLmStats& st : c->stats) gave_up = gave_up || st.status == -2;
    if (gave_up && !c->no_persist_once) {
      // a persistent launch gave up: the same chain again on the launches per iteration (same output parity, same tags)
      b->depth_persist_bails++;
      b->depth_persist_strikes++;
      b->depth_persist_clean = 0;
      for (int i : c->ids) b->next_par[i] ^= 1;
      c->no_persist_once = true;
      const int rc = batch_chain_run(b, c);
      c->no_persist_once = false;
      return rc;
    }
    if (!gave_up && b->depth_persist_strikes > 0 && ++b->depth_persist_clean >= 1024) { b->depth_persist_strikes = 0; b->depth_persist_clean = 0; }
  }
  c->stage = 3;
  c->complete.store(1, std::memory_order_release);
  return 0;
}

// Helper thread: runs the posted chains in order.
static void batch_worker_main(odo_tracker_batch* b) {
  (void)hipSetDevice(b->ctx_b->device);
  int idle_spins = 0;
  for (;;) {
    const long done = b->w_done.load(std::memory_order_relaxed);
    if (b->w_posted.load(std::memory_order_acquire) <= done) {
      if (b->w_quit.load(std::memory_order_acquire)) return;
      if (++idle_spins > 20000000) std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    idle_spins = 0;
    (void)batch_chain_run(b, b->w_ring[done & 1]);
    b->w_done.store(done + 1, std::memory_order_release);
  }
}
// Hands a chain to the helper thread (or runs it right here when there is none). At most two may be outstanding.
static void batch_chain_post(odo_tracker_batch* b, BatchChain* c) {
  c->complete.store(0, std::memory_order_relaxed);
  if (!b->worker.joinable()) { (void)batch_chain_run(b, c); return; }
  const long n = b->w_posted.load(std::memory_order_relaxed);
  b->w_ring[n & 1] = c;
  b->w_posted.store(n + 1, std::memory_order_release);
}
// Waits until the helper thread has run everything posted so far (bounded).
static int batch_wait_chains(odo_tracker_batch* b) {
  if (!b->worker.joinable()) return 0;
  const long posted = b->w_posted.load(std::memory_order_acquire);
  const auto q0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (b->w_done.load(std::memory_order_acquire) < posted) {
    if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - q0 > std::chrono::seconds(10)) {
      b->dead = 1;   // the helper still owns its chains: every later call fails fast instead of refilling them under it
      return fail("    depth failed! (the depth stream's helper thread did not finish within 10 s)");
    }
  }
  return 0;
}

// Takes a complete chain's results over: per slot the parity, status, statistics, the pair the job was computed from, the
// candidate lists' per-level counts. Called by the thread that owns the tracker state, never by the helper.
static int batch_chain_absorb(odo_tracker_batch* b, BatchChain* c) {
  if (c->stage != 3) return 0;
  c->stage = 0;
  if (c->err) return -1;
  for (size_t e = 0; e < c->ids.size(); e++) {
    const int i = c->ids[e];
    b->depth[i]->last = c->stats[e];
    b->res_par[i] = c->par[e];
    b->res_ok[i] = c->stats[e].status == 0;
    b->res_left[i] = c->left[i]; b->res_right[i] = c->right[i];
    b->res_tag[i] = c->tag[e];
    memcpy(b->lm[i]->cand[c->par[e]].h_npts, b->h_cand_npts + ((size_t)i * 2 + c->par[e]) * ODO_MAX_LEVELS, sizeof(int) * ODO_MAX_LEVELS);
  }
  return 0;
}

// Sets a chain up for the listed slots (the job itself is described by the caller through the fields set here).
static void batch_chain_fill(odo_tracker_batch* b, BatchChain* c, const std::vector<int>& ids, const float* const* left,
                             const float* const* right, bool first, bool img_is_next, int tag_add, hipEvent_t img_ready) {
  c->ids = ids;
  c->left.assign(b->S, nullptr); c->right.assign(b->S, nullptr);
  for (int i : ids) { c->left[i] = left[i]; c->right[i] = right[i]; }
  c->first = first; c->img_is_next = img_is_next; c->tag_add = tag_add; c->img_ready = img_ready;
  c->with_lists = b->with_lists;
  c->err = 0;
}

// Everything the helper thread has been given is run to completion and taken over (a chain posted ahead keeps its results
// with its slots), and the device is quiet afterwards.
static int batch_quiesce(odo_tracker_batch* b) {
  if (batch_wait_chains(b)) return -1;
  (void)batch_chain_absorb(b, &b->late);
  (void)batch_chain_absorb(b, &b->ahead);
  for (odo_ctx* c : {b->ctx_a, b->ctx_b, b->ctx_c}) HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}

// The batched twin of odo_tracker_quiesce: chains posted ahead are run to completion, the early-started Solve of the next lock
// step is dropped, announcements are void, all streams idle. Afterwards nothing reads a caller-owned frame buffer.
extern "C" int odo_tracker_batch_quiesce(odo_tracker_batch* b) {
  if (!b) return fail("NULL batch tracker");
  HIP_OK(hipSetDevice(b->ctx_a->device));
  if (batch_quiesce(b)) return -1;
  if (b->ctx_a->lm_batch_job) b->ctx_a->lm_batch_job->active = 0;   // the batched Solve started early for the next lock step is dropped
  for (int i = 0; i < b->S; i++) {
    b->lm[i]->job.active = 0;
    b->hint_next[i] = b->hint_next_right[i] = b->prefetched[i] = nullptr;
  }
  return 0;
}

// Frame 0 of the slots in b->ids (ref: :95-145).
static int batch_init_set(odo_tracker_batch* b, const float* const* left_dev, const float* const* right_dev,
                          const float* abs_pose0) {
  HIP_OK(hipSetDevice(b->ctx_a->device));
  if (batch_quiesce(b)) return -1;
  const odo_tracker_params& p = b->p;
  const int n = (int)b->ids.size();
  if (n == 0) return 0;
  b->with_lists = batch_lists_ahead(n);
  for (int i : b->ids) {
    b->frame_id[i] = 0; b->alive[i] = 0; b->hint_next[i] = b->hint_next_right[i] = b->prefetched[i] = nullptr;
    b->res_left[i] = b->res_right[i] = nullptr;
  }
  hipStream_t sa = b->ctx_a->stream;
  std::vector<const float*> src(left_dev, left_dev + b->S);
  if (batch_build_pyramids(b, b->ids, src, b->kf_img, 0, sa)) return -1;                  // :130
  HIP_OK(hipEventRecord(b->ev_cur_img, sa));
  BatchChain* c = &b->late;
  batch_chain_fill(b, c, b->ids, left_dev, right_dev, true, false, 0, b->ev_cur_img);
  if (batch_chain_run(b, c)) return -1;                                                  // :102 (on this thread: nothing to overlap)
  if (batch_chain_absorb(b, c)) return -1;
  HIP_OK(hipStreamSynchronize(sa));
  const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int bad = -1;
  for (int i : b->ids) {
    const float* a0 = abs_pose0 ? abs_pose0 + 16 * (size_t)i : eye;
    memcpy(&b->kf_abs[16 * (size_t)i], a0, sizeof(float) * 16);        // :143
    memcpy(&b->pose_to_kf[16 * (size_t)i], a0, sizeof(float) * 16);    // :98
    odo_lm_reset(b->lm[i], eye, p.lm_lambda);                          // :77-81
    b->lm[i]->kf_img_ver = b->lm[i]->kf_dep_ver = 0;
    b->n_keyframes[i] = 1;
    b->alive[i] = b->res_ok[i];
    b->last_valid[i] = b->depth[i]->last.n_valid;
    b->last_depth_iters[i] = b->depth[i]->last.iters;
    if (!b->res_ok[i] && bad < 0) bad = i;
    b->out_par[i] = b->res_par[i];
    if (b->res_ok[i]) {
      const int par = b->res_par[i];
      std::swap(b->kf_dep[i], b->pre_dep[par][i]);                     // :141: the frame's depth pyramid is the keyframe's
      if (b->with_lists) (void)lm_adopt_candidate(b->lm[i], b->kf_img[i], b->kf_dep[i], 0, par);
    }
    b->res_left[i] = b->res_right[i] = nullptr;
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

// Optional: the left images of the NEXT step (NULL entries allowed). Their pyramids are built during the current step on a
// stream of their own instead of at the head of the next step.
extern "C" int odo_tracker_batch_hint_next(odo_tracker_batch* b, const float* const* next_left_dev) {
  if (!b) return fail("NULL batch tracker");
  for (int i = 0; i < b->S; i++) { b->hint_next[i] = next_left_dev ? next_left_dev[i] : nullptr; b->hint_next_right[i] = nullptr; }
  return 0;
}
// The same with the right images: the next step's ComputeDepth + depth pyramids + candidate lists (they depend on the images
// only) are enqueued a step early as well, so the depth stream works a step ahead of the pose LM.
extern "C" int odo_tracker_batch_hint_next_pair(odo_tracker_batch* b, const float* const* next_left_dev,
                                                const float* const* next_right_dev) {
  if (!b) return fail("NULL batch tracker");
  for (int i = 0; i < b->S; i++) {
    b->hint_next[i] = next_left_dev ? next_left_dev[i] : nullptr;
    b->hint_next_right[i] = (b->hint_next[i] && next_right_dev) ? next_right_dev[i] : nullptr;
  }
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
  if (b->dead) return fail("odo_tracker_batch_track: this tracker timed out with work in flight earlier and is dead: destroy it");
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
  b->with_lists = batch_lists_ahead(n);
  hipStream_t sa = b->ctx_a->stream;
  hipStream_t sb = b->overlap ? b->ctx_b->stream : sa;
  hipStream_t sc = b->ctx_c->stream;
  // ---- the chain posted a step ago for the frames announced then: normally complete long ago (it had the whole previous
  // step). Its results stay with their slots whether or not the announced pair is the one that came.
  if (batch_wait_chains(b)) return -1;
  (void)batch_chain_absorb(b, &b->late);
  if (batch_chain_absorb(b, &b->ahead)) return -1;
  if (n == 0) return 0;
  // ---- image pyramids of this step: prefetched during the last step (stream C), or built now (stream A)
  std::vector<int> build;
  bool any_prefetched = false, wrong_hint = false;
  for (int i : b->ids) {
    b->frame_id[i]++;
    if (b->prefetched[i] && b->prefetched[i] == left_dev[i]) { std::swap(b->cur_img[i], b->next_img[i]); any_prefetched = true; }
    else { build.push_back(i); if (b->prefetched[i]) wrong_hint = true; }
    b->prefetched[i] = nullptr;
  }
  if (any_prefetched) HIP_OK(hipStreamWaitEvent(sa, b->ev_next, 0));
  {
    std::vector<const float*> src(left_dev, left_dev + S);
    if (batch_build_pyramids(b, build, src, b->cur_img, 0, sa)) return -1;               // :205
  }
  HIP_OK(hipEventRecord(b->ev_cur_img, sa));
  // ---- stream B, this step's frames: slots whose collected job is not for this pair get one now (beside the Solve, :226)
  BatchChain* late = &b->late;
  {
    std::vector<int> need;
    for (int i : b->ids)
      if (!(b->res_left[i] == left_dev[i] && b->res_right[i] == right_dev[i] && b->res_tag[i] == b->frame_id[i])) need.push_back(i);
    batch_chain_fill(b, late, need, left_dev, right_dev, false, false, 0, b->ev_cur_img);
  }
  const bool late_now = b->overlap != 0;
  if (late_now && !late->ids.empty()) batch_chain_post(b, late);
  // ---- the next step: pyramids on stream C; announced pairs get their stream-B job now, behind this step's (the helper runs
  // the chains one after the other: a slot's depth object serves one job at a time)
  std::vector<int> nxt, nxt_pair;
  for (int i = 0; i < S; i++) {
    if (!b->hint_next[i] || !b->alive[i]) continue;
    nxt.push_back(i);
    if (b->hint_next_right[i] && b->depth_ahead) nxt_pair.push_back(i);
  }
  if (!nxt.empty()) {
    if (wrong_hint) {
      // A frame other than the announced one came: the Solve started early for the announced frame is abandoned, but its
      // launches on stream A may still be reading next_img — the rebuild on stream C goes behind them.
      HIP_OK(hipEventRecord(b->ev_a, sa));
      HIP_OK(hipStreamWaitEvent(sc, b->ev_a, 0));
    }
    if (batch_build_pyramids(b, nxt, b->hint_next, b->next_img, S, sc)) return -1;
    HIP_OK(hipEventRecord(b->ev_next, sc));
    for (int i : nxt) b->prefetched[i] = b->hint_next[i];
  }
  BatchChain* ahead = &b->ahead;
  batch_chain_fill(b, ahead, nxt_pair, b->hint_next.data(), b->hint_next_right.data(), false, true, 1, b->ev_next);
  if (!ahead->ids.empty()) batch_chain_post(b, ahead);
  for (int i = 0; i < S; i++) b->hint_next[i] = b->hint_next_right[i] = nullptr;
  // ---- pose LM of this step (stream A); the helper thread feeds stream B meanwhile
  std::vector<float> T((size_t)n * 16);
  std::vector<int> st(n, 0);
  std::vector<odo_lm*> lms(n);
  std::vector<const odo_pyr*> kfi(n), kfd(n), cur(n);
  for (int e = 0; e < n; e++) { const int i = b->ids[e]; lms[e] = b->lm[i]; kfi[e] = b->kf_img[i]; kfd[e] = b->kf_dep[i]; cur[e] = b->cur_img[i]; }
  const auto f1 = std::chrono::steady_clock::now();
  const int lm_rc = lm_solve_batch(n, lms.data(), kfi.data(), kfd.data(), cur.data(), T.data(), st.data(), nullptr, nullptr);  // :215
  const auto f2 = std::chrono::steady_clock::now();
  if (!late_now && !late->ids.empty()) batch_chain_post(b, late);   // program order: runs right here
  if (lm_rc < 0) return -1;
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
  // this step's depth results: the late chain is in front of the ahead chain in the helper's queue
  if (!late->ids.empty()) {
    if (b->worker.joinable()) {
      const auto q0 = std::chrono::steady_clock::now();
      long spins = 0;
      while (!late->complete.load(std::memory_order_acquire)) {
        if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - q0 > std::chrono::seconds(10)) {
          b->dead = 1;   // the helper still owns the chain: nothing of this tracker may be reused
          return fail("    depth failed! (the depth stream's helper thread did not finish within 10 s)");
        }
      }
    }
    if (batch_chain_absorb(b, late)) return -1;
  }
  const auto f3 = std::chrono::steady_clock::now();
  int any_depth_fail = 0;
  for (int e = 0; e < n; e++) {
    const int i = b->ids[e];
    const int par = b->res_par[i];
    b->out_par[i] = par;
    b->last_evals[i] = b->lm[i]->last_evals;
    b->last_valid[i] = b->depth[i]->last.n_valid;   // NB: of the slot's latest collected job (this frame's)
    b->last_depth_iters[i] = b->depth[i]->last.iters;
    if (!b->res_ok[i]) { status[i] = -1; b->alive[i] = 0; any_depth_fail = 1; continue; }  // :230-232
    const float* Ti = &T[16 * (size_t)e];
    float ang[3];
    motion_angles(Ti, ang);                                                                // :253
    const float mot[6] = {fabsf(ang[0]), fabsf(ang[1]), fabsf(ang[2]), fabsf(Ti[12]), fabsf(Ti[13]), fabsf(Ti[14])};
    float mag = 0.0f;
    for (int k = 0; k < 6; k++) mag += mot[k] * p.keyframe_weight[k];                       // :257
    if (mag > p.keyframe_motion_th) {                                                      // :258
      std::swap(b->kf_img[i], b->cur_img[i]);                                              // :259 (the :251 rebuild has the same content)
      std::swap(b->kf_dep[i], b->pre_dep[par][i]);
      memcpy(&b->kf_abs[16 * (size_t)i], &curp[16 * (size_t)e], sizeof(float) * 16);       // :260
      b->n_keyframes[i]++;
      if (is_new_keyframe) is_new_keyframe[i] = 1;
      if (b->with_lists) (void)lm_adopt_candidate(b->lm[i], b->kf_img[i], b->kf_dep[i], b->frame_id[i], par);
    }
    odo_lm_reset(b->lm[i], Ti, 0.01f);                                                     // :261 / :268
    if (motion_mag) motion_mag[i] = mag;
    status[i] = st[e] ? 1 : 0;
  }
  // ---- the next lock step's Solve: its inputs are all known now — the keyframes (promotions taken), the announced frames'
  // pyramids (stream C, ev_next), the initial poses (:261 / :268 Reset above). Its coarse launch and first step launches go out
  // here and run while the caller gets its results and comes back (lm_solve_batch collects the job if the next call brings
  // exactly the announced frames, else it is abandoned like any finished Solve's queued launches).
  if (b->early_solve && !nxt.empty()) {
    std::vector<odo_lm*> l2;
    std::vector<const odo_pyr*> ki, kd, cu;
    for (int i : nxt)
      if (b->alive[i]) { l2.push_back(b->lm[i]); ki.push_back(b->kf_img[i]); kd.push_back(b->kf_dep[i]); cu.push_back(b->next_img[i]); }
    if (!l2.empty()) {
      HIP_OK(hipStreamWaitEvent(sa, b->ev_next, 0));
      if (lm_batch_begin((int)l2.size(), l2.data(), ki.data(), kd.data(), cu.data()) < 0) return -1;
    }
  }
  if (any_depth_fail) fail("    depth failed!");
  b->tm_frame_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f0).count();
  b->tm_head_us += std::chrono::duration<double, std::micro>(f1 - f0).count();
  b->tm_solve_us += std::chrono::duration<double, std::micro>(f2 - f1).count();
  b->tm_depth_wait_us += std::chrono::duration<double, std::micro>(f3 - f2).count();
  b->tm_frames++;
  return 0;
}

// Host-clock averages per lock step since the last call (microseconds): whole call, head (tables + pyramid + chain fronts), the
// batched Solve (depth launches are pumped from its wait loops), the wait for this step's depth chain after the Solve.
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
  const int par = b->out_par[seq];
  if (val) *val = b->d_val[par][seq];
  if (disp) *disp = b->d_disp[par][seq];
  if (dep) *dep = b->d_dep[par][seq];
  return 0;
}
extern "C" int odo_tracker_batch_size(const odo_tracker_batch* b) { return b ? b->S : 0; }
extern "C" odo_lm* odo_tracker_batch_lm(odo_tracker_batch* b, int slot) { return (b && slot >= 0 && slot < b->S) ? b->lm[slot] : nullptr; }
extern "C" odo_ctx* odo_tracker_batch_ctx(odo_tracker_batch* b) { return b ? b->ctx_a : nullptr; }
