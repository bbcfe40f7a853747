// tracker.hip.h — the runner's frame loop (run_odometry_kitti_offline.cpp:58-145, 198-271) as a C-ABI object.
// Host logic only (keyframe policy, pose chaining); all image work is enqueued on three HIP streams (pose LM, depth, the
// next frame's image pyramid).
// Included by odometry_hip.hip after the pyramid / LM / depth objects are defined.
#pragma once

// One frame's stream-B work: ComputeDepth of the pair, the depth pyramid, the keyframe-candidate point lists. Two of them
// exist (slot 0 / 1, each with its own outputs) so that the depth stream can work on the NEXT frame while the pose LM is
// still on this one (odo_tracker_hint_next_pair).
struct TrackerJob {
  const float *left, *right;
  odo_pyr* img;           // the frame's image pyramid (source of the candidate lists)
  hipEvent_t img_ready;   // event after which `img` is complete (stream A: ev_cur_img, stream C: ev_next)
  int slot;
  long tag;               // frame id (tags the candidate lists)
  int with_lists;
  // progress (the thread that runs the job)
  DepthJob dj;
  int stage;              // 0 idle, 1 depth launches in flight, 3 all enqueued
  int err;
  // result
  int rc;
  char msg[256];
  DepthLmStats stats;
};

struct odo_tracker {
  odo_tracker_params p;
  odo_ctx* ctx_a;  // pyramids of the incoming frame + pose LM
  odo_ctx* ctx_b;  // ComputeDepth + the frame's keyframe-candidate pyramids
  odo_ctx* ctx_c;  // the NEXT frame's image pyramid when the next Solve starts early (a stream of its own: it delays neither chain)
  odo_lm* lm;
  odo_depth* depth;
  odo_pyr *kf_img, *kf_dep, *cur_img, *pre_img, *next_img;
  odo_pyr* pre_dep[2];         // per job slot
  const float* hint_next;      // device image the caller announced as the next frame (odo_tracker_hint_next)
  const float* hint_next_right;  // its right image (odo_tracker_hint_next_pair), NULL: left only
  const float* prefetched;     // image whose pyramid already sits in next_img
  uint8_t* d_val[2];           // per job slot
  float *d_disp[2], *d_dep[2];
  int out_slot;                // slot that holds the last tracked frame's outputs
  float kf_abs[16];
  float pose_to_kf[16];
  int n_keyframes, frame_id;
  int last_evals, last_depth_iters, last_valid;
  hipEvent_t ev_cur_img;  // stream A: the current frame's image pyramid is complete (stream B reads it for the candidate lists)
  TrackerJob jobs[2];
  TrackerJob* cur_job;    // job the calling thread is pumping (overlap_depth 0 / 1)
  double tm_solve_us, tm_depth_us, tm_frame_us, tm_wait_us; long tm_frames;  // host-clock averages (diagnostics)
  int cand_lists;    // 1: keyframe-candidate point lists are built every frame on stream B (ODO_NO_CAND_LISTS=1 turns it off)
  // overlap_depth == 2: a helper host thread feeds stream B (ComputeDepth + candidate pyramids) while the calling
  // thread feeds stream A (pose LM). Host launch rate, not the GPU, bounds a latency-bound frame loop.
  // Jobs go through a two-entry ring: job n lives in jobs[n & 1]; the caller posts (w_posted++), the helper runs them in order
  // (w_done++). At most two are outstanding: this frame's and, when the next pair was announced, the next frame's.
  std::thread worker;
  std::atomic<long> w_posted, w_done;
  std::atomic<int> w_quit;   // 1: the worker leaves its loop once the ring is empty
  long ahead_job;            // index of the job already posted for the NEXT frame, -1: none
  // Early start of the next Solve (overlap_depth != 0, next frame announced with odo_tracker_hint_next): the next frame's
  // pyramid is built on a third stream at the start of the call, and as soon as this frame's Solve has returned — initial pose and
  // keyframe decision are known then — the next Solve's launches go out on the LM stream while the depth stream finishes and
  // the host does its bookkeeping (odo_lm_solve_begin). Same launches, earlier; ODO_NO_EARLY_SOLVE=1 turns it off.
  int early_solve;
  int chain_solve;           // 1: ... and queued BEHIND this frame's Solve before its result exists (lm_chain_begin; opt-in: ODO_CHAIN_SOLVE=1 — measured no faster)
  long chain_used, chain_wasted;   // chained Solves adopted / that ran for nothing (the host's keyframe test disagreed with the guard)
  double dbg_pre_us, dbg_spin_us, dbg_chain_us, dbg_verdict_us, dbg_post_us, dbg_relaunch_us; long dbg_n, dbg_relaunch_n;   // ODO_TRACK_DEBUG: host time per call, by phase
  int depth_ahead;           // 1: with the next PAIR announced, the next frame's stream-B job is posted a frame early (ODO_NO_DEPTH_AHEAD=1: off)
  hipEvent_t ev_next;        // stream C: next_img is complete
  int arm_enabled;           // armed Solves are in use (lm_enable_arming; ODO_NO_ARM=1 turns them off)
  int arm_pending, arm_on;   // the next Solve is to be armed from the wait loop (tracker_poll_next) / has been armed and waits for its word
  int next_ready;            // 1: ev_next was seen complete by the host while it waited for the Solve (tracker_poll_next): the next Solve's
                             // launches then go out without a wait packet in front of them (1.5 us of host time + the packet's processing)
};

static void tracker_worker_main(odo_tracker* t);

extern "C" int odo_tracker_default_params(odo_tracker_params* p) {
  if (!p) return fail("NULL params");
  memset(p, 0, sizeof(*p));
  p->rows = 376; p->cols = 1241; p->levels = 4;
  p->lm_lambda = 0.01f; p->lm_precision = 0.995f;
  const int mi[4] = {10, 20, 30, 30};
  for (int i = 0; i < 4; i++) p->lm_max_iters[i] = mi[i];
  p->lm_robust = 1; p->lm_huber_delta = 28.0f;
  p->grad_th = 8.0f; p->ssd_th = 900.0f; p->photo_th = 15.0f;
  p->min_depth = 0.1f; p->max_depth = 30.0f;
  p->depth_lambda = 0.01f; p->depth_huber_delta = 28.0f; p->depth_precision = 0.995f;
  p->depth_max_iters = 50; p->boundary = 4; p->max_residuals = 80000;
  p->max_disparity = 0; p->any_size = 0;
  p->K = kKitti00;
  p->baseline = 386.1448f / 718.856f;
  const float w[6] = {0.1f / 3.3f, 1.0f / 3.3f, 0.1f / 3.3f, 1.0f / 3.3f, 0.1f / 3.3f, 1.0f / 3.3f};
  for (int i = 0; i < 6; i++) p->keyframe_weight[i] = w[i];
  p->keyframe_motion_th = 1.1f;
  p->smooth_image = 1;
  p->overlap_depth = 2;
  return 0;
}

extern "C" int odo_tracker_destroy(odo_tracker* t) {
  if (!t) return 0;
  if (t->worker.joinable()) {
    t->w_quit.store(1, std::memory_order_release);  // a job in flight finishes first; the flag is the worker's to read only
    t->worker.join();
  }
  if (t->ctx_a) (void)hipStreamSynchronize(t->ctx_a->stream);
  if (t->ctx_b) (void)hipStreamSynchronize(t->ctx_b->stream);
  if (t->ctx_c) (void)hipStreamSynchronize(t->ctx_c->stream);
  if (t->lm && getenv("ODO_TRACK_DEBUG")) fprintf(stderr, "[track] armed Solves: %ld started on the host's word, %ld told to return\n", t->lm->arm_used, t->lm->arm_aborted);
  odo_lm_destroy(t->lm);
  odo_depth_destroy(t->depth);
  odo_pyr* ps[] = {t->kf_img, t->kf_dep, t->cur_img, t->pre_img, t->pre_dep[0], t->pre_dep[1], t->next_img};
  for (odo_pyr* q : ps) odo_pyramid_destroy(q);
  for (int k = 0; k < 2; k++) {
    if (t->d_val[k]) (void)hipFree(t->d_val[k]);
    if (t->d_disp[k]) (void)hipFree(t->d_disp[k]);
    if (t->d_dep[k]) (void)hipFree(t->d_dep[k]);
  }
  if (t->ev_cur_img) (void)hipEventDestroy(t->ev_cur_img);
  if (t->ev_next) (void)hipEventDestroy(t->ev_next);
  odo_ctx_destroy(t->ctx_c);
  odo_ctx_destroy(t->ctx_b);
  odo_ctx_destroy(t->ctx_a);
  delete t;
  return 0;
}

extern "C" int odo_tracker_create(int device, const odo_tracker_params* p, odo_tracker** out) {
  if (!p || !out) return fail("odo_tracker_create: NULL arg");
  *out = nullptr;
  odo_tracker* t = new (std::nothrow) odo_tracker();
  if (!t) return fail("out of memory");
  t->ctx_a = t->ctx_b = t->ctx_c = nullptr; t->lm = nullptr; t->depth = nullptr;
  t->kf_img = t->kf_dep = t->cur_img = t->pre_img = t->next_img = nullptr;
  t->hint_next = t->hint_next_right = t->prefetched = nullptr;
  for (int k = 0; k < 2; k++) {
    t->pre_dep[k] = nullptr; t->d_val[k] = nullptr; t->d_disp[k] = t->d_dep[k] = nullptr;
    memset(&t->jobs[k], 0, sizeof(TrackerJob));
    t->jobs[k].slot = k;
  }
  t->cur_job = nullptr; t->out_slot = 0; t->ahead_job = -1;
  t->ev_cur_img = nullptr;
  t->n_keyframes = t->frame_id = t->last_evals = t->last_depth_iters = t->last_valid = 0;
  t->cand_lists = getenv("ODO_NO_CAND_LISTS") ? 0 : 1;
  t->tm_solve_us = t->tm_depth_us = t->tm_frame_us = t->tm_wait_us = 0.0; t->tm_frames = 0;
  t->w_posted.store(0); t->w_done.store(0); t->w_quit.store(0);
  t->ev_next = nullptr; t->next_ready = 0; t->arm_pending = t->arm_on = t->arm_enabled = 0;
  t->early_solve = getenv("ODO_NO_EARLY_SOLVE") ? 0 : 1;
  // Chained Solves are OFF unless ODO_CHAIN_SOLVE=1: measured (round 4) they close the 14 us the GPU idles between
  // two Solves (rocprofv3: fine -> next coarse gap 14.2 -> 0.0 us) and the frame rate does not move (3 250 both ways): the LM
  // kernels' own wall time grows by what the gap gave (same cycle counts: the clock, not the work).
  t->chain_solve = (t->early_solve && getenv("ODO_CHAIN_SOLVE") && !getenv("ODO_NO_CHAIN_SOLVE")) ? 1 : 0;
  t->chain_used = t->chain_wasted = 0;
  t->dbg_pre_us = t->dbg_spin_us = t->dbg_chain_us = t->dbg_verdict_us = t->dbg_post_us = t->dbg_relaunch_us = 0.0; t->dbg_n = t->dbg_relaunch_n = 0;
  t->depth_ahead = getenv("ODO_NO_DEPTH_AHEAD") ? 0 : 1;
  t->p = *p;
  float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(t->pose_to_kf, eye, sizeof(eye));
  memcpy(t->kf_abs, eye, sizeof(eye));
  const size_t n = (size_t)p->rows * p->cols;
  // Stream A carries the latency-critical LM chain: a high-priority stream gets its hardware queue from a pool of its
  // own, so no other stream of the process (the depth stream, a collective's stream, another tracker) can sit ahead of a
  // dependent LM launch in the same queue. ODO_LM_PRIORITY=0 falls back to a normal stream.
  // (read per tracker: several trackers in ONE process should not all draw from the small high-priority pool)
  const bool lm_prio = !(getenv("ODO_LM_PRIORITY") && atoi(getenv("ODO_LM_PRIORITY")) == 0);
  bool ok = (lm_prio ? odo_ctx_create_high_priority(device, &t->ctx_a) : odo_ctx_create(device, &t->ctx_a)) == 0 &&
            odo_ctx_create(device, &t->ctx_b) == 0 && odo_ctx_create(device, &t->ctx_c) == 0;
  ok = ok && odo_lm_create(t->ctx_a, p->lm_lambda, p->lm_precision, p->lm_max_iters, p->levels, eye, p->lm_robust,
                           p->lm_huber_delta, &p->K, &t->lm) == 0;
  ok = ok && odo_depth_create(t->ctx_b, p->grad_th, p->ssd_th, p->photo_th, p->min_depth, p->max_depth, p->depth_lambda,
                              p->depth_huber_delta, p->depth_precision, p->depth_max_iters, p->boundary, &p->K, p->baseline,
                              p->max_residuals, p->max_disparity, p->any_size, &t->depth) == 0;
  ok = ok && pyr_alloc(t->ctx_a, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &t->cur_img, false) == 0;
  ok = ok && pyr_alloc(t->ctx_a, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &t->next_img, false) == 0;
  ok = ok && pyr_alloc(t->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &t->kf_img, false) == 0;
  ok = ok && pyr_alloc(t->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_DEPTH, &t->kf_dep, false) == 0;
  ok = ok && pyr_alloc(t->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_IMAGE, &t->pre_img, false) == 0;
  for (int k = 0; k < 2; k++) {
    ok = ok && pyr_alloc(t->ctx_b, p->rows, p->cols, p->levels, ODO_PYR_DEPTH, &t->pre_dep[k], false) == 0;
    ok = ok && hipMalloc((void**)&t->d_val[k], n) == hipSuccess && hipMalloc((void**)&t->d_disp[k], sizeof(float) * n) == hipSuccess &&
         hipMalloc((void**)&t->d_dep[k], sizeof(float) * n) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&t->ev_cur_img, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&t->ev_next, hipEventDisableTiming) == hipSuccess;
  if (ok && !getenv("ODO_LM_TRACE")) t->lm->record = 0;   // nobody reads the per-evaluation rows / cost statistics of a tracker's Solves
  // Armed Solves (lm_arm_begin): with the helper thread only (overlap_depth == 2: the calling thread does nothing but the pose LM's
  // launches); ODO_NO_ARM=1 turns them off.
  t->arm_enabled = 0;
  if (ok && p->overlap_depth == 2 && !getenv("ODO_NO_ARM") && !t->chain_solve) { ok = lm_enable_arming(t->lm) == 0; t->arm_enabled = ok ? 1 : 0; }
  // (only a tracker that chains its Solves needs the finishing launch to leave pose + guard word on the device: sixteen stores and two
  //  releases between the last evaluation and the launch's exit, which an armed launch behind it waits for)
  if (ok && t->chain_solve) lm_set_chain_rule(t->lm, p->keyframe_weight, p->keyframe_motion_th);
  if (!ok) {
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", g_err);
    odo_tracker_destroy(t);
    return fail("odo_tracker_create failed: %s", keep);
  }
  if (p->overlap_depth == 2) t->worker = std::thread(tracker_worker_main, t);
  *out = t;
  return 0;
}

// ComputeDepth of the pair and the frame's keyframe-candidate pyramids, all on stream B (ref: :226-252), as a
// resumable job: begin enqueues the front, pump issues one more launch when the device is ready for it.
static int tracker_job_begin(odo_tracker* t, TrackerJob* j) {
  const odo_tracker_params& p = t->p;
  j->err = 0;
  if (depth_check_size(t->depth, p.rows, p.cols)) return -1;
  if (depth_ensure(t->depth, p.rows, p.cols)) return -1;
  if (depth_job_begin(t->depth, &j->dj, j->left, j->right, p.rows, p.cols, t->d_val[j->slot], t->d_disp[j->slot], t->d_dep[j->slot], 2))
    return -1;
  j->stage = 1;
  return 0;
}
static void tracker_job_pump_one(odo_tracker* t, TrackerJob* j) {
  if (j->stage != 1) return;
  const int r = depth_job_pump(t->depth, &j->dj);
  if (r < 0) { j->err = 1; j->stage = 3; return; }
  if (r > 0) {
    const odo_tracker_params& p = t->p;
    if (!j->img && pyr_build(t->pre_img, j->left, p.smooth_image)) j->err = 1;  // :130 (frame 0 only: see odo_tracker_track for :251)
    if (pyr_build(t->pre_dep[j->slot], t->d_dep[j->slot], 0)) j->err = 1;                                  // :252
    // This frame may become the next keyframe (:258-265): compact its valid-depth pixels into point lists now, on this
    // stream, beside the Solve — the image pyramid is the one built for the frame's Solve (same image, same arithmetic
    // as the :251 rebuild). If the frame is not promoted the lists are simply overwritten two frames later.
    if (j->with_lists) {
      if (hipStreamWaitEvent(t->ctx_b->stream, j->img_ready, 0) != hipSuccess ||
          lm_build_candidate(t->lm, j->img, t->pre_dep[j->slot], t->ctx_b->stream, j->tag, j->slot))
        t->lm->cand[j->slot].tag = -1;
    } else {
      t->lm->cand[j->slot].tag = -1;
    }
    if (depth_job_stats(t->depth, &j->dj)) j->err = 1;  // completion word AFTER the pyramids: it covers them too
    j->stage = 3;
  }
}
// Called from the Solve's wait loop (overlap_depth == 2, next frame announced): has stream C finished the next frame's pyramid?
static void tracker_poll_next(void* arg) {
  odo_tracker* t = (odo_tracker*)arg;
  if (!t->next_ready) {
    if (hipEventQuery(t->ev_next) == hipSuccess) t->next_ready = 1;
    else (void)hipGetLastError();   // (hipErrorNotReady is not an error, and must not be the thread's last error when the next HIP_OK looks)
    return;
  }
  // The next frame's pyramid is complete: its Solve can be ARMED (lm_arm_begin) — the coarse launch queued behind this frame's Solve, no
  // wait packet in front of it.
  if (t->arm_pending) {
    t->arm_pending = 0;
    t->arm_on = lm_arm_begin(t->lm, t->kf_img, t->kf_dep, t->next_img, 0.01f) == 0;
  }
}
static void tracker_job_pump(void* arg) {
  odo_tracker* t = (odo_tracker*)arg;
  if (t->cur_job) tracker_job_pump_one(t, t->cur_job);
}
// The NEXT frame's image pyramid (ref: :205 of the next iteration) on a stream of its own, enqueued at the start of this
// frame's call: complete long before this frame's Solve ends, so the next Solve can start the moment it does.
static int tracker_next_pyramid(odo_tracker* t, const float* next_left) {
  t->next_img->ctx = t->ctx_c;
  if (pyr_build(t->next_img, next_left, t->p.smooth_image)) return -1;
  HIP_OK(hipEventRecord(t->ev_next, t->ctx_c->stream));
  return 0;
}
static int tracker_job_drain(odo_tracker* t, TrackerJob* j) {
  while (j->stage == 1) tracker_job_pump_one(t, j);
  j->stage = 0;
  return j->err ? -1 : 0;
}
// Whole job on the calling thread: launches, then the wait for its completion word (or a full sync) and its statistics.
static int tracker_job_run(odo_tracker* t, TrackerJob* j, bool full_sync) {
  int rc = tracker_job_begin(t, j);
  if (rc == 0) rc = tracker_job_drain(t, j);
  if (rc == 0) rc = depth_finish(t->depth, full_sync);
  if (rc == 2) {   // the persistent depth-LM launch gave up: the whole job again (depth_finish has switched this run to the step launches)
    rc = tracker_job_begin(t, j);
    if (rc == 0) rc = tracker_job_drain(t, j);
    if (rc == 0) rc = depth_finish(t->depth, full_sync);
    if (rc == 2) rc = -1;
  }
  j->stats = t->depth->last;
  j->rc = rc;
  if (rc) snprintf(j->msg, sizeof(j->msg), "%s", g_err);
  return rc;
}
static void tracker_job_fill(odo_tracker* t, TrackerJob* j, const float* left, const float* right, odo_pyr* img, hipEvent_t img_ready,
                             long tag, int with_lists) {
  j->left = left; j->right = right; j->img = img; j->img_ready = img_ready; j->tag = tag; j->with_lists = with_lists;
  j->stage = 0; j->err = 0; j->rc = 0; j->msg[0] = 0;
}

// Helper thread (overlap_depth == 2): runs the stream-B jobs of the ring in order, each up to its completion word.
static void tracker_worker_main(odo_tracker* t) {
  (void)hipSetDevice(t->ctx_b->device);
  int idle_spins = 0;
  for (;;) {
    const long done = t->w_done.load(std::memory_order_relaxed);
    if (t->w_posted.load(std::memory_order_acquire) <= done) {
      if (t->w_quit.load(std::memory_order_acquire)) return;
      // Spin while a sequence is being tracked (the next job arrives within a fraction of a millisecond and a
      // sleeping thread wakes ~100 us late); back off only after ~10 ms without work.
      if (++idle_spins > 20000000) std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    // A job that was already waiting when the previous one finished starts in the MIDDLE of a frame (a frame that was not announced
    // brings two jobs at once; so does a depth stream that has fallen behind): its persistent depth launch would then be dispatched
    // while the pose LM's persistent launch holds its XCD, and the two can end up waiting for each other's undispatched workgroups
    // (see g_lm_fine_dispatch) — the depth launch gives up after its whole wait bound and the job runs twice, 0.9 ms in all. Such a
    // job takes the launch-per-iteration path from the start (0.06 ms slower, hidden); in step with the frames every job starts
    // with its frame, under the pose LM's coarse launch, and keeps the persistent launch.
    static const bool persist_always = getenv("ODO_DEPTH_PERSIST_ALWAYS") != nullptr;
    if (idle_spins == 0 && done > 0 && !persist_always) t->depth->persist_off_once = 1;
    idle_spins = 0;
    const auto w0 = std::chrono::steady_clock::now();
#ifdef ODO_DIAG   // diagnostic build only (tools/interference_probe.py): the stream-B job left out, to price what it costs the pose LM beside it
    static const bool diag_skip = getenv("ODO_DIAG_SKIP_DEPTH") != nullptr;
    if (diag_skip && done >= 2) { t->jobs[done & 1].rc = 0; t->w_done.store(done + 1, std::memory_order_release); continue; }
#endif
    (void)tracker_job_run(t, &t->jobs[done & 1], false);
    t->tm_depth_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    t->w_done.store(done + 1, std::memory_order_release);
  }
}

// Waits until job `index` of the ring has been run. Bounded: a job is ~0.2 ms of launches plus the depth LM; after 5 s the
// worker is presumed stuck behind a lost device (the tracker stays destroyable: odo_tracker_destroy joins the worker whenever
// it comes back).
static int tracker_wait_job(odo_tracker* t, long index) {
  const auto q0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (t->w_done.load(std::memory_order_acquire) <= index) {
    if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - q0 > std::chrono::seconds(5))
      return fail("    depth failed! (the depth stream's helper thread did not finish within 5 s)");
  }
  t->tm_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - q0).count();
  return 0;
}
static int tracker_wait_idle(odo_tracker* t) {
  const long posted = t->w_posted.load(std::memory_order_acquire);
  return posted > 0 ? tracker_wait_job(t, posted - 1) : 0;
}

// Everything the tracker still has in flight on behalf of frames the caller handed it — the job posted ahead for an announced
// pair, the prefetched pyramid of an announced image, an early-started Solve, straggling LM launches — is run to completion or
// dropped, and all three streams are idle afterwards. After this call no launch, queued or yet to be issued by the helper thread,
// reads a caller-owned frame buffer: the caller may free or overwrite its frames (announcements are void).
extern "C" int odo_tracker_quiesce(odo_tracker* t) {
  if (!t) return fail("NULL tracker");
  HIP_OK(hipSetDevice(t->ctx_a->device));
  if (tracker_wait_idle(t)) return -1;
  t->ahead_job = -1;
  lm_arm_abort(t->lm);   // (an armed launch would hold stream A until its word came)
  HIP_OK(hipStreamSynchronize(t->ctx_a->stream));
  HIP_OK(hipStreamSynchronize(t->ctx_b->stream));
  HIP_OK(hipStreamSynchronize(t->ctx_c->stream));
  t->lm->job.active = 0;                   // a Solve started early for an announced frame is dropped
  t->lm->chained.active = 0;
  t->prefetched = t->hint_next = t->hint_next_right = nullptr;  // announcements are void
  return 0;
}

extern "C" int odo_tracker_init(odo_tracker* t, const float* left, const float* right, const float abs_pose0[16]) {
  if (!t || !left || !right || !abs_pose0) return fail("odo_tracker_init: NULL arg");
  HIP_OK(hipSetDevice(t->ctx_a->device));
  // Re-initialisation of a tracker that has been tracking: a job posted ahead for the previous sequence's next frame, the
  // tail of the last frame (a prefetched next pyramid, straggling LM launches, an early Solve) may still be running.
  // Start from an idle helper thread and a quiet device and build everything on stream B again.
  if (odo_tracker_quiesce(t)) return -1;
  t->pre_img->ctx = t->ctx_b;
  t->lm->cand[0].tag = t->lm->cand[1].tag = -1;
  TrackerJob* j = &t->jobs[0];
  // frame 0 has no Solve pyramid: the first Solve builds the keyframe lists itself (with_lists = 0)
  tracker_job_fill(t, j, left, right, nullptr, nullptr, 0, 0);
  if (tracker_job_run(t, j, true)) {                            // :102, :130-131
    if (t->depth->last.status != 0) fail("Init 0-th frame failed!");   // :103-106
    return -1;
  }
  std::swap(t->kf_img, t->pre_img);                             // :141 first keyframe
  std::swap(t->kf_dep, t->pre_dep[0]);
  t->out_slot = 0;
  memcpy(t->kf_abs, abs_pose0, sizeof(float) * 16);             // :143
  memcpy(t->pose_to_kf, abs_pose0, sizeof(float) * 16);         // :98 pose_to_keyframe = cur_pose
  float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  odo_lm_reset(t->lm, eye, t->p.lm_lambda);                     // :77-81 init_relative_affine = I
  t->n_keyframes = 1;
  t->frame_id = 0;
  t->last_valid = j->stats.n_valid;
  t->last_depth_iters = j->stats.iters;
  return 0;
}

// Inverse of a 4x4 (Eigen Matrix4f::inverse(), ref: run_odometry_kitti_offline.cpp:218): Gauss-Jordan in fp64,
// rounded to fp32. Column-major in/out.
static bool invert4(const float* m, float* out) {
  double a[4][8];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { a[i][j] = m[j * 4 + i]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int i = c + 1; i < 4; i++) if (fabs(a[i][c]) > fabs(a[piv][c])) piv = i;
    if (a[piv][c] == 0.0) return false;
    if (piv != c) for (int j = 0; j < 8; j++) std::swap(a[c][j], a[piv][j]);
    const double d = a[c][c];
    for (int j = 0; j < 8; j++) a[c][j] /= d;
    for (int i = 0; i < 4; i++) if (i != c) { const double f = a[i][c]; for (int j = 0; j < 8; j++) a[i][j] -= f * a[c][j]; }
  }
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[j * 4 + i] = (float)a[i][4 + j];
  return true;
}
static void matmul4(const float* A, const float* B, float* C) {  // column-major fp32, k ascending
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      C[j * 4 + i] = ((A[0 * 4 + i] * B[j * 4 + 0] + A[1 * 4 + i] * B[j * 4 + 1]) + A[2 * 4 + i] * B[j * 4 + 2]) + A[3 * 4 + i] * B[j * 4 + 3];
}

// Sophus SO3::angleX/Y/Z (ref: third_party/Sophus/sophus/so3.hpp:127-154): SO3(R) -> q -> R', then the rotation
// closest to a 2x2 block of R' (makeRotationMatrix, polar factor) and its SO2 log = atan2(c - b, a + d).
static void motion_angles(const float* T, float ang[3]) {
  Se3 s;
  se3_from_colmajor(T, &s);
  float R[9];
  quat_to_rot(s, R);
  ang[0] = atan2f(R[7] - R[5], R[4] + R[8]);
  ang[1] = atan2f(R[2] - R[6], R[0] + R[8]);
  ang[2] = atan2f(R[3] - R[1], R[0] + R[4]);
}

extern "C" int odo_tracker_track(odo_tracker* t, const float* left, const float* right, float pose_to_keyframe[16],
                                 float abs_pose[16], int* is_new_keyframe, float* motion_mag, int* solve_status) {
  if (!t || !left || !right) return fail("odo_tracker_track: NULL arg");
  const odo_tracker_params& p = t->p;
  const auto f0 = std::chrono::steady_clock::now();
  HIP_OK(hipSetDevice(t->ctx_a->device));
  t->frame_id++;
  // :251 rebuilds the pyramid :205 built from the same image: bit-identical, so a promoted frame's :205 pyramid becomes the
  // keyframe pyramid (buffer swap below) and no second build is launched.
  const bool had_prefetch = t->prefetched == left;
  if (had_prefetch) {
    std::swap(t->cur_img, t->next_img);  // :205 — this frame's pyramid was built during the last call
  } else {
    t->lm->job.active = 0;               // an early Solve (if any) was started on another image
    t->cur_img->ctx = t->ctx_a;
    if (pyr_build(t->cur_img, left, p.smooth_image)) return -1;                        // :205
    if (t->cand_lists) HIP_OK(hipEventRecord(t->ev_cur_img, t->ctx_a->stream));  // before the stream-B job can ask for it
  }
  t->prefetched = nullptr;
  const float* next_left = t->hint_next;
  const float* next_right = t->hint_next_right;
  t->hint_next = t->hint_next_right = nullptr;
  const bool early = t->early_solve && p.overlap_depth != 0 && next_left != nullptr;
  const bool ahead = early && t->depth_ahead && p.overlap_depth == 2 && next_right != nullptr;
  // ---- this frame's stream-B job: posted a frame ago (the pair was announced), or now
  long my_job = -1;          // ring index (overlap_depth == 2)
  TrackerJob* jk = nullptr;
  if (p.overlap_depth == 2) {
    if (t->ahead_job >= 0) {
      TrackerJob* ja = &t->jobs[t->ahead_job & 1];
      if (had_prefetch && ja->left == left && ja->right == right) {
        my_job = t->ahead_job;
        jk = ja;
      } else if (tracker_wait_idle(t)) {   // announced one pair, given another: let the stale job finish, its outputs are dropped
        return -1;
      }
      t->ahead_job = -1;
    }
    if (my_job < 0) {
      // nothing may be left in the ring (a call that returned early after posting its job leaves one there): at most two jobs —
      // this frame's and the next frame's — are ever outstanding
      if (tracker_wait_idle(t)) return -1;
      my_job = t->w_posted.load(std::memory_order_relaxed);
      jk = &t->jobs[my_job & 1];
      tracker_job_fill(t, jk, left, right, t->cur_img, had_prefetch ? t->ev_next : t->ev_cur_img, (long)t->frame_id, t->cand_lists);
      t->w_posted.store(my_job + 1, std::memory_order_release);
    }
  } else {
    jk = &t->jobs[0];
    tracker_job_fill(t, jk, left, right, t->cur_img, had_prefetch ? t->ev_next : t->ev_cur_img, (long)t->frame_id, t->cand_lists);
  }
  if (p.overlap_depth == 1) {
    // stream B: ComputeDepth + candidate pyramids, concurrent with the Solve on stream A. The front of the job is
    // enqueued now; its depth-LM launches are issued from the pose LM's wait loop (one host thread feeds both).
    if (tracker_job_begin(t, jk)) return -1;
    t->cur_job = jk;
    t->lm->idle_pump = tracker_job_pump;
    t->lm->idle_arg = t;
  }
  // ---- the next frame: its pyramid on stream C right away; with the pair announced its stream-B job goes into the ring too
  // (the helper runs it as soon as it is done with this frame's: the depth stream works a frame ahead of the pose LM)
  t->next_ready = 0;
  if (early && tracker_next_pyramid(t, next_left)) return -1;
  if (early && p.overlap_depth == 2 && !t->chain_solve) { t->lm->idle_pump = tracker_poll_next; t->lm->idle_arg = t; }
  // ---- the next frame's Solve ARMED: its coarse launch is queued behind this frame's Solve from the wait loop, as soon as the next
  // pyramid is complete, and reads this frame's pose and the keyframe decision from a word the host writes (lm_arm_go / lm_arm_abort
  // below). This frame's Solve must be in flight (started early in the last call); otherwise the next Solve starts the ordinary way.
  struct ArmGuard {   // whatever path leaves this call: an armed launch never stays without its word
    odo_tracker* t;
    ~ArmGuard() { t->arm_pending = 0; if (t->arm_on) { t->arm_on = 0; lm_arm_abort(t->lm); } }
  } arm_guard{t};
  t->arm_pending = 0; t->arm_on = 0;
  if (early && t->arm_enabled && had_prefetch && p.overlap_depth == 2 && !t->chain_solve && lm_job_matches(t->lm, t->kf_img, t->kf_dep, t->cur_img) &&
      t->lm->job.launches == 2)
    t->arm_pending = 1;
  if (ahead) {
    const long n = t->w_posted.load(std::memory_order_relaxed);
    TrackerJob* ja = &t->jobs[n & 1];   // the other slot: this frame's job is n - 1 (or done long ago)
    tracker_job_fill(t, ja, next_left, next_right, t->next_img, t->ev_next, (long)t->frame_id + 1, t->cand_lists);
    t->w_posted.store(n + 1, std::memory_order_release);
    t->ahead_job = n;
  }
  // ---- the next frame's Solve, queued behind this frame's before its result exists: same keyframe, initial pose = this Solve's
  // result taken on the device, guarded by the runner's keyframe test on the device (lm_chain_begin). This frame's Solve must be in
  // flight for that: started early in the last call (or chained then), else started here.
  bool chained = false;
  const auto d0 = std::chrono::steady_clock::now();
  auto d1 = d0, d2 = d0;
  if (early && t->chain_solve) {
    if (!lm_job_matches(t->lm, t->kf_img, t->kf_dep, t->cur_img)) {
      if (had_prefetch && hipStreamWaitEvent(t->ctx_a->stream, t->ev_next, 0) != hipSuccess) return fail("hipStreamWaitEvent failed");
      if (odo_lm_solve_begin(t->lm, t->kf_img, t->kf_dep, t->cur_img) < 0) return -1;
    }
    // (ev_next is re-recorded by tracker_next_pyramid above: the next frame's pyramid on stream C — ~15 us of work enqueued a moment
    //  ago, while this frame's Solve has a few hundred us to go: wait for it HERE, on the host, so that no wait packet sits between
    //  this Solve's last launch and the chained Solve's first; after 100 us the stream waits for it instead)
    bool ready = false;
    d1 = std::chrono::steady_clock::now();
    for (const auto w0 = std::chrono::steady_clock::now(); !ready;) {
      ready = hipEventQuery(t->ev_next) == hipSuccess;
      if (!ready && std::chrono::steady_clock::now() - w0 > std::chrono::microseconds(100)) break;
    }
    (void)hipGetLastError();   // (hipErrorNotReady is not an error)
    d2 = std::chrono::steady_clock::now();
    if (ready || hipStreamWaitEvent(t->ctx_a->stream, t->ev_next, 0) == hipSuccess)
      chained = lm_chain_begin(t->lm, t->kf_img, t->kf_dep, t->next_img, 0.01f) == 0;
  }
  const auto d3 = std::chrono::steady_clock::now();
  t->dbg_pre_us += std::chrono::duration<double, std::micro>(d0 - f0).count();
  t->dbg_spin_us += std::chrono::duration<double, std::micro>(d2 - d1).count();
  t->dbg_chain_us += std::chrono::duration<double, std::micro>(d3 - d2).count();
  float T[16];
  const auto s0 = std::chrono::steady_clock::now();
  const int st = odo_lm_solve(t->lm, t->kf_img, t->kf_dep, t->cur_img, T);             // :215 (collects an early start)
  const int solved_token = t->lm->last_token, solved_slot = t->lm->last_slot;
  lap(0);
  const auto s1 = std::chrono::steady_clock::now();
  t->tm_solve_us += std::chrono::duration<double, std::micro>(s1 - s0).count();
  t->lm->idle_pump = nullptr;
  if (next_left && !early) {
    // The caller told us which device image comes next (offline / batched runs know): its pyramid is built now, on
    // the idle stream A, instead of at the head of the next call where the Solve would wait for it.
    t->next_img->ctx = t->ctx_a;
    if (pyr_build(t->next_img, next_left, p.smooth_image)) return -1;
    HIP_OK(hipEventRecord(t->ev_next, t->ctx_a->stream));
  }
  if (next_left) t->prefetched = next_left;
  // :218 — the runner stores the frame's pose BEFORE it computes the depth (:215-232): a frame whose ComputeDepth fails
  // still reports its pose (and the runner then leaves its loop).
  memcpy(t->pose_to_kf, T, sizeof(T));
  float inv[16], cur[16];
  // (called once the next Solve's launches are out — the keyframe test below needs T only, and every host instruction in front of those
  //  launches is idle time of the pose-LM chain)
  auto write_pose = [&]() {
    if (!invert4(T, inv))   // singular pose_to_keyframe (the failure pseudo-identity has (3,3) = 0): Eigen's inverse() of a
      for (int i = 0; i < 16; i++) inv[i] = __builtin_nanf("");  // singular matrix is inf / NaN, not zeros
    matmul4(t->kf_abs, inv, cur);                                                        // :218
    if (pose_to_keyframe) memcpy(pose_to_keyframe, T, sizeof(T));
    if (abs_pose) memcpy(abs_pose, cur, sizeof(cur));
    if (solve_status) *solve_status = st;
    if (is_new_keyframe) *is_new_keyframe = 0;
    if (motion_mag) *motion_mag = 0.0f;
  };
  lap(1);
  const float mag = motion_magnitude(T, p.keyframe_weight);                            // :253-257 (odo_math.h: the guard's own function)
  const bool promote = mag > p.keyframe_motion_th;                                     // :258
  // The next Solve's inputs: keyframe (unchanged unless this frame is promoted), next frame's pyramid (stream C, ev_next),
  // initial pose = T (:261 / :268 Reset). Started here, it runs while stream B finishes this frame's depth.
  auto start_next_solve = [&]() -> int {
    lap(2);
    if (!t->next_ready && hipStreamWaitEvent(t->ctx_a->stream, t->ev_next, 0) != hipSuccess) return 1;
    lap(3);
    return lm_solve_begin(t->lm, t->kf_img, t->kf_dep, t->next_img, false) < 0 ? -1 : 0;   // (the device is current: set at the top of this call)
  };
  bool reset_done = false;
  const auto v0 = std::chrono::steady_clock::now();
  if (chained) {
    // what did the guard tell the chained Solve? It runs (1) exactly when the device's keyframe test kept the keyframe and the Solve
    // succeeded; the host's own test decides what counts: agreement and "keep" -> the chained Solve IS the next Solve.
    int verdict = lm_chain_verdict(t->lm, solved_token, solved_slot);
    if (verdict == 0) {   // no word (a redone Solve, a finalize launch): its launches saw a guard without their token and returned
      verdict = 2;
    }
    if (verdict == 1 && !promote && st == 0) {
      odo_lm_reset(t->lm, T, 0.01f);                                                   // :268 (the pose and lambda it started from)
      lm_chain_adopt(t->lm);
      reset_done = true;
      t->chain_used++;
    } else {
      t->lm->chained.active = 0;
      if (verdict == 1) t->chain_wasted++;   // it runs for nothing: the Solve started below queues behind it
      chained = false;
    }
  }
  t->dbg_verdict_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - v0).count();
  t->arm_pending = 0;
  if (t->arm_on) {
    t->arm_on = 0;
    if (!promote && st == 0) {
      odo_lm_reset(t->lm, T, 0.01f);                                                   // :268 (the pose and lambda the armed Solve starts from)
      if (lm_arm_go(t->lm, T) == 0) reset_done = true;
    } else {
      lm_arm_abort(t->lm);
    }
  }
  if (early && !promote && !reset_done) {
    odo_lm_reset(t->lm, T, 0.01f);                                                     // :268
    reset_done = true;
    if (start_next_solve() < 0) { write_pose(); return -1; }
    t->dbg_relaunch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s1).count();
    t->dbg_relaunch_n++;
  }
  write_pose();
  // ---- collect this frame's stream-B job
  if (p.overlap_depth == 2) {
    if (tracker_wait_job(t, my_job)) { t->lm->job.active = 0; return -1; }
  } else {
    if (p.overlap_depth == 1) {
      int rc = tracker_job_drain(t, jk);
      t->cur_job = nullptr;
      if (rc == 0) rc = depth_finish(t->depth, false);
      if (rc == 2) rc = tracker_job_run(t, jk, false);   // persistent depth-LM launch gave up: the job again, on the step launches
      jk->stats = t->depth->last;
      jk->rc = rc;
      if (rc) snprintf(jk->msg, sizeof(jk->msg), "%s", g_err);
    } else {
      (void)tracker_job_run(t, jk, false);                                             // :226-252 in program order
    }
  }
  t->out_slot = jk->slot;
  t->last_depth_iters = jk->stats.iters;
  t->last_valid = jk->stats.n_valid;
  if (jk->rc) { t->lm->job.active = 0; return fail("    depth failed! (%s)", jk->msg); }   // :230-232
  int new_kf = 0;
  if (promote) {
    std::swap(t->kf_img, t->cur_img);                                                  // :259 (the :251 rebuild == the :205 pyramid)
    std::swap(t->kf_dep, t->pre_dep[jk->slot]);
    memcpy(t->kf_abs, cur, sizeof(cur));                                               // :260
    t->n_keyframes++;
    new_kf = 1;
    if (t->cand_lists) lm_adopt_candidate(t->lm, t->kf_img, t->kf_dep, (long)t->frame_id, jk->slot);  // lists built beside the Solve
  }
  if (!reset_done) odo_lm_reset(t->lm, T, 0.01f);                                      // :261 / :268 (both branches)
  if (early && promote && start_next_solve() < 0) return -1;   // against the new keyframe, as soon as its lists are adopted
  if (pose_to_keyframe) memcpy(pose_to_keyframe, T, sizeof(T));
  if (abs_pose) memcpy(abs_pose, cur, sizeof(cur));
  if (is_new_keyframe) *is_new_keyframe = new_kf;
  if (motion_mag) *motion_mag = mag;
  if (solve_status) *solve_status = st;
  t->last_evals = t->lm->last_evals;
  t->tm_frame_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f0).count();
  t->tm_frames++;
  t->dbg_post_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s1).count();
#ifdef ODO_DIAG
  g_lap_n++;
  if ((t->dbg_n + 1) % 500 == 0 && getenv("ODO_TRACK_DEBUG"))
    fprintf(stderr, "[relaunch laps us] result seen -> Solve returned %.2f | pose math %.2f | magnitude + reset %.2f | WaitEvent %.2f | check + SetDevice %.2f | "
            "keyframe check %.2f | args %.2f | coarse launch %.2f | fine launch %.2f | pump + GetLastError %.2f\n", g_lap_us[0] / g_lap_n, g_lap_us[1] / g_lap_n,
            g_lap_us[2] / g_lap_n, g_lap_us[3] / g_lap_n, g_lap_us[4] / g_lap_n, g_lap_us[5] / g_lap_n, g_lap_us[6] / g_lap_n, g_lap_us[7] / g_lap_n, g_lap_us[8] / g_lap_n,
            g_lap_us[9] / g_lap_n);
#endif
  if (++t->dbg_n % 500 == 0 && getenv("ODO_TRACK_DEBUG"))
    fprintf(stderr, "[track] per call: before the chain %.1f us, waiting for the next pyramid %.1f, chain launches %.1f, guard verdict %.1f, "
            "after the Solve returned %.1f (Solve returned -> the next Solve's launches issued: %.2f)\n", t->dbg_pre_us / t->dbg_n, t->dbg_spin_us / t->dbg_n, t->dbg_chain_us / t->dbg_n,
            t->dbg_verdict_us / t->dbg_n, t->dbg_post_us / t->dbg_n, t->dbg_relaunch_us / (t->dbg_relaunch_n ? t->dbg_relaunch_n : 1));
  return 0;
}

// Host-clock averages per tracked frame since the last call (microseconds): whole track() call, Solve (stream A, calling
// thread), the stream-B job (helper thread), and the time the calling thread waited for the helper after its own work.
extern "C" int odo_tracker_timing(odo_tracker* t, double out[4]) {
  if (!t || !out) return fail("NULL arg");
  const double n = t->tm_frames > 0 ? (double)t->tm_frames : 1.0;
  out[0] = t->tm_frame_us / n; out[1] = t->tm_solve_us / n; out[2] = t->tm_depth_us / n; out[3] = t->tm_wait_us / n;
  t->tm_solve_us = t->tm_depth_us = t->tm_frame_us = t->tm_wait_us = 0.0; t->tm_frames = 0;
  return 0;
}

// Optional: announce the left image of the NEXT frame before calling odo_tracker_track for the current one.
extern "C" int odo_tracker_hint_next(odo_tracker* t, const float* next_left_dev) {
  if (!t) return fail("NULL tracker");
  t->hint_next = next_left_dev;
  t->hint_next_right = nullptr;
  return 0;
}
// The same with the right image: the next frame's ComputeDepth + candidate pyramids are then enqueued a frame early as well
// (the depth stream runs a frame ahead of the pose LM, so a short Solve no longer waits for it).
extern "C" int odo_tracker_hint_next_pair(odo_tracker* t, const float* next_left_dev, const float* next_right_dev) {
  if (!t) return fail("NULL tracker");
  t->hint_next = next_left_dev;
  t->hint_next_right = next_left_dev ? next_right_dev : nullptr;
  return 0;
}

extern "C" int odo_tracker_stats(const odo_tracker* t, int* lm_evals, int* depth_iters, int* n_valid, int* n_kf) {
  if (!t) return fail("NULL tracker");
  if (lm_evals) *lm_evals = t->last_evals;
  if (depth_iters) *depth_iters = t->last_depth_iters;
  if (n_valid) *n_valid = t->last_valid;
  if (n_kf) *n_kf = t->n_keyframes;
  return 0;
}
extern "C" int odo_tracker_outputs(const odo_tracker* t, const uint8_t** val, const float** disp, const float** dep) {
  if (!t) return fail("NULL tracker");
  if (val) *val = t->d_val[t->out_slot];
  if (disp) *disp = t->d_disp[t->out_slot];
  if (dep) *dep = t->d_dep[t->out_slot];
  return 0;
}
extern "C" int odo_tracker_chain_stats(const odo_tracker* t, long* adopted, long* wasted) {
  if (!t) return fail("NULL tracker");
  if (adopted) *adopted = t->chain_used;
  if (wasted) *wasted = t->chain_wasted;
  return 0;
}
extern "C" int odo_tracker_arm_stats(const odo_tracker* t, long* started, long* returned) {
  if (!t) return fail("NULL tracker");
  if (started) *started = t->lm->arm_used;
  if (returned) *returned = t->lm->arm_aborted;
  return 0;
}
extern "C" odo_lm* odo_tracker_lm(odo_tracker* t) { return t ? t->lm : nullptr; }
extern "C" odo_depth* odo_tracker_depth(odo_tracker* t) { return t ? t->depth : nullptr; }
extern "C" odo_ctx* odo_tracker_ctx(odo_tracker* t) { return t ? t->ctx_a : nullptr; }

// bench.py roofline leg: `reps` launches of the evaluation kernel (residual / normal-equation pass, without the LM
// update) on `level` at pose T, each bracketed by HIP events on the stream it is launched on. Returns mean / min
// launch duration, the algorithmic bytes of one launch and the residual count.
static int lm_time_eval(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, int level,
                        const float* T_colmajor, int reps, float* mean_us, float* min_us, double* algorithmic_bytes,
                        int* n_points) {
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
  HIP_OK(hipMemcpyAsync(m->d_init, T_colmajor, sizeof(float) * 16, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(lm_force_state_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_init, level);
  std::vector<hipEvent_t> ev(2 * (size_t)reps);
  for (auto& e : ev) HIP_OK(hipEventCreate(&e));
  for (int w = 0; w < 3; w++) lm_launch_eval(m, v, k, level, nblk);  // warm caches
  for (int i = 0; i < reps; i++) lm_launch_eval(m, v, k, level, nblk, ev[2 * i], ev[2 * i + 1]);
  double* d_acc = nullptr;
  HIP_OK(hipMalloc((void**)&d_acc, sizeof(double) * ODO_NACC));
  hipLaunchKernelGGL(lm_sum_partials_kernel, dim3(1), dim3(256), 0, s, m->d_partials, nblk, d_acc);
  double acc[ODO_NACC];
  HIP_OK(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  HIP_OK(hipFree(d_acc));
  double sum = 0.0, mn = 1e30;
  for (int i = 0; i < reps; i++) {
    float ms = 0.0f;
    HIP_OK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    sum += ms; if (ms < mn) mn = ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (mean_us) *mean_us = (float)(sum / reps * 1000.0);
  if (min_us) *min_us = (float)(mn * 1000.0);
  if (algorithmic_bytes) *algorithmic_bytes = lm_level_bytes(m, level, v.rows, v.cols, nblk);
  if (n_points) *n_points = (int)acc[28];
  return 0;
}

extern "C" int odo_lm_time_eval(odo_lm* m, const odo_pyr* kf_img, const odo_pyr* kf_dep, const odo_pyr* cur_img, int level,
                                const float T_colmajor[16], int reps, float* mean_us, float* min_us,
                                double* algorithmic_bytes, int* n_points) {
  if (!T_colmajor || reps < 1) return fail("odo_lm_time_eval: bad arg");
  if (lm_check_pyrs(m, kf_img, kf_dep, cur_img)) return -1;
  if (level < 0 || level >= m->n_levels) return fail("odo_lm_time_eval: bad level");
  return lm_time_eval(m, kf_img, kf_dep, cur_img, level, T_colmajor, reps, mean_us, min_us, algorithmic_bytes, n_points);
}

// The same for n streams in ONE launch (lm_dense_eval_batch_kernel): `level` must be dense for every optimiser. Times `reps`
// event-bracketed batched launches; algorithmic_bytes is the total over the streams (12 B per interior pixel each).
extern "C" int odo_lm_time_eval_batch(int n, odo_lm* const* lms, const odo_pyr* const* kf_img, const odo_pyr* const* kf_dep,
                                      const odo_pyr* const* cur_img, int level, const float T_colmajor[16], int reps,
                                      float* mean_us, float* min_us, double* algorithmic_bytes, int* n_points_total) {
  if (n < 1 || n > 64 || !lms || !kf_img || !kf_dep || !cur_img || !T_colmajor || reps < 1) return fail("odo_lm_time_eval_batch: bad arg");
  for (int i = 0; i < n; i++) {
    if (!lms[i] || lms[i]->ctx != lms[0]->ctx) return fail("odo_lm_time_eval_batch: the optimisers must share one context");
    if (lm_check_pyrs(lms[i], kf_img[i], kf_dep[i], cur_img[i])) return -1;
    if (level < 0 || level >= lms[i]->n_levels) return fail("odo_lm_time_eval_batch: bad level");
    if (lm_prepare_keyframe(lms[i], kf_img[i], kf_dep[i])) return -1;
    if (lms[i]->use_list[level] || lms[i]->robust == 2) return fail("odo_lm_time_eval_batch: level %d of stream %d is not a dense Huber / L2 level", level, i);
  }
  odo_ctx* cx = lms[0]->ctx;
  hipStream_t s = cx->stream;
  HIP_OK(hipSetDevice(cx->device));
  DenseBatchItem* h = nullptr;
  DenseBatchItem* d = nullptr;
  std::vector<hipEvent_t> ev;
  double* d_acc = nullptr;
  struct Cleanup {   // every early return below (HIP_OK) gives the tables and the events back
    DenseBatchItem*& h; DenseBatchItem*& d; std::vector<hipEvent_t>& ev; double*& d_acc;
    ~Cleanup() {
      for (auto& e : ev) if (e) (void)hipEventDestroy(e);
      if (d_acc) (void)hipFree(d_acc);
      if (d) (void)hipFree(d);
      if (h) (void)hipHostFree(h);
    }
  } cleanup{h, d, ev, d_acc};
  HIP_OK(hipHostMalloc((void**)&h, sizeof(DenseBatchItem) * (size_t)n, hipHostMallocDefault));
  HIP_OK(hipMalloc((void**)&d, sizeof(DenseBatchItem) * (size_t)n));
  int max_nblk = 1;
  double bytes = 0.0;
  for (int i = 0; i < n; i++) {
    odo_lm* m = lms[i];
    LevelView v;
    v.I1 = kf_img[i]->dev + kf_img[i]->off[level];
    v.I2 = cur_img[i]->dev + cur_img[i]->off[level];
    v.D1 = kf_dep[i]->dev + kf_dep[i]->off[level];
    v.rows = kf_img[i]->r[level]; v.cols = kf_img[i]->c[level];
    const DenseLevel L = lm_dense_level(v, lm_level_k(m, level), 0);
    memset(&h[i], 0, sizeof(h[i]));
    h[i].L = L; h[i].st = m->d_state; h[i].scale_sqr = m->d_scale; h[i].partials = m->d_partials;
    h[i].expect_level = level; h[i].robust = m->robust; h[i].huber_delta = m->huber_delta;
    if (L.nblk > max_nblk) max_nblk = L.nblk;
    bytes += lm_level_bytes(m, level, v.rows, v.cols, L.nblk);
    HIP_OK(hipMemcpyAsync(m->d_init, T_colmajor, sizeof(float) * 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lm_force_state_kernel, dim3(1), dim3(64), 0, s, m->d_state, m->d_init, level);
  }
  HIP_OK(hipMemcpyAsync(d, h, sizeof(DenseBatchItem) * (size_t)n, hipMemcpyHostToDevice, s));
  ev.assign(2 * (size_t)reps, nullptr);
  for (auto& e : ev) HIP_OK(hipEventCreate(&e));
  for (int w = 0; w < 3; w++) launch_dense_eval_batch(d, n, max_nblk, s, nullptr, nullptr, lms[0]->dense_plain_div);
  for (int i = 0; i < reps; i++) launch_dense_eval_batch(d, n, max_nblk, s, ev[2 * i], ev[2 * i + 1], lms[0]->dense_plain_div);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(s));
  double sum = 0.0, mn = 1e30;
  for (int i = 0; i < reps; i++) {
    float ms = 0.0f;
    HIP_OK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    sum += ms; if (ms < mn) mn = ms;
  }
  // residual counts: fold each stream's partial rows (column 28)
  int npts = 0;
  HIP_OK(hipMalloc((void**)&d_acc, sizeof(double) * ODO_NACC));
  for (int i = 0; i < n; i++) {
    hipLaunchKernelGGL(lm_sum_partials_kernel, dim3(1), dim3(256), 0, s, lms[i]->d_partials, h[i].L.nblk, d_acc);
    double acc[ODO_NACC];
    HIP_OK(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    npts += (int)acc[28];
  }
  if (mean_us) *mean_us = (float)(sum / reps * 1000.0);
  if (min_us) *min_us = (float)(mn * 1000.0);
  if (algorithmic_bytes) *algorithmic_bytes = bytes;
  if (n_points_total) *n_points_total = npts;
  return 0;
}

extern "C" int odo_tracker_time_residual(odo_tracker* t, int level, int reps, float* mean_us, float* min_us,
                                         double* algorithmic_bytes, int* n_points) {
  if (!t || reps < 1 || level < 0 || level >= t->p.levels) return fail("odo_tracker_time_residual: bad arg");
  return lm_time_eval(t->lm, t->kf_img, t->kf_dep, t->cur_img, level, t->pose_to_kf, reps, mean_us, min_us,
                      algorithmic_bytes, n_points);
}

// Config 5 leg of bench.py: event-timed stages of the disparity front end on device-resident images:
// us[0] blur (both images), us[1] point selection, us[2] epipolar SSD scan; candidates = SSD evaluations of one scan.
extern "C" int odo_depth_time_stages(odo_depth* d, const float* left_dev, const float* right_dev, int rows, int cols,
                                     int reps, float us[3], double* candidates, int* n_selected) {
  if (!d || !left_dev || !right_dev || !us || reps < 1) return fail("odo_depth_time_stages: bad arg");
  if (depth_check_size(d, rows, cols)) return -1;
  HIP_OK(hipSetDevice(d->ctx->device));
  if (depth_ensure(d, rows, cols)) return -1;
  hipStream_t s = d->ctx->stream;
  hipEvent_t e[6];  // dispatch-bound start / stop events per stage
  for (auto& x : e) HIP_OK(hipEventCreate(&x));
  double tot[3] = {0, 0, 0};
  for (int r = -2; r < reps; r++) {  // two warm-up rounds
    hipExtLaunchKernelGGL(blur3x3_kernel, grid2d(cols, rows, 2), dim3(256), 0, s, e[0], e[1], 0, left_dev, d->d_bl, right_dev,
                          d->d_br, rows, cols, d->d_val, d->d_disp, d->d_dep);
    hipExtLaunchKernelGGL(depth_select_kernel, dim3(kSelBlocks), dim3(kSelThreads), 0, s, e[2], e[3], 0,
                          (const float*)d->d_bl, rows, cols, d->boundary, d->grad_th, d->d_val, d->d_pts, d->d_cnt);
    depth_launch_scan(d, s, rows, cols, d->d_disp, d->d_dep, e[4], e[5]);
    HIP_OK(hipStreamSynchronize(s));
    if (r >= 0)
      for (int k = 0; k < 3; k++) { float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e[2 * k], e[2 * k + 1])); tot[k] += ms * 1000.0; }
  }
  for (auto& x : e) (void)hipEventDestroy(x);
  for (int k = 0; k < 3; k++) us[k] = (float)(tot[k] / reps);
  // count the candidates of one scan from the point list
  std::vector<uint32_t> pts(kSelBlocks * kSelCap);
  std::vector<int> cnt(kSelBlocks);
  HIP_OK(hipMemcpy(pts.data(), d->d_pts, sizeof(uint32_t) * pts.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(cnt.data(), d->d_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
  double cand = 0.0;
  int nsel = 0;
  for (int b = 0; b < kSelBlocks; b++)
    for (int k = 0; k < cnt[b]; k++) {
      const int x = (int)(pts[b * kSelCap + k] & 0xffffu);
      int lo = d->boundary;
      if (d->max_disparity > 0 && x - d->max_disparity > lo) lo = x - d->max_disparity;
      cand += (x > lo) ? (x - lo) : 0;
      nsel++;
    }
  if (candidates) *candidates = cand;
  if (n_selected) *n_selected = nsel;
  return 0;
}
