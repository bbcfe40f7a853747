/* track_resident.c — the whole frame loop of run_odometry_kitti_offline.cpp:198-271 through the C ABI (odo_tracker_*), plain C,
 * frames resident on the device: what an offline evaluation (KITTI: all frames on disk) does for full speed.
 *
 *   python examples/make_frames.py frames.bin 40
 *   gcc -O2 -Iinclude examples/track_resident.c -o track_resident -Lodometry_amd/lib -lodometry_hip -Wl,-rpath,$PWD/odometry_amd/lib -lm
 *   ./track_resident frames.bin [--passes N] [--no-announce] [--rel-bin out.bin]
 *
 * The next stereo pair is announced before each frame is tracked (odo_tracker_hint_next_pair): its image pyramid, ComputeDepth
 * and keyframe-candidate lists then run a frame ahead and the next Solve starts the moment this one returns. Results do not
 * depend on the announcements (--no-announce gives the same poses, slower).
 * Prints one line per frame (like examples/run_odometry_synth.cpp) and, with --passes, "TRACK_FPS <frames/s>" on stderr. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "odometry_hip.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    if ((call) != 0) {                                                     \
      fprintf(stderr, "%s failed: %s\n", #call, odo_last_error());         \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s frames.bin [--passes N] [--no-announce] [--rel-bin out.bin]\n", argv[0]); return 2; }
  int passes = 0, announce = 1;
  const char* rel_out = NULL;
  for (int i = 2; i < argc; i++) {
    if (!strcmp(argv[i], "--passes") && i + 1 < argc) passes = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-announce")) announce = 0;
    else if (!strcmp(argv[i], "--rel-bin") && i + 1 < argc) rel_out = argv[++i];
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { printf("cannot read %s\n", argv[1]); return 2; }
  int hdr[3];
  if (fread(hdr, sizeof(int), 3, f) != 3) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  const size_t px = (size_t)rows * cols;

  odo_tracker_params p;
  CHECK(odo_tracker_default_params(&p));     /* the runner's constants (ref: :41-145) */
  p.rows = rows; p.cols = cols;
  p.any_size = !(rows == 376 && cols == 1241);
  odo_tracker* trk = NULL;
  CHECK(odo_tracker_create(0, &p, &trk));
  odo_ctx* ctx = odo_tracker_ctx(trk);

  /* all frames to the device once (pinned staging: plain asynchronous DMAs) */
  float** left = (float**)calloc((size_t)n, sizeof(float*));
  float** right = (float**)calloc((size_t)n, sizeof(float*));
  float* host = (float*)odo_host_alloc(sizeof(float) * px);
  if (!host) { fprintf(stderr, "odo_host_alloc failed: %s\n", odo_last_error()); return 1; }
  for (int k = 0; k < n; k++) {
    for (int side = 0; side < 2; side++) {
      if (fread(host, sizeof(float), px, f) != px) { printf("short file\n"); return 2; }
      void* d = NULL;
      CHECK(odo_dev_alloc(ctx, sizeof(float) * px, &d));
      CHECK(odo_dev_upload(ctx, d, host, sizeof(float) * px));
      (side ? right : left)[k] = (float*)d;
    }
  }
  fclose(f);
  odo_host_free(host);

  const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float* rel = (float*)calloc((size_t)n * 16, sizeof(float));
  double secs = 0.0;
  for (int pass = 0; pass <= passes; pass++) {
    const double t0 = now_s();
    CHECK(odo_tracker_init(trk, left[0], right[0], eye));   /* frame 0: ref :95-145 */
    int kf = 0;
    for (int k = 1; k < n; k++) {
      if (announce && k + 1 < n) CHECK(odo_tracker_hint_next_pair(trk, left[k + 1], right[k + 1]));
      float T[16], A[16], mag = 0.0f;
      int new_kf = 0, st = 0;
      if (odo_tracker_track(trk, left[k], right[k], T, A, &new_kf, &mag, &st) != 0) {
        printf("    depth failed!\n");   /* ref :230-232: the runner leaves its loop */
        break;
      }
      kf += new_kf;
      if (pass == 0) {
        memcpy(rel + (size_t)k * 16, T, sizeof(T));
        printf("frame %d kf %d motion %.4f  t = [% .5f % .5f % .5f]\n", k, kf, mag, A[12], A[13], A[14]);
      } else if (memcmp(rel + (size_t)k * 16, T, sizeof(T)) != 0) {
        fprintf(stderr, "TRACK_MISMATCH pass %d frame %d\n", pass, k);
        return 3;
      }
    }
    if (pass > 0) secs += now_s() - t0;
    if (pass == 0) printf("Total keyframes: %d\n", kf);
  }
  if (passes > 0) fprintf(stderr, "TRACK_FPS %.1f FRAMES %d PASSES %d\n", (double)(n - 1) * passes / secs, (n - 1) * passes, passes);
  if (rel_out) {
    FILE* g = fopen(rel_out, "wb");
    if (!g) return 1;
    fwrite(rel + 16, sizeof(float), (size_t)(n - 1) * 16, g);
    fclose(g);
  }
  for (int k = 0; k < n; k++) { odo_dev_free(ctx, left[k]); odo_dev_free(ctx, right[k]); }
  free(left); free(right); free(rel);
  odo_tracker_destroy(trk);
  return 0;
}
