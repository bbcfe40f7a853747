// dense_ablate.hip — A/B timing of the dense-level evaluation kernel variants on a synthetic 1920x1080 level
// (BASELINE.json configs[2]) in ONE process: HIP events around back-to-back launches, sums cross-checked between the
// variants that compute the real thing. Build + run (GPU box):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
//         -o /tmp/dense_ablate tools/microbench/dense_ablate.hip && /tmp/dense_ablate
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define ODO_DENSE_KERNELS 1
#include "../../odometry_amd/csrc/kernels.hip.h"
using namespace odo;

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

static float tex(float x, float y) {
  const float v = 128.0f + 40.0f * sinf(0.05f * x + 0.3f * sinf(0.02f * y)) + 30.0f * sinf(0.11f * y + 0.07f * x) +
                  20.0f * sinf(0.31f * x) * sinf(0.23f * y);
  return floorf(fminf(fmaxf(v, 0.0f), 255.0f));
}

__global__ void set_state(LmState* st, const float* T, int level) {
  if (threadIdx.x == 0) {
    LmState s;
    memset(&s, 0, sizeof(s));
    s.level = level; s.active = 1;
    for (int i = 0; i < 16; i++) s.T[i] = T[i];
    *st = s;
  }
}
__global__ void sum_rows(const double* __restrict__ partials, int nblk, double* __restrict__ out) {
  const int q = threadIdx.x;
  if (q < ODO_NACC) {
    double v = 0.0;
    for (int b = 0; b < nblk; b++) v += partials[(size_t)b * ODO_NACC + q];
    out[q] = v;
  }
}

struct Result { double us; double acc[ODO_NACC]; };
static const char* g_filter = nullptr;

template <typename F>
static Result run(const char* name, hipStream_t s, double* d_part, int nblk, double* d_acc, F launch, double bytes, const Result* ref) {
  CK(hipMemsetAsync(d_part, 0, sizeof(double) * 4096 * ODO_NACC, s));
  if (g_filter && !strstr(name, g_filter)) return Result();
  for (int i = 0; i < 3; i++) launch(nullptr, nullptr);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 40;
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; i++) launch(nullptr, nullptr);
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // kernel durations proper: start / stop events bound to each dispatch
  std::vector<hipEvent_t> ev(2 * reps);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < reps; i++) launch(ev[2 * i], ev[2 * i + 1]);
  CK(hipStreamSynchronize(s));
  double kus = 0.0;
  for (int i = 0; i < reps; i++) { float m2 = 0; CK(hipEventElapsedTime(&m2, ev[2 * i], ev[2 * i + 1])); kus += m2 * 1000.0 / reps; }
  for (auto& e : ev) (void)hipEventDestroy(e);
  hipLaunchKernelGGL(sum_rows, dim3(1), dim3(64), 0, s, d_part, nblk, d_acc);
  Result r;
  CK(hipMemcpyAsync(r.acc, d_acc, sizeof(r.acc), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  r.us = ms * 1000.0 / reps;
  double worst = 0.0;
  if (ref)
    for (int q = 0; q < ODO_NACC; q++) {
      const double d = fabs(r.acc[q] - ref->acc[q]) / (fabs(ref->acc[q]) + 1e-300);
      if (d > worst) worst = d;
    }
  printf("%-52s kernel %7.2f us (frac %.3f) | back-to-back %7.2f us  N=%.0f err=%.6e  %s%.1e\n", name, kus, bytes / kus * 1e-3 / 8000.0, r.us,
         r.acc[28], r.acc[27] / (r.acc[28] > 0 ? r.acc[28] : 1), ref ? "max rel diff vs ref " : "", worst);
  fflush(stdout);
  return r;
}

int main(int argc, char** argv) {
  const int rows = argc > 2 ? atoi(argv[2]) : 1080, cols = argc > 1 ? atoi(argv[1]) : 1920;
  g_filter = argc > 3 ? argv[3] : nullptr;
  const size_t n = (size_t)rows * cols;
  std::vector<float> I1(n), I2(n), D1(n);
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      I1[(size_t)y * cols + x] = tex((float)x, (float)y);
      I2[(size_t)y * cols + x] = tex((float)x + 1.3f, (float)y + 0.4f);
      const float Z = 6.0f + 0.004f * (float)y + 2.0f * sinf(0.003f * (float)x);  // smooth surface, 6-12 m
      D1[(size_t)y * cols + x] = 1.0f / Z;
    }
  float *dI1, *dI2, *dD1, *dT;
  CK(hipMalloc(&dI1, n * 4)); CK(hipMalloc(&dI2, n * 4)); CK(hipMalloc(&dD1, n * 4)); CK(hipMalloc(&dT, 64));
  CK(hipMemcpy(dI1, I1.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dI2, I2.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dD1, D1.data(), n * 4, hipMemcpyHostToDevice));
  // small motion: rotation about y by 0.002 rad, translation (0.02, -0.01, 0.05); column-major
  const float c = cosf(0.002f), sn = sinf(0.002f);
  const float T[16] = {c, 0, -sn, 0, 0, 1, 0, 0, sn, 0, c, 0, 0.02f, -0.01f, 0.05f, 1};
  CK(hipMemcpy(dT, T, 64, hipMemcpyHostToDevice));
  LmState* d_st; CK(hipMalloc(&d_st, sizeof(LmState)));
  double *d_part, *d_acc; CK(hipMalloc(&d_part, sizeof(double) * 4096 * ODO_NACC)); CK(hipMalloc(&d_acc, sizeof(double) * 32));
  float* d_scale; CK(hipMalloc(&d_scale, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipLaunchKernelGGL(set_state, dim3(1), dim3(64), 0, s, d_st, dT, 0);
  const LevelK k = make_level_k(1100.0f, 959.5f, 539.5f, 0);
  const double bytes = 12.0 * (double)(rows - 8) * (cols - 8);
  printf("level %dx%d, interior %d px, algorithmic bytes %.2f MB\n", cols, rows, (rows - 8) * (cols - 8), bytes * 1e-6);
  LevelView v; v.I1 = dI1; v.I2 = dI2; v.D1 = dD1; v.rows = rows; v.cols = cols;
  const int robust = 1; const float hd = 28.0f;
  Result ref = run("round-1 lm_residual_dense_kernel (1280x256)", s, d_part, 1280, d_acc, [&](hipEvent_t a, hipEvent_t b) {
    hipExtLaunchKernelGGL(lm_residual_dense_kernel, dim3(1280), dim3(256), 0, s, a, b, 0, v, k, (const LmState*)d_st, 0, robust, hd, (const float*)d_scale, d_part);
  }, bytes, nullptr);
  DenseLevel L; memset(&L, 0, sizeof(L));
  L.I1 = dI1; L.I2 = dI2; L.D1 = dD1; L.rows = rows; L.cols = cols; L.k = k;
  L.fast_ok = dense_fast_ok(k.fl, k.cx, k.cy, rows, cols);
  printf("fast_ok %d\n", L.fast_ok);
#define RUN(NAME, BLOCK, FLAGS, MAXB, WAVES)                                                                                   \
  do {                                                                                                                  \
    DenseLevel LL = L; dense_level_geometry(&LL, BLOCK, MAXB);                                                          \
    char nm[96]; snprintf(nm, sizeof(nm), "%s [%d x %d, %d w/SIMD]", NAME, LL.nblk, BLOCK, WAVES);                                        \
    run(nm, s, d_part, LL.nblk, d_acc, [&](hipEvent_t a, hipEvent_t b) {                                                \
      hipExtLaunchKernelGGL((lm_dense_eval_kernel<BLOCK, FLAGS, WAVES>), dim3(LL.nblk), dim3(BLOCK), 0, s, a, b, 0, LL, (const LmState*)d_st, 0, robust, hd, \
                         (const float*)d_scale, d_part);                                                                \
    }, bytes, ((FLAGS) & ~1) ? nullptr : &ref);                                                                         \
  } while (0)
  RUN("unpipelined, shared reciprocals", 256, 32, 1024, 4);
  RUN("pipelined, plain divisions", 256, 1, 1024, 4);
  RUN("pipelined, shared reciprocals", 256, 0, 1280, 5);
  RUN("pipelined, shared reciprocals", 256, 0, 1024, 4);
  RUN("pipelined, shared reciprocals", 256, 0, 768, 3);
  RUN("pipelined, shared reciprocals", 256, 0, 2048, 4);
  RUN("pipelined, shared reciprocals", 512, 0, 512, 4);
  RUN("pipelined, shared reciprocals", 1024, 0, 256, 4);
  RUN("pipelined ablate: no normal-equation products", 256, 2, 1024, 4);
  // FETCH_SIZE calibration for this access pattern (one dword per lane, coalesced rows): streams D1 and I1 of every interior
  // pixel (2 x 4 B x interior = a known byte count) and almost nothing else (bogus geometry: the taps are skipped)
  RUN("calibration: D1 + I1 streamed, no taps", 256, 62, 1024, 4);
  return 0;
}
