// tests/hostemu.cpp — TEST INFRASTRUCTURE. Compiles odometry_amd/csrc/odo_math.h (the arithmetic the HIP
// kernels execute) for the host and steps it serially, so the per-point chain, the SE(3) update and the LM
// state machine can be checked against the oracle on a machine without a GPU. Never linked into the product.
#include "../odometry_amd/csrc/odo_math.h"

#include <string.h>

using namespace odo;

extern "C" {

static int g_emu_bilinear = 0;
void emu_set_sampling(int bilinear) { g_emu_bilinear = bilinear; }   // the device's odo_lm_set_sampling

int emu_lm_accumulate(const float* I1, const float* I2, const float* D1, int rows, int cols, int level, const float* T,
                      int robust, float huber_delta, float f0, float cx0, float cy0, double* acc) {
  LevelK k = make_level_k(f0, cx0, cy0, level);
  k.bilinear = g_emu_bilinear;
  for (int i = 0; i < ODO_NACC; i++) acc[i] = 0.0;
  for (int y = 4; y < rows - 4; y++)
    for (int x = 4; x < cols - 4; x++) {
      const size_t o = (size_t)y * cols + x;
      if (!depth_valid(D1[o])) continue;
      const PointK p = make_point(x, y, D1[o], I1[o], k);
      float r, J[6];
      if (!point_residual(p, T, k, I2, rows, cols, &r, J)) continue;
      accumulate_row(acc, r, robust_weight(r, robust, huber_delta, 1.0f), J);
    }
  return acc[28] > 0.0 ? 0 : -1;
}

// Full Solve with the device state machine (robust 0/1), pyramids stored level after level.
int emu_lm_solve(const float* img1, const float* dep1, const float* img2, int rows, int cols, int n_levels,
                 const int* max_iters, float lambda, float precision, int robust, float huber_delta, float f0, float cx0,
                 float cy0, const float* init, float* out, int* n_evals) {
  LmState s;
  lm_begin_solve(&s, init);
  long off[8];
  int rr[8], cc[8];
  long tot = 0;
  int r = rows, c = cols;
  for (int l = 0; l < n_levels; l++) { off[l] = tot; rr[l] = r; cc[l] = c; tot += (long)r * c; r /= 2; c /= 2; }
  for (int l = n_levels - 1; l >= 0; l--) {
    lm_begin_level(&s, l, lambda, max_iters[l]);
    for (int it = 0; it < max_iters[l]; it++) {
      if (!(s.active && s.level == l)) continue;  // what a stale launch does
      double acc[ODO_NACC];
      emu_lm_accumulate(img1 + off[l], img2 + off[l], dep1 + off[l], rr[l], cc[l], l, s.T, robust, huber_delta, f0, cx0,
                        cy0, acc);
      lm_consume(&s, acc, precision, max_iters[l]);
    }
  }
  if (s.status == 0) se3_to_colmajor(s.cur, out);
  else { memset(out, 0, sizeof(float) * 16); out[0] = out[5] = out[10] = 1.0f; }
  *n_evals = s.n_evals;
  return s.status;
}

void emu_se3_exp(const float* a, float* M) { Se3 s; se3_exp(a, &s); se3_to_colmajor(s, M); }
void emu_se3_roundtrip(const float* Min, float* Mout) { Se3 s; se3_from_colmajor(Min, &s); se3_to_colmajor(s, Mout); }
void emu_se3_left_update(const float* d6, const float* cur, float* out) {
  Se3 d, c, o;
  se3_exp(d6, &d);
  se3_from_colmajor(cur, &c);
  se3_left_update(d, c, &o);
  se3_to_colmajor(o, out);
}
void emu_solve_damped(const double* acc, float lambda, float* delta) { solve_damped(acc, lambda, delta); }
void emu_sincos(float x, float* s, float* c) { sincos_f(x, s, c); }
float emu_ssd8(const float* L, const float* R) { return ssd8_tree(L, R); }
float emu_cx_level(float c, int level) { return cx_level(c, level); }
// The device's LM state machine (lm_begin_level / lm_consume: what lm_step_kernel and lm_coarse_kernel run) replayed on a given
// sequence of errors, the poses tagged through their x translation: the twin of orc_lm_schedule / ref_lm_schedule.
int emu_lm_schedule(const float* errs, int n_errs, float lambda0, float precision, int max_iters, int* rec, int* final_current) {
  LmState s;
  const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  lm_begin_solve(&s, eye);
  lm_begin_level(&s, 0, lambda0, max_iters);
  int k = 0;
  while (s.active && k < n_errs) {
    double acc[ODO_NACC];
    for (int i = 0; i < ODO_NACC; i++) acc[i] = 0.0;
    acc[27] = (double)errs[k]; acc[28] = 1.0;          // err_now = float(acc[27] / acc[28]) = errs[k]; a zero system: zero step
    lm_consume(&s, acc, precision, max_iters);
    memcpy(&rec[5 * k + 0], &s.lambda, sizeof(int));
    memcpy(&rec[5 * k + 1], &s.err_last, sizeof(int));
    rec[5 * k + 2] = (int)s.cur.tx; rec[5 * k + 3] = (int)s.last.tx;
    const int broke = (!s.active && (s.stop_reason == 1 || s.stop_reason == 2)) ? 1 : 0;
    rec[5 * k + 4] = broke;
    k++;
    if (broke) break;
    s.inc.tx = (float)k;                               // the pose solved after this evaluation
  }
  *final_current = (int)s.cur.tx;
  return k;
}
// The device's inverse-depth LM driver (depth_lm_begin / depth_lm_decide / depth_lm_advance: what every block of
// depth_lm_step_kernel runs) replayed on a given sequence of errors, the depth vectors as tags (0 = the scan's depths, k + 1 = the
// vector solved after evaluation k; pre starts as the zero vector, tag -1): the twin of orc_depth_lm_schedule /
// ref_depth_lm_schedule. rec = 5 ints per evaluation {lambda bits, err_last bits, current tag, pre tag, broke}.
int emu_depth_lm_schedule(const float* errs, int n_errs, float lambda0, float precision, int max_iters, int* rec, int* final_current,
                          int* iters) {
  DepthLmState st;
  depth_lm_begin(&st, lambda0, max_iters);
  int cur = 0, pre = -1, tmp = 0, k = 0;
  while (!st.done && k < n_errs) {
    const int mode = depth_lm_decide(&st, errs[k], precision);
    if (mode == 0) cur = pre;                            // the kernel's `c = pf_pre`
    else if (mode == 1 || mode == 3) { cur = tmp; pre = cur; }
    memcpy(&rec[5 * k + 0], &st.lambda, sizeof(int));
    memcpy(&rec[5 * k + 1], &st.err_last, sizeof(int));
    rec[5 * k + 2] = cur; rec[5 * k + 3] = pre; rec[5 * k + 4] = (mode >= 2) ? 1 : 0;
    k++;
    depth_lm_advance(&st, mode, max_iters);
    if (mode >= 2) break;
    tmp = k;
  }
  *final_current = cur;
  *iters = st.iter;
  return k;
}
}
