// odo_math.h — per-point photometric chain, SE(3) update and damped 6x6 solve.
//
// Single source for the arithmetic the HIP kernels execute. Every fp32 operation here rounds once
// (compile with -ffp-contract=off; FMA is only used where it is exact by construction), so the
// results are a function of the inputs alone. Marked ODO_HD so tests can also compile these
// functions for the host and step them against the oracle without a GPU (tests/hostemu.cpp) —
// the product library only ever calls them from device code.
//
// Arithmetic spec (what "parity mode" means) follows the reference line by line:
//   reproject / warp / gradient / Jacobian  ref: src/lm_optimizer.cpp:190-237,
//                                                include/image_processing_global.h:22-69
//   LM accept/reject + damping               ref: src/lm_optimizer.cpp:117-155
//   SE3 exp / matrix / ctor-from-4x4          ref: third_party/Sophus/sophus/se3.hpp:272-278,495-502,765-786
//                                                third_party/Sophus/sophus/so3.hpp:302-304,463-468,577-611
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ODO_HD __host__ __device__ __forceinline__
#else
#define ODO_HD static inline
#endif

#define ODO_NACC 29  // 21 upper-tri JtWJ + 6 JtWr + sum(w r^2) + N

namespace odo {

// ---------------------------------------------------------------------------------------------
// Level intrinsics. ref: image_processing_global.h:22-28 (GetCxLevel), :35 (718.856f / pow(2.0f, level)).
// std::pow(float,int) is double, so the focal length of a level is carried in double (exact: power-of-2 scale).
// ---------------------------------------------------------------------------------------------
struct LevelK {
  double fl;   // f0 / 2^level
  float cx, cy;
  int bilinear;  // 0 = the reference's floor sampling (parity mode); 1 = bilinear sampling (odo_lm_set_sampling, non-parity)
};

ODO_HD float cx_level(float c, int level) {
  float v = c;
  for (int i = 0; i < level; i++) v = (v + 0.5f) / 2.0f + 0.5f;
  return v;
}
ODO_HD LevelK make_level_k(float f0, float cx0, float cy0, int level) {
  LevelK k;
  double s = 1.0;
  for (int i = 0; i < level; i++) s *= 2.0;
  k.fl = (double)f0 / s;
  k.cx = cx_level(cx0, level);
  k.cy = cx_level(cy0, level);
  k.bilinear = 0;
  return k;
}

// ---------------------------------------------------------------------------------------------
// Keyframe-side constants of one semi-dense point: everything that does not depend on the pose.
// (Back-projection and the geometric Jacobian are evaluated at the UN-warped point,
//  ref: lm_optimizer.cpp:219-234, so they can be computed once per keyframe.)
// ---------------------------------------------------------------------------------------------
struct PointK {
  float X, Y, Z;       // ReprojectToCameraFrame, ref: image_processing_global.h:35-38
  float i1;            // keyframe intensity I1(y,x)
  float fx_z;          // jw(0,0) = jw(1,1)
  float jw02, jw03, jw04, jw05, jw12, jw13, jw14, jw15;
};

ODO_HD bool depth_valid(float d) { return !(fabsf(d - 0.0f) < 0.01f); }  // ref: lm_optimizer.cpp:193

// make_point in two halves (the dense evaluation kernel runs them at different times): the back-projected point ...
ODO_HD void point_xyz(int x, int y, float inv_depth, const LevelK& k, float* X, float* Y, float* Z) {
  const float z = 1.0f / inv_depth;                                   // :198
  // The reference evaluates these quotients in double (std::pow returns double) and rounds to float.
  // Both operands are exactly representable in fp32 (fl = f0 / 2^level), and a correctly rounded fp64 quotient
  // of two fp32 values rounds to the correctly rounded fp32 quotient (53 >= 2*24 + 2), so the IEEE fp32 divide
  // below is bit-identical and four times cheaper on the device.
  const float flf = (float)k.fl;
  *X = (z * ((float)x - k.cx)) / flf;                                 // h:35
  *Y = (z * ((float)y - k.cy)) / flf;                                 // h:36
  *Z = z;
}
// ... and the geometric Jacobian at the un-warped point, from p->X, p->Y, p->Z.
ODO_HD void point_jacobian(PointK* p, const LevelK& k) {
  const float flf = (float)k.fl;
  const float fx_z = flf / p->Z;                                      // :223
  const float xy = p->X * p->Y, xx = p->X * p->X, yy = p->Y * p->Y, zz = p->Z * p->Z;
  p->fx_z = fx_z;
  p->jw02 = (-fx_z * p->X) / p->Z;                                    // :232
  p->jw03 = (-fx_z * xy) / p->Z;
  p->jw04 = (float)(k.fl * (1.0 + (double)(xx / zz)));
  p->jw05 = -fx_z * p->Y;
  p->jw12 = (-fx_z * p->Y) / p->Z;                                    // :233
  p->jw13 = (float)(-k.fl * (1.0 + (double)(yy / zz)));
  p->jw14 = -p->jw03;  // = (fx_z * xy) / Z (ref :233): IEEE negation commutes with multiply and divide, bit for bit
  p->jw15 = fx_z * p->X;
}
ODO_HD PointK make_point(int x, int y, float inv_depth, float i1, const LevelK& k) {
  PointK p;
  point_xyz(x, y, inv_depth, k, &p.X, &p.Y, &p.Z);
  p.i1 = i1;
  point_jacobian(&p, k);
  return p;
}

// Warp by T (column-major 4x4) and project (ref: image_processing_global.h:42-51). Returns false when the point lies
// behind the camera (:43-46).
ODO_HD bool warp_point_uv(const PointK& p, const float* T, const LevelK& k, float* u, float* v) {
  const float t0 = ((T[0] * p.X + T[4] * p.Y) + T[8] * p.Z) + T[12];
  const float t1 = ((T[1] * p.X + T[5] * p.Y) + T[9] * p.Z) + T[13];
  const float t2 = ((T[2] * p.X + T[6] * p.Y) + T[10] * p.Z) + T[14];
  if (!(t2 > 0.0f)) return false;
  *u = (float)(k.fl * (double)t0 / (double)t2 + (double)k.cx);
  *v = (float)(k.fl * (double)t1 / (double)t2 + (double)k.cy);
  return true;
}
// ... then floor. Returns false when the point is skipped (ref: image_processing_global.h:54-59). ui/vi = floor(u), floor(v).
ODO_HD bool warp_point(const PointK& p, const float* T, const LevelK& k, int rows, int cols, int* ui, int* vi) {
  float u, v;
  if (!warp_point_uv(p, T, k, &u, &v)) return false;
  const float fu = floorf(u), fv = floorf(v);
  if (!(fu < (float)cols) || !(fv < (float)rows) || !(fu >= 0.0f) || !(fv >= 0.0f)) return false;
  *ui = (int)fu;
  *vi = (int)fv;
  return true;
}

// Residual + Jacobian row from the five I2 values around the floor-sampled pixel (centre, left, right, up, down; the
// neighbours index-clamped to the image): ref lm_optimizer.cpp:215-234, image_processing_global.h:62-69.
ODO_HD void residual_jacobian_taps(const PointK& p, float tc, float tl, float tr, float tu, float td, float* r, float J[6]) {
  const float gx = 0.5f * (tr - tl);
  const float gy = 0.5f * (td - tu);
  *r = tc - p.i1;
  J[0] = gx * p.fx_z + gy * 0.0f;
  J[1] = gx * 0.0f + gy * p.fx_z;
  J[2] = gx * p.jw02 + gy * p.jw12;
  J[3] = gx * p.jw03 + gy * p.jw13;
  J[4] = gx * p.jw04 + gy * p.jw14;
  J[5] = gx * p.jw05 + gy * p.jw15;
}
// The same, reading the taps. I2: level image of the current frame, row-major.
ODO_HD void residual_jacobian(const PointK& p, const float* I2, int rows, int cols, int ui, int vi, float* r, float J[6]) {
  const int px = (ui - 1 >= 0) ? ui - 1 : 0, nx = (ui + 1 < cols) ? ui + 1 : cols - 1;
  const int py = (vi - 1 >= 0) ? vi - 1 : 0, ny = (vi + 1 < rows) ? vi + 1 : rows - 1;
  const float* row = I2 + (size_t)vi * cols;
  residual_jacobian_taps(p, row[ui], row[px], row[nx], I2[(size_t)py * cols + ui], I2[(size_t)ny * cols + ui], r, J);
}

// ---------------------------------------------------------------------------------------------
// Bilinear sampling — a NON-PARITY option (BASELINE.json north_star names it; the reference samples at floor(u), floor(v),
// ref: lm_optimizer.cpp:208-217, so the default stays floor). The warped point (u, v) lies in the cell of the four pixels
// (x0, y0) .. (x0 + 1, y0 + 1), x0 = floor(u), y0 = floor(v), a = u - x0, b = v - y0; I2 is interpolated inside the cell and
// the image gradient is the derivative of that interpolant (so the objective is continuous in the pose and the Jacobian is
// its true derivative). The geometric Jacobian still comes from the un-warped point (:219-234), as in parity mode. A point
// whose cell is not entirely inside the image is skipped. Fixed operation order, fp32, one rounding per operation.
// ---------------------------------------------------------------------------------------------
ODO_HD bool bilinear_cell(float u, float v, int rows, int cols, int* x0, int* y0, float* a, float* b) {
  const float fu = floorf(u), fv = floorf(v);
  if (!(fu >= 0.0f) || !(fv >= 0.0f) || !(fu + 1.0f < (float)cols) || !(fv + 1.0f < (float)rows)) return false;
  *x0 = (int)fu; *y0 = (int)fv;
  *a = u - fu; *b = v - fv;
  return true;
}
ODO_HD void residual_jacobian_bilinear(const PointK& p, float i00, float i10, float i01, float i11, float a, float b, float* r,
                                       float J[6]) {
  const float dx0 = i10 - i00, dx1 = i11 - i01;      // horizontal differences on the upper / lower edge of the cell
  const float top = i00 + a * dx0, bot = i01 + a * dx1;
  const float val = top + b * (bot - top);
  const float gx = dx0 + b * (dx1 - dx0);
  const float dy0 = i01 - i00, dy1 = i11 - i10;      // vertical differences on the left / right edge
  const float gy = dy0 + a * (dy1 - dy0);
  *r = val - p.i1;
  J[0] = gx * p.fx_z + gy * 0.0f;
  J[1] = gx * 0.0f + gy * p.fx_z;
  J[2] = gx * p.jw02 + gy * p.jw12;
  J[3] = gx * p.jw03 + gy * p.jw13;
  J[4] = gx * p.jw04 + gy * p.jw14;
  J[5] = gx * p.jw05 + gy * p.jw15;
}
// Warp + sample + Jacobian row in the level's sampling mode. Returns false when the point produces no residual.
// kMayBilinear = false: a caller that is never launched with bilinear sampling on (the trackers' lean LM kernels) compiles the
// non-parity path out instead of branching around it.
template <bool kMayBilinear = true>
ODO_HD bool point_residual(const PointK& p, const float* T, const LevelK& k, const float* I2, int rows, int cols, float* r,
                           float J[6]) {
  if (kMayBilinear && k.bilinear) {
    float u, v, a, b;
    int x0, y0;
    if (!warp_point_uv(p, T, k, &u, &v) || !bilinear_cell(u, v, rows, cols, &x0, &y0, &a, &b)) return false;
    const float* r0 = I2 + (size_t)y0 * cols + x0;
    residual_jacobian_bilinear(p, r0[0], r0[1], r0[cols], r0[cols + 1], a, b, r, J);
    return true;
  }
  int ui, vi;
  if (!warp_point(p, T, k, rows, cols, &ui, &vi)) return false;
  residual_jacobian(p, I2, rows, cols, ui, vi, r, J);
  return true;
}
// The residual alone (t-distribution scale pass, ref: lm_optimizer.cpp:338-358).
ODO_HD bool point_residual_only(const PointK& p, const float* T, const LevelK& k, const float* I2, int rows, int cols, float* r) {
  if (k.bilinear) {
    float J[6];
    return point_residual(p, T, k, I2, rows, cols, r, J);
  }
  int ui, vi;
  if (!warp_point(p, T, k, rows, cols, &ui, &vi)) return false;
  *r = I2[(size_t)vi * cols + ui] - p.i1;
  return true;
}

// Robust weight (ref: lm_optimizer.cpp:249-262). scale_sqr only used by mode 2.
ODO_HD float robust_weight(float r, int robust, float huber_delta, float scale_sqr) {
  if (robust == 1) return (fabsf(r) <= huber_delta) ? 1.0f : huber_delta / fabsf(r);
  if (robust == 2) return (200.0f + 1.0f) / (200.0f + r * r / scale_sqr);
  return 1.0f;
}

// acc += one weighted row. Products of two fp32 values are exact in fp64, so fma == mul-then-add here.
ODO_HD void accumulate_row(double acc[ODO_NACC], float r, float w, const float J[6]) {
  double jw[6], Jd[6];
  for (int a = 0; a < 6; a++) { jw[a] = (double)(J[a] * w); Jd[a] = (double)J[a]; }
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) { acc[k] = fma(jw[a], Jd[b], acc[k]); k++; }
  const double rd = (double)r;
  for (int a = 0; a < 6; a++) acc[21 + a] = fma(jw[a], rd, acc[21 + a]);
  acc[27] = fma((double)(r * w), rd, acc[27]);
  acc[28] += 1.0;
}

// ---------------------------------------------------------------------------------------------
// sin/cos of an fp32 argument, evaluated in fp64 and rounded once (deterministic on host and device).
// ---------------------------------------------------------------------------------------------
ODO_HD void sincos_f(float xf, float* s_out, float* c_out) {
  const double x = (double)xf;
  const double kf = floor(x * 6.36619772367581382433e-01 + 0.5);
  const double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
  const double r2 = r * r;
  double ps = 1.0 / 355687428096000.0;
  ps = ps * r2 - 1.0 / 1307674368000.0;
  ps = ps * r2 + 1.0 / 6227020800.0;
  ps = ps * r2 - 1.0 / 39916800.0;
  ps = ps * r2 + 1.0 / 362880.0;
  ps = ps * r2 - 1.0 / 5040.0;
  ps = ps * r2 + 1.0 / 120.0;
  ps = ps * r2 - 1.0 / 6.0;
  ps = ps * r2 + 1.0;
  const double sr = ps * r;
  double pc = 1.0 / 6402373705728000.0;
  pc = pc * r2 - 1.0 / 20922789888000.0;
  pc = pc * r2 + 1.0 / 87178291200.0;
  pc = pc * r2 - 1.0 / 479001600.0;
  pc = pc * r2 + 1.0 / 3628800.0;
  pc = pc * r2 - 1.0 / 40320.0;
  pc = pc * r2 + 1.0 / 720.0;
  pc = pc * r2 - 1.0 / 24.0;
  pc = pc * r2 + 1.0 / 2.0;
  pc = pc * r2;
  const double cr = 1.0 - pc;
  const int q = (int)((long long)kf & 3);
  double s, c;
  if (q == 0) { s = sr; c = cr; }
  else if (q == 1) { s = cr; c = -sr; }
  else if (q == 2) { s = -sr; c = -cr; }
  else { s = -cr; c = sr; }
  *s_out = (float)s;
  *c_out = (float)c;
}

// ---------------------------------------------------------------------------------------------
// SE(3) as Sophus stores it: unit quaternion (x,y,z,w) + translation, fp32.
// ---------------------------------------------------------------------------------------------
struct Se3 {
  float qx, qy, qz, qw, tx, ty, tz;
};

// Eigen Quaternion(Matrix3) — trace-based, no renormalisation (so3.hpp:463-468). R row-major.
// The three "largest diagonal element" cases are spelled out so every index is a compile-time constant.
ODO_HD void rot_to_quat(const float R[9], Se3* o) {
  float t = (R[0] + R[4]) + R[8];
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    o->qw = 0.5f * t;
    t = 0.5f / t;
    o->qx = (R[7] - R[5]) * t;
    o->qy = (R[2] - R[6]) * t;
    o->qz = (R[3] - R[1]) * t;
    return;
  }
  int i = 0;
  if (R[4] > R[0]) i = 1;
  if (R[8] > (i == 1 ? R[4] : R[0])) i = 2;
  if (i == 0) {        // j = 1, k = 2
    t = sqrtf(((R[0] - R[4]) - R[8]) + 1.0f);
    o->qx = 0.5f * t;
    t = 0.5f / t;
    o->qw = (R[7] - R[5]) * t;
    o->qy = (R[3] + R[1]) * t;
    o->qz = (R[6] + R[2]) * t;
  } else if (i == 1) { // j = 2, k = 0
    t = sqrtf(((R[4] - R[8]) - R[0]) + 1.0f);
    o->qy = 0.5f * t;
    t = 0.5f / t;
    o->qw = (R[2] - R[6]) * t;
    o->qz = (R[7] + R[5]) * t;
    o->qx = (R[1] + R[3]) * t;
  } else {             // j = 0, k = 1
    t = sqrtf(((R[8] - R[0]) - R[4]) + 1.0f);
    o->qz = 0.5f * t;
    t = 0.5f / t;
    o->qw = (R[3] - R[1]) * t;
    o->qx = (R[2] + R[6]) * t;
    o->qy = (R[5] + R[7]) * t;
  }
}

// Eigen Quaternion::toRotationMatrix (so3.hpp:302-304). R row-major.
ODO_HD void quat_to_rot(const Se3& s, float R[9]) {
  const float tx = 2.0f * s.qx, ty = 2.0f * s.qy, tz = 2.0f * s.qz;
  const float twx = tx * s.qw, twy = ty * s.qw, twz = tz * s.qw;
  const float txx = tx * s.qx, txy = ty * s.qx, txz = tz * s.qx;
  const float tyy = ty * s.qy, tyz = tz * s.qy, tzz = tz * s.qz;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}

ODO_HD void se3_from_colmajor(const float M[16], Se3* o) {  // se3.hpp:495-502
  float R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = M[j * 4 + i];
  rot_to_quat(R, o);
  o->tx = M[12]; o->ty = M[13]; o->tz = M[14];
}
ODO_HD void se3_to_colmajor(const Se3& s, float M[16]) {     // se3.hpp:272-278
  float R[9];
  quat_to_rot(s, R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[j * 4 + i] = R[i * 3 + j];
  M[3] = 0.0f; M[7] = 0.0f; M[11] = 0.0f;
  M[12] = s.tx; M[13] = s.ty; M[14] = s.tz; M[15] = 1.0f;
}

// SE3::exp, a = [upsilon; omega] (se3.hpp:765-786, so3.hpp:577-611).
ODO_HD void se3_exp(const float a[6], Se3* o) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = (ox * ox + oy * oy) + oz * oz;
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  float imag, real;
  if (theta < 1e-5f) {
    const float theta_po4 = theta_sq * theta_sq;
    imag = (0.5f - (float)(1.0 / 48.0) * theta_sq) + (float)(1.0 / 3840.0) * theta_po4;
    real = (1.0f - (float)(1.0 / 8.0) * theta_sq) + (float)(1.0 / 384.0) * theta_po4;
  } else {
    float sh, ch;
    sincos_f(half_theta, &sh, &ch);
    imag = sh / theta;
    real = ch;
  }
  o->qw = real; o->qx = imag * ox; o->qy = imag * oy; o->qz = imag * oz;
  const float Om[9] = {0.0f, -oz, oy, oz, 0.0f, -ox, -oy, ox, 0.0f};
  float Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Om2[i * 3 + j] = (Om[i * 3 + 0] * Om[0 * 3 + j] + Om[i * 3 + 1] * Om[1 * 3 + j]) + Om[i * 3 + 2] * Om[2 * 3 + j];
  float V[9];
  if (theta < 1e-5f) {
    quat_to_rot(*o, V);
  } else {
    float st, ct;
    sincos_f(theta, &st, &ct);
    const float tsq = theta * theta;
    const float ca = (1.0f - ct) / tsq;
    const float cb = (theta - st) / (tsq * theta);
    for (int i = 0; i < 9; i++) {
      const float id = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
      V[i] = (id + ca * Om[i]) + cb * Om2[i];
    }
  }
  o->tx = (V[0] * a[0] + V[1] * a[1]) + V[2] * a[2];
  o->ty = (V[3] * a[0] + V[4] * a[1]) + V[5] * a[2];
  o->tz = (V[6] * a[0] + V[7] * a[1]) + V[8] * a[2];
}

// inc = SE3(delta.matrix() * cur.matrix()) (lm_optimizer.cpp:152-153): plain 4x4 fp32 product, k ascending.
// se3_left_update_mat: the same with cur.matrix() handed in (C, column-major, as se3_to_colmajor writes it) — the persistent LM
// kernels keep the matrix of the current estimate from the time it was computed as the candidate's (lm_state_machine_hot).
ODO_HD void se3_left_update_mat(const Se3& delta, const float C[16], Se3* out) {
  float D[16], M[16];
  se3_to_colmajor(delta, D);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      M[j * 4 + i] = ((D[0 * 4 + i] * C[j * 4 + 0] + D[1 * 4 + i] * C[j * 4 + 1]) + D[2 * 4 + i] * C[j * 4 + 2]) +
                     D[3 * 4 + i] * C[j * 4 + 3];
  se3_from_colmajor(M, out);
}
ODO_HD void se3_left_update(const Se3& delta, const Se3& cur, Se3* out) {
  float C[16];
  se3_to_colmajor(cur, C);
  se3_left_update_mat(delta, C, out);
}

// Damped normal equations A = JtWJ + lambda*diag(JtWJ), b = -JtWr (lm_optimizer.cpp:145-151), solved in fp64
// ("fp64 LDL^T or QR of the fp64-accumulated system"). A is symmetric positive semi-definite
// (a weighted Gram matrix with a scaled-up diagonal), so elimination runs down the diagonal without row exchanges —
// the LDL^T order of operations. A zero pivot (a Jacobian column that is identically zero) leaves that step
// component at zero. Back substitution multiplies by the reciprocal pivots. The step is rounded to fp32.
// Written with compile-time indices only, so on the device the 6x7 system lives in registers.
ODO_HD void solve_damped(const double acc[ODO_NACC], float lambda, float delta[6]) {
  double A[6][7];
  {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int b = a; b < 6; b++) { A[a][b] = acc[k]; A[b][a] = acc[k]; k++; }
    }
  }
#pragma unroll
  for (int a = 0; a < 6; a++) {
    A[a][a] = A[a][a] + (double)lambda * A[a][a];
    A[a][6] = -acc[21 + a];
  }
  bool ok[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    ok[c] = (fabs(A[c][c]) > 0.0);
    if (ok[c]) {
#pragma unroll
      for (int i = c + 1; i < 6; i++) {
        const double f = A[i][c] / A[c][c];
#pragma unroll
        for (int j = c; j < 7; j++) A[i][j] = A[i][j] - f * A[c][j];
      }
    }
  }
  double rinv[6];
#pragma unroll
  for (int c = 0; c < 6; c++) rinv[c] = ok[c] ? 1.0 / A[c][c] : 0.0;
  double xs[6];
#pragma unroll
  for (int c = 5; c >= 0; c--) {
    double s = A[c][6];
#pragma unroll
    for (int j = c + 1; j < 6; j++) s = s - A[c][j] * xs[j];
    xs[c] = ok[c] ? s * rinv[c] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < 6; c++) delta[c] = (float)xs[c];
}

// ---------------------------------------------------------------------------------------------
// LM state machine of one Solve (lm_optimizer.cpp:73-160), advanced one evaluation at a time.
// ---------------------------------------------------------------------------------------------
struct LmState {  // <= 64 dwords: the update kernel copies it with one wavefront
  Se3 cur, inc, last;
  float T[16];        // inc.matrix(), column-major: the pose the next evaluation uses
  float lambda, err_last, err_now;
  int level;          // level being optimised
  int iter;           // iter_count inside the level
  int active;         // 1 while the level's while-loop is running
  int status;         // 0 ok, -1 failed (N == 0)
  int n_evals;        // evaluations consumed so far
  int stop_reason;    // last stop: 1 precision, 2 lambda, 3 max iters
  int iters_level[8]; // evaluations per level (diagnostic; the reference never fills its own stats)
  float delta[6];     // last solved step
  // fused pipeline: an evaluation whose partial sums have been written but not consumed yet
  int pending;        // 1 = partials of an evaluation at s->T wait in the partial buffer
  int pending_nblk;   // number of partial rows that evaluation wrote
  int max_iters;      // iteration budget of the level being optimised (max_iterations_[level])
  int finished;       // 1 once every level has been optimised (or the Solve failed): nothing left to launch
};

ODO_HD void lm_begin_solve(LmState* s, const float init_colmajor[16]) {
  se3_from_colmajor(init_colmajor, &s->cur);  // :76
  s->inc = s->cur;                            // :77
  s->last = s->cur;                           // :78
  s->status = 0;
  s->n_evals = 0;
  s->active = 0;
  s->stop_reason = 0;
  s->pending = 0;
  s->pending_nblk = 0;
  s->max_iters = 0;
  s->finished = 0;
  s->err_now = 0.0f;
  for (int i = 0; i < 8; i++) s->iters_level[i] = 0;
  for (int i = 0; i < 6; i++) s->delta[i] = 0.0f;
}

ODO_HD void lm_begin_level(LmState* s, int level, float lambda0, int max_iters) {
  s->level = level;
  s->iter = 0;
  s->err_last = 1e+10f;        // :111
  s->lambda = lambda0;         // :113
  s->inc = s->cur;             // :115
  s->active = (s->status == 0 && max_iters > 0) ? 1 : 0;
  s->max_iters = max_iters;
  se3_to_colmajor(s->inc, s->T);
}

// Consume the accumulators of one evaluation at pose s->T (lm_optimizer.cpp:123-154), in two halves so the
// device can run the 6x6 solve across the lanes of a wavefront in between.
// lm_decide: error, accept / reject, lambda rule, stop tests (:129-143). Returns true when a step must be solved.
ODO_HD bool lm_decide(LmState* s, const double acc[ODO_NACC], float precision) {
  s->n_evals++;
  {  // iters_level[level]++ with compile-time indices, so a register-resident state never needs dynamic indexing
    const int li = s->level & 7;
#pragma unroll
    for (int i = 0; i < 8; i++) s->iters_level[i] += (i == li) ? 1 : 0;
  }
  if (!(acc[28] > 0.0)) {      // :244-248 -> :123-126
    s->status = -1;
    s->active = 0;
    return false;
  }
  const float err_now = (float)(acc[27] / acc[28]);  // :129
  s->err_now = err_now;
  if (err_now > s->err_last) {                       // :131
    s->lambda = s->lambda * 5.0f;
    if (s->lambda > 1e+5f) { s->active = 0; s->stop_reason = 2; return false; }
    s->cur = s->last;
  } else {
    s->cur = s->inc;
    s->last = s->cur;
    const float err_diff = err_now / s->err_last;
    if (err_diff > precision) { s->active = 0; s->stop_reason = 1; return false; }
    s->err_last = err_now;
    s->lambda = fmaxf(s->lambda / 5.0f, 1e-5f);
  }
  return true;
}
// lm_apply_step: inc = SE3(exp(delta).matrix() * cur.matrix()), iter++ (:152-154) and the loop test (:117).
ODO_HD void lm_apply_step(LmState* s, int max_iters) {
  Se3 d;
  se3_exp(s->delta, &d);                             // :152
  se3_left_update(d, s->cur, &s->inc);               // :153
  se3_to_colmajor(s->inc, s->T);
  s->iter++;                                         // :154
  if (!(max_iters > s->iter)) { s->active = 0; s->stop_reason = 3; }
}
ODO_HD void lm_consume(LmState* s, const double acc[ODO_NACC], float precision, int max_iters) {
  if (!lm_decide(s, acc, precision)) return;
  solve_damped(acc, s->lambda, s->delta);            // :145-151
  lm_apply_step(s, max_iters);
}

// ---------------------------------------------------------------------------------------------
// The runner's keyframe test on a pose_to_keyframe (ref: run_odometry_kitti_offline.cpp:144-145,253-258): Sophus SO3::angleX/Y/Z of
// the R -> q -> R round trip (third_party/Sophus/sophus/so3.hpp:127-154), |angles| and |translation| weighted and summed in order.
// One source for the trackers' host code and for the device-side guard of a chained Solve (the two atan2f may differ in the last
// bit; the guard is only a hint — the host's decision counts, see lm_chain_begin).
// ---------------------------------------------------------------------------------------------
ODO_HD void motion_angles3(const float* T, float ang[3]) {
  Se3 s;
  se3_from_colmajor(T, &s);
  float R[9];
  quat_to_rot(s, R);
  ang[0] = atan2f(R[7] - R[5], R[4] + R[8]);
  ang[1] = atan2f(R[2] - R[6], R[0] + R[8]);
  ang[2] = atan2f(R[3] - R[1], R[0] + R[4]);
}
ODO_HD float motion_magnitude(const float* T, const float w[6]) {
  float ang[3];
  motion_angles3(T, ang);                                                              // :253
  const float mot[6] = {fabsf(ang[0]), fabsf(ang[1]), fabsf(ang[2]), fabsf(T[12]), fabsf(T[13]), fabsf(T[14])};
  float mag = 0.0f;
  for (int i = 0; i < 6; i++) mag += mot[i] * w[i];                                     // :257
  return mag;
}

// A CHEAP bound on the same quantity for the device-side guard of a chained Solve — a hint, not the decision (the host's
// motion_magnitude decides): angles from the pose's own rotation block (no R -> q -> R round trip: 1e-7) through a polynomial
// arctangent (|error| < 2.1e-4 rad: 2e-4 of magnitude at the runner's weights). The guard lets the chained Solve run only if the bound stays below the threshold by kMotionSlack.
ODO_HD float atan2_approx(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = (mx > 0.0f) ? mn / mx : 0.0f;
  const float s2 = a * a;
  float r = ((-0.0464964749f * s2 + 0.15931422f) * s2 - 0.327622764f) * s2 * a + a;   // atan(a), a in [0, 1]: max error 2.1e-4
  if (ay > ax) r = 1.57079637f - r;
  if (x < 0.0f) r = 3.14159274f - r;
  return (y < 0.0f) ? -r : r;
}
ODO_HD float motion_magnitude_approx(const float* T, const float w[6]) {
  // T column-major: R[i][j] = T[j * 4 + i]; angleX = atan2(R21 - R12, R11 + R22), ... as motion_angles3
  const float ax = atan2_approx(T[1 * 4 + 2] - T[2 * 4 + 1], T[1 * 4 + 1] + T[2 * 4 + 2]);
  const float ay = atan2_approx(T[2 * 4 + 0] - T[0 * 4 + 2], T[0 * 4 + 0] + T[2 * 4 + 2]);
  const float az = atan2_approx(T[0 * 4 + 1] - T[1 * 4 + 0], T[0 * 4 + 0] + T[1 * 4 + 1]);
  return fabsf(ax) * w[0] + fabsf(ay) * w[1] + fabsf(az) * w[2] + fabsf(T[12]) * w[3] + fabsf(T[13]) * w[4] + fabsf(T[14]) * w[5];
}
#define ODO_MOTION_SLACK 1e-3f

// ---------------------------------------------------------------------------------------------
// The inverse-depth LM's driver (ref: src/depth_estimate.cpp:92-96,141,150-161,167): the state every block of
// depth_lm_step_kernel carries from launch to launch, and the rule one evaluation's error is judged by. Pinned to the
// reference's own lines through the host build (tests/test_ref_pin.py, emu_depth_lm_schedule).
// ---------------------------------------------------------------------------------------------
struct DepthLmState {
  float lambda, err_last, err_now;
  int iter;
  int done;
};
ODO_HD void depth_lm_begin(DepthLmState* st, float lambda0, int max_iters) {
  st->lambda = lambda0;                              // :92
  st->err_last = 1e+10f;                             // :94
  st->err_now = 0.0f;                                // :95
  st->iter = 0;                                      // :93
  st->done = (max_iters > 0) ? 0 : 1;                // :141
}
// Returns the mode: 0 reject + carry on (current = pre, :153), 1 accept + carry on (current = tmp, pre = current, :155-156),
// 2 reject + break (:152), 3 accept + break (:158). Lambda and err_last are updated as the reference's lines do.
ODO_HD int depth_lm_decide(DepthLmState* st, float err_now, float precision) {
  st->err_now = err_now;
  if (err_now > st->err_last) {                      // :150
    st->lambda = st->lambda * 10.0f;                 // :151
    return (st->lambda > 1e+5f) ? 2 : 0;             // :152
  }
  const float err_diff = err_now / st->err_last;     // :157
  if (err_diff > precision) return 3;                // :158
  st->err_last = err_now;                            // :159
  st->lambda = fmaxf(st->lambda / 10.0f, 1e-7f);     // :160
  return 1;
}
// iter_count++ (:167) and the loop test (:141) after a step that carries on; a break ends the loop as it stands
ODO_HD void depth_lm_advance(DepthLmState* st, int mode, int max_iters) {
  if (mode == 2 || mode == 3) { st->done = 1; return; }
  st->iter++;
  if (!(max_iters > st->iter)) st->done = 1;
}

// ---------------------------------------------------------------------------------------------
// Depth estimator pieces (ref: src/depth_estimate.cpp).
// ---------------------------------------------------------------------------------------------
// 8-tap SSD with the AVX hadd tree of ComputeSsdPattern8Sse (depth_estimate.cpp:435-453).
// Lane order s0=(0,+2) s1=(-1,+1) s2=(+2,0) s3=(0,0) s4=(-2,0) s5=(+1,-1) s6=(-1,-1) s7=(0,-2) as (dx,dy).
ODO_HD float ssd8_tree(const float L[8], const float R[8]) {
  float s[8];
  for (int i = 0; i < 8; i++) { const float d = L[i] - R[i]; s[i] = d * d; }
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

}  // namespace odo
