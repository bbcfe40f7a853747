/*
 * oracle/odo_oracle.c — CPU restatement of the reference's photometric-LM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker (or as the timed CPU baseline), never as the thing shipped or measured as the GPU path.
 *
 * PARITY UNPINNED, except: the disparity scan, its SSD tree and GetCxLevel (orc_disparity_scan / ssd8 / cx_level), the camera
 * pyramid's intrinsic rule, and the accept / reject / lambda / stop schedules of both LM loops (lm_rule / orc_lm_schedule,
 * depth_lm_rule / orc_depth_lm_schedule) are checked bit for bit against the reference's own lines
 * (src/depth_estimate.cpp:92-96,141,150-161,167-168,380-395,435-453; include/image_processing_global.h:22-28;
 * src/camera.cpp:61-65; src/lm_optimizer.cpp:110-115,117,131-143,154-155), compiled from /root/reference by
 * oracle/make_ref_fixtures.py: tests/golden/ssd_ref.npz, cx_level_ref.npz, lm_schedule_ref.npz, depth_lm_schedule_ref.npz,
 * tests/test_ref_pin.py.
 * For everything else: the reference (WangYuTum/odometry) cannot be built here (no Eigen, no OpenCV,
 * empty nanogui submodule, std::sqrtf) and none of its tests pins a numeric result, so this file is a
 * from-scratch restatement that follows the cited reference lines op for op.  Where the reference
 * delegates to OpenCV / Eigen (blur, pyrDown, dense products, the 6x6 solve, quaternion<->matrix) the
 * published definition of that routine is restated; the choices made where the library leaves the
 * association order open are written next to each function.  All fp32 arithmetic is evaluated without
 * FMA contraction (compile with -ffp-contract=off), sums over pixels are accumulated in fp64.
 *
 * Citations "ref:" are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LEVELS 8

/* ------------------------------------------------------------------------------------------------
 * Types shared with the ctypes wrapper (oracle/oracle.py).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float f0;  /* level-0 focal length  (ref: include/image_processing_global.h:35  718.856f)   */
  float cx0; /* level-0 principal x   (ref: include/image_processing_global.h:35  607.1928)   */
  float cy0; /* level-0 principal y   (ref: include/image_processing_global.h:36  185.2157)   */
} orc_intr;

typedef struct {
  float lambda;                    /* ref: src/lm_optimizer.cpp:26  */
  float precision;                 /* ref: src/lm_optimizer.cpp:27  */
  int n_levels;
  int max_iters[ORC_MAX_LEVELS];   /* indexed by level, ref: src/lm_optimizer.cpp:117 */
  int robust;                      /* 0 none, 1 Huber, 2 t-dist; ref: src/lm_optimizer.cpp:249-262 */
  float huber_delta;
  orc_intr K;
} orc_lm_params;

/* One record per evaluation of the hot loop (ref: src/lm_optimizer.cpp:117-155). */
typedef struct {
  int level, iter, n_res, accepted; /* accepted: 1 good step, 0 rejected */
  float err, lambda_after;
  int stop;                         /* 0 continue, 1 precision break, 2 lambda break */
  float delta[6];                   /* step solved after this evaluation (zeros when the loop broke) */
  float pose[16];                   /* column-major inc_estimate.matrix() used for this evaluation */
} orc_lm_trace;

typedef struct {
  float grad_th, ssd_th, photo_th, min_depth, max_depth, lambda, huber_delta, precision;
  int max_iters, boundary;
  float baseline;
  int max_residuals;
  float f0;             /* ref: src/depth_estimate.cpp:214,273 hard-code 718.856f */
  int max_disparity;    /* 0 = reference range [boundary, x) (ref: src/depth_estimate.cpp:382) */
  int any_size;         /* 0 = keep the 376x1241 guard (ref: src/depth_estimate.cpp:46-49) */
} orc_depth_params;

typedef struct {
  int n_selected, n_matched, n_valid, iters;
  float cost;
} orc_depth_stats;

/* ------------------------------------------------------------------------------------------------
 * Deterministic sin/cos for fp32 arguments.
 * The reference calls std::sin/std::cos on float (Sophus, ref: third_party/Sophus/sophus/so3.hpp:600-602,
 * se3.hpp:780-782), i.e. libm sinf/cosf.  To make the result independent of the libm in use (glibc
 * here, ocml on the GPU) the value is computed in fp64 (Cody-Waite reduction + Taylor series, error
 * < 1e-17) and rounded once to fp32 — i.e. the correctly rounded sinf/cosf except in ~1e-9 of cases.
 * ---------------------------------------------------------------------------------------------- */
static double orc_sin_kernel(double r) {
  const double r2 = r * r;
  double p = 1.0 / 355687428096000.0;            /* 1/17! */
  p = p * r2 - 1.0 / 1307674368000.0;            /* 1/15! */
  p = p * r2 + 1.0 / 6227020800.0;               /* 1/13! */
  p = p * r2 - 1.0 / 39916800.0;                 /* 1/11! */
  p = p * r2 + 1.0 / 362880.0;                   /* 1/9!  */
  p = p * r2 - 1.0 / 5040.0;                     /* 1/7!  */
  p = p * r2 + 1.0 / 120.0;                      /* 1/5!  */
  p = p * r2 - 1.0 / 6.0;                        /* 1/3!  */
  p = p * r2 + 1.0;
  return p * r;
}
static double orc_cos_kernel(double r) {
  const double r2 = r * r;
  double p = 1.0 / 6402373705728000.0;           /* 1/18! */
  p = p * r2 - 1.0 / 20922789888000.0;           /* 1/16! */
  p = p * r2 + 1.0 / 87178291200.0;              /* 1/14! */
  p = p * r2 - 1.0 / 479001600.0;                /* 1/12! */
  p = p * r2 + 1.0 / 3628800.0;                  /* 1/10! */
  p = p * r2 - 1.0 / 40320.0;                    /* 1/8!  */
  p = p * r2 + 1.0 / 720.0;                      /* 1/6!  */
  p = p * r2 - 1.0 / 24.0;                       /* 1/4!  */
  p = p * r2 + 1.0 / 2.0;                        /* 1/2!  */
  p = p * r2;
  return 1.0 - p;
}
static void orc_sincos_d(double x, double* s, double* c) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
  const double pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi */
  const double kf = floor(x * two_over_pi + 0.5);
  const double r = (x - kf * pio2_hi) - kf * pio2_lo;
  const long long k = (long long)kf;
  const double sr = orc_sin_kernel(r), cr = orc_cos_kernel(r);
  switch ((int)(k & 3)) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
  }
}
float orc_sinf(float x) { double s, c; orc_sincos_d((double)x, &s, &c); return (float)s; }
float orc_cosf(float x) { double s, c; orc_sincos_d((double)x, &s, &c); return (float)c; }

/* ------------------------------------------------------------------------------------------------
 * Pyramids
 * ---------------------------------------------------------------------------------------------- */
static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * n - 2 - i;
  }
  return i;
}

/* cv::GaussianBlur(src, dst, Size(3,3), 0) on CV_32F, BORDER_REFLECT_101
 * (ref: src/image_processing_global.cpp:30, src/depth_estimate.cpp:256-257).
 * Separable {1/4, 1/2, 1/4}: row pass then column pass, each mid*0.5f + (prev+next)*0.25f. */
int orc_blur3x3(const float* src, int rows, int cols, float* dst) {
  float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
  if (!tmp) return -1;
  for (int y = 0; y < rows; y++) {
    const float* s = src + (size_t)y * cols;
    float* t = tmp + (size_t)y * cols;
    for (int x = 0; x < cols; x++) {
      const float p = s[reflect101(x - 1, cols)], n = s[reflect101(x + 1, cols)];
      t[x] = s[x] * 0.5f + (p + n) * 0.25f;
    }
  }
  for (int y = 0; y < rows; y++) {
    const float* tp = tmp + (size_t)reflect101(y - 1, rows) * cols;
    const float* tn = tmp + (size_t)reflect101(y + 1, rows) * cols;
    const float* tm = tmp + (size_t)y * cols;
    float* d = dst + (size_t)y * cols;
    for (int x = 0; x < cols; x++) d[x] = tm[x] * 0.5f + (tp[x] + tn[x]) * 0.25f;
  }
  free(tmp);
  return 0;
}

/* cv::pyrDown(src, dst, Size(cols/2, rows/2)) on CV_32F, BORDER_REFLECT_101
 * (ref: src/image_processing_global.cpp:38,46).
 * Horizontal: row[x] = s[2x]*6 + (s[2x-1]+s[2x+1])*4 + s[2x-2] + s[2x+2];
 * vertical:   dst[x] = (r2[x]*6 + (r1[x]+r3[x])*4 + r0[x] + r4[x]) * (1/256). */
int orc_pyrdown(const float* src, int rows, int cols, float* dst) {
  const int dr = rows / 2, dc = cols / 2;
  float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * dc);
  if (!tmp) return -1;
  for (int y = 0; y < rows; y++) {
    const float* s = src + (size_t)y * cols;
    float* t = tmp + (size_t)y * dc;
    for (int x = 0; x < dc; x++) {
      const float s0 = s[reflect101(2 * x - 2, cols)], s1 = s[reflect101(2 * x - 1, cols)];
      const float s2 = s[reflect101(2 * x, cols)], s3 = s[reflect101(2 * x + 1, cols)];
      const float s4 = s[reflect101(2 * x + 2, cols)];
      t[x] = ((s2 * 6.0f + (s1 + s3) * 4.0f) + s0) + s4;
    }
  }
  for (int y = 0; y < dr; y++) {
    const float* r0 = tmp + (size_t)reflect101(2 * y - 2, rows) * dc;
    const float* r1 = tmp + (size_t)reflect101(2 * y - 1, rows) * dc;
    const float* r2 = tmp + (size_t)reflect101(2 * y, rows) * dc;
    const float* r3 = tmp + (size_t)reflect101(2 * y + 1, rows) * dc;
    const float* r4 = tmp + (size_t)reflect101(2 * y + 2, rows) * dc;
    float* d = dst + (size_t)y * dc;
    for (int x = 0; x < dc; x++)
      d[x] = (((r2[x] * 6.0f + (r1[x] + r3[x]) * 4.0f) + r0[x]) + r4[x]) * (1.0f / 256.0f);
  }
  free(tmp);
  return 0;
}

void orc_level_dims(int rows, int cols, int level, int* r, int* c) {
  for (int l = 0; l < level; l++) { rows /= 2; cols /= 2; }
  *r = rows; *c = cols;
}
/* number of floats of an n_levels pyramid stored level after level */
long orc_pyramid_size(int rows, int cols, int n_levels) {
  long n = 0;
  for (int l = 0; l < n_levels; l++) { n += (long)rows * cols; rows /= 2; cols /= 2; }
  return n;
}
long orc_level_offset(int rows, int cols, int level) { return orc_pyramid_size(rows, cols, level); }

/* GaussianImagePyramidNaive (ref: src/image_processing_global.cpp:12-56).
 * L0 = blur(in) if smooth else copy; L1 = pyrDown(in)  -- from the UNSMOOTHED input (:38);
 * Lk = pyrDown(L(k-1)). out holds the levels back to back. */
int orc_image_pyramid(const float* img, int rows, int cols, int n_levels, int smooth, float* out) {
  if (n_levels < 1 || n_levels > ORC_MAX_LEVELS) return -1;
  if (smooth) { if (orc_blur3x3(img, rows, cols, out)) return -1; }
  else memcpy(out, img, sizeof(float) * (size_t)rows * cols);
  const float* prev = img;
  int pr = rows, pc = cols;
  float* dst = out + (size_t)rows * cols;
  for (int l = 1; l < n_levels; l++) {
    if (orc_pyrdown(prev, pr, pc, dst)) return -1;
    prev = dst; pr /= 2; pc /= 2;
    dst += (size_t)pr * pc;
  }
  return 0;
}

/* cv::medianBlur(src, dst, 3) on CV_32F (ref: src/image_processing_global.cpp:77; OpenCV is not vendored: restated from its
 * documented definition — the median of the 3x3 neighbourhood, BORDER_REPLICATE). A median is a selection, not arithmetic: any
 * correct implementation returns the same bits (inputs without NaN). */
int orc_median3x3(const float* src, int rows, int cols, float* dst) {
  if (rows < 1 || cols < 1) return -1;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      float v[9];
      int k = 0;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          int yy = y + dy, xx = x + dx;
          yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
          xx = xx < 0 ? 0 : (xx >= cols ? cols - 1 : xx);
          v[k++] = src[(size_t)yy * cols + xx];
        }
      for (int i = 1; i < 9; i++) {   /* insertion sort of nine */
        const float t = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; }
        v[j + 1] = t;
      }
      dst[(size_t)y * cols + x] = v[4];
    }
  return 0;
}

/* MedianDepthPyramidNaive (ref: src/image_processing_global.cpp:58-113): L0 = copy, or the 3x3 median when `smooth`
 * (:76-80; no caller of the reference passes true), Lk(y,x) = L(k-1)(2y+1, 2x+1). */
int orc_depth_pyramid_ex(const float* dep, int rows, int cols, int n_levels, int smooth, float* out);
int orc_depth_pyramid(const float* dep, int rows, int cols, int n_levels, float* out) {
  return orc_depth_pyramid_ex(dep, rows, cols, n_levels, 0, out);
}
int orc_depth_pyramid_ex(const float* dep, int rows, int cols, int n_levels, int smooth, float* out) {
  if (n_levels < 1 || n_levels > ORC_MAX_LEVELS) return -1;
  if (smooth) { if (orc_median3x3(dep, rows, cols, out)) return -1; }
  else memcpy(out, dep, sizeof(float) * (size_t)rows * cols);
  const float* prev = out;
  int pr = rows, pc = cols;
  float* dst = out + (size_t)rows * cols;
  for (int l = 1; l < n_levels; l++) {
    const int r = pr / 2, c = pc / 2;
    for (int y = 0; y < r; y++)
      for (int x = 0; x < c; x++) dst[(size_t)y * c + x] = prev[(size_t)(2 * y + 1) * pc + (2 * x + 1)];
    prev = dst; pr = r; pc = c;
    dst += (size_t)r * c;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SE(3) pieces (Sophus / Eigen semantics, fp32)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float qx, qy, qz, qw; float t[3]; } orc_se3;

/* Eigen Quaternion(Matrix3), used by Sophus SO3(R)
 * (ref: third_party/Sophus/sophus/so3.hpp:463-468). No renormalisation. R row-major 3x3. */
static void rot_to_quat(const float R[9], orc_se3* o) {
  float q[4]; /* x y z w */
  float t = (R[0] + R[4]) + R[8];
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(((R[i * 3 + i] - R[j * 3 + j]) - R[k * 3 + k]) + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  o->qx = q[0]; o->qy = q[1]; o->qz = q[2]; o->qw = q[3];
}

/* Eigen Quaternion::toRotationMatrix, used by SO3::matrix()
 * (ref: third_party/Sophus/sophus/so3.hpp:302-304). R row-major 3x3. */
static void quat_to_rot(const orc_se3* s, float R[9]) {
  const float x = s->qx, y = s->qy, z = s->qz, w = s->qw;
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}

/* SE3(Matrix4) ctor (ref: third_party/Sophus/sophus/se3.hpp:495-502). M column-major 4x4. */
static void se3_from_colmajor(const float M[16], orc_se3* o) {
  float R[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = M[j * 4 + i];
  rot_to_quat(R, o);
  o->t[0] = M[12]; o->t[1] = M[13]; o->t[2] = M[14];
}
/* SE3::matrix() (ref: third_party/Sophus/sophus/se3.hpp:272-278). Column-major 4x4 out. */
static void se3_to_colmajor(const orc_se3* s, float M[16]) {
  float R[9];
  quat_to_rot(s, R);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[j * 4 + i] = R[i * 3 + j];
  M[3] = 0.0f; M[7] = 0.0f; M[11] = 0.0f;
  M[12] = s->t[0]; M[13] = s->t[1]; M[14] = s->t[2]; M[15] = 1.0f;
}

/* SE3::exp (ref: third_party/Sophus/sophus/se3.hpp:765-786) with SO3::expAndTheta
 * (ref: so3.hpp:577-611). a = [upsilon(3); omega(3)]. */
static void se3_exp(const float a[6], orc_se3* o) {
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
    imag = orc_sinf(half_theta) / theta;
    real = orc_cosf(half_theta);
  }
  o->qw = real; o->qx = imag * ox; o->qy = imag * oy; o->qz = imag * oz;
  /* Omega = hat(omega), Omega_sq = Omega*Omega (3x3 products, k ascending) */
  const float Om[9] = {0.0f, -oz, oy, oz, 0.0f, -ox, -oy, ox, 0.0f};
  float Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Om2[i * 3 + j] = (Om[i * 3 + 0] * Om[0 * 3 + j] + Om[i * 3 + 1] * Om[1 * 3 + j]) + Om[i * 3 + 2] * Om[2 * 3 + j];
  float V[9];
  if (theta < 1e-5f) {
    quat_to_rot(o, V);
  } else {
    const float tsq = theta * theta;
    const float ca = (1.0f - orc_cosf(theta)) / tsq;
    const float cb = (theta - orc_sinf(theta)) / (tsq * theta);
    for (int i = 0; i < 9; i++) {
      const float id = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
      V[i] = (id + ca * Om[i]) + cb * Om2[i];
    }
  }
  for (int i = 0; i < 3; i++) o->t[i] = (V[i * 3 + 0] * a[0] + V[i * 3 + 1] * a[1]) + V[i * 3 + 2] * a[2];
}
void orc_se3_exp(const float a[6], float M_colmajor[16]) {
  orc_se3 s; se3_exp(a, &s); se3_to_colmajor(&s, M_colmajor);
}
/* SE3(M) then .matrix(): the R -> q -> R round trip every pose goes through. */
void orc_se3_roundtrip(const float M_in[16], float M_out[16]) {
  orc_se3 s; se3_from_colmajor(M_in, &s); se3_to_colmajor(&s, M_out);
}

/* inc = SE3(delta.matrix() * cur.matrix()) (ref: src/lm_optimizer.cpp:152-153).
 * 4x4 fp32 product, k ascending, no FMA. */
static void se3_left_update(const orc_se3* delta, const orc_se3* cur, orc_se3* out) {
  float D[16], C[16], M[16];
  se3_to_colmajor(delta, D);
  se3_to_colmajor(cur, C);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      M[j * 4 + i] = ((D[0 * 4 + i] * C[j * 4 + 0] + D[1 * 4 + i] * C[j * 4 + 1]) + D[2 * 4 + i] * C[j * 4 + 2]) +
                     D[3 * 4 + i] * C[j * 4 + 3];
  se3_from_colmajor(M, out);
}
void orc_se3_left_update(const float delta6[6], const float cur_colmajor[16], float out_colmajor[16]) {
  orc_se3 d, c, o;
  se3_exp(delta6, &d);
  se3_from_colmajor(cur_colmajor, &c);
  se3_left_update(&d, &c, &o);
  se3_to_colmajor(&o, out_colmajor);
}

/* ------------------------------------------------------------------------------------------------
 * Per-pixel chain: reproject, warp, gradient, residual, Jacobian row
 * ---------------------------------------------------------------------------------------------- */
/* GetCxLevel (ref: include/image_processing_global.h:22-28) */
static float cx_level(float c, int level) {
  float v = c;
  for (int i = 0; i < level; i++) v = (v + 0.5f) / 2.0f + 0.5f;
  return v;
}
float orc_cx_level(float c, int level) { return cx_level(c, level); }

/* One row of ComputeResidualJacobianNaive (ref: src/lm_optimizer.cpp:190-237 with
 * include/image_processing_global.h:31-69). T column-major 4x4. Returns 1 and fills r, J[6] when the
 * pixel contributes, 0 when it is skipped. Mixed float/double promotions follow SURVEY appendix A10:
 * std::pow(2.0f, level) is double, so every expression it touches is evaluated in double and rounded
 * to float on assignment. */
/* Sampling of the current image: 0 = floor (the reference, parity mode); 1 = bilinear — the oracle of the NON-PARITY option
 * odo_lm_set_sampling(ODO_SAMPLE_BILINEAR) (BASELINE.json north_star names bilinear sampling; the reference has no such mode,
 * so this definition — interpolate I2 in the 2x2 cell around (u, v), gradient = derivative of the interpolant, skip points whose
 * cell leaves the image — is this repository's, shared op for op with odo_math.h:residual_jacobian_bilinear). */
static int g_sampling = 0;
void orc_set_sampling(int mode) { g_sampling = mode; }

static int pixel_row(const float* I1, const float* I2, const float* D1, int rows, int cols, int x, int y,
                     const float T[16], double fl, float cxl, float cyl, float* r_out, float J[6]) {
  const float d = D1[(size_t)y * cols + x];
  if (fabsf(d - 0.0f) < 0.01f) return 0;                                  /* :193 */
  const float z = 1.0f / d;                                               /* :198 */
  /* ReprojectToCameraFrame h:35-38 */
  const float X = (float)((double)(z * ((float)x - cxl)) / fl);
  const float Y = (float)((double)(z * ((float)y - cyl)) / fl);
  const float Z = z;
  /* WarpPixel h:43 tmp = T * P (P.w = 1) */
  const float t0 = ((T[0] * X + T[4] * Y) + T[8] * Z) + T[12] * 1.0f;
  const float t1 = ((T[1] * X + T[5] * Y) + T[9] * Z) + T[13] * 1.0f;
  const float t2 = ((T[2] * X + T[6] * Y) + T[10] * Z) + T[14] * 1.0f;
  if (!(t2 > 0.0f)) return 0;                                             /* h:45 (NaN also skipped) */
  const float u = (float)(fl * (double)t0 / (double)t2 + (double)cxl);    /* h:50 */
  const float v = (float)(fl * (double)t1 / (double)t2 + (double)cyl);    /* h:51 */
  const float fu = floorf(u), fv = floorf(v);
  float gx, gy;
  if (g_sampling == 1) {
    if (!(fu >= 0.0f) || !(fv >= 0.0f) || !(fu + 1.0f < (float)cols) || !(fv + 1.0f < (float)rows)) return 0;
    const int x0 = (int)fu, y0 = (int)fv;
    const float a = u - fu, b = v - fv;
    const float* c0 = I2 + (size_t)y0 * cols + x0;
    const float i00 = c0[0], i10 = c0[1], i01 = c0[cols], i11 = c0[cols + 1];
    const float dx0 = i10 - i00, dx1 = i11 - i01;
    const float top = i00 + a * dx0, bot = i01 + a * dx1;
    const float val = top + b * (bot - top);
    gx = dx0 + b * (dx1 - dx0);
    const float dy0 = i01 - i00, dy1 = i11 - i10;
    gy = dy0 + a * (dy1 - dy0);
    *r_out = val - I1[(size_t)y * cols + x];
  } else {
    if (!(fu < (float)cols) || !(fv < (float)rows) || !(fu >= 0.0f) || !(fv >= 0.0f)) return 0; /* h:54-56 */
    const int ui = (int)fu, vi = (int)fv;                                   /* :208-209 */
    /* ComputePixelGradient h:62-69 (index clamping) */
    const int px = (ui - 1 >= 0) ? ui - 1 : 0, nx = (ui + 1 < cols) ? ui + 1 : cols - 1;
    const int py = (vi - 1 >= 0) ? vi - 1 : 0, ny = (vi + 1 < rows) ? vi + 1 : rows - 1;
    gx = 0.5f * (I2[(size_t)vi * cols + nx] - I2[(size_t)vi * cols + px]);
    gy = 0.5f * (I2[(size_t)ny * cols + ui] - I2[(size_t)py * cols + ui]);
    *r_out = I2[(size_t)vi * cols + ui] - I1[(size_t)y * cols + x];         /* :217 */
  }
  /* geometric Jacobian at the UN-warped point :223-233 */
  const float fx_z = (float)(fl / (double)Z);
  const float xy = X * Y, xx = X * X, yy = Y * Y, zz = Z * Z;
  const float jw02 = (-fx_z * X) / Z;
  const float jw03 = (-fx_z * xy) / Z;
  const float jw04 = (float)(fl * (1.0 + (double)(xx / zz)));
  const float jw05 = -fx_z * Y;
  const float jw12 = (-fx_z * Y) / Z;
  const float jw13 = (float)(-fl * (1.0 + (double)(yy / zz)));
  const float jw14 = (fx_z * xy) / Z;
  const float jw15 = fx_z * X;
  /* jaco.row = grad * jw :234 (1x2 * 2x6, g0*jw0 + g1*jw1) */
  J[0] = gx * fx_z + gy * 0.0f;
  J[1] = gx * 0.0f + gy * fx_z;
  J[2] = gx * jw02 + gy * jw12;
  J[3] = gx * jw03 + gy * jw13;
  J[4] = gx * jw04 + gy * jw14;
  J[5] = gx * jw05 + gy * jw15;
  return 1;
}

/* ComputeScaleNaive (ref: src/lm_optimizer.cpp:338-358); the per-pass sum is accumulated in fp64. */
static long g_tdist_passes = 0, g_tdist_calls = 0;   /* diagnostic: passes over the residuals / calls (orc_tdist_counters) */
void orc_tdist_counters(long* passes, long* calls, int reset) {
  if (passes) *passes = g_tdist_passes;
  if (calls) *calls = g_tdist_calls;
  if (reset) g_tdist_passes = g_tdist_calls = 0;
}
static float tdist_scale(const float* r, int n) {
  float init_sigma = 5.0f, cur = 5.0f;
  const float vee = 200.0f;
  int guard = 0;
  g_tdist_calls++;
  do {
    g_tdist_passes++;
    init_sigma = cur;
    const float sigma_sqr = cur * cur;
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
      const float e2 = r[i] * r[i];
      sum += (double)(e2 * (1.0f + vee) / (vee + e2 / sigma_sqr));
    }
    cur = sqrtf((float)(sum / (double)n));
  } while (fabsf(cur - init_sigma) >= 1e-3f && ++guard < 1000);
  return cur;
}

/* Accumulators: acc[0..20] upper triangle of J^T W J row-major (00,01,..,05,11,..,55),
 * acc[21..26] J^T W r, acc[27] sum w r^2, acc[28] N.
 * Terms: jw_a = fl32(J_a * w) (JtW = J^T * W, ref :145), A_ab += jw_a * J_b, b_a += jw_a * r (ref :146,149),
 * err += fl32(r*w) * r (ref :129); products and sums in fp64.
 * Optional dumps of the first `dump_cap` rows: r, w, J (row-major 6). Returns 0, or -1 when N == 0
 * (ref :244-248). */
/* ---- timing-only variant in the reference's own shape (SURVEY 8d "faithful restatement") ----
 * ComputeResidualJacobianNaive materialises J (N x 6), the weights and the residuals for the whole frame
 * (ref: src/lm_optimizer.cpp:187-188 resize to rows*cols, :242-243 conservativeResize copy), re-evaluates
 * std::pow(2.0f, level) and GetCxLevel at every use site for every pixel WITH VALID DEPTH (ref: :193-196 skips the others first;
 * include/image_processing_global.h:22-28,35-36,50-51; src/lm_optimizer.cpp:223-233), and OptimizeCameraPose then forms JtW (6 x N temporary), JtW*J, JtW*r and r^T W r as
 * separate fp32 passes (ref: :129,145-149). Same per-pixel arithmetic as pixel_row; the sums are fp32, so its
 * trajectory is not bit-comparable with the oracle proper — it exists only so that the CPU time of the reference's
 * shape can be reported next to the fused restatement. Selected with orc_set_reference_shape(1). */
static int g_reference_shape = 0;   /* 1: sequential fp32 sums; 2: eight strided fp32 partial sums + a pairwise combine (the
                                      * association of an AVX2 reduction, which is what Eigen emits under -mavx2) */
void orc_set_reference_shape(int on) { g_reference_shape = on; }
static float dot_f32(const float* a, size_t sa, const float* b, size_t sb, const float* c, int n) {
  /* sum_i a[i*sa] * b[i*sb] (* c[i] when c != NULL), fp32, association per g_reference_shape */
  if (g_reference_shape != 2) {
    float s = 0.0f;
    for (int i = 0; i < n; i++) s += c ? a[(size_t)i * sa] * c[i] * b[(size_t)i * sb] : a[(size_t)i * sa] * b[(size_t)i * sb];
    return s;
  }
  float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; k++) {
      const float p = c ? a[(size_t)(i + k) * sa] * c[i + k] * b[(size_t)(i + k) * sb] : a[(size_t)(i + k) * sa] * b[(size_t)(i + k) * sb];
      lane[k] += p;
    }
  float s = ((lane[0] + lane[4]) + (lane[2] + lane[6])) + ((lane[1] + lane[5]) + (lane[3] + lane[7]));
  for (; i < n; i++) s += c ? a[(size_t)i * sa] * c[i] * b[(size_t)i * sb] : a[(size_t)i * sa] * b[(size_t)i * sb];
  return s;
}
/* The reference evaluates std::pow(2.0f, level) and GetCxLevel at every use site, for every pixel that PASSES the depth test
 * (ref: src/lm_optimizer.cpp:193-196 `continue`s before any of them): ReprojectToCameraFrame 2 pow + 2 GetCxLevel
 * (include/image_processing_global.h:35-36), WarpPixel 2 + 2 after its Z' > 0 test (h:45,50-51), the Jacobian 4 pow once the warp
 * succeeded (src/lm_optimizer.cpp:223-224,232-233). Does a reference build really call pow there? Yes: GCC 11 -O3 -march=haswell
 * -mavx2 (ref: CMakeLists.txt:19) keeps `call pow@PLT` inside the pixel loop for exactly this pattern (pow may set errno under the
 * default -fmath-errno, so it is neither hoisted nor merged) — checked with a stand-alone loop of the same shape. The volatile
 * base below only stops THIS file's compiler from folding pow(2.0, level) because `level` is visible as a small constant range
 * after inlining; it adds no call the reference does not make. */
static volatile float g_two = 2.0f;
#define REF_FL() ((double)K->f0 / pow((double)g_two, (double)level))

/* pixel_row with the reference's call pattern: same arithmetic, same results bit for bit (fl, cxl, cyl are the same values
 * wherever they are recomputed), the reference's amount of work per pixel. Floor sampling only. */
static int pixel_row_reference_shape(const float* I1, const float* I2, const float* D1, int rows, int cols, int x, int y,
                                     const float T[16], const orc_intr* K, int level, float* r_out, float J[6]) {
  const float d = D1[(size_t)y * cols + x];
  if (fabsf(d - 0.0f) < 0.01f) return 0;                                  /* :193-196: nothing below runs for invalid depth */
  const float z = 1.0f / d;                                               /* :198 */
  const float X = (float)((double)(z * ((float)x - cx_level(K->cx0, level))) / REF_FL());   /* h:35 */
  const float Y = (float)((double)(z * ((float)y - cx_level(K->cy0, level))) / REF_FL());   /* h:36 */
  const float Z = z;
  const float t0 = ((T[0] * X + T[4] * Y) + T[8] * Z) + T[12] * 1.0f;     /* h:43 */
  const float t1 = ((T[1] * X + T[5] * Y) + T[9] * Z) + T[13] * 1.0f;
  const float t2 = ((T[2] * X + T[6] * Y) + T[10] * Z) + T[14] * 1.0f;
  if (!(t2 > 0.0f)) return 0;                                             /* h:45 */
  const float u = (float)(REF_FL() * (double)t0 / (double)t2 + (double)cx_level(K->cx0, level));   /* h:50 */
  const float v = (float)(REF_FL() * (double)t1 / (double)t2 + (double)cx_level(K->cy0, level));   /* h:51 */
  const float fu = floorf(u), fv = floorf(v);
  if (!(fu < (float)cols) || !(fv < (float)rows) || !(fu >= 0.0f) || !(fv >= 0.0f)) return 0;      /* h:54-56 */
  const int ui = (int)fu, vi = (int)fv;                                   /* :208-209 */
  const int px = (ui - 1 >= 0) ? ui - 1 : 0, nx = (ui + 1 < cols) ? ui + 1 : cols - 1;             /* h:62-69 */
  const int py = (vi - 1 >= 0) ? vi - 1 : 0, ny = (vi + 1 < rows) ? vi + 1 : rows - 1;
  const float gx = 0.5f * (I2[(size_t)vi * cols + nx] - I2[(size_t)vi * cols + px]);
  const float gy = 0.5f * (I2[(size_t)ny * cols + ui] - I2[(size_t)py * cols + ui]);
  *r_out = I2[(size_t)vi * cols + ui] - I1[(size_t)y * cols + x];         /* :217 */
  const float fx_z = (float)(REF_FL() / (double)Z);                       /* :223 */
  const float fy_z = (float)(REF_FL() / (double)Z);                       /* :224 (fy = fx: the same value, evaluated again) */
  const float xy = X * Y, xx = X * X, yy = Y * Y, zz = Z * Z;
  const float jw02 = (-fx_z * X) / Z;
  const float jw03 = (-fx_z * xy) / Z;
  const float jw04 = (float)(REF_FL() * (1.0 + (double)(xx / zz)));       /* :232 */
  const float jw05 = -fx_z * Y;
  const float jw12 = (-fy_z * Y) / Z;
  const float jw13 = (float)(-REF_FL() * (1.0 + (double)(yy / zz)));      /* :233 */
  const float jw14 = (fy_z * xy) / Z;
  const float jw15 = fy_z * X;
  J[0] = gx * fx_z + gy * 0.0f;                                           /* :234 */
  J[1] = gx * 0.0f + gy * fy_z;
  J[2] = gx * jw02 + gy * jw12;
  J[3] = gx * jw03 + gy * jw13;
  J[4] = gx * jw04 + gy * jw14;
  J[5] = gx * jw05 + gy * jw15;
  return 1;
}

static int lm_accumulate_reference_shape(const float* I1, const float* I2, const float* D1, int rows, int cols, int level,
                                         const float T[16], int robust, float huber_delta, const orc_intr* K,
                                         double acc[29]) {
  const size_t cap = (size_t)rows * cols;
  float* J = (float*)malloc(sizeof(float) * cap * 6);   /* :187 jaco.resize(rows*cols, 6) */
  float* W = (float*)malloc(sizeof(float) * cap);       /* :188 */
  float* r = (float*)malloc(sizeof(float) * cap);
  if (!J || !W || !r) { free(J); free(W); free(r); return -1; }
  int n = 0;
  for (int y = 4; y < rows - 4; y++)
    for (int x = 4; x < cols - 4; x++)
      if (pixel_row_reference_shape(I1, I2, D1, rows, cols, x, y, T, K, level, &r[n], &J[(size_t)n * 6])) n++;
  for (int i = 0; i < 29; i++) acc[i] = 0.0;
  if (n == 0) { free(J); free(W); free(r); return -1; }
  /* :242-243 conservativeResize(n, 6): Eigen reallocates and copies */
  float* J2 = (float*)malloc(sizeof(float) * (size_t)n * 6);
  float* r2 = (float*)malloc(sizeof(float) * (size_t)n);
  memcpy(J2, J, sizeof(float) * (size_t)n * 6);
  memcpy(r2, r, sizeof(float) * (size_t)n);
  free(J); free(r);
  /* :249-262 weights, a separate pass */
  float scale_sqr = 1.0f;
  if (robust == 2) { const float sg = tdist_scale(r2, n); scale_sqr = sg * sg; }
  for (int i = 0; i < n; i++) {
    const float ri = r2[i];
    float w = 1.0f;
    if (robust == 1) w = (fabsf(ri) <= huber_delta) ? 1.0f : huber_delta / fabsf(ri);
    else if (robust == 2) w = (200.0f + 1.0f) / (200.0f + ri * ri / scale_sqr);
    W[i] = w;
  }
  /* :129 err = r^T W r / N */
  const float err = dot_f32(r2, 1, r2, 1, W, n);
  /* :145 JtW = J^T * W.asDiagonal(), a 6 x N temporary */
  float* JtW = (float*)malloc(sizeof(float) * (size_t)n * 6);
  for (int a = 0; a < 6; a++)
    for (int i = 0; i < n; i++) JtW[(size_t)a * n + i] = J2[(size_t)i * 6 + a] * W[i];
  /* :146 JtWJ = JtW * J (all 36 entries), :149 b = -JtW * r */
  float A[6][6], b[6];
  for (int a = 0; a < 6; a++) {
    for (int c = 0; c < 6; c++) A[a][c] = dot_f32(JtW + (size_t)a * n, 1, J2 + c, 6, NULL, n);
    b[a] = dot_f32(JtW + (size_t)a * n, 1, r2, 1, NULL, n);
  }
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int c = a; c < 6; c++) acc[k++] = (double)A[a][c];
  for (int a = 0; a < 6; a++) acc[21 + a] = (double)b[a];
  acc[27] = (double)err;
  acc[28] = (double)n;
  free(J2); free(r2); free(W); free(JtW);
  return 0;
}

int orc_lm_accumulate(const float* I1, const float* I2, const float* D1, int rows, int cols, int level,
                      const float T[16], int robust, float huber_delta, const orc_intr* K, double acc[29],
                      float* sigma_out, int dump_cap, float* dump_r, float* dump_w, float* dump_J) {
  if (g_reference_shape && !dump_cap && !sigma_out)
    return lm_accumulate_reference_shape(I1, I2, D1, rows, cols, level, T, robust, huber_delta, K, acc);
  const double fl = (double)K->f0 / pow(2.0, (double)level);
  const float cxl = cx_level(K->cx0, level), cyl = cx_level(K->cy0, level);
  const size_t cap = (size_t)rows * cols;
  float* r = (float*)malloc(sizeof(float) * cap);
  float* J = (float*)malloc(sizeof(float) * cap * 6);
  if (!r || !J) { free(r); free(J); return -1; }
  int n = 0;
  for (int y = 4; y < rows - 4; y++)
    for (int x = 4; x < cols - 4; x++)
      if (pixel_row(I1, I2, D1, rows, cols, x, y, T, fl, cxl, cyl, &r[n], &J[(size_t)n * 6])) n++;
  for (int i = 0; i < 29; i++) acc[i] = 0.0;
  if (sigma_out) *sigma_out = 0.0f;
  if (n == 0) { free(r); free(J); return -1; }
  float scale_sqr = 1.0f;
  if (robust == 2) {
    const float s = tdist_scale(r, n);
    if (sigma_out) *sigma_out = s;
    scale_sqr = s * s;
  }
  for (int i = 0; i < n; i++) {
    const float ri = r[i];
    float w = 1.0f;
    if (robust == 1) w = (fabsf(ri) <= huber_delta) ? 1.0f : huber_delta / fabsf(ri);  /* :254 */
    else if (robust == 2) w = (200.0f + 1.0f) / (200.0f + ri * ri / scale_sqr);        /* :260 */
    const float* Ji = &J[(size_t)i * 6];
    float jw[6];
    for (int a = 0; a < 6; a++) jw[a] = Ji[a] * w;
    int k = 0;
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++) acc[k++] += (double)jw[a] * (double)Ji[b];
    for (int a = 0; a < 6; a++) acc[21 + a] += (double)jw[a] * (double)ri;
    acc[27] += (double)(ri * w) * (double)ri;
    if (i < dump_cap) {
      if (dump_r) dump_r[i] = ri;
      if (dump_w) dump_w[i] = w;
      if (dump_J) memcpy(&dump_J[(size_t)i * 6], Ji, sizeof(float) * 6);
    }
  }
  acc[28] = (double)n;
  free(r); free(J);
  return 0;
}

/* Damped normal equations (ref: src/lm_optimizer.cpp:145-151): A = JtWJ + lambda*diag(JtWJ), b = -JtWr.
 * The reference solves in fp32 with colPivHouseholderQr; here ("fp64 LDL^T or QR of the
 * fp64-accumulated system") the symmetric positive semi-definite system is eliminated down its diagonal in fp64
 * without row exchanges (the LDL^T order of operations); a zero pivot (an identically zero Jacobian column) yields a
 * zero step component; back substitution multiplies by the reciprocal pivot; the step is rounded to fp32. */
static void solve_damped(const double acc[29], float lambda, float delta[6]) {
  double A[6][7];
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) { A[a][b] = acc[k]; A[b][a] = acc[k]; k++; }
  for (int a = 0; a < 6; a++) {
    A[a][a] = A[a][a] + (double)lambda * A[a][a];
    A[a][6] = -acc[21 + a];
  }
  int piv_ok[6];
  for (int c = 0; c < 6; c++) {
    piv_ok[c] = fabs(A[c][c]) > 0.0;
    if (!piv_ok[c]) continue;
    for (int i = c + 1; i < 6; i++) {
      const double f = A[i][c] / A[c][c];
      for (int j = c; j < 7; j++) A[i][j] = A[i][j] - f * A[c][j];
    }
  }
  double xs[6];
  for (int c = 5; c >= 0; c--) {
    if (!piv_ok[c]) { xs[c] = 0.0; continue; }
    const double rinv = 1.0 / A[c][c];
    double s = A[c][6];
    for (int j = c + 1; j < 6; j++) s = s - A[c][j] * xs[j];
    xs[c] = s * rinv;
  }
  for (int c = 0; c < 6; c++) delta[c] = (float)xs[c];
}
/* Sensitivity study only (oracle/sensitivity.py): the reference's own solver shape — fp32 system, fp32 column-pivoted
 * Householder QR (Eigen's colPivHouseholderQr().solve(), ref: src/lm_optimizer.cpp:150-151; Eigen is not vendored: restated
 * from the textbook algorithm — pivot on the largest remaining column norm, reflect, rank threshold eps * 6 * |max pivot|,
 * back-substitute the leading rank x rank block, undo the permutation). Selected with orc_set_solver(1). */
static int g_solver = 0;
void orc_set_solver(int s) { g_solver = s; }
static void solve_damped_qr_f32(const double acc[29], float lambda, float delta[6]) {
  float A[6][6], b[6];
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int c = a; c < 6; c++) { A[a][c] = (float)acc[k]; A[c][a] = (float)acc[k]; k++; }
  for (int a = 0; a < 6; a++) { A[a][a] = A[a][a] + lambda * A[a][a]; b[a] = -(float)acc[21 + a]; }   /* :147-150 */
  int perm[6] = {0, 1, 2, 3, 4, 5};
  float maxpiv = 0.0f;
  int rank = 6;
  for (int c = 0; c < 6; c++) {
    int best = c;
    float bn = -1.0f;
    for (int j = c; j < 6; j++) {
      float nn = 0.0f;
      for (int i = c; i < 6; i++) nn += A[i][j] * A[i][j];
      if (nn > bn) { bn = nn; best = j; }
    }
    if (best != c) {
      for (int i = 0; i < 6; i++) { const float t = A[i][c]; A[i][c] = A[i][best]; A[i][best] = t; }
      const int t = perm[c]; perm[c] = perm[best]; perm[best] = t;
    }
    float tail = 0.0f;
    for (int i = c + 1; i < 6; i++) tail += A[i][c] * A[i][c];
    const float c0 = A[c][c];
    float beta, tau;
    if (tail == 0.0f) { beta = c0; tau = 0.0f; }
    else {
      beta = sqrtf(c0 * c0 + tail);
      if (c0 >= 0.0f) beta = -beta;
      for (int i = c + 1; i < 6; i++) A[i][c] = A[i][c] / (c0 - beta);   /* essential part of the reflector */
      tau = (beta - c0) / beta;
    }
    /* apply H = I - tau v v^T (v = [1; essential]) to the trailing columns and to b */
    if (tau != 0.0f) {
      for (int j = c + 1; j < 6; j++) {
        float d = A[c][j];
        for (int i = c + 1; i < 6; i++) d += A[i][c] * A[i][j];
        d *= tau;
        A[c][j] -= d;
        for (int i = c + 1; i < 6; i++) A[i][j] -= d * A[i][c];
      }
      float d = b[c];
      for (int i = c + 1; i < 6; i++) d += A[i][c] * b[i];
      d *= tau;
      b[c] -= d;
      for (int i = c + 1; i < 6; i++) b[i] -= d * A[i][c];
    }
    A[c][c] = beta;
    if (fabsf(beta) > maxpiv) maxpiv = fabsf(beta);
  }
  const float thr = 1.1920929e-07f * 6.0f * maxpiv;
  rank = 0;
  for (int c = 0; c < 6; c++) if (fabsf(A[c][c]) > thr) rank++; else break;
  float xs[6] = {0, 0, 0, 0, 0, 0};
  for (int c = rank - 1; c >= 0; c--) {
    float sum = b[c];
    for (int j = c + 1; j < rank; j++) sum -= A[c][j] * xs[j];
    xs[c] = sum / A[c][c];
  }
  for (int c = 0; c < 6; c++) delta[perm[c]] = xs[c];
}
void orc_solve_damped(const double acc[29], float lambda, float delta[6]) {
  if (g_solver == 1) solve_damped_qr_f32(acc, lambda, delta); else solve_damped(acc, lambda, delta);
}

/* LevenbergMarquardtOptimizer::OptimizeCameraPose (ref: src/lm_optimizer.cpp:73-160).
 * img1/dep1 = keyframe pyramids, img2 = current pyramid, each stored level after level.
 * init / out: column-major 4x4. trace may be NULL. Returns 0, or -1 (then out = pseudo-identity whose
 * (3,3) is 0, ref :48-52,60-65). */
/* The accept / reject / lambda / stop rule of one evaluation of the LM loop (ref: src/lm_optimizer.cpp:131-143). Returns the
 * stop code (0 carry on, 1 precision break :140, 2 lambda break :134); *accepted tells which branch ran. Pinned to the reference's
 * own lines by tests/test_ref_pin.py (oracle/make_ref_fixtures.py compiles :110-115,117,131-143,154-155 as they stand). */
static int lm_rule(float err_now, float* err_last, float* lambda, float precision, int* accepted) {
  if (err_now > *err_last) {                                              /* :131 */
    *accepted = 0;
    *lambda = *lambda * 5.0f;                                             /* :133 */
    return (*lambda > 1e+5f) ? 2 : 0;                                     /* :134 */
  }
  *accepted = 1;
  const float err_diff = err_now / *err_last;                             /* :139 */
  if (err_diff > precision) return 1;                                     /* :140 */
  *err_last = err_now;                                                    /* :141 */
  *lambda = fmaxf(*lambda / 5.0f, 1e-5f);                                 /* :142 */
  return 0;
}
/* The LM driver of one level replayed on a given sequence of errors, estimates as tags (0 = the level's starting pose, k + 1 = the
 * pose solved after evaluation k): rec = 5 ints per evaluation {lambda bits, err_last bits, current tag, last tag, broke}. Same
 * loop as orc_lm_solve's (ref: :110-155); the twin of ref_lm_schedule in oracle/make_ref_fixtures.py. */
int orc_lm_schedule(const float* errs, int n_errs, float lambda0, float precision, int max_iters, int* rec, int* final_current) {
  int cur = 0, last = 0, inc = 0, iter = 0, k = 0;
  float err_last = 1e+10f, lambda = lambda0;
  inc = cur;
  while (max_iters > iter) {
    if (k >= n_errs) break;
    int accepted;
    const int stop = lm_rule(errs[k], &err_last, &lambda, precision, &accepted);
    if (accepted) { cur = inc; last = cur; }
    else if (!stop) cur = last;
    memcpy(&rec[5 * k + 0], &lambda, sizeof(int));
    memcpy(&rec[5 * k + 1], &err_last, sizeof(int));
    rec[5 * k + 2] = cur; rec[5 * k + 3] = last; rec[5 * k + 4] = stop ? 1 : 0;
    k++;
    if (stop) break;
    inc = k;
    iter++;
  }
  *final_current = cur;
  return k;
}

int orc_lm_solve(const float* img1, const float* dep1, const float* img2, int rows, int cols,
                 const orc_lm_params* p, const float init[16], float out[16], orc_lm_trace* trace,
                 int trace_cap, int* n_trace) {
  orc_se3 cur, inc, last, delta;
  se3_from_colmajor(init, &cur);                                          /* :76 */
  inc = cur;                                                              /* :77 */
  last = cur;                                                             /* :78 */
  int nt = 0;
  int status = 0;
  for (int l = p->n_levels - 1; l >= 0 && status == 0; l--) {            /* :92 */
    int r, c;
    orc_level_dims(rows, cols, l, &r, &c);
    const long off = orc_level_offset(rows, cols, l);
    int iter = 0;
    float err_last = 1e+10f, err_now = 0.0f;
    float lambda = p->lambda;                                             /* :113 */
    inc = cur;                                                            /* :115 */
    while (p->max_iters[l] > iter) {                                      /* :117 */
      float T[16];
      double acc[29];
      se3_to_colmajor(&inc, T);
      if (orc_lm_accumulate(img1 + off, img2 + off, dep1 + off, r, c, l, T, p->robust, p->huber_delta, &p->K, acc,
                            NULL, 0, NULL, NULL, NULL)) { status = -1; break; }  /* :123-126 */
      err_now = (float)(acc[27] / acc[28]);                               /* :129 */
      orc_lm_trace* tr = (trace && nt < trace_cap) ? &trace[nt] : NULL;
      if (tr) {
        memset(tr, 0, sizeof(*tr));
        tr->level = l; tr->iter = iter; tr->n_res = (int)acc[28]; tr->err = err_now;
        memcpy(tr->pose, T, sizeof(T));
      }
      nt++;
      int accepted;
      const int stop = lm_rule(err_now, &err_last, &lambda, p->precision, &accepted);   /* :131-143 */
      if (accepted) { cur = inc; last = cur; }                            /* :137-138 (before the precision break of :140) */
      else if (!stop) cur = last;                                         /* :135 (after the lambda break of :134) */
      if (tr) { tr->accepted = accepted; tr->stop = stop; tr->lambda_after = lambda; }
      if (stop) break;
      float dv[6];
      orc_solve_damped(acc, lambda, dv);                                  /* :145-151 */
      se3_exp(dv, &delta);                                                /* :152 */
      se3_left_update(&delta, &cur, &inc);                                /* :153 */
      if (tr) memcpy(tr->delta, dv, sizeof(dv));
      iter++;
    }
  }
  if (n_trace) *n_trace = nt;
  if (status) {
    memset(out, 0, sizeof(float) * 16);
    out[0] = 1.0f; out[5] = 1.0f; out[10] = 1.0f;                         /* (3,3) stays 0, ref :48-52 */
    return -1;
  }
  se3_to_colmajor(&cur, out);                                             /* :158 */
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Depth estimator
 * ---------------------------------------------------------------------------------------------- */
static int cmp_float(const void* a, const void* b) {
  const float fa = *(const float*)a, fb = *(const float*)b;
  return (fa > fb) - (fa < fb);
}

/* ComputeSsdPattern8Sse (ref: src/depth_estimate.cpp:435-453): AVX lane order and hadd tree
 * ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)) with s0=(0,+2) s1=(-1,+1) s2=(+2,0) s3=(0,0) s4=(-2,0)
 * s5=(+1,-1) s6=(-1,-1) s7=(0,-2) as (dx,dy). L[8] holds the left taps in that lane order. */
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("fp-contract=off")))  /* the reference's SSD is AVX intrinsics: no contraction there under any flag */
#endif
static float ssd8(const float L[8], const float* img, int cols, int x, int y) {
  const float* pp = img + (size_t)(y - 2) * cols;
  const float* p = img + (size_t)(y - 1) * cols;
  const float* c = img + (size_t)y * cols;
  const float* n = img + (size_t)(y + 1) * cols;
  const float* nn = img + (size_t)(y + 2) * cols;
  const float R[8] = {nn[x], n[x - 1], c[x + 2], c[x], c[x - 2], p[x + 1], p[x - 1], pp[x]};
  float s[8];
  for (int i = 0; i < 8; i++) { const float dlt = L[i] - R[i]; s[i] = dlt * dlt; }
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
float orc_ssd8_tree(const float s[8]) { return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])); }
/* one candidate: left taps in lane order against the right image at (x, y) — tests/test_ref_pin.py */
float orc_ssd8_at(const float L[8], const float* img, int cols, int x, int y) { return ssd8(L, img, cols, x, y); }

/* The epipolar scan of DisparityDepthEstimate on its own (ref: src/depth_estimate.cpp:345-398): for every marked pixel of the
 * BLURRED pair the 8-tap SSD of every candidate column, first strict minimum, threshold, disparity and inverse depth. Optional
 * outputs best / match (smallest SSD and its column per marked pixel) exist for tests/test_ref_pin.py, which compares this
 * function with the reference's own lines compiled by oracle/make_ref_fixtures.py. */
void orc_disparity_scan(const float* L, const float* R, const uint8_t* val, int rows, int cols, int bnd, float ssd_th, float fx,
                        float baseline, int max_disparity, float* disp, float* dep, float* best_out, int* match_out,
                        int* n_sel_out, int* n_match_out) {
  int n_sel = 0, n_match = 0;
  for (int y = bnd; y < rows - bnd; y++)                                  /* :346 */
    for (int x = bnd; x < cols - bnd; x++) {                              /* :352 */
      if (val[(size_t)y * cols + x] == 0) continue;
      n_sel++;
      const float* pp = L + (size_t)(y - 2) * cols; const float* pr = L + (size_t)(y - 1) * cols;
      const float* cr = L + (size_t)y * cols; const float* nr = L + (size_t)(y + 1) * cols;
      const float* nn = L + (size_t)(y + 2) * cols;
      const float Lp[8] = {nn[x], nr[x - 1], cr[x + 2], cr[x], cr[x - 2], pr[x + 1], pr[x - 1], pp[x]}; /* :380-381 */
      float best = 1e+10f;
      int match = -1;
      int lo = bnd;
      if (max_disparity > 0 && x - max_disparity > lo) lo = x - max_disparity;
      for (int rx = lo; rx < x; rx++) {                                   /* :382 */
        const float s = ssd8(Lp, R, cols, rx, y);
        if (s < best) { best = s; match = rx; }                           /* :385-386 strict < */
      }
      if (best_out) best_out[(size_t)y * cols + x] = best;
      if (match_out) match_out[(size_t)y * cols + x] = match;
      if (best > ssd_th) continue;                                        /* :388 */
      const float dsp = (float)abs(x - match);                            /* :391 */
      disp[(size_t)y * cols + x] = dsp;
      dep[(size_t)y * cols + x] = dsp / (fx * baseline);                  /* :394 */
      n_match++;
    }
  if (n_sel_out) *n_sel_out = n_sel;
  if (n_match_out) *n_match_out = n_match;
}

/* DisparityDepthEstimate (ref: src/depth_estimate.cpp:244-401). val must be zeroed by the caller;
 * disp/dep are zero-filled here. bl/br receive the blurred images if non-NULL. */
static int disparity_depth(const float* left, const float* right, int rows, int cols, const orc_depth_params* p,
                           uint8_t* val, float* disp, float* dep, orc_depth_stats* st) {
  const size_t npx = (size_t)rows * cols;
  float* L = (float*)malloc(sizeof(float) * npx);
  float* R = (float*)malloc(sizeof(float) * npx);
  float* grad = (float*)malloc(sizeof(float) * npx);
  if (!L || !R || !grad) { free(L); free(R); free(grad); return -1; }
  orc_blur3x3(left, rows, cols, L);                                       /* :256 */
  orc_blur3x3(right, rows, cols, R);                                      /* :257 */
  const int bnd = p->boundary;
  const int block_w = (cols - bnd * 2) / 32, block_h = (rows - bnd * 2) / 16;  /* :303-304 */
  const int bsz = block_w * block_h;
  float* bg = (float*)malloc(sizeof(float) * (size_t)(bsz > 0 ? bsz : 1));
  int n_sel = 0, n_match = 0;
  for (int b = 0; b < 16 * 32 && bsz > 0; b++) {                          /* :310 */
    const int sy = bnd + (b / 32) * block_h, sx = bnd + (b % 32) * block_w;
    int cnt = 0;
    for (int y = sy; y < sy + block_h; y++)
      for (int x = sx; x < sx + block_w; x++) {
        const float gx = 0.5f * (L[(size_t)y * cols + x + 1] - L[(size_t)y * cols + x - 1]);
        const float gy = 0.5f * (L[(size_t)(y + 1) * cols + x] - L[(size_t)(y - 1) * cols + x]);
        const float m = sqrtf(gx * gx + gy * gy);                         /* :321 */
        grad[(size_t)y * cols + x] = m;
        bg[cnt++] = m;
      }
    qsort(bg, (size_t)bsz, sizeof(float), cmp_float);                     /* nth_element :328 */
    const float th = bg[bsz / 2] + p->grad_th;                            /* :329 */
    int vc = 0;
    for (int y = sy; y < sy + block_h && vc < 80; y++)
      for (int x = sx; x < sx + block_w; x++) {
        if (vc >= 80) break;                                              /* :334 */
        if (grad[(size_t)y * cols + x] > th) { val[(size_t)y * cols + x] = 1; vc++; }
      }
  }
  orc_disparity_scan(L, R, val, rows, cols, bnd, p->ssd_th, p->f0, p->baseline, p->max_disparity, disp, dep, NULL, NULL,
                     &n_sel, &n_match);
  if (st) { st->n_selected = n_sel; st->n_matched = n_match; }
  free(L); free(R); free(grad); free(bg);
  return 0;
}

/* The accept / reject / lambda / stop rule of one evaluation of the inverse-depth LM (ref: src/depth_estimate.cpp:150-161). Returns
 * the stop code (0 carry on, 1 precision break :158, 2 lambda break :152); *accepted tells which branch ran. Pinned to the
 * reference's own lines by tests/test_ref_pin.py (oracle/make_ref_fixtures.py compiles :92-96,141,150-161,167 as they stand). */
static int depth_lm_rule(float err_now, float* err_last, float* lambda, float precision, int* accepted) {
  if (err_now > *err_last) {                                              /* :150 */
    *accepted = 0;
    *lambda = *lambda * 10.0f;                                            /* :151 */
    return (*lambda > 1e+5f) ? 2 : 0;                                     /* :152 */
  }
  *accepted = 1;
  const float err_diff = err_now / *err_last;                             /* :157 */
  if (err_diff > precision) return 1;                                     /* :158 */
  *err_last = err_now;                                                    /* :159 */
  *lambda = fmaxf(*lambda / 10.0f, 1e-7f);                                /* :160 */
  return 0;
}
/* The inverse-depth LM driver replayed on a given sequence of errors, depth vectors as tags (0 = the scan's depths, k + 1 = the
 * vector solved after evaluation k, -1 = the zero vector pre_depth starts as): rec = 5 ints per evaluation {lambda bits, err_last
 * bits, current tag, pre tag, broke}; *iters = iter_count at :171. Same loop as depth_optimization's below; the twin of
 * ref_depth_lm_schedule in oracle/make_ref_fixtures.py. */
int orc_depth_lm_schedule(const float* errs, int n_errs, float lambda0, float precision, int max_iters, int* rec, int* final_current,
                          int* iters) {
  int cur = 0, pre = -1, tmp = 0, iter = 0, k = 0;
  float err_last = 1e+10f, lambda = lambda0;
  while (max_iters > iter) {
    if (k >= n_errs) break;
    int accepted;
    const int stop = depth_lm_rule(errs[k], &err_last, &lambda, precision, &accepted);
    if (accepted) { cur = tmp; pre = cur; }
    else if (!stop) cur = pre;
    memcpy(&rec[5 * k + 0], &lambda, sizeof(int));
    memcpy(&rec[5 * k + 1], &err_last, sizeof(int));
    rec[5 * k + 2] = cur; rec[5 * k + 3] = pre; rec[5 * k + 4] = stop ? 1 : 0;
    k++;
    if (stop) break;
    tmp = k;
    iter++;
  }
  *final_current = cur;
  *iters = iter;
  return k;
}

/* DepthOptimization + ComputeResidualJacobian (ref: src/depth_estimate.cpp:80-198, 200-242).
 * Uses the UNBLURRED images (:67). err sums are accumulated in fp64 and rounded to fp32. */
static int depth_optimization(const float* left, const float* right, int rows, int cols, const orc_depth_params* p,
                              uint8_t* val, float* dep, orc_depth_stats* st) {
  int n = 0;
  for (size_t i = 0; i < (size_t)rows * cols; i++) n += (val[i] == 1);
  int* cx = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int* cy = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  float* buf = (float*)malloc(sizeof(float) * (size_t)(n + 1) * 6);
  float *cur = buf, *pre = buf + (n + 1), *tmp = buf + 2 * (n + 1), *res = buf + 3 * (n + 1), *jtwj = buf + 4 * (n + 1),
        *bb = buf + 5 * (n + 1);
  int k = 0;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++)
      if (val[(size_t)y * cols + x] == 1) { cx[k] = x; cy[k] = y; cur[k] = dep[(size_t)y * cols + x]; k++; }  /* :106-114 */
  for (int i = 0; i < n; i++) { res[i] = 0.0f; pre[i] = 0.0f; tmp[i] = cur[i]; jtwj[i] = 1.0f; bb[i] = 0.0f; }
  float lambda = p->lambda, err_last = 1e+10f, err_now = 0.0f;
  int iter = 0;
  const float tx = p->baseline, fx = p->f0;
  while (p->max_iters > iter) {                                           /* :141 */
    /* ComputeResidualJacobian :200-242 */
    double esum = 0.0;
    int n_act = 0;
    for (int i = 0; i < n; i++) {
      const float wf = floorf((float)cx[i] - tx * fx * tmp[i]);           /* :217 */
      if (!(wf >= 2.0f) || !(wf <= (float)(cols - 2))) {                  /* :219 (NaN -> sentinel) */
        jtwj[i] = 0.0f; bb[i] = 0.0f; res[i] = -1000.0f; continue;
      }
      const int wx = (int)wf;
      const float* Rr = right + (size_t)cy[i] * cols;
      const float r_i = left[(size_t)cy[i] * cols + cx[i]] - Rr[wx];      /* :226 */
      const float w_i = (fabsf(r_i) <= p->huber_delta) ? 1.0f : p->huber_delta / fabsf(r_i);
      const float r_diff = tx * fx * 0.5f * (Rr[wx + 1] - Rr[wx - 1]);    /* :229 */
      res[i] = fabsf(r_i);
      n_act++;
      esum += (double)(r_i * r_i * w_i);                                  /* :233 */
      jtwj[i] = r_diff * r_diff * w_i;                                    /* :234 */
      bb[i] = -r_diff * w_i * r_i;                                        /* :235 */
    }
    err_now = (1.0f / (float)n_act) * (float)esum;                        /* :239 */
    int accepted;
    const int stop = depth_lm_rule(err_now, &err_last, &lambda, p->precision, &accepted);   /* :150-161 */
    if (accepted) {
      memcpy(cur, tmp, sizeof(float) * (size_t)n);                        /* :155 */
      memcpy(pre, cur, sizeof(float) * (size_t)n);                        /* :156 */
    } else if (!stop) memcpy(cur, pre, sizeof(float) * (size_t)n);        /* :153 */
    if (stop) break;
    for (int i = 0; i < n; i++) {                                         /* :164-166 */
      const float A = jtwj[i] + lambda * jtwj[i];
      const float dd = (1.0f / A) * bb[i];
      tmp[i] = dd + cur[i];
    }
    iter++;
  }
  int n_valid = 0;
  for (int i = 0; i < n; i++) {                                           /* :176-191 */
    const size_t o = (size_t)cy[i] * cols + cx[i];
    if (res[i] > p->photo_th || res[i] == -1000.0f) { val[o] = 0; dep[o] = 0.0f; }
    else if (1.0f / cur[i] > p->max_depth || 1.0f / cur[i] < p->min_depth) { val[o] = 0; dep[o] = 0.0f; }
    else { val[o] = 1; dep[o] = cur[i]; n_valid++; }
  }
  if (st) { st->iters = iter; st->cost = err_now; st->n_valid = n_valid; }
  free(cx); free(cy); free(buf);
  return (n_valid < 500) ? -1 : 0;                                        /* :192-197 */
}

/* stage: 1 = disparity only (val/disp/dep after DisparityDepthEstimate), 2 = full ComputeDepth. */
int orc_compute_depth(const float* left, const float* right, int rows, int cols, const orc_depth_params* p,
                      int stage, uint8_t* val, float* disp, float* dep, orc_depth_stats* st) {
  if (st) memset(st, 0, sizeof(*st));
  if (!p->any_size && (rows != 376 || cols != 1241)) return -1;          /* :46-49 */
  memset(val, 0, (size_t)rows * cols);
  memset(disp, 0, sizeof(float) * (size_t)rows * cols);
  memset(dep, 0, sizeof(float) * (size_t)rows * cols);
  if (disparity_depth(left, right, rows, cols, p, val, disp, dep, st)) return -1;
  if (stage == 1) return 0;
  return depth_optimization(left, right, rows, cols, p, val, dep, st);
}

/* ------------------------------------------------------------------------------------------------
 * Runner step (ref: run_odometry_kitti_offline.cpp:198-271), used as the timed CPU baseline.
 * Tracks `cur_left` against the keyframe pyramids and estimates the depth of the current stereo pair.
 * kf_img/kf_dep: keyframe pyramids; out_pose: pose_to_keyframe; cur_img_pyr/cur_dep_pyr: rebuilt
 * pyramids of the current frame (the runner builds the image pyramid twice, :205 and :251).
 * ---------------------------------------------------------------------------------------------- */
int orc_track_frame(const float* kf_img, const float* kf_dep, const float* cur_left, const float* cur_right, int rows,
                    int cols, const orc_lm_params* lp, const orc_depth_params* dp, const float init[16],
                    float out_pose[16], float* cur_img_pyr, float* cur_dep_pyr, uint8_t* val, float* disp, float* dep,
                    orc_depth_stats* st) {
  if (orc_image_pyramid(cur_left, rows, cols, lp->n_levels, 1, cur_img_pyr)) return -1;       /* :205 */
  const int s = orc_lm_solve(kf_img, kf_dep, cur_img_pyr, rows, cols, lp, init, out_pose, NULL, 0, NULL); /* :215 */
  if (orc_compute_depth(cur_left, cur_right, rows, cols, dp, 2, val, disp, dep, st)) return -2; /* :229 */
  if (orc_image_pyramid(cur_left, rows, cols, lp->n_levels, 1, cur_img_pyr)) return -1;       /* :251 */
  if (orc_depth_pyramid(dep, rows, cols, lp->n_levels, cur_dep_pyr)) return -1;               /* :252 */
  return s;
}

/* ------------------------------------------------------------------------------------------------
 * Camera model: CameraPyramid::ConfigureCamera / UndistortRectify
 * (ref: src/camera.cpp:40-69,71-82). The arithmetic lives in OpenCV (cv::initUndistortRectifyMap, cv::remap), which is
 * not vendored: restated from the documented definitions, PARITY UNPINNED like the rest of this file.
 * ---------------------------------------------------------------------------------------------- */
/* Intrinsic pyramid (ref: src/camera.cpp:44-66): out[l] = {fx, fy, f_theta, cx, cy}; P is the 3x4 row-major
 * rectified projection. */
void orc_camera_intrinsics(const double P[12], int levels, double out[][5]) {
  double fx = P[0], fy = P[5], cx = P[2], cy = P[6], ft = P[1];
  for (int l = 0; l < levels; l++) {
    out[l][0] = fx; out[l][1] = fy; out[l][2] = ft; out[l][3] = cx; out[l][4] = cy;
    fx = fx / 2.0; fy = fy / 2.0; ft = ft / 2.0;
    cx = (cx + 0.5) / 2.0 + 0.5;
    cy = (cy + 0.5) / 2.0 + 0.5;
  }
}

static int inv3(const double m[9], double out[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c01 = m[5] * m[6] - m[3] * m[8];
  const double c02 = m[3] * m[7] - m[4] * m[6];
  const double det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
  if (!(fabs(det) > 0.0)) return -1;
  const double id = 1.0 / det;
  out[0] = c00 * id; out[1] = (m[2] * m[7] - m[1] * m[8]) * id; out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  out[3] = c01 * id; out[4] = (m[0] * m[8] - m[2] * m[6]) * id; out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  out[6] = c02 * id; out[7] = (m[1] * m[6] - m[0] * m[7]) * id; out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return 0;
}

/* cv::initUndistortRectifyMap(K_raw, dist = (k1, k2, p1, p2), R, P, size, CV_32FC1) (ref: src/camera.cpp:68).
 * raw = {fx, fy, f_theta (ignored by OpenCV), cx, cy}. Maps are rows x cols fp32. Returns -1 when P[:, :3]*R is
 * singular. */
int orc_camera_init_maps(const double raw[5], const double dist[4], const double R[9], const double P[12], int rows,
                         int cols, float* mapx, float* mapy) {
  double PR[9], iR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      PR[i * 3 + j] = (P[i * 4 + 0] * R[0 * 3 + j] + P[i * 4 + 1] * R[1 * 3 + j]) + P[i * 4 + 2] * R[2 * 3 + j];
  if (inv3(PR, iR)) return -1;
  const double fx = raw[0], fy = raw[1], cx = raw[3], cy = raw[4];
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
  for (int v = 0; v < rows; v++)
    for (int u = 0; u < cols; u++) {
      const double du = (double)u, dv = (double)v;
      const double _x = (iR[0] * du + iR[1] * dv) + iR[2];
      const double _y = (iR[3] * du + iR[4] * dv) + iR[5];
      const double _w = (iR[6] * du + iR[7] * dv) + iR[8];
      const double w = 1.0 / _w;
      const double x = _x * w, y = _y * w;
      const double x2 = x * x, y2 = y * y;
      const double r2 = x2 + y2, _2xy = (2.0 * x) * y;
      const double kr = 1.0 + (k2 * r2 + k1) * r2;
      const double xd = (x * kr + p1 * _2xy) + p2 * (r2 + 2.0 * x2);
      const double yd = (y * kr + p1 * (r2 + 2.0 * y2)) + p2 * _2xy;
      mapx[(size_t)v * cols + u] = (float)(fx * xd + cx);
      mapy[(size_t)v * cols + u] = (float)(fy * yd + cy);
    }
  return 0;
}

/* cv::remap(src, dst, mapx, mapy, INTER_LINEAR, BORDER_CONSTANT, border_value) on CV_32F (ref: src/camera.cpp:80):
 * 5-bit fixed-point source coordinates, fp32 bilinear weights, taps outside the source read border_value. */
void orc_camera_remap(const float* src, int srows, int scols, const float* mapx, const float* mapy, int drows, int dcols,
                      float border_value, float* dst) {
  for (int v = 0; v < drows; v++)
    for (int u = 0; u < dcols; u++) {
      const size_t o = (size_t)v * dcols + u;
      const int sx = (int)rintf(mapx[o] * 32.0f), sy = (int)rintf(mapy[o] * 32.0f);
      const int ix = sx >> 5, iy = sy >> 5;
      const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
      const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
      const int x0 = ix >= 0 && ix < scols, x1 = ix + 1 >= 0 && ix + 1 < scols;
      const int y0 = iy >= 0 && iy < srows, y1 = iy + 1 >= 0 && iy + 1 < srows;
      const float s00 = (x0 && y0) ? src[(size_t)iy * scols + ix] : border_value;
      const float s01 = (x1 && y0) ? src[(size_t)iy * scols + ix + 1] : border_value;
      const float s10 = (x0 && y1) ? src[(size_t)(iy + 1) * scols + ix] : border_value;
      const float s11 = (x1 && y1) ? src[(size_t)(iy + 1) * scols + ix + 1] : border_value;
      dst[o] = ((s00 * w00 + s01 * w01) + s10 * w10) + s11 * w11;
    }
}
