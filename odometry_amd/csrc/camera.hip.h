// camera.hip.h — CameraPyramid on the device: rectified intrinsics per pyramid level, the
// undistort + rectify lookup maps and the remap of raw camera frames.
// Replaces odometry::CameraPyramid (ref: include/camera.h:16-119; src/camera.cpp:12-38 ctor, :40-69 ConfigureCamera,
// :71-82 UndistortRectify). The two OpenCV calls behind it — cv::initUndistortRectifyMap and cv::remap — are restated
// from their documented definitions (the library is not vendored; parity unpinned):
//   map:   [x y w]^T = (P[:, :3] * R)^-1 * [u v 1]^T; x' = x/w, y' = y/w; r2 = x'^2 + y'^2;
//          kr = 1 + (k2*r2 + k1)*r2; xd = x'*kr + p1*2x'y' + p2*(r2 + 2x'^2); yd = y'*kr + p1*(r2 + 2y'^2) + p2*2x'y';
//          map_x = fx*xd + cx, map_y = fy*yd + cy   — fp64, stored as fp32 (CV_32FC1 maps).
//   remap: INTER_LINEAR with OpenCV's 5-bit fixed-point coordinates (INTER_BITS = 5): sx = rint(map_x * 32),
//          ix = sx >> 5, ax = sx & 31, weights (1 - ay/32)(1 - ax/32) ... as fp32 products, value =
//          ((S00*w00 + S01*w01) + S10*w10) + S11*w11; BORDER_CONSTANT: a tap outside the source reads border_value.
// Included at the end of odometry_hip.hip (shares its context, error and allocation helpers).
#pragma once

namespace odo {

struct CamCoef {     // everything the map kernel needs, fp64
  double iR[9];      // (P[:, :3] * R)^-1, row-major
  double fx, fy, cx, cy;  // raw camera matrix (skew is ignored, as cv::initUndistortRectifyMap does)
  double k1, k2, p1, p2;  // radial k1, k2 and tangential p1, p2 (the reference's "r1", "r2")
};

// One map entry.
ODO_HD void undistort_map_entry(const CamCoef& c, int u, int v, float* mx, float* my) {
  const double du = (double)u, dv = (double)v;
  const double _x = (c.iR[0] * du + c.iR[1] * dv) + c.iR[2];
  const double _y = (c.iR[3] * du + c.iR[4] * dv) + c.iR[5];
  const double _w = (c.iR[6] * du + c.iR[7] * dv) + c.iR[8];
  const double w = 1.0 / _w;
  const double x = _x * w, y = _y * w;
  const double x2 = x * x, y2 = y * y;
  const double r2 = x2 + y2, _2xy = (2.0 * x) * y;
  const double kr = 1.0 + (c.k2 * r2 + c.k1) * r2;
  const double xd = (x * kr + c.p1 * _2xy) + c.p2 * (r2 + 2.0 * x2);
  const double yd = (y * kr + c.p1 * (r2 + 2.0 * y2)) + c.p2 * _2xy;
  *mx = (float)(c.fx * xd + c.cx);
  *my = (float)(c.fy * yd + c.cy);
}

__global__ void __launch_bounds__(256) undistort_map_kernel(CamCoef c, int rows, int cols, float* __restrict__ mapx,
                                                            float* __restrict__ mapy) {
  const int u = blockIdx.x * 64 + (threadIdx.x & 63);
  const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= cols || v >= rows) return;
  float mx, my;
  undistort_map_entry(c, u, v, &mx, &my);
  mapx[(size_t)v * cols + u] = mx;
  mapy[(size_t)v * cols + u] = my;
}

// One output pixel per thread: two coalesced map reads, four gathered source taps, one coalesced store
// (algorithmic bytes: 8 B of maps + 4 B of source footprint + 4 B written per output pixel; HBM / L2-gather bound).
__global__ void __launch_bounds__(256) remap_bilinear_kernel(const float* __restrict__ src, int srows, int scols,
                                                             const float* __restrict__ mapx, const float* __restrict__ mapy,
                                                             float* __restrict__ dst, int drows, int dcols,
                                                             float border_value) {
  const int u = blockIdx.x * 64 + (threadIdx.x & 63);
  const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= dcols || v >= drows) return;
  const size_t o = (size_t)v * dcols + u;
  const int sx = (int)rintf(mapx[o] * 32.0f), sy = (int)rintf(mapy[o] * 32.0f);
  const int ix = sx >> 5, iy = sy >> 5;
  const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
  const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
  const bool x0 = (unsigned)ix < (unsigned)scols, x1 = (unsigned)(ix + 1) < (unsigned)scols;
  const bool y0 = (unsigned)iy < (unsigned)srows, y1 = (unsigned)(iy + 1) < (unsigned)srows;
  const float s00 = (x0 && y0) ? src[(size_t)iy * scols + ix] : border_value;
  const float s01 = (x1 && y0) ? src[(size_t)iy * scols + ix + 1] : border_value;
  const float s10 = (x0 && y1) ? src[(size_t)(iy + 1) * scols + ix] : border_value;
  const float s11 = (x1 && y1) ? src[(size_t)(iy + 1) * scols + ix + 1] : border_value;
  dst[o] = ((s00 * w00 + s01 * w01) + s10 * w10) + s11 * w11;
}

}  // namespace odo

// ------------------------------------------------------------------------------------------------
struct odo_camera {
  odo_ctx* ctx;
  int levels;
  double raw[5];      // fx, fy, f_theta, cx, cy of the raw camera matrix (ref: src/camera.cpp:16-25)
  double dist[4];     // k1, k2, r1, r2 (ref: :26-30)
  double sensor_w, sensor_h;
  int res_w, res_h;
  bool configured;
  double intr[ODO_MAX_LEVELS][5];  // rectified fx, fy, f_theta, cx, cy per level (ref: :50-66)
  int map_rows, map_cols;
  float* d_mapx;
  float* d_mapy;
  float* d_src;       // staging for the host-buffer entry point
  float* d_dst;
  size_t src_cap, dst_cap;
};

// 3x3 inverse by cofactors in fp64, fixed operation order (part of the arithmetic contract: the CPU checker repeats it).
static bool cam_inv3(const double m[9], double out[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c01 = m[5] * m[6] - m[3] * m[8];
  const double c02 = m[3] * m[7] - m[4] * m[6];
  const double det = (m[0] * c00 + m[1] * c01) + m[2] * c02;
  if (!(fabs(det) > 0.0)) return false;
  const double id = 1.0 / det;
  out[0] = c00 * id; out[1] = (m[2] * m[7] - m[1] * m[8]) * id; out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  out[3] = c01 * id; out[4] = (m[0] * m[8] - m[2] * m[6]) * id; out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  out[6] = c02 * id; out[7] = (m[1] * m[6] - m[0] * m[7]) * id; out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return true;
}

extern "C" int odo_camera_create(odo_ctx* ctx, int levels, double fx, double fy, double f_theta, double cx, double cy,
                                 double k1, double k2, double r1, double r2, double sensor_width, double sensor_height,
                                 int resolution_width, int resolution_height, odo_camera** out) {
  if (!ctx || !out) return fail("odo_camera_create: NULL arg");
  *out = nullptr;
  if (levels < 1 || levels > ODO_MAX_LEVELS) return fail("odo_camera_create: levels %d out of [1, %d]", levels, ODO_MAX_LEVELS);
  odo_camera* c = new (std::nothrow) odo_camera();
  if (!c) return fail("odo_camera_create: out of memory");
  c->ctx = ctx;
  c->levels = levels;
  c->raw[0] = fx; c->raw[1] = fy; c->raw[2] = f_theta; c->raw[3] = cx; c->raw[4] = cy;
  c->dist[0] = k1; c->dist[1] = k2; c->dist[2] = r1; c->dist[3] = r2;
  c->sensor_w = sensor_width; c->sensor_h = sensor_height;
  c->res_w = resolution_width; c->res_h = resolution_height;
  c->configured = false;
  *out = c;
  return 0;
}

extern "C" int odo_camera_configure(odo_camera* c, const double R_rowmajor[9], const double P_rowmajor[12], int new_width,
                                    int new_height) {
  if (!c || !R_rowmajor || !P_rowmajor) return fail("odo_camera_configure: NULL arg");
  if (new_width < 1 || new_height < 1) return fail("odo_camera_configure: bad size %dx%d", new_width, new_height);
  const double* P = P_rowmajor;
  // intrinsic pyramid (ref: src/camera.cpp:44-66): f halves, c <- (c + 0.5) / 2 + 0.5, all in double
  double fx = P[0], fy = P[5], cx = P[2], cy = P[6], ft = P[1];
  for (int l = 0; l < c->levels; l++) {
    c->intr[l][0] = fx; c->intr[l][1] = fy; c->intr[l][2] = ft; c->intr[l][3] = cx; c->intr[l][4] = cy;
    fx = fx / 2.0; fy = fy / 2.0; ft = ft / 2.0;
    cx = (cx + 0.5) / 2.0 + 0.5;
    cy = (cy + 0.5) / 2.0 + 0.5;
  }
  // lookup maps (ref: :68 cv::initUndistortRectifyMap(intrinsic_raw_, distortion_param_, R, P, size, CV_32FC1, ...))
  double PR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      PR[i * 3 + j] = (P[i * 4 + 0] * R_rowmajor[0 * 3 + j] + P[i * 4 + 1] * R_rowmajor[1 * 3 + j]) + P[i * 4 + 2] * R_rowmajor[2 * 3 + j];
  odo::CamCoef k;
  if (!cam_inv3(PR, k.iR)) return fail("odo_camera_configure: P[:, :3] * R is singular");
  k.fx = c->raw[0]; k.fy = c->raw[1]; k.cx = c->raw[3]; k.cy = c->raw[4];
  k.k1 = c->dist[0]; k.k2 = c->dist[1]; k.p1 = c->dist[2]; k.p2 = c->dist[3];
  HIP_OK(hipSetDevice(c->ctx->device));
  const size_t n = (size_t)new_width * new_height;
  if (c->d_mapx) { (void)hipFree(c->d_mapx); c->d_mapx = nullptr; }
  if (c->d_mapy) { (void)hipFree(c->d_mapy); c->d_mapy = nullptr; }
  HIP_OK(hipMalloc((void**)&c->d_mapx, n * sizeof(float)));
  HIP_OK(hipMalloc((void**)&c->d_mapy, n * sizeof(float)));
  c->map_rows = new_height; c->map_cols = new_width;
  hipLaunchKernelGGL(odo::undistort_map_kernel, dim3((new_width + 63) / 64, (new_height + 3) / 4), dim3(256), 0, c->ctx->stream,
                     k, new_height, new_width, c->d_mapx, c->d_mapy);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(c->ctx->stream));
  c->configured = true;
  return 0;
}

extern "C" int odo_camera_levels(const odo_camera* c) { return c ? c->levels : -1; }

extern "C" int odo_camera_intrinsics(const odo_camera* c, int level, double out5[5]) {
  if (!c || !out5) return fail("odo_camera_intrinsics: NULL arg");
  if (!c->configured) return fail("odo_camera_intrinsics: ConfigureCamera has not run");
  if (level < 0 || level >= c->levels) return fail("odo_camera_intrinsics: level %d out of range", level);
  for (int i = 0; i < 5; i++) out5[i] = c->intr[level][i];
  return 0;
}

extern "C" int odo_camera_raw(const odo_camera* c, double raw5[5], double dist4[4], double sensor2[2], int resolution2[2]) {
  if (!c) return fail("odo_camera_raw: NULL arg");
  if (raw5) for (int i = 0; i < 5; i++) raw5[i] = c->raw[i];
  if (dist4) for (int i = 0; i < 4; i++) dist4[i] = c->dist[i];
  if (sensor2) { sensor2[0] = c->sensor_w; sensor2[1] = c->sensor_h; }
  if (resolution2) { resolution2[0] = c->res_w; resolution2[1] = c->res_h; }
  return 0;
}

extern "C" int odo_camera_map_size(const odo_camera* c, int* rows, int* cols) {
  if (!c || !c->configured) return fail("odo_camera_map_size: not configured");
  if (rows) *rows = c->map_rows;
  if (cols) *cols = c->map_cols;
  return 0;
}

extern "C" int odo_camera_download_maps(const odo_camera* c, float* mapx_host, float* mapy_host) {
  if (!c || !c->configured) return fail("odo_camera_download_maps: not configured");
  HIP_OK(hipSetDevice(c->ctx->device));
  const size_t bytes = (size_t)c->map_rows * c->map_cols * sizeof(float);
  if (mapx_host && copy_to_user_host(c->ctx, mapx_host, c->d_mapx, bytes)) return -1;
  if (mapy_host && copy_to_user_host(c->ctx, mapy_host, c->d_mapy, bytes)) return -1;
  return 0;
}

// Device-resident source and destination (dense rows): the per-frame path.
extern "C" int odo_camera_undistort_rectify_dev(odo_camera* c, const float* src_dev, int src_rows, int src_cols, float* dst_dev,
                                                float border_value) {
  if (!c || !src_dev || !dst_dev) return fail("odo_camera_undistort_rectify_dev: NULL arg");
  if (!c->configured) return fail("odo_camera_undistort_rectify_dev: ConfigureCamera has not run");
  if (src_rows < 1 || src_cols < 1) return fail("odo_camera_undistort_rectify_dev: bad source size");
  HIP_OK(hipSetDevice(c->ctx->device));
  hipLaunchKernelGGL(odo::remap_bilinear_kernel, dim3((c->map_cols + 63) / 64, (c->map_rows + 3) / 4), dim3(256), 0,
                     c->ctx->stream, src_dev, src_rows, src_cols, (const float*)c->d_mapx, (const float*)c->d_mapy, dst_dev,
                     c->map_rows, c->map_cols, border_value);
  HIP_OK(hipGetLastError());
  return 0;
}

// Host buffers (the shim's cv::Mat path): upload, remap, download, synchronous.
extern "C" int odo_camera_undistort_rectify(odo_camera* c, const float* src, int src_rows, int src_cols, float* dst,
                                            float border_value) {
  if (!c || !src || !dst) return fail("odo_camera_undistort_rectify: NULL arg");
  if (!c->configured) return fail("odo_camera_undistort_rectify: ConfigureCamera has not run");
  HIP_OK(hipSetDevice(c->ctx->device));
  const size_t sb = (size_t)src_rows * src_cols * sizeof(float), db = (size_t)c->map_rows * c->map_cols * sizeof(float);
  if (c->src_cap < sb) {
    if (c->d_src) (void)hipFree(c->d_src);
    c->d_src = nullptr; c->src_cap = 0;
    HIP_OK(hipMalloc((void**)&c->d_src, sb));
    c->src_cap = sb;
  }
  if (c->dst_cap < db) {
    if (c->d_dst) (void)hipFree(c->d_dst);
    c->d_dst = nullptr; c->dst_cap = 0;
    HIP_OK(hipMalloc((void**)&c->d_dst, db));
    c->dst_cap = db;
  }
  if (upload_rows_async(c->ctx, c->d_src, src, sb, sb, 1)) return -1;
  if (odo_camera_undistort_rectify_dev(c, c->d_src, src_rows, src_cols, c->d_dst, border_value)) return -1;
  return copy_to_user_host(c->ctx, dst, c->d_dst, db);
}

extern "C" int odo_camera_destroy(odo_camera* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->d_mapx) (void)hipFree(c->d_mapx);
  if (c->d_mapy) (void)hipFree(c->d_mapy);
  if (c->d_src) (void)hipFree(c->d_src);
  if (c->d_dst) (void)hipFree(c->d_dst);
  delete c;
  return 0;
}
