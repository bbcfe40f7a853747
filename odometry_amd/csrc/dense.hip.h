// dense.hip.h — evaluation of DENSE pyramid levels (every interior pixel carries depth: BASELINE.json configs[2]).
// Included by kernels.hip.h. Same per-pixel arithmetic as the reference's scan (ref: src/lm_optimizer.cpp:163-264,
// include/image_processing_global.h:22-69) and bit-identical results to odo::make_point / warp_point, organised for the
// chip instead of for a raster loop:
//   * 2-D work units (a 64-pixel column strip x one row per wavefront) dealt so that the blocks that share an XCD
//     (blockIdx % 8, MI355X guide: a label, speed only) own one horizontal band of the image: every I2 line that the
//     floor-sampled taps touch is fetched into ONE L2, once, and the vertically adjacent rows of a block hit it in L1;
//   * the nine IEEE fp32 divisions and two fp64 divisions per pixel share four reciprocals (see "Shared-reciprocal
//     division" below): ~150 fewer VALU cycles per pixel-wave in a kernel that is VALU-issue bound;
//   * the next unit's D1 / I1 loads are issued before the current unit is evaluated.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include "odo_math.h"

// The kernels below are compiled in a translation unit of their own (dense_kernels.hip; round 2 gave it -fno-slp-vectorize: hipcc's
// SLP vectoriser turns pairs of fp32 operations into v_pk_mul_f32 / v_pk_fma_f32, measured at ~9.5 cycles per wave-instruction
// against ~3.3 for a scalar fp32 operation — tools/microbench/valu_rates.hip; since round 5 the whole library is built that
// way); everything else sees only the level description, its host-side helpers and the launcher declared at the end.
namespace odo {

// The fp64 forms of the shared-reciprocal division below (see there): also used by the persistent LM kernels' warp
// (kernels.hip.h point_residual_g), hence outside the dense unit's guard.
__device__ __forceinline__ double rcp_refined_d(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double div_shared_d(double a, double b, double y) {
  const double q0 = a * y;
  const double r0 = __builtin_fma(-b, q0, a);
  return __builtin_fma(r0, y, q0);
}

#ifdef ODO_DENSE_KERNELS
// ---------------------------------------------------------------------------------------------
// Shared-reciprocal division.
// hipcc -fhip-fp32-correctly-rounded-divide-sqrt expands a / b into
//     bs = v_div_scale(b)   y0 = v_rcp(bs)   e = fma(-bs, y0, 1)   y1 = fma(e, y0, y0)
//     as = v_div_scale(a)   q0 = as * y1     r0 = fma(-bs, q0, as) q1 = fma(r0, y1, q0)
//     r1 = fma(-bs, q1, as) q2 = v_div_fmas(r1, y1, q1)            result = v_div_fixup(q2, b, a)
// (fp64: two Newton steps on the reciprocal, one residual correction). v_div_scale only rescales operands whose
// exponents are extreme (zero / denormal operands, |b| >= 2^126, a / b denormal, exponent(a) - exponent(b) >= 96, |a| <
// 2^-103), v_div_fmas is a plain fma when nothing was scaled and v_div_fixup only overrides special values. For
// operands inside the guarded range below the sequence therefore IS the one written here, operation for operation, and
// the refined reciprocal y1 depends on b alone: several numerators over one denominator share it. The callers guard the
// range per wavefront and fall back to the plain `/` otherwise, so results never depend on which path ran.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_refined(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_shared(float a, float b, float y) {
  const float q0 = a * y;
  const float r0 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-b, q1, a);
  return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ float recip_shared(float b, float y) {  // 1.0f / b: q0 = 1 * y is exact
  const float r0 = __builtin_fmaf(-b, y, 1.0f);
  const float q1 = __builtin_fmaf(r0, y, y);
  const float r1 = __builtin_fmaf(-b, q1, 1.0f);
  return __builtin_fmaf(r1, y, q1);
}

#endif  // ODO_DENSE_KERNELS

// Host-side part of the guard: intrinsics of a level for which the unscaled sequences are exact for every pixel whose
// inverse depth passes the per-lane test |d| <= 4096 (|d| >= 0.01 holds for every evaluated pixel,
// ref: src/lm_optimizer.cpp:193): focal length in [1, 65536], principal point within +-65536 and not closer than 2^-8 to
// an integer without being one (x - cx is then 0 or at least 2^-8 in magnitude), image at most 65535 wide / high.
// With z in [2^-12, 2^7]: |X|, |Y| in {0} u [2^-36, 2^24], f/Z in [2^-7, 2^28], every numerator in {0} u [2^-86, 2^76],
// every quotient normal, every exponent difference below 96.
static inline int dense_fast_ok(double fl, float cx, float cy, int rows, int cols) {
  auto frac_ok = [](float c) {
    const float r = c - floorf(c);
    const float dist = r < 0.5f ? r : 1.0f - r;
    return dist == 0.0f || dist >= 1.0f / 256.0f;
  };
  return fl >= 1.0 && fl <= 65536.0 && fabsf(cx) <= 65536.0f && fabsf(cy) <= 65536.0f && frac_ok(cx) && frac_ok(cy) &&
         rows <= 65535 && cols <= 65535;
}

#ifdef ODO_DENSE_KERNELS
// odo::point_xyz / point_jacobian / make_point with the shared reciprocals: bit-identical for guarded operands.
// yfl = rcp_refined((float)k.fl).
__device__ __forceinline__ void point_xyz_shared(int x, int y, float inv_depth, const LevelK& k, float flf, float yfl, float* X,
                                                 float* Y, float* Z) {
  const float z = recip_shared(inv_depth, rcp_refined(inv_depth));   // :198
  *X = div_shared(z * ((float)x - k.cx), flf, yfl);                  // h:35
  *Y = div_shared(z * ((float)y - k.cy), flf, yfl);                  // h:36
  *Z = z;
}
__device__ __forceinline__ void point_jacobian_shared(PointK* p, const LevelK& k, float flf) {
  const float z = p->Z;
  const float yz = rcp_refined(z);
  const float fx_z = div_shared(flf, z, yz);                         // :223
  const float xy = p->X * p->Y, xx = p->X * p->X, yy = p->Y * p->Y, zz = z * z;
  const float yzz = rcp_refined(zz);
  p->fx_z = fx_z;
  p->jw02 = div_shared(-fx_z * p->X, z, yz);                         // :232
  p->jw03 = div_shared(-fx_z * xy, z, yz);
  p->jw04 = (float)(k.fl * (1.0 + (double)div_shared(xx, zz, yzz)));
  p->jw05 = -fx_z * p->Y;
  p->jw12 = div_shared(-fx_z * p->Y, z, yz);                         // :233
  p->jw13 = (float)(-k.fl * (1.0 + (double)div_shared(yy, zz, yzz)));
  p->jw14 = -p->jw03;
  p->jw15 = fx_z * p->X;
}
__device__ __forceinline__ PointK make_point_shared(int x, int y, float inv_depth, float i1, const LevelK& k, float flf, float yfl) {
  PointK p;
  point_xyz_shared(x, y, inv_depth, k, flf, yfl, &p.X, &p.Y, &p.Z);
  p.i1 = i1;
  point_jacobian_shared(&p, k, flf);
  return p;
}

// odo::warp_point with one fp64 reciprocal for both image coordinates (ref: include/image_processing_global.h:42-59).
// Exact whenever t0, t1 are finite and t2 is a finite positive float (every float is a normal double and no quotient of
// two of them leaves the double range); the caller guards the pose.
__device__ __forceinline__ bool warp_uv_shared(const PointK& p, const float* T, const LevelK& k, float* u, float* v) {
  const float t0 = ((T[0] * p.X + T[4] * p.Y) + T[8] * p.Z) + T[12];
  const float t1 = ((T[1] * p.X + T[5] * p.Y) + T[9] * p.Z) + T[13];
  const float t2 = ((T[2] * p.X + T[6] * p.Y) + T[10] * p.Z) + T[14];
  if (!(t2 > 0.0f)) return false;
  const double t2d = (double)t2;
  const double y = rcp_refined_d(t2d);
  *u = (float)(div_shared_d(k.fl * (double)t0, t2d, y) + (double)k.cx);
  *v = (float)(div_shared_d(k.fl * (double)t1, t2d, y) + (double)k.cy);
  return true;
}
__device__ __forceinline__ bool warp_point_shared(const PointK& p, const float* T, const LevelK& k, int rows, int cols, int* ui, int* vi) {
  float u, v;
  if (!warp_uv_shared(p, T, k, &u, &v)) return false;
  const float fu = floorf(u), fv = floorf(v);
  if (!(fu < (float)cols) || !(fv < (float)rows) || !(fu >= 0.0f) || !(fv >= 0.0f)) return false;
  *ui = (int)fu;
  *vi = (int)fv;
  return true;
}
#endif  // ODO_DENSE_KERNELS

// ---------------------------------------------------------------------------------------------
// Work decomposition of one dense level.
// ---------------------------------------------------------------------------------------------
struct DenseLevel {
  const float* I1;  // keyframe image level
  const float* I2;  // current image level
  const float* D1;  // keyframe inverse depth level
  int rows, cols;
  LevelK k;
  int nblk;       // blocks that evaluate the level (rows of partial sums it writes)
  int n_strips;   // 64-pixel column strips of the interior
  int n_rg;       // row groups of the interior (one row per wavefront of a block)
  int fast_ok;    // dense_fast_ok() of the level's intrinsics
  int max_iters;  // max_iterations_[level] (ref: src/lm_optimizer.cpp:117)
};

constexpr int kDenseMaxBlocks = 1280;  // five 256-thread blocks per CU

// One stream's entry of the table a batched evaluation launch reads (lm_dense_eval_batch_kernel).
struct DenseBatchItem {
  DenseLevel L;
  const LmState* st;
  const float* scale_sqr;
  double* partials;
  int expect_level, robust;
  float huber_delta;
  int pad_;
};

// Fills the decomposition fields for `block_threads`-wide blocks. A block walks `units` (strip, row-group) pairs; the
// grid is capped at max_blocks; levels with fewer units than that get one block per unit.
static inline void dense_level_geometry(DenseLevel* L, int block_threads, int max_blocks) {
  const int R = block_threads / 64;
  const int iw = L->cols - 8, ih = L->rows - 8;
  if (iw <= 0 || ih <= 0) { L->n_strips = 0; L->n_rg = 0; L->nblk = 1; return; }
  L->n_strips = (iw + 63) / 64;
  L->n_rg = (ih + R - 1) / R;
  const long units = (long)L->n_strips * L->n_rg;
  long g = units;  // one unit per block while the grid fits (small levels are latency bound: the less serial work the better)
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  L->nblk = (int)g;
}

#ifdef ODO_DENSE_KERNELS
#ifndef ODO_KREDPAD
#define ODO_KREDPAD 1
constexpr int kRedPad = 8;  // see kernels.hip.h
#endif
// The units of block `b`: XCD group g = b % G owns the band of row groups [g n_rg / G, (g + 1) n_rg / G); inside the band
// the units run strip by strip, top to bottom, and the group's blocks take consecutive runs of them.
struct DenseRun { int strip, rg, rg0, rg1, count; };
__device__ __forceinline__ DenseRun dense_block_run(const DenseLevel& L, int b) {
  // 32-bit arithmetic throughout (64-bit integer division costs several hundred cycles on the device): a band holds at most
  // 1024 strips x 2048 row groups = 2^21 units (65535 x 65535 image), a group at most 128 blocks.
  DenseRun r;
  const unsigned G = L.nblk < 8 ? (unsigned)L.nblk : 8u;
  const unsigned g = (unsigned)b % G, j = (unsigned)b / G;
  const unsigned nj = ((unsigned)L.nblk - g + G - 1u) / G;  // blocks of this group
  r.rg0 = (int)(g * (unsigned)L.n_rg / G);
  r.rg1 = (int)((g + 1u) * (unsigned)L.n_rg / G);
  const unsigned nrg = (unsigned)(r.rg1 - r.rg0);
  const unsigned U = nrg * (unsigned)L.n_strips;
  const unsigned u0 = j * U / nj, u1 = (j + 1u) * U / nj;  // (j + 1) * U <= 129 * 2^21 < 2^32
  r.count = (int)(u1 - u0);
  r.strip = nrg > 0 ? (int)(u0 / nrg) : 0;
  r.rg = nrg > 0 ? r.rg0 + (int)(u0 % nrg) : 0;
  return r;
}

// One pixel, first half: keyframe point constants and the floor-sampled pixel it lands on. FAST selects the
// shared-reciprocal forms (bit-identical inside the guard). kFlags bits 1..4 are ablations for
// tools/microbench/dense_ablate.hip only (they change the result): 2 = no normal-equation products, 4 = no I2 taps,
// 8 = no point constants, 16 = fp32 projection.
template <bool FAST, int kFlags>
__device__ __forceinline__ bool dense_point(int x, int y, float d, float i1, const DenseLevel& L, float flf, float yfl,
                                            const float* T, PointK* po, int* ui, int* vi) {
  PointK p;
  if (kFlags & 8) {
    p.X = ((float)x - L.k.cx) * d; p.Y = ((float)y - L.k.cy) * d; p.Z = d + 1.0f; p.i1 = i1; p.fx_z = flf * d;
    p.jw02 = p.X * d; p.jw03 = p.Y * d; p.jw04 = flf + p.X; p.jw05 = flf - p.Y; p.jw12 = p.X - d; p.jw13 = p.Y - d;
    p.jw14 = -p.jw03; p.jw15 = p.fx_z * p.X;
  } else if (FAST) {
    p = make_point_shared(x, y, d, i1, L.k, flf, yfl);
  } else {
    p = make_point(x, y, d, i1, L.k);
  }
  *po = p;
  if (kFlags & 16) {
    const float t0 = ((T[0] * p.X + T[4] * p.Y) + T[8] * p.Z) + T[12];
    const float t1 = ((T[1] * p.X + T[5] * p.Y) + T[9] * p.Z) + T[13];
    const float t2 = ((T[2] * p.X + T[6] * p.Y) + T[10] * p.Z) + T[14];
    if (!(t2 > 0.0f)) return false;
    const float rz = __builtin_amdgcn_rcpf(t2);
    const float fu = floorf(t0 * rz * flf + L.k.cx), fv = floorf(t1 * rz * flf + L.k.cy);
    if (!(fu < (float)L.cols) || !(fv < (float)L.rows) || !(fu >= 0.0f) || !(fv >= 0.0f)) return false;
    *ui = (int)fu; *vi = (int)fv;
    return true;
  }
  if (FAST) return warp_point_shared(p, T, L.k, L.rows, L.cols, ui, vi);
  return warp_point(p, T, L.k, L.rows, L.cols, ui, vi);
}

// Second half: five I2 taps, Jacobian row, weight, 29 fp64 products (one copy of this code for both paths above).
template <int kFlags>
__device__ __forceinline__ void dense_accumulate(const PointK& p, int ui, int vi, const DenseLevel& L, int robust,
                                                 float huber_delta, float scale_sqr, double acc[ODO_NACC]) {
  float r, J[6];
  if (kFlags & 4) {
    const float gx = (float)ui * 0.001f, gy = (float)vi * 0.002f;
    r = gx - p.i1;
    J[0] = gx * p.fx_z; J[1] = gy * p.fx_z; J[2] = gx * p.jw02 + gy * p.jw12; J[3] = gx * p.jw03 + gy * p.jw13;
    J[4] = gx * p.jw04 + gy * p.jw14; J[5] = gx * p.jw05 + gy * p.jw15;
  } else {
    residual_jacobian(p, L.I2, L.rows, L.cols, ui, vi, &r, J);
  }
  const float w = robust_weight(r, robust, huber_delta, scale_sqr);
  if (kFlags & 2) {
    acc[27] += (double)(((((J[0] + J[1]) + J[2]) + J[3]) + J[4]) + J[5]) * (double)(r * w);
    acc[28] += 1.0;
  } else {
    accumulate_row(acc, r, w, J);
  }
}

// The evaluation of one dense level by one block: acc += every residual of the block's units at pose T.
// kFlags bit 0: never take the shared-reciprocal path (A/B timing and the parity test of the fast path).
template <int kBlock, int kFlags>
__device__ __forceinline__ void dense_eval_block(const DenseLevel& L, int b, const float* T, int robust, float huber_delta,
                                                 float scale_sqr, double acc[ODO_NACC]) {
  constexpr int R = kBlock / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  DenseRun run = dense_block_run(L, b);
  if (run.count <= 0) return;
  const float flf = (float)L.k.fl;
  const float yfl = rcp_refined(flf);
  bool pose_ok = true;  // wave-uniform: every entry of the pose finite and of moderate size (see warp_point_shared)
#pragma unroll
  for (int i = 0; i < 16; i++) pose_ok = pose_ok && (fabsf(T[i]) <= 1048576.0f);
  const bool fast_level = !(kFlags & 1) && L.fast_ok && pose_ok;
  const int x_end = L.cols - 4, y_end = L.rows - 4;
  int strip = run.strip, rg = run.rg;
  // software pipeline: the inverse depth / intensity of the next unit are in flight while this one is evaluated
  auto fetch = [&](int s, int g, float* d, float* i1, int* xo, int* yo) {
    const int x = 4 + s * 64 + lane, y = 4 + g * R + wv;
    *xo = x; *yo = y;
    *d = 0.0f; *i1 = 0.0f;
    if (x < x_end && y < y_end) {
      const size_t o = (size_t)y * L.cols + x;
      *d = L.D1[o];
      *i1 = L.I1[o];
    }
  };
  float d_n, i1_n;
  int x_n, y_n;
  fetch(strip, rg, &d_n, &i1_n, &x_n, &y_n);
  for (int it = 0; it < run.count; it++) {
    const float d = d_n, i1 = i1_n;
    const int x = x_n, y = y_n;
    if (++rg == run.rg1) { rg = run.rg0; strip++; }
    if (it + 1 < run.count) fetch(strip, rg, &d_n, &i1_n, &x_n, &y_n);
    // d == 0 outside the image and for pixels without depth: depth_valid() is false for both (:193)
    const bool valid = depth_valid(d);
    const bool lane_ok = !valid || (fabsf(d) <= 4096.0f);
    PointK p;
    int ui = 0, vi = 0;
    bool hit = false;
    if (fast_level && __all(lane_ok)) {
      if (valid) hit = dense_point<true, kFlags>(x, y, d, i1, L, flf, yfl, T, &p, &ui, &vi);
    } else {
      if (valid) hit = dense_point<false, kFlags>(x, y, d, i1, L, flf, yfl, T, &p, &ui, &vi);
    }
    if (hit) dense_accumulate<kFlags>(p, ui, vi, L, robust, huber_delta, scale_sqr, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined form of dense_eval_block (the one the product uses). A pixel is evaluated in two stages:
//   A: inverse depth -> back-projected point -> warp -> the five I2 taps are ISSUED (nothing waits for them);
//   B: (one unit later) Jacobian constants from the point, taps consumed, weight, 29 fp64 products.
// While stage B of unit i runs, the taps of unit i + 1 and the D1 / I1 loads of unit i + 2 are in flight, so a wave
// covers its own memory latency: with 58 accumulator registers per thread only four waves fit a SIMD, too few to hide
// an Infinity-Cache / HBM round trip behind each other.
// ---------------------------------------------------------------------------------------------
struct DensePix {
  float X, Y, Z, i1;
  float tc, tl, tr, tu, td;  // I2 at the floor-sampled pixel and its index-clamped neighbours (ref: h:62-69); in bilinear
                             // mode the four cell corners: tc = (x0, y0), tr = (x0+1, y0), td = (x0, y0+1), tl = (x0+1, y0+1)
  float fa, fb;              // bilinear mode: position inside the cell
  bool hit;                  // the pixel produces a residual
};

// Load through a wave-uniform base (SGPR pair) + a 32-bit byte offset (one VGPR): no 64-bit address arithmetic per tap.
__device__ __forceinline__ float ld_f32(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int kFlags>
__device__ __forceinline__ void dense_stage_a(int x, int y, float d, float i1, bool fast, const DenseLevel& L, float flf,
                                              float yfl, const float* T, DensePix* q) {
  PointK p;
  p.X = 0.0f; p.Y = 0.0f; p.Z = 0.0f;
  int ui = 0, vi = 0;
  bool hit = false;
  float fa = 0.0f, fb = 0.0f;
  const bool bil = L.k.bilinear != 0;  // wave-uniform
  if (depth_valid(d)) {  // :193 (d == 0 stands for "outside the image" too)
    if (fast) point_xyz_shared(x, y, d, L.k, flf, yfl, &p.X, &p.Y, &p.Z);
    else point_xyz(x, y, d, L.k, &p.X, &p.Y, &p.Z);
    if (bil) {
      float u, v;
      hit = (fast ? warp_uv_shared(p, T, L.k, &u, &v) : warp_point_uv(p, T, L.k, &u, &v)) &&
            bilinear_cell(u, v, L.rows, L.cols, &ui, &vi, &fa, &fb);
    } else {
      hit = fast ? warp_point_shared(p, T, L.k, L.rows, L.cols, &ui, &vi) : warp_point(p, T, L.k, L.rows, L.cols, &ui, &vi);
    }
  }
  if (!hit) { ui = 0; vi = 0; }  // the taps are issued unconditionally (no control flow around the loads): a pixel that
                                 // produces no residual reads pixel (0, 0) and ignores it
  q->X = p.X; q->Y = p.Y; q->Z = p.Z; q->i1 = i1; q->hit = hit; q->fa = fa; q->fb = fb;
  // floor mode: left / right / up / down neighbours, index-clamped; bilinear mode: the cell's other three corners (the cell
  // is inside the image for a hit, and for a miss ui = vi = 0 keeps every index in range as long as the image is >= 2 x 2)
  const int px = bil ? ((ui + 1 < L.cols && vi + 1 < L.rows) ? ui + 1 : ui) : ((ui - 1 >= 0) ? ui - 1 : 0);
  const int nx = (ui + 1 < L.cols) ? ui + 1 : L.cols - 1;
  const int py = bil ? vi : ((vi - 1 >= 0) ? vi - 1 : 0), ny = (vi + 1 < L.rows) ? vi + 1 : L.rows - 1;
  const unsigned cols = (unsigned)L.cols;
  const unsigned rowo = (unsigned)vi * cols;
  q->tc = ld_f32(L.I2, (rowo + (unsigned)ui) * 4u);
  q->tl = ld_f32(L.I2, ((bil ? (unsigned)ny * cols : rowo) + (unsigned)px) * 4u);
  q->tr = ld_f32(L.I2, (rowo + (unsigned)nx) * 4u);
  q->tu = ld_f32(L.I2, ((unsigned)py * cols + (unsigned)ui) * 4u);
  q->td = ld_f32(L.I2, ((unsigned)ny * cols + (unsigned)ui) * 4u);
}

template <int kFlags>
__device__ __forceinline__ void dense_stage_b(const DensePix& q, bool fast, const DenseLevel& L, float flf, int robust,
                                              float huber_delta, float scale_sqr, double acc[ODO_NACC]) {
  if (!q.hit) return;
  PointK p;
  p.X = q.X; p.Y = q.Y; p.Z = q.Z; p.i1 = q.i1;
  if (fast) point_jacobian_shared(&p, L.k, flf);
  else point_jacobian(&p, L.k);
  float r, J[6];
  if (L.k.bilinear) residual_jacobian_bilinear(p, q.tc, q.tr, q.td, q.tl, q.fa, q.fb, &r, J);
  else residual_jacobian_taps(p, q.tc, q.tl, q.tr, q.tu, q.td, &r, J);
  const float w = robust_weight(r, robust, huber_delta, scale_sqr);
  if (kFlags & 2) {
    acc[27] += (double)(((((J[0] + J[1]) + J[2]) + J[3]) + J[4]) + J[5]) * (double)(r * w);
    acc[28] += 1.0;
  } else {
    accumulate_row(acc, r, w, J);
  }
}

template <int kBlock, int kFlags>
__device__ __forceinline__ void dense_eval_block_pipelined(const DenseLevel& L, int b, const float* T, int robust,
                                                           float huber_delta, float scale_sqr, double acc[ODO_NACC]) {
  constexpr int R = kBlock / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const DenseRun run = dense_block_run(L, b);
  if (run.count <= 0) return;
  const float flf = (float)L.k.fl;
  const float yfl = rcp_refined(flf);
  bool pose_ok = true;  // wave-uniform: every entry of the pose finite and of moderate size (see warp_point_shared)
#pragma unroll
  for (int i = 0; i < 16; i++) pose_ok = pose_ok && (fabsf(T[i]) <= 1048576.0f);
  const bool fast_level = !(kFlags & 1) && L.fast_ok && pose_ok;
  const int x_end = L.cols - 4, y_end = L.rows - 4;
  int strip = run.strip, rg = run.rg;
  auto advance = [&]() { if (++rg == run.rg1) { rg = run.rg0; strip++; } };
  // The loaded values are only touched one iteration later (an early use would drain every older load, the taps in
  // flight included): fetch() returns them raw together with the in-image flag.
  auto fetch = [&](float* d, float* i1, int* xo, int* yo, bool* in) {  // the unit at (strip, rg)
    const int x = 4 + strip * 64 + lane, y = 4 + rg * R + wv;
    *xo = x; *yo = y;
    *in = (x < x_end && y < y_end);
    const unsigned o = *in ? ((unsigned)y * (unsigned)L.cols + (unsigned)x) * 4u : 0u;  // unconditional loads, see stage A
    *d = ld_f32(L.D1, o);
    *i1 = ld_f32(L.I1, o);
  };
  auto guard = [&](float d) {  // wave-uniform: may this unit take the shared-reciprocal forms?
    const bool lane_ok = !depth_valid(d) || (fabsf(d) <= 4096.0f);
    return fast_level && __all(lane_ok);
  };
  // The loop body has no block-level control flow around its loads: every step issues exactly two D1 / I1 loads and
  // five taps (units past the end of the run read pixel (0, 0) and are marked invalid), so the compiler's s_waitcnt
  // counts are exact and stage B waits for ITS taps only, with the seven younger loads still in flight. Two steps per
  // trip with the two pixel slots trading roles: no register copies between steps.
  float dA, iA, dB, iB;
  int xA, yA, xB, yB;
  bool inA, inB;
  fetch(&dA, &iA, &xA, &yA, &inA);   // unit 0
  advance();
  fetch(&dB, &iB, &xB, &yB, &inB);   // unit 1
  advance();
  DensePix pa, pb;
  bool fa, fb;
  {
    const float d = inA ? dA : 0.0f;
    fa = guard(d);
    dense_stage_a<kFlags>(xA, yA, d, iA, fa, L, flf, yfl, T, &pa);  // unit 0: taps in flight
  }
  for (int u = 0; u < run.count; u += 2) {
    {  // unit u + 1 through stage A (slot b), unit u + 2 loaded (regs A), unit u through stage B (slot a)
      const float d = (inB && u + 1 < run.count) ? dB : 0.0f, iv = iB;
      const int x = xB, y = yB;
      fetch(&dA, &iA, &xA, &yA, &inA);
      advance();
      fb = guard(d);
      dense_stage_a<kFlags>(x, y, d, iv, fb, L, flf, yfl, T, &pb);
      dense_stage_b<kFlags>(pa, fa, L, flf, robust, huber_delta, scale_sqr, acc);
    }
    {  // unit u + 2 through stage A (slot a), unit u + 3 loaded (regs B), unit u + 1 through stage B (slot b)
      const float d = (inA && u + 2 < run.count) ? dA : 0.0f, iv = iA;
      const int x = xA, y = yA;
      fetch(&dB, &iB, &xB, &yB, &inB);
      advance();
      fa = guard(d);
      dense_stage_a<kFlags>(x, y, d, iv, fa, L, flf, yfl, T, &pa);
      dense_stage_b<kFlags>(pb, fb, L, flf, robust, huber_delta, scale_sqr, acc);
    }
  }
}

// Deterministic block reduction of 29 fp64 accumulators per thread for any block width (multiple of 64, <= 1024):
// rounds of kQ quantities through sh[kQ][kBlock + pad]; thread (q, s) sums every kSub-th entry in ascending order, then a
// kSub-lane xor tree — the association order depends only on the block width.
template <int kBlock>
__device__ __forceinline__ void block_reduce_acc_w(const double acc[ODO_NACC], double* __restrict__ out) {
  constexpr int kQ = (kBlock <= 256) ? 15 : (kBlock <= 512 ? 8 : 4);  // 31.7 KB / 33.3 KB / 33.0 KB of LDS
  constexpr int kSub = kBlock / kQ >= 64 ? 64 : (kBlock / kQ >= 32 ? 32 : (kBlock / kQ >= 16 ? 16 : 8));
  constexpr int kW = kBlock + kRedPad;
  __shared__ double shw[kQ][kW];
  const int t = threadIdx.x;
  constexpr int kRounds = (ODO_NACC + kQ - 1) / kQ;
#pragma unroll
  for (int round = 0; round < kRounds; round++) {
    const int q0 = round * kQ;
    const int nq = (ODO_NACC - q0) < kQ ? (ODO_NACC - q0) : kQ;
    if (round) __syncthreads();
#pragma unroll
    for (int q = 0; q < kQ; q++)
      if (q < nq) shw[q][t] = acc[q0 + q];
    __syncthreads();
    if (t < nq * kSub) {
      const int q = t / kSub, s = t % kSub;
      double v = 0.0;
#pragma unroll 8
      for (int i = 0; i < kBlock / kSub; i++) v += shw[q][i * kSub + s];
#pragma unroll
      for (int o = kSub / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kSub);
      if (s == 0) out[q0 + q] = v;
    }
  }
}

// Residual / Jacobian / normal-equation pass over one dense level at the pose held in the device-resident LM state.
// Replaces ComputeResidualJacobianNaive + the three product passes (ref: src/lm_optimizer.cpp:163-264,129,145-149).
// One row of 29 fp64 partial sums per block. Stale launches (level already stopped) return immediately.
template <int kBlock, int kFlags, int kWaves /* waves per SIMD the register allocation is held to */>
__global__ void __launch_bounds__(kBlock, kWaves) lm_dense_eval_kernel(DenseLevel L, const LmState* __restrict__ st, int expect_level,
                                                                int robust, float huber_delta,
                                                                const float* __restrict__ scale_sqr_ptr,
                                                                double* __restrict__ partials) {
  if (!(st->active != 0 && st->level == expect_level)) return;
  float T[16];   // the pose is wave-uniform: pinned to scalar registers (16 vector registers less in a kernel held to 128 of them)
#pragma unroll
  for (int i = 0; i < 16; i++) T[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(st->T[i])));
  const float scale_sqr = (robust == 2) ? *scale_sqr_ptr : 1.0f;
  double acc[ODO_NACC];
#pragma unroll
  for (int q = 0; q < ODO_NACC; q++) acc[q] = 0.0;
  if (kFlags & 32) dense_eval_block<kBlock, kFlags>(L, blockIdx.x, T, robust, huber_delta, scale_sqr, acc);  // microbench A/B only
  else dense_eval_block_pipelined<kBlock, kFlags>(L, blockIdx.x, T, robust, huber_delta, scale_sqr, acc);
  block_reduce_acc_w<kBlock>(acc, partials + (size_t)blockIdx.x * ODO_NACC);
}

// Several independent streams (optimisers) in ONE launch: blockIdx.y picks the stream's entry of a table in device memory, block
// (x, y) does exactly what block x of the stream's own launch does (same units, same reduction order: bit-identical results).
// One 1080p level is too small to fill the chip for long (2 M pixels: ~18 us, of which launch ramp and tail are a large part);
// S levels side by side are S times the bytes in one ramp. grid = (largest nblk of any stream, rounded up to a multiple of 8 so
// that blockIdx.x % 8 stays the XCD label of every row of the grid; S).
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uniform_f64(double d) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(d)), __builtin_amdgcn_readfirstlane(__double2loint(d)));
}
template <int kBlock, int kFlags, int kWaves>
__global__ void __launch_bounds__(kBlock, kWaves) lm_dense_eval_batch_kernel(const DenseBatchItem* __restrict__ items) {
  const DenseBatchItem& it = items[blockIdx.y];
  if ((int)blockIdx.x >= it.L.nblk) return;
  if (!(it.st->active != 0 && it.st->level == it.expect_level)) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; i++) T[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(it.st->T[i])));
  const float scale_sqr = (it.robust == 2) ? *it.scale_sqr : 1.0f;
  double acc[ODO_NACC];
#pragma unroll
  for (int q = 0; q < ODO_NACC; q++) acc[q] = 0.0;
  // The level description is wave-uniform but comes from memory: pin every field to scalar registers (in the single-stream kernel it
  // is a kernel argument and lives there anyway; left to itself the compiler keeps part of it in vector registers here — 128 VGPRs
  // + 20 B of scratch per lane against 116 and none).
  DenseLevel L;
  L.I1 = uniform_ptr(it.L.I1); L.I2 = uniform_ptr(it.L.I2); L.D1 = uniform_ptr(it.L.D1);
  L.rows = __builtin_amdgcn_readfirstlane(it.L.rows); L.cols = __builtin_amdgcn_readfirstlane(it.L.cols);
  L.k.fl = uniform_f64(it.L.k.fl);
  L.k.cx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(it.L.k.cx)));
  L.k.cy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(it.L.k.cy)));
  L.k.bilinear = __builtin_amdgcn_readfirstlane(it.L.k.bilinear);
  L.nblk = __builtin_amdgcn_readfirstlane(it.L.nblk); L.n_strips = __builtin_amdgcn_readfirstlane(it.L.n_strips);
  L.n_rg = __builtin_amdgcn_readfirstlane(it.L.n_rg); L.fast_ok = __builtin_amdgcn_readfirstlane(it.L.fast_ok);
  L.max_iters = __builtin_amdgcn_readfirstlane(it.L.max_iters);
  const int robust = __builtin_amdgcn_readfirstlane(it.robust);
  const float huber_delta = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(it.huber_delta)));
  dense_eval_block_pipelined<kBlock, kFlags>(L, blockIdx.x, T, robust, huber_delta, scale_sqr, acc);
  block_reduce_acc_w<kBlock>(acc, it.partials + (size_t)blockIdx.x * ODO_NACC);
}

#endif  // ODO_DENSE_KERNELS

// Launcher (defined in dense_kernels.hip): one evaluation of level `L` at the pose in `st` on stream `s`; e0 / e1, when
// given, are start / stop events bound to the dispatch. plain_div = 1 selects the plain IEEE divisions (A/B and parity
// test of the shared-reciprocal path). Grid = L.nblk blocks of kDenseBlock threads.
constexpr int kDenseBlock = 256, kDenseWaves = 4, kDenseGridCap = 1024;
void launch_dense_eval(const DenseLevel& L, const LmState* st, int expect_level, int robust, float huber_delta,
                       const float* scale_sqr_ptr, double* partials, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int plain_div);
// The batched twin: n streams, d_items = device table of n DenseBatchItem, max_nblk = largest L.nblk among them.
void launch_dense_eval_batch(const DenseBatchItem* d_items, int n, int max_nblk, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int plain_div);

}  // namespace odo
