// kernels.hip.h — CDNA4 (gfx950) kernels of the tracking hot path. 64-wide wavefronts throughout.
// Included once by odometry_hip.hip. Compile with -ffp-contract=off (see odo_math.h).
#pragma once
#include <hip/hip_runtime.h>
#include "odo_math.h"

// Phase stamps (ODO_COARSE_STAMPS / ODO_DEPTH_STAMPS: cycle sums of the persistent kernels' phases, the launch timeline) are a
// DIAGNOSTIC BUILD: -DODO_PHASE_STAMPS=1 (python -m odometry_amd.build --stamps -> lib/libodometry_hip_stamps.so). In the product
// build ODO_DBG() is a constant null pointer and every stamp — the counter reads, the laps, the LDS words — is compiled out: left in
// as run-time branches on a kernel argument they cost the headline 1.5 % (round 6: 3 618 -> 3 561 frames/s when the state machine's
// sums moved to LDS; profiles/r06_state_machine_ab.md), by what they do to register allocation and scheduling, not by executing.
#ifndef ODO_PHASE_STAMPS
#define ODO_PHASE_STAMPS 0
#endif
#if ODO_PHASE_STAMPS
#define ODO_DBG(a) ((a).dbg)
#else
#define ODO_DBG(a) ((unsigned long long*)nullptr)
#endif

// Translation units. The machine scheduler that suits a kernel depends on what binds it (profiles/r06_state_machine_ab.md, section
// 5): the single tracker's LM chain — lm_coarse_kernel, lm_fine_kernel and their variants — is ONE wave working through ~1 200
// dependent instructions per evaluation and wants the ILP-first list scheduler (+ 1.3-3 % on the headline); every other kernel of
// the library is a throughput kernel and wants the occupancy-first one (the batched LM kernels + 10-14 % at S = 1 ... 4, the batched
// tracker + 3 % at S = 8). So the chain kernels compile in a unit of their own, lm_chain_kernels.hip: it includes this header with
// ODO_LM_CHAIN_TU set — every other kernel is then a never-instantiated template (declared, never emitted) —, defines the launchers
// declared at the end of this header, and nothing else. odometry_hip.hip (the main unit) does not see the chain kernels at all.
#ifndef ODO_KERNEL
#define ODO_KERNEL __global__      // a kernel definition
#define ODO_KERNEL_T __global__    // ... one that is a template already
#endif
#ifdef ODO_LM_CHAIN_TU
#define ODO_DEVICE_VAR static __device__
#else
#define ODO_DEVICE_VAR __device__
#endif

namespace odo {

constexpr int kWave = 64;
#define ODO_MAX_LEVELS_K 8

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
  return i;
}

// =============================================================================================
// Pyramid kernels (P1/P2 of SURVEY section 8a)
// =============================================================================================

// 3x3 Gaussian blur {1/4,1/2,1/4} separable, reflect-101 (cv::GaussianBlur ksize 3, sigma 0;
// ref: src/image_processing_global.cpp:30, src/depth_estimate.cpp:256-257). The row pass is recomputed
// for the three rows a pixel needs — identical fp32 values to a materialised intermediate.
// grid.z selects one of up to two images (left/right blurred in one launch).
// zero_u8 / zero_f0 / zero_f1 (optional): images of the same size cleared by the z == 0 slice — the depth estimator's
// zero-filled outputs without three extra fill launches.
__device__ __forceinline__ void blur3x3_kernel_body(int side, const float* __restrict__ src0, float* __restrict__ dst0,
                                                       const float* __restrict__ src1, float* __restrict__ dst1,
                                                       int rows, int cols, uint8_t* __restrict__ zero_u8,
                                                       float* __restrict__ zero_f0,
                                                       float* __restrict__ zero_f1) {
  const float* __restrict__ src = side ? src1 : src0;
  float* __restrict__ dst = side ? dst1 : dst0;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cols || y >= rows) return;
  if (side == 0 && zero_u8) {
    const size_t o = (size_t)y * cols + x;
    zero_u8[o] = 0; zero_f0[o] = 0.0f; zero_f1[o] = 0.0f;
  }
  const int xp = reflect101(x - 1, cols), xn = reflect101(x + 1, cols);
  const int yy[3] = {reflect101(y - 1, rows), y, reflect101(y + 1, rows)};
  float t[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float* r = src + (size_t)yy[k] * cols;
    t[k] = r[x] * 0.5f + (r[xp] + r[xn]) * 0.25f;
  }
  dst[(size_t)y * cols + x] = t[1] * 0.5f + (t[0] + t[2]) * 0.25f;
}
ODO_KERNEL void __launch_bounds__(256) blur3x3_kernel(const float* __restrict__ src0, float* __restrict__ dst0,
                                                       const float* __restrict__ src1, float* __restrict__ dst1,
                                                       int rows, int cols, uint8_t* __restrict__ zero_u8 = nullptr,
                                                       float* __restrict__ zero_f0 = nullptr,
                                                       float* __restrict__ zero_f1 = nullptr) {
  blur3x3_kernel_body((int)blockIdx.z, src0, dst0, src1, dst1, rows, cols, zero_u8, zero_f0, zero_f1);
}

// 5x5 pyrDown [1,4,6,4,1]/16 per axis sampled at (2x,2y), reflect-101, dst = (rows/2, cols/2)
// (cv::pyrDown; ref: src/image_processing_global.cpp:38,46). Horizontal pass recomputed per source row.
ODO_KERNEL void __launch_bounds__(256) pyrdown_kernel(const float* __restrict__ src, int rows, int cols,
                                                       float* __restrict__ dst) {
  const int dr = rows / 2, dc = cols / 2;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dc || y >= dr) return;
  int xs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x - 2 + k, cols);
  float h[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const float* s = src + (size_t)reflect101(2 * y - 2 + k, rows) * cols;
    h[k] = ((s[xs[2]] * 6.0f + (s[xs[1]] + s[xs[3]]) * 4.0f) + s[xs[0]]) + s[xs[4]];
  }
  dst[(size_t)y * dc + x] = (((h[2] * 6.0f + (h[1] + h[3]) * 4.0f) + h[0]) + h[4]) * (1.0f / 256.0f);
}

// Depth decimation L_k(y,x) = L_{k-1}(2y+1, 2x+1) (ref: src/image_processing_global.cpp:85-89,99-103).
// cv::medianBlur(src, dst, 3) on fp32 (ref: src/image_processing_global.cpp:77, DepthPyramid with smooth = true): median of the
// 3x3 neighbourhood, replicated border, by the 19-exchange sorting network for nine values (a selection: bit-exact by definition).
__device__ __forceinline__ void med_cx(float& a, float& b) { const float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }
ODO_KERNEL void __launch_bounds__(256) median3x3_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cols || y >= rows) return;
  const int xm = x > 0 ? x - 1 : 0, xp = x + 1 < cols ? x + 1 : cols - 1;
  const int ym = y > 0 ? y - 1 : 0, yp = y + 1 < rows ? y + 1 : rows - 1;
  const float* r0 = src + (size_t)ym * cols;
  const float* r1 = src + (size_t)y * cols;
  const float* r2 = src + (size_t)yp * cols;
  float p0 = r0[xm], p1 = r0[x], p2 = r0[xp], p3 = r1[xm], p4 = r1[x], p5 = r1[xp], p6 = r2[xm], p7 = r2[x], p8 = r2[xp];
  med_cx(p1, p2); med_cx(p4, p5); med_cx(p7, p8); med_cx(p0, p1); med_cx(p3, p4); med_cx(p6, p7);
  med_cx(p1, p2); med_cx(p4, p5); med_cx(p7, p8); med_cx(p0, p3); med_cx(p5, p8); med_cx(p4, p7);
  med_cx(p3, p6); med_cx(p1, p4); med_cx(p2, p5); med_cx(p4, p7); med_cx(p4, p2); med_cx(p6, p4);
  med_cx(p4, p2);
  dst[(size_t)y * cols + x] = p4;
}

ODO_KERNEL void __launch_bounds__(256) decimate_odd_kernel(const float* __restrict__ src, int cols,
                                                            float* __restrict__ dst, int dr, int dc) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dc || y >= dr) return;
  dst[(size_t)y * dc + x] = src[(size_t)(2 * y + 1) * cols + (2 * x + 1)];
}

// Whole image pyramid in ONE launch (levels <= 4): a block owns the 32x32 / 16x16 / 8x8 / 4x4 region of levels
// 0 / 1 / 2 / 3 under one 32x32 tile of the input and recomputes, in LDS, the halo of the intermediate levels it needs
// (53x53 input -> 25x25 of L1 -> 11x11 of L2 -> 4x4 of L3). Every value is produced by exactly the arithmetic of
// blur3x3_kernel / pyrdown_kernel (same taps, same association order, reflect-101 applied at the level being read), so
// the result is bit-identical to the level-by-level kernels; it replaces four dependent launches by one.
struct PyrOut {
  float* lvl[4];
  int rows[4], cols[4];
  int n_levels;
  int smooth;
};

constexpr int kPT = 32;                    // input tile
constexpr int kPIn = 53, kPInS = 54;       // input halo tile (+ padded stride)
constexpr int kPL1 = 25, kPL2 = 11;

__device__ __forceinline__ float pd_h(const float* row, int x0, int x1, int x2, int x3, int x4) {
  return ((row[x2] * 6.0f + (row[x1] + row[x3]) * 4.0f) + row[x0]) + row[x4];
}
__device__ __forceinline__ float pd_v(float r0, float r1, float r2, float r3, float r4) {
  return (((r2 * 6.0f + (r1 + r3) * 4.0f) + r0) + r4) * (1.0f / 256.0f);
}

constexpr int kPyrThreads = 1024;  // the phases are short dependent LDS passes: more threads = fewer trips per thread
// kThreads: 1 024 for one frame (short dependent LDS passes: more threads = fewer trips per thread, lowest latency); the batched
// tracker, whose launches carry many frames, uses fewer threads per tile so that more tiles are resident per CU (throughput).
template <int kThreads>
__device__ __forceinline__ void image_pyramid_fused_kernel_body(const float* __restrict__ src, PyrOut o) {
  __shared__ float in_t[kPIn * kPInS];
  __shared__ float h_t[kPIn * kPL1];   // horizontal pass (reused per level)
  __shared__ float l1_t[kPL1 * kPL1];
  __shared__ float l2_t[kPL2 * kPL2];
  const int t = threadIdx.x;
  const int ty = blockIdx.y, tx = blockIdx.x;
  const int R0 = o.rows[0], C0 = o.cols[0];
  const int iy0 = kPT * ty - 14, ix0 = kPT * tx - 14;  // input halo origin (level-0 coordinates, may be negative)
  for (int e = t; e < kPIn * kPIn; e += kThreads) {
    const int i = e / kPIn, j = e % kPIn;
    in_t[i * kPInS + j] = src[(size_t)reflect101(iy0 + i, R0) * C0 + reflect101(ix0 + j, C0)];
  }
  __syncthreads();
  // ---- level 0: blur (or copy) of the owned 32x32 ----
  for (int e = t; e < kPT * kPT; e += kThreads) {
    const int y = kPT * ty + e / kPT, x = kPT * tx + e % kPT;
    if (y < R0 && x < C0) {
      const int i = y - iy0, j = x - ix0;
      float v;
      if (o.smooth) {
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float* row = &in_t[(i - 1 + k) * kPInS];
          r[k] = row[j] * 0.5f + (row[j - 1] + row[j + 1]) * 0.25f;
        }
        v = r[1] * 0.5f + (r[0] + r[2]) * 0.25f;
      } else {
        v = in_t[i * kPInS + j];
      }
      o.lvl[0][(size_t)y * C0 + x] = v;
    }
  }
  if (o.n_levels < 2) return;
  // ---- level 1 halo tile: rows/cols [16t-6, 16t+18] clipped to the level ----
  const int R1 = o.rows[1], C1 = o.cols[1];
  const int y1lo = max(16 * ty - 6, 0), x1lo = max(16 * tx - 6, 0);
  const int y1hi = min(16 * ty + 18, R1 - 1), x1hi = min(16 * tx + 18, C1 - 1);
  const int n1y = y1hi - y1lo + 1, n1x = x1hi - x1lo + 1;
  // horizontal pass over every input row of the halo, for the needed L1 columns
  for (int e = t; e < kPIn * kPL1; e += kThreads) {
    const int i = e / kPL1, jx = e % kPL1;
    if (jx < n1x) {
      const int x1 = x1lo + jx;
      const float* row = &in_t[i * kPInS];
      int xs[5];
#pragma unroll
      for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x1 - 2 + k, C0) - ix0;
      h_t[i * kPL1 + jx] = pd_h(row, xs[0], xs[1], xs[2], xs[3], xs[4]);
    }
  }
  __syncthreads();
  for (int e = t; e < kPL1 * kPL1; e += kThreads) {
    const int iy = e / kPL1, jx = e % kPL1;
    if (iy < n1y && jx < n1x) {
      const int y1 = y1lo + iy;
      float r[5];
#pragma unroll
      for (int k = 0; k < 5; k++) r[k] = h_t[(reflect101(2 * y1 - 2 + k, R0) - iy0) * kPL1 + jx];
      const float v = pd_v(r[0], r[1], r[2], r[3], r[4]);
      l1_t[iy * kPL1 + jx] = v;
      const int x1 = x1lo + jx;
      if (y1 >= 16 * ty && y1 < 16 * ty + 16 && x1 >= 16 * tx && x1 < 16 * tx + 16) o.lvl[1][(size_t)y1 * C1 + x1] = v;
    }
  }
  if (o.n_levels < 3) return;
  __syncthreads();
  // ---- level 2 halo tile: [8t-2, 8t+8] clipped ----
  const int R2 = o.rows[2], C2 = o.cols[2];
  const int y2lo = max(8 * ty - 2, 0), x2lo = max(8 * tx - 2, 0);
  const int y2hi = min(8 * ty + 8, R2 - 1), x2hi = min(8 * tx + 8, C2 - 1);
  const int n2y = y2hi - y2lo + 1, n2x = x2hi - x2lo + 1;
  for (int e = t; e < kPL1 * kPL2; e += kThreads) {  // horizontal pass over L1 halo rows
    const int i = e / kPL2, jx = e % kPL2;
    if (i < n1y && jx < n2x) {
      const int x2 = x2lo + jx;
      const float* row = &l1_t[i * kPL1];
      int xs[5];
#pragma unroll
      for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x2 - 2 + k, C1) - x1lo;
      h_t[i * kPL2 + jx] = pd_h(row, xs[0], xs[1], xs[2], xs[3], xs[4]);
    }
  }
  __syncthreads();
  for (int e = t; e < kPL2 * kPL2; e += kThreads) {
    const int iy = e / kPL2, jx = e % kPL2;
    if (iy < n2y && jx < n2x) {
      const int y2 = y2lo + iy;
      float r[5];
#pragma unroll
      for (int k = 0; k < 5; k++) r[k] = h_t[(reflect101(2 * y2 - 2 + k, R1) - y1lo) * kPL2 + jx];
      const float v = pd_v(r[0], r[1], r[2], r[3], r[4]);
      l2_t[iy * kPL2 + jx] = v;
      const int x2 = x2lo + jx;
      if (y2 >= 8 * ty && y2 < 8 * ty + 8 && x2 >= 8 * tx && x2 < 8 * tx + 8) o.lvl[2][(size_t)y2 * C2 + x2] = v;
    }
  }
  if (o.n_levels < 4) return;
  __syncthreads();
  // ---- level 3: the owned 4x4 ----
  const int R3 = o.rows[3], C3 = o.cols[3];
  for (int e = t; e < kPL2 * 4; e += kThreads) {  // horizontal pass over L2 halo rows for the 4 owned columns
    const int i = e / 4, jx = e % 4;
    const int x3 = 4 * tx + jx;
    if (i < n2y && x3 < C3) {
      const float* row = &l2_t[i * kPL2];
      int xs[5];
#pragma unroll
      for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x3 - 2 + k, C2) - x2lo;
      h_t[i * 4 + jx] = pd_h(row, xs[0], xs[1], xs[2], xs[3], xs[4]);
    }
  }
  __syncthreads();
  if (t < 16) {
    const int y3 = 4 * ty + t / 4, x3 = 4 * tx + t % 4;
    if (y3 < R3 && x3 < C3) {
      float r[5];
#pragma unroll
      for (int k = 0; k < 5; k++) r[k] = h_t[(reflect101(2 * y3 - 2 + k, R2) - y2lo) * 4 + (t % 4)];
      o.lvl[3][(size_t)y3 * C3 + x3] = pd_v(r[0], r[1], r[2], r[3], r[4]);
    }
  }
}
ODO_KERNEL void __launch_bounds__(kPyrThreads) image_pyramid_fused_kernel(const float* __restrict__ src, PyrOut o) {
  image_pyramid_fused_kernel_body<kPyrThreads>(src, o);
}
// Large images (many more tiles than CUs): 256 threads per tile, eight tiles resident per CU instead of two.
constexpr int kPyrThreadsWide = 256;
ODO_KERNEL void __launch_bounds__(kPyrThreadsWide) image_pyramid_fused_wide_kernel(const float* __restrict__ src, PyrOut o) {
  image_pyramid_fused_kernel_body<kPyrThreadsWide>(src, o);
}

// Whole depth pyramid in one launch: L_k(Y,X) = L_0(2^k Y + 2^k - 1, 2^k X + 2^k - 1), the composition of the
// reference's odd decimations (ref: src/image_processing_global.cpp:85-89,99-103). Thread <-> level-0 pixel.
__device__ __forceinline__ void depth_pyramid_fused_kernel_body(const float* src, PyrOut o) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= o.cols[0] || y >= o.rows[0]) return;
  const float v = src[(size_t)y * o.cols[0] + x];
  if (src != o.lvl[0]) o.lvl[0][(size_t)y * o.cols[0] + x] = v;   // in place when level 0 was filled by the median filter
#pragma unroll
  for (int l = 1; l < 4; l++) {
    if (l < o.n_levels) {
      const int m = (1 << l) - 1;
      if (((y + 1) & m) == 0 && ((x + 1) & m) == 0) {
        const int Y = y >> l, X = x >> l;
        if (Y < o.rows[l] && X < o.cols[l]) o.lvl[l][(size_t)Y * o.cols[l] + X] = v;
      }
    }
  }
}
ODO_KERNEL void __launch_bounds__(256) depth_pyramid_fused_kernel(const float* src, PyrOut o) {
  depth_pyramid_fused_kernel_body(src, o);
}

// =============================================================================================
// Pose LM kernels (L2/L3/L5 of SURVEY section 8a)
// =============================================================================================

struct LevelView {
  const float* I1;  // keyframe image level
  const float* I2;  // current image level
  const float* D1;  // keyframe inverse depth level
  int rows, cols;
};

constexpr int kLmBlock = 256;
#ifndef ODO_KREDPAD
#define ODO_KREDPAD 1
constexpr int kRedPad = 8;  // sh[q][256+8] doubles: q-stride shifts 16 banks -> at most 2-way conflicts
#endif

// Deterministic block reduction of 29 fp64 accumulators per thread -> out[29].
// Thread t < 232 owns quantity q = t>>3 and sums the 32 values sh[q][i*8 + (t&7)] in ascending i,
// then an 8-lane xor tree. The association order is fixed, so results are run-to-run identical.
__device__ __forceinline__ void block_reduce_acc(const double acc[ODO_NACC], double* __restrict__ out) {
  __shared__ double sh[ODO_NACC][kLmBlock + kRedPad];
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < ODO_NACC; q++) sh[q][t] = acc[q];
  __syncthreads();
  if (t < ODO_NACC * 8) {
    const int q = t >> 3, s = t & 7;
    double v = 0.0;
#pragma unroll 8
    for (int i = 0; i < kLmBlock / 8; i++) v += sh[q][i * 8 + s];
    v += __shfl_xor(v, 4, 8);
    v += __shfl_xor(v, 2, 8);
    v += __shfl_xor(v, 1, 8);
    if (s == 0) out[q] = v;
  }
}

// Same reduction in two rounds of 15 + 14 quantities through a buffer half the size (31.7 KB): five 256-thread
// blocks fit a CU instead of two. Used by the dense scan, which is throughput bound and wants the occupancy; the
// association order is identical to block_reduce_acc, so both give the same bits.
__device__ __forceinline__ void block_reduce_acc_2r(const double acc[ODO_NACC], double* __restrict__ out) {
  __shared__ double sh2[15][kLmBlock + kRedPad];
  const int t = threadIdx.x;
#pragma unroll
  for (int round = 0; round < 2; round++) {
    const int q0 = round * 15, nq = round == 0 ? 15 : ODO_NACC - 15;
    if (round) __syncthreads();
#pragma unroll
    for (int q = 0; q < 15; q++)
      if (q < nq) sh2[q][t] = acc[q0 + q];
    __syncthreads();
    if (t < nq * 8) {
      const int q = t >> 3, s = t & 7;
      double v = 0.0;
#pragma unroll 8
      for (int i = 0; i < kLmBlock / 8; i++) v += sh2[q][i * 8 + s];
      v += __shfl_xor(v, 4, 8);
      v += __shfl_xor(v, 2, 8);
      v += __shfl_xor(v, 1, 8);
      if (s == 0) out[q0 + q] = v;
    }
  }
}

// Residual / Jacobian / normal-equation pass over the interior of one level, reading the pose from the
// device-resident LM state (no host round trip between iterations). Dense scan: thread <-> interior pixel
// (grid-stride), the reference's own iteration space (ref: src/lm_optimizer.cpp:190-191). Coalesced reads of
// D1/I1; the five I2 taps of neighbouring pixels land in neighbouring cache lines (L1/L2 resident).
// Writes one 29-vector of fp64 partials per block. `expect_level` guards against stale launches: once the
// level's loop has stopped (state->active == 0) the launch returns immediately.
ODO_KERNEL void __launch_bounds__(kLmBlock) lm_residual_dense_kernel(LevelView v, LevelK k, const LmState* __restrict__ st,
                                                                      int expect_level, int robust, float huber_delta,
                                                                      const float* __restrict__ scale_sqr_ptr,
                                                                      double* __restrict__ partials) {
  if (!(st->active != 0 && st->level == expect_level)) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; i++) T[i] = st->T[i];
  const float scale_sqr = (robust == 2) ? *scale_sqr_ptr : 1.0f;
  double acc[ODO_NACC];
#pragma unroll
  for (int q = 0; q < ODO_NACC; q++) acc[q] = 0.0;
  const int iw = v.cols - 8, ih = v.rows - 8;
  const int n = (iw > 0 && ih > 0) ? iw * ih : 0;
  // (x, y) of the interior raster index advance incrementally: one integer division per thread instead of one per pixel
  const int stride = gridDim.x * kLmBlock;
  const int sy = (iw > 0) ? stride / iw : 0, sx = (iw > 0) ? stride % iw : 0;  // wave-uniform
  int idx = blockIdx.x * kLmBlock + threadIdx.x;
  int yi = (iw > 0) ? idx / iw : 0, xi = (iw > 0) ? idx % iw : 0;
  for (; idx < n; idx += stride, yi += sy, xi += sx) {
    if (xi >= iw) { xi -= iw; yi++; }
    const int y = 4 + yi, x = 4 + xi;
    const size_t o = (size_t)y * v.cols + x;
    const float d = v.D1[o];
    if (!depth_valid(d)) continue;
    const PointK p = make_point(x, y, d, v.I1[o], k);
    int ui, vi;
    if (!warp_point(p, T, k, v.rows, v.cols, &ui, &vi)) continue;
    float r, J[6];
    residual_jacobian(p, v.I2, v.rows, v.cols, ui, vi, &r, J);
    const float w = robust_weight(r, robust, huber_delta, scale_sqr);
    accumulate_row(acc, r, w, J);
  }
  block_reduce_acc_2r(acc, partials + (size_t)blockIdx.x * ODO_NACC);
}

}  // namespace odo
#include "dense.hip.h"
namespace odo {

// ---------------------------------------------------------------------------------------------
// Semi-dense keyframe point lists. Everything of a residual that does not depend on the pose (back-projected
// point, keyframe intensity, geometric Jacobian at the un-warped point; ref: src/lm_optimizer.cpp:193-234) is
// computed ONCE per keyframe and level, compacted in raster order — the reference's iteration order — into
// three float4 arrays + one float array (52 B per point, lane-consecutive 16-B loads).
// ---------------------------------------------------------------------------------------------
struct PointList {
  float4* a;  // X, Y, Z, i1
  float4* b;  // fx_z, jw02, jw03, jw04
  float4* c;  // jw05, jw12, jw13, jw14
  float* d;   // jw15
};

struct KfLevels {
  const float* I1[ODO_MAX_LEVELS_K];
  const float* D1[ODO_MAX_LEVELS_K];
  int rows[ODO_MAX_LEVELS_K], cols[ODO_MAX_LEVELS_K];
  int row_base[ODO_MAX_LEVELS_K + 1];  // first global row index of each level (interior rows only)
  int n_levels;
};

__device__ __forceinline__ int kf_find_level(const KfLevels& kl, int grow) {
  int l = 0;
  while (l + 1 < kl.n_levels && grow >= kl.row_base[l + 1]) l++;
  return l;
}

// Pass 1: one block per interior row (all levels in one launch): number of valid depths in the row.
// KL = KfLevels (a kernel argument, by value) or const KfLevels& (a table entry in global memory: dynamic level indexing
// then stays plain loads instead of a private copy).
template <class KL>
__device__ __forceinline__ void kf_count_kernel_body(KL kl, int* __restrict__ rowcnt) {
  __shared__ int sh[4];
  const int l = kf_find_level(kl, blockIdx.x);
  const int y = 4 + (blockIdx.x - kl.row_base[l]);
  const int cols = kl.cols[l];
  const float* D = kl.D1[l] + (size_t)y * cols;
  int c = 0;
  for (int x = 4 + threadIdx.x; x < cols - 4; x += 256) c += depth_valid(D[x]) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) rowcnt[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// npts_zero: the per-level totals kf_fill_kernel writes (levels without interior rows stay at 0), cleared here so that no
// memset operation is needed in front of the pair of launches.
ODO_KERNEL void __launch_bounds__(256) kf_count_kernel(KfLevels kl, int* __restrict__ rowcnt, int* __restrict__ npts_zero) {
  if (npts_zero && blockIdx.x == 0 && threadIdx.x < ODO_MAX_LEVELS_K) npts_zero[threadIdx.x] = 0;
  kf_count_kernel_body<KfLevels>(kl, rowcnt);
}

// Pass 2: one block per interior row: its offset from the row counts above it, ordered compaction (ballot prefix), the
// per-point constants; the last row of a level writes the level's total.
// l = the level of this block's row (kf_find_level), pl = that level's list.
template <class KL>
__device__ __forceinline__ void kf_fill_kernel_body(KL kl, int l, PointList pl, float f0, float cx0, float cy0,
                                                      const int* __restrict__ rowcnt, int* __restrict__ npts, int dense_above = 0) {
  __shared__ int wave_tot[4];
  __shared__ int base_sh;
  if (dense_above > 0) {
    // A level that will run the dense scan anyway — more than dense_above points AND more than half of its interior with
    // depth (the host's rule, lm_take_counts) — needs its COUNT only: skip the 52 B per point its list would take (130 MB per
    // keyframe on a dense 1080p pyramid). Every block of the level reaches the same verdict from the same row counts.
    const int r0 = kl.row_base[l], r1 = kl.row_base[l + 1];
    int part = 0;
    for (int i = r0 + (int)threadIdx.x; i < r1; i += 256) part += rowcnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = part;
    __syncthreads();
    const int total = (wave_tot[0] + wave_tot[1]) + (wave_tot[2] + wave_tot[3]);
    __syncthreads();
    const long interior = (long)(kl.rows[l] - 8) * (kl.cols[l] - 8);
    if (total > dense_above && 2L * total > interior) {
      if (threadIdx.x == 0 && (int)blockIdx.x == r1 - 1) npts[l] = total;
      return;
    }
  }
  const int y = 4 + (blockIdx.x - kl.row_base[l]);
  const int cols = kl.cols[l];
  const LevelK k = make_level_k(f0, cx0, cy0, l);
  const float* D = kl.D1[l] + (size_t)y * cols;
  const float* I = kl.I1[l] + (size_t)y * cols;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  {  // this row's offset = number of points in the rows of the level above it (fixed order: exact integer sums)
    const int r0 = kl.row_base[l], my = blockIdx.x - r0;
    int part = 0;
    for (int i = t; i < my; i += 256) part += rowcnt[r0 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) wave_tot[wv] = part;
    __syncthreads();
    if (t == 0) base_sh = (wave_tot[0] + wave_tot[1]) + (wave_tot[2] + wave_tot[3]);
    __syncthreads();
  }
  for (int x0 = 4; x0 < cols - 4; x0 += 256) {
    const int x = x0 + t;
    float d = 0.0f;
    const bool f = (x < cols - 4) && depth_valid(d = D[x]);
    const unsigned long long bal = __ballot(f);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(bal);
    __syncthreads();
    int off = base_sh;
    for (int w = 0; w < wv; w++) off += wave_tot[w];
    if (f) {
      const PointK p = make_point(x, y, d, I[x], k);
      const int i = off + pre;
      pl.a[i] = make_float4(p.X, p.Y, p.Z, p.i1);
      pl.b[i] = make_float4(p.fx_z, p.jw02, p.jw03, p.jw04);
      pl.c[i] = make_float4(p.jw05, p.jw12, p.jw13, p.jw14);
      pl.d[i] = p.jw15;
    }
    __syncthreads();
    if (t == 0) base_sh += (wave_tot[0] + wave_tot[1]) + (wave_tot[2] + wave_tot[3]);
    __syncthreads();
  }
  // the last row of a level knows the level's total
  if (t == 0 && (int)blockIdx.x == kl.row_base[l + 1] - 1) npts[l] = base_sh;
}
ODO_KERNEL void __launch_bounds__(256) kf_fill_kernel(KfLevels kl, float f0, float cx0, float cy0,
                                                      const int* __restrict__ rowcnt, int* __restrict__ npts, PointList pl0, PointList pl1,
                                                      PointList pl2, PointList pl3, PointList pl4, PointList pl5,
                                                      PointList pl6, PointList pl7, int dense_above) {
  const int l = kf_find_level(kl, blockIdx.x);
  const PointList pl = l == 0 ? pl0 : l == 1 ? pl1 : l == 2 ? pl2 : l == 3 ? pl3 : l == 4 ? pl4 : l == 5 ? pl5 : l == 6 ? pl6 : pl7;
  kf_fill_kernel_body<KfLevels>(kl, l, pl, f0, cx0, cy0, rowcnt, npts, dense_above);
}

__device__ __forceinline__ PointK load_point(const PointList& pl, int i) {
  const float4 a = pl.a[i], b = pl.b[i], c = pl.c[i];
  PointK p;
  p.X = a.x; p.Y = a.y; p.Z = a.z; p.i1 = a.w;
  p.fx_z = b.x; p.jw02 = b.y; p.jw03 = b.z; p.jw04 = b.w;
  p.jw05 = c.x; p.jw12 = c.y; p.jw13 = c.z; p.jw14 = c.w;
  p.jw15 = pl.d[i];
  return p;
}

// odo::point_residual for the persistent kernels, whose level table lives in LDS: a pointer read from LDS is a generic pointer in a
// vector register pair — the five taps became FLAT loads (counted on the LDS counter as well as the memory one: every wait for an
// LDS read then also waits for them) behind 64-bit address arithmetic per tap. Here the image base is pinned to a scalar register
// pair (it is wave-uniform) and the taps are GLOBAL loads at base + a 32-bit byte offset. Same taps, same arithmetic.
__device__ __forceinline__ const float* lm_uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float ldg_f32(const float* base_uniform, unsigned byte_off) {
  typedef __attribute__((address_space(1))) const char GlobalBytes;
  typedef __attribute__((address_space(1))) const float GlobalFloat;
  return *(GlobalFloat*)((GlobalBytes*)base_uniform + byte_off);
}
template <bool kMayBilinear>
__device__ __forceinline__ bool point_residual_g(const PointK& p, const float* T, const LevelK& k, const float* I2_uniform, int rows,
                                                 int cols, float* r, float J[6]) {
  if (kMayBilinear && k.bilinear) return point_residual<true>(p, T, k, I2_uniform, rows, cols, r, J);
  // odo::warp_point (ref: include/image_processing_global.h:42-59) with ONE fp64 reciprocal for both image coordinates (dense.hip.h,
  // "Shared-reciprocal division": 13 fp64 operations instead of 22). u = float(fl * double(t0) / double(t2) + double(cx)): the
  // operands are floats, so as doubles they are normal, every quotient stays inside the double range, v_div_scale rescales nothing
  // and the compiler's division sequence IS rcp_refined_d + div_shared_d operation for operation whenever t0, t1, t2 are finite. When
  // one of them is not (a pose gone to infinity), both forms give a non-finite u or v and the point is skipped below either way.
  const float t0 = ((T[0] * p.X + T[4] * p.Y) + T[8] * p.Z) + T[12];
  const float t1 = ((T[1] * p.X + T[5] * p.Y) + T[9] * p.Z) + T[13];
  const float t2 = ((T[2] * p.X + T[6] * p.Y) + T[10] * p.Z) + T[14];
  if (!(t2 > 0.0f)) return false;
  const double t2d = (double)t2, y = rcp_refined_d(t2d);
  const float u = (float)(div_shared_d(k.fl * (double)t0, t2d, y) + (double)k.cx);
  const float v = (float)(div_shared_d(k.fl * (double)t1, t2d, y) + (double)k.cy);
  const float fu = floorf(u), fv = floorf(v);
  if (!(fu < (float)cols) || !(fv < (float)rows) || !(fu >= 0.0f) || !(fv >= 0.0f)) return false;
  const int ui = (int)fu, vi = (int)fv;
  // (ref: lm_optimizer.cpp:215-217, image_processing_global.h:62-69 — odo::residual_jacobian's taps)
  const int px = (ui - 1 >= 0) ? ui - 1 : 0, nx = (ui + 1 < cols) ? ui + 1 : cols - 1;
  const int py = (vi - 1 >= 0) ? vi - 1 : 0, ny = (vi + 1 < rows) ? vi + 1 : rows - 1;
  const unsigned row = (unsigned)vi * (unsigned)cols, ucols = (unsigned)cols;
  const float tc = ldg_f32(I2_uniform, (row + (unsigned)ui) * 4u), tl = ldg_f32(I2_uniform, (row + (unsigned)px) * 4u),
              tr = ldg_f32(I2_uniform, (row + (unsigned)nx) * 4u), tu = ldg_f32(I2_uniform, ((unsigned)py * ucols + (unsigned)ui) * 4u),
              td = ldg_f32(I2_uniform, ((unsigned)ny * ucols + (unsigned)ui) * 4u);
  // odo::residual_jacobian_taps without its two products by the zeros of Jw — J[0] = gx fx_z + gy 0, J[1] = gx 0 + gy fx_z (ref:
  // lm_optimizer.cpp:235, jw(1,0) = jw(0,1) = 0): for finite gradients the sum with a signed zero can only turn a -0 into a +0, and a
  // zero entry of J reaches the sums as a zero addend of accumulators that start at +0 — every sum is the same bit pattern.
  const float gx = 0.5f * (tr - tl), gy = 0.5f * (td - tu);
  *r = tc - p.i1;
  J[0] = gx * p.fx_z;
  J[1] = gy * p.fx_z;
  J[2] = gx * p.jw02 + gy * p.jw12;
  J[3] = gx * p.jw03 + gy * p.jw13;
  J[4] = gx * p.jw04 + gy * p.jw14;
  J[5] = gx * p.jw05 + gy * p.jw15;
  return true;
}

// Residual / Jacobian / normal-equation pass over a keyframe point list (semi-dense levels): one point per thread
// (grid-stride beyond the block cap), 52 B of list + five I2 taps per point. Same arithmetic as the dense scan.
ODO_KERNEL void __launch_bounds__(kLmBlock) lm_residual_list_kernel(PointList pl, int n, const float* __restrict__ I2, int rows,
                                                                     int cols, LevelK k, const LmState* __restrict__ st,
                                                                     int expect_level, int robust, float huber_delta,
                                                                     const float* __restrict__ scale_sqr_ptr,
                                                                     double* __restrict__ partials) {
  if (!(st->active != 0 && st->level == expect_level)) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; i++) T[i] = st->T[i];
  const float scale_sqr = (robust == 2) ? *scale_sqr_ptr : 1.0f;
  double acc[ODO_NACC];
#pragma unroll
  for (int q = 0; q < ODO_NACC; q++) acc[q] = 0.0;
  for (int idx = blockIdx.x * kLmBlock + threadIdx.x; idx < n; idx += gridDim.x * kLmBlock) {
    const PointK p = load_point(pl, idx);
    float r, J[6];
    if (!point_residual(p, T, k, I2, rows, cols, &r, J)) continue;
    const float w = robust_weight(r, robust, huber_delta, scale_sqr);
    accumulate_row(acc, r, w, J);
  }
  block_reduce_acc(acc, partials + (size_t)blockIdx.x * ODO_NACC);
}

// t-distribution pass 1 over a point list: residual per point (NaN = skipped).
ODO_KERNEL void __launch_bounds__(kLmBlock) lm_residual_only_list_kernel(PointList pl, int n, const float* __restrict__ I2,
                                                                          int rows, int cols, LevelK k,
                                                                          const LmState* __restrict__ st, int expect_level,
                                                                          float* __restrict__ res) {
  if (!(st->active != 0 && st->level == expect_level)) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; i++) T[i] = st->T[i];
  for (int idx = blockIdx.x * kLmBlock + threadIdx.x; idx < n; idx += gridDim.x * kLmBlock) {
    const PointK p = load_point(pl, idx);
    float r = __builtin_nanf(""), rr;
    if (point_residual_only(p, T, k, I2, rows, cols, &rr)) r = rr;
    res[idx] = r;
  }
}

// t-distribution mode (robust == 2) needs every residual before any weight
// (ComputeScaleNaive, ref: src/lm_optimizer.cpp:338-358): pass 1 stores r per interior pixel (NaN = skipped).
ODO_KERNEL void __launch_bounds__(kLmBlock) lm_residual_only_kernel(LevelView v, LevelK k, const LmState* __restrict__ st,
                                                                     int expect_level, float* __restrict__ res) {
  if (!(st->active != 0 && st->level == expect_level)) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; i++) T[i] = st->T[i];
  const int iw = v.cols - 8, ih = v.rows - 8;
  const int n = (iw > 0 && ih > 0) ? iw * ih : 0;
  for (int idx = blockIdx.x * kLmBlock + threadIdx.x; idx < n; idx += gridDim.x * kLmBlock) {
    const int y = 4 + idx / iw, x = 4 + idx % iw;
    const size_t o = (size_t)y * v.cols + x;
    const float d = v.D1[o];
    float r = __builtin_nanf("");
    if (depth_valid(d)) {
      const PointK p = make_point(x, y, d, v.I1[o], k);
      float rr;
      if (point_residual_only(p, T, k, v.I2, v.rows, v.cols, &rr)) r = rr;
    }
    res[idx] = r;
  }
}

static_assert(sizeof(LmState) <= 64 * sizeof(int), "LmState must fit one wavefront-wide copy");
static_assert(sizeof(LmState) == 64 * sizeof(int), "LmState is exactly 64 dwords");

struct LmTraceRow {
  int level, iter, n_res, accepted, stop;
  float err, lambda_after;
  float delta[6];
};
constexpr int kTraceCap = 128;

// Damped 6x6 solve spread over the lanes of ONE wavefront: lane (i*8 + j) owns A[i][j] of the augmented 6x7
// system (j == 6 is the right-hand side). Exactly the operations of odo::solve_damped (elimination down the diagonal
// in fp64, zero pivot -> zero component, reciprocal-pivot back substitution), but a column step is ~40 wave
// instructions instead of ~400 single-lane ones: a lone lane pays full issue latency per instruction, which made
// the serial solve the longest kernel of the tracker. acc: 29 fp64 accumulators (LDS). delta_out: 6 floats (LDS).
__device__ __forceinline__ double readlane_d(double v, int src_lane) {  // src_lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void solve_damped_wave_regs(const double* acc, float lambda, float delta_out[6]) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  const bool in = (i < 6 && j < 7);
  double a = 0.0;
  if (in) {
    if (j < 6) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      a = acc[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
      if (i == j) a = a + (double)lambda * a;
    } else {
      a = -acc[21 + i];
    }
  }
  // Elimination down the diagonal (no row exchanges: the system is symmetric positive semi-definite). The pivot is
  // wave-uniform (v_readlane -> SGPRs); the pivot row and this lane's column-c entry are two independent
  // ds_bpermute gathers per step.
  unsigned okmask = 0u;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const double piv = readlane_d(a, c * 8 + c);
    if (fabs(piv) > 0.0) {  // wave-uniform
      okmask |= 1u << c;
      const double prow = __shfl(a, c * 8 + j, 64);
      const double mycol = __shfl(a, (i < 6 ? i : 5) * 8 + c, 64);
      if (in && i > c && j >= c) {
        // (Round 5, measured and dropped: the division spelled out as the compiler's own sequence — v_div_scale, v_rcp_f64, two
        //  Newton steps, v_div_fmas, v_div_fixup — with the pivot's refined reciprocal computed under the latency of the two
        //  gathers and used whenever v_div_scale scales nothing: bit-identical, five dependent fp64 operations off the chain per
        //  column, and 2 % SLOWER over the whole frame, 3 485 -> 3 408 frames/s.)
        const double f = mycol / piv;
        a = a - f * prow;
      }
    }
  }
  // Reciprocal pivots: the diagonal lanes divide side by side (one divide latency for all six).
  const double rdiag = (in && i == j && ((okmask >> i) & 1u)) ? 1.0 / a : 0.0;
  // Back substitution on wave-uniform values (compile-time indices, xs[] stays in registers).
  double xs[6];
#pragma unroll
  for (int c = 5; c >= 0; c--) {
    double s = readlane_d(a, c * 8 + 6);
#pragma unroll
    for (int jj = c + 1; jj < 6; jj++) s = s - readlane_d(a, c * 8 + jj) * xs[jj];
    xs[c] = ((okmask >> c) & 1u) ? s * readlane_d(rdiag, c * 8 + c) : 0.0;
  }
#pragma unroll
  for (int c = 0; c < 6; c++) delta_out[c] = (float)xs[c];  // wave-uniform: every lane holds the step
}
// The same, step written to memory (LDS) by lane 0.
__device__ __forceinline__ void solve_damped_wave(const double* acc, float lambda, float* delta_out) {
  float d[6];
  solve_damped_wave_regs(acc, lambda, d);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 6; c++) delta_out[c] = d[c];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// exp() of the state machine with its four fp64 polynomial chains side by side. odo::sincos_f evaluates a sine and a cosine
// polynomial (8 dependent multiply / add pairs each: contraction is off) one after the other, and se3_exp calls it twice
// (theta / 2 for the quaternion, theta for the translation's V matrix): ~1 000 cycles of a 6 900-cycle state machine on a
// wave whose 64 lanes all compute the same thing. Here lane group g = lane & 3 evaluates ONE of the four chains — g = 0: sin
// of xa, 1: cos of xa, 2: sin of xb, 3: cos of xb — with its own argument and its own coefficients, and the four results are
// read back with v_readlane. Every lane performs exactly the operations odo::sincos_f performs for its function, in its
// order (x - K is x + (-K)): the results are bit-identical. Must be called by a full wavefront with wave-uniform arguments.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_pair_lanes(float xa_f, float xb_f, float* sa, float* ca, float* sb, float* cb) {
  const int g = threadIdx.x & 3;
  const bool is_cos = (g & 1) != 0;
  const double x = (double)((g & 2) ? xb_f : xa_f);
  const double kf = floor(x * 6.36619772367581382433e-01 + 0.5);
  const double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
  const double r2 = r * r;
  // coefficients, highest power first; the eight Horner steps of both polynomials have the same shape p = p * r2 + k
  double p = is_cos ? 1.0 / 6402373705728000.0 : 1.0 / 355687428096000.0;
  p = p * r2 + (is_cos ? -(1.0 / 20922789888000.0) : -(1.0 / 1307674368000.0));
  p = p * r2 + (is_cos ? 1.0 / 87178291200.0 : 1.0 / 6227020800.0);
  p = p * r2 + (is_cos ? -(1.0 / 479001600.0) : -(1.0 / 39916800.0));
  p = p * r2 + (is_cos ? 1.0 / 3628800.0 : 1.0 / 362880.0);
  p = p * r2 + (is_cos ? -(1.0 / 40320.0) : -(1.0 / 5040.0));
  p = p * r2 + (is_cos ? 1.0 / 720.0 : 1.0 / 120.0);
  p = p * r2 + (is_cos ? -(1.0 / 24.0) : -(1.0 / 6.0));
  p = p * r2 + (is_cos ? 1.0 / 2.0 : 1.0);
  // sine: sr = ps * r; cosine: cr = 1 - pc * r2
  const double tail = p * (is_cos ? r2 : r);
  const double v = is_cos ? 1.0 - tail : tail;
  const int q = (int)((long long)kf & 3);
  const double sra = readlane_d(v, 0), cra = readlane_d(v, 1), srb = readlane_d(v, 2), crb = readlane_d(v, 3);
  const int qa = __builtin_amdgcn_readlane(q, 0), qb = __builtin_amdgcn_readlane(q, 2);
  // odo::sincos_f's quadrant table — q 0: (s, c), 1: (c, -s), 2: (-s, -c), 3: (-c, s) — without its four-way branch: swap on bit 0,
  // the sine's sign from bit 1, the cosine's from bit 1 of q + 1; on the wave-uniform bit patterns, so a handful of scalar selects and
  // XORs (the branches compiled to ~50 scalar instructions and eight jumps per pair). Negation = the sign bit, as -x is.
  auto quadrant = [](double sr, double cr, int qq, float* s_out, float* c_out) {
    const unsigned long long sb = (unsigned long long)__double_as_longlong(sr), cb_ = (unsigned long long)__double_as_longlong(cr);
    const bool swap = (qq & 1) != 0;
    unsigned long long s0 = swap ? cb_ : sb, c0 = swap ? sb : cb_;
    s0 ^= (unsigned long long)(unsigned)(qq & 2) << 62;
    c0 ^= (unsigned long long)(unsigned)((qq + 1) & 2) << 62;
    *s_out = (float)__longlong_as_double((long long)s0);
    *c_out = (float)__longlong_as_double((long long)c0);
  };
  quadrant(sra, cra, qa, sa, ca);
  quadrant(srb, crb, qb, sb, cb);
}

// odo::se3_exp with the two sincos calls replaced by one sincos_pair_lanes (same operations otherwise, same order).
__device__ __forceinline__ void se3_exp_wave(const float a[6], Se3* o) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = (ox * ox + oy * oy) + oz * oz;
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  const bool small = theta < 1e-5f;   // wave-uniform
  float sh = 0.0f, ch = 0.0f, st = 0.0f, ct = 0.0f;
  if (!small) sincos_pair_lanes(half_theta, theta, &sh, &ch, &st, &ct);
  float imag, real;
  if (small) {
    const float theta_po4 = theta_sq * theta_sq;
    imag = (0.5f - (float)(1.0 / 48.0) * theta_sq) + (float)(1.0 / 3840.0) * theta_po4;
    real = (1.0f - (float)(1.0 / 8.0) * theta_sq) + (float)(1.0 / 384.0) * theta_po4;
  } else {
    imag = sh / theta;
    real = ch;
  }
  o->qw = real; o->qx = imag * ox; o->qy = imag * oy; o->qz = imag * oz;
  const float Om[9] = {0.0f, -oz, oy, oz, 0.0f, -ox, -oy, ox, 0.0f};
  float Om2[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      Om2[i * 3 + j] = (Om[i * 3 + 0] * Om[0 * 3 + j] + Om[i * 3 + 1] * Om[1 * 3 + j]) + Om[i * 3 + 2] * Om[2 * 3 + j];
  float V[9];
  if (small) {
    quat_to_rot(*o, V);
  } else {
    const float tsq = theta * theta;
    const float ca = (1.0f - ct) / tsq;
    const float cb = (theta - st) / (tsq * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const float id = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
      V[i] = (id + ca * Om[i]) + cb * Om2[i];
    }
  }
  o->tx = (V[0] * a[0] + V[1] * a[1]) + V[2] * a[2];
  o->ty = (V[3] * a[0] + V[4] * a[1]) + V[5] * a[2];
  o->tz = (V[6] * a[0] + V[7] * a[1]) + V[8] * a[2];
}
// odo::lm_apply_step for a full wavefront (every lane carries the state): exp through se3_exp_wave.
__device__ __forceinline__ void lm_apply_step_wave(LmState* s, int max_iters) {
  Se3 d;
  se3_exp_wave(s->delta, &d);                        // :152
  se3_left_update(d, s->cur, &s->inc);               // :153
  se3_to_colmajor(s->inc, s->T);
  s->iter++;                                         // :154
  if (!(max_iters > s->iter)) { s->active = 0; s->stop_reason = 3; }
}

// ---------------------------------------------------------------------------------------------
// Row-sharing block accumulation of the normal equations (fused LM kernels).
// Instead of 29 fp64 products per thread followed by a 29 x T transpose through LDS, every thread publishes the ROW of
// its point — J[6], fl32(J*w)[6], r, fl32(r*w): 14 floats — and 29 x S threads (S sub-lanes per quantity) each sum one
// quantity over the rows of every S-th point: acc_q = sum_p fma(double(A[p]), double(B[p])), A/B two of the 14 row
// entries. Same products as odo::accumulate_row (fp32 operands, exact in fp64), a fixed association order, a quarter of
// the LDS traffic and 47 fewer instructions in the evaluation itself. A point that is skipped publishes a zero row.
// ---------------------------------------------------------------------------------------------
constexpr int kRowFloats = 15;  // rows 0..5 J, 6..11 JW, 12 r, 13 rw, 14 valid (1 / 0: its square sums to the count)
// Layout of one row: the value of point t sits at (t % S) * (T / S + 4) + t / S — the T / S values a sub-lane s sums (points s, s + S,
// ...) are contiguous and 16-byte aligned, so its chain reads them as T / S / 4 ds_read_b128 instead of T / S ds_read_b32 per
// operand; the stride T / S + 4 = 36 keeps the stores of a wave (64 consecutive points) on distinct banks two by two.
constexpr int kRowSub = 8;      // S: sub-lanes per quantity in every kernel (one accumulation order everywhere)
template <int T> struct RowBuf {
  static constexpr int kStride = T / kRowSub + 4;
  static constexpr int W = kRowSub * kStride;
  static constexpr int kBytes = kRowFloats * W * (int)sizeof(float);
};

__device__ __forceinline__ void rows_store(float* __restrict__ rows, int W, int t, const float J[6], float w, float r, bool valid) {
  const int p = (t % kRowSub) * (W / kRowSub) + t / kRowSub;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    rows[a * W + p] = valid ? J[a] : 0.0f;
    rows[(6 + a) * W + p] = valid ? J[a] * w : 0.0f;
  }
  rows[12 * W + p] = valid ? r : 0.0f;
  rows[13 * W + p] = valid ? r * w : 0.0f;
  rows[14 * W + p] = valid ? 1.0f : 0.0f;
}
// The same for callers that hand in all-zero J, w, r for a point without a residual (every caller below initialises them so and
// writes them only behind a successful point_residual): the fourteen selects against `valid` are then no-ops — fourteen
// instructions per point in a phase that is issue-bound at two waves per SIMD.
__device__ __forceinline__ void rows_store_z(float* __restrict__ rows, int W, int t, const float J[6], float w, float r, bool valid) {
  const int p = (t % kRowSub) * (W / kRowSub) + t / kRowSub;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    rows[a * W + p] = J[a];
    rows[(6 + a) * W + p] = J[a] * w;
  }
  rows[12 * W + p] = r;
  rows[13 * W + p] = r * w;
  rows[14 * W + p] = valid ? 1.0f : 0.0f;
}
// Which two rows quantity q multiplies: q < 21: (JW[a], J[b]) for the upper triangle in row-major order;
// 21..26: (JW[a], r); 27: (rw, r); 28: (valid, valid) = the number of residuals.
__device__ __forceinline__ void rows_of_quantity(int q, int* rowA, int* rowB) {
  int a = 0, k = q;
  while (a < 5 && k >= 6 - a) { k -= 6 - a; a++; }   // q < 21: a = row of the upper triangle, k = b - a
  if (q < 21) { *rowA = 6 + a; *rowB = a + k; }
  else if (q < 27) { *rowA = 6 + (q - 21); *rowB = 12; }
  else if (q == 27) { *rowA = 13; *rowB = 12; }
  else { *rowA = 14; *rowB = 14; }
}
// The butterfly over the eight sub-lanes of a quantity (xor 4, then 2, then 1 — the order is part of the sums' bits) on DPP instead
// of ds_bpermute: quad_perm for xor 1 and xor 2; xor 4 = row_shl:4 for the lanes whose bit 2 is clear, row_shr:4 for the others
// (both sources lie inside the lane's row of 16). Three dependent ~110-cycle trips through the LDS crossbar become ~15 VALU moves.
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rows_butterfly8(double acc) {
  const double up = dpp_d<0x104>(acc), dn = dpp_d<0x114>(acc);   // row_shl:4: lane i <- i + 4; row_shr:4: lane i <- i - 4
  acc += (threadIdx.x & 4) ? dn : up;                            // xor 4
  acc += dpp_d<0x4E>(acc);                                       // quad_perm [2, 3, 0, 1]: xor 2
  acc += dpp_d<0xB1>(acc);                                       // quad_perm [1, 0, 3, 2]: xor 1
  return acc;
}
// One round: thread (q, s), t = q * S + s < 29 * S, adds the products of the points s, s + S, ... of the block, in that order.
template <int T, int S>
__device__ __forceinline__ double rows_accumulate(const float* __restrict__ rows, int rowA, int rowB, int s, double acc) {
  static_assert(S == kRowSub && (T / S) % 4 == 0, "the row layout is built for S sub-lanes");
  constexpr int W = RowBuf<T>::W;
  const float4* A = (const float4*)(rows + rowA * W + s * RowBuf<T>::kStride);
  const float4* B = (const float4*)(rows + rowB * W + s * RowBuf<T>::kStride);
#pragma unroll
  for (int i = 0; i < T / S / 4; i++) {
    const float4 a = A[i], b = B[i];
    acc = fma((double)a.x, (double)b.x, acc);
    acc = fma((double)a.y, (double)b.y, acc);
    acc = fma((double)a.z, (double)b.z, acc);
    acc = fma((double)a.w, (double)b.w, acc);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// t-distribution weights inside the fused kernels (robust == 2: ComputeScaleNaive, ref: src/lm_optimizer.cpp:257-261,338-358).
// The scale is a fixed-point iteration sigma <- sqrt(mean_i f(r_i, sigma)): every pass is one more sum over ALL residuals of the
// evaluation, so it is one more reduction through whatever joins the evaluation's points — a wave, the workgroup's LDS (coarse
// kernel), the L2 exchange (persistent kernel). One summation order everywhere, defined on the point list alone:
//   partial[c] = wave_sum64 of the terms of points 64 c .. 64 c + 63 (a skipped point's term is +0);
//   total      = wave_sum64 over lanes l of (partial[l] + partial[l + 64] + ... ascending)
// with wave_sum64 the butterfly below (xor 1, xor 2, mirror of 8, mirror of 16 on DPP; the four row sums as (r0 + r1) + (r2 + r3)).
// Terms are fp32 values >= +0 summed in fp64, so chunks without points (+0) change nothing: the coarse kernel, the persistent
// kernel and lm_tdist_scale_kernel (the unfused pipeline, point-list levels) give the same sigma bit for bit.
// Must be called by a full wavefront.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum64(double v) {
  v += dpp_d<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_d<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_d<0x141>(v);   // row_half_mirror: lane i <- 7 - i (the other quad of the eight)
  v += dpp_d<0x140>(v);   // row_mirror: lane i <- 15 - i (the other eight of the row)
  const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
  return (r0 + r1) + (r2 + r3);
}
// One residual's term of a scale pass (ref: src/lm_optimizer.cpp:350-351), nu = 200.
__device__ __forceinline__ float tdist_term(float e2, float sigma_sqr) { return e2 * (1.0f + 200.0f) / (200.0f + e2 / sigma_sqr); }
// sigma of the next pass from the pass's total and the number of residuals (ref: :353), and the loop test (:354).
__device__ __forceinline__ float tdist_next_sigma(double total, int n) { return sqrtf((float)(total / (double)n)); }
__device__ __forceinline__ bool tdist_converged(float nxt, float cur) { return !(fabsf(nxt - cur) >= 1e-3f); }
constexpr int kTdistMaxPasses = 1000;   // the restatement's guard (a scale iteration that does not settle)

// Fixed-point iteration for the t-distribution scale over the stored residuals (unfused pipeline); one workgroup, sums in fp64
// with a fixed association order (ref: src/lm_optimizer.cpp:338-358). Writes sigma^2. res[i] = NaN: point i gave no residual.
// list_order (point-list levels, n <= 64 * kTdistChunksMax): the summation order of the fused kernels (see wave_sum64) — the same
// sigma bit for bit whichever pipeline evaluates the level. Otherwise (dense levels: up to millions of residuals) a strided
// per-thread sum and a tree.
constexpr int kTdistChunksMax = 640;   // 64-point chunks of the largest point-list level (160 virtual blocks)
ODO_KERNEL void __launch_bounds__(1024) lm_tdist_scale_kernel(const float* __restrict__ res, int n,
                                                               const LmState* __restrict__ st, int expect_level,
                                                               float* __restrict__ scale_sqr_out, int list_order,
                                                               int* __restrict__ only_if = nullptr) {
  if (!(st->active != 0 && st->level == expect_level)) return;
  // only_if: the fall-back of lm_tdist_scale_multi_kernel — runs only when that launch gave up, and clears the flag
  if (only_if) {
    if (__hip_atomic_load(only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    __syncthreads();   // (every thread has read the flag)
    if (threadIdx.x == 0) {
      __hip_atomic_store(only_if, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      only_if[1] += 1;   // scale passes redone here, for odo_lm_tdist_stats: this kernel sums in another association than the multi-workgroup
    }                    // launch, so after a give-up sigma (and the pose) may differ in the last bits from a run without one
  }
  __shared__ double shs[1024];
  __shared__ int shn[1024];
  __shared__ float sh_sigma;
  __shared__ int sh_done;
  const int t = threadIdx.x;
  float cur = 5.0f;
  if (list_order) {
    double* part = shs;                 // [kTdistChunksMax] chunk sums of the pass (kTdistChunksMax <= 1024)
    const int lane = t & 63, wv = t >> 6;
    const int nchunk = (n + 63) / 64;
    int n_valid = 0;
    // the wave's residuals (chunks wv, wv + 16, ...) stay in registers for every pass: one trip to memory per launch, not per pass
    constexpr int kPer = kTdistChunksMax / 16;
    float e2[kPer];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < kPer; u++) {
      const int c = wv + 16 * u, i = 64 * c + lane;
      const float r = (c < nchunk && i < n) ? res[i] : __builtin_nanf("");
      const bool ok = (r == r);
      e2[u] = ok ? r * r : -1.0f;   // (< 0: no residual)
      cnt += __popcll(__ballot(ok));
    }
    for (int pass = 0; pass < kTdistMaxPasses; pass++) {
      const float sigma_sqr = cur * cur;
#pragma unroll
      for (int u = 0; u < kPer; u++) {
        const int c = wv + 16 * u;
        if (c < nchunk) {   // wave-uniform
          const double ws = wave_sum64(e2[u] >= 0.0f ? (double)tdist_term(e2[u], sigma_sqr) : 0.0);
          if (lane == 0) part[c] = ws;
        }
      }
      if (pass == 0 && lane == 0) shn[wv] = cnt;
      __syncthreads();
      if (pass == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) tot += shn[w];
        n_valid = tot;
      }
      double a = 0.0;   // (the one order: a virtual block's four chunk sums first, then the virtual blocks lane-strided, then across lanes)
      for (int v = lane; 4 * v < nchunk; v += 64) {
        const int c = 4 * v;
        const double p1 = (c + 1 < nchunk) ? part[c + 1] : 0.0, p2 = (c + 2 < nchunk) ? part[c + 2] : 0.0, p3 = (c + 3 < nchunk) ? part[c + 3] : 0.0;
        a += ((part[c] + p1) + p2) + p3;
      }
      const double total = wave_sum64(a);   // every wave for itself: nothing to broadcast
      const float nxt = (n_valid > 0) ? tdist_next_sigma(total, n_valid) : cur;
      const bool done = (n_valid == 0) || tdist_converged(nxt, cur);
      cur = nxt;
      if (done) break;
      __syncthreads();   // the chunk sums have been read: the next pass may overwrite them
    }
    if (t == 0) *scale_sqr_out = cur * cur;
    return;
  }
  for (int guard = 0; guard < kTdistMaxPasses; guard++) {
    const float init_sigma = cur;
    const float sigma_sqr = cur * cur;
    double s = 0.0;
    int cnt = 0;
    for (int i = t; i < n; i += 1024) {
      const float r = res[i];
      if (r == r) {
        s += (double)tdist_term(r * r, sigma_sqr);
        cnt++;
      }
    }
    shs[t] = s;
    shn[t] = cnt;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (t < o) { shs[t] += shs[t + o]; shn[t] += shn[t + o]; }
      __syncthreads();
    }
    if (t == 0) {
      const float nxt = (shn[0] > 0) ? tdist_next_sigma(shs[0], shn[0]) : cur;
      sh_sigma = nxt;
      sh_done = (shn[0] == 0) || tdist_converged(nxt, init_sigma);
    }
    __syncthreads();
    cur = sh_sigma;
    const int done = sh_done;
    __syncthreads();
    if (done) break;
  }
  if (t == 0) *scale_sqr_out = cur * cur;
}

// =============================================================================================
// Fused LM iteration: ONE launch per evaluation.
// Every block first re-derives, redundantly and bit-identically, the LM step that the previous launch's partial
// sums imply (fold -> accept/reject -> wave-parallel 6x6 solve -> exp / compose), then evaluates its share of the
// points at the new pose and writes the next partial sums. No separate update kernel, no kernel boundary and no
// trip through memory between "new pose known" and "residuals at the new pose". State and partials are
// double-buffered by launch sequence number; one extra block that evaluates nothing (the last of the grid) alone
// publishes the state, the trace row, the progress words and — when the Solve has finished — the result.
// =============================================================================================
struct StepLevel {   // everything a launch needs to evaluate one pyramid level
  PointList pl;
  int n, nblk;       // points of the keyframe list, blocks that evaluate them
  const float* I2;   // current image, this level
  int rows, cols;
  LevelK k;
  int max_iters;     // max_iterations_[level] (ref: src/lm_optimizer.cpp:117)
  int pad_;
};
struct StepArgs {
  StepLevel lv[ODO_MAX_LEVELS_K];
  int n_levels;
  const LmState* st_in;
  LmState* st_out;
  const double* part_in;
  double* part_out;
  float lambda0, precision;
  int robust;
  float huber_delta;
  LmTraceRow* trace;
  float* cost_stat;
  int* host_prog;
  int seq;
  int first_of_solve;   // 1: the state is initialised from `init` (lm_begin_solve) instead of being loaded
  float init[16];       // affine_init_, column-major (ref: src/lm_optimizer.cpp:76-78)
  unsigned long long* dbg;  // diagnostic (ODO_COARSE_STAMPS): cycle sums of the coarse kernel's phases, else null
  float* out;           // host-mapped result (42 floats, see lm_write_result), written by the launch that finishes the Solve
  int* done_flag;       // host-mapped: set to `token` when `out` is complete
  int token;            // identifies this Solve in the progress and completion words
  // batched Solves (one table entry per sequence, constant for the whole Solve; the launch number comes as a kernel argument):
  LmState* st2[2];      // both state buffers: launch `seq` reads st2[seq & 1], writes st2[(seq + 1) & 1]
  double* part2[2];     // idem for the partial sums
  int min_level;        // lm_coarse_kernel: levels >= min_level run inside the workgroup (n_levels = none)
  int fine_lo;          // lm_fine_kernel_batch: the persistent launch takes levels [fine_lo, min_level) (>= min_level: none)
  unsigned long long* xbuf;  // ... and exchanges this sequence's partial rows through this buffer (kFineXbufWords words)
  unsigned fine_dispatch;    // process-wide number of this persistent launch (g_lm_fine_dispatch; epochs are per optimiser and may coincide)
  unsigned fine_epoch;       // tags of the exchange: (fine_epoch << 8) + evaluation; the host never repeats an epoch on a buffer it has not cleared
  unsigned fine_wait;        // bound of one wait of the persistent launch in wall-clock ticks (0: kFineWaitTicks)
  int fine_home;             // the XCC id of the XCD this optimiser's persistent launch runs on (fine_on_home; < 0: class 0 wherever it lands)
  // Chained Solves (the tracker's next Solve queued BEHIND the one in flight, before its result exists: lm_chain_begin). The launch
  // that finishes a Solve leaves its pose and a guard word in device memory; a chained Solve starts from that pose — what Reset
  // would have handed it (ref: run_odometry_kitti_offline.cpp:261,268) — provided the guard says the Solve it follows succeeded and
  // the runner's keyframe test (ref: :253-258) keeps the keyframe; otherwise every launch of it returns at once.
  float* chain_pose;         // [16] column-major pose of the last finished Solve (device memory)
  int* chain_guard;          // (token << 2) | 2 (promote) | 1 (succeeded) of the last finished Solve
  int chain_in_token;        // != 0: this Solve is chained behind the Solve with that token
  int chain_out;             // 1: the finishing launch writes chain_pose / chain_guard (kf_rule: six weights + the threshold)
  float kf_rule[7];
  // hand-over to the unfused pipeline (dense fine levels): the device stops walking the pyramid below stop_level, reports the
  // Solve "finished" there and leaves its state in final_state, from which the host carries on level by level
  int stop_level;       // 0: the fused pipeline covers every level
  LmState* final_state; // where the finishing launch copies the state (NULL: nowhere)
  // bench.py roofline: a sampled launch records its own execution span — span[0] = min over blocks of the device wall clock
  // (100 MHz) at entry, span[1] = max at exit; NULL: not sampled (two fire-and-forget device-scope atomics per block when on)
  unsigned long long* span;
  // An ARMED coarse launch (lm_coarse_armed_kernel; the tracker's next Solve, queued behind the Solve in flight before that one has
  // returned — lm_arm_begin): host-mapped, 17 data-tagged granules {value, tag = token} that the HOST writes once that Solve has
  // returned and the runner's keyframe test is done: [0..15] the initial pose (what Reset hands the Solve, ref:
  // run_odometry_kitti_offline.cpp:261,268), [16] 1 = go, 2 = return at once. Neither the host's launch call nor the dispatch lies
  // between two Solves: the launch starts the moment the persistent launch in front of it retires, and finds its word waiting. NULL: not armed.
  unsigned long long* arm;
};
// What changes from launch to launch of one Solve.
struct StepLaunch {
  const LmState* st_in; LmState* st_out;
  const double* part_in; double* part_out;
  int seq, first_of_solve;
  unsigned long long* span;
};
// The level table of a launch's StepArgs into LDS, by all threads at once (a dword each). `src` = the StepArgs itself, whose first member
// the table is: the kernarg segment of a single launch (lm_kernarg_words: the StepArgs is the kernel's first parameter) or the
// sequence's entry of a batched launch's table. Thread 0 copying it member by member went through scalar loads in rounds of what the
// scalar registers hold — three to four dependent trips to device memory, 4.1 us of the coarse launch's 5.8 us in front of its first
// evaluation (profiles/r06_relaunch_path.md).
static_assert(offsetof(StepArgs, lv) == 0, "lm_copy_levels reads the level table at offset 0 of the StepArgs");
__device__ __forceinline__ const unsigned* lm_kernarg_words() { return (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ void lm_copy_levels(StepLevel* lv_sh, const unsigned* __restrict__ src) {
  constexpr int kWords = (int)(sizeof(StepLevel) * ODO_MAX_LEVELS_K / sizeof(unsigned));
  for (int i = (int)threadIdx.x; i < kWords; i += (int)blockDim.x) ((unsigned*)lv_sh)[i] = src[i];
}
// In a batched launch (gridDim.y sequences) one block in eight takes part, a different residue per sequence: several hundred
// atomics on one address serialise in the L2 (~10 ns each: ~4 us for the 424 blocks of an S = 8 step launch, on a 16 us kernel).
__device__ __forceinline__ bool lm_span_block() {
  return threadIdx.x == 0 && (gridDim.y == 1 || ((blockIdx.x ^ blockIdx.y) & 7u) == 0u);
}
__device__ __forceinline__ void lm_span_begin(unsigned long long* span) {
  if (span && lm_span_block()) atomicMin(span, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void lm_span_end(unsigned long long* span) {
  if (span && lm_span_block()) atomicMax(span + 1, (unsigned long long)wall_clock64());
}

constexpr int kFoldChunk = 20;  // the 160 rows of a point-list grid in one round of loads per segment

// The LM state machine of one evaluation, run by wave 0 of a block (all threads of the block must call it; it ends
// with a block barrier): accept / reject, the 6x6 solve across the wavefront, exp / compose, the trace row, and — when
// `lv` is given — the walk down the pyramid (ref: src/lm_optimizer.cpp:92,110-115,123-156).
// Every lane of wave 0 carries the whole 64-dword state in registers and runs the scalar code redundantly on
// wave-uniform values: one LDS read of the state and one write-back, no LDS round trip inside decide / apply / walk and
// nothing to broadcast (the solve returns the step in every lane).
// pending: acc_sh holds the 29 sums of an evaluation at s_sh.T that has not been consumed yet.
__device__ __forceinline__ void lm_state_machine(bool pending, const StepLevel* lv, int n_levels, float lambda0,
                                                 float precision, LmState& s_sh, double* acc_sh, LmTraceRow* __restrict__ trace,
                                                 float* __restrict__ cost_stat, bool publisher,
                                                 unsigned long long* smdbg = nullptr, int stop_level = 0) {
  const int t = threadIdx.x;
  if (t < 64) {
    unsigned long long c0 = smdbg ? __builtin_readcyclecounter() : 0, c1 = c0, c2 = c0, c3 = c0;
    LmState s = s_sh;
    const int iter0 = s.iter, lvl0 = s.level;
    const float err_last0 = s.err_last;
    bool need_step = false;
    if (pending) need_step = lm_decide(&s, acc_sh, precision);
    if (smdbg) { c1 = __builtin_readcyclecounter(); c2 = c1; c3 = c1; }
    if (need_step) {
      float d[6];
      solve_damped_wave_regs(acc_sh, s.lambda, d);
#pragma unroll
      for (int i = 0; i < 6; i++) s.delta[i] = d[i];
      if (smdbg) { asm volatile("" ::"v"(d[0]), "v"(d[5])); c2 = __builtin_readcyclecounter(); }
#ifdef ODO_NO_LANE_SINCOS
      lm_apply_step(&s, s.max_iters);
#else
      lm_apply_step_wave(&s, s.max_iters);
#endif
      if (smdbg) { asm volatile("" ::"v"(s.T[0]), "v"(s.T[14])); c3 = __builtin_readcyclecounter(); }
    }
    if (smdbg && t == 0) { smdbg[0] += c1 - c0; smdbg[1] += c2 - c1; smdbg[2] += c3 - c2; smdbg[3] += 1; }
    if (pending) {
      // trace == NULL: an optimiser whose per-evaluation rows and per-level cost statistics nobody reads (the trackers' own):
      // the ~80 instructions of the row are then not on the critical wave (~600 cycles per evaluation of the coarse kernel)
      if (publisher && t == 0 && trace) {
        const int ev = s.n_evals - 1;
        if (ev < kTraceCap) {
          LmTraceRow& r = trace[ev];
          r.level = lvl0;
          r.iter = iter0;
          r.n_res = (int)acc_sh[28];
          r.err = s.err_now;
          r.accepted = (s.status == 0 && !(s.err_now > err_last0)) ? 1 : 0;
          r.stop = (s.stop_reason == 3 || s.active) ? 0 : s.stop_reason;
          r.lambda_after = s.lambda;
#pragma unroll
          for (int i = 0; i < 6; i++) r.delta[i] = (s.active || s.stop_reason == 3) ? s.delta[i] : 0.0f;
        }
        int evals_this_level = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) evals_this_level += (i == (lvl0 & 7)) ? s.iters_level[i] : 0;
        if (iter0 == 0 && evals_this_level == 1) cost_stat[lvl0 * 2 + 0] = s.err_now;
        cost_stat[lvl0 * 2 + 1] = s.err_now;
      }
      s.pending = 0;
    }
    if (lv) {
      while (!s.active && s.status == 0 && !s.finished) {
        const int next = (s.level < 0) ? n_levels - 1 : s.level - 1;
        if (next < stop_level) { s.finished = 1; break; }   // stop_level > 0: the levels below are the host's (unfused pipeline)
        s.stop_reason = 0;
        lm_begin_level(&s, next, lambda0, lv[next].max_iters);  // ref: src/lm_optimizer.cpp:110-115
      }
      if (s.status != 0) s.finished = 1;
    }
    if (t == 0) s_sh = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// The same state machine for the kernels that run MANY evaluations per launch (lm_coarse_kernel, lm_fine_kernel): wave 0 keeps the
// part of the state an evaluation reads and writes in registers from one evaluation to the next (LmHot, wave-uniform) instead of
// copying all 64 dwords out of LDS and back every time, and only what the other waves read goes to LDS per evaluation: T, level,
// active, status, finished. Same operations on the same values as lm_decide / solve_damped / lm_apply_step / lm_begin_level:
//   * `last` is not carried: every path through :131-143 leaves last == cur (accept: last = cur = inc; reject: cur = last), so the
//     reject branch copies nothing and lm_hot_store writes cur into both;
//   * cur.matrix() — the C of se3_left_update, :153 — is not recomputed from the quaternion every evaluation: it IS the T of the
//     evaluation in which cur was the candidate (se3_to_colmajor of the same seven floats), taken over on accept (Tc); a level's
//     first T = matrix(inc = cur) is Tc as well (:115).
// Round 5: the single wave that runs this issues one instruction per >= 4 cycles whatever its EXEC mask
// (tools/microbench/exec_mask_rates.hip), so the state machine's time is its instruction count.
// ---------------------------------------------------------------------------------------------------------------
struct LmHot {
  Se3 cur, inc;
  float Tc[12];   // cur.matrix(): rows 0..2 of columns 0..3 (column-major, as in LmState::T without its constant bottom row)
  float lambda, err_last;
  int level, iter, max_iters, active, status, finished, n_evals, stop_reason;
};
__device__ __forceinline__ void lm_hot_load(LmHot& h, const LmState& s) {   // wave 0, every lane
  h.cur = s.cur; h.inc = s.inc;
  float M[16];
  se3_to_colmajor(s.cur, M);
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) h.Tc[c * 3 + r] = M[c * 4 + r];
  h.lambda = s.lambda; h.err_last = s.err_last;
  h.level = s.level; h.iter = s.iter; h.max_iters = s.max_iters; h.active = s.active; h.status = s.status; h.finished = s.finished;
  h.n_evals = s.n_evals; h.stop_reason = s.stop_reason;
}
__device__ __forceinline__ void lm_hot_store(const LmHot& h, LmState& s) {  // one lane
  s.cur = h.cur; s.inc = h.inc; s.last = h.cur;
  s.lambda = h.lambda; s.err_last = h.err_last;
  s.level = h.level; s.iter = h.iter; s.max_iters = h.max_iters; s.active = h.active; s.status = h.status; s.finished = h.finished;
  s.n_evals = h.n_evals; s.stop_reason = h.stop_reason;
  s.pending = 0;
}
// All threads of the block call it (it ends with a block barrier); acc_sh holds the 29 sums of the evaluation at s_sh.T.
__device__ __forceinline__ void lm_state_machine_hot(LmHot& h, const StepLevel* lv, int n_levels, float lambda0, float precision,
                                                     LmState& s_sh, const double* acc_sh, LmTraceRow* __restrict__ trace,
                                                     float* __restrict__ cost_stat, bool publisher,
                                                     unsigned long long* smdbg = nullptr, int stop_level = 0) {
  const int t = threadIdx.x;
  if (t < 64) {
    unsigned long long c0 = smdbg ? __builtin_readcyclecounter() : 0, c1 = c0, c2 = c0, c3 = c0;
    const int iter0 = h.iter, lvl0 = h.level;
    const float err_last0 = h.err_last;
    // the T of the evaluation being consumed = matrix(inc): cur's on accept (read ahead of the decision: one LDS trip beside the sums')
    float Tprev[12];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 3; r++) Tprev[c * 3 + r] = s_sh.T[c * 4 + r];
    // ---- lm_decide (odo_math.h; ref: src/lm_optimizer.cpp:123-143) ----
    h.n_evals++;
    if (t == 0) s_sh.iters_level[lvl0 & 7] += 1;
    bool need_step = false;
    float err_now = 0.0f;
    bool have_err = false;
    if (!(acc_sh[28] > 0.0)) {                           // :244-248 -> :123-126
      h.status = -1;
      h.active = 0;
    } else {
      err_now = (float)(acc_sh[27] / acc_sh[28]);        // :129
      have_err = true;
      if (err_now > h.err_last) {                        // :131
        h.lambda = h.lambda * 5.0f;
        if (h.lambda > 1e+5f) { h.active = 0; h.stop_reason = 2; }
        else need_step = true;                           // (cur = last: they are equal)
      } else {
        h.cur = h.inc;                                   // (and last = cur)
#pragma unroll
        for (int i = 0; i < 12; i++) h.Tc[i] = Tprev[i];
        const float err_diff = err_now / h.err_last;
        if (err_diff > precision) { h.active = 0; h.stop_reason = 1; }
        else {
          h.err_last = err_now;
          h.lambda = fmaxf(h.lambda / 5.0f, 1e-5f);
          need_step = true;
        }
      }
    }
    if (smdbg) { c1 = __builtin_readcyclecounter(); c2 = c1; c3 = c1; }
    float d[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool new_T = false;
    float Tn[16];
#pragma unroll
    for (int i = 0; i < 16; i++) Tn[i] = 0.0f;
    if (need_step) {
      solve_damped_wave_regs(acc_sh, h.lambda, d);       // :145-151
      if (smdbg) { asm volatile("" ::"v"(d[0]), "v"(d[5])); c2 = __builtin_readcyclecounter(); }
      Se3 dd;
      se3_exp_wave(d, &dd);                              // :152
      const float C[16] = {h.Tc[0], h.Tc[1], h.Tc[2], 0.0f, h.Tc[3], h.Tc[4], h.Tc[5], 0.0f,
                           h.Tc[6], h.Tc[7], h.Tc[8], 0.0f, h.Tc[9], h.Tc[10], h.Tc[11], 1.0f};
      se3_left_update_mat(dd, C, &h.inc);                // :153
      se3_to_colmajor(h.inc, Tn);
      new_T = true;
      h.iter++;                                          // :154
      if (!(h.max_iters > h.iter)) { h.active = 0; h.stop_reason = 3; }
      if (smdbg) { asm volatile("" ::"v"(Tn[0]), "v"(Tn[14])); c3 = __builtin_readcyclecounter(); }
    }
    if (smdbg && t == 0) { smdbg[0] += c1 - c0; smdbg[1] += c2 - c1; smdbg[2] += c3 - c2; smdbg[3] += 1; }
    if (t == 0) {
      if (have_err) s_sh.err_now = err_now;
      if (need_step) {
#pragma unroll
        for (int i = 0; i < 6; i++) s_sh.delta[i] = d[i];
      }
    }
    if (publisher && t == 0 && trace) {   // (see lm_state_machine)
      const int ev = h.n_evals - 1;
      const float e = s_sh.err_now;
      if (ev < kTraceCap) {
        LmTraceRow& r = trace[ev];
        r.level = lvl0;
        r.iter = iter0;
        r.n_res = (int)acc_sh[28];
        r.err = e;
        r.accepted = (h.status == 0 && !(e > err_last0)) ? 1 : 0;
        r.stop = (h.stop_reason == 3 || h.active) ? 0 : h.stop_reason;
        r.lambda_after = h.lambda;
#pragma unroll
        for (int i = 0; i < 6; i++) r.delta[i] = (h.active || h.stop_reason == 3) ? d[i] : 0.0f;
      }
      const int evals_this_level = s_sh.iters_level[lvl0 & 7];
      if (iter0 == 0 && evals_this_level == 1) cost_stat[lvl0 * 2 + 0] = e;
      cost_stat[lvl0 * 2 + 1] = e;
    }
    if (lv) {
      while (!h.active && h.status == 0 && !h.finished) {
        const int next = (h.level < 0) ? n_levels - 1 : h.level - 1;
        if (next < stop_level) { h.finished = 1; break; }
        h.stop_reason = 0;
        // lm_begin_level (ref: src/lm_optimizer.cpp:110-115)
        h.level = next;
        h.iter = 0;
        h.err_last = 1e+10f;
        h.lambda = lambda0;
        h.inc = h.cur;
        h.max_iters = lv[next].max_iters;
        h.active = (h.status == 0 && h.max_iters > 0) ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
          for (int r = 0; r < 3; r++) Tn[c * 4 + r] = h.Tc[c * 3 + r];
          Tn[c * 4 + 3] = (c == 3) ? 1.0f : 0.0f;
        }
        new_T = true;
      }
      if (h.status != 0) h.finished = 1;
    }
    if (t == 0) {   // what the other waves read
      if (new_T) {
#pragma unroll
        for (int i = 0; i < 16; i++) s_sh.T[i] = Tn[i];
      }
      s_sh.level = h.level; s_sh.active = h.active; s_sh.status = h.status; s_sh.finished = h.finished;
    }
  }
  __syncthreads();
}

// Sums the per-block partials in a fixed order and advances the LM state machine by one evaluation
// (accept / reject, damped 6x6 solve, exp, left-compose; ref: src/lm_optimizer.cpp:129-154). One workgroup:
// all threads fold the partials, wave 0 runs lm_state_machine.
// STAMP builds (diagnostic entry odo_debug_update_stamps only) record s_memtime at phase boundaries into `stamps`.
constexpr int kUpdThreads = 1024;
#define ODO_STAMP(i) do { if (STAMP && threadIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
template <bool STAMP>
__device__ __forceinline__ void lm_update_body(LmState* __restrict__ st, const double* __restrict__ partials,
                                               int nblk, int expect_level, float precision, int max_iters,
                                               LmTraceRow* __restrict__ trace, float* __restrict__ cost_stat,
                                               int* __restrict__ host_prog, int seq,
                                               unsigned long long* __restrict__ stamps) {
  ODO_STAMP(0);
  if (!(st->active != 0 && st->level == expect_level)) {
    // stale launch (the level's loop already stopped): only report progress to the polling host
    if (host_prog && threadIdx.x == 0) __hip_atomic_store(host_prog, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __shared__ double sh[32][32];
  __shared__ double acc_sh[32];
  __shared__ LmState s_sh;  // LDS copy: dynamic indexing (iters_level[level]) stays out of scratch memory
  const int t = threadIdx.x;
  const int q = t & 31, seg = t >> 5;
  // fold the per-block partials: 32 segments x 29 quantities, every load of a thread in flight at once (<= 5 for the
  // 160-block point-list grids), then a fixed-order combine — the association order depends only on nblk, so
  // results are run-to-run identical.
  double v = 0.0;
  if (q < ODO_NACC) {
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int b = seg;
    for (; b + 96 < nblk; b += 128) {
      v0 += partials[(size_t)b * ODO_NACC + q];
      v1 += partials[(size_t)(b + 32) * ODO_NACC + q];
      v2 += partials[(size_t)(b + 64) * ODO_NACC + q];
      v3 += partials[(size_t)(b + 96) * ODO_NACC + q];
    }
    for (; b < nblk; b += 32) v0 += partials[(size_t)b * ODO_NACC + q];
    v = (v0 + v1) + (v2 + v3);
  }
  ODO_STAMP(1);
  sh[seg][q] = v;
  if (t < (int)(sizeof(LmState) / sizeof(int))) ((int*)&s_sh)[t] = ((const int*)st)[t];  // cooperative state copy
  __syncthreads();
  if (t < ODO_NACC) {
    double a = 0.0;
#pragma unroll
    for (int g = 0; g < 32; g++) a += sh[g][t];
    acc_sh[t] = a;
  }
  if (t == 0) s_sh.max_iters = max_iters;   // (what lm_begin_level stored: the argument is the same number)
  __syncthreads();
  ODO_STAMP(2);
  // wave 0: accept / reject, the 6x6 solve across the wavefront, exp / compose, trace row, cost statistics — the state
  // machine of the fused kernels (every lane carries the state; the four sin / cos chains in four lane groups)
  lm_state_machine(true, nullptr, 0, 0.0f, precision, s_sh, acc_sh, trace, cost_stat, true);
  ODO_STAMP(3); ODO_STAMP(4); ODO_STAMP(5); ODO_STAMP(6);
  if (t < (int)(sizeof(LmState) / sizeof(int))) ((int*)st)[t] = ((const int*)&s_sh)[t];  // cooperative write-back
  if (t == 0) {
    const LmState& s = s_sh;
    if (host_prog) {
      // host-mapped progress words: [2 + level] = 1 once the level's loop has stopped (the host then skips the
      // launches it has not issued yet), [0] = sequence number of the last update launch that has run.
      if (!s.active) __hip_atomic_store(host_prog + 2 + expect_level, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_prog, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  ODO_STAMP(7);
}
template <bool STAMP>
ODO_KERNEL_T void __launch_bounds__(kUpdThreads) lm_update_kernel(LmState* __restrict__ st, const double* __restrict__ partials,
                                                         int nblk, int expect_level, float precision, int max_iters,
                                                         LmTraceRow* __restrict__ trace, float* __restrict__ cost_stat,
                                                         int* __restrict__ host_prog, int seq,
                                                         unsigned long long* __restrict__ stamps) {
  lm_update_body<STAMP>(st, partials, nblk, expect_level, precision, max_iters, trace, cost_stat, host_prog, seq, stamps);
}
// The unfused pipeline for several streams at once (dense levels of a batched Solve): one table entry per stream, constant for
// a pyramid level; blockIdx.x = stream. Each block is the stream's own single-stream launch.
struct UpdItem {
  LmState* st;
  const double* partials;
  int nblk, expect_level;
  float precision;
  int max_iters;
  LmTraceRow* trace;
  float* cost_stat;
  int* host_prog;
  float lambda0;
  const float* init;   // lm_begin_solve_batch_kernel: affine_init_, column-major, device memory
  float* out;          // lm_finalize_batch_kernel: 26 result floats, device memory
};
ODO_KERNEL void __launch_bounds__(kUpdThreads) lm_update_batch_kernel(const UpdItem* __restrict__ items, int seq) {
  const UpdItem& q = items[blockIdx.x];
  lm_update_body<false>(q.st, q.partials, q.nblk, q.expect_level, q.precision, q.max_iters, q.trace, q.cost_stat, q.host_prog, seq,
                        nullptr);
}

// Prologue shared by the step kernel and the finalize kernel: leaves the advanced state in s_sh.
// fold_sh: >= 8 x 32 doubles of scratch. All 256 threads must call it. `lv` non-null: also walk the pyramid —
// when the level's loop has ended, begin the next coarser-to-finer level right here (ref: src/lm_optimizer.cpp:92,
// 110-115,156), so the launch that learns "level l is done" is also the first evaluation of level l-1.
__device__ __forceinline__ void lm_fused_prologue(const LmState* __restrict__ st_in, const double* __restrict__ part_in,
                                                  const StepLevel* lv, int n_levels, float lambda0, float precision,
                                                  LmState& s_sh, double* fold_sh, double* acc_sh,
                                                  LmTraceRow* __restrict__ trace, float* __restrict__ cost_stat, bool publisher,
                                                  const float* init /* non-null: first launch of a Solve */, int stop_level = 0,
                                                  const float* __restrict__ init_dev = nullptr /* a chained Solve: the pose the Solve before it left in device memory */) {
  const int t = threadIdx.x;
  if (init) {
    if (t == 0) {
      float m[16];
      // (the device-side pose through atomic loads: plain loads would let the compiler fold the two sources into ONE pointer —
      //  kernel-argument address or global — and a by-value kernel argument whose address is taken that way is mirrored in scratch
      //  memory: 1 032 B per lane in lm_step_kernel, launches 7.8 -> 18 us)
      if (init_dev) { for (int i = 0; i < 16; i++) m[i] = __hip_atomic_load(init_dev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else { for (int i = 0; i < 16; i++) m[i] = init[i]; }
      lm_begin_solve(&s_sh, m);
      s_sh.level = -1; s_sh.iter = 0; s_sh.lambda = 0.0f; s_sh.err_last = 1e+10f;
      for (int i = 0; i < 16; i++) s_sh.T[i] = m[i];
    }
    if (publisher && t < 16 && cost_stat) cost_stat[t] = 0.0f;
  } else if (t < (int)(sizeof(LmState) / sizeof(int))) {
    ((int*)&s_sh)[t] = ((const int*)st_in)[t];
  }
  __syncthreads();
  // (Prefetching this thread's keyframe point here, to overlap its latency with the fold and the solve, was measured
  // and is slower — 7.15 vs 6.98 us per launch: the extra loads compete with the partial rows.)
  const bool pending = s_sh.pending != 0;  // block-uniform
  if (pending) {
    // (Loading the partial rows speculatively, together with the state, was measured and is slower: at the coarse
    // levels it fetches ~100 rows that are not needed — 9.5 vs 8.9 us per launch.)
    const int nblk = s_sh.pending_nblk;
    const int q = t & 31, seg = t >> 5;
    double v = 0.0;
    if (q < ODO_NACC && seg < 8) {  // 8 segments of 32 lanes fold; wider blocks (coarse kernel) leave the rest idle
      for (int b0 = seg; b0 < nblk; b0 += 8 * kFoldChunk) {
        double r[kFoldChunk];
#pragma unroll
        for (int u = 0; u < kFoldChunk; u++) {
          const int b = b0 + 8 * u;
          r[u] = (b < nblk) ? part_in[(size_t)b * ODO_NACC + q] : 0.0;  // all loads of the chunk in flight together
        }
#pragma unroll
        for (int u = 0; u < kFoldChunk; u++) v += r[u];
      }
    }
    if (seg < 8) fold_sh[seg * 32 + q] = v;
    __syncthreads();
    if (t < ODO_NACC) {
      double a = 0.0;
#pragma unroll
      for (int g = 0; g < 8; g++) a += fold_sh[g * 32 + t];
      acc_sh[t] = a;
    }
    __syncthreads();
  }
  lm_state_machine(pending, lv, n_levels, lambda0, precision, s_sh, acc_sh, trace, cost_stat, publisher, nullptr, stop_level);
}

// End of a Solve: affine_ = current_estimate.matrix() (ref: src/lm_optimizer.cpp:158) or the pseudo-identity on failure
// (ref: :48-52,60-65), status, counters and cost statistics into host-mapped memory, then the completion word.
// One thread. out: 16 pose (column-major), status, n_evals, 8 evaluations per level, 16 cost statistics.
// chain (optional): the arguments of the Solve, for the hand-over to a chained Solve (chain_out); word 43 of the result block then
// tells the host what the guard says: (token << 2) | 1 = a Solve chained behind this one runs, | 2 = its launches return at once
// (failure, or the keyframe test fires).
struct ChainOut { float* pose; int* guard; int on; float w0, w1, w2, w3, w4, w5, th; };
__device__ __forceinline__ ChainOut chain_out_of(const StepArgs& a) {   // (by value, field by field: the address of a by-value kernel
  return ChainOut{a.chain_pose, a.chain_guard, a.chain_out, a.kf_rule[0], a.kf_rule[1], a.kf_rule[2], a.kf_rule[3], a.kf_rule[4],   // argument would mirror all of it in scratch memory)
                  a.kf_rule[5], a.kf_rule[6]};
}
__device__ __forceinline__ void lm_write_result(const LmState& s, const float* cost_stat, float* __restrict__ out,
                                                int* __restrict__ done_flag, int token, const ChainOut chain = ChainOut{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0}) {
  float m[16];
  if (s.status == 0) {
    se3_to_colmajor(s.cur, m);
  } else {
    for (int i = 0; i < 16; i++) m[i] = 0.0f;
    m[0] = 1.0f; m[5] = 1.0f; m[10] = 1.0f;
  }
  for (int i = 0; i < 16; i++) out[i] = m[i];
  out[16] = (float)s.status;
  out[17] = (float)s.n_evals;
  for (int i = 0; i < 8; i++) out[18 + i] = (float)s.iters_level[i];
  if (cost_stat) { for (int i = 0; i < 16; i++) out[26 + i] = cost_stat[i]; }   // (an optimiser that records nothing: the host fills in the zeros —
                                                                                //  sixteen stores to host memory less in front of the completion word)
  __hip_atomic_store(done_flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the host has the result from here on
  if (chain.on) {
    // ... and the Solve queued behind this launch gets its start: pose, then the guard (the launch ends behind these stores). The
    // verdict goes to the host as well, tagged with the token (word 43 of the result block, written last).
    for (int i = 0; i < 16; i++) chain.pose[i] = m[i];
    const float w[6] = {chain.w0, chain.w1, chain.w2, chain.w3, chain.w4, chain.w5};
    // (a bound, not the test itself — the host's motion_magnitude decides, ref: :258: within ODO_MOTION_SLACK of the threshold the
    //  chained Solve is told to return, and the host starts the next Solve the ordinary way if its own test keeps the keyframe)
    const bool promote = (s.status != 0) || !(motion_magnitude_approx(m, w) < chain.th - ODO_MOTION_SLACK);
    __hip_atomic_store(chain.guard, (token << 2) | (promote ? 2 : 0) | 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((int*)out + 43, (token << 2) | (promote ? 2 : 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// A chained Solve's launches look at the guard first (every thread; the word is final: the Solve it belongs to has ended).
__device__ __forceinline__ bool lm_chain_skip(const StepArgs& a) {
  if (a.chain_in_token == 0) return false;
  return __hip_atomic_load(a.chain_guard, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ((a.chain_in_token << 2) | 1);
}

constexpr int kProgSeqBits = 12;  // progress word = (token << 12) | launches finished: stale launches of an earlier Solve
                                  // that drain after the host has moved on cannot be mistaken for this Solve's progress
__device__ __forceinline__ void lm_fused_publish(LmState& s_sh, LmState* __restrict__ st_out, int* __restrict__ host_prog,
                                                 int seq, int token, const float* cost_stat, float* __restrict__ out,
                                                 int* __restrict__ done_flag, LmState* __restrict__ final_state = nullptr,
                                                 const ChainOut chain = ChainOut{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0}) {
  const int t = threadIdx.x;
  if (t < 64) {
    if (t < (int)(sizeof(LmState) / sizeof(int))) ((int*)st_out)[t] = ((const int*)&s_sh)[t];
    if (final_state && s_sh.finished && t < (int)(sizeof(LmState) / sizeof(int))) ((int*)final_state)[t] = ((const int*)&s_sh)[t];
    if (t == 0) {
      // the launch that learns that every level is done hands the result to the host itself (no finalize launch)
      if (s_sh.finished && out) lm_write_result(s_sh, cost_stat, out, done_flag, token, chain);
      if (host_prog) {
        // host-mapped progress: [1] = token once every level is done (the host stops issuing launches),
        // [0] = number of launches of this Solve that have finished
        if (s_sh.finished) __hip_atomic_store(host_prog + 1, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_prog, (token << kProgSeqBits) | (seq + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// One generic LM step. Which level it works on is decided on the device (the prologue walks the pyramid), so the host
// issues identical launches until the device reports that the Solve is finished; the grid is sized for the largest
// level and the blocks a coarser level does not need stop after the (redundant, parallel) prologue.
__device__ __forceinline__ void lm_step_body(const StepArgs& a, const StepLaunch& q) {
  if (lm_chain_skip(a)) return;
  __builtin_amdgcn_s_setprio(3);  // see lm_coarse_kernel
  lm_span_begin(q.span);
  __shared__ LmState s_sh;
  __shared__ double fold_sh[8 * 32];
  __shared__ double acc_sh[32];
  // The last block of the grid evaluates no points: it publishes the state, the trace row and the host progress word
  // (a system-scope release, ~0.5 us) while the other blocks are still evaluating.
  const bool publisher = (blockIdx.x == gridDim.x - 1);
  if (ODO_DBG(a) && publisher && threadIdx.x == 0 && q.seq < 56) ODO_DBG(a)[16 + 2 * q.seq] = wall_clock64();  // diagnostic timeline
  lm_fused_prologue(q.st_in, q.part_in, a.lv, a.n_levels, a.lambda0, a.precision, s_sh, fold_sh, acc_sh, a.trace, a.cost_stat,
                    publisher, q.first_of_solve ? a.init : nullptr, a.stop_level);   // (a chained Solve has no step launches: lm_chain_begin)
  const bool run = (s_sh.active != 0 && s_sh.status == 0);  // block-uniform
  const int lvl = run ? s_sh.level : 0;
  const StepLevel& L = a.lv[lvl];
  if (run && !publisher && (int)blockIdx.x < L.nblk) {
    __shared__ float rows_sh[kRowFloats * RowBuf<kLmBlock>::W];  // the block's point rows (see rows_store)
    constexpr int kS = 8;  // sub-lanes per quantity: 29 x 8 = 232 accumulating threads
    const int my_q = threadIdx.x / kS, my_s = threadIdx.x % kS;
    int rowA = 0, rowB = 0;
    if (my_q < ODO_NACC) rows_of_quantity(my_q, &rowA, &rowB);
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; i++) T[i] = s_sh.T[i];
    double accq = 0.0;
    const int first = blockIdx.x * kLmBlock;
    for (int base = first; base < L.n; base += L.nblk * kLmBlock) {  // one round unless the level has > 160 x 256 points
      const int idx = base + threadIdx.x;
      float r = 0.0f, w = 0.0f, J[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      bool valid = false;
      if (idx < L.n) {
        const PointK p = load_point(L.pl, idx);
        if (point_residual(p, T, L.k, L.I2, L.rows, L.cols, &r, J)) {
          w = robust_weight(r, a.robust, a.huber_delta, 1.0f);
          valid = true;
        }
      }
      if (base > first) __syncthreads();  // the previous round's rows have been consumed
      rows_store(rows_sh, RowBuf<kLmBlock>::W, threadIdx.x, J, w, r, valid);
      __syncthreads();
      if (my_q < ODO_NACC) accq = rows_accumulate<kLmBlock, kS>(rows_sh, rowA, rowB, my_s, accq);
    }
    accq = rows_butterfly8(accq);
    if (my_q < ODO_NACC && my_s == 0) q.part_out[(size_t)blockIdx.x * ODO_NACC + my_q] = accq;
  }
  if (publisher) {
    if (threadIdx.x == 0 && run) { s_sh.pending = 1; s_sh.pending_nblk = L.nblk; }
    __syncthreads();
    lm_fused_publish(s_sh, q.st_out, a.host_prog, q.seq, a.token, a.cost_stat, a.out, a.done_flag, a.final_state, chain_out_of(a));
    if (ODO_DBG(a) && threadIdx.x == 0 && q.seq < 56) ODO_DBG(a)[16 + 2 * q.seq + 1] = wall_clock64();
  }
  lm_span_end(q.span);
}

ODO_KERNEL void __launch_bounds__(kLmBlock) lm_step_kernel(StepArgs a) {
  const StepLaunch q = {a.st_in, a.st_out, a.part_in, a.part_out, a.seq, a.first_of_solve, a.span};
  lm_step_body(a, q);
}
// Several independent Solves in the SAME launches: blockIdx.y picks the sequence's entry of a table in device memory that
// stays constant for the whole Solve; every sequence runs its own state machine and finishes in its own time (the blocks of
// a finished sequence return after the prologue). grid = (largest grid of any sequence, number of sequences).
ODO_KERNEL void __launch_bounds__(kLmBlock) lm_step_kernel_batch(const StepArgs* __restrict__ table, int seq, int first_of_solve,
                                                                 unsigned long long* span) {
  const StepArgs& a = table[blockIdx.y];
  const StepLaunch q = {a.st2[seq & 1], a.st2[(seq + 1) & 1], a.part2[seq & 1], a.part2[(seq + 1) & 1], seq, first_of_solve, span};
  lm_step_body(a, q);
}

// Coarse pyramid levels inside ONE workgroup. A level with a few thousand points does not fill more than a handful of
// CUs, so spreading it over blocks only buys kernel boundaries and trips through L2 for the state and the partial
// sums. Here a single 512-thread workgroup loops evaluate -> reduce (LDS) -> state machine for every level
// >= min_level: no launch, no global partials, no state reload between iterations. It hands over to the generic step
// launches with the next level already begun.
#ifndef ODO_COARSE_BLOCK
#define ODO_COARSE_BLOCK 512
#endif
// 512 threads, two rounds over a level of <= 1024 points. (1024 threads / one round / 32 sub-lanes per accumulated quantity was
// measured in round 2: evaluation 4 100 -> 4 430 cycles, reduction 2 650 -> 3 960 per iteration: sixteen waves on one CU pay
// more at the barriers than the second round costs.)
constexpr int kCoarseBlock = ODO_COARSE_BLOCK;
constexpr int kCoarseLdsBytes = 2 * RowBuf<kLmBlock>::kBytes;  // [2][15][288] floats: point rows of two virtual blocks
constexpr int kCoarseMaxPoints = 1024;  // levels with more points go to the multi-block step kernel (measured: a 512-thread
                                        // workgroup walking ~2000 points four per thread is no faster than seven blocks)

// ComputeScaleNaive inside one workgroup (ref: src/lm_optimizer.cpp:338-358): every thread of the workgroup calls it with the squared
// residuals of its points — one per round of the level (nr = 1 or 2 rounds of 512 points; valid = false: no residual). Chunk = 64
// consecutive points = one wave of one round: chunk (2 rd + half) * 4 + (wave & 3) = rd * 8 + wave, summed in the one order
// (wave_sum64 per chunk — chunks without points are +0 —, a virtual block's four chunks, then the virtual blocks: see fine_tdist_sigma). part: [2][16] doubles, cnt: [16] ints of LDS. One workgroup barrier per pass.
constexpr int kCoarseRounds = kCoarseMaxPoints / kCoarseBlock;   // 2
constexpr int kCoarseChunks = kCoarseRounds * (kCoarseBlock / kWave);   // 16
__device__ __forceinline__ float coarse_tdist_sigma(double (*part)[kCoarseChunks], int* cnt, const float e2[kCoarseRounds],
                                                    const bool valid[kCoarseRounds], int nr) {
  constexpr int kW = kCoarseBlock / kWave;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int my_cnt[kCoarseRounds];
#pragma unroll
  for (int rd = 0; rd < kCoarseRounds; rd++) my_cnt[rd] = __popcll(__ballot(rd < nr && valid[rd]));
  float sigma = 5.0f;
  int n_total = 0;
  for (int pass = 0; pass < kTdistMaxPasses; pass++) {
#pragma unroll
    for (int rd = 0; rd < kCoarseRounds; rd++) {
      const double ws = wave_sum64((rd < nr && valid[rd]) ? (double)tdist_term(e2[rd], sigma * sigma) : 0.0);
      if (lane == 0) { part[pass & 1][rd * kW + wv] = ws; if (pass == 0) cnt[rd * kW + wv] = my_cnt[rd]; }
    }
    __syncthreads();   // (the other parity is still being read by nobody: a wave passes this barrier only after its reads of pass - 1)
    if (pass == 0) {
      int tot = 0;
#pragma unroll
      for (int i = 0; i < kCoarseChunks; i++) tot += cnt[i];
      n_total = tot;
    }
    // (the one order: a virtual block's four chunk sums first — G = ((c0 + c1) + c2) + c3, virtual block v = 2 * round + half —, then
    //  the virtual blocks across lanes)
    double G = 0.0;
    if (lane < kCoarseChunks / 4) {
      const double* pv = &part[pass & 1][(lane >> 1) * kW + 4 * (lane & 1)];
      G = ((pv[0] + pv[1]) + pv[2]) + pv[3];
    }
    const double total = wave_sum64(G);
    const float nxt = (n_total > 0) ? tdist_next_sigma(total, n_total) : sigma;
    const bool done = (n_total == 0) || tdist_converged(nxt, sigma);
    sigma = nxt;
    if (done) break;   // workgroup-uniform: every wave folds the same values in the same order
  }
  return sigma;
}
// kFull = false: the trackers' build — Huber / L2 weights, floor sampling (the parity mode) and no per-evaluation trace rows (an
// optimiser nobody asked to record):
// the t-distribution scale passes with their td_* registers and the trace / cost-statistics writes are compiled out, not branched
// around (+ 1.5 % on the headline, profiles/r06_state_machine_ab.md: what a single latency-bound wave does not execute still costs
// it registers and scheduling freedom). kFull = true: everything decided at run time from StepArgs (robust == 2, trace != NULL).
#if ODO_PHASE_STAMPS
// Diagnostic build: where the wall clock of a Solve goes OUTSIDE the evaluation loops (100 MHz ticks, device memory, thread 0 of the
// publishing workgroup at exit): [0] / [1] exit stamps of the last coarse / fine launch, [4..8] coarse prologue, loop, epilogue, previous
// fine exit -> this entry, launches; [9..13] the same for the fine launch (with coarse exit -> fine entry). Read by lm_chain_diag_read.
ODO_DEVICE_VAR unsigned long long g_lm_diag[24];   // [16..18] coarse prologue: entry -> level table in LDS, -> lm_fused_prologue done, -> hot state loaded
#endif
constexpr unsigned long long kArmWaitTicks = 200000000ull;   // an armed launch waits at most 2 s of the 100 MHz wall clock for its word; then it reports a
                                                             // give-up (status -2: the host redoes the Solve on the step launches, like a persistent launch's)
template <bool kFull, bool kArmed = false>
__device__ __forceinline__ void lm_coarse_body(const StepArgs& a, const StepLaunch& q, int min_level, const unsigned* __restrict__ lv_src) {
#if ODO_PHASE_STAMPS
  const unsigned long long w_entry = (unsigned long long)wall_clock64();
#endif
  if (!kArmed && lm_chain_skip(a)) return;
  // The pose LM is the latency-critical chain of a frame, while the depth stream floods the CUs with throughput work
  // (selection, SSD scan) at the same time: raise this workgroup's issue priority on the SIMDs it shares with them.
  __builtin_amdgcn_s_setprio(3);
  lm_span_begin(q.span);
  __shared__ LmState s_sh;
  extern __shared__ __attribute__((aligned(16))) double red_sh[];  // kCoarseLdsBytes: reduction buffer; its head doubles as the fold scratch
                                                                   // (16-byte aligned whatever static LDS precedes it: the row sums read it with ds_read_b128 —
                                                                   //  68 bytes of static LDS in front of it cost the coarse launch 1 us per iteration)
  __shared__ double acc_sh[32];
  // The level table moves to LDS (static-index copy): every later access indexes it with the level read from the state,
  // and a dynamic index into the by-value kernel argument makes the compiler mirror all of `a` in scratch memory.
  __shared__ StepLevel lv_sh[ODO_MAX_LEVELS_K];
  // ODO_COARSE_STAMPS: the state machine's phase sums stay in LDS and go out ONCE, at exit — round 5 added them to the host-mapped
  // counters from inside the loop (four read-modify-writes over PCIe per iteration), which the "state-machine" lap then measured
  __shared__ unsigned long long sm_sh[4];
  // (an armed launch: the first read of the host's word travels beside the level table's loads)
  unsigned long long arm_g0 = 0;
  if (kArmed && threadIdx.x < 17) arm_g0 = __hip_atomic_load(a.arm + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  lm_copy_levels(lv_sh, lv_src);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) sm_sh[i] = 0;
  }
  __shared__ __attribute__((aligned(16))) float arm_pose_sh[16];
  __shared__ __attribute__((aligned(16))) int arm_verdict_sh4[4];
  if (kArmed) {
    // The first wave reads the host's word (one read of host memory per pass: the trip over PCIe paces the loop). 17 granules, every
    // one tagged with this Solve's token: no flag, no fence. Normally the word is there: the host wrote it while the launch in front
    // of this one was retiring.
    if (threadIdx.x < 64) {
      const int lane = (int)threadIdx.x;
      unsigned long long g = arm_g0;
      bool got = __all(lane >= 17 || (int)(g >> 32) == a.token);
      const unsigned long long t0 = (unsigned long long)wall_clock64();
      for (int spin = 0; !got; spin++) {
        if (lane < 17) g = __hip_atomic_load(a.arm + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        got = __all(lane >= 17 || (int)(g >> 32) == a.token);
        if (!got && (spin & 15) == 15 && (unsigned long long)wall_clock64() - t0 > (a.fine_wait ? (unsigned long long)a.fine_wait : kArmWaitTicks)) break;
      }
      if (lane < 16) arm_pose_sh[lane] = __uint_as_float((unsigned)g);
      if (lane == 16) arm_verdict_sh4[0] = got ? (int)(unsigned)g : 3;
    }
    __syncthreads();
    if (arm_verdict_sh4[0] == 3) {   // nobody answered (a host thread that lost its processor for seconds, a dead process): a give-up, reported
      if (threadIdx.x < 64) {        // like the persistent launch's — the launches queued behind this one find a finished Solve
        if (threadIdx.x < (int)(sizeof(LmState) / sizeof(int))) ((int*)&s_sh)[threadIdx.x] = 0;
        if (threadIdx.x == 0) { s_sh.status = -2; s_sh.active = 0; s_sh.finished = 1; }
      }
      __syncthreads();
      lm_fused_publish(s_sh, q.st_out, a.host_prog, q.seq, a.token, a.cost_stat, a.out, a.done_flag, a.final_state);
      return;
    }
    if (arm_verdict_sh4[0] != 1) return;   // the keyframe changes or the Solve before this one failed: nothing was touched
  } else {
    __syncthreads();
  }
#if ODO_PHASE_STAMPS
  const unsigned long long w_p1 = (unsigned long long)wall_clock64();
#endif
  // state in (or initialised), a pending evaluation of an earlier launch consumed, pyramid walk started
  lm_fused_prologue(q.st_in, q.part_in, lv_sh, a.n_levels, a.lambda0, a.precision, s_sh, red_sh, acc_sh, (kFull ? a.trace : (LmTraceRow*)nullptr), a.cost_stat, true,
                    kArmed ? arm_pose_sh : (q.first_of_solve ? a.init : nullptr), a.stop_level, (!kArmed && a.chain_in_token) ? a.chain_pose : nullptr);
#if ODO_PHASE_STAMPS
  const unsigned long long w_p2 = (unsigned long long)wall_clock64();
#endif
  LmHot hot;                    // wave 0's copy of the state between evaluations (lm_state_machine_hot; the other waves never look at theirs)
  lm_hot_load(hot, s_sh);
  if (ODO_DBG(a) && threadIdx.x == 0 && q.seq < 56) ODO_DBG(a)[16 + 2 * q.seq] = wall_clock64();
  unsigned long long c_eval = 0, c_red = 0, c_sm = 0, c_it = 0, c_last = ODO_DBG(a) ? __builtin_readcyclecounter() : 0;
  const unsigned long long c_begin = c_last, w_begin = ODO_DBG(a) ? (unsigned long long)wall_clock64() : 0;
  auto lap = [&](unsigned long long& sum) {
    if (ODO_DBG(a)) { const unsigned long long now = __builtin_readcyclecounter(); sum += now - c_last; c_last = now; }
  };
  // Accumulation order = the step / fine kernels': virtual blocks of 256 points (29 x 8 sub-lane chains over every eighth row, an
  // 8-lane butterfly), their 232-B partial rows folded segment by segment (lm_fused_prologue's order). A level therefore gives the
  // same sums bit for bit whichever kernel evaluates it — the single tracker sends a 830-point level to the persistent launch,
  // the batched one keeps it here. The two halves of the workgroup work on two virtual blocks at a time.
  static_assert(kCoarseBlock == 2 * kLmBlock, "two 256-point virtual blocks per round");
  constexpr int kS = 8;
  constexpr int kVbMax = (kCoarseMaxPoints + kLmBlock - 1) / kLmBlock;
  __shared__ double part_sh[kVbMax][32];
  __shared__ double sc_part[2][kCoarseChunks];   // t-distribution scale passes: the chunk sums (wave x round), by pass parity
  __shared__ int sc_cnt[kCoarseChunks];
  const int tl = threadIdx.x & (kLmBlock - 1), half = threadIdx.x >> 8;
  float* rows_sh = (float*)red_sh + half * (kRowFloats * RowBuf<kLmBlock>::W);  // [2][15][256 + 8] floats
  const int my_q = tl / kS, my_s = tl % kS;
  int rowA = 0, rowB = 0;
  if (my_q < ODO_NACC) rows_of_quantity(my_q, &rowA, &rowB);
  // the keyframe points of the level stay in registers from iteration to iteration (as in lm_fine_body): one dependent trip to L2 / L1
  // less in every evaluation of a level that is a latency chain of ~200 points on four waves
  PointK cpt[kCoarseRounds];
  bool cpt_ok[kCoarseRounds] = {false, false};
  int cpt_level = -1;
  for (int guard = 0; guard < 4096; guard++) {
    const bool run = (s_sh.active != 0 && s_sh.status == 0 && s_sh.level >= min_level);  // block-uniform
    if (!run) break;
    c_it++;
    const StepLevel& L = lv_sh[s_sh.level];
    const float* I2u = lm_uniform_ptr(L.I2);   // (point_residual_g)
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; i++) T[i] = s_sh.T[i];
    const int nvb = L.nblk;  // = ceil(n / 256), at least 1 (lm_grid_for)
    if (s_sh.level != cpt_level) {
      cpt_level = s_sh.level;
#pragma unroll
      for (int rd = 0; rd < kCoarseRounds; rd++) {
        const int vb = 2 * rd + half, idx = vb * kLmBlock + tl;
        cpt_ok[rd] = vb < nvb && idx < L.n;
        if (cpt_ok[rd]) cpt[rd] = load_point(L.pl, idx);
      }
    }
    // (Staging the 29 KB level image in LDS as well was measured: no faster.)
    // t-distribution weights (ref: src/lm_optimizer.cpp:257-261,338-358) need the scale of ALL residuals of the evaluation before any
    // weight: the residuals of every round are evaluated first and kept (with their Jacobian rows) in registers, then the scale
    // iteration runs over them (coarse_tdist_sigma), then the rounds go through the row sums as usual
    float td_r[kCoarseRounds] = {0.0f, 0.0f}, td_J[kCoarseRounds][6] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
    bool td_valid[kCoarseRounds] = {false, false};
    float td_scale_sqr = 1.0f;
    if ((kFull && a.robust == 2)) {
      float e2[kCoarseRounds];
#pragma unroll
      for (int rd = 0; rd < kCoarseRounds; rd++) {
        if (cpt_ok[rd]) td_valid[rd] = point_residual_g<kFull>(cpt[rd], T, L.k, I2u, L.rows, L.cols, &td_r[rd], td_J[rd]);
        e2[rd] = td_r[rd] * td_r[rd];
      }
      const float sg = coarse_tdist_sigma(sc_part, sc_cnt, e2, td_valid, (nvb + 1) / 2);
      td_scale_sqr = sg * sg;
    }
#pragma unroll
    for (int vb0 = 0; vb0 < 2 * kCoarseRounds; vb0 += 2) {  // one round per 512 points (kCoarseMaxPoints: two)
      if (vb0 >= nvb) break;
      const int vb = vb0 + half;
      float r = 0.0f, w = 0.0f, J[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      bool valid = false;
      if ((kFull && a.robust == 2)) {
        r = td_r[vb0 / 2]; valid = td_valid[vb0 / 2];
#pragma unroll
        for (int i = 0; i < 6; i++) J[i] = td_J[vb0 / 2][i];
        if (valid) w = robust_weight(r, 2, a.huber_delta, td_scale_sqr);
      } else if (cpt_ok[vb0 / 2]) {
        if (point_residual_g<kFull>(cpt[vb0 / 2], T, L.k, I2u, L.rows, L.cols, &r, J)) {
          w = robust_weight(r, kFull ? a.robust : (a.robust == 1 ? 1 : 0), a.huber_delta, 1.0f);
          valid = true;
        }
      }
      if (vb0 > 0) __syncthreads();  // the previous round's rows have been consumed
      rows_store_z(rows_sh, RowBuf<kLmBlock>::W, tl, J, w, r, valid);
      __syncthreads();  // rows visible (and, first round, everyone has read s_sh.T)
      if (vb0 == 0) lap(c_eval);
      double accq = 0.0;
      if (vb < nvb) {   // (uniform over the four waves of a half)
        if (my_q < ODO_NACC) accq = rows_accumulate<kLmBlock, kS>(rows_sh, rowA, rowB, my_s, accq);
        accq = rows_butterfly8(accq);
      }
      if (nvb == 1) {
        // One virtual block (<= 256 points: the usual coarsest level): its partial row IS the level's sums — the fold below would add
        // it to zeros, segment 0 first, then the seven empty segments; the same additions here, by the thread that holds the value,
        // without the two trips through LDS and the two barriers around them.
        if (half == 0 && my_q < ODO_NACC && my_s == 0) {
          double v = 0.0;
          v += accq;              // segment 0: rows 0, 8, ... = row 0
          double acc = 0.0;
          acc += v;
#pragma unroll
          for (int g = 1; g < 8; g++) acc += 0.0;   // segments 1 .. 7 are empty (IEEE: kept as written, -0 + 0 = +0)
          acc_sh[my_q] = acc;
        }
      } else if (vb < nvb && my_q < ODO_NACC && my_s == 0) {
        part_sh[vb][my_q] = accq;
      }
    }
    __syncthreads();
    if (nvb > 1) {  // the fold of lm_fused_prologue: segment seg adds rows seg, seg + 8, ... in ascending order, then the segments in order
      const int fq = threadIdx.x & 31, fseg = threadIdx.x >> 5;
      double* fold_sh = red_sh;  // the rows have been consumed (barrier above)
      if (fseg < 8) {
        double v = 0.0;
        if (fq < ODO_NACC)
          for (int b = fseg; b < nvb; b += 8) v += part_sh[b][fq];
        fold_sh[fseg * 32 + fq] = v;
      }
      __syncthreads();
      if (threadIdx.x < ODO_NACC) {
        double acc = 0.0;
#pragma unroll
        for (int g = 0; g < 8; g++) acc += fold_sh[g * 32 + threadIdx.x];
        acc_sh[threadIdx.x] = acc;
      }
      // (no workgroup barrier here: the 29 sums are written by lanes of wave 0 and read by wave 0 alone — lm_state_machine — and
      //  a wave's LDS accesses stay in order; the other waves wait at the state machine's closing barrier)
    }
    lap(c_red);
    lm_state_machine_hot(hot, lv_sh, a.n_levels, a.lambda0, a.precision, s_sh, acc_sh, (kFull ? a.trace : (LmTraceRow*)nullptr), a.cost_stat, true,
                         ODO_DBG(a) ? sm_sh : nullptr, a.stop_level);
    lap(c_sm);
  }
#if ODO_PHASE_STAMPS
  const unsigned long long w_loop_end = (unsigned long long)wall_clock64();
#endif
  if (threadIdx.x == 0) lm_hot_store(hot, s_sh);   // (read back by wave 0 only, below: a wave's LDS accesses stay in order)
  lm_fused_publish(s_sh, q.st_out, a.host_prog, q.seq, a.token, a.cost_stat, a.out, a.done_flag, a.final_state, chain_out_of(a));
#if ODO_PHASE_STAMPS
  if (ODO_DBG(a) && threadIdx.x == 0) {
    const unsigned long long now = (unsigned long long)wall_clock64();
    g_lm_diag[4] += w_begin - w_entry; g_lm_diag[5] += w_loop_end - w_begin; g_lm_diag[6] += now - w_loop_end;
    g_lm_diag[16] += w_p1 - w_entry; g_lm_diag[17] += w_p2 - w_p1; g_lm_diag[18] += w_begin - w_p2;
    if (g_lm_diag[1] && w_entry > g_lm_diag[1] && w_entry - g_lm_diag[1] < 100000ull) { g_lm_diag[7] += w_entry - g_lm_diag[1]; g_lm_diag[3] += w_entry - g_lm_diag[2]; g_lm_diag[14] += 1; }
    g_lm_diag[0] = now; g_lm_diag[8] += 1;
  }
#endif
  if (ODO_DBG(a) && threadIdx.x == 0) {
    if (q.seq < 56) ODO_DBG(a)[16 + 2 * q.seq + 1] = wall_clock64();
    ODO_DBG(a)[0] += c_eval; ODO_DBG(a)[1] += c_red; ODO_DBG(a)[2] += c_sm; ODO_DBG(a)[3] += c_it;
    ODO_DBG(a)[4] += __builtin_readcyclecounter() - c_begin; ODO_DBG(a)[5] += 1;
    ODO_DBG(a)[6] += (unsigned long long)wall_clock64() - w_begin;   // 100 MHz ticks beside the shader cycles of [4]: the shader clock under this load
#pragma unroll
    for (int i = 0; i < 4; i++) ODO_DBG(a)[8 + i] += sm_sh[i];
  }
  lm_span_end(q.span);
}

#ifdef ODO_LM_CHAIN_TU   // (the chain unit: see the top of this header)
static __global__ void __launch_bounds__(kCoarseBlock) lm_coarse_kernel(StepArgs a, int min_level) {   // Huber / L2, nothing recorded
  const StepLaunch q = {a.st_in, a.st_out, a.part_in, a.part_out, a.seq, a.first_of_solve, a.span};
  lm_coarse_body<false>(a, q, min_level, lm_kernarg_words());
}
static __global__ void __launch_bounds__(kCoarseBlock) lm_coarse_armed_kernel(StepArgs a, int min_level) {   // lm_coarse_kernel waiting for the host's word (StepArgs::arm)
  const StepLaunch q = {a.st_in, a.st_out, a.part_in, a.part_out, a.seq, a.first_of_solve, a.span};
  lm_coarse_body<false, true>(a, q, min_level, lm_kernarg_words());
}
static __global__ void __launch_bounds__(kCoarseBlock) lm_coarse_full_kernel(StepArgs a, int min_level) {   // t-distribution weights and / or trace rows
  const StepLaunch q = {a.st_in, a.st_out, a.part_in, a.part_out, a.seq, a.first_of_solve, a.span};
  lm_coarse_body<true>(a, q, min_level, lm_kernarg_words());
}
#endif  // ODO_LM_CHAIN_TU
// Batched twin (see lm_step_kernel_batch): one workgroup per sequence, each with its own min_level (a sequence without a
// coarse level only initialises its state, begins its first level and publishes).
// (lean = every sequence of the table is a trackers' optimiser — Huber / L2, floor sampling, nothing recorded: see lm_coarse_body)
template <bool kFull>
__device__ __forceinline__ void lm_coarse_batch_entry(const StepArgs* __restrict__ table, int seq, int first_of_solve, unsigned long long* span) {
  const StepArgs& a = table[blockIdx.y];
  const StepLaunch q = {a.st2[seq & 1], a.st2[(seq + 1) & 1], a.part2[seq & 1], a.part2[(seq + 1) & 1], seq, first_of_solve, span};
  lm_coarse_body<kFull>(a, q, a.min_level, (const unsigned*)&table[blockIdx.y]);
}
ODO_KERNEL void __launch_bounds__(kCoarseBlock) lm_coarse_kernel_batch(const StepArgs* __restrict__ table, int seq, int first_of_solve,
                                                                       unsigned long long* span) {
  lm_coarse_batch_entry<false>(table, seq, first_of_solve, span);
}
ODO_KERNEL void __launch_bounds__(kCoarseBlock) lm_coarse_full_kernel_batch(const StepArgs* __restrict__ table, int seq, int first_of_solve,
                                                                            unsigned long long* span) {
  lm_coarse_batch_entry<true>(table, seq, first_of_solve, span);
}

// =============================================================================================
// The fine levels in ONE launch: a persistent kernel whose K workgroups exchange their partial sums through L2.
//
// A step launch per evaluation costs a kernel boundary (1.5-1.8 us), a reload of state and partial rows (two dependent trips to
// L2) and the keyframe point's record (another) every time. Here K workgroups stay resident for all evaluations of the levels the
// coarse kernel does not take. Per evaluation every workgroup (a) evaluates its virtual blocks — virtual block vb is exactly the
// step kernel's block vb: points vb * 256 + t (+ rounds), the same row publication and the same 29 x 8 accumulation, so the 232-B
// partial row of a virtual block is bit for bit the step kernel's — (b) publishes each row as 58 data-tagged 8-byte granules
// {32 bits of payload, 32-bit tag = launch epoch of the buffer and evaluation number} with plain (or agent-scope) stores, (c) gathers ALL
// rows by polling the granules themselves with agent-scope loads — no flag, no fence, one trip through L2 (MI355X guide,
// handoff-1to1: 0.8-1.0 us; tools/microbench/xcd_allgather.hip: 0.8-1.3 us for 4-16 workgroups) — in the fold's own access
// pattern (thread (q, seg) reads rows seg, seg + 8, ... of quantity q and adds them in that order: lm_fused_prologue's fold),
// then (d) runs the state machine redundantly, as every block of a step launch does. Nothing but the rows crosses workgroups.
// Rows are double-buffered by evaluation parity: a workgroup can be at most one evaluation ahead of the slowest (it cannot
// finish gathering evaluation e before every workgroup has published e).
//
// Placement: blocks are dealt round-robin over the 8 XCDs, so the launch has 8 x K blocks of which every eighth takes part
// (they share an XCD and its L2; the others return at once). That is for speed only: the workgroups tell each other their XCC id
// first, and only if all agree do the rows go out as plain stores (they stay in that L2); otherwise as agent-scope stores, which
// are coherent at any placement. Every wait is bounded by the device wall clock (kFineWaitTicks: 4 ms): a workgroup that waits longer gives up
// and the Solve reports status -2 (the host redoes it on the step launches) instead of hanging the device.
// =============================================================================================
constexpr int kFineGran = 2 * ODO_NACC;          // granules per partial row
constexpr int kFineRowsMax = 160;                // = kLmListMaxBlocks (host): partial rows of the largest point-list level
constexpr int kFineChunk = 8;                    // rows a folding thread keeps in flight
constexpr unsigned kFineWaitTicks = 400000u;     // default bound of one wait, in ticks of the 100 MHz wall clock: 4 ms — three orders of
                                                 // magnitude above a normal exchange (~1 us), short enough that a give-up is a hiccup of one
                                                 // frame, not a stall (StepArgs::fine_wait overrides: ODO_LM_FINE_WAIT_US)
// A bounded wait: the clocks are looked at every 32nd poll only (s_memrealtime is a scalar memory operation of ~100 cycles).
// The bound is met when `limit` ticks of the 100 MHz wall clock have passed AND as many SHADER cycles as that time holds at the
// nominal 2.4 GHz: in the first milliseconds of a process (the shader clock ramps up from 95 MHz) and under power capping everything
// a wait is for — another persistent kernel vacating its CUs, the slowest workgroup's evaluation — takes longer in wall-clock time,
// and a fixed wall-clock bound then gives up on launches that are merely slow (seen: one depth job in ~ 400 frames at the start of a
// process against one in 5 700 later — 0.7 ms inside a 20-step measurement). The stretch is capped: at 4 x `limit` of wall-clock
// time the wait is over whatever the cycle counter says — the worst case a host-side timeout has to allow for is 16 ms for the pose
// LM's default (4 ms) and 2 ms for the depth launch's (0.5 ms).
struct FineDeadline {
  unsigned long long t0, c0;
  unsigned limit;
  static __device__ __forceinline__ FineDeadline begin(unsigned limit_ticks) {
    return FineDeadline{(unsigned long long)wall_clock64(), (unsigned long long)__builtin_readcyclecounter(), limit_ticks};
  }
  __device__ __forceinline__ bool expired(int spin) const {
    if ((spin & 31) != 31) return false;
    const unsigned long long dt = (unsigned long long)wall_clock64() - t0;
    if (dt <= (unsigned long long)limit) return false;
    if (dt > 4ull * (unsigned long long)limit) return true;   // (a hard wall-clock cap whatever the shader clock does: 4 x the bound)
    return (unsigned long long)__builtin_readcyclecounter() - c0 > 24ull * (unsigned long long)limit;
  }
};
// ComputeScaleNaive over a DENSE level (ref: src/lm_optimizer.cpp:338-358; up to 2 M residuals): the fixed-point iteration of
// lm_tdist_scale_kernel spread over G <= 128 workgroups that keep their residuals in registers (<= 32 per thread) and meet once per
// pass: every workgroup publishes the fp64 sum of its terms as tagged 8-byte granules (agent scope: the workgroups sit on all XCDs),
// wave 0 of every workgroup gathers the G sums, folds them in a fixed order and hands sigma to its workgroup — every workgroup
// computes the same sigma and the same convergence decision. One pass is one trip through the fabric (~3 us) instead of one
// workgroup walking every residual (70-140 us at 1080p). Sums: per thread in ascending index, wave_sum64, the eight waves of a
// workgroup in order, the workgroups g = lane, lane + 64 in order per lane, wave_sum64 — fixed for a given n (G depends on n only).
// Every wait is bounded (wait_ticks of the 100 MHz clock): a launch whose workgroups cannot all be resident sets *gave_up and the
// single-workgroup kernel, queued behind it with only_if = gave_up, does the level.
constexpr int kTsThreads = 512, kTsPerThread = 32, kTsMaxWg = 128;
constexpr int kTsXbufWords = 2 * kTsMaxWg * 4;   // [pass parity][workgroup]{sum hi | tag, sum lo | tag, count | tag, -}
ODO_KERNEL void __launch_bounds__(kTsThreads) lm_tdist_scale_multi_kernel(const float* __restrict__ res, int n,
                                                                          const LmState* __restrict__ st, int expect_level,
                                                                          float* __restrict__ scale_sqr_out,
                                                                          unsigned long long* __restrict__ xbuf, unsigned epoch,
                                                                          unsigned wait_ticks, int* __restrict__ gave_up, int fault) {
  if (!(st->active != 0 && st->level == expect_level)) return;   // (the state does not change while this launch runs: grid-uniform)
  const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const size_t stride = (size_t)G * kTsThreads;
  float e2[kTsPerThread];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < kTsPerThread; k++) {
    const size_t i = (size_t)g * kTsThreads + t + (size_t)k * stride;
    const float r = (i < (size_t)n) ? res[i] : __builtin_nanf("");
    const bool ok = (r == r);
    e2[k] = ok ? r * r : -1.0f;   // (a squared residual is never negative: < 0 marks "no residual")
    cnt += ok ? 1 : 0;
  }
  __shared__ double wsum[kTsThreads / 64];
  __shared__ double wcnt[kTsThreads / 64];
  __shared__ float sh_sigma;
  __shared__ int sh_done, sh_bail;
  const unsigned wait_limit = wait_ticks ? wait_ticks : 400000u;
  float cur = 5.0f;
  double n_valid = 0.0;
  for (int pass = 0; pass < kTdistMaxPasses; pass++) {
    const float sigma_sqr = cur * cur;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kTsPerThread; k++)
      if (e2[k] >= 0.0f) s += (double)tdist_term(e2[k], sigma_sqr);
    const double ws = wave_sum64(s);
    if (pass == 0) {
      const double wc = wave_sum64((double)cnt);
      if (lane == 0) wcnt[wv] = wc;
    }
    if (lane == 0) wsum[wv] = ws;
    __syncthreads();
    if (t < 64) {
      double P = 0.0, C = 0.0;
#pragma unroll
      for (int w = 0; w < kTsThreads / 64; w++) { P += wsum[w]; if (pass == 0) C += wcnt[w]; }
      const unsigned tag = (epoch << 10) | (unsigned)(pass + 1);
      unsigned long long* mine = xbuf + ((size_t)(pass & 1) * kTsMaxWg + g) * 4;
      if (lane == 0 && !(fault && g == 1)) {   // (fault: test hook — workgroup 1 never publishes)
        const unsigned long long bits = (unsigned long long)__double_as_longlong(P);
        __hip_atomic_store(mine + 0, ((bits >> 32) << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, ((bits & 0xffffffffull) << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2, ((unsigned long long)(unsigned)(int)C << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      double a = 0.0, c = 0.0;
      bool all = true;
      const FineDeadline dl = FineDeadline::begin(wait_limit);
      for (int gg = lane; gg < G; gg += 64) {
        const unsigned long long* theirs = xbuf + ((size_t)(pass & 1) * kTsMaxWg + gg) * 4;
        unsigned long long w0 = 0, w1 = 0, w2 = 0;
        bool got = false;
        for (int spin = 0; !got; spin++) {
          w0 = __hip_atomic_load(theirs + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1 = __hip_atomic_load(theirs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w2 = __hip_atomic_load(theirs + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          got = ((unsigned)w0 == tag) && ((unsigned)w1 == tag) && ((unsigned)w2 == tag);
          if (!got && dl.expired(spin)) break;
        }
        all = all && got;
        a += __longlong_as_double((long long)(((w0 >> 32) << 32) | (w1 >> 32)));
        c += (double)(int)(unsigned)(w2 >> 32);
      }
      const bool everyone = __all(all);
      const double total = wave_sum64(a);
      if (pass == 0) n_valid = wave_sum64(c);
      const float nxt = (n_valid > 0.0) ? tdist_next_sigma(total, (int)n_valid) : cur;
      const bool done = !(n_valid > 0.0) || tdist_converged(nxt, cur);
      if (lane == 0) { sh_sigma = nxt; sh_done = done ? 1 : 0; sh_bail = everyone ? 0 : 1; }
    }
    __syncthreads();
    if (sh_bail) {   // a workgroup of this launch never published: the single-workgroup kernel behind this launch does the level
      if (t == 0) __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    cur = sh_sigma;
    if (sh_done) break;   // workgroup- and grid-uniform
  }
  if (g == 0 && t == 0) *scale_sqr_out = cur * cur;
}

constexpr int kFineThreads = 2 * kLmBlock;       // a workgroup works on TWO virtual blocks at a time, one per half
constexpr int kFineKMax = 32;                     // workgroups of one launch: they wait for each other, so each needs a CU of the XCD (32) to itself
// t-distribution scale passes (robust == 2): one fp64 sum per 64-point chunk (wave) and pass, double-buffered by pass parity,
// as a 16-byte granule pair {upper 32 bits | tag, lower 32 bits | tag}; the chunk's residual count as {count | tag} (pass 0 only).
constexpr int kScaleChunks = 4 * 2 * kFineKMax;  // 256: four waves per virtual block, <= 2 * kFineKMax virtual blocks per level
constexpr int kScaleWords = 2 * kScaleChunks * 2 + kScaleChunks;
constexpr int kFineScaleOff = 2 * kFineRowsMax * kFineGran + kFineKMax;   // (a multiple of 2 words: the pairs are 16-byte aligned)
static_assert(kFineScaleOff % 2 == 0, "scale granule pairs must be 16-byte aligned");
constexpr int kFineXbufWords = kFineScaleOff + kScaleWords;  // two row buffers + one placement word per workgroup + the scale area
constexpr int kTdistMaxEvals = 1023;             // evaluations of one launch the scale tags can tell apart (10 bits)
typedef unsigned FineG2 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) FineG2 FineG2Global;   // global_load (not flat_load) for the coherent gather loads
// One partial sum -> two granules {upper 32 bits | tag}, {lower 32 bits | tag}, each ONE 8-byte store. local: every workgroup of the
// launch sits on the same XCD (checked at run time) — plain stores then leave the line in that XCD's L2, where the L1-bypassing
// gather loads find it (MI355X guide: an agent-scope store drops the line and the reader pays the trip to the fabric);
// otherwise agent-scope write-through stores, correct at any placement.
__device__ __forceinline__ void fine_publish(unsigned long long* __restrict__ buf, int vb, int q, double acc, unsigned tag, bool local) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
  unsigned long long* g = buf + (size_t)vb * kFineGran + 2 * q;
  const unsigned long long g0 = ((bits >> 32) << 32) | tag, g1 = (bits << 32) | tag;
  if (local) {
    g[0] = g0;
    g[1] = g1;
  } else {
    __hip_atomic_store(g, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ int fine_xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
// WHICH XCD a persistent launch sits on. The blocks of a launch are dealt round-robin over the 8 XCDs, so its eight block classes
// (blockIdx.x & 7) sit on eight XCDs — but on which one a given class lands changes from launch to launch (the dispatcher carries
// on where the previous dispatch stopped). Two persistent launches of one process that end up on the SAME XCD — the pose LM's and
// the depth LM's, which run side by side — cannot both get their workgroups resident (a pose-LM workgroup takes a CU's registers
// whole): each holds some CUs and waits for the rest until the wait bound ends one of them (measured: a depth job in ~40 redone,
// 4 ms each time). So every persistent user has a HOME XCD — a hardware XCC id the host read with xcc_probe_kernel and hands out
// per device: an odo_lm the first, its estimator the second, the next tracker the third and fourth ... — and the blocks that find
// themselves there take part: no block waits for another one to decide this (a class that sits behind the OTHER persistent launch
// may not even start before that one ends). home < 0 (the probe did not see eight distinct ids): class 0, wherever it lands.
__device__ __forceinline__ bool fine_on_home(int home) {
  return home >= 0 ? fine_xcc_id() == home : (blockIdx.x & 7u) == 0u;
}
ODO_KERNEL void xcc_probe_kernel(int* __restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = fine_xcc_id();
}
struct XccIds { int id[8]; };   // the eight XCC ids of the device (id[0] < 0: unknown)
// ComputeScaleNaive across the workgroups of the persistent launch (ref: src/lm_optimizer.cpp:338-358): called by every thread of a
// workgroup that holds a virtual block of the level (workgroup barriers inside; a half without a virtual block contributes zeros).
// Per pass every wave sums its 64 terms (wave_sum64), the four waves of a half add their chunk sums up through LDS — G = ((c0 + c1) +
// c2) + c3 — and ONE tagged granule pair per virtual block goes out; every wave gathers all of them with L1-bypassing 16-byte loads
// (lane v: virtual block v) and folds them with wave_sum64: every wave of every workgroup computes the same sigma and the same
// convergence decision, nothing is broadcast. <= 64 publishers and one load per gathering lane where one sum per WAVE (<= 256
// publishers, four loads per lane) cost 2.8 k cycles per pass: 2.3 k (configs[0]: 0.885 -> 0.81 ms). The same association — chunk,
// virtual block, lanes — in coarse_tdist_sigma and lm_tdist_scale_kernel: one sigma bit for bit on every pipeline.
// A workgroup can be at most one pass ahead of the slowest (it cannot finish gathering pass p before every one has published p),
// hence the two parities. tag_ev = (launch epoch & 0xfff) << 20 | evaluation << 10; the pass number + 1 fills the low 10 bits (never
// 0: a cleared buffer matches nothing). Returns sigma; sets *bail when a wait ran out (the Solve then reports -2 like the row
// exchange does). td_ws: [2][8] doubles, td_cnt: [8] ints of LDS.
__device__ __forceinline__ float fine_tdist_sigma(unsigned long long* __restrict__ sbuf, int chunk, int nchunk, unsigned tag_ev,
                                                          float e2, bool valid, bool local, unsigned wait_limit, int* bail,
                                                          double (*td_ws)[8], int* td_cnt) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int vb = chunk >> 2, nvb = nchunk >> 2;
  unsigned long long* cnts = sbuf + 2 * kScaleChunks * 2;
  const int my_cnt = __popcll(__ballot(valid));
  float sigma = 5.0f;
  int n_total = 0;
  for (int pass = 0; pass < kTdistMaxPasses; pass++) {
    const unsigned tag = tag_ev | (unsigned)(pass + 1);
    const double ws = wave_sum64(valid ? (double)tdist_term(e2, sigma * sigma) : 0.0);
    if (lane == 0) { td_ws[pass & 1][wv] = ws; if (pass == 0) td_cnt[wv] = my_cnt; }
    __syncthreads();
    unsigned long long* sums = sbuf + (size_t)(pass & 1) * kScaleChunks * 2;
    if ((wv & 3) == 0 && lane == 0) {
      const double G = ((td_ws[pass & 1][wv] + td_ws[pass & 1][wv + 1]) + td_ws[pass & 1][wv + 2]) + td_ws[pass & 1][wv + 3];
      const int C = td_cnt[wv] + td_cnt[wv + 1] + td_cnt[wv + 2] + td_cnt[wv + 3];
      const unsigned long long bits = (unsigned long long)__double_as_longlong(G);
      const unsigned long long g0 = ((bits >> 32) << 32) | tag, g1 = (bits << 32) | tag;
      const unsigned long long gc = ((unsigned long long)(unsigned)C << 32) | tag;
      if (local) {
        sums[2 * vb] = g0; sums[2 * vb + 1] = g1;
        if (pass == 0) cnts[vb] = gc;
      } else {
        __hip_atomic_store(sums + 2 * vb, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sums + 2 * vb + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pass == 0) __hip_atomic_store(cnts + vb, gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    FineG2 g = {0, 0, 0, 0};
    unsigned long long gc = 0;
    bool all = false;
    FineDeadline dl = {0ull, 0ull, wait_limit};
    __builtin_amdgcn_s_setprio(0);
    for (int spin = 0; !all; spin++) {
      if (spin == 1) dl = FineDeadline::begin(wait_limit);
      if (spin > 0) { if (dl.expired(spin)) break; __builtin_amdgcn_s_sleep(1); }
      bool mine = true;
      if (lane < nvb) {
        g = *(const volatile FineG2Global*)(sums + 2 * lane);
        if (pass == 0) gc = __hip_atomic_load(cnts + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mine = (g.x == tag) && (g.z == tag) && (pass != 0 || (unsigned)gc == tag);
      }
      all = __all(mine);
    }
    __builtin_amdgcn_s_setprio(3);
    if (!all) { if (lane == 0) *bail = 1; break; }
    double part = 0.0, cnt = 0.0;
    if (lane < nvb) {
      part = __longlong_as_double((long long)(((unsigned long long)g.y << 32) | (unsigned long long)g.w));
      if (pass == 0) cnt = (double)(unsigned)(gc >> 32);
    }
    const double total = wave_sum64(part);
    if (pass == 0) n_total = (int)wave_sum64(cnt);
    const float nxt = (n_total > 0) ? tdist_next_sigma(total, n_total) : sigma;
    const bool done = (n_total == 0) || tdist_converged(nxt, sigma);
    sigma = nxt;
    if (done) break;
  }
  return sigma;
}
// kTdist: the t-distribution build (robust == 2: the scale passes of fine_tdist_sigma in front of the weights). A template
// parameter, not a branch: the Huber / L2 kernel's register count is part of how it shares its CUs with the depth stream
// (tests/test_abi.py::test_kernel_register_budgets), and the scale loop's gather registers would cost it 27 VGPRs.
// kTrace = false: no per-evaluation trace rows / cost statistics (see lm_coarse_body's kFull) — the trackers' optimisers.
template <bool kTdist, bool kTrace>
__device__ __forceinline__ void lm_fine_body(const StepArgs& a, const StepLaunch& q, int K, int w, unsigned long long* __restrict__ xbuf,
                                             int fault, int lo_level, const unsigned* __restrict__ lv_src) {
#if ODO_PHASE_STAMPS
  const unsigned long long w_entry = (unsigned long long)wall_clock64();
#endif
  __builtin_amdgcn_s_setprio(3);  // see lm_coarse_kernel
  const int t = threadIdx.x, tl = t & (kLmBlock - 1), half = t >> 8;
  const bool publisher = (w == 0);
  if (publisher && t == 0 && q.span) atomicMin(q.span, (unsigned long long)wall_clock64());
  __shared__ LmState s_sh;
  __shared__ double fold_sh[8 * 32];
  __shared__ double acc_sh[32];
  __shared__ StepLevel lv_sh[ODO_MAX_LEVELS_K];  // dynamic level index: keep the by-value argument out of scratch memory
  __shared__ float rows_sh2[2][kRowFloats * RowBuf<kLmBlock>::W];
  __shared__ int bail_sh, local_sh;
  __shared__ unsigned long long sm_sh[4];   // ODO_COARSE_STAMPS: phase sums of the publisher's state machine, flushed once at exit
  float* rows_sh = rows_sh2[half];
  const unsigned tag_base = a.fine_epoch << 8;   // unique per launch on this buffer (lm_fine_next_epoch): a stale granule cannot pass for a new one
  unsigned long long* place = xbuf + 2 * kFineRowsMax * kFineGran;  // [K] {xcc id, launch epoch} words
  const unsigned wait_limit = a.fine_wait ? a.fine_wait : kFineWaitTicks;
  lm_copy_levels(lv_sh, lv_src);
  if (t == 0) {
    bail_sh = 0;
    local_sh = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) sm_sh[i] = 0;
    // where am I? (agent-scope store: this exchange must work at any placement)
    __hip_atomic_store(place + w, ((unsigned long long)(unsigned)fine_xcc_id() << 32) | tag_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // state in (left by the coarse launch, or initialised), a pending evaluation of an earlier launch consumed, pyramid walk started
  lm_fused_prologue(q.st_in, q.part_in, lv_sh, a.n_levels, a.lambda0, a.precision, s_sh, fold_sh, acc_sh, (kTrace ? a.trace : (LmTraceRow*)nullptr), a.cost_stat,
                    publisher, q.first_of_solve ? a.init : nullptr, a.stop_level, a.chain_in_token ? a.chain_pose : nullptr);
  LmHot hot;                    // wave 0's copy of the state between evaluations (lm_state_machine_hot)
  lm_hot_load(hot, s_sh);
  // does every workgroup of this launch share my XCD? (wave 0, lane i asks about workgroup i; the answer is the same everywhere)
  if (t < 64) {
    bool same = true, got = false;
    if (t < K) {
      unsigned long long pw = 0;
      const FineDeadline dl = FineDeadline::begin(wait_limit);
      for (int spin = 0; !got; spin++) {
        pw = __hip_atomic_load(place + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = ((unsigned)pw == tag_base);
        if (!got && dl.expired(spin)) break;
      }
      same = got && ((int)(pw >> 32) == fine_xcc_id());
    } else {
      got = true;
    }
    const bool all_got = __all(got), all_same = __all(same);
    if (t == 0) { local_sh = (all_got && all_same) ? 1 : 0; if (!all_got) bail_sh = 1; }
  }
  __syncthreads();
  const bool local = local_sh != 0;
  constexpr int kS = 8;  // sub-lanes per quantity: 29 x 8 = 232 accumulating threads per half (the step kernel's)
  const int my_q = tl / kS, my_s = tl % kS;
  int rowA = 0, rowB = 0;
  if (my_q < ODO_NACC) rows_of_quantity(my_q, &rowA, &rowB);
  const int fq = t & 31, fseg = t >> 5;  // the fold's thread map (segments 0..7 fold: the first half of the workgroup)
  PointK pt;
  bool pt_ok = false;
  int pt_level = -1;
  unsigned long long c_eval = 0, c_xchg = 0, c_sm = 0, c_it = 0, c_last = ODO_DBG(a) ? __builtin_readcyclecounter() : 0;
#if ODO_PHASE_STAMPS
  unsigned long long c_first = 0, n_first = 0;   // the first evaluation of every level (keyframe points and taps not in this XCD's L2 yet)
#endif
  const unsigned long long c_begin = c_last, w_begin = ODO_DBG(a) ? (unsigned long long)wall_clock64() : 0;
  auto lap = [&](unsigned long long& sum) {
    if (ODO_DBG(a)) { const unsigned long long now = __builtin_readcyclecounter(); sum += now - c_last; c_last = now; }
  };
  for (int ev = 0; ev < 4096; ev++) {
    const bool run = (s_sh.active != 0 && s_sh.status == 0 && !s_sh.finished && !bail_sh && s_sh.level >= lo_level);  // block-uniform
    if (!run) break;   // (a level below lo_level has begun: the step launches behind this one carry on from the state it leaves)
    c_it++;
    const StepLevel& L = lv_sh[s_sh.level];
    const float* I2u = lm_uniform_ptr(L.I2);   // (point_residual_g)
    const int nblk = L.nblk;
    const unsigned tag = tag_base + 1u + (unsigned)(ev % 255);   // never tag_base itself: that is the placement word's
    unsigned long long* buf = xbuf + (size_t)(ev & 1) * kFineRowsMax * kFineGran;
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; i++) T[i] = s_sh.T[i];
    // ---- (a) + (b): my virtual blocks, two at a time (half h of the workgroup works on virtual block vb0 + h * K) ----
#if ODO_PHASE_STAMPS
    const bool first_of_level = (s_sh.level != pt_level);
#endif
    const bool resident = (nblk <= 2 * K) && (L.n <= nblk * kLmBlock);  // one point per thread covers my share of the level
    if (resident) {
      // The keyframe point stays in registers for the whole level: one trip to L2 per level instead of one per evaluation.
      const int vb = w + half * K;
      if (s_sh.level != pt_level) {
        pt_level = s_sh.level;
        const int idx = vb * kLmBlock + tl;
        pt_ok = vb < nblk && idx < L.n;
        if (pt_ok) pt = load_point(L.pl, idx);
      }
      float r = 0.0f, wgt = 0.0f, J[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      bool valid = false;
      if (pt_ok && point_residual_g<kTdist || kTrace>(pt, T, L.k, I2u, L.rows, L.cols, &r, J)) {
        wgt = robust_weight(r, (kTdist || kTrace) ? a.robust : (a.robust == 1 ? 1 : 0), a.huber_delta, 1.0f);
        valid = true;
      }
      if (kTdist && w < nblk) {   // (workgroup-uniform: the scale passes hold workgroup barriers; a half without a virtual block adds zeros)
        // t-distribution weights need the scale of ALL residuals of this evaluation first (ref: src/lm_optimizer.cpp:257-261)
        __shared__ double td_ws_sh[2][8];
        __shared__ int td_cnt_sh[8];
        const float sg = fine_tdist_sigma(xbuf + kFineScaleOff, vb * 4 + (tl >> 6), 4 * nblk,
                                          ((a.fine_epoch & 0xfffu) << 20) | (((unsigned)ev & 0x3ffu) << 10), r * r, valid, local, wait_limit, &bail_sh,
                                          td_ws_sh, td_cnt_sh);
        if (valid) wgt = robust_weight(r, 2, a.huber_delta, sg * sg);
      }
      rows_store_z(rows_sh, RowBuf<kLmBlock>::W, tl, J, wgt, r, valid);   // (the state machine's closing barrier separates this
      __syncthreads();                                                  //  from the previous evaluation's row sums)
      if (bail_sh) break;   // (block-uniform behind the barrier) a scale pass ran out of time
      if (vb < nblk) {
        double accq = 0.0;
        if (my_q < ODO_NACC) accq = rows_accumulate<kLmBlock, kS>(rows_sh, rowA, rowB, my_s, accq);
        accq = rows_butterfly8(accq);
        if (my_q < ODO_NACC && my_s == 0 && !(fault && vb == 0)) fine_publish(buf, vb, my_q, accq, tag, local);   // fault: row 0 never appears
      }
    } else {
      // Levels of more virtual blocks than the 2 K the workgroups keep in registers (lm_plan_levels: up to 2 K x fine_passes, two
      // passes by default — a keyframe near the reference's point cap, bench.py's `saturated_keyframe`): the points are re-read from
      // L2 every evaluation, pass by pass. (The register allocator also needs this branch: without it the kernel takes 245 VGPRs
      // instead of 203, see test_kernel_register_budgets.)
      const int rounds = (L.n + nblk * kLmBlock - 1) / (nblk * kLmBlock);  // > 1 only beyond 160 x 256 points (a round without
      for (int vb0 = w; vb0 < nblk; vb0 += 2 * K) {                        //  points adds zero rows: the sums do not change)
        const int vb = vb0 + half * K;
        double accq = 0.0;
        for (int rd = 0; rd < rounds; rd++) {
          const int idx = (vb + rd * nblk) * kLmBlock + tl;
          float r = 0.0f, wgt = 0.0f, J[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
          bool valid = false;
          if (vb < nblk && idx < L.n) {
            const PointK p = load_point(L.pl, idx);
            if (point_residual_g<kTdist || kTrace>(p, T, L.k, I2u, L.rows, L.cols, &r, J)) {
              wgt = robust_weight(r, (kTdist || kTrace) ? a.robust : (a.robust == 1 ? 1 : 0), a.huber_delta, 1.0f);
              valid = true;
            }
          }
          __syncthreads();  // the previous round's rows have been consumed
          rows_store_z(rows_sh, RowBuf<kLmBlock>::W, tl, J, wgt, r, valid);
          __syncthreads();
          if (my_q < ODO_NACC) accq = rows_accumulate<kLmBlock, kS>(rows_sh, rowA, rowB, my_s, accq);
        }
        accq = rows_butterfly8(accq);
        if (vb < nblk && my_q < ODO_NACC && my_s == 0 && !(fault && vb == 0)) fine_publish(buf, vb, my_q, accq, tag, local);   // fault: as above
      }
    }
#if ODO_PHASE_STAMPS
    { const unsigned long long before = c_eval; lap(c_eval); if (first_of_level) { c_first += c_eval - before; n_first++; } }
#else
    lap(c_eval);
#endif
    // ---- (c): gather every row in the fold's order (lm_fused_prologue: segment seg adds rows seg, seg + 8, ... ascending) ----
    double v = 0.0;
    __builtin_amdgcn_s_setprio(0);  // waiting is not urgent: the depth stream's waves on this CU go first while we poll
    if (fq < ODO_NACC && fseg < 8) {
      for (int b0 = fseg; b0 < nblk; b0 += 8 * kFineChunk) {
        FineG2 g2[kFineChunk];   // {hi granule, lo granule} of one double: one 16-byte load that bypasses L1
        bool all = false;
        FineDeadline dl = {0ull, 0ull, wait_limit};
        for (int spin = 0; !all; spin++) {
          if (spin == 1) dl = FineDeadline::begin(wait_limit);   // the first pass succeeds three times in four: no clock read then
          if (spin > 0) { if (dl.expired(spin)) break; __builtin_amdgcn_s_sleep(1); }
          all = true;
#pragma unroll
          for (int u = 0; u < kFineChunk; u++) {
            const int b = b0 + 8 * u;
            if (b < nblk) g2[u] = *(const volatile FineG2Global*)(buf + (size_t)b * kFineGran + 2 * fq);  // re-read every pass, sc0 sc1
          }
#pragma unroll
          for (int u = 0; u < kFineChunk; u++) {
            const int b = b0 + 8 * u;
            if (b < nblk) all = all && (g2[u].x == tag) && (g2[u].z == tag);
          }
        }
        if (!all) bail_sh = 1;  // a workgroup of this Solve never published: report failure, do not hang
#pragma unroll
        for (int u = 0; u < kFineChunk; u++) {
          const int b = b0 + 8 * u;
          if (b < nblk) v += __longlong_as_double((long long)(((unsigned long long)g2[u].y << 32) | (unsigned long long)g2[u].w));
        }
      }
    }
    __builtin_amdgcn_s_setprio(3);
    if (fseg < 8) fold_sh[fseg * 32 + fq] = v;
    __syncthreads();
    if (t < ODO_NACC) {
      double acc = 0.0;
#pragma unroll
      for (int g = 0; g < 8; g++) acc += fold_sh[g * 32 + t];
      acc_sh[t] = acc;
    }
    // (no second barrier: written and read by wave 0 only, see lm_coarse_body; bail_sh was set in front of the barrier above)
    lap(c_xchg);
    if (bail_sh) break;
    // ---- (d): the state machine, every workgroup for itself ----
    lm_state_machine_hot(hot, lv_sh, a.n_levels, a.lambda0, a.precision, s_sh, acc_sh, (kTrace ? a.trace : (LmTraceRow*)nullptr), a.cost_stat, publisher,
                         (ODO_DBG(a) && publisher) ? sm_sh : nullptr, a.stop_level);
    lap(c_sm);
  }
#if ODO_PHASE_STAMPS
  const unsigned long long w_loop_end = (unsigned long long)wall_clock64();
#endif
  if (t == 0) lm_hot_store(hot, s_sh);
  if (bail_sh && t == 0) { s_sh.status = -2; s_sh.active = 0; s_sh.finished = 1; }   // -2: gave up waiting (the host redoes the Solve)
  __syncthreads();
  if (publisher) {
    lm_fused_publish(s_sh, q.st_out, a.host_prog, q.seq, a.token, a.cost_stat, a.out, a.done_flag, a.final_state, chain_out_of(a));
#if ODO_PHASE_STAMPS
    const unsigned long long w_published = (unsigned long long)wall_clock64();
#endif
    if (ODO_DBG(a) && t == 0) { ODO_DBG(a)[128] += c_eval; ODO_DBG(a)[129] += c_xchg; ODO_DBG(a)[130] += c_sm; ODO_DBG(a)[131] += c_it; ODO_DBG(a)[132] += 1; ODO_DBG(a)[133] += local ? 1 : 0;
      ODO_DBG(a)[134] += __builtin_readcyclecounter() - c_begin; ODO_DBG(a)[135] += (unsigned long long)wall_clock64() - w_begin;
#pragma unroll
      for (int i = 0; i < 4; i++) ODO_DBG(a)[136 + i] += sm_sh[i]; }
    if (t == 0 && q.span) atomicMax(q.span + 1, (unsigned long long)wall_clock64());
#if ODO_PHASE_STAMPS
    if (ODO_DBG(a) && t == 0) {
      const unsigned long long now = w_published;   // (the stamped build's own counter updates over PCIe follow it)
      g_lm_diag[2] = w_loop_end;
      g_lm_diag[19] += c_first; g_lm_diag[20] += n_first;
      g_lm_diag[9] += w_begin - w_entry; g_lm_diag[10] += w_loop_end - w_begin; g_lm_diag[11] += now - w_loop_end;
      if (g_lm_diag[0] && w_entry > g_lm_diag[0] && w_entry - g_lm_diag[0] < 100000ull) { g_lm_diag[12] += w_entry - g_lm_diag[0]; g_lm_diag[15] += 1; }
      g_lm_diag[1] = now; g_lm_diag[13] += 1;
    }
#endif
  }
}
// grid = 8 * K blocks: the class (blockIdx.x & 7) that sits on the optimiser's home XCD takes part, the others return at once
// "The dispatch of a pose-LM persistent launch is in progress": block 0 of its grid writes the launch's number (process-wide, from the
// host: several optimisers in one process never write the same value) into word 0 on entry, the last block into word 1 — the dispatcher deals the blocks of a grid in order, so word 0 != word 1 means that some blocks of that
// launch have not been dispatched yet. Read by depth_lm_persistent_kernel, whose own dispatch can be what they are waiting for: every
// block of a launch visits the XCD the dispatcher deals it to, also the seven eighths that return at once, and a pose-LM block (416
// VGPRs per SIMD) finds no CU on an XCD filled with the depth launch's 80 workgroups while the depth launch's next block finds none
// on the XCD the pose LM's resident workgroups fill: two half-dispatched persistent launches, each waiting for workgroups that are
// never dispatched (seen with ODO_LOG_GIVEUPS: all 80 depth workgroups resident in the end, the last of the grid 0.5 ms behind the
// first; about once per process start — launches 4 and 8 of bench.py's tracker — and once in ~ 5 000 frames later).
ODO_DEVICE_VAR unsigned g_lm_fine_dispatch[2];   // (the chain unit's own copy is never used: its kernels get the main unit's address as an argument)
__device__ __forceinline__ bool lm_fine_mid_dispatch() {
  return __hip_atomic_load(&g_lm_fine_dispatch[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
         __hip_atomic_load(&g_lm_fine_dispatch[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool kTdist, bool kTrace>
__device__ __forceinline__ void lm_fine_entry(const StepArgs& a, int K, unsigned long long* __restrict__ xbuf, int fault, int lo_level,
                                              unsigned* dispatch_words) {
  // (dispatch_words = the MAIN unit's g_lm_fine_dispatch, handed over as an argument: the depth launches, compiled there, read it)
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) __hip_atomic_store(&dispatch_words[0], a.fine_dispatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == gridDim.x - 1) __hip_atomic_store(&dispatch_words[1], a.fine_dispatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!fine_on_home(a.fine_home) || lm_chain_skip(a)) return;
  const StepLaunch q = {a.st_in, a.st_out, a.part_in, a.part_out, a.seq, a.first_of_solve, a.span};
  lm_fine_body<kTdist, kTrace>(a, q, K, (int)(blockIdx.x >> 3), xbuf, fault, lo_level, lm_kernarg_words());
}
#ifdef ODO_LM_CHAIN_TU
static __global__ void __launch_bounds__(kFineThreads) lm_fine_kernel(StepArgs a, int K, unsigned long long* __restrict__ xbuf, int fault,
                                                               int lo_level, unsigned* dispatch_words) {
  lm_fine_entry<false, false>(a, K, xbuf, fault, lo_level, dispatch_words);
}
// The same for an optimiser that records its trace rows (odo_lm_set_record: the LevenbergMarquardtOptimizer objects of the tests).
static __global__ void __launch_bounds__(kFineThreads) lm_fine_trace_kernel(StepArgs a, int K, unsigned long long* __restrict__ xbuf, int fault,
                                                                     int lo_level, unsigned* dispatch_words) {
  lm_fine_entry<false, true>(a, K, xbuf, fault, lo_level, dispatch_words);
}
// The same with t-distribution weights (a.robust == 2).
static __global__ void __launch_bounds__(kFineThreads) lm_fine_tdist_kernel(StepArgs a, int K, unsigned long long* __restrict__ xbuf, int fault,
                                                                     int lo_level, unsigned* dispatch_words) {
  lm_fine_entry<true, true>(a, K, xbuf, fault, lo_level, dispatch_words);
}
#endif  // ODO_LM_CHAIN_TU
// Batched twin: the sequences of a batched Solve each get an XCD (sequence i: the blocks with blockIdx.x % 8 == i % 8; beyond
// eight sequences two or more share an XCD, K workgroups each). grid = 8 * K * ceil(n / 8). A sequence whose levels do not
// fit (fine_lo >= min_level) takes no part: its blocks return at once and its levels follow on the batched step launches.
template <bool kTrace>
__device__ __forceinline__ void lm_fine_batch_entry(const StepArgs* __restrict__ table, int n, int K, int seq, int first_of_solve,
                                                    unsigned long long* span, int fault, const XccIds& xcc, unsigned* dispatch_words) {
  // sequence i on the i-th XCD of the device, whichever block class sits there in this launch (XCC ids unknown: on class i % 8, as dealt)
  int r = (int)(blockIdx.x & 7u);
  if (xcc.id[0] >= 0) {
    const int mine = fine_xcc_id();
#pragma unroll
    for (int c = 0; c < 8; c++) if (xcc.id[c] == mine) r = c;
  }
  if (threadIdx.x == 0 && n > 0) {   // (the batched launch is a pose-LM persistent launch like any other: see g_lm_fine_dispatch)
    if (blockIdx.x == 0) __hip_atomic_store(&dispatch_words[0], table[0].fine_dispatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == gridDim.x - 1) __hip_atomic_store(&dispatch_words[1], table[0].fine_dispatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int wa = (int)(blockIdx.x >> 3);
  const int i = (wa / K) * 8 + r, w = wa % K;
  if (i >= n) return;
  const StepArgs& a = table[i];
  const StepLaunch q = {a.st2[seq & 1], a.st2[(seq + 1) & 1], a.part2[seq & 1], a.part2[(seq + 1) & 1], seq, first_of_solve,
                        (i == 0) ? span : nullptr};
  if (a.fine_lo >= a.min_level) {
    // nothing to evaluate here, but the launch number is the whole batch's: one workgroup carries the sequence's state from this
    // launch's input buffer to its output buffer (prologue + publish, no level at or above lo_level = none), as a step launch of a
    // finished sequence does
    if (w == 0) lm_fine_body<false, kTrace>(a, q, 1, 0, a.xbuf, 0, ODO_MAX_LEVELS_K + 1, (const unsigned*)&table[i]);
    return;
  }
  lm_fine_body<false, kTrace>(a, q, K, w, a.xbuf, fault, a.fine_lo, (const unsigned*)&table[i]);   // (a batched Solve never carries t-distribution weights: lm_batch_begin)
}
ODO_KERNEL void __launch_bounds__(kFineThreads) lm_fine_kernel_batch(const StepArgs* __restrict__ table, int n, int K, int seq,
                                                                     int first_of_solve, unsigned long long* span, int fault, XccIds xcc,
                                                                     unsigned* dispatch_words) {
  lm_fine_batch_entry<false>(table, n, K, seq, first_of_solve, span, fault, xcc, dispatch_words);   // lean: nothing recorded, floor sampling
}
ODO_KERNEL void __launch_bounds__(kFineThreads) lm_fine_trace_kernel_batch(const StepArgs* __restrict__ table, int n, int K, int seq,
                                                                           int first_of_solve, unsigned long long* span, int fault, XccIds xcc,
                                                                     unsigned* dispatch_words) {
  lm_fine_batch_entry<true>(table, n, K, seq, first_of_solve, span, fault, xcc, dispatch_words);
}

// End of a fused Solve when no step launch has reported it (no launch was issued at all, or the host is not polling):
// consume the last pending evaluation, then hand the result over like lm_fused_publish does.
struct FinalizeArgs {
  const LmState* st_in;
  const double* part_in;
  float precision;
  LmTraceRow* trace;
  float* cost_stat;
  LmState* st_out;
  float* out;          // 42 floats: pose, status, n_evals, evaluations per level, cost statistics (host-mapped)
  int* done_flag;      // host-mapped: set to `token` when `out` is complete
  int token;
  int first_of_solve;  // no evaluation was launched (all budgets 0): the state comes from `init`
  float init[16];
};
ODO_KERNEL void __launch_bounds__(kLmBlock) lm_fused_finalize_kernel(FinalizeArgs a) {
  __shared__ LmState s_sh;
  __shared__ double fold_sh[8 * 32];
  __shared__ double acc_sh[32];
  lm_fused_prologue(a.st_in, a.part_in, nullptr, 0, 0.0f, a.precision, s_sh, fold_sh, acc_sh, a.trace,
                    a.cost_stat, true, a.first_of_solve ? a.init : nullptr);
  if (threadIdx.x < (int)(sizeof(LmState) / sizeof(int))) ((int*)a.st_out)[threadIdx.x] = ((const int*)&s_sh)[threadIdx.x];
  if (threadIdx.x == 0) lm_write_result(s_sh, a.cost_stat, a.out, a.done_flag, a.token);
}

// Test entry for solve_damped_wave (one wavefront).
ODO_KERNEL void solve_damped_wave_test_kernel(const double* __restrict__ acc_in, float lambda, float* __restrict__ out) {
  __shared__ double acc_sh[32];
  __shared__ float delta_sh[6];
  if (threadIdx.x < ODO_NACC) acc_sh[threadIdx.x] = acc_in[threadIdx.x];
  __syncthreads();
  solve_damped_wave(acc_sh, lambda, delta_sh);
  __syncthreads();
  if (threadIdx.x < 6) out[threadIdx.x] = delta_sh[threadIdx.x];
}

ODO_KERNEL void lm_begin_solve_kernel(LmState* __restrict__ st, const float* __restrict__ init, float* __restrict__ cost_stat) {
  if (threadIdx.x == 0) {
    LmState s;
    float m[16];
    for (int i = 0; i < 16; i++) m[i] = init[i];
    lm_begin_solve(&s, m);
    s.level = -1; s.iter = 0; s.lambda = 0.0f; s.err_last = 1e+10f;
    for (int i = 0; i < 16; i++) s.T[i] = m[i];
    *st = s;
    for (int i = 0; i < 16; i++) cost_stat[i] = 0.0f;
  }
}

ODO_KERNEL void lm_begin_level_kernel(LmState* __restrict__ st, int level, float lambda0, int max_iters) {
  if (threadIdx.x == 0) {
    LmState s = *st;
    s.stop_reason = 0;
    lm_begin_level(&s, level, lambda0, max_iters);
    *st = s;
  }
}

// affine_ = current_estimate.matrix() (ref: src/lm_optimizer.cpp:158), or the pseudo-identity whose (3,3)
// is 0 on failure (ref: :48-52,60-65). out[16] = pose, out[16] = status as float.
__device__ __forceinline__ void lm_finalize_body(const LmState* __restrict__ st, float* __restrict__ out);
ODO_KERNEL void lm_finalize_kernel(const LmState* __restrict__ st, float* __restrict__ out) { lm_finalize_body(st, out); }
ODO_KERNEL void lm_finalize_batch_kernel(const UpdItem* __restrict__ items) { lm_finalize_body(items[blockIdx.x].st, items[blockIdx.x].out); }
ODO_KERNEL void lm_begin_solve_batch_kernel(const UpdItem* __restrict__ items) {
  if (threadIdx.x == 0) {
    const UpdItem& q = items[blockIdx.x];
    LmState s;
    float m[16];
    for (int i = 0; i < 16; i++) m[i] = q.init[i];
    lm_begin_solve(&s, m);
    s.level = -1; s.iter = 0; s.lambda = 0.0f; s.err_last = 1e+10f;
    for (int i = 0; i < 16; i++) s.T[i] = m[i];
    *q.st = s;
    for (int i = 0; i < 16; i++) q.cost_stat[i] = 0.0f;
  }
}
ODO_KERNEL void lm_begin_level_batch_kernel(const UpdItem* __restrict__ items) {
  if (threadIdx.x == 0) {
    const UpdItem& q = items[blockIdx.x];
    LmState s = *q.st;
    s.stop_reason = 0;
    lm_begin_level(&s, q.expect_level, q.lambda0, q.max_iters);
    *q.st = s;
  }
}
__device__ __forceinline__ void lm_finalize_body(const LmState* __restrict__ st, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    float m[16];
    if (st->status == 0) {
      se3_to_colmajor(st->cur, m);
    } else {
      for (int i = 0; i < 16; i++) m[i] = 0.0f;
      m[0] = 1.0f; m[5] = 1.0f; m[10] = 1.0f;
    }
    for (int i = 0; i < 16; i++) out[i] = m[i];
    out[16] = (float)st->status;
    out[17] = (float)st->n_evals;
    for (int i = 0; i < 8; i++) out[18 + i] = (float)st->iters_level[i];
  }
}

// =============================================================================================
// Depth estimator kernels (D2/D3/D5 of SURVEY section 8a)
// =============================================================================================
constexpr int kSelCap = 80;      // ref: src/depth_estimate.cpp:334
constexpr int kSelBlocks = 512;  // 16 x 32 blocks, ref: :302
// 512 threads (two keys per thread at KITTI's 874-pixel tiles): a 1024-thread block does not fit beside a workgroup of the
// persistent LM launch on its CUs (4 waves per SIMD x 32 VGPRs against the 96 it leaves), so an eighth of the grid waited for the
// Solve to end and the depth job ran 230 us instead of 205; alone the kernel takes 9.7 us either way (256 threads: 10.9).
#ifndef ODO_SEL_THREADS
#define ODO_SEL_THREADS 512
#endif
constexpr int kSelThreads = ODO_SEL_THREADS;
constexpr int kSelMaxElems = 4096;

// Point selection (ref: src/depth_estimate.cpp:300-342): one workgroup per 16x32 grid block.
// |grad| on the blurred left image, block median via an LDS bitonic sort, threshold = median + grad_th,
// first <= 80 pixels in raster order above the threshold. Outputs the mask and a fixed-slot point list
// pts[block*80 + k] = x | y<<16, cnt[block].
__device__ __forceinline__ void depth_select_kernel_body(const float* __restrict__ L, int rows, int cols, int bnd,
                                                                    float grad_th, uint8_t* __restrict__ val,
                                                                    uint32_t* __restrict__ pts, int* __restrict__ cnt) {
  __shared__ float mag[kSelMaxElems];
  __shared__ int wave_tot[kSelThreads / kWave];
  __shared__ int base_sh;
  const int t = threadIdx.x;
  const int bw = (cols - bnd * 2) / 32, bh = (rows - bnd * 2) / 16;
  const int bsz = bw * bh;
  const int b = blockIdx.x;
  if (bsz <= 0 || bsz > kSelMaxElems) {
    if (t == 0) cnt[b] = 0;
    return;
  }
  const int sy = bnd + (b / 32) * bh, sx = bnd + (b % 32) * bw;
  constexpr int kPer = kSelMaxElems / kSelThreads;  // keys per thread (1 at KITTI size: 38 x 23 = 874 per block)
  unsigned keys[kPer];
#pragma unroll
  for (int u = 0; u < kPer; u++) {
    const int e = t + u * kSelThreads;
    keys[u] = 0xffffffffu;  // padding sorts last
    if (e < bsz) {
      const int y = sy + e / bw, x = sx + e % bw;
      const float gx = 0.5f * (L[(size_t)y * cols + x + 1] - L[(size_t)y * cols + x - 1]);
      const float gy = 0.5f * (L[(size_t)(y + 1) * cols + x] - L[(size_t)(y - 1) * cols + x]);
      const float m = sqrtf(gx * gx + gy * gy);  // :321
      mag[e] = m;
      keys[u] = __float_as_uint(m);  // m >= +0: the bit pattern orders like the value
    }
  }
  __syncthreads();
  // Block median = the element nth_element(size / 2) leaves at that position (:328) = the k-th smallest, k = bsz / 2:
  // MSB-first radix select, 8 bits per pass. The keys that still match the prefix are counted into a 256-bin LDS histogram
  // (one ds_add per key), wave 0 finds the bin that holds the k-th key with a prefix scan (four bins per lane), and every
  // thread narrows (prefix, k) identically. Four passes of ~20 instructions per wave — the 2-bit / four-ballot variant this
  // replaces spent ~1 000 instructions per wave (16 passes), and with 32 waves resident per CU the selection is
  // instruction-issue bound (13 us -> see DESIGN 5.2).
  __shared__ int hist[256];
  __shared__ int sel_sh[2];
  unsigned prefix = 0u, pmask = 0u;
  int kth = bsz / 2;
  {
    const int lane = t & 63;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int shift = 24 - 8 * pass;
      if (t < 256) hist[t] = 0;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kPer; u++) {
        if (u * kSelThreads >= bsz) break;  // block-uniform: no keys in this round
        const int e = t + u * kSelThreads;
        if (e < bsz && (keys[u] & pmask) == prefix) atomicAdd(&hist[(keys[u] >> shift) & 255u], 1);
      }
      __syncthreads();
      if (t < 64) {
        const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const int sum = (c0 + c1) + (c2 + c3);
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o, 64);
          if (lane >= o) incl += v;
        }
        const int excl = incl - sum;
        if (excl <= kth && kth < incl) {   // exactly one lane: the matching keys number more than kth
          int below = excl, d = 4 * lane;
          if (kth >= below + c0) {
            below += c0; d++;
            if (kth >= below + c1) {
              below += c1; d++;
              if (kth >= below + c2) { below += c2; d++; }
            }
          }
          sel_sh[0] = d;
          sel_sh[1] = below;
        }
      }
      __syncthreads();
      kth -= sel_sh[1];
      prefix |= (unsigned)sel_sh[0] << shift;
      pmask |= 255u << shift;
    }
  }
  const float median = __uint_as_float(prefix);
  const float th = median + grad_th;  // :328-329
  if (t == 0) base_sh = 0;
  __syncthreads();
  const int lane = t & 63, wv = t >> 6;
  for (int c0 = 0; c0 < bsz; c0 += kSelThreads) {
    const int e = c0 + t;
    const bool f = (e < bsz) && (mag[e] > th);  // :335
    const unsigned long long bal = __ballot(f);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(bal);
    __syncthreads();
    int off = base_sh;
    for (int w = 0; w < wv; w++) off += wave_tot[w];
    const int rank = off + pre;
    if (f && rank < kSelCap) {
      const int y = sy + e / bw, x = sx + e % bw;
      if (val) val[(size_t)y * cols + x] = 1;  // :336 (NULL: a selection prepared ahead of ComputeDepth, odo_depth_prepare_left_dev)
      pts[b * kSelCap + rank] = (uint32_t)x | ((uint32_t)y << 16);
    }
    __syncthreads();
    if (t == 0) {
      int tot = base_sh;
      for (int w = 0; w < kSelThreads / kWave; w++) tot += wave_tot[w];
      base_sh = tot;
    }
    __syncthreads();
  }
  if (t == 0) cnt[b] = (base_sh < kSelCap) ? base_sh : kSelCap;
}
ODO_KERNEL void __launch_bounds__(kSelThreads) depth_select_kernel(const float* __restrict__ L, int rows, int cols, int bnd,
                                                                    float grad_th, uint8_t* __restrict__ val,
                                                                    uint32_t* __restrict__ pts, int* __restrict__ cnt) {
  depth_select_kernel_body(L, rows, cols, bnd, grad_th, val, pts, cnt);
}

// Epipolar line search (ref: src/depth_estimate.cpp:345-398): one wavefront per selected point. Each lane keeps its first strict
// minimum over ascending candidate columns, then a wave-wide (ssd, right_x) argmin where ties take the lowest right_x — the
// sequential strict-< scan's answer (:385-386). SSD uses the AVX hadd tree (ssd8_tree).
//
// Main loop: a lane owns FOUR consecutive candidates c0 .. c0 + 3 (c0 = base + 4 * lane) and fetches the 26 right-image values
// their 32 taps touch with seven vector loads (row y: 8 values, y - 1: 6, y + 1 / y + 2 / y - 2: 4 each) — 6.5 dwords per
// candidate instead of 8, 7 requests per 4 candidates instead of 32, and a wave's 1 KB request straddles one extra cache line
// where every 256-B request of the one-candidate-per-lane form straddles one too. The scan is bound by the vector L1's line
// rate (the blurred pair is L2 resident; 18 k points each streaming five rows leave nothing to reuse in a 32 KB L1), so lines per
// candidate are what it costs: 23.4 -> 20.2 us for the reference's full range. Fewer than 256 candidates left: lanes =
// consecutive candidates, stride 64. The slot header (count, packed coordinates, eight left taps) comes through the scalar
// cache: the wave's index is uniform, so none of it occupies the vector memory path.
//
// Measured and not kept (round 3, all bit-identical; tools/scan_probe.py): a persistent grid (256 .. 2048
// blocks walking the slots: 53 .. 22 us — static striding balances worse than the dispatcher); groups of 2 .. 16 slots per wave
// with their headers fetched together (23.3 .. 50.9 us — the points of a group then scan one after the other); the points of a
// group walked in lock step, 32 taps in flight per wave (21.9 .. 24.6 us; best for +-128 px: 10.1 against 11.9 us); right-image
// rows of a (tile, 256-column chunk) staged in LDS by 512-thread workgroups with a 64-bit atomic-min merge and a resolve kernel
// (35.1 + 3.9 us: per (point, chunk) bookkeeping outweighs the cheaper taps). With ONE candidate per point the kernel still
// takes 8 .. 11 us in every shape: the floor is the per-point chain of dependent trips to L2, not the scan.
// Round 5 (profiles/r05_scan_rows_ab.md, commit 95f24c6): a workgroup per IMAGE ROW — the five right rows y - 2 .. y + 2 staged in
// LDS once (25 KB), the row's points listed in LDS with their left taps, a wave per point reading taps with ds_read_b128 — removes
// that chain altogether (staging + list: 6 us) and is still 2 x slower (32-35 us with 2-8 workgroups per row, static assignment): rows
// hold 0 .. 500 points and up to 4.7 x the mean number of candidates, and what bounds the scan is VALU issue — 26 instructions per
// candidate + ~150 per point = 7 M wave-instructions, ~13 us on 1 024 SIMDs at the measured 4 cycles each —, which the wave-per-slot
// grid spreads evenly and a row-wise grid does not. (An LDS work counter fetched by `if (lane == 0) atomicAdd` in front of
// readfirstlane made hipcc 7.2 build a loop that never ended; fetched by all lanes it cost 75 us.) -fno-slp-vectorize (the packed
// v_pk_add / v_pk_mul_f32 the compiler picks here issue at 8 cycles against 3 + 3): 2 % here — and 4 % of the pose-LM chain, which is
// why the whole library is built with it (build.py).
struct __attribute__((packed, aligned(4))) ScanF4 { float v[4]; };
struct __attribute__((packed, aligned(4))) ScanF2 { float v[2]; };
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// (best, match) <- the lexicographic minimum with the lane the DPP pattern pairs this one with (every pattern used is a
// permutation inside a row of 16 lanes, so every lane has a source)
template <int CTRL> __device__ __forceinline__ void scan_min_step(float& best, int& match) {
  const float ob = dpp_f<CTRL>(best);
  const int om = dpp_i<CTRL>(match);
  const bool take = ob < best || (ob == best && om < match);
  best = take ? ob : best;
  match = take ? om : match;
}
__device__ __forceinline__ void depth_disparity_kernel_body(const float* __restrict__ L, const float* __restrict__ R,
                                                               int rows, int cols, int bnd, int max_disp, float ssd_th,
                                                               float f0, float baseline, const uint32_t* __restrict__ pts,
                                                               const int* __restrict__ cnt, float* __restrict__ disp,
                                                               float* __restrict__ dep, float* __restrict__ d0,
                                                               uint8_t* __restrict__ matched) {
  const int lane = threadIdx.x & 63;
  int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));  // wave-uniform
  if (max_disp <= 0) {
    // Reference range [bnd, x): a point's work grows with its column, and the dispatcher hands out blocks in order — the 16 x 32
    // selection blocks go out column by column from the RIGHT, heaviest first, so the launch does not end on its longest points
    // (17.8 -> 16.7 us on bench.py's frame; with a +-128 px window every point costs the same and the plain order is 4 % faster).
    const int bb = slot / kSelCap;
    slot = ((bb % 16) * 32 + (31 - bb / 16)) * kSelCap + slot % kSelCap;
  }
  const int b = slot / kSelCap, k = slot % kSelCap;
  if (b >= kSelBlocks) return;
  const uint32_t pk = pts[slot];  // requested with the count, not after it (an unused slot's word is never looked at)
  if (k >= cnt[b]) return;
  const int x = (int)(pk & 0xffffu), y = (int)(pk >> 16);
  const float* lpp = L + (size_t)(y - 2) * cols;
  const float* lp = L + (size_t)(y - 1) * cols;
  const float* lc = L + (size_t)y * cols;
  const float* ln = L + (size_t)(y + 1) * cols;
  const float* lnn = L + (size_t)(y + 2) * cols;
  const float Lp[8] = {lnn[x], ln[x - 1], lc[x + 2], lc[x], lc[x - 2], lp[x + 1], lp[x - 1], lpp[x]};  // :380-381
  const float* rpp = R + (size_t)(y - 2) * cols;
  const float* rp = R + (size_t)(y - 1) * cols;
  const float* rc = R + (size_t)y * cols;
  const float* rn = R + (size_t)(y + 1) * cols;
  const float* rnn = R + (size_t)(y + 2) * cols;
  int lo = bnd;
  if (max_disp > 0 && x - max_disp > lo) lo = x - max_disp;
  float best = 1e+10f;  // :367
  int match = 0x7fffffff;
  int base = lo;
  // :382 — four consecutive candidates per lane while a whole 256-candidate trip fits below x
  for (; base + 4 * kWave <= x; base += 4 * kWave) {
    const int c0 = base + 4 * lane;
    const ScanF4 C0 = *(const ScanF4*)(rc + c0 - 2), C1 = *(const ScanF4*)(rc + c0 + 2);
    const ScanF4 P0 = *(const ScanF4*)(rp + c0 - 1);
    const ScanF2 P1 = *(const ScanF2*)(rp + c0 + 3);
    const ScanF4 N0 = *(const ScanF4*)(rn + c0 - 1);
    const ScanF4 NN = *(const ScanF4*)(rnn + c0);
    const ScanF4 PP = *(const ScanF4*)(rpp + c0);
    const float C[8] = {C0.v[0], C0.v[1], C0.v[2], C0.v[3], C1.v[0], C1.v[1], C1.v[2], C1.v[3]};  // rc[c0 - 2 + i]
    const float P[6] = {P0.v[0], P0.v[1], P0.v[2], P0.v[3], P1.v[0], P1.v[1]};                      // rp[c0 - 1 + i]
#pragma unroll
    for (int u = 0; u < 4; u++) {  // ascending columns: the strict < keeps the first minimum
      const float Rq[8] = {NN.v[u], N0.v[u], C[u + 4], C[u + 2], C[u], P[u + 2], P[u], PP.v[u]};
      const float s = ssd8_tree(Lp, Rq);
      if (s < best) { best = s; match = c0 + u; }
    }
  }
#ifndef ODO_SCAN_UNROLL
#define ODO_SCAN_UNROLL 2
#endif
  // the rest: lanes = consecutive candidate columns. ODO_SCAN_UNROLL candidates per lane and trip, all their taps loaded before
  // any is used (more loads in flight per wave)
  int rx = base + lane;
  for (; rx + (ODO_SCAN_UNROLL - 1) * kWave < x; rx += ODO_SCAN_UNROLL * kWave) {
    float Rq[ODO_SCAN_UNROLL][8];
#pragma unroll
    for (int u = 0; u < ODO_SCAN_UNROLL; u++) {
      const int c = rx + u * kWave;
      Rq[u][0] = rnn[c]; Rq[u][1] = rn[c - 1]; Rq[u][2] = rc[c + 2]; Rq[u][3] = rc[c]; Rq[u][4] = rc[c - 2];
      Rq[u][5] = rp[c + 1]; Rq[u][6] = rp[c - 1]; Rq[u][7] = rpp[c];
    }
#pragma unroll
    for (int u = 0; u < ODO_SCAN_UNROLL; u++) {
      const float s = ssd8_tree(Lp, Rq[u]);
      if (s < best) { best = s; match = rx + u * kWave; }
    }
  }
  for (; rx < x; rx += kWave) {
    const float Rp[8] = {rnn[rx], rn[rx - 1], rc[rx + 2], rc[rx], rc[rx - 2], rp[rx + 1], rp[rx - 1], rpp[rx]};
    const float s = ssd8_tree(Lp, Rp);
    if (s < best) { best = s; match = rx; }  // :385-386
  }
  // argmin over the wave without a trip through LDS: inside each row of 16 lanes by DPP (xor 1, xor 2, mirror of 8, mirror of 16),
  // then the four row results through scalar registers
  scan_min_step<0xB1>(best, match);   // quad_perm [1, 0, 3, 2]
  scan_min_step<0x4E>(best, match);   // quad_perm [2, 3, 0, 1]
  scan_min_step<0x141>(best, match);  // row_half_mirror
  scan_min_step<0x140>(best, match);  // row_mirror
  float wb = lane_f(best, 0);
  int wm = __builtin_amdgcn_readlane(match, 0);
#pragma unroll
  for (int r = 1; r < 4; r++) {
    const float ob = lane_f(best, 16 * r);
    const int om = __builtin_amdgcn_readlane(match, 16 * r);
    const bool take = ob < wb || (ob == wb && om < wm);
    wb = take ? ob : wb;
    wm = take ? om : wm;
  }
  if (lane == 0) {
    float dd = 0.0f;
    const bool hit = !(wb > ssd_th);  // :388
    if (hit) {
      const float dsp = (float)(x - wm);     // :391
      dd = dsp / (f0 * baseline);            // :394
      disp[(size_t)y * cols + x] = dsp;
      dep[(size_t)y * cols + x] = dd;
    }
    d0[slot] = dd;
    matched[slot] = hit ? 1 : 0;  // counted later by a reduction (a single-address atomic serialises at ~12 ns each)
  }
}
ODO_KERNEL void __launch_bounds__(256) depth_disparity_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                               int rows, int cols, int bnd, int max_disp, float ssd_th,
                                                               float f0, float baseline, const uint32_t* __restrict__ pts,
                                                               const int* __restrict__ cnt, float* __restrict__ disp,
                                                               float* __restrict__ dep, float* __restrict__ d0,
                                                               uint8_t* __restrict__ matched) {
  depth_disparity_kernel_body(L, R, rows, cols, bnd, max_disp, ssd_th, f0, baseline, pts, cnt, disp, dep, d0, matched);
}

// ComputeDepth's three output images are zero everywhere but at the selected points (ref: src/depth_estimate.cpp:388-397,176-191 write
// at the points; the images are zero-filled in front, forced deviation #14): per point slot {pixel index, val, disp, dep}, 13 bytes
// instead of 9 per pixel — what a caller whose images live in host memory it owns (cv::Mat) needs to rebuild them there. Unused
// slots get index 0xffffffff.
ODO_KERNEL void __launch_bounds__(256) depth_compact_outputs_kernel(const uint32_t* __restrict__ pts, const int* __restrict__ cnt, int cols,
                                                                    const uint8_t* __restrict__ val, const float* __restrict__ disp,
                                                                    const float* __restrict__ dep, uint32_t* __restrict__ o_idx,
                                                                    float* __restrict__ o_disp, float* __restrict__ o_dep,
                                                                    uint8_t* __restrict__ o_val) {
  const int s = (int)(blockIdx.x * 256 + threadIdx.x);
  if (s >= kSelBlocks * kSelCap) return;
  uint32_t idx = 0xffffffffu;
  float ds = 0.0f, dp = 0.0f;
  uint8_t v = 0;
  if ((s % kSelCap) < cnt[s / kSelCap]) {
    const uint32_t pk = pts[s];
    idx = (pk >> 16) * (uint32_t)cols + (pk & 0xffffu);
    v = val[idx]; ds = disp[idx]; dp = dep[idx];
  }
  o_idx[s] = idx; o_disp[s] = ds; o_dep[s] = dp; o_val[s] = v;
}

struct DepthLmStats {
  int iters;
  float cost;
  int n_valid;
  int n_selected;
  int n_matched;
  int status;
};

// DepthOptimization (ref: src/depth_estimate.cpp:80-198, 200-242). All points share one accept/reject decision
// per iteration (:150-161). One launch per evaluation, one thread per point slot; the decision for evaluation
// k-1 is re-derived at the top of launch k by EVERY block from the per-block partial sums of launch k-1 (fixed
// summation order -> all blocks agree bit for bit), so no grid barrier and no host round trip is needed.
// State and partials are double-buffered by launch parity. Uses the UNBLURRED images (:67).
// DepthLmState, depth_lm_begin / depth_lm_decide / depth_lm_advance: odo_math.h (shared with the host build the tests pin)
constexpr int kDlmBlock = 256;
constexpr int kDlmBlocks = kSelBlocks * kSelCap / kDlmBlock;  // 160

// Host-mapped progress words of the depth LM: [0] = index of the last launch whose block 0 has run (+1),
// [1] = 1 once the loop has stopped (the host then skips the launches it has not issued yet).
__device__ __forceinline__ void dlm_report(int* host_prog, int k, int done) {
  if (!host_prog) return;
  if (done) __hip_atomic_store(host_prog + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_prog, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void depth_lm_step_kernel_body(
    int k, const float* __restrict__ left, const float* __restrict__ right, int cols, const uint32_t* __restrict__ pts,
    const int* __restrict__ cnt, const float* __restrict__ d0, float* __restrict__ scratch /* 6 x nslots */,
    DepthLmState* __restrict__ state /* [2] */, double* __restrict__ part_e /* [2][blocks] */,
    int* __restrict__ part_n /* [2][blocks] */, float tx, float fx, float huber_delta, float lambda0, float precision,
    int max_iters, int* __restrict__ host_prog) {
  constexpr int nslots = kSelBlocks * kSelCap;
  __shared__ double shs[kDlmBlock / 64];
  __shared__ int shn[kDlmBlock / 64];
  const int t = threadIdx.x;
  const int s = blockIdx.x * kDlmBlock + t;
  float* cur = scratch;
  float* pre = scratch + nslots;
  float* tmp = scratch + 2 * nslots;
  float* res = scratch + 3 * nslots;  // residual of the last evaluated step (:231), -1000 = out of range
  float* jt = scratch + 4 * nslots;   // diagonal of JtWJ (:234)
  float* bb = scratch + 5 * nslots;   // -JtWr (:235)
  const bool ok = (s % kSelCap) < cnt[s / kSelCap];
  DepthLmState st;
  float my_tmp;
  // Everything below that does not depend on the accept / reject decision is fetched up front, in one batch with the
  // state and the partial sums: a launch then has two dependent trips to memory (this batch, the right-image taps)
  // instead of five.
  float pf_pre = 0.0f, pf_tmp = 0.0f, pf_jt = 0.0f, pf_bb = 0.0f, pf_left = 0.0f;
  uint32_t pf_pk = 0u;
  if (ok) {
    pf_pk = pts[s];
    if (k > 0) { pf_pre = pre[s]; pf_tmp = tmp[s]; pf_jt = jt[s]; pf_bb = bb[s]; }
    pf_left = left[(size_t)(pf_pk >> 16) * cols + (pf_pk & 0xffffu)];
  }
  if (k == 0) {
    depth_lm_begin(&st, lambda0, max_iters);
    const float v = ok ? d0[s] : 0.0f;
    cur[s] = v; pre[s] = 0.0f; tmp[s] = v; res[s] = 0.0f; jt[s] = 1.0f; bb[s] = 0.0f;
    my_tmp = v;
    if (st.done) { if (blockIdx.x == 0 && t == 0) { state[1] = st; dlm_report(host_prog, k, 1); } return; }
  } else {
    st = state[k & 1];
    if (st.done) { if (blockIdx.x == 0 && t == 0) { state[(k + 1) & 1] = st; dlm_report(host_prog, k, 1); } return; }
    // ---- decision for evaluation k-1 (every block, identical arithmetic) ----
    const double* pe = part_e + ((k - 1) & 1) * kDlmBlocks;
    const int* pn = part_n + ((k - 1) & 1) * kDlmBlocks;
    // fixed-order fold of the 160 per-block partials: every wave does it redundantly (three rows per lane, then a
    // butterfly), so no LDS and no block barrier sit between the partial sums and the decision
    double fe = 0.0;
    int fn = 0;
    {
      const int lane = t & 63;
      double fnd = 0.0;
#pragma unroll
      for (int j = 0; j < (kDlmBlocks + 63) / 64; j++) {
        const int b = lane + 64 * j;
        if (b < kDlmBlocks) { fe += pe[b]; fnd += (double)pn[b]; }
      }
      fe = wave_sum64(fe);                 // (the order of depth_lm_persistent_kernel's fold: both give the same bits)
      fn = (int)wave_sum64(fnd);           // integers: exact
    }
    const float err_now = (1.0f / (float)fn) * (float)fe;  // :239
    const int mode = depth_lm_decide(&st, err_now, precision);  // 0 reject+continue, 1 accept+continue, 2 reject+break, 3 accept+break
    my_tmp = 0.0f;
    if (mode != 2 && ok) {
      float c;
      if (mode == 0) c = pf_pre;            // :153
      else { c = pf_tmp; pre[s] = c; }      // :155-156
      cur[s] = c;
      if (mode != 3) {
        const float jj = pf_jt;
        const float A = jj + st.lambda * jj;  // :164
        const float dd = (1.0f / A) * pf_bb;  // :165
        my_tmp = dd + c;                      // :166
        tmp[s] = my_tmp;
      }
    }
    depth_lm_advance(&st, mode, max_iters);   // :167, :141
    if (st.done) { if (blockIdx.x == 0 && t == 0) { state[(k + 1) & 1] = st; dlm_report(host_prog, k, 1); } return; }
  }
  // ---- evaluation k at tmp (ComputeResidualJacobian :200-242) ----
  double esum = 0.0;
  int nact = 0;
  if (ok) {
    const uint32_t pk = pf_pk;
    const int x = (int)(pk & 0xffffu), y = (int)(pk >> 16);
    const float wf = floorf((float)x - tx * fx * my_tmp);  // :217
    if (!(wf >= 2.0f) || !(wf <= (float)(cols - 2))) {     // :219-223
      jt[s] = 0.0f; bb[s] = 0.0f; res[s] = -1000.0f;
    } else {
      const int wx = (int)wf;
      const float* Rr = right + (size_t)y * cols;
      const float r_i = pf_left - Rr[wx];                                                // :226
      const float w_i = (fabsf(r_i) <= huber_delta) ? 1.0f : huber_delta / fabsf(r_i);   // :228
      const float r_diff = tx * fx * 0.5f * (Rr[wx + 1] - Rr[wx - 1]);                   // :229
      res[s] = fabsf(r_i);
      nact = 1;
      esum = (double)(r_i * r_i * w_i);                                                  // :233
      jt[s] = r_diff * r_diff * w_i;                                                     // :234
      bb[s] = -r_diff * w_i * r_i;                                                       // :235
    }
  }
  // block sum: wave_sum64 inside each wave, then the four wave sums in a fixed order (one barrier)
  esum = wave_sum64(esum);
  nact = __popcll(__ballot(nact != 0));
  if ((t & 63) == 0) { shs[t >> 6] = esum; shn[t >> 6] = nact; }
  __syncthreads();
  if (t == 0) {
    part_e[(k & 1) * kDlmBlocks + blockIdx.x] = (shs[0] + shs[1]) + (shs[2] + shs[3]);
    part_n[(k & 1) * kDlmBlocks + blockIdx.x] = (shn[0] + shn[1]) + (shn[2] + shn[3]);
    if (blockIdx.x == 0) { state[(k + 1) & 1] = st; dlm_report(host_prog, k, 0); }
  }
}
ODO_KERNEL void __launch_bounds__(kDlmBlock) depth_lm_step_kernel(
    int k, const float* __restrict__ left, const float* __restrict__ right, int cols, const uint32_t* __restrict__ pts,
    const int* __restrict__ cnt, const float* __restrict__ d0, float* __restrict__ scratch /* 6 x nslots */,
    DepthLmState* __restrict__ state /* [2] */, double* __restrict__ part_e /* [2][blocks] */,
    int* __restrict__ part_n /* [2][blocks] */, float tx, float fx, float huber_delta, float lambda0, float precision,
    int max_iters, int* __restrict__ host_prog) {
  depth_lm_step_kernel_body(k, left, right, cols, pts, cnt, d0, scratch, state, part_e, part_n, tx, fx, huber_delta, lambda0, precision, max_iters, host_prog);
}

// =============================================================================================
// DepthOptimization in ONE launch (round 4): the whole iteration of ref: src/depth_estimate.cpp:141-168 plus the write-back and
// filters of :176-191 inside a persistent kernel, instead of a launch per iteration (~27 dependent launches of 4.8 us per frame).
// K = 80 workgroups of 512 threads on one XCD; workgroup g owns the two 256-slot virtual blocks 2 g, 2 g + 1 — exactly blocks
// of depth_lm_step_kernel, wave w of it the 64 slots the step kernel's wave (w & 3) of block 2 g + (w >> 2) has — and every thread
// ONE slot, whose whole state (current / previous / trial inverse depth, diagonal of JtWJ, -JtWr, last residual, the left pixel)
// stays in registers for all iterations. Per iteration only the global error crosses workgroups (:150, the one accept / reject
// decision all points share): every virtual block's {sum w r^2, N} goes out as ONE tagged 16-byte granule pair, wave 0 of every
// workgroup gathers the 160 pairs with L1-bypassing loads (no flag, no fence) and folds them in depth_lm_step_kernel's order, so
// both kernels give the same bits: the step launches stay as the fall-back (a launch that cannot get its workgroups resident
// gives up within the wait bound, reports it through the statistics, and the host runs the job again on them) and as the batched
// tracker's path. An iteration is two trips through the XCD's L2 (the pairs; the three right-image taps) and two workgroup barriers.
// (First shape, measured: 32 workgroups x 256 threads x FIVE slots per thread, every wave gathering: 7.4 k cycles per iteration —
// gather 2.8 k, decide + update 1.3 k, evaluation 1.9 k, five wave sums + publish 1.4 k — no faster than the launches it replaced.)
// A granule = {32 bits of payload | 16 bits of extra payload | 16-bit tag}; tag = (launch epoch & 0xff) << 8 | iteration + 1 (the
// host clears the buffer whenever the epoch's low byte starts over, so a stale granule cannot pass for a new one); the pairs are
// double-buffered by iteration parity (wave 0 cannot be two iterations ahead of the slowest: it needs everybody's pair of k - 1).
// =============================================================================================
constexpr int kDpThreads = 512;
constexpr int kDpVb = kDpThreads / kDlmBlock;    // virtual blocks per workgroup: 2
constexpr int kDpK = kDlmBlocks / kDpVb;         // workgroups: 80 (they wait for each other: all must be resident at once — three fit a CU.
                                                 // 40 x 1 024 threads would need <= 64 VGPRs to fit two per CU: 13 spills)
constexpr int kDpMaxIters = 250;                 // the 8-bit iteration field of the tags
constexpr int kDpXbufWords = 2 * kDlmBlocks * 2 + kDpK + 2;   // two parities of 160 pairs + one placement word per workgroup + the "launch abandoned" word
static_assert(kDlmBlocks % kDpVb == 0 && kDpK <= kDpThreads, "every workgroup owns the same number of virtual blocks; one thread per workgroup");
struct DepthPersistArgs {
  const float *left, *right;
  int cols;
  const uint32_t* pts;
  const int* cnt;
  const float* d0;
  const uint8_t* matched;
  DepthLmState* state_out;      // final driver state (iterations, last error) for depth_stats_kernel
  float tx, fx, huber_delta, lambda0, precision;
  int max_iters;
  float photo_th, min_depth, max_depth;
  uint8_t* val;
  float* dep;
  int* counts;                  // [kDlmBlocks][3] {valid, selected, matched} per virtual block
  unsigned long long* xbuf;     // kDpXbufWords
  unsigned epoch;
  unsigned wait_ticks;          // bound of one wait (100 MHz wall clock); 0: kFineWaitTicks
  int* gave_up;                 // device word: set to 1 by a workgroup whose wait ran out (depth_stats_kernel reads and clears it)
  int fault;                    // test hook (ODO_DEPTH_PERSIST_FAULT): virtual block 0's pair is never published
  unsigned long long* dbg;      // diagnostic (ODO_DEPTH_STAMPS): cycle sums of workgroup 0's phases, else null
  int home;                     // the XCC id of the XCD this estimator's persistent launch runs on (fine_on_home); < 0: by block class
  int cls;                      // the block class (blockIdx % 8) that takes part when home < 0
};
// (the launch's workgroup g of kDpK; which blocks of the grid those are is the kernel's business)
__device__ __forceinline__ void depth_lm_persistent_body(const DepthPersistArgs& a, const int g) {
  const unsigned wait_limit = a.wait_ticks ? a.wait_ticks : kFineWaitTicks;
  const unsigned long long t_entry = (unsigned long long)wall_clock64();
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __shared__ double sh_e[kDpThreads / 64];
  __shared__ int sh_n[kDpThreads / 64];
  __shared__ int sh_c[kDpThreads / 64][3];
  __shared__ double fold_e;
  __shared__ int fold_n, local_sh, bail_sh;
  unsigned long long* place = a.xbuf + 2 * kDlmBlocks * 2;
  const unsigned ep = (a.epoch & 0xffu) << 8;
  unsigned long long* abandoned = place + kDpK;   // = epoch + 1 once a workgroup of this launch has given up: the ones dispatched later follow at once
  if (t == 0) {
    local_sh = 1;
    bail_sh = (__hip_atomic_load(abandoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(a.epoch + 1u)) ? 1 : 0;
    __hip_atomic_store(place + g, ((unsigned long long)(unsigned)fine_xcc_id() << 32) | (a.epoch + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- my slot: everything an iteration needs, in registers ----
  const int s = g * kDpThreads + t;               // = (2 g + (wv >> 2)) * 256 + (t & 255): the step kernel's slot of the same wave
  const bool ok = (s % kSelCap) < a.cnt[s / kSelCap];
  const uint32_t pk = ok ? a.pts[s] : 0u;
  const int px = (int)(pk & 0xffffu), py = (int)(pk >> 16);
  const float lft = ok ? a.left[(size_t)py * a.cols + px] : 0.0f;
  const float* Rr = a.right + (size_t)py * a.cols;
  float cur = ok ? a.d0[s] : 0.0f, pre = 0.0f, tmp = cur, res = 0.0f, jt = 1.0f, bb = 0.0f;   // :121-137
  __syncthreads();
  // does every workgroup of this launch share my XCD? (then plain stores stay in its L2, where the gather loads find them)
  if (t < kDpK) {   // thread i asks about workgroup i
    unsigned long long pw = 0;
    bool got = false;
    const FineDeadline dl = FineDeadline::begin(wait_limit);
    for (int spin = 0; !got; spin++) {
      pw = __hip_atomic_load(place + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      got = ((unsigned)pw == a.epoch + 1u);
      if (!got && dl.expired(spin)) break;
      // 60 us without the others AND a pose-LM persistent launch half dispatched: the two launches are in each other's way (see
      // g_lm_fine_dispatch) and this one, which is not the frame's critical chain, yields at once instead of after the whole bound
      if (!got && (spin & 31) == 31 &&
          (((unsigned long long)wall_clock64() - dl.t0 > 6000ull && lm_fine_mid_dispatch()) ||
           __hip_atomic_load(abandoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(a.epoch + 1u))) break;
    }
    if (!got) {
      bail_sh = 1;
      __hip_atomic_store(abandoned, (unsigned long long)(a.epoch + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else if ((int)(pw >> 32) != fine_xcc_id()) local_sh = 0;
  }
  __syncthreads();
  const bool local = local_sh != 0;
  DepthLmState st;
  depth_lm_begin(&st, a.lambda0, a.max_iters);      // :92-96
  const bool placed = bail_sh == 0;   // (workgroup-uniform: read behind the barrier above)
  unsigned long long c_gather = 0, c_decide = 0, c_eval = 0, c_sum = 0, c_it = 0, c_last = ODO_DBG(a) ? __builtin_readcyclecounter() : 0;
  const unsigned long long c_begin = c_last;
  auto lap = [&](unsigned long long& sum) {
    if (ODO_DBG(a)) { const unsigned long long now = __builtin_readcyclecounter(); sum += now - c_last; c_last = now; }
  };
  int exit_k = -1;
  for (int k = 0; placed && !st.done; k++) {   // (st: every thread derives the same state)
    exit_k = k;
    c_it++;
    if (k > 0) {
      if (t < 64) {
        // ---- wave 0: gather the pairs of evaluation k - 1, fold them in depth_lm_step_kernel's order ----
        const unsigned tag = ep | (unsigned)k;         // evaluation k - 1 was published with iteration field k
        const unsigned long long* buf = a.xbuf + (size_t)((k - 1) & 1) * kDlmBlocks * 2;
        FineG2 q[(kDlmBlocks + 63) / 64];
        bool all = false;
        FineDeadline dl = {0ull, 0ull, wait_limit};
        for (int spin = 0; !all; spin++) {
          if (spin == 1) dl = FineDeadline::begin(wait_limit);
          if (spin > 0 && dl.expired(spin)) break;
          bool mine = true;
#pragma unroll
          for (int u = 0; u < (kDlmBlocks + 63) / 64; u++) {
            const int b = lane + 64 * u;
            if (b < kDlmBlocks) q[u] = *(const volatile FineG2Global*)(buf + 2 * b);
          }
#pragma unroll
          for (int u = 0; u < (kDlmBlocks + 63) / 64; u++) {
            const int b = lane + 64 * u;
            if (b < kDlmBlocks) mine = mine && ((q[u].x & 0xffffu) == tag) && ((q[u].z & 0xffffu) == tag);
          }
          all = __all(mine);
        }
        double fe = 0.0, fnd = 0.0;
#pragma unroll
        for (int u = 0; u < (kDlmBlocks + 63) / 64; u++) {
          const int b = lane + 64 * u;
          if (b < kDlmBlocks) {
            fe += __longlong_as_double((long long)(((unsigned long long)q[u].y << 32) | (unsigned long long)q[u].w));
            fnd += (double)(q[u].x >> 16);
          }
        }
        fe = wave_sum64(fe);
        const int fn = (int)wave_sum64(fnd);
        if (lane == 0) { fold_e = fe; fold_n = fn; if (!all) bail_sh = 1; }
      }
      __syncthreads();      // the error of evaluation k - 1 is in LDS — or a wait ran out, for the whole workgroup at the same point
      lap(c_gather);
      if (bail_sh) break;
      // ---- decision for evaluation k - 1 (identical arithmetic in every thread) ----
      const float err_now = (1.0f / (float)fold_n) * (float)fold_e;   // :239
      const int mode = depth_lm_decide(&st, err_now, a.precision);    // :150-161
      if (mode != 2 && ok) {
        float c;
        if (mode == 0) c = pre;                       // :153
        else { c = tmp; pre = c; }                    // :155-156
        cur = c;
        if (mode != 3) {
          const float A = jt + st.lambda * jt;        // :164
          const float dd = (1.0f / A) * bb;           // :165
          tmp = dd + c;                               // :166
        }
      }
      depth_lm_advance(&st, mode, a.max_iters);       // :167, :141
      if (ODO_DBG(a)) asm volatile("" ::"v"(tmp));
      lap(c_decide);
      if (st.done) break;
    }
    // ---- evaluation k at tmp (ComputeResidualJacobian :200-242) ----
    double esum = 0.0;
    bool act = false;
    if (ok) {
      const float wf = floorf((float)px - a.tx * a.fx * tmp);           // :217
      if (!(wf >= 2.0f) || !(wf <= (float)(a.cols - 2))) {              // :219-223
        jt = 0.0f; bb = 0.0f; res = -1000.0f;
      } else {
        const int wx = (int)wf;
        const float r_i = lft - Rr[wx];                                                          // :226
        const float w_i = (fabsf(r_i) <= a.huber_delta) ? 1.0f : a.huber_delta / fabsf(r_i);     // :228
        const float r_diff = a.tx * a.fx * 0.5f * (Rr[wx + 1] - Rr[wx - 1]);                     // :229
        res = fabsf(r_i);
        act = true;
        esum = (double)(r_i * r_i * w_i);                                                        // :233
        jt = r_diff * r_diff * w_i;                                                              // :234
        bb = -r_diff * w_i * r_i;                                                                // :235
      }
    }
    if (ODO_DBG(a)) asm volatile("" ::"v"(esum));
    lap(c_eval);
    {
      const double ws = wave_sum64(esum);
      const int wn = __popcll(__ballot(act));
      if (lane == 0) { sh_e[wv] = ws; sh_n[wv] = wn; }
    }
    __syncthreads();
    if (t < kDpVb) {
      const int vb = g * kDpVb + t;
      const double e = (sh_e[4 * t] + sh_e[4 * t + 1]) + (sh_e[4 * t + 2] + sh_e[4 * t + 3]);     // depth_lm_step_kernel's block sum
      const int n = (sh_n[4 * t] + sh_n[4 * t + 1]) + (sh_n[4 * t + 2] + sh_n[4 * t + 3]);
      const unsigned tag = ep | (unsigned)(k + 1);
      const unsigned long long bits = (unsigned long long)__double_as_longlong(e);
      const unsigned long long g0 = ((bits >> 32) << 32) | ((unsigned long long)(unsigned)n << 16) | tag, g1 = (bits << 32) | tag;
      unsigned long long* dst = a.xbuf + (size_t)(k & 1) * kDlmBlocks * 2 + 2 * vb;
      if (!(a.fault && vb == 0)) {
        if (local) { dst[0] = g0; dst[1] = g1; }
        else {
          __hip_atomic_store(dst, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(dst + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    lap(c_sum);
    // (sh_e / sh_n are rewritten only behind the next iteration's first barrier, which wave 0 — their reader — reaches after this)
  }
  __syncthreads();
  const bool bailed = bail_sh != 0;   // (workgroup-uniform behind the barrier)
  if (bailed && t == 0) {
    __hip_atomic_store(a.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // diagnostic words behind the flag (ODO_LOG_GIVEUPS prints them): a workgroup that gave up, its XCC id, whether it had been
    // placed (0: the wait for the other workgroups' placement words ran out; 1: a wait for an iteration's sums), how many gave up
    a.gave_up[1] = g; a.gave_up[2] = fine_xcc_id(); a.gave_up[3] = placed ? 1 : 0;
    atomicAdd(a.gave_up + 4, 1);
    // per workgroup: the iteration it was in (-1: never placed), the 100 MHz clock at its entry and at its exit (low 32 bits)
    a.gave_up[16 + 4 * g] = exit_k; a.gave_up[16 + 4 * g + 1] = (int)(unsigned)t_entry; a.gave_up[16 + 4 * g + 2] = (int)(unsigned)wall_clock64();
    a.gave_up[16 + 4 * g + 3] = fine_xcc_id();
  }
  // ---- write-back + filters (:176-191) and the per-block counts of depth_finalize_kernel ----
  bool good = false;
  if (ok && !bailed) {
    const size_t o = (size_t)py * a.cols + px;
    good = !(res > a.photo_th || res == -1000.0f);
    if (good && (1.0f / cur > a.max_depth || 1.0f / cur < a.min_depth)) good = false;
    a.val[o] = good ? 1 : 0;
    a.dep[o] = good ? cur : 0.0f;
  }
  {
    const int c0 = __popcll(__ballot(good)), c1 = __popcll(__ballot(ok)), c2 = __popcll(__ballot(ok && a.matched[s] != 0));
    if (lane == 0) { sh_c[wv][0] = c0; sh_c[wv][1] = c1; sh_c[wv][2] = c2; }
  }
  __syncthreads();
  if (t < kDpVb * 3) {
    const int j = t / 3, qn = t % 3;
    a.counts[(g * kDpVb + j) * 3 + qn] = (sh_c[4 * j][qn] + sh_c[4 * j + 1][qn]) + (sh_c[4 * j + 2][qn] + sh_c[4 * j + 3][qn]);
  }
  if (g == 0 && t == 0) *a.state_out = st;
  if (ODO_DBG(a) && g == 0 && t == 0) {
    ODO_DBG(a)[0] += c_gather; ODO_DBG(a)[1] += c_decide; ODO_DBG(a)[2] += c_eval; ODO_DBG(a)[3] += c_sum; ODO_DBG(a)[4] += c_it; ODO_DBG(a)[5] += 1;
    ODO_DBG(a)[6] += local ? 1 : 0; ODO_DBG(a)[7] += __builtin_readcyclecounter() - c_begin;
  }
}

ODO_KERNEL void __launch_bounds__(kDpThreads) depth_lm_persistent_kernel(DepthPersistArgs a) {
  // Which eighth of the grid takes part: block class `cls` (blocks with blockIdx % 8 == cls land on one XCD: the dispatcher deals every
  // grid round-robin from the same XCD). NOT the pose LM's class 0: on one XCD the two persistent launches cannot share a CU (416 +
  // 160 VGPRs per SIMD), so whenever they overlapped in time one waited for the other's CUs, and when both were dispatched at the same
  // moment each got part of the XCD and waited for workgroups that could not be placed (ODO_LOG_GIVEUPS: 76 of 80 workgroups entered
  // together, the last four only when the first ones had given up — on XCC 6, the pose LM's, every time). home >= 0: by XCC id.
  if (a.home >= 0 ? fine_xcc_id() != a.home : (int)(blockIdx.x & 7u) != a.cls) return;
  depth_lm_persistent_body(a, (int)(blockIdx.x >> 3));
}
// Batched twin (odo_tracker_batch, up to four sequences): sequence j's 80 workgroups are block class (tab[0].cls + j) % 8 — an XCD of its
// own beside the batched pose LM's, whose sequences sit on classes 0 .. 3 —, each with its own exchange buffer, epoch and give-up word.
ODO_KERNEL void __launch_bounds__(kDpThreads) depth_lm_persistent_batch_kernel(const DepthPersistArgs* __restrict__ tab, int n, XccIds xcc) {
  int r = (int)(blockIdx.x & 7u);   // the XCD this block sits on, named as lm_fine_kernel_batch names it (XCC ids unknown: the block class)
  if (xcc.id[0] >= 0) {
    const int mine = fine_xcc_id();
#pragma unroll
    for (int c = 0; c < 8; c++) if (xcc.id[c] == mine) r = c;
  }
  const int j = (r - tab[0].cls) & 7;
  if (j >= n) return;
  depth_lm_persistent_body(tab[j], (int)(blockIdx.x >> 3));
}

// Write-back + filters (ref: src/depth_estimate.cpp:176-191) and per-block counts {valid, selected, matched}.
// run_lm == 0: disparity-only entry (counts only, mask/depth untouched).
__device__ __forceinline__ void depth_finalize_kernel_body(int run_lm, int cols, const uint32_t* __restrict__ pts,
                                                                   const int* __restrict__ cnt,
                                                                   const uint8_t* __restrict__ matched,
                                                                   const float* __restrict__ scratch, float photo_th,
                                                                   float min_depth, float max_depth,
                                                                   uint8_t* __restrict__ val, float* __restrict__ dep,
                                                                   int* __restrict__ counts /* [blocks][3] */) {
  constexpr int nslots = kSelBlocks * kSelCap;
  __shared__ int sh[3][kDlmBlock];
  const int t = threadIdx.x;
  const int s = blockIdx.x * kDlmBlock + t;
  const bool ok = (s % kSelCap) < cnt[s / kSelCap];
  int good = 0;
  if (ok && run_lm) {
    const uint32_t pk = pts[s];
    const size_t o = (size_t)(pk >> 16) * cols + (pk & 0xffffu);
    const float c = scratch[s];
    const float rs = scratch[3 * nslots + s];
    bool g = !(rs > photo_th || rs == -1000.0f);
    if (g && (1.0f / c > max_depth || 1.0f / c < min_depth)) g = false;
    val[o] = g ? 1 : 0;
    dep[o] = g ? c : 0.0f;
    good = g;
  }
  sh[0][t] = good;
  sh[1][t] = ok ? 1 : 0;
  sh[2][t] = (ok && matched[s]) ? 1 : 0;
  __syncthreads();
  for (int o = kDlmBlock / 2; o > 0; o >>= 1) {
    if (t < o) { sh[0][t] += sh[0][t + o]; sh[1][t] += sh[1][t + o]; sh[2][t] += sh[2][t + o]; }
    __syncthreads();
  }
  if (t < 3) counts[blockIdx.x * 3 + t] = sh[t][0];
}
ODO_KERNEL void __launch_bounds__(kDlmBlock) depth_finalize_kernel(int run_lm, int cols, const uint32_t* __restrict__ pts,
                                                                   const int* __restrict__ cnt,
                                                                   const uint8_t* __restrict__ matched,
                                                                   const float* __restrict__ scratch, float photo_th,
                                                                   float min_depth, float max_depth,
                                                                   uint8_t* __restrict__ val, float* __restrict__ dep,
                                                                   int* __restrict__ counts /* [blocks][3] */) {
  depth_finalize_kernel_body(run_lm, cols, pts, cnt, matched, scratch, photo_th, min_depth, max_depth, val, dep, counts);
}

// gave_up (optional): device word a persistent depth-LM launch sets when one of its waits ran out — the statistics then carry
// status -2 (the host runs the job again on the step launches) and the word is cleared for the next job.
__device__ __forceinline__ void depth_stats_kernel_body(int run_lm, int n_launches, const int* __restrict__ counts,
                                                                const DepthLmState* __restrict__ state,
                                                                DepthLmStats* __restrict__ stats /* host-mapped */,
                                                                int* __restrict__ done_flag, int token, int* __restrict__ gave_up = nullptr) {
  __shared__ int sh[3][kDlmBlock];
  const int t = threadIdx.x;
  for (int q = 0; q < 3; q++) sh[q][t] = (t < kDlmBlocks) ? counts[t * 3 + q] : 0;
  __syncthreads();
  for (int o = kDlmBlock / 2; o > 0; o >>= 1) {
    if (t < o) { sh[0][t] += sh[0][t + o]; sh[1][t] += sh[1][t + o]; sh[2][t] += sh[2][t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    const DepthLmState st = state[n_launches & 1];  // the state written by the last launch
    stats->iters = run_lm ? st.iter : 0;
    stats->cost = run_lm ? st.err_now : 0.0f;
    stats->n_valid = sh[0][0];
    stats->n_selected = sh[1][0];
    stats->n_matched = sh[2][0];
    stats->status = (run_lm && sh[0][0] < 500) ? -1 : 0;  // :192-197
    if (gave_up && *gave_up) { stats->status = -2; *gave_up = 0; }
    __hip_atomic_store(done_flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
ODO_KERNEL void __launch_bounds__(kDlmBlock) depth_stats_kernel(int run_lm, int n_launches, const int* __restrict__ counts,
                                                                const DepthLmState* __restrict__ state,
                                                                DepthLmStats* __restrict__ stats /* host-mapped */,
                                                                int* __restrict__ done_flag, int token, int* __restrict__ gave_up) {
  depth_stats_kernel_body(run_lm, n_launches, counts, state, stats, done_flag, token, gave_up);
}

// Launchers of the single tracker's chain kernels (defined in lm_chain_kernels.hip, see the top of this header).
// variant of the fine launch: 0 = lean (Huber / L2, floor sampling, nothing recorded), 1 = trace rows and / or bilinear sampling, 2 =
// t-distribution weights. dispatch_words: device address of the main unit's g_lm_fine_dispatch.
#if ODO_PHASE_STAMPS
void lm_chain_diag_read(unsigned long long out[24]);
#endif
hipError_t lm_chain_setup();   // the coarse kernels' dynamic LDS limit; once per process and device
void launch_lm_coarse(bool lean, hipStream_t s, const StepArgs& a, int min_level);
void launch_lm_coarse_armed(hipStream_t s, const StepArgs& a, int min_level);   // lean build only (a trackers' optimiser)
void launch_lm_fine(int variant, int blocks, hipStream_t s, const StepArgs& a, int K, unsigned long long* xbuf, int fault, int lo_level,
                    unsigned* dispatch_words);

}  // namespace odo
