// What would one LM iteration's exchange cost if the K workgroups of a persistent kernel, all on one XCD, replaced the
// launch-per-evaluation chain?  Every iteration each workgroup (a) "evaluates" (spins), (b) publishes its rows of partial sums as
// data-tagged 8-byte granules {payload, iteration} with sc1 (write-through) stores, (c) gathers ALL rows by polling the granules
// themselves with sc1 loads (no flag, no fence: MI355X guide, handoff-1to1), (d) wave 0 "runs the state machine" (spins).
// Reported: time per iteration minus the two spins = the price of the exchange, idle and with a streaming kernel beside it.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/xcd_allgather tools/microbench/xcd_allgather.hip && /tmp/xcd_allgather
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

constexpr int kThreads = 512;
constexpr int kGran = 58;  // granules per row: 29 doubles as {hi, tag} {lo, tag}

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ void spin(long long cycles) {
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
}
__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// grid = 8 * K blocks; only blocks with blockIdx.x % 8 == 0 take part (they share an XCD), the others exit at once
__global__ void __launch_bounds__(kThreads) coop(int K, int iters, int rows_per_wg, unsigned long long* buf, int spin_eval, int spin_sm,
                                                 unsigned long long* out, int* xcc) {
  if (blockIdx.x % 8 != 0) return;
  const int w = blockIdx.x / 8, t = threadIdx.x;
  if (t == 0) xcc[w] = xcc_id();
  const int rows = K * rows_per_wg, n = rows * kGran;
  unsigned long long acc = 0;
  const unsigned long long t0 = wall();
  for (int it = 1; it <= iters; it++) {
    unsigned long long* b = buf + (size_t)(it & 1) * n;
    spin(spin_eval);
    // publish my rows
    for (int i = t; i < rows_per_wg * kGran; i += kThreads) {
      const unsigned long long g = ((unsigned long long)(unsigned)(w * 1000 + i) << 32) | (unsigned)it;
      __hip_atomic_store(b + (size_t)w * rows_per_wg * kGran + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // gather every row: poll the granules themselves
    for (int i = t; i < n; i += kThreads) {
      unsigned long long g;
      do { g = __hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)g != (unsigned)it);
      acc += g >> 32;
    }
    __syncthreads();
    if (t < 64) spin(spin_sm);
    __syncthreads();
  }
  const unsigned long long t1 = wall();
  if (t == 0) { out[2 * w] = t1 - t0; }
  out[2 * w + 1] = acc;  // keep the loads
}

__global__ void stream_load(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  unsigned long long *buf, *out;
  int* xcc;
  hipMalloc(&buf, sizeof(unsigned long long) * 2 * 256 * kGran);
  hipMalloc(&out, sizeof(unsigned long long) * 2 * 64);
  hipMalloc(&xcc, sizeof(int) * 64);
  const size_t nbig = 64u << 20;  // 1 GiB of float4 traffic per rep pair
  float4 *a, *b;
  hipMalloc(&a, nbig * sizeof(float4) / 4);
  hipMalloc(&b, nbig * sizeof(float4) / 4);
  hipMemset(a, 0, nbig * sizeof(float4) / 4);
  const int iters = 2000;
  for (int load = 0; load < 2; load++) {
    for (int K : {1, 2, 4, 8, 16, 32}) {
      for (int rows_per_wg : {1, 4}) {
        for (int spins : {0, 1}) {
          const int se = spins ? 4000 : 0, ss = spins ? 6400 : 0;
          hipMemsetAsync(buf, 0, sizeof(unsigned long long) * 2 * 256 * kGran, s);
          hipStreamSynchronize(s);
          if (load) hipLaunchKernelGGL(stream_load, dim3(1024), dim3(256), 0, s2, a, b, nbig / 4, 6);  // 4 blocks per CU: half the wave slots stay free
          const auto h0 = std::chrono::steady_clock::now();
          hipLaunchKernelGGL(coop, dim3(8 * K), dim3(kThreads), 0, s, K, iters, rows_per_wg, buf, se, ss, out, xcc);
          hipStreamSynchronize(s);
          const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
          hipStreamSynchronize(s2);
          std::vector<unsigned long long> o(2 * K);
          std::vector<int> x(K);
          hipMemcpy(o.data(), out, sizeof(unsigned long long) * 2 * K, hipMemcpyDeviceToHost);
          hipMemcpy(x.data(), xcc, sizeof(int) * K, hipMemcpyDeviceToHost);
          bool same = true;
          for (int i = 1; i < K; i++) same = same && x[i] == x[0];
          const double us_it = wall_us / iters;
          printf("%s K %2d rows/wg %d spins %5d+%5d cycles: %.2f us per iteration (host wall), %.0f device cycles per iteration, "
                 "exchange %.2f us; same XCD: %s (xcc %d)\n",
                 load ? "LOADED" : "idle  ", K, rows_per_wg, se, ss, us_it, (double)o[0] / iters,
                 us_it - (se + ss) / 2400.0, same ? "yes" : "NO", x[0]);
        }
      }
    }
  }
  return 0;
}
