// What does ONE all-to-all exchange of a single fp64 per wave cost between the workgroups of a persistent launch on one XCD?
// (The t-distribution scale passes of lm_fine_tdist_kernel and the iterations of the persistent depth LM are chains of exactly this.)
// Every iteration each wave publishes one tagged 16-byte granule pair; then, per variant,
//   mode 0: wave 0 of each workgroup gathers all pairs (lane l: pairs l, l + 64, ...) and hands the sum to the other waves through LDS,
//   mode 1: every wave gathers for itself.
// Knobs: plain vs agent-scope stores, a delay (s_sleep units) before the first poll, s_sleep between polls.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/xse tools/microbench/xcd_scalar_exchange.hip && /tmp/xse
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef unsigned G2 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) G2 G2Global;

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int N> __device__ __forceinline__ void nap() { if (N > 0) __builtin_amdgcn_s_sleep(N); }

struct Slot { double v; unsigned tag; unsigned pad; };

template <int kThreads, int kMode, int kPlain, int kPre, int kBetween>
__global__ void __launch_bounds__(kThreads) coop(int K, int iters, unsigned long long* buf, unsigned long long* out, int* xcc) {
  if (blockIdx.x % 8 != 0) return;
  constexpr int W = kThreads / 64;
  const int g = blockIdx.x / 8, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __shared__ Slot slot[2];
  if (t == 0) { xcc[g] = xcc_id(); slot[0].tag = slot[1].tag = 0; }
  __syncthreads();
  const int nchunk = K * W, chunk = g * W + wv;
  double acc = 0.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 1; it <= iters; it++) {
    const unsigned tag = (unsigned)it;
    unsigned long long* b = buf + (size_t)(it & 1) * 2 * 256;
    const double mine = (double)(chunk + it) + acc * 1e-9;
    if (lane == 0) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
      const unsigned long long g0 = ((bits >> 32) << 32) | tag, g1 = (bits << 32) | tag;
      if (kPlain) { b[2 * chunk] = g0; b[2 * chunk + 1] = g1; }
      else {
        __hip_atomic_store(b + 2 * chunk, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(b + 2 * chunk + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    double total = 0.0;
    if (kMode == 1 || wv == 0) {
      nap<kPre>();
      G2 q[4];
      bool ok = false;
      for (int spin = 0; !ok; spin++) {
        if (spin > 0) nap<kBetween>();
        bool m = true;
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = lane + 64 * u; if (c < nchunk) q[u] = *(const volatile G2Global*)(b + 2 * c); }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = lane + 64 * u; if (c < nchunk) m = m && q[u].x == tag && q[u].z == tag; }
        ok = __all(m);
      }
      double part = 0.0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = lane + 64 * u;
        if (c < nchunk) part += __longlong_as_double((long long)(((unsigned long long)q[u].y << 32) | q[u].w));
      }
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      total = part;
      if (kMode == 0 && lane == 0) {
        slot[it & 1].v = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&slot[it & 1].tag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      __builtin_amdgcn_s_setprio(0);
      while (__hip_atomic_load(&slot[it & 1].tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != tag) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_s_setprio(3);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      total = slot[it & 1].v;
    }
    acc += total;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (t == 0) out[2 * g] = t1 - t0;
  if (lane == 0) out[2 * g + 1] = (unsigned long long)acc;
}

template <int kThreads, int kMode, int kPlain, int kPre, int kBetween>
void run(const char* what, hipStream_t s, unsigned long long* buf, unsigned long long* out, int* xcc) {
  const int iters = 4000;
  for (int K : {1, 2, 4, 8, 16, 32}) {
    if (K * (kThreads / 64) > 256) continue;
    hipMemsetAsync(buf, 0, sizeof(unsigned long long) * 4 * 256, s);
    hipStreamSynchronize(s);
    hipLaunchKernelGGL((coop<kThreads, kMode, kPlain, kPre, kBetween>), dim3(8 * K), dim3(kThreads), 0, s, K, iters, buf, out, xcc);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> o(2 * K);
    std::vector<int> x(K);
    hipMemcpy(o.data(), out, sizeof(unsigned long long) * 2 * K, hipMemcpyDeviceToHost);
    hipMemcpy(x.data(), xcc, sizeof(int) * K, hipMemcpyDeviceToHost);
    bool same = true;
    for (int i = 1; i < K; i++) same = same && x[i] == x[0];
    printf("%-58s K %2d waves %3d: %6.0f cycles per exchange%s\n", what, K, K * (kThreads / 64), (double)o[0] / iters, same ? "" : "  (NOT one XCD)");
  }
}

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long *buf, *out;
  int* xcc;
  hipMalloc(&buf, sizeof(unsigned long long) * 4 * 256);
  hipMalloc(&out, sizeof(unsigned long long) * 2 * 64);
  hipMalloc(&xcc, sizeof(int) * 64);
  run<512, 0, 1, 0, 0>("512 thr, wave 0 gathers + LDS, plain, no delay", s, buf, out, xcc);
  run<512, 0, 1, 0, 1>("512 thr, wave 0 gathers + LDS, plain, sleep 1 between", s, buf, out, xcc);
  run<512, 0, 1, 2, 0>("512 thr, wave 0 gathers + LDS, plain, pre-delay 2", s, buf, out, xcc);
  run<512, 0, 1, 4, 0>("512 thr, wave 0 gathers + LDS, plain, pre-delay 4", s, buf, out, xcc);
  run<512, 0, 1, 8, 0>("512 thr, wave 0 gathers + LDS, plain, pre-delay 8", s, buf, out, xcc);
  run<512, 0, 0, 0, 0>("512 thr, wave 0 gathers + LDS, agent-scope stores", s, buf, out, xcc);
  run<512, 1, 1, 0, 0>("512 thr, every wave gathers, plain, no delay", s, buf, out, xcc);
  run<512, 1, 1, 4, 0>("512 thr, every wave gathers, plain, pre-delay 4", s, buf, out, xcc);
  run<256, 1, 1, 0, 0>("256 thr, every wave gathers, plain, no delay", s, buf, out, xcc);
  run<256, 0, 1, 0, 0>("256 thr, wave 0 gathers + LDS, plain, no delay", s, buf, out, xcc);
  run<64, 1, 1, 0, 0>(" 64 thr, every wave gathers, plain, no delay", s, buf, out, xcc);
  run<64, 1, 1, 4, 0>(" 64 thr, every wave gathers, plain, pre-delay 4", s, buf, out, xcc);
  return 0;
}
