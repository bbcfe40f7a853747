// exec_mask_rates.hip — does a wave64 VALU instruction get cheaper when only 16 (or 32) of its lanes are enabled? One wave per SIMD,
// one block per CU, a single dependent chain and eight independent ones, EXEC = all 64 / the low 32 / the low 16 / one lane.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/exec_mask_rates tools/microbench/exec_mask_rates.hip && /tmp/exec_mask_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kIters = 4096;

template <int CHAINS, int OP>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, float seed, int lanes) {
  float a[CHAINS]; double d[CHAINS];
  for (int i = 0; i < CHAINS; i++) { a[i] = seed + i + threadIdx.x * 1e-3f; d[i] = a[i]; }
  const float m = 1.0000001f, c = 1e-7f; const double dm = 1.0000000001, dc = 1e-9;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)(threadIdx.x & 63) < lanes) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
      for (int i = 0; i < CHAINS; i++) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc));
        if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = 0; for (int i = 0; i < CHAINS; i++) s += a[i] + (float)d[i];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, int OP>
static void run(const char* name) {
  float* out; unsigned long long* cyc; CK(hipMalloc(&out, 64)); CK(hipHostMalloc(&cyc, 64));
  for (int lanes : {64, 32, 16, 1}) {
    for (int r = 0; r < 3; r++) { hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256), dim3(256), 0, 0, out, cyc, 1.0f, lanes); CK(hipDeviceSynchronize()); }
    printf("%-10s %d chain(s), EXEC = %2d lanes: %.2f cycles per instruction\n", name, CHAINS, lanes, (double)cyc[0] / ((double)kIters * CHAINS));
  }
}
int main() {
  run<1, 0>("v_fma_f32"); run<8, 0>("v_fma_f32");
  run<1, 2>("v_mul_f32"); run<8, 2>("v_mul_f32");
  run<1, 1>("v_fma_f64"); run<8, 1>("v_fma_f64");
  return 0;
}
