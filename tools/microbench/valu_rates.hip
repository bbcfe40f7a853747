// valu_rates.hip — issue cost of the VALU instructions the dense evaluation kernel is made of, measured the way that
// kernel runs them: W waves per SIMD (W = 1, 2, 4, 8), independent chains, every CU busy.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rates tools/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

enum { FMA32, PKFMA32, FMA64, CVT64, CVT32, RCP32, RCP64, MUL32, ADD64, MIX, PKMUL32, PKADD32, ADD32 };
constexpr int kIters = 2048, kChains = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
  float a[kChains]; double d[kChains]; float2 p[kChains];
  for (int i = 0; i < kChains; i++) { a[i] = seed + i + threadIdx.x * 1e-3f; d[i] = a[i]; p[i] = make_float2(a[i], a[i] + 1); }
  const float m = 1.0000001f, c = 1e-7f; const double dm = 1.0000000001, dc = 1e-9;
  const float2 m2 = make_float2(m, m), c2 = make_float2(c, c);
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < kChains; i++) {
      if (OP == FMA32) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); }
      if (OP == MUL32) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
      if (OP == PKFMA32) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2)); }
      if (OP == PKMUL32) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2)); }
      if (OP == PKADD32) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2)); }
      if (OP == ADD32) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
      if (OP == FMA64) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc)); }
      if (OP == ADD64) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc)); }
      if (OP == CVT64) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i])); }
      if (OP == CVT32) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i])); }
      if (OP == RCP32) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])); }
      if (OP == RCP64) { asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i])); }
      if (OP == MIX) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc)); }
    }
  }
  float s = 0; for (int i = 0; i < kChains; i++) s += a[i] + (float)d[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
static void run(const char* name, int per_iter_instr) {
  float* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256-thread block = one wave per SIMD of a CU
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / 5;
    const double instr_per_simd = (double)wps * kIters * kChains * per_iter_instr;
    printf("%-12s %d waves/SIMD: %8.1f us -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, us, us * 2400.0 / instr_per_simd);
  }
}
int main() {
  run<FMA32>("v_fma_f32", 1);
  run<MUL32>("v_mul_f32", 1);
  run<ADD32>("v_add_f32", 1);
  run<PKFMA32>("v_pk_fma_f32", 1);
  run<PKMUL32>("v_pk_mul_f32", 1);
  run<PKADD32>("v_pk_add_f32", 1);
  if (getenv("VALU_RATES_F32_ONLY")) return 0;
  run<FMA64>("v_fma_f64", 1);
  run<ADD64>("v_add_f64", 1);
  run<CVT64>("cvt_f64_f32", 1);
  run<CVT32>("cvt_f32_f64", 1);
  run<RCP32>("v_rcp_f32", 1);
  run<RCP64>("v_rcp_f64", 1);
  run<MIX>("fma32+fma64", 2);
  return 0;
}
