#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_trivial(int* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += n; }
__global__ void k_spin(int* p, int cycles) {
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
__global__ void k_hostflag(int* p, int* host, int seq, int cycles) {
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { p[0] += 1; __hip_atomic_store(host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  int* h; hipHostMalloc(&h, 64, hipHostMallocMapped); int* hd; hipHostGetDevicePointer((void**)&hd, h, 0);
  const int N = 2000;
  for (int grid : {1, 161}) {
    for (int cyc : {0, 5000, 15000}) {
      for (int rep = 0; rep < 2; rep++) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, s, d, cyc);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        if (rep) printf("grid %3d spin %5d cycles (%.2f us): %.2f us per launch -> boundary+launch %.2f us\n", grid, cyc, cyc / 2400.0, us, us - cyc / 2400.0);
      }
    }
  }
  for (int cyc : {5000, 15000}) {
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_hostflag, dim3(161), dim3(256), 0, s, d, hd, i, cyc);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("hostflag grid 161 spin %d: %.2f us per launch -> overhead %.2f\n", cyc, us, us - cyc / 2400.0);
  }
  // just-in-time issue: wait for the flag of launch i-1 before issuing launch i+1 (run-ahead 2)
  for (int cyc : {15000}) {
    volatile int* hv = h; h[0] = -1;
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      while (i - hv[0] > 2) {}
      hipLaunchKernelGGL(k_hostflag, dim3(161), dim3(256), 0, s, d, hd, i, cyc);
    }
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("run-ahead 2, grid 161 spin %d: %.2f us per launch -> overhead %.2f\n", cyc, us, us - cyc / 2400.0);
  }
  return 0;
}
