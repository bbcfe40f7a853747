// What does hipLaunchKernelGGL cost the HOST for a kernel whose by-value argument is 64 B ... 1.5 KB (the pose LM's StepArgs is ~1.3 KB)?
// Host time of the call itself (steady_clock around 2 000 launches on a stream that is kept nearly empty: a sync every 16 launches).
//   hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { unsigned w[N]; };
template <int N> __global__ void k(Blob<N> b, unsigned* out) { if (threadIdx.x == 0 && b.w[0] == 0xdeadbeefu) out[0] = b.w[N - 1]; }
template <int N> static void run(hipStream_t s, unsigned* d) {
  Blob<N> b{}; b.w[0] = 1;
  for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, b, d);
  (void)hipStreamSynchronize(s);
  double sum = 0; int n = 0;
  for (int rep = 0; rep < 125; rep++) {
    for (int i = 0; i < 16; i++) {
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, b, d);
      sum += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); n++;
    }
    (void)hipStreamSynchronize(s);
  }
  // launch -> completion latency of a single launch on an idle stream
  double lat = 0;
  for (int rep = 0; rep < 200; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, b, d);
    (void)hipStreamSynchronize(s);
    lat += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("argument %5d B: launch call %.2f us, launch + sync on an idle stream %.2f us\n", (int)sizeof(Blob<N>), sum / n, lat / 200);
}
int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned* d; (void)hipMalloc(&d, 64);
  run<16>(s, d); run<64>(s, d); run<128>(s, d); run<256>(s, d); run<336>(s, d); run<384>(s, d);
  int lo, hi; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t sp; (void)hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi);
  printf("high-priority stream:\n");
  run<16>(sp, d); run<336>(sp, d);
  return 0;
}
