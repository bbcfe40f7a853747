// Does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a launch to ONE XCD of an MI355X, and which mask bits are that XCD's?
// A persistent kernel whose workgroups must share an XCD is launched today as 8 x K blocks of which every eighth takes part (the
// dispatcher deals blocks round-robin over the XCDs); the other seven eighths return at once but must still be placed — on XCDs that
// another persistent kernel may have filled. With a stream confined to one XCD the grid could be the K
// workgroups themselves.   hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void where(int* out) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = (int)(x & 0xf);
  }
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  printf("%s: %d CUs\n", p.name, ncu);
  int* d; hipMalloc(&d, 4096 * sizeof(int));
  std::vector<int> h(4096);
  for (int pattern = 0; pattern < 3; pattern++) {
    for (int x = 0; x < 8; x += (pattern == 2 ? 8 : 1)) {
      std::vector<uint32_t> mask(words, 0u);
      for (int i = 0; i < ncu; i++) {
        const bool on = pattern == 0 ? (i % 8 == x) : pattern == 1 ? (i / (ncu / 8) == x) : true;
        if (on) mask[i / 32] |= 1u << (i % 32);
      }
      hipStream_t s;
      if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
      hipMemsetAsync(d, 0xff, 4096 * sizeof(int), s);
      hipLaunchKernelGGL(where, dim3(256), dim3(512), 0, s, d);
      hipMemcpyAsync(h.data(), d, 256 * sizeof(int), hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
      int hist[16] = {0};
      for (int b = 0; b < 256; b++) hist[h[b] & 15]++;
      printf("%s x = %d: blocks per XCC id:", pattern == 0 ? "bits i %% 8 == x" : pattern == 1 ? "bits i / 32 == x" : "all bits", x);
      for (int k = 0; k < 8; k++) printf(" %d", hist[k]);
      printf("\n");
      hipStreamDestroy(s);
    }
  }
  return 0;
}
