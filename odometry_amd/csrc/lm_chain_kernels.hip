// lm_chain_kernels.hip — the single tracker's pose-LM chain (lm_coarse_kernel, lm_fine_kernel and their variants) as a translation
// unit of its own, compiled with the ILP-first machine scheduler (odometry_amd/build.py), plus the host-side launchers. See the top
// of kernels.hip.h: these kernels are one wave working through ~1 200 dependent instructions per evaluation; every other kernel of
// the library is a throughput kernel and compiles in the main unit under the occupancy-first scheduler.
#include <hip/hip_runtime.h>
#define ODO_LM_CHAIN_TU 1
// every other kernel of the header becomes a function template nobody instantiates: declared, never emitted (a plain `static
// __global__` is emitted whether launched or not: the unit would carry a second copy of every kernel of the library)
#define ODO_KERNEL template <int kNotInThisUnit = 0> static __global__
#define ODO_KERNEL_T static __global__
#include "kernels.hip.h"

namespace odo {

hipError_t lm_chain_setup() {
  // the single-workgroup coarse kernel reduces through its dynamic LDS block (gfx950: up to 160 KB per workgroup)
  hipError_t e = hipFuncSetAttribute((const void*)lm_coarse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseLdsBytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)lm_coarse_armed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseLdsBytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)lm_coarse_full_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseLdsBytes);
}

void launch_lm_coarse_armed(hipStream_t s, const StepArgs& a, int min_level) {
  hipLaunchKernelGGL(lm_coarse_armed_kernel, dim3(1), dim3(kCoarseBlock), kCoarseLdsBytes, s, a, min_level);
}

void launch_lm_coarse(bool lean, hipStream_t s, const StepArgs& a, int min_level) {
  if (lean) hipLaunchKernelGGL(lm_coarse_kernel, dim3(1), dim3(kCoarseBlock), kCoarseLdsBytes, s, a, min_level);
  else hipLaunchKernelGGL(lm_coarse_full_kernel, dim3(1), dim3(kCoarseBlock), kCoarseLdsBytes, s, a, min_level);
}

void launch_lm_fine(int variant, int blocks, hipStream_t s, const StepArgs& a, int K, unsigned long long* xbuf, int fault, int lo_level,
                    unsigned* dispatch_words) {
  if (variant == 2)
    hipLaunchKernelGGL(lm_fine_tdist_kernel, dim3(blocks), dim3(kFineThreads), 0, s, a, K, xbuf, fault, lo_level, dispatch_words);
  else if (variant == 1)
    hipLaunchKernelGGL(lm_fine_trace_kernel, dim3(blocks), dim3(kFineThreads), 0, s, a, K, xbuf, fault, lo_level, dispatch_words);
  else
    hipLaunchKernelGGL(lm_fine_kernel, dim3(blocks), dim3(kFineThreads), 0, s, a, K, xbuf, fault, lo_level, dispatch_words);
}

#if ODO_PHASE_STAMPS
void lm_chain_diag_read(unsigned long long out[24]) {   // diagnostic build: the wall-clock sums the chain kernels keep in device memory
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lm_diag), sizeof(unsigned long long) * 24) != hipSuccess) memset(out, 0, sizeof(unsigned long long) * 24);
}
#endif

}  // namespace odo
