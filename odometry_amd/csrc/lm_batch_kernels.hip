// lm_batch_kernels.hip — the BATCHED pose-LM kernels (lm_step_kernel_batch, lm_coarse_kernel_batch, lm_fine_kernel_batch: S sequences
// per launch, odo_tracker_batch / odo_lm_solve_batch) as a translation unit of their own, compiled with the occupancy-first machine
// scheduler (odometry_amd/build.py), plus their host-side launchers. See the top of kernels.hip.h: the main unit's ILP-first
// scheduler serves the single tracker's one-wave latency chain and costs these throughput kernels 10-14 % at S = 1 ... 4.
#include <hip/hip_runtime.h>
#define ODO_LM_BATCH_TU 1
// every other kernel of the header becomes a function template nobody instantiates: declared, never emitted (a plain `static
// __global__` is emitted whether launched or not: the unit would carry a second copy of every kernel of the library)
#define ODO_KERNEL template <int kNotInThisUnit = 0> static __global__
#define ODO_KERNEL_T static __global__
#include "kernels.hip.h"

namespace odo {

void launch_lm_step_batch(int grid_x, int n, hipStream_t s, const StepArgs* table, int seq, int first_of_solve, unsigned long long* span) {
  hipLaunchKernelGGL(lm_step_kernel_batch, dim3(grid_x, n), dim3(kLmBlock), 0, s, table, seq, first_of_solve, span);
}

void launch_lm_coarse_batch(bool lean, int n, hipStream_t s, const StepArgs* table, int seq, int first_of_solve, unsigned long long* span) {
  if (lean) hipLaunchKernelGGL(lm_coarse_kernel_batch, dim3(1, n), dim3(kCoarseBlock), kCoarseLdsBytes, s, table, seq, first_of_solve, span);
  else hipLaunchKernelGGL(lm_coarse_full_kernel_batch, dim3(1, n), dim3(kCoarseBlock), kCoarseLdsBytes, s, table, seq, first_of_solve, span);
}

void launch_lm_fine_batch(bool lean, int blocks, hipStream_t s, const StepArgs* table, int n, int K, int seq, int first_of_solve,
                          unsigned long long* span, int fault, const XccIds& xcc, unsigned* dispatch_words) {
  if (lean)
    hipLaunchKernelGGL(lm_fine_kernel_batch, dim3(blocks), dim3(kFineThreads), 0, s, table, n, K, seq, first_of_solve, span, fault, xcc,
                       dispatch_words);
  else
    hipLaunchKernelGGL(lm_fine_trace_kernel_batch, dim3(blocks), dim3(kFineThreads), 0, s, table, n, K, seq, first_of_solve, span, fault,
                       xcc, dispatch_words);
}

}  // namespace odo
