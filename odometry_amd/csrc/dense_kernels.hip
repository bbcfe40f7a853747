// dense_kernels.hip — the dense-level evaluation kernels (dense.hip.h) as a translation unit of their own (compiles beside the
// main one; see dense.hip.h), plus their host-side launcher.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#define ODO_DENSE_KERNELS 1
#include "dense.hip.h"

namespace odo {

void launch_dense_eval(const DenseLevel& L, const LmState* st, int expect_level, int robust, float huber_delta,
                       const float* scale_sqr_ptr, double* partials, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int plain_div) {
  const dim3 grid(L.nblk), block(kDenseBlock);
  if (plain_div) {
    if (e0 && e1)
      hipExtLaunchKernelGGL((lm_dense_eval_kernel<kDenseBlock, 1, kDenseWaves>), grid, block, 0, s, e0, e1, 0, L, st, expect_level, robust,
                            huber_delta, scale_sqr_ptr, partials);
    else
      hipLaunchKernelGGL((lm_dense_eval_kernel<kDenseBlock, 1, kDenseWaves>), grid, block, 0, s, L, st, expect_level, robust, huber_delta,
                         scale_sqr_ptr, partials);
  } else {
    if (e0 && e1)
      hipExtLaunchKernelGGL((lm_dense_eval_kernel<kDenseBlock, 0, kDenseWaves>), grid, block, 0, s, e0, e1, 0, L, st, expect_level, robust,
                            huber_delta, scale_sqr_ptr, partials);
    else
      hipLaunchKernelGGL((lm_dense_eval_kernel<kDenseBlock, 0, kDenseWaves>), grid, block, 0, s, L, st, expect_level, robust, huber_delta,
                         scale_sqr_ptr, partials);
  }
}

void launch_dense_eval_batch(const DenseBatchItem* d_items, int n, int max_nblk, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int plain_div) {
  const dim3 grid((max_nblk + 7) / 8 * 8, n), block(kDenseBlock);
  if (plain_div) {
    if (e0 && e1) hipExtLaunchKernelGGL((lm_dense_eval_batch_kernel<kDenseBlock, 1, kDenseWaves>), grid, block, 0, s, e0, e1, 0, d_items);
    else hipLaunchKernelGGL((lm_dense_eval_batch_kernel<kDenseBlock, 1, kDenseWaves>), grid, block, 0, s, d_items);
  } else {
    if (e0 && e1) hipExtLaunchKernelGGL((lm_dense_eval_batch_kernel<kDenseBlock, 0, kDenseWaves>), grid, block, 0, s, e0, e1, 0, d_items);
    else hipLaunchKernelGGL((lm_dense_eval_batch_kernel<kDenseBlock, 0, kDenseWaves>), grid, block, 0, s, d_items);
  }
}

}  // namespace odo
