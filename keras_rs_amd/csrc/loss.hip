// K7 -- the loss at the end of the DLRM step: binary cross-entropy on the sigmoid output of the top MLP.
//
// Reference: model.compile(loss=keras.losses.BinaryCrossentropy()) (examples/ml_perf/main.py:201-210) on the output of
// the top MLP, whose last Dense has a sigmoid activation (examples/ml_perf/model.py:105-163).  Keras computes, with
// from_logits=False and the default reduction (mean over the batch of the mean over the last axis):
//     p = clip(pred, eps, 1 - eps), eps = keras.backend.epsilon() = 1e-7
//     loss = mean( -(y log p + (1 - y) log(1 - p)) )
// Forward and backward are ONE pass here (the prediction is read once): the loss scalar and
//     dL/dpred_i = scale / n * ((1 - y_i) / (1 - p_i) - y_i / p_i)    (0 where the clip is active)
// Per-thread sums over a fixed stride, a fixed-order tree in LDS per workgroup, and -- with the caller's `partials`
// scratch (KRS_BCE_MAX_BLOCKS floats) -- one partial per workgroup added in workgroup order by a second tiny launch:
// the loss is run-to-run bit-identical (no atomics).  Without scratch ONE workgroup walks all n predictions.
#include <algorithm>

#include "krs_common.h"

namespace krs {
namespace {

constexpr int kBceThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kBceThreads) void bce_kernel(const T* pred, const float* labels, int64_t n, float eps,
                                                          float scale, float* loss, T* dpred, float* partials) {
  __shared__ float part[kBceThreads];
  const float hi = 1.0f - eps, inv_n = 1.0f / (float)n;
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * kBceThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBceThreads) {
    float x;
    if constexpr (sizeof(T) == 2) x = bf16_to_f32(pred[i]);
    else x = pred[i];
    const float y = labels[i];
    const float p = fminf(fmaxf(x, eps), hi);
    acc -= y * logf(p) + (1.0f - y) * logf(1.0f - p);
    if (dpred) {
      const bool inside = x >= eps && x <= hi;
      const float g = inside ? scale * inv_n * ((1.0f - y) / (1.0f - p) - y / p) : 0.0f;
      if constexpr (sizeof(T) == 2) dpred[i] = f32_to_bf16(g);
      else dpred[i] = g;
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kBceThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (partials) partials[blockIdx.x] = part[0];
    else *loss = part[0] * inv_n;
  }
}

__global__ void bce_finish_kernel(const float* partials, int n_blocks, int64_t n, float* loss) {
  float s = 0.0f;
  for (int b = 0; b < n_blocks; ++b) s += partials[b];     // workgroup order
  *loss = s / (float)n;
}

}  // namespace
}  // namespace krs

extern "C" int krs_bce_fwd_bwd(const void* pred, int pred_dtype, const float* labels, int64_t n, float epsilon,
                               float grad_scale, float* loss, void* dpred, float* partials, void* stream) {
  using namespace krs;
  KRS_REQUIRE(pred && labels && loss, "bce: null argument");
  KRS_REQUIRE(n > 0, "bce: empty batch");
  KRS_REQUIRE(pred_dtype == KRS_F32 || pred_dtype == KRS_BF16, "bce: bad dtype");
  KRS_REQUIRE(epsilon > 0.0f && epsilon < 0.5f, "bce: epsilon must be in (0, 0.5)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int blocks = partials ? (int)std::min<int64_t>(KRS_BCE_MAX_BLOCKS, ceil_div(n, kBceThreads)) : 1;
  if (blocks == 1) partials = nullptr;
  if (pred_dtype == KRS_BF16)
    hipLaunchKernelGGL(bce_kernel<uint16_t>, dim3(blocks), dim3(kBceThreads), 0, st, reinterpret_cast<const uint16_t*>(pred),
                       labels, n, epsilon, grad_scale, loss, reinterpret_cast<uint16_t*>(dpred), partials);
  else
    hipLaunchKernelGGL(bce_kernel<float>, dim3(blocks), dim3(kBceThreads), 0, st, reinterpret_cast<const float*>(pred), labels,
                       n, epsilon, grad_scale, loss, reinterpret_cast<float*>(dpred), partials);
  if (partials) hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(1), 0, st, partials, blocks, n, loss);
  KRS_CHECK_LAUNCH("krs_bce_fwd_bwd");
  return KRS_OK;
}
