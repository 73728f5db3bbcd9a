// K7 -- the loss at the end of the DLRM step: binary cross-entropy on the sigmoid output of the top MLP.
//
// Reference: model.compile(loss=keras.losses.BinaryCrossentropy()) (examples/ml_perf/main.py:201-210) on the output of
// the top MLP, whose last Dense has a sigmoid activation (examples/ml_perf/model.py:105-163).  Keras computes, with
// from_logits=False and the default reduction (mean over the batch of the mean over the last axis):
//     p = clip(pred, eps, 1 - eps), eps = keras.backend.epsilon() = 1e-7
//     loss = mean( -(y log p + (1 - y) log(1 - p)) )
// Forward and backward are ONE pass here (the prediction is read once): the loss scalar and
//     dL/dpred_i = scale / n * ((1 - y_i) / (1 - p_i) - y_i / p_i)    (0 where the clip is active)
// One workgroup walks the n predictions (n = the batch: 65,536 at C3, 128 KB of input): per-thread sums over a fixed
// stride, then a fixed-order tree in LDS, so the loss is run-to-run bit-identical (no atomics).
#include "krs_common.h"

namespace krs {
namespace {

constexpr int kBceThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kBceThreads) void bce_kernel(const T* pred, const float* labels, int64_t n, float eps,
                                                          float scale, float* loss, T* dpred) {
  __shared__ float part[kBceThreads];
  const float hi = 1.0f - eps, inv_n = 1.0f / (float)n;
  float acc = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += kBceThreads) {
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
  if (threadIdx.x == 0) *loss = part[0] * inv_n;
}

}  // namespace
}  // namespace krs

extern "C" int krs_bce_fwd_bwd(const void* pred, int pred_dtype, const float* labels, int64_t n, float epsilon,
                               float grad_scale, float* loss, void* dpred, void* stream) {
  using namespace krs;
  KRS_REQUIRE(pred && labels && loss, "bce: null argument");
  KRS_REQUIRE(n > 0, "bce: empty batch");
  KRS_REQUIRE(pred_dtype == KRS_F32 || pred_dtype == KRS_BF16, "bce: bad dtype");
  KRS_REQUIRE(epsilon > 0.0f && epsilon < 0.5f, "bce: epsilon must be in (0, 0.5)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pred_dtype == KRS_BF16)
    hipLaunchKernelGGL(bce_kernel<uint16_t>, dim3(1), dim3(kBceThreads), 0, st, reinterpret_cast<const uint16_t*>(pred),
                       labels, n, epsilon, grad_scale, loss, reinterpret_cast<uint16_t*>(dpred));
  else
    hipLaunchKernelGGL(bce_kernel<float>, dim3(1), dim3(kBceThreads), 0, st, reinterpret_cast<const float*>(pred), labels,
                       n, epsilon, grad_scale, loss, reinterpret_cast<float*>(dpred));
  KRS_CHECK_LAUNCH("krs_bce_fwd_bwd");
  return KRS_OK;
}
