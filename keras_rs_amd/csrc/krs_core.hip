// Library-wide entry points of libkrs_hip.so.
#include "krs_common.h"

namespace krs {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace krs

extern "C" int krs_version(void) { return KRS_VERSION; }
extern "C" const char* krs_last_error(void) { return krs::error_buffer(); }

static_assert(sizeof(krs_table) == 32, "krs_table layout (mirrored in keras_rs_amd/_lib.py)");
static_assert(sizeof(krs_feature) == 24, "krs_feature layout");
static_assert(sizeof(krs_gemm_epilogue) == 80, "krs_gemm_epilogue layout");

// ---- krs_store_f32: a few host floats into device memory as kernel arguments (see include/krs.h) ----------------------
namespace krs {
namespace {
constexpr int kStoreChunk = 32;
struct StoreVals {
  float v[kStoreChunk];
};
__global__ void store_f32_kernel(char* dst, int64_t stride, StoreVals vals, int count) {
  const int i = threadIdx.x;
  if (i < count) *reinterpret_cast<float*>(dst + (int64_t)i * stride) = vals.v[i];
}
}  // namespace
}  // namespace krs

extern "C" int krs_store_f32(void* dst, int64_t stride_bytes, const float* values_host, int count, void* stream) {
  using namespace krs;
  KRS_REQUIRE(count >= 0 && stride_bytes >= 4 && stride_bytes % 4 == 0, "krs_store_f32: bad count / stride");
  if (count == 0) return KRS_OK;
  KRS_REQUIRE(dst && values_host, "krs_store_f32: null pointer");
  KRS_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 3) == 0, "krs_store_f32: dst must be 4-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < count; base += kStoreChunk) {
    const int n = count - base < kStoreChunk ? count - base : kStoreChunk;
    StoreVals vals;
    for (int i = 0; i < kStoreChunk; ++i) vals.v[i] = i < n ? values_host[base + i] : 0.0f;
    hipLaunchKernelGGL(store_f32_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<char*>(dst) + (int64_t)base * stride_bytes,
                       stride_bytes, vals, n);
    KRS_CHECK_LAUNCH("store_f32_kernel");
  }
  return KRS_OK;
}
