// Shared device/host helpers of libkrs_hip.so (gfx950 only).
#ifndef KRS_COMMON_H_
#define KRS_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/krs.h"

namespace krs {

// ---- thread-local error text behind krs_last_error() ----------------------
char* error_buffer();
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define KRS_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return ::krs::fail(KRS_ERR_INVALID, __VA_ARGS__); \
  } while (0)

#define KRS_CHECK_LAUNCH(what)                                                           \
  do {                                                                                   \
    hipError_t e__ = hipGetLastError();                                                  \
    if (e__ != hipSuccess)                                                               \
      return ::krs::fail(KRS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e__));        \
  } while (0)

#define KRS_HIP(call)                                                                    \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess)                                                               \
      return ::krs::fail(KRS_ERR_LAUNCH, "%s: %s", #call, hipGetErrorString(e__));       \
  } while (0)

// ---- bf16 <-> f32, bit-identical to oracle/krs_oracle.c -------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// f32 -> bf16 is gfx950's v_cvt_pk_bf16_f32: round to nearest even, NaN quieted with its payload
// kept -- checked against the oracle's integer formula on all 2^32 inputs
// (scripts/exp/cvt_bf16_check.hip: zero mismatches, NaNs included).
typedef __bf16 krs_bf16x2 __attribute__((ext_vector_type(2)));
typedef float krs_f32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16x2 (lo = a, hi = b)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((krs_f32x2){a, b}, krs_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, f) & 0xffffu); }

// runtime-dtype element access (generic / fallback kernels only)
__device__ __forceinline__ float ld_elem(const void* p, int dtype, int64_t i) {
  return dtype == KRS_BF16 ? bf16_to_f32(((const uint16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st_elem(void* p, int dtype, int64_t i, float v) {
  if (dtype == KRS_BF16)
    ((uint16_t*)p)[i] = f32_to_bf16(v);
  else
    ((float*)p)[i] = v;
}
__device__ __forceinline__ int64_t ld_index(const void* p, int is64, int64_t i) {
  return is64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}

__host__ __device__ __forceinline__ int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace krs
#endif
