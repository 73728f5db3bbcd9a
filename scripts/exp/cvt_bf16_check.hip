// Exhaustive check: v_cvt_pk_bf16_f32 (gfx950) against the software RNE of krs_common.h / the oracle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ uint16_t sw(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__global__ void k(unsigned long long* bad, unsigned long long* bad_nan, uint32_t* first) {
  const uint64_t n = 1ull << 32;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float f = __uint_as_float((uint32_t)i);
    const bf16x2 r = __builtin_convertvector((f32x2){f, f}, bf16x2);
    const uint16_t hw = (uint16_t)(__builtin_bit_cast(uint32_t, r) & 0xffff);
    const uint16_t s = sw(f);
    if (hw != s) {
      const bool nan = (((uint32_t)i) & 0x7fffffffu) > 0x7f800000u;
      if (nan) atomicAdd(bad_nan, 1ull); else { if (atomicAdd(bad, 1ull) == 0) { first[0] = (uint32_t)i; first[1] = hw; first[2] = s; } }
    }
  }
}
int main() {
  unsigned long long *bad, *bad_nan; uint32_t* first;
  hipMalloc(&bad, 8); hipMalloc(&bad_nan, 8); hipMalloc(&first, 12);
  hipMemset(bad, 0, 8); hipMemset(bad_nan, 0, 8); hipMemset(first, 0, 12);
  k<<<4096, 256>>>(bad, bad_nan, first);
  unsigned long long h[2]; uint32_t f[3];
  hipMemcpy(&h[0], bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&h[1], bad_nan, 8, hipMemcpyDeviceToHost);
  hipMemcpy(f, first, 12, hipMemcpyDeviceToHost);
  printf("mismatch non-nan %llu nan %llu first %08x hw %04x sw %04x\n", h[0], h[1], f[0], f[1], f[2]);
  return 0;
}
