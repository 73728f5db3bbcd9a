// Calibration kernels: streaming read, write, copy with 16 B/lane (development aid).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_copy(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) d[i] = s[i];
}
__global__ void k_copy_nt(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
__global__ void k_read(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  u32x4 a = {0, 0, 0, 0};
  for (; i < n; i += st) a ^= s[i];
  if (a.x == 0x12345678u && a.y == 77u) d[0] = a;
}
__global__ void k_write(u32x4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  u32x4 a = {1, 2, 3, 4};
  for (; i < n; i += st) d[i] = a;
}
extern "C" void membw(int which, const void* s, void* d, size_t bytes, int blocks, void* stream) {
  size_t n = bytes / 16;
  hipStream_t st = (hipStream_t)stream;
  if (which == 0) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (const u32x4*)s, (u32x4*)d, n);
  if (which == 1) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, st, (const u32x4*)s, (u32x4*)d, n);
  if (which == 2) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, st, (u32x4*)d, n);
  if (which == 3) hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, st, (const u32x4*)s, (u32x4*)d, n);
}
