// Device-wide exclusive scan of int32 arrays (shared by K2's plan and K6): three launches, no atomics,
// deterministic.  1024 elements per workgroup: per-block sums -> one-workgroup scan of the sums -> per-block
// scan with its base.  `in` and `out` may alias.
#ifndef KRS_SCAN_H_
#define KRS_SCAN_H_

#include "krs_common.h"

namespace krs {
namespace scan {

constexpr int kTile = 1024;   // 256 threads x 4 consecutive elements

// in-place exclusive scan of a (short) int32 array by ONE workgroup of 1024 threads; total -> *total (optional)
static __global__ __launch_bounds__(1024) void block_kernel(int32_t* a, int64_t n, int64_t* total) {
  __shared__ long long wsum[16];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = i < n ? a[i] : 0;
    long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    long long off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (i < n) a[i] = (int32_t)(off + x - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + x;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

static __global__ __launch_bounds__(256) void sums_kernel(const int32_t* a, int64_t n, int32_t* sums) {
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * kTile + threadIdx.x * 4;
  int c = 0;
  for (int k = 0; k < 4; ++k)
    if (i0 + k < n) c += a[i0 + k];
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

static __global__ __launch_bounds__(256) void apply_kernel(const int32_t* in, int32_t* out, int64_t n,
                                                           const int32_t* sums) {
  __shared__ int wsum[4];
  const int64_t i0 = (int64_t)blockIdx.x * kTile + threadIdx.x * 4;
  int v[4], c = 0;
  for (int k = 0; k < 4; ++k) {
    v[k] = i0 + k < n ? in[i0 + k] : 0;
    c += v[k];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int s = sums[blockIdx.x] + x - c;
  for (int w = 0; w < wave; ++w) s += wsum[w];
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) out[i0 + k] = s;
    s += v[k];
  }
}

inline size_t workspace_bytes(int64_t n) { return (size_t)(ceil_div(n > 0 ? n : 1, kTile) + 1) * sizeof(int32_t) + 256; }

// out[i] = sum of in[0..i); `sums` = workspace of workspace_bytes(n); total (optional device int64) = sum of all
inline void exclusive(const int32_t* in, int32_t* out, int64_t n, int32_t* sums, int64_t* total, hipStream_t st) {
  if (n <= 0) return;
  const int blocks = (int)ceil_div(n, kTile);
  hipLaunchKernelGGL(sums_kernel, dim3(blocks), dim3(256), 0, st, in, n, sums);
  hipLaunchKernelGGL(block_kernel, dim3(1), dim3(1024), 0, st, sums, (int64_t)blocks, total);
  hipLaunchKernelGGL(apply_kernel, dim3(blocks), dim3(256), 0, st, in, out, n, sums);
}

}  // namespace scan
}  // namespace krs
#endif
