// Probe (north_star: "LDS staging of hot embedding rows"): does serving the hottest rows of a table from LDS
// beat leaving them to L2 / MALL?  One table [V, 128] bf16, bags of L lookups with power-law ids
// id = floor(V * u^e) -- the BEST case for staging: the hot rows are ids 0 .. K-1, so membership is `id < K`
// (no hash, no per-batch ranking).  Persistent workgroups (2 per CU); variant B copies rows 0 .. K-1 (K * 256 B)
// into LDS once per workgroup and reads hot lookups from there.  Same lane mapping as K1: 16 lanes per row,
// fp32 accumulation, 8 row loads in flight.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/exp/hotrow_probe.hip -o scripts/exp/hotrow_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int K>   // K = rows staged in LDS (0 = none)
__global__ __launch_bounds__(256) void pool(const uint16_t* table, const int32_t* ids, uint16_t* out, int n_bags, int L) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if (K > 0) {
    for (int i = threadIdx.x; i < K * 16; i += 256)
      reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(table)[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  for (int bag = wave * 4 + g; bag < n_bags; bag += n_waves * 4) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int32_t* bi = ids + (int64_t)bag * L;
    for (int l0 = 0; l0 < L; l0 += 8) {
      u32x4 raw[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int id = bi[l0 + k];
        if (K > 0 && id < K) raw[k] = reinterpret_cast<const u32x4*>(lds)[id * 16 + sub];
        else raw[k] = reinterpret_cast<const u32x4*>(table)[(int64_t)id * 16 + sub];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += __uint_as_float(raw[k][j] << 16);
          acc[2 * j + 1] += __uint_as_float(raw[k][j] & 0xffff0000u);
        }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__float_as_uint(acc[2 * j]) >> 16) | (__float_as_uint(acc[2 * j + 1]) & 0xffff0000u);
    reinterpret_cast<u32x4*>(out)[(int64_t)bag * 16 + sub] = o;
  }
}

template <int K>
float run(const uint16_t* t, const int32_t* ids, uint16_t* out, int n_bags, int L) {
  auto kern = pool<K>;
  const size_t lds = (size_t)K * 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(512), dim3(256), lds, 0, t, ids, out, n_bags, L);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(512), dim3(256), lds, 0, t, ids, out, n_bags, L);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 100.0f;
}

int main() {
  const int64_t V = 4000000; const int L = 32, n_bags = 1 << 19;      // 16.8 M lookups, 1 GB table
  uint16_t *t, *out; int32_t* ids;
  hipMalloc(&t, V * 256); hipMalloc(&out, (size_t)n_bags * 256); hipMalloc(&ids, (size_t)n_bags * L * 4);
  hipMemset(t, 0x3c, V * 256);
  std::vector<int32_t> h((size_t)n_bags * L);
  for (double e : {1.0, 4.0, 8.0}) {
    uint64_t x = 88172645463325252ULL; double hot256 = 0, hot64 = 0;
    for (auto& v : h) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const double u = (double)(x >> 11) / 9007199254740992.0;
      v = (int32_t)std::fmin((double)(V - 1), std::floor(V * std::pow(u, e)));
      hot256 += v < 256; hot64 += v < 64;
    }
    hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const double gb = (double)n_bags * L * 260 + (double)n_bags * 256;
    const float a = run<0>(t, ids, out, n_bags, L), b64 = run<64>(t, ids, out, n_bags, L), b256 = run<256>(t, ids, out, n_bags, L);
    printf("id = V*u^%.0f: lookups among the hottest 64 / 256 rows %.1f %% / %.1f %%;  no staging %.1f us (%.2f TB/s)   "
           "64 rows in LDS %.1f us   256 rows in LDS %.1f us\n", e, 100 * hot64 / h.size(), 100 * hot256 / h.size(), a,
           gb / a / 1e6, b64, b256);
  }
  return 0;
}
