// Round-4 probe (development aid, not product): what does the OPERAND LAYOUT do to the LDS-DMA stream of the
// 256x256 ring GEMM, and what does a 4-wave workgroup (128x128 per wave) do to its LDS-read + MFMA loop?
//   M = 65536, N = 512, K = 3456 bf16 (h = x U of the C3 cross layer); BK = 32 per ring stage, 4 stages.
//   MODE 0 full, 1 DMA only, 2 LDS reads + MFMA only.
//   LA / LB: layout of the A / B operand: 0 row-major [rows][K], 1 tiled [K/32][rows][32], 2 fm128 [K/128][rows][128]
//   PF: row-major A only -- every 4th block each thread touches one 128-byte line of the 256-byte row segments
//       PF blocks ahead (an L2 prefetch that asks HBM for whole 256-byte pieces at one time).
//   build: hipcc --offload-arch=gfx950 -O3 scripts/exp/gemm_probe2.hip -o scripts/exp/gemm_probe2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

constexpr int NSTAGE = 4, BKB = 64, OPA = 256 * BKB, STAGE = 2 * OPA;

template <int L>
__device__ __forceinline__ int64_t op_off(int64_t row, int64_t t, int64_t rows, int64_t ld) {
  if (L == 0) return row * ld * 2 + t * 64;
  if (L == 1) return t * rows * 64 + row * 64;
  return (t >> 2) * rows * 256 + row * 256 + (t & 3) * 64;
}

// WAVES = 8: 2(M) x 4(N) waves of 128x64; WAVES = 4: 2 x 2 waves of 128x128
template <int MODE, int LA, int LB, int PF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* a, const char* b, float* c, int64_t m, int64_t n, int64_t kk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NI = 16 / WAVES;  // DMA instructions per operand piece per wave (16 rows x 64 B each)
  constexpr int FB = WAVES == 8 ? 2 : 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = WAVES == 8 ? wave >> 2 : wave >> 1, wn = WAVES == 8 ? wave & 3 : wave & 1;
  const int64_t nt = n / 256;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m0 = ((slot / nt) * 8 + xcd) * 256, n0 = (slot % nt) * 256;
  const int64_t ntiles = kk / 32;
  int rowi[NI], cci[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    rowi[i] = (wave * NI + i) * 16 + (lane >> 2);
    cci[i] = ((lane & 3) ^ ((rowi[i] >> 2) & 3)) * 16;
  }
  auto issue = [&](int64_t t, int stage) {
    char* sa = smem + stage * STAGE + (wave * NI) * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      __builtin_amdgcn_global_load_lds((gptr)(a + op_off<LA>(m0 + rowi[i], t, m, kk) + cci[i]), (lptr)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(b + op_off<LB>(n0 + rowi[i], t, n, kk) + cci[i]), (lptr)(sa + OPA + i * 1024), 16, 0, 0);
    }
  };
  f32x16 acc[4][FB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int frow = lane & 31, fhalf = lane >> 5, key = (frow >> 2) & 3;
  const int a_lane = (wm * 128 + frow) * 64 + ((fhalf ^ key) << 4);
  const int b_lane = OPA + (wn * (FB * 32) + frow) * 64 + ((fhalf ^ key) << 4);
  uint32_t pfv = 0;
  if (MODE != 2)
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s, s);
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t % NSTAGE);
    if (MODE != 2) {
      // block t has landed: allow the NSTAGE-2 younger blocks (2*NI instructions each) [+ prefetch loads] in flight
      if constexpr (PF > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NSTAGE - 2) + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NSTAGE - 2)) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE != 2) {
      const int64_t nx = t + NSTAGE - 1;
      issue(nx < ntiles ? nx : ntiles - 1, (int)(nx % NSTAGE));
      if constexpr (PF > 0 && LA == 0) {
        if ((t & 3) == 0) {
          const int64_t tp = (t + PF < ntiles ? t + PF : ntiles - 4) & ~(int64_t)3;
          const int tid = threadIdx.x;
          constexpr int TPR = WAVES * 64 / 256;  // threads per row: 2 (8 waves) | 1 (4 waves)
          const char* pa = a + (m0 + tid / TPR) * kk * 2 + tp * 64 + (TPR == 2 ? (tid & 1) * 128 : 0);
          // (LDS-DMA of 4 bytes per lane into a scratch corner: no destination register to keep alive)
          char* scr = smem + NSTAGE * STAGE + wave * 256;
          __builtin_amdgcn_global_load_lds((gptr)pa, (lptr)scr, 4, 0, 0);
          if (TPR == 1) __builtin_amdgcn_global_load_lds((gptr)(pa + 128), (lptr)scr, 4, 0, 0);
        }
      }
    }
    if (MODE != 1) {
      const char* st = smem + cur * STAGE;
#pragma unroll
      for (int hk = 0; hk < 2; ++hk) {
        const int x = hk * 32;
        u32x4 fa[4], fb[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j) fb[j] = *reinterpret_cast<const u32x4*>(st + (b_lane ^ x) + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + (a_lane ^ x) + i * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < FB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = (float)pfv;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) c[threadIdx.x] = s;
}

template <int MODE, int LA, int LB, int PF, int WAVES>
void run(const char* name, const char* a, const char* b, float* c, int64_t m, int64_t n, int64_t kk) {
  const size_t lds = (size_t)NSTAGE * STAGE + 4096;
  auto kern = k<MODE, LA, LB, PF, WAVES>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = (unsigned)((m / 256) * (n / 256));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, a, b, c, m, n, kk);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, a, b, c, m, n, kk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %8.1f us  %7.1f TF/s  (%s)\n", name, ms * 100, 2.0 * m * n * kk / (ms * 1e-4) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int64_t m = 65536, n = 512, kk = 3456;
  char *a, *b;
  float* c;
  hipMalloc(&a, m * kk * 2);
  hipMalloc(&b, n * kk * 2);
  hipMalloc(&c, 4096);
  std::vector<uint16_t> h(m * kk);
  uint32_t x = 12345;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = (uint16_t)(0x3c00 + ((x >> 16) & 0x3ff) | ((x >> 31) << 15));
  }
  hipMemcpy(a, h.data(), m * kk * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), n * kk * 2, hipMemcpyHostToDevice);
#define R(MODE, LA, LB, PF, W, NAME) run<MODE, LA, LB, PF, W>(NAME, a, b, c, m, n, kk)
  for (int rep = 0; rep < 2; ++rep) {
    printf("-- 8 waves (2x4 of 128x64), DMA stream alone\n");
    R(1, 0, 0, 0, 8, "DMA  A row      B row");
    R(1, 0, 1, 0, 8, "DMA  A row      B tiled");
    R(1, 2, 0, 0, 8, "DMA  A fm128    B row");
    R(1, 2, 1, 0, 8, "DMA  A fm128    B tiled");
    R(1, 1, 0, 0, 8, "DMA  A tiled    B row");
    R(1, 1, 1, 0, 8, "DMA  A tiled    B tiled");
    R(1, 0, 0, 8, 8, "DMA  A row+pf8  B row");
    R(1, 0, 1, 8, 8, "DMA  A row+pf8  B tiled");
    R(1, 0, 1, 16, 8, "DMA  A row+pf16 B tiled");
    printf("-- 8 waves, full loop\n");
    R(0, 0, 0, 0, 8, "full A row      B row");
    R(0, 0, 1, 0, 8, "full A row      B tiled");
    R(0, 2, 1, 0, 8, "full A fm128    B tiled");
    R(0, 1, 1, 0, 8, "full A tiled    B tiled");
    R(0, 0, 1, 8, 8, "full A row+pf8  B tiled");
    R(2, 0, 0, 0, 8, "LDS reads + MFMA alone");
    printf("-- 4 waves (2x2 of 128x128)\n");
    R(2, 0, 0, 0, 4, "LDS reads + MFMA alone");
    R(1, 0, 0, 0, 4, "DMA  A row      B row");
    R(1, 0, 1, 0, 4, "DMA  A row      B tiled");
    R(1, 2, 1, 0, 4, "DMA  A fm128    B tiled");
    R(1, 1, 1, 0, 4, "DMA  A tiled    B tiled");
    R(0, 0, 0, 0, 4, "full A row      B row");
    R(0, 0, 1, 0, 4, "full A row      B tiled");
    R(0, 2, 1, 0, 4, "full A fm128    B tiled");
    R(0, 1, 1, 0, 4, "full A tiled    B tiled");
  }
  return 0;
}
