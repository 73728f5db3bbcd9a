// Probe: where does the 256x256 LDS-DMA GEMM loop spend its time?  MODE 0 full, 1 DMA only, 2 MFMA only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROW_BYTES = 128;
template <int MODE, int NSTAGE, int BKB, bool TILED>  // BKB: bytes of K per row per stage (128 or 64); TILED: [K/BK][rows][BK] operand layout
__global__ __launch_bounds__(512) void k(const char* a, const char* b, float* c, int64_t m, int64_t n, int64_t kk,
                                          int64_t lda, int64_t ldb, int64_t mwrap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = 2;
  constexpr int OPA = 256 * BKB;
  constexpr int STAGE = 512 * BKB;
  constexpr int RPI = 1024 / BKB;          // rows per DMA instruction (8 | 16)
  constexpr int CPR = BKB / 16;            // 16-B chunks per row (8 | 4)
  constexpr int NI = 256 / RPI / 8;        // DMA instrs per operand per wave (4 | 2)
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t nt = n / 256;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m0 = (((slot / nt) * 8 + xcd) * 256) % mwrap, n0 = (slot % nt) * 256;
  const int64_t ntiles = kk * ES / BKB;
  const char* asrc[NI]; const char* bsrc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = (wave * NI + i) * RPI + lane / CPR;
    const int cc = (lane % CPR) ^ ((r >> 1) & (CPR - 1));
    asrc[i] = a + (TILED ? (m0 + r) * BKB : ((m0 + r) * lda) * ES) + cc * 16;
    bsrc[i] = b + (TILED ? (n0 + r) * BKB : ((n0 + r) * ldb) * ES) + cc * 16;
  }
  auto issue = [&](int64_t t, int stage) {
    char* sa = smem + stage * STAGE + (wave * NI) * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      __builtin_amdgcn_global_load_lds((gptr)(asrc[i] + t * (TILED ? m * BKB : BKB)), (lptr)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(bsrc[i] + t * (TILED ? n * BKB : BKB)), (lptr)(sa + OPA + i * 1024), 16, 0, 0);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int frow = lane & 31, fhalf = lane >> 5;
  int aoff[4], boff[2], akey[4], bkey[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + frow; aoff[i] = ra * BKB; akey[i] = (ra >> 1) & (CPR - 1); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + frow; boff[j] = OPA + rb * BKB; bkey[j] = (rb >> 1) & (CPR - 1); }
  // prologue: NSTAGE-1 tiles in flight
  if (MODE != 2)
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s, s);
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t % NSTAGE);
    // wait until tile t has landed: allow (NSTAGE-2) younger tiles in flight
    if (MODE != 2) {
      if constexpr (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (NSTAGE == 3) { if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else { if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t visible to all; everyone done with tile t-1
    if (MODE != 2) { const int64_t nx = t + NSTAGE - 1; issue(nx < ntiles ? nx : ntiles - 1, (int)(nx % NSTAGE)); }
    if (MODE != 1) {
      const char* st = smem + cur * STAGE;
#pragma unroll
      for (int ks = 0; ks < BKB / 32; ++ks) {
        const int cc = ks * 2 + fhalf;
        u32x4 fa[4], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x4*>(st + boff[j] + ((cc ^ bkey[j]) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + aoff[i] + ((cc ^ akey[i]) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) c[threadIdx.x] = s;
}
template <int MODE, int NSTAGE, int BKB, bool TILED = false>
void run(const char* name, const char* a, const char* b, float* c, int64_t m, int64_t n, int64_t kk, int64_t ld = 0, int64_t mwrap = 0) {
  if (!ld) ld = kk;
  if (!mwrap) mwrap = m;
  const size_t lds = (size_t)NSTAGE * 512 * BKB;
  auto kern = k<MODE, NSTAGE, BKB, TILED>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = (unsigned)((m / 256) * (n / 256));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a, b, c, m, n, kk, ld, ld, mwrap);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a, b, c, m, n, kk, ld, ld, mwrap);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %8.1f us  %7.1f TF/s  (err %s)\n", name, ms * 100, 2.0 * m * n * kk / (ms * 1e-4) / 1e12, hipGetErrorString(hipGetLastError()));
}
int main() {
  const int64_t m = 65536, n = 512, kk = 3456;
  char *a, *b; float* c;
  hipMalloc(&a, m * 4096 * 2); hipMalloc(&b, n * 4096 * 2); hipMalloc(&c, 4096);
  std::vector<uint16_t> h(m * kk);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 16) & 0x3ff) | ((x >> 31) << 15)); }
  hipMemcpy(a, h.data(), m * kk * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), n * kk * 2, hipMemcpyHostToDevice);
  run<0, 2, 128>("full   2 stages BK=64", a, b, c, m, n, kk);
  run<1, 2, 128>("DMA    2 stages BK=64", a, b, c, m, n, kk);
  run<2, 2, 128>("MFMA   2 stages BK=64", a, b, c, m, n, kk);
  run<0, 4, 64>("full   4 stages BK=32", a, b, c, m, n, kk);
  run<1, 4, 64>("DMA    4 stages BK=32", a, b, c, m, n, kk);
  run<2, 4, 64>("MFMA   4 stages BK=32", a, b, c, m, n, kk);
  run<0, 2, 128, true>("full   2 stages BK=64 tiled", a, b, c, m, n, kk);
  run<1, 2, 128, true>("DMA    2 stages BK=64 tiled", a, b, c, m, n, kk);
  run<0, 4, 64, true>("full   4 stages BK=32 tiled", a, b, c, m, n, kk);
  run<1, 4, 64, true>("DMA    4 stages BK=32 tiled", a, b, c, m, n, kk);
  // row stride of the operands (elements): 54 / 55 / 56 / 64 lines of 128 B per row
  run<1, 2, 128>("DMA    2 stages BK=64 ld=3456", a, b, c, m, n, kk, 3456);
  run<1, 2, 128>("DMA    2 stages BK=64 ld=3520", a, b, c, m, n, kk, 3520);
  run<1, 2, 128>("DMA    2 stages BK=64 ld=3584", a, b, c, m, n, kk, 3584);
  run<1, 2, 128>("DMA    2 stages BK=64 ld=4096", a, b, c, m, n, kk, 4096);
  run<0, 2, 128>("full   2 stages BK=64 ld=3520", a, b, c, m, n, kk, 3520);
  run<1, 4, 64>("DMA    4 stages BK=32 ld=3520", a, b, c, m, n, kk, 3520);
  // A confined to its first rows (cache-resident after the first touch): is the row-major limit HBM-side?
  run<1, 2, 128>("DMA    2 stages BK=64 A=8192 rows", a, b, c, m, n, kk, 0, 8192);
  run<1, 4, 64>("DMA    4 stages BK=32 A=8192 rows", a, b, c, m, n, kk, 0, 8192);
  run<1, 2, 128, true>("DMA    2 stages BK=64 tiled A=8192 rows", a, b, c, m, n, kk, 0, 8192);
  run<1, 4, 64, true>("DMA    4 stages BK=32 tiled A=8192 rows", a, b, c, m, n, kk, 0, 8192);
  run<1, 2, 128>("DMA    2 stages BK=64 A=2048 rows", a, b, c, m, n, kk, 0, 2048);
  run<1, 4, 64>("DMA    4 stages BK=32 A=2048 rows", a, b, c, m, n, kk, 0, 2048);
  run<0, 2, 128>("full   2 stages BK=64 A=2048 rows", a, b, c, m, n, kk, 0, 2048);
  return 0;
}
