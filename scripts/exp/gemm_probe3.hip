// Round-4 probe (development aid, not product): the operand stream of the 256x256 bf16 tile through VGPRs
// (global_load_dwordx4 -> ds_write_b128, double-buffered LDS, D blocks of loads in flight) instead of the LDS-DMA ring
// the product kernel uses -- the vendor library's best kernel for these shapes (rocprofv3: MT256x256x64, four waves of
// 128x128, 16x16x32 MFMA, no direct-to-LDS) is 8-18 % faster than krs_gemm on h = x U / dh = dz K^T
// (profiles/r4y_vendor_gemm_compare.txt), and the LDS-DMA stream alone costs 174 us of the 205.
//   M = 65536, N = 512, K = 3456 bf16; MODE 0 full loop, 1 operand stream alone (no fragment reads, no MFMA)
//   WAVES 8 (2 x 4 waves of 128x64) | 4 (2 x 2 of 128x128);  BKB = bytes of K per row and block: 64 (32 k) | 128 (64 k:
//   whole 128-byte lines per row);  D = blocks of global loads in flight;  MI = 32 (32x32x16 MFMA) | 16 (16x16x32)
//   build: hipcc --offload-arch=gfx950 -O3 scripts/exp/gemm_probe3.hip -o scripts/exp/gemm_probe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int BKB>
__device__ __forceinline__ int swz(int row) { return BKB == 64 ? (row >> 2) & 3 : (row >> 1) & 7; }

template <int MODE, int WAVES, int BKB, int D, int MI, int VAR = 0>
__global__ __launch_bounds__(WAVES * 64) void kv(const char* __restrict__ a, const char* __restrict__ b, float* c,
                                                 int64_t m, int64_t n, int64_t kk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = WAVES * 64, CPR = BKB / 16, NI = 256 * CPR / NT, PIECE = 256 * BKB, STG = 2 * PIECE;
  constexpr int FB = WAVES == 8 ? 2 : 4;     // 32-column B fragments per wave (32x32 form)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = WAVES == 8 ? wave >> 2 : wave >> 1, wn = WAVES == 8 ? wave & 3 : wave & 1;
  const int64_t nt = n / 256;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m0 = ((slot / nt) * 8 + xcd) * 256, n0 = (slot % nt) * 256;
  const int ntiles = (int)(kk * 2 / BKB);
  // thread's chunk i of a piece: row = i * (NT / CPR) + tid / CPR, 16-byte chunk tid % CPR -- one 32-bit lane offset, the
  // rest is uniform (scalar registers)
  constexpr int RPI = NT / CPR;   // rows between a thread's consecutive chunks (a multiple of 16: the swizzle key is the same)
  const char* ab = a + m0 * kk * 2;
  const char* bb = b + n0 * kk * 2;
  const uint32_t ldb2 = (uint32_t)(kk * 2);
  const uint32_t vo = (uint32_t)(tid / CPR) * ldb2 + (uint32_t)(tid % CPR) * 16;
  const int lo0 = (tid / CPR) * BKB + (((tid % CPR) ^ swz<BKB>(tid / CPR)) << 4);
  u32x4 ra[D][NI], rb[D][NI];
  auto gload = [&](int t, int s) {
    const uint32_t off = (uint32_t)(t < ntiles ? t : ntiles - 1) * BKB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const size_t uni = (size_t)(i * RPI) * ldb2 + off;
      ra[s][i] = *reinterpret_cast<const u32x4*>(ab + uni + vo);
      rb[s][i] = *reinterpret_cast<const u32x4*>(bb + uni + vo);
    }
  };
  auto lstore = [&](int s, int stage) {
    char* st = smem + stage * STG;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      *reinterpret_cast<u32x4*>(st + lo0 + i * RPI * BKB) = ra[s][i];
      *reinterpret_cast<u32x4*>(st + PIECE + lo0 + i * RPI * BKB) = rb[s][i];
    }
  };
  // accumulators: 32x32 form acc[4][FB] of 16, 16x16 form acc16[8][2*FB] of 4 -- the same 128 / 256 registers
  f32x16 acc[4][FB];
  f32x4 acc16[8][2 * FB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2 * FB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.0f;

  auto compute = [&](int stage, int half) {   // half: -1 all k-steps of the block, 0 / 1 its first / second half
    const char* st = smem + stage * STG;
    if constexpr (MI == 32) {
      const int frow = lane & 31, fhalf = lane >> 5, key = swz<BKB>(frow);
#pragma unroll
      for (int ks = 0; ks < BKB / 32; ++ks) {
        if (half >= 0 && (ks >= BKB / 64) != (half == 1)) continue;
        const int cho = ((ks * 2 + fhalf) ^ key) << 4;
        u32x4 fa[4], fb[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j)
          fb[j] = *reinterpret_cast<const u32x4*>(st + PIECE + (wn * (FB * 32) + j * 32 + frow) * BKB + cho);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + (wm * 128 + i * 32 + frow) * BKB + cho);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < FB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      // 16x16x32: lane (r = lane & 15, q = lane >> 4) holds k = q * 8 .. + 8 of row r
      const int frow = lane & 15, q = lane >> 4, key = swz<BKB>(frow);
#pragma unroll
      for (int ks = 0; ks < BKB / 64; ++ks) {
        if (half >= 0 && BKB >= 128 && (ks >= BKB / 128) != (half == 1)) continue;
        if (half == 1 && BKB < 128) continue;
        const int cho = ((ks * 4 + q) ^ key) << 4;
        u32x4 fa[8], fb[2 * FB];
#pragma unroll
        for (int j = 0; j < 2 * FB; ++j)
          fb[j] = *reinterpret_cast<const u32x4*>(st + PIECE + (wn * (FB * 32) + j * 16 + frow) * BKB + cho);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + (wm * 128 + i * 16 + frow) * BKB + cho);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 2 * FB; ++j)
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                  __builtin_bit_cast(bf16x8, fb[j]), acc16[i][j], 0, 0, 0);
      }
    }
  };

#pragma unroll
  for (int s = 0; s < D; ++s) gload(s, s);
  lstore(0, 0);
  gload(D, 0);
  for (int j0 = 0; j0 < ntiles; j0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int j = j0 + s;
      __syncthreads();   // block j is in LDS for everyone; nobody reads stage (j + 1) & 1 any more
      if constexpr (VAR == 1) {          // the LDS writes of the next block between the two halves of this block's MFMAs
        if constexpr (MODE == 0) compute(j & 1, 0);
        lstore((s + 1) % D, (j + 1) & 1);
        gload(j + 1 + D, (s + 1) % D);
        if constexpr (MODE == 0) compute(j & 1, 1);
      } else if constexpr (VAR == 2) {   // ... behind them (the loads have had the whole block's compute to arrive)
        if constexpr (MODE == 0) compute(j & 1, -1);
        lstore((s + 1) % D, (j + 1) & 1);
        gload(j + 1 + D, (s + 1) % D);
      } else {
        lstore((s + 1) % D, (j + 1) & 1);
        gload(j + 1 + D, (s + 1) % D);
        if constexpr (MODE == 0) {
          if constexpr (VAR == 3) __builtin_amdgcn_s_setprio(1);
          compute(j & 1, -1);
          if constexpr (VAR == 3) __builtin_amdgcn_s_setprio(0);
        }
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2 * FB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc16[i][j][r];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int i = 0; i < NI; ++i) s += (float)(ra[d][i][0] ^ rb[d][i][3]);   // the trailing loads stay live
  if (s == 12345.678f) c[threadIdx.x] = s + smem[tid];
}

// ---- hand-pipelined form: 8 waves, 64-k blocks (128-byte rows), 32x32x16 MFMA --------------------------------------------
// Per 16-k step a wave issues, in this order: the fragment reads of the NEXT step (second register set), a quarter of the
// operand stream's work (steps 0 / 1: ds_write the A / B chunks of block j+1, whose loads were issued one block ago;
// steps 2 / 3: global loads of block j+2), then the 8 MFMAs of THIS step -- so the LDS latency and the stream's issue slots
// sit under MFMA time of the same wave.  One barrier per block, before the last step's MFMAs: behind it everybody's
// ds_writes of block j+1 are visible (first fragment read of the next block) and nobody reads stage j & 1 any more.
// SPLIT: 1 = waves 4-7 run the stream quarters in the opposite step order (their LDS writes fall where waves 0-3 load)
template <int SPLIT, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64) void kp(const char* __restrict__ a, const char* __restrict__ b, float* c,
                                          int64_t m, int64_t n, int64_t kk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = 128, NT = WAVES * 64, CPR = 8, NI = 256 * CPR / NT, RPI = NT / CPR, PIECE = 256 * BKB, STG = 2 * PIECE;
  constexpr int FB = WAVES == 8 ? 2 : 4;   // 32-column B fragments per wave: 128x64 | 128x128 per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = WAVES == 8 ? wave >> 2 : wave >> 1, wn = WAVES == 8 ? wave & 3 : wave & 1;
  const int64_t nt = n / 256;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m0 = ((slot / nt) * 8 + xcd) * 256, n0 = (slot % nt) * 256;
  const int ntiles = (int)(kk * 2 / BKB);
  const char* ab = a + m0 * kk * 2;
  const char* bb = b + n0 * kk * 2;
  const uint32_t ldb2 = (uint32_t)(kk * 2);
  const uint32_t vo = (uint32_t)(tid / CPR) * ldb2 + (uint32_t)(tid % CPR) * 16;
  const int lo0 = (tid / CPR) * BKB + (((tid % CPR) ^ swz<BKB>(tid / CPR)) << 4);
  u32x4 ra[NI], rb[NI];
  auto gload_a = [&](int t) {
    const uint32_t off = (uint32_t)(t < ntiles ? t : ntiles - 1) * BKB;
#pragma unroll
    for (int i = 0; i < NI; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ab + ((size_t)(i * RPI) * ldb2 + off) + vo);
  };
  auto gload_b = [&](int t) {
    const uint32_t off = (uint32_t)(t < ntiles ? t : ntiles - 1) * BKB;
#pragma unroll
    for (int i = 0; i < NI; ++i) rb[i] = *reinterpret_cast<const u32x4*>(bb + ((size_t)(i * RPI) * ldb2 + off) + vo);
  };
  auto lstore_a = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(smem + stage * STG + lo0 + i * RPI * BKB) = ra[i];
  };
  auto lstore_b = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(smem + stage * STG + PIECE + lo0 + i * RPI * BKB) = rb[i];
  };
  f32x16 acc[4][FB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int frow = lane & 31, fhalf = lane >> 5, key = swz<BKB>(frow);
  const int a_row = (wm * 128 + frow) * BKB, b_row = PIECE + (wn * (FB * 32) + frow) * BKB;
  u32x4 fa[2][4], fb[2][FB];
  auto rd = [&](int stage, int ks, int buf) {
    const char* st = smem + stage * STG;
    const int cho = ((ks * 2 + fhalf) ^ key) << 4;
#pragma unroll
    for (int j = 0; j < FB; ++j) fb[buf][j] = *reinterpret_cast<const u32x4*>(st + b_row + j * 32 * BKB + cho);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const u32x4*>(st + a_row + i * 32 * BKB + cho);
  };
  auto mm = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < FB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[buf][i]),
                                                            __builtin_bit_cast(bf16x8, fb[buf][j]), acc[i][j], 0, 0, 0);
  };
#define SB() __builtin_amdgcn_sched_barrier(0)
  // prologue: block 0 in LDS, block 1 in flight, first fragments read
  gload_a(0); gload_b(0);
  lstore_a(0); lstore_b(0);
  gload_a(1); gload_b(1);
  __syncthreads();
  rd(0, 0, 0);
  const bool flip = SPLIT == 1 && wave >= WAVES / 2;
  for (int j = 0; j < ntiles; ++j) {
    const int cur = j & 1, nxt = cur ^ 1;
    SB(); rd(cur, 1, 1); SB();
    if (!flip) {
      lstore_a(nxt); SB(); mm(0); SB();
      rd(cur, 2, 0); SB(); lstore_b(nxt); SB(); mm(1); SB();
      rd(cur, 3, 1); SB(); gload_a(j + 2); SB(); mm(0); SB();
      gload_b(j + 2); SB();
    } else {
      lstore_b(nxt); SB(); mm(0); SB();
      rd(cur, 2, 0); SB(); lstore_a(nxt); SB(); mm(1); SB();
      rd(cur, 3, 1); SB(); gload_b(j + 2); SB(); mm(0); SB();
      gload_a(j + 2); SB();
    }
    __syncthreads();
    SB(); rd(nxt, 0, 0); SB();
    mm(1); SB();
  }
#undef SB
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
#pragma unroll
  for (int i = 0; i < NI; ++i) s += (float)(ra[i][0] ^ rb[i][3]);
  s += (float)(fa[0][0][0] ^ fb[0][1][2]);
  if (s == 12345.678f) c[threadIdx.x] = s + smem[tid];
}

template <int SPLIT, int WAVES = 8>
void runp(const char* name, const char* a, const char* b, float* c, int64_t m, int64_t n, int64_t kk) {
  const size_t lds = 2 * 2 * 256 * 128;
  auto kern = kp<SPLIT, WAVES>;
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
  printf("%-64s %8.1f us  %7.1f TF/s  (%s)\n", name, ms * 100, 2.0 * m * n * kk / (ms * 1e-4) / 1e12,
         hipGetErrorString(hipGetLastError()));
}

template <int MODE, int WAVES, int BKB, int D, int MI, int VAR = 0>
void run(const char* name, const char* a, const char* b, float* c, int64_t m, int64_t n, int64_t kk) {
  const size_t lds = 2 * 2 * 256 * BKB;
  auto kern = kv<MODE, WAVES, BKB, D, MI, VAR>;
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
  printf("%-64s %8.1f us  %7.1f TF/s  (%s)\n", name, ms * 100, 2.0 * m * n * kk / (ms * 1e-4) / 1e12,
         hipGetErrorString(hipGetLastError()));
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
#define R(MODE, W, BKB, D, MI, NAME) run<MODE, W, BKB, D, MI>(NAME, a, b, c, m, n, kk)
#define RV(MODE, W, BKB, D, MI, VAR, NAME) run<MODE, W, BKB, D, MI, VAR>(NAME, a, b, c, m, n, kk)
  for (int rep = 0; rep < 2; ++rep) {
    printf("-- hand-pipelined (8 waves, 64-k blocks, 32x32x16)\n");
    runp<0>("pipelined: next step's fragment reads + a stream quarter, then 8 MFMAs", a, b, c, m, n, kk);
    runp<1>("pipelined, waves 4-7 with the stream quarters in the other order", a, b, c, m, n, kk);
    runp<0, 4>("pipelined, FOUR waves of 128x128", a, b, c, m, n, kk);
    R(0, 8, 128, 1, 32, "full  64-k blocks  1 in flight  32x32x16  (compiler-scheduled reference)");
    RV(0, 8, 128, 1, 32, 1, "full  64-k blocks  1 in flight  32x32x16  LDS writes between the halves");
  }
  return 0;
}
