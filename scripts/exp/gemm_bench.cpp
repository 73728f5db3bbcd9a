// Native A/B bench of krs_gemm at the C3 FeatureCross shapes (development aid; no torch import, so a
// GPU-box call costs seconds instead of minutes).  Links libkrs_hip.so through the C ABI.
//   build: hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o scripts/exp/gemm_bench \
//            -I include -L keras_rs_amd -lkrs_hip -Wl,-rpath,'$ORIGIN/../../keras_rs_amd'
//   run:   scripts/exp/gemm_bench [rounds] [B]
// For every pipeline setting (krs_gemm_set_option) it times the seven products / passes of one cross layer,
// interleaved over `rounds` rounds (median), and checks every output bit for bit against pipeline 0.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "krs.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)
#define KK(x)                                                      \
  do {                                                             \
    int r_ = (x);                                                  \
    if (r_ != 0) {                                                 \
      printf("krs error %d: %s (%s)\n", r_, krs_last_error(), #x); \
      exit(1);                                                     \
    }                                                              \
  } while (0)

__global__ void fill_bf16(uint16_t* p, int64_t rows, int64_t cols, int64_t ld, uint32_t seed, float scale) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    const float f = ((float)(x & 0xffffff) / 16777216.0f - 0.5f) * scale;
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    p[(i / cols) * ld + (i % cols)] = (uint16_t)(u >> 16);
  }
}
__global__ void diff_count(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

struct Buf {
  void* p = nullptr;
  int64_t rows = 0, cols = 0, ld = 0;
  size_t es = 2;
  size_t bytes() const { return (size_t)rows * ld * es; }
};
Buf alloc(int64_t rows, int64_t cols, int64_t ld, size_t es = 2) {
  // KRS_STAGGER=bytes: the i-th buffer starts (i % 16) * bytes into its allocation (do equal-sized streams
  // that start on the same DRAM bank / channel phase slow each other down?)
  static const size_t stagger = getenv("KRS_STAGGER") ? (size_t)atoll(getenv("KRS_STAGGER")) : 0;
  static int count = 0;
  Buf b; b.rows = rows; b.cols = cols; b.ld = ld; b.es = es;
  const size_t off = (size_t)(count++ % 16) * stagger;
  char* base;
  CK(hipMalloc(&base, b.bytes() + 16 * stagger));
  CK(hipMemset(base, 0, b.bytes() + 16 * stagger));
  b.p = base + off;
  return b;
}
void fill(Buf& b, uint32_t seed, float scale) {
  hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)b.p, b.rows, b.cols, b.ld, seed, scale);
}
unsigned long long diffs(const Buf& a, const Buf& b) {
  static unsigned long long* d = nullptr;
  if (!d) CK(hipMalloc(&d, 8));
  CK(hipMemset(d, 0, 8));
  hipLaunchKernelGGL(diff_count, dim3(2048), dim3(256), 0, 0, (const uint32_t*)a.p, (const uint32_t*)b.p,
                     (int64_t)(a.bytes() / 4), d);
  unsigned long long h;
  CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  return h;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 7;
  const int64_t B = argc > 2 ? atoll(argv[2]) : 65536;
  const int64_t d = 3456, pj = 512;
  const int64_t xpad = argc > 3 ? atoll(argv[3]) : 0;   // extra columns in the row stride of the [B, d] matrices
  const int64_t hpad = argc > 4 ? atoll(argv[4]) : 0;   // ... and of the [B, p] ones
  printf("krs %d  B=%lld d=%lld p=%lld  ld(x)=%lld ld(h)=%lld rounds=%d\n", krs_version(), (long long)B, (long long)d,
         (long long)pj, (long long)(d + xpad), (long long)(pj + hpad), rounds);
  Buf x0 = alloc(B, d, d + xpad), x = alloc(B, d, d + xpad), g = alloc(B, d, d + xpad);
  Buf Ut = alloc(pj, d, d), Vt = alloc(d, pj, pj), U = alloc(d, pj, pj), V = alloc(pj, d, d);
  Buf bias = alloc(1, d, d, 4);
  // KRS_ZERO=1: all-zero operands (the chip clocks to its power budget: same binary, less switching -- how much of a
  // product's time is the clock?)
  const float zs = getenv("KRS_ZERO") ? 0.0f : 1.0f;
  fill(x0, 1, 1.0f * zs); fill(x, 2, 1.0f * zs); fill(g, 3, 1.0f * zs);
  fill(Ut, 4, 0.05f * zs); fill(Vt, 5, 0.05f * zs); fill(U, 6, 0.05f * zs); fill(V, 7, 0.05f * zs);
  const int pipes[] = {0, 5, 4};   // 0 = the 128x128 two-stage kernels (reference schedule), 5 = the 32-k ring on 256x256 tiles for every product, 4 = default (64-k ring where K allows)
  constexpr int NP = 3;
  // outputs per pipeline
  Buf h[NP], y[NP], u[NP], dz[NP], dx0[NP], dk[NP], dh[NP], du[NP], dx[NP];
  float* dbias;
  CK(hipMalloc(&dbias, d * 4));
  for (int i = 0; i < NP; ++i) {
    h[i] = alloc(B, pj, pj + hpad); y[i] = alloc(B, d, d + xpad); u[i] = alloc(B, d, d + xpad);
    dz[i] = alloc(B, d, d + xpad); dx0[i] = alloc(B, d, d + xpad);
    dk[i] = alloc(pj, d, d, 4); dh[i] = alloc(B, pj, pj + hpad); du[i] = alloc(d, pj, pj, 4); dx[i] = alloc(B, d, d + xpad);
  }
  // (the K-contiguous products take split-K too when their output is too small to fill the chip: B <= ~24k)
  const size_t wsb = std::max(std::max(krs_gemm_workspace_bytes(pj, d, B, 1), krs_gemm_workspace_bytes(d, pj, B, 1)),
                              std::max(krs_gemm_workspace_bytes(B, pj, d, 0), krs_gemm_workspace_bytes(B, d, pj, 0)));
  void* ws;
  CK(hipMalloc(&ws, wsb ? wsb : 16));
  printf("split-K workspace %.1f MB\n", wsb / 1e6);
  const size_t ws2b = krs_gemm_cross_bwd_workspace_bytes(B, d);
  void* ws2;
  CK(hipMalloc(&ws2, ws2b ? ws2b : 16));
  hipStream_t st = 0;
  struct Case { const char* name; double flops; double bytes; };
  const double F = 2.0 * B * d * pj;
  const Case cases[8] = {
      {"fwd1 h = x U          (K=3456,N=512)", F, (double)(B * d + B * pj) * 2},
      {"fwd2 y = cross(h V)   (K=512,N=3456)", F, (double)(B * pj + 4 * B * d) * 2},
      {"bwd  elementwise dz, dx0", 0, (double)5 * B * d * 2},
      {"dK   = h^T dz         (split-K)", F, (double)(B * pj + B * d) * 2},
      {"dh   = dz V^T         (K=3456,N=512)", F, (double)(B * d + B * pj) * 2},
      {"dU   = x^T dh         (split-K)", F, (double)(B * d + B * pj) * 2},
      {"dx   = dh U^T + g     (K=512,N=3456)", F, (double)(B * pj + 2 * B * d) * 2},
      {"dx + elementwise bwd of the layer below", F, (double)(B * pj + 7 * B * d) * 2},
  };
  auto run_case = [&](int c, int i) {
    krs_gemm_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    switch (c) {
      case 0:
        KK(krs_gemm(x.p, x.ld, 0, Ut.p, Ut.ld, 1, h[i].p, h[i].ld, B, pj, d, KRS_BF16, KRS_BF16, nullptr, ws, wsb, st));
        break;
      case 1:
        ep.bias = (const float*)bias.p; ep.x0 = x0.p; ep.x = x.p; ep.ldx = x.ld; ep.u_out = u[i].p; ep.ldu = u[i].ld;
        KK(krs_gemm(h[i].p, h[i].ld, 0, Vt.p, Vt.ld, 1, y[i].p, y[i].ld, B, d, pj, KRS_BF16, KRS_BF16, &ep, ws, wsb, st));
        break;
      case 2:
        KK(krs_cross_epilogue_bwd(g.p, u[i].p, x0.p, x.p, dz[i].p, dx0[i].p, getenv("KRS_EW_ACC") ? 1 : 0, nullptr, dbias, B, d, x.ld, 0.0f,
                                  KRS_ACT_NONE, KRS_BF16, nullptr, 0, st));
        break;
      case 3:
        KK(krs_gemm(h[i].p, h[i].ld, 1, dz[i].p, dz[i].ld, 0, dk[i].p, dk[i].ld, pj, d, B, KRS_BF16, KRS_F32, nullptr, ws,
                    wsb, st));
        break;
      case 4:
        KK(krs_gemm(dz[i].p, dz[i].ld, 0, V.p, V.ld, 1, dh[i].p, dh[i].ld, B, pj, d, KRS_BF16, KRS_BF16, nullptr, ws, wsb,
                    st));
        break;
      case 5:
        KK(krs_gemm(x.p, x.ld, 1, dh[i].p, dh[i].ld, 0, du[i].p, du[i].ld, d, pj, B, KRS_BF16, KRS_F32, nullptr, ws, wsb,
                    st));
        break;
      case 6:
        ep.r = g.p; ep.ldr = g.ld; ep.beta = 1.0f;
        KK(krs_gemm(dh[i].p, dh[i].ld, 0, U.p, U.ld, 1, dx[i].p, dx[i].ld, B, d, pj, KRS_BF16, KRS_BF16, &ep, ws, wsb, st));
        break;
      case 7:   // krs_gemm_cross_bwd: the same product + dz / dL/dx0 (accumulating) / bias gradient of the layer below
        KK(krs_gemm_cross_bwd(dh[i].p, dh[i].ld, U.p, U.ld, g.p, g.ld, 1.0f, dx[i].p, dx[i].ld, x0.p, u[i].p, dz[i].p, dx0[i].p,
                              x.ld, 1, nullptr, 0, dbias, B, d, pj, KRS_ACT_NONE, KRS_BF16, ws2, ws2b, st));
        break;
    }
  };
  // warm-up + outputs of every pipeline (cases in dependency order)
  for (int i = 0; i < NP; ++i) {
    KK(krs_gemm_set_option(KRS_GEMM_OPT_PIPELINE, pipes[i]));
    for (int c = 0; c < 8; ++c) run_case(c, i);
  }
  CK(hipDeviceSynchronize());
  for (int i = 1; i < NP; ++i) {
    printf("pipeline %d vs 0: mismatching words  h %llu  y %llu  u %llu  dK %llu  dh %llu  dU %llu  dx %llu\n", pipes[i],
           diffs(h[0], h[i]), diffs(y[0], y[i]), diffs(u[0], u[i]), diffs(dk[0], dk[i]), diffs(dh[0], dh[i]),
           diffs(du[0], du[i]), diffs(dx[0], dx[i]));
  }
  // a sampled fp64 check of pipeline 0's h against the host (guards the baseline itself)
  {
    std::vector<uint16_t> hx((size_t)4 * x.ld), hu((size_t)Ut.rows * Ut.ld), hh((size_t)4 * h[0].ld);
    CK(hipMemcpy(hx.data(), x.p, hx.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hu.data(), Ut.p, hu.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hh.data(), h[0].p, hh.size() * 2, hipMemcpyDeviceToHost));
    auto f = [](uint16_t v) { uint32_t w = (uint32_t)v << 16; float r; memcpy(&r, &w, 4); return (double)r; };
    double worst = 0;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < pj; c += 37) {
        double s = 0;
        for (int k = 0; k < d; ++k) s += f(hx[(size_t)r * x.ld + k]) * f(hu[(size_t)c * Ut.ld + k]);
        worst = std::max(worst, std::abs(s - f(hh[(size_t)r * h[0].ld + c])) / (std::abs(s) + 1e-3));
      }
    printf("pipeline 0: h vs fp64 host, worst relative error on the sample %.3g (bf16 rounding ~4e-3)\n", worst);
  }
  // timing: rounds x pipelines x cases, one event pair per call
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t[NP][8];
  for (int r = 0; r < rounds; ++r)
    for (int c = 0; c < 8; ++c)
      for (int i = 0; i < NP; ++i) {
        KK(krs_gemm_set_option(KRS_GEMM_OPT_PIPELINE, pipes[i]));
        run_case(c, i);  // untimed: same-variant warm caches / clocks
        CK(hipEventRecord(e0, st));
        run_case(c, i);
        run_case(c, i);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t[i][c].push_back(ms * 500.0f);  // us per call
      }
  printf("%-40s", "case (median us | TF/s or GB/s)");
  for (int i = 0; i < NP; ++i) printf("   pipe %d          ", pipes[i]);
  printf("\n");
  double tot[NP] = {0, 0, 0};
  for (int c = 0; c < 8; ++c) {
    printf("%-40s", cases[c].name);
    for (int i = 0; i < NP; ++i) {
      std::sort(t[i][c].begin(), t[i][c].end());
      const double us = t[i][c][t[i][c].size() / 2];
      if (c != 7) tot[i] += us;     // (case 7 replaces cases 2 + 6 of a lower layer: listed, not summed)
      if (cases[c].flops > 0) printf("  %7.1f us %6.0f TF", us, cases[c].flops / us / 1e6);
      else printf("  %7.1f us %6.0f GB", us, cases[c].bytes / us / 1e3);
    }
    printf("\n");
  }
  printf("%-40s", "layer fwd+bwd total");
  for (int i = 0; i < NP; ++i) printf("  %7.1f us          ", tot[i]);
  printf("\n");
  return 0;
}
