// K4 -- DotInteraction (DLRM pairwise dots) forward and backward.
//
// Replaces DotInteraction.call (keras_rs/src/layers/feature_interaction/dot_interaction.py:170-203):
// stack -> matmul(X, X^T) -> tril gather / masked flatten, which the reference
// runs as a stack copy, a batched matmul and a gather.  The op moves
// B*F*D*s + B*F(F-1)/2*s bytes for 2*B*F^2*D flops: HBM-bound.
//
// Forward: one wave64 per sample.  The F <= 32 feature rows of the sample are
// read straight from HBM into MFMA operand registers (lane = feature row, 16 B
// of the row per K step; P = X X^T so the A and the B operand are the SAME
// registers), one 32x32 accumulator tile per sample, and the lower triangle is
// written from the accumulator layout (lanes 0..31 = consecutive columns of one
// row -> contiguous stores).  No LDS, no stack copy.
// Backward: dX = (G + G^T) X per sample on the vector ALU (see dot_bwd_valu_kernel).
// Shapes outside the fast path (32 < F <= 64, odd D, unaligned) use plain kernels.
#include <algorithm>
#include <type_traits>

#include "krs_common.h"

namespace krs {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxFast = 32;

struct DotParams {
  const void* feat[kMaxFast];
  int64_t ld[kMaxFast];
  void* gfeat[kMaxFast];  // backward outputs
  int64_t gld[kMaxFast];
  int n_feats;
  int64_t batch;
  int dim;
  int self_inter;
  int skip_gather;
  void* out;          // forward output / backward grad_out
  int64_t out_ld;
  uint32_t acc_mask;  // backward: bit f set = gfeat[f] already holds a gradient, add to it
};

__host__ __device__ constexpr int tri_cols(int rows, bool self) { return self ? rows * (rows + 1) / 2 : rows * (rows - 1) / 2; }
__device__ __forceinline__ bool pair_kept(int i, int j, int self_inter) { return self_inter ? j <= i : j < i; }
// column of pair (i,j) in the gathered output (row-major lower triangle, dot_interaction.py:118-132)
__device__ __forceinline__ int64_t pair_col(int i, int j, int F, int self_inter, int skip_gather) {
  if (skip_gather) return (int64_t)i * F + j;
  return self_inter ? (int64_t)i * (i + 1) / 2 + j : (int64_t)i * (i - 1) / 2 + j;
}

// select p.feat[f] / p.ld[f] without dynamically indexing the kernarg struct
__device__ __forceinline__ void pick_feature(const DotParams& p, int f, const char*& ptr, int64_t& ld) {
  ptr = nullptr;
  ld = 0;
#pragma unroll
  for (int i = 0; i < kMaxFast; ++i)
    if (i == f) {
      ptr = reinterpret_cast<const char*>(p.feat[i]);
      ld = p.ld[i];
    }
}

// KS > 0: the number of k steps is known at compile time (8 = D 128 in bf16): the sample's loads are issued as
// one batch and the NEXT sample's loads are in flight while this one is multiplied and written (a run-time loop
// of load -> MFMA pairs was one HBM latency per k step).  The pooled dots leave through a wave-private LDS row in
// output order, so a sample is written by ceil(cols / 64) stores of 128 contiguous bytes instead of one 2-byte
// scattered store per (row, lane) -- 27 partial-line stores per sample at F = 27.
template <int ES, int KS>
__global__ __launch_bounds__(256) void dot_fwd_mfma_kernel(const DotParams p) {
  typedef typename std::conditional<ES == 2, uint16_t, float>::type elem_t;
  __shared__ elem_t stage_all[4][kMaxFast * kMaxFast];
  const int lane = threadIdx.x & 63;
  const int f = lane & 31;
  const int half = lane >> 5;
  const int F = p.n_feats;
  const char* base;
  int64_t ld;
  pick_feature(p, f < F ? f : 0, base, ld);
  constexpr int VE = 16 / ES;  // elements per 16-byte piece
  constexpr int NV = KS > 0 ? KS : 1;
  const int ksteps = KS > 0 ? KS : (p.dim + 2 * VE - 1) / (2 * VE);
  const int64_t waves = (int64_t)gridDim.x * 4;
  const int ncols = p.skip_gather ? F * F : tri_cols(F, p.self_inter != 0);
  elem_t* stage = stage_all[threadIdx.x >> 6];
  elem_t* outp = reinterpret_cast<elem_t*>(p.out);
  const u32x4 zero = {0, 0, 0, 0};
  auto load_all = [&](int64_t b, u32x4(&v)[NV]) {   // KS > 0 only: every address is valid (f clamped, k < dim)
    const char* row = base + b * ld * ES + half * 16;
#pragma unroll
    for (int ks = 0; ks < NV; ++ks) v[ks] = *reinterpret_cast<const u32x4*>(row + ks * 32);
  };
  u32x4 cur[NV], nxt[NV];
  int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if constexpr (KS > 0)
    if (b < p.batch) load_all(b, cur);
  for (; b < p.batch; b += waves) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if constexpr (KS > 0) {
      load_all(b + waves < p.batch ? b + waves : b, nxt);
#pragma unroll
      for (int ks = 0; ks < NV; ++ks) {
        const u32x4 v = f < F ? cur[ks] : zero;
        if constexpr (ES == 2) {
          const bf16x8 x = __builtin_bit_cast(bf16x8, v);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x = __uint_as_float(v[q]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < NV; ++ks) cur[ks] = nxt[ks];
    } else {
      const char* row = base + b * ld * ES;
      for (int ks = 0; ks < ksteps; ++ks) {
        const int k = ks * 2 * VE + half * VE;
        u32x4 v = zero;
        if (f < F && k < p.dim) v = *reinterpret_cast<const u32x4*>(row + (int64_t)k * ES);
        if constexpr (ES == 2) {
          const bf16x8 x = __builtin_bit_cast(bf16x8, v);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x = __uint_as_float(v[q]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc, 0, 0, 0);
          }
        }
      }
    }
    // accumulator: P[i][j], j = lane & 31, i = (r & 3) + 8*(r >> 2) + 4*half  ->  the sample's output row in LDS
    const int j = f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < F && j < F) {
        const bool keep = pair_kept(i, j, p.self_inter);
        const float v = keep ? acc[r] : 0.0f;
        if (p.skip_gather || keep) {
          const int c = (int)pair_col(i, j, F, p.self_inter, p.skip_gather);
          if constexpr (ES == 2) stage[c] = f32_to_bf16(v); else stage[c] = v;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // wave-private row: LDS operations of a wave are in order
    elem_t* orow = outp + b * p.out_ld;
    for (int e = lane; e < ncols; e += 64) orow[e] = stage[e];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// Backward fast path (F <= 32, even D, 4-byte aligned rows, < 2 GiB per operand): fp32 FMAs, no MFMA.
// dX[i] = sum_j Gs[i][j] X[j] is 2*F^2*D flops per sample against 2*F*D*s bytes: ~27 flop/B at
// F = 27, far below the VALU ridge, and the contraction index (the feature) is the strided one in
// memory, so feeding MFMA would need a transposing LDS pass.  Instead every lane owns two adjacent
// columns d of ONE sample (a wave covers 128 columns: one sample at D = 128, 128/D samples for
// smaller D), keeps the sample's rows x[j] and accumulators as float2 registers straight from
// coalesced global loads, and walks the pairs j < i: one LDS broadcast read of G[i][j] (the
// sample's gradient row, staged as fp32 in lower-triangle order whatever the output layout was)
// feeds acc[i] += g x[j] and acc[j] += g x[i] as two v_pk_fma_f32.
// The pair loops are fully unrolled over FR >= F rows (template), so registers AND the LDS offsets
// are static; rows F..FR-1 are skipped by wave-uniform branches.
// One iteration = (group of spw samples, 128-column chunk); the loads of iteration n+1 are issued
// before the FMAs of iteration n, so HBM latency hides under the arithmetic.
typedef float f32x2 __attribute__((ext_vector_type(2)));


// ACC: features whose bit is set in p.acc_mask receive `existing + dX` (fp32 sum, one rounding): the existing
// values are requested together with the sample's rows (same clamped addresses), one iteration ahead.
template <int ES, int FR, bool SELF, bool MULTI, bool ACC>  // MULTI: several samples per wave (D < 128)
__global__ __launch_bounds__(64) void dot_bwd_valu_kernel(const DotParams p, int lps_log2, int spw) {
  constexpr int GSTRIDE = tri_cols(FR, SELF) + 1;                    // LDS floats per sample
  constexpr int NG = MULTI ? 16 : (tri_cols(FR, SELF) + 63) / 64;    // gradient elements per lane in flight
  // LDS: [64] {pointer, row stride in bytes} of the F inputs and F outputs, then [2][spw][GSTRIDE]
  // gradient rows.  The pointer table lives in LDS (not in SGPRs) on purpose: 4 * FR kernel
  // arguments kept live across the loop would spill, and LDS reads cannot be hoisted out of it.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* tab = reinterpret_cast<u32x4*>(smem);
  float* g_l = reinterpret_cast<float*>(smem + 64 * sizeof(u32x4));
  const int lane = threadIdx.x;
  const int F = p.n_feats;
  const int lps = 1 << lps_log2;          // lanes per sample
  const int slot = lane >> lps_log2;      // which of the wave's samples this lane works on
  const int dp = lane & (lps - 1);
  const int ncols = p.skip_gather ? F * F : tri_cols(F, SELF);  // columns of the incoming gradient
  const int gtotal = spw * ncols;
  const int chunks = (p.dim + 2 * lps - 1) / (2 * lps);
  const int64_t n_groups = (p.batch + spw - 1) / spw;
  // this workgroup's groups are blockIdx.x + k * gridDim.x; local iteration n = k * chunks + chunk
  const int64_t my_groups = blockIdx.x < n_groups ? (n_groups - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  const int64_t n_iters = my_groups * chunks;
  typedef typename std::conditional<ES == 2, uint32_t, f32x2>::type raw_t;   // two adjacent columns
  typedef typename std::conditional<ES == 2, uint16_t, float>::type gelem_t;

  {
    const int f = lane & 31;
    uint64_t ptr = 0;
    int64_t stride = 0;
#pragma unroll
    for (int q = 0; q < kMaxFast; ++q)  // static kernarg indices
      if (q == f) {
        ptr = reinterpret_cast<uint64_t>(lane < 32 ? p.feat[q] : p.gfeat[q]);
        stride = lane < 32 ? p.ld[q] : p.gld[q];
      }
    tab[lane] = u32x4{(uint32_t)ptr, (uint32_t)(ptr >> 32), (uint32_t)stride * ES, 0u};
  }

  raw_t xr[FR];
  raw_t ar[ACC ? FR : 1];
  gelem_t gr[NG];
  auto issue = [&](int64_t it) {
    const int64_t k = it / chunks;
    const int ch = (int)(it - k * chunks);
    const int64_t grp = blockIdx.x + k * gridDim.x;
    const int64_t b = grp * spw + slot;
    const bool live = slot < spw && b < p.batch;
    const int d = ch * 2 * lps + 2 * dp;
    const uint32_t bc = (uint32_t)(live ? b : p.batch - 1);   // clamped: every load is unconditional
    const uint32_t dc = (live && d < p.dim) ? d : 0;
    if (ch == 0) {
#pragma unroll
      for (int q = 0; q < NG; ++q) {  // clamped, unconditional, converted only when consumed
        const int idx = min(q * 64 + lane, gtotal - 1);
        const int sidx = MULTI ? idx / ncols : 0;
        const int c = idx - sidx * ncols;
        const int64_t bb = min(grp * spw + sidx, p.batch - 1);
        gr[q] = reinterpret_cast<const gelem_t __attribute__((address_space(1)))*>(
            reinterpret_cast<uint64_t>(p.out))[bb * p.out_ld + c];
      }
    }
#pragma unroll
    for (int j = 0; j < FR; ++j) {  // entries j >= F repeat the last feature (host side)
      const u32x4 e = tab[j];
      const uint64_t addr = (((uint64_t)e[1] << 32) | e[0]) + (uint64_t)bc * e[2] + dc * ES;
      xr[j] = *reinterpret_cast<const raw_t __attribute__((address_space(1)))*>(addr);
      if constexpr (ACC) {
        const u32x4 ge = tab[32 + j];
        const uint64_t gaddr = (((uint64_t)ge[1] << 32) | ge[0]) + (uint64_t)bc * ge[2] + dc * ES;
        ar[j] = *reinterpret_cast<const raw_t __attribute__((address_space(1)))*>(gaddr);
      }
    }
  };

  if (n_iters > 0) issue(0);
  int parity = 0;
  for (int64_t it = 0; it < n_iters; ++it) {
    const int64_t k = it / chunks;
    const int ch = (int)(it - k * chunks);
    const int64_t grp = blockIdx.x + k * gridDim.x;
    const int64_t b = grp * spw + slot;
    const int d = ch * 2 * lps + 2 * dp;
    const bool act = slot < spw && b < p.batch && d < p.dim;
    if (ch == 0) {
      parity ^= 1;
      float* dstg = g_l + parity * spw * GSTRIDE;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const int idx = q * 64 + lane;
        const int sidx = MULTI ? idx / ncols : 0;
        int c = idx - sidx * ncols;
        bool keep = idx < gtotal;
        if (p.skip_gather) {  // [F, F] masked layout -> lower-triangle order
          const int i = c / F, j = c - i * F;
          keep = keep && (SELF ? j <= i : j < i);
          c = tri_cols(i, SELF) + j;
        }
        float v;
        if constexpr (ES == 2) v = bf16_to_f32(gr[q]); else v = gr[q];
        if (keep) dstg[sidx * GSTRIDE + c] = v;
      }
    }
    f32x2 x[FR], acc[FR];
    raw_t ac[ACC ? FR : 1];  // this iteration's existing gradients (ar is re-filled below)
#pragma unroll
    for (int j = 0; j < FR; ++j) {
      if constexpr (ACC) {
        ac[j] = ar[j];
        asm volatile("" : "+v"(ac[j]));
      }
      if constexpr (ES == 2) x[j] = f32x2{__uint_as_float(xr[j] << 16), __uint_as_float(xr[j] & 0xffff0000u)};
      else x[j] = xr[j];
      // pin the unpacked value here: left alone, the compiler sinks the unpack into the FMA section
      // and then waits on the loads issued below (and the previous stores) through the in-order vmcnt
      asm volatile("" : "+v"(x[j]));
      acc[j] = f32x2{0.0f, 0.0f};
    }
    // single-wave workgroup: LDS operations of a wave complete in order, the barrier only keeps the
    // compiler from moving the reads below over the writes above (no vmcnt wait as __syncthreads has)
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    issue(it + 1 < n_iters ? it + 1 : it);
    const float* grow = g_l + parity * spw * GSTRIDE + (MULTI ? (slot < spw ? slot : 0) * GSTRIDE : 0);
#pragma unroll
    for (int i = 0; i < FR; ++i) {
      // The wave-uniform branch also makes every row its own basic block: as one straight-line
      // block the compiler hoists all F(F-1)/2 LDS reads to the top and spills.
      if (i < F) {
#pragma unroll
        for (int j = 0; j < i; ++j) {
          const float g = grow[tri_cols(i, SELF) + j];
          const f32x2 g2 = {g, g};
          acc[i] += g2 * x[j];
          acc[j] += g2 * x[i];
        }
        if constexpr (SELF) {
          const float g = 2.0f * grow[tri_cols(i, SELF) + i];
          acc[i] += f32x2{g, g} * x[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FR; ++i) {
      if (i < F && act) {
        if constexpr (ACC) {
          if ((p.acc_mask >> i) & 1u) {  // wave-uniform
            if constexpr (ES == 2) acc[i] += f32x2{__uint_as_float(ac[i] << 16), __uint_as_float(ac[i] & 0xffff0000u)};
            else acc[i] += ac[i];
          }
        }
        const u32x4 e = tab[32 + i];
        const uint64_t dst = (((uint64_t)e[1] << 32) | e[0]) + (uint64_t)(uint32_t)b * e[2] + d * ES;
        if constexpr (ES == 2)
          *reinterpret_cast<uint32_t __attribute__((address_space(1)))*>(dst) = pack_bf16x2(acc[i][0], acc[i][1]);
        else
          *reinterpret_cast<f32x2 __attribute__((address_space(1)))*>(dst) = acc[i];
      }
    }
  }
}

template <int ES, int FR, bool SELF, bool MULTI>
void launch_dot_bwd(const DotParams& p, int lps_log2, int spw, unsigned blocks, hipStream_t st) {
  const size_t lds = 64 * 16 + (size_t)2 * spw * (tri_cols(FR, SELF) + 1) * sizeof(float);  // table + double buffer
  if (p.acc_mask)
    hipLaunchKernelGGL((dot_bwd_valu_kernel<ES, FR, SELF, MULTI, true>), dim3(blocks), dim3(64), lds, st, p, lps_log2, spw);
  else
    hipLaunchKernelGGL((dot_bwd_valu_kernel<ES, FR, SELF, MULTI, false>), dim3(blocks), dim3(64), lds, st, p, lps_log2, spw);
}

// ---- backward on the matrix cores (bf16, F <= 32, D a multiple of 32 up to 128, 16-byte aligned rows) ----
// dX = Gs X per sample is C = A^T B with A[j][i] = Gs[j][i] (32 x 32, symmetric; every entry is ONE element of the
// incoming gradient, or twice one on the diagonal: exactly representable in bf16) and B[j][d] = X[j][d].  Both
// operands are strided in the contraction index j (the feature), which is what gfx950's transposing LDS read
// serves: the sample's rows go to LDS as they lie in memory (the K-strided image of gemm_tn_glds_kernel) and
// every MFMA operand is two ds_read_b64_tr_b16.  8 MFMA per sample at D = 128 replace ~750 packed FMAs and ~340
// LDS broadcast reads per lane (that kernel issued ~5000 instructions per sample and ran at 3 TB/s).  The result
// leaves through a wave-private fp32 LDS block per 32 columns so that lanes hold 8 consecutive columns of one
// feature row: 16-byte loads of the stored gradient (ACC) and 16-byte stores.  The loads of sample n+1 are issued
// before the MFMAs of sample n.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 dot_read_tr16(uint32_t addr) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* trptr;
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr)(uintptr_t)addr));
}

template <bool SELF, bool ACC, int NG, int NB>   // NG: gradient elements per lane; NB: 32-column blocks provided for (D <= 32 NB)
__global__ __launch_bounds__(128) void dot_bwd_mfma_kernel(const DotParams p, int x_bytes) {
  constexpr int kWavesPerWg = 2, GB = 2048, SST = 36;   // Gs image bytes; staging row stride (floats)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = p.n_feats, nblk = p.dim / 32;
  // LDS: [64] {pointer, row stride in bytes} of the F inputs and F outputs (shared), then per wave
  // X image | Gs image | staging block [32][SST] floats
  u32x4* tab = reinterpret_cast<u32x4*>(smem);
  char* mine = smem + 64 * sizeof(u32x4) + (size_t)wave * (x_bytes + GB + 32 * SST * 4);
  char* ximg = mine;
  char* gimg = mine + x_bytes;
  float* stage = reinterpret_cast<float*>(mine + x_bytes + GB);
  if (threadIdx.x < 64) {
    const int f = threadIdx.x & 31;
    uint64_t ptr = 0;
    int64_t stride = 0;
#pragma unroll
    for (int q = 0; q < kMaxFast; ++q)  // static kernarg indices
      if (q == f) {
        ptr = reinterpret_cast<uint64_t>(threadIdx.x < 32 ? p.feat[q] : p.gfeat[q]);
        stride = threadIdx.x < 32 ? p.ld[q] : p.gld[q];
      }
    tab[threadIdx.x] = u32x4{(uint32_t)ptr, (uint32_t)(ptr >> 32), (uint32_t)stride * 2u, 0u};
  }
  for (int i = lane * 16; i < x_bytes + GB; i += 64 * 16) *reinterpret_cast<u32x4*>(mine + i) = u32x4{0, 0, 0, 0};
  __syncthreads();
  const int ncols = p.skip_gather ? F * F : tri_cols(F, SELF);
  const int xrounds = (F + 3) / 4;                 // 4 feature rows per load instruction, 16 lanes x 16 B per row
  const int cpr = p.dim / 8;                       // 16-byte chunks per feature row
  const int items = F * 4;                         // output items of a 32-column block: (feature, chunk of 8)
  const int64_t waves = (int64_t)gridDim.x * kWavesPerWg;
  u32x4 xr[8];                                     // up to 32 rows
  uint16_t gr[NG];
  u32x4 ar[ACC ? NB * 2 : 1];                      // stored gradients: [block][round]
  const uint16_t* gout = reinterpret_cast<const uint16_t*>(p.out);
  auto issue = [&](int64_t b) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = t * 4 + (lane >> 4);
      if (t < xrounds) {
        const u32x4 e = tab[f < F ? f : F - 1];
        const int c = lane & 15;
        const uint64_t addr = (((uint64_t)e[1] << 32) | e[0]) + (uint64_t)b * e[2] + (uint32_t)(c < cpr ? c : 0) * 16u;
        xr[t] = *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(addr);
      }
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) gr[q] = gout[b * p.out_ld + min(q * 64 + lane, ncols - 1)];
    if constexpr (ACC) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (nb < nblk) {
            const int it = min(t * 64 + lane, items - 1);
            const u32x4 e = tab[32 + (it >> 2)];
            const uint64_t addr = (((uint64_t)e[1] << 32) | e[0]) + (uint64_t)b * e[2] + (uint32_t)(nb * 32 + (it & 3) * 8) * 2u;
            ar[nb * 2 + t] = *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(addr);
          }
    }
  };
  const uint32_t xbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ximg;
  const uint32_t gbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)gimg;
  const int g4 = lane >> 4, ii = lane & 15;
  const uint32_t frag_x = (uint32_t)((ii >> 2) * 64 + (g4 & 1) * 32 + (ii & 3) * 8 + (g4 >> 1) * 2048);
  const uint32_t frag_g = (uint32_t)((ii >> 2) * 64 + (g4 & 1) * 32 + (ii & 3) * 8 + (g4 >> 1) * 512);
  int64_t b = (int64_t)blockIdx.x * kWavesPerWg + wave;
  if (b < p.batch) issue(b);
  for (; b < p.batch; b += waves) {
    // ---- this sample's rows and gradient matrix into LDS ----
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = t * 4 + (lane >> 4), c = lane & 15;
      if (t < xrounds && f < F && c < cpr) {
        const int col = c * 8;
        *reinterpret_cast<u32x4*>(ximg + (col >> 7) * 8192 + (f >> 2) * 1024 + (((col & 127) >> 5) * 4 + (f & 3)) * 64 +
                                  (col & 31) * 2) = xr[t];
      }
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int c = q * 64 + lane;
      if (c < ncols) {
        int i, j;
        if (p.skip_gather) {
          i = c / F;
          j = c - i * F;
        } else if (SELF) {
          i = (int)((sqrtf(8.0f * c + 1.0f) - 1.0f) * 0.5f);
          while (i * (i + 1) / 2 > c) --i;
          while ((i + 1) * (i + 2) / 2 <= c) ++i;
          j = c - i * (i + 1) / 2;
        } else {
          i = (int)((sqrtf(8.0f * c + 1.0f) + 1.0f) * 0.5f);
          while (i * (i - 1) / 2 > c) --i;
          while ((i + 1) * i / 2 <= c) ++i;
          j = c - i * (i - 1) / 2;
        }
        if (SELF ? j <= i : j < i) {
          uint16_t* gi = reinterpret_cast<uint16_t*>(gimg);
          if (i == j) {
            gi[(i >> 2) * 128 + (i & 3) * 32 + i] = f32_to_bf16(2.0f * bf16_to_f32(gr[q]));
          } else {
            gi[(j >> 2) * 128 + (j & 3) * 32 + i] = gr[q];   // image[k = j][column i]
            gi[(i >> 2) * 128 + (i & 3) * 32 + j] = gr[q];   // image[k = i][column j]
          }
        }
      }
    }
    u32x4 ac[ACC ? NB * 2 : 1];
    if constexpr (ACC) {
#pragma unroll
      for (int q = 0; q < NB * 2; ++q) ac[q] = ar[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // wave-private LDS: operations of a wave are in order
    if (b + waves < p.batch) issue(b + waves);
    // ---- C[i][d] = sum_j Gs[j][i] X[j][d] ----
    u32x4 fg[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x2 lo = dot_read_tr16(gbase + ks * 1024 + frag_g), hi = dot_read_tr16(gbase + ks * 1024 + frag_g + 256);
      fg[ks] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nb < nblk) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint32_t a = xbase + (nb >> 2) * 8192 + (nb & 3) * 256 + ks * 4096 + frag_x;
          const u32x2 lo = dot_read_tr16(a), hi = dot_read_tr16(a + 1024);
          const u32x4 fx = u32x4{lo.x, lo.y, hi.x, hi.y};
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fg[ks]), __builtin_bit_cast(bf16x8, fx),
                                                       acc, 0, 0, 0);
        }
        // accumulator: C[i][d], d = lane & 31, i = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SST + (lane & 31)] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int it = t * 64 + lane;
          if (it < items) {
            const int i = it >> 2, c8 = (it & 3) * 8;
            const float4 v0 = *reinterpret_cast<const float4*>(stage + i * SST + c8);
            const float4 v1 = *reinterpret_cast<const float4*>(stage + i * SST + c8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (ACC) {
              if ((p.acc_mask >> i) & 1u) {
                const u32x4 e = ac[nb * 2 + t];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  v[2 * q] += __uint_as_float(e[q] << 16);
                  v[2 * q + 1] += __uint_as_float(e[q] & 0xffff0000u);
                }
              }
            }
            const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            const u32x4 e = tab[32 + i];
            const uint64_t dst = (((uint64_t)e[1] << 32) | e[0]) + (uint64_t)b * e[2] + (uint32_t)(nb * 32 + c8) * 2u;
            *reinterpret_cast<u32x4 __attribute__((address_space(1)))*>(dst) = o;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }
}

// ---- plain kernels (any D / alignment, F <= 64); the pointer tables travel as kernel arguments ----
constexpr int kMaxGeneric = 64;
struct DotGenericParams {
  const void* feat[kMaxGeneric];
  int64_t ld[kMaxGeneric];
  void* gfeat[kMaxGeneric];
  int64_t gld[kMaxGeneric];
  int n_feats;
  int64_t batch;
  int dim;
  int self_inter;
  int skip_gather;
  void* out;
  int64_t out_ld;
  int dtype;
  uint64_t acc_mask;
};

__global__ __launch_bounds__(256) void dot_fwd_generic_kernel(const DotGenericParams p) {
  const int F = p.n_feats;
  const int64_t total = p.batch * F * F;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t b = idx / (F * F);
    const int e = (int)(idx - b * F * F);
    const int i = e / F, j = e - i * F;
    const bool keep = pair_kept(i, j, p.self_inter);
    if (!keep && !p.skip_gather) continue;
    float acc = 0.0f;
    if (keep)
      for (int c = 0; c < p.dim; ++c)
        acc = fmaf(ld_elem(p.feat[i], p.dtype, b * p.ld[i] + c), ld_elem(p.feat[j], p.dtype, b * p.ld[j] + c), acc);
    st_elem(p.out, p.dtype, b * p.out_ld + pair_col(i, j, F, p.self_inter, p.skip_gather), acc);
  }
}

__global__ __launch_bounds__(256) void dot_bwd_generic_kernel(const DotGenericParams p) {
  const int F = p.n_feats;
  const int64_t total = p.batch * F * p.dim;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t b = idx / ((int64_t)F * p.dim);
    const int e = (int)(idx - b * F * p.dim);
    const int i = e / p.dim, c = e - i * p.dim;
    float acc = 0.0f;
    for (int j = 0; j < F; ++j) {
      float gs = 0.0f;
      if (pair_kept(i, j, p.self_inter))
        gs += ld_elem(p.out, p.dtype, b * p.out_ld + pair_col(i, j, F, p.self_inter, p.skip_gather));
      if (pair_kept(j, i, p.self_inter))
        gs += ld_elem(p.out, p.dtype, b * p.out_ld + pair_col(j, i, F, p.self_inter, p.skip_gather));
      acc = fmaf(gs, ld_elem(p.feat[j], p.dtype, b * p.ld[j] + c), acc);
    }
    if ((p.acc_mask >> i) & 1ull) acc += ld_elem(p.gfeat[i], p.dtype, b * p.gld[i] + c);
    st_elem(p.gfeat[i], p.dtype, b * p.gld[i] + c, acc);
  }
}

// ---- more than 64 features: the plain kernels over BLOCKS of features ----------------------------------------
// The reference has no limit on the number of features (dot_interaction.py:134-205); pointer tables of any length do
// not fit kernel arguments, so F > 64 runs one launch per (block of 32 features i, chunk of 128 features j): the
// forward fills the pairs (i, j) of the output; the backward adds the chunk's share of dX_i = sum_j (G_ij + G_ji) X_j
// to the gradients of the block -- the first chunk stores (or adds to the existing value of an accumulated feature),
// later chunks add to what the buffer holds; launches of one stream run in order, so j ascends as in the one-launch
// kernel.  Up to 128 features that is ONE chunk: fp32 sum, one rounding, exactly the plain kernel's arithmetic;
// beyond, a bf16 gradient is rounded once per chunk of 128 (fp32 gradients are exact sums either way).
constexpr int kBlockI = 32, kBlockJ = 128;
struct DotBlockParams {
  const void* fi[kBlockI];
  int64_t ldi[kBlockI];
  void* gi[kBlockI];
  int64_t gldi[kBlockI];
  const void* fj[kBlockJ];
  int64_t ldj[kBlockJ];
  int i0, ni, j0, nj, n_feats;
  int64_t batch;
  int dim, self_inter, skip_gather, dtype;
  void* out;
  int64_t out_ld;
  uint32_t acc_bits;     // backward: features of the i block whose gradient buffer already holds a value to add to
  int first_j;           // backward: the first chunk of the row of launches
};

__global__ __launch_bounds__(256) void dot_fwd_block_kernel(const DotBlockParams p) {
  const int64_t per = (int64_t)p.ni * p.nj;
  const int64_t total = p.batch * per;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t b = idx / per;
    const int e = (int)(idx - b * per);
    const int ii = e / p.nj, jj = e - ii * p.nj;
    const int i = p.i0 + ii, j = p.j0 + jj;
    const bool keep = pair_kept(i, j, p.self_inter);
    if (!keep && !p.skip_gather) continue;
    float acc = 0.0f;
    if (keep)
      for (int c = 0; c < p.dim; ++c)
        acc = fmaf(ld_elem(p.fi[ii], p.dtype, b * p.ldi[ii] + c), ld_elem(p.fj[jj], p.dtype, b * p.ldj[jj] + c), acc);
    st_elem(p.out, p.dtype, b * p.out_ld + pair_col(i, j, p.n_feats, p.self_inter, p.skip_gather), acc);
  }
}

__global__ __launch_bounds__(256) void dot_bwd_block_kernel(const DotBlockParams p) {
  const int64_t per = (int64_t)p.ni * p.dim;
  const int64_t total = p.batch * per;
  const int F = p.n_feats;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t b = idx / per;
    const int e = (int)(idx - b * per);
    const int ii = e / p.dim, c = e - ii * p.dim;
    const int i = p.i0 + ii;
    float acc = 0.0f;
    for (int jj = 0; jj < p.nj; ++jj) {
      const int j = p.j0 + jj;
      float gs = 0.0f;
      if (pair_kept(i, j, p.self_inter))
        gs += ld_elem(p.out, p.dtype, b * p.out_ld + pair_col(i, j, F, p.self_inter, p.skip_gather));
      if (pair_kept(j, i, p.self_inter))
        gs += ld_elem(p.out, p.dtype, b * p.out_ld + pair_col(j, i, F, p.self_inter, p.skip_gather));
      acc = fmaf(gs, ld_elem(p.fj[jj], p.dtype, b * p.ldj[jj] + c), acc);
    }
    if (!p.first_j || ((p.acc_bits >> ii) & 1u)) acc += ld_elem(p.gi[ii], p.dtype, b * p.gldi[ii] + c);
    st_elem(p.gi[ii], p.dtype, b * p.gldi[ii] + c, acc);
  }
}

// host side of the two: the launches of one call
int run_dot_blocks(bool backward, const void* const* feats, const int64_t* ld, int n_feats, int64_t batch, int dim,
                   int dtype, int self_interaction, int skip_gather, void* out, int64_t out_ld, void* const* grad_feats,
                   const int64_t* grad_feat_ld, uint64_t accumulate_mask, hipStream_t st) {
  for (int i0 = 0; i0 < n_feats; i0 += kBlockI) {
    DotBlockParams p{};
    p.i0 = i0; p.ni = std::min(kBlockI, n_feats - i0); p.n_feats = n_feats; p.batch = batch; p.dim = dim;
    p.self_inter = self_interaction != 0; p.skip_gather = skip_gather != 0; p.dtype = dtype; p.out = out; p.out_ld = out_ld;
    for (int k = 0; k < p.ni; ++k) {
      p.fi[k] = feats[i0 + k]; p.ldi[k] = ld[i0 + k];
      if (backward) { p.gi[k] = grad_feats[i0 + k]; p.gldi[k] = grad_feat_ld[i0 + k]; }
      if (i0 + k < 64 && ((accumulate_mask >> (i0 + k)) & 1ull)) p.acc_bits |= 1u << k;
    }
    // forward without skip_gather: pairs with j > i are not part of the output
    const int j_end = (!backward && !skip_gather) ? std::min(n_feats, i0 + p.ni) : n_feats;
    for (int j0 = 0; j0 < j_end; j0 += kBlockJ) {
      p.j0 = j0; p.nj = std::min(kBlockJ, j_end - j0); p.first_j = j0 == 0;
      for (int k = 0; k < p.nj; ++k) { p.fj[k] = feats[j0 + k]; p.ldj[k] = ld[j0 + k]; }
      const int64_t work = backward ? batch * p.ni * dim : batch * p.ni * p.nj;
      const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(work, 256), 65536);
      if (backward) hipLaunchKernelGGL(dot_bwd_block_kernel, dim3(blocks), dim3(256), 0, st, p);
      else hipLaunchKernelGGL(dot_fwd_block_kernel, dim3(blocks), dim3(256), 0, st, p);
    }
  }
  KRS_CHECK_LAUNCH("dot_interaction block kernels");
  return KRS_OK;
}

bool fast_ok(const void* const* feats, const int64_t* ld, int n_feats, int dim, int es) {
  if (n_feats > kMaxFast) return false;
  const int ve = 16 / es;
  if (dim % ve) return false;
  for (int f = 0; f < n_feats; ++f) {
    if (reinterpret_cast<uintptr_t>(feats[f]) & 15) return false;
    if (ld[f] % ve) return false;
  }
  return true;
}

int check_args(const void* const* feats, const int64_t* ld, int n_feats, int64_t batch, int dim, int dtype,
               const void* out) {
  KRS_REQUIRE(feats && ld && out, "dot_interaction: null argument");
  KRS_REQUIRE(n_feats > 0 && batch >= 0 && dim > 0, "dot_interaction: bad sizes");
  KRS_REQUIRE(dtype == KRS_F32 || dtype == KRS_BF16, "dot_interaction: bad dtype");
  for (int f = 0; f < n_feats; ++f) KRS_REQUIRE(feats[f], "dot_interaction: null feature pointer");
  return KRS_OK;
}

}  // namespace
}  // namespace krs

using namespace krs;

extern "C" int krs_dot_interaction_fwd(const void* const* feats, const int64_t* ld, int n_feats, int64_t batch,
                                       int dim, int dtype, int self_interaction, int skip_gather, void* out,
                                       int64_t out_ld, void* stream) {
  if (int rc = check_args(feats, ld, n_feats, batch, dim, dtype, out)) return rc;
  if (batch == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int es = dtype == KRS_BF16 ? 2 : 4;
  if (fast_ok(feats, ld, n_feats, dim, es)) {
    DotParams p{};
    for (int f = 0; f < n_feats; ++f) { p.feat[f] = feats[f]; p.ld[f] = ld[f]; }
    p.n_feats = n_feats; p.batch = batch; p.dim = dim; p.self_inter = self_interaction != 0;
    p.skip_gather = skip_gather != 0; p.out = out; p.out_ld = out_ld;
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(batch, 4), 256 * 8);
    if (es == 2 && dim == 128) hipLaunchKernelGGL((dot_fwd_mfma_kernel<2, 8>), dim3(blocks), dim3(256), 0, st, p);
    else if (es == 2 && dim == 64) hipLaunchKernelGGL((dot_fwd_mfma_kernel<2, 4>), dim3(blocks), dim3(256), 0, st, p);
    else if (es == 2) hipLaunchKernelGGL((dot_fwd_mfma_kernel<2, 0>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dot_fwd_mfma_kernel<4, 0>), dim3(blocks), dim3(256), 0, st, p);
    KRS_CHECK_LAUNCH("dot_fwd_mfma_kernel");
    return KRS_OK;
  }
  if (n_feats > kMaxGeneric)
    return run_dot_blocks(false, feats, ld, n_feats, batch, dim, dtype, self_interaction, skip_gather, out, out_ld,
                          nullptr, nullptr, 0, st);
  DotGenericParams g{};
  for (int f = 0; f < n_feats; ++f) { g.feat[f] = feats[f]; g.ld[f] = ld[f]; }
  g.n_feats = n_feats; g.batch = batch; g.dim = dim; g.self_inter = self_interaction != 0;
  g.skip_gather = skip_gather != 0; g.out = out; g.out_ld = out_ld; g.dtype = dtype;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(batch * n_feats * n_feats, 256), 65536);
  hipLaunchKernelGGL(dot_fwd_generic_kernel, dim3(blocks), dim3(256), 0, st, g);
  KRS_CHECK_LAUNCH("dot_fwd_generic_kernel");
  return KRS_OK;
}

extern "C" int krs_dot_interaction_bwd(const void* const* feats, const int64_t* ld, int n_feats, int64_t batch,
                                       int dim, int dtype, int self_interaction, int skip_gather,
                                       const void* grad_out, int64_t grad_ld, void* const* grad_feats,
                                       const int64_t* grad_feat_ld, void* stream) {
  return krs_dot_interaction_bwd_accumulate(feats, ld, n_feats, batch, dim, dtype, self_interaction, skip_gather,
                                            grad_out, grad_ld, grad_feats, grad_feat_ld, 0, stream);
}

extern "C" int krs_dot_interaction_bwd_accumulate(const void* const* feats, const int64_t* ld, int n_feats,
                                                  int64_t batch, int dim, int dtype, int self_interaction,
                                                  int skip_gather, const void* grad_out, int64_t grad_ld,
                                                  void* const* grad_feats, const int64_t* grad_feat_ld,
                                                  uint64_t accumulate_mask, void* stream) {
  if (int rc = check_args(feats, ld, n_feats, batch, dim, dtype, grad_out)) return rc;
  if (n_feats < 64) accumulate_mask &= (uint64_t(1) << n_feats) - 1;
  KRS_REQUIRE(grad_feats && grad_feat_ld, "dot_interaction_bwd: null gradient outputs");
  if (batch == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int es = dtype == KRS_BF16 ? 2 : 4;
  // pairs of adjacent columns per lane: rows 2*es-byte aligned, even D; 32-bit byte offsets
  bool fast = n_feats <= kMaxFast && dim % 2 == 0;
  for (int f = 0; fast && f < n_feats; ++f)
    fast = !(reinterpret_cast<uintptr_t>(feats[f]) % (2 * es)) && !(reinterpret_cast<uintptr_t>(grad_feats[f]) % (2 * es)) &&
           ld[f] % 2 == 0 && grad_feat_ld[f] % 2 == 0 && ld[f] > 0 && grad_feat_ld[f] > 0 &&
           (batch * ld[f] + dim) * es < (int64_t(1) << 31) && (batch * grad_feat_ld[f] + dim) * es < (int64_t(1) << 31);
  // matrix-core path: bf16, rows of whole 32-column blocks (one 16-lane group loads a row: D <= 128), 16-byte aligned rows
  bool mfma_ok = es == 2 && n_feats <= kMaxFast && dim % 32 == 0 && dim <= 128 && !getenv("KRS_DOT_BWD_VALU") &&
                 (skip_gather || tri_cols(n_feats, self_interaction != 0) > 0);
  for (int f = 0; mfma_ok && f < n_feats; ++f)
    mfma_ok = !(reinterpret_cast<uintptr_t>(feats[f]) % 16) && !(reinterpret_cast<uintptr_t>(grad_feats[f]) % 16) &&
              ld[f] % 8 == 0 && grad_feat_ld[f] % 8 == 0 && ld[f] > 0 && grad_feat_ld[f] > 0 &&
              ld[f] * 2 < (int64_t(1) << 32) && grad_feat_ld[f] * 2 < (int64_t(1) << 32);
  if (mfma_ok) {
    DotParams p{};
    for (int f = 0; f < kMaxFast; ++f) {
      const int q = std::min(f, n_feats - 1);
      p.feat[f] = feats[q]; p.ld[f] = ld[q]; p.gfeat[f] = grad_feats[q]; p.gld[f] = grad_feat_ld[q];
    }
    p.n_feats = n_feats; p.batch = batch; p.dim = dim; p.self_inter = self_interaction != 0;
    p.skip_gather = skip_gather != 0; p.out = const_cast<void*>(grad_out); p.out_ld = grad_ld;
    p.acc_mask = (uint32_t)accumulate_mask;
    const int x_bytes = (dim + 127) / 128 * 8192;
    const size_t lds = 64 * 16 + (size_t)2 * (x_bytes + 2048 + 32 * 36 * 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(batch, 2), 256 * 8);
    const bool self = self_interaction != 0, acc = accumulate_mask != 0;
#define KRS_DOT_MFMA(SELF, ACC, NG, NB) \
  hipLaunchKernelGGL((dot_bwd_mfma_kernel<SELF, ACC, NG, NB>), dim3(blocks), dim3(128), lds, st, p, x_bytes)
#define KRS_DOT_MFMA_NB(SELF, ACC, NG) \
  { KRS_DOT_MFMA(SELF, ACC, NG, 4); }
#define KRS_DOT_MFMA_NG(SELF, ACC) \
  { if (skip_gather) KRS_DOT_MFMA_NB(SELF, ACC, 16) else KRS_DOT_MFMA_NB(SELF, ACC, 9) }
    if (self) { if (acc) KRS_DOT_MFMA_NG(true, true) else KRS_DOT_MFMA_NG(true, false) }
    else { if (acc) KRS_DOT_MFMA_NG(false, true) else KRS_DOT_MFMA_NG(false, false) }
#undef KRS_DOT_MFMA_NG
#undef KRS_DOT_MFMA_NB
#undef KRS_DOT_MFMA
    KRS_CHECK_LAUNCH("dot_bwd_mfma_kernel");
    return KRS_OK;
  }
  if (fast) {
    DotParams p{};
    for (int f = 0; f < kMaxFast; ++f) {
      const int q = std::min(f, n_feats - 1);  // slots past F repeat the last feature: loads stay valid
      p.feat[f] = feats[q]; p.ld[f] = ld[q]; p.gfeat[f] = grad_feats[q]; p.gld[f] = grad_feat_ld[q];
    }
    p.n_feats = n_feats; p.batch = batch; p.dim = dim; p.self_inter = self_interaction != 0;
    p.skip_gather = skip_gather != 0; p.out = const_cast<void*>(grad_out); p.out_ld = grad_ld;
    p.acc_mask = (uint32_t)accumulate_mask;
    const int ncols = skip_gather ? n_feats * n_feats : tri_cols(n_feats, self_interaction != 0);
    int lps_log2 = 0;
    while ((2 << lps_log2) < dim && lps_log2 < 6) ++lps_log2;     // lanes per sample: dim/2 up to 64
    const int spw = std::min(64 >> lps_log2, std::max(1, 1024 / std::max(ncols, 1)));
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(batch, spw), 256 * 64);
    // one sample per wave: rows rounded up to 28 (the 26 + 1 features of DLRM) or 32; several
    // samples per wave (small D) and the [F, F] layout always take the 32-row build
    const bool multi = spw > 1 || ncols > 64 * ((tri_cols(kMaxFast, true) + 63) / 64);
    const bool self = self_interaction != 0;
#define KRS_DOT_BWD(ES)                                                                        \
  if (multi) { if (self) launch_dot_bwd<ES, 32, true, true>(p, lps_log2, spw, blocks, st);      \
               else launch_dot_bwd<ES, 32, false, true>(p, lps_log2, spw, blocks, st); }        \
  else if (n_feats <= 28 && !skip_gather) {                                                     \
               if (self) launch_dot_bwd<ES, 28, true, false>(p, lps_log2, spw, blocks, st);     \
               else launch_dot_bwd<ES, 28, false, false>(p, lps_log2, spw, blocks, st); }       \
  else {       if (self) launch_dot_bwd<ES, 32, true, false>(p, lps_log2, spw, blocks, st);     \
               else launch_dot_bwd<ES, 32, false, false>(p, lps_log2, spw, blocks, st); }
    if (es == 2) { KRS_DOT_BWD(2) } else { KRS_DOT_BWD(4) }
#undef KRS_DOT_BWD
    KRS_CHECK_LAUNCH("dot_bwd_valu_kernel");
    return KRS_OK;
  }
  if (n_feats > kMaxGeneric)
    return run_dot_blocks(true, feats, ld, n_feats, batch, dim, dtype, self_interaction, skip_gather,
                          const_cast<void*>(grad_out), grad_ld, grad_feats, grad_feat_ld, accumulate_mask, st);
  DotGenericParams g{};
  for (int f = 0; f < n_feats; ++f) {
    g.feat[f] = feats[f]; g.ld[f] = ld[f]; g.gfeat[f] = grad_feats[f]; g.gld[f] = grad_feat_ld[f];
  }
  g.n_feats = n_feats; g.batch = batch; g.dim = dim; g.self_inter = self_interaction != 0;
  g.skip_gather = skip_gather != 0; g.out = const_cast<void*>(grad_out); g.out_ld = grad_ld; g.dtype = dtype;
  g.acc_mask = accumulate_mask;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(batch * n_feats * dim, 256), 65536);
  hipLaunchKernelGGL(dot_bwd_generic_kernel, dim3(blocks), dim3(256), 0, st, g);
  KRS_CHECK_LAUNCH("dot_bwd_generic_kernel");
  return KRS_OK;
}
