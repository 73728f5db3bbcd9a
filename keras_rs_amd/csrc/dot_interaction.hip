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
// Backward: dX = (G + G^T) X per sample; G+G^T (32x32) and X are staged in LDS
// as fp32 and contracted with v_mfma_f32_32x32x2_f32.
// Shapes outside the fast path (32 < F <= 64, odd D, unaligned) use plain kernels.
#include <algorithm>

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
};

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

template <int ES>
__global__ __launch_bounds__(256) void dot_fwd_mfma_kernel(const DotParams p) {
  const int lane = threadIdx.x & 63;
  const int f = lane & 31;
  const int half = lane >> 5;
  const int F = p.n_feats;
  const char* base;
  int64_t ld;
  pick_feature(p, f < F ? f : 0, base, ld);
  constexpr int VE = 16 / ES;  // elements per 16-byte piece
  const int ksteps = (p.dim + 2 * VE - 1) / (2 * VE);
  const int64_t waves = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < p.batch; b += waves) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const char* row = base + b * ld * ES;
    for (int ks = 0; ks < ksteps; ++ks) {
      const int k = ks * 2 * VE + half * VE;
      u32x4 v = {0, 0, 0, 0};
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
    // accumulator: P[i][j], j = lane & 31, i = (r & 3) + 8*(r >> 2) + 4*half
    const int j = f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < F && j < F) {
        const bool keep = pair_kept(i, j, p.self_inter);
        if (p.skip_gather)
          st_elem(p.out, ES == 2 ? KRS_BF16 : KRS_F32, b * p.out_ld + (int64_t)i * F + j, keep ? acc[r] : 0.0f);
        else if (keep)
          st_elem(p.out, ES == 2 ? KRS_BF16 : KRS_F32,
                  b * p.out_ld + pair_col(i, j, F, p.self_inter, 0), acc[r]);
      }
    }
  }
}

// Backward fast path: one wave (64-thread workgroup) per sample at a time.
// LDS: X as fp32 [32][dim + 4], Gs = G + G^T as fp32 [32][33].
template <int ES, int NBLK>  // NBLK = ceil(dim / 32) column blocks of the output
__global__ __launch_bounds__(64) void dot_bwd_mfma_kernel(const DotParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int F = p.n_feats;
  const int xs = p.dim + 4;  // padded row stride (floats)
  float* X = reinterpret_cast<float*>(smem);
  float* Gs = X + 32 * xs;
  // per-feature pointer tables, staged once so they can be indexed dynamically
  const char** t_feat = reinterpret_cast<const char**>(Gs + 32 * 33 + 1 + ((32 * xs + 32 * 33 + 1) & 1));
  int64_t* t_ld = reinterpret_cast<int64_t*>(t_feat + 32);
  char** t_gfeat = reinterpret_cast<char**>(t_ld + 32);
  int64_t* t_gld = reinterpret_cast<int64_t*>(t_gfeat + 32);
  float* g_l = reinterpret_cast<float*>(t_gld + 32);  // the sample's gradient row (<= 1024 entries)
  const int n_gcols = p.skip_gather ? F * F : (p.self_inter ? F * (F + 1) / 2 : F * (F - 1) / 2);
  if (lane < 32) {
    const char* fp = nullptr; int64_t fl = 0; char* gp = nullptr; int64_t gl = 0;
#pragma unroll
    for (int q = 0; q < kMaxFast; ++q)
      if (q == lane) {
        fp = reinterpret_cast<const char*>(p.feat[q]); fl = p.ld[q];
        gp = reinterpret_cast<char*>(p.gfeat[q]); gl = p.gld[q];
      }
    t_feat[lane] = fp; t_ld[lane] = fl; t_gfeat[lane] = gp; t_gld[lane] = gl;
  }
  __syncthreads();
  constexpr int VE = 16 / ES;
  const int dt = ES == 2 ? KRS_BF16 : KRS_F32;
  const int pieces = p.dim / VE;  // 16-byte pieces per feature row

  for (int64_t b = blockIdx.x; b < p.batch; b += gridDim.x) {
    // stage X (zero rows for f >= F): loads are issued in batches of 8 before any is consumed
    const int total = 32 * pieces;
    for (int base = 0; base < total; base += 64 * 8) {
      u32x4 raw[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = base + q * 64 + lane;
        const int f = min(idx / pieces, F - 1);            // clamped: always a real row
        const int pc = idx % pieces;
        raw[q] = *reinterpret_cast<const u32x4*>(t_feat[f] + (b * t_ld[f] + (int64_t)pc * VE) * ES);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = base + q * 64 + lane;
        if (idx < total) {
          const int f = idx / pieces, pc = idx % pieces;
          const bool live = f < F;
          float* dst = X + f * xs + pc * VE;
          if constexpr (ES == 2) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              dst[2 * w] = live ? __uint_as_float(raw[q][w] << 16) : 0.0f;
              dst[2 * w + 1] = live ? __uint_as_float(raw[q][w] & 0xffff0000u) : 0.0f;
            }
          } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) dst[w] = live ? __uint_as_float(raw[q][w]) : 0.0f;
          }
        }
      }
    }
    // the sample's gradient row goes to LDS with coalesced loads first ...
    for (int c = lane; c < n_gcols; c += 64) g_l[c] = ld_elem(p.out, dt, b * p.out_ld + c);
    __syncthreads();
    // ... then Gs[i][j] = G[i][j]*kept(i,j) + G[j][i]*kept(j,i) is built from LDS
    for (int e = lane; e < 1024; e += 64) {
      const int i = e >> 5, j = e & 31;
      float g = 0.0f;
      if (i < F && j < F) {
        const int cij = (int)pair_col(i, j, F, p.self_inter, p.skip_gather);
        const int cji = (int)pair_col(j, i, F, p.self_inter, p.skip_gather);
        const float gij = g_l[pair_kept(i, j, p.self_inter) ? cij : 0];
        const float gji = g_l[pair_kept(j, i, p.self_inter) ? cji : 0];
        g = (pair_kept(i, j, p.self_inter) ? gij : 0.0f) + (pair_kept(j, i, p.self_inter) ? gji : 0.0f);
      }
      Gs[i * 33 + j] = g;
    }
    __syncthreads();

    f32x16 acc[NBLK];
#pragma unroll
    for (int n = 0; n < NBLK; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    const int row = lane & 31;
    const int half = lane >> 5;
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int j = 2 * s + half;
      const float a = Gs[row * 33 + j];  // A[i = row][k = j]
#pragma unroll
      for (int n = 0; n < NBLK; ++n) {
        const int d = n * 32 + row;
        const float bv = d < p.dim ? X[j * xs + d] : 0.0f;  // B[k = j][n = d]
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[n], 0, 0, 0);
      }
    }
    // dX[i][d]: d = n*32 + (lane & 31), i = (r & 3) + 8*(r >> 2) + 4*half
#pragma unroll
    for (int n = 0; n < NBLK; ++n) {
      const int d = n * 32 + row;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (i < F && d < p.dim) st_elem(t_gfeat[i], dt, b * t_gld[i] + d, acc[n][r]);
      }
    }
    __syncthreads();
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
    st_elem(p.gfeat[i], p.dtype, b * p.gld[i] + c, acc);
  }
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
    if (es == 2) hipLaunchKernelGGL(dot_fwd_mfma_kernel<2>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(dot_fwd_mfma_kernel<4>, dim3(blocks), dim3(256), 0, st, p);
    KRS_CHECK_LAUNCH("dot_fwd_mfma_kernel");
    return KRS_OK;
  }
  if (n_feats > kMaxGeneric)
    return fail(KRS_ERR_UNSUPPORTED, "dot_interaction: at most %d features (got %d)", kMaxGeneric, n_feats);
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
  if (int rc = check_args(feats, ld, n_feats, batch, dim, dtype, grad_out)) return rc;
  KRS_REQUIRE(grad_feats && grad_feat_ld, "dot_interaction_bwd: null gradient outputs");
  if (batch == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int es = dtype == KRS_BF16 ? 2 : 4;
  bool fast = fast_ok(feats, ld, n_feats, dim, es) && dim <= 256;
  if (fast) {
    DotParams p{};
    for (int f = 0; f < n_feats; ++f) {
      p.feat[f] = feats[f]; p.ld[f] = ld[f]; p.gfeat[f] = grad_feats[f]; p.gld[f] = grad_feat_ld[f];
    }
    p.n_feats = n_feats; p.batch = batch; p.dim = dim; p.self_inter = self_interaction != 0;
    p.skip_gather = skip_gather != 0; p.out = const_cast<void*>(grad_out); p.out_ld = grad_ld;
    const size_t lds = (size_t)(32 * (dim + 4) + 32 * 33 + 2) * sizeof(float) + 32 * 4 * 8 + 1024 * sizeof(float);
    const unsigned blocks = (unsigned)std::min<int64_t>(batch, 256 * 32);
    const int nblk = (dim + 31) / 32;
#define KRS_DOT_BWD(ES, NB)                                                                     \
  hipLaunchKernelGGL((dot_bwd_mfma_kernel<ES, NB>), dim3(blocks), dim3(64), lds, st, p)
#define KRS_DOT_BWD_ES(ES)                                                                      \
  switch (nblk) {                                                                               \
    case 1: KRS_DOT_BWD(ES, 1); break;                                                          \
    case 2: KRS_DOT_BWD(ES, 2); break;                                                          \
    case 3: KRS_DOT_BWD(ES, 3); break;                                                          \
    case 4: KRS_DOT_BWD(ES, 4); break;                                                          \
    case 5: KRS_DOT_BWD(ES, 5); break;                                                          \
    case 6: KRS_DOT_BWD(ES, 6); break;                                                          \
    case 7: KRS_DOT_BWD(ES, 7); break;                                                          \
    default: KRS_DOT_BWD(ES, 8); break;                                                         \
  }
    if (es == 2) { KRS_DOT_BWD_ES(2) } else { KRS_DOT_BWD_ES(4) }
#undef KRS_DOT_BWD_ES
#undef KRS_DOT_BWD
    KRS_CHECK_LAUNCH("dot_bwd_mfma_kernel");
    return KRS_OK;
  }
  if (n_feats > kMaxGeneric)
    return fail(KRS_ERR_UNSUPPORTED, "dot_interaction: at most %d features (got %d)", kMaxGeneric, n_feats);
  DotGenericParams g{};
  for (int f = 0; f < n_feats; ++f) {
    g.feat[f] = feats[f]; g.ld[f] = ld[f]; g.gfeat[f] = grad_feats[f]; g.gld[f] = grad_feat_ld[f];
  }
  g.n_feats = n_feats; g.batch = batch; g.dim = dim; g.self_inter = self_interaction != 0;
  g.skip_gather = skip_gather != 0; g.out = const_cast<void*>(grad_out); g.out_ld = grad_ld; g.dtype = dtype;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(batch * n_feats * dim, 256), 65536);
  hipLaunchKernelGGL(dot_bwd_generic_kernel, dim3(blocks), dim3(256), 0, st, g);
  KRS_CHECK_LAUNCH("dot_bwd_generic_kernel");
  return KRS_OK;
}
