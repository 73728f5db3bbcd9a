// Dense-layer helpers of the ml_perf step around the cross stack: backward of a Dense layer's
// bias + activation epilogue, the per-step cast (+ K-contiguous copy) of the weights, and the dense weights' Adagrad in one
// launch.  (Part of feature_cross.hip until round 5.)
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "krs_dense_common.h"

using namespace krs;

// Backward of a Dense layer's bias + activation epilogue: dz = g * act'(y) from the saved OUTPUT y, and
// dbias = column sums of dz (fp32, of the unrounded products), in one pass (it was a compare, a cast, a multiply
// and a separate column-sum launch per layer).  Same walk as cross_bwd_vec_kernel: a thread owns V columns and
// walks the rows of its wave's chunk with two rows of loads in flight.
struct DenseBwdParams {
  const void* g;
  const void* y;
  void* dz;
  float* dbias;
  float* partial;   // as CrossParams::partial
  int64_t m, n, ldg, ldy, ldz;
  int act, dtype;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void dense_act_bwd_vec_kernel(const DenseBwdParams p, int rows_per_block) {
  __shared__ float red[4][64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = ((int64_t)blockIdx.x * 64 + lane) * V < p.n;
  const int64_t col = live ? ((int64_t)blockIdx.x * 64 + lane) * V : 0;
  const int64_t r0 = ((int64_t)blockIdx.y * 4 + wave) * rows_per_block;
  const int64_t r1 = min(p.m, r0 + rows_per_block);
  const int64_t rend = live ? r1 : r0;
  float db[V];
#pragma unroll
  for (int k = 0; k < V; ++k) db[k] = 0.0f;
  typedef typename RowVec<T, V>::raw_t raw_t;
  constexpr int AHEAD = 2;
  raw_t rg[AHEAD], ry[AHEAD];
  const void* ysrc = p.y ? p.y : p.g;   // no activation: the value is ignored
  const int64_t ldy = p.y ? p.ldy : p.ldg;
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) {
    const int64_t ra = max(min(r0 + a, r1 - 1), (int64_t)0);
    rg[a] = RowVec<T, V>::load_raw(p.g, ra * p.ldg + col);
    ry[a] = RowVec<T, V>::load_raw(ysrc, ra * ldy + col);
  }
  for (int64_t i = r0; i < rend; ++i) {
    float g[V], y[V], dz[V];
    RowVec<T, V>::unpack(rg[0], g);
    RowVec<T, V>::unpack(ry[0], y);
#pragma unroll
    for (int a = 0; a + 1 < AHEAD; ++a) { rg[a] = rg[a + 1]; ry[a] = ry[a + 1]; }
    {
      const int64_t rn = min(i + AHEAD, r1 - 1);
      rg[AHEAD - 1] = RowVec<T, V>::load_raw(p.g, rn * p.ldg + col);
      ry[AHEAD - 1] = RowVec<T, V>::load_raw(ysrc, rn * ldy + col);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      dz[k] = g[k] * act_grad_from_output(p.act, y[k]);
      db[k] += dz[k];
    }
    if (p.dz) RowVec<T, V>::store(p.dz, i * p.ldz + col, dz);
  }
  if (p.dbias) {
#pragma unroll
    for (int k = 0; k < V; ++k) red[wave][lane * V + k] = db[k];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * V; c += 256) {
      const int64_t cc = (int64_t)blockIdx.x * 64 * V + c;
      const float s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
      if (cc < p.n) {
        if (p.partial) p.partial[(int64_t)blockIdx.y * p.n + cc] = s;
        else atomicAdd(p.dbias + cc, s);
      }
    }
  }
}
__global__ __launch_bounds__(64) void dense_act_bwd_scalar_kernel(const DenseBwdParams p, int rows_per_block) {
  const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (col >= p.n) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(p.m, r0 + rows_per_block);
  float db = 0.0f;
  for (int64_t i = r0; i < r1; ++i) {
    const float yv = p.y ? ld_elem(p.y, p.dtype, i * p.ldy + col) : 0.0f;
    const float dz = ld_elem(p.g, p.dtype, i * p.ldg + col) * act_grad_from_output(p.act, yv);
    db += dz;
    if (p.dz) st_elem(p.dz, p.dtype, i * p.ldz + col, dz);
  }
  if (p.dbias) {
    if (p.partial) p.partial[(int64_t)blockIdx.y * p.n + col] = db;
    else atomicAdd(p.dbias + col, db);
  }
}

extern "C" int krs_dense_act_bwd(const void* g, int64_t ld_g, const void* y, int64_t ld_y, void* dz, int64_t ld_dz,
                                 float* dbias, int64_t m, int64_t n, int act, int dtype, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  KRS_REQUIRE(g && (dz || dbias), "dense_act_bwd: null operand");
  KRS_REQUIRE(act == KRS_ACT_NONE || y, "dense_act_bwd: an activation needs the saved output y");
  KRS_REQUIRE(m >= 0 && n >= 0 && ld_g >= n && (!y || ld_y >= n) && (!dz || ld_dz >= n), "dense_act_bwd: bad sizes");
  KRS_REQUIRE(dtype == KRS_F32 || dtype == KRS_BF16, "dense_act_bwd: dtype must be f32 or bf16");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool two_stage = dbias && workspace && m > 0 && n > 0;
  if (two_stage) KRS_REQUIRE(workspace_bytes >= krs_colsum_workspace_bytes(m, n), "dense_act_bwd: workspace too small");
  if (dbias && !two_stage) KRS_HIP(hipMemsetAsync(dbias, 0, (size_t)n * sizeof(float), st));
  if (m == 0 || n == 0) return KRS_OK;
  DenseBwdParams p{g, y, dz, dbias, two_stage ? reinterpret_cast<float*>(workspace) : nullptr, m, n, ld_g, ld_y, ld_dz, act, dtype};
  const int v = dtype == KRS_BF16 ? 8 : 4;
  bool vec = n % v == 0 && ld_g % v == 0 && (!y || ld_y % v == 0) && (!dz || ld_dz % v == 0);
  for (const void* q : {g, y, (const void*)dz}) vec = vec && reinterpret_cast<uintptr_t>(q) % 16 == 0;
  const ColChunks cc = col_chunks(m, vec ? n / v : n);
  const int64_t strips = ceil_div(vec ? n / v : n, 64);
  if (vec) {
    const dim3 grid4((unsigned)strips, (unsigned)cc.groups4);
    if (dtype == KRS_BF16) hipLaunchKernelGGL((dense_act_bwd_vec_kernel<uint16_t, 8>), grid4, dim3(256), 0, st, p, cc.rows_per_block);
    else hipLaunchKernelGGL((dense_act_bwd_vec_kernel<float, 4>), grid4, dim3(256), 0, st, p, cc.rows_per_block);
  } else {
    hipLaunchKernelGGL(dense_act_bwd_scalar_kernel, dim3((unsigned)strips, (unsigned)cc.chunks), dim3(64), 0, st, p,
                       cc.rows_per_block);
  }
  KRS_CHECK_LAUNCH("dense_act_bwd_kernel");
  if (two_stage) return finish_colsum(p.partial, vec ? cc.groups4 : cc.chunks, n, dbias, st);
  return KRS_OK;
}

// Weight preparation of a Dense / FeatureCross step: dst = cast(src) and dst_t = cast(src)^T in one pass over
// a 64 x 64 tile staged in LDS (padded rows: conflict-free in both directions).  The weights are a few MB, so
// the separate cast + transposed copy of every step were launch-bound (four ~16 us launches per cross layer).
__global__ __launch_bounds__(256) void cast_transpose_kernel(const void* src, int64_t rows, int64_t cols, int64_t lds_,
                                                             int src_dtype, void* dst, int64_t ldd, void* dst_t,
                                                             int64_t ldt, int dst_dtype) {
  __shared__ float tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t r = r0 + ty * 16 + k, c = c0 + tx;
    if (r < rows && c < cols) {
      const float v = ld_elem(src, src_dtype, r * lds_ + c);
      tile[ty * 16 + k][tx] = v;
      if (dst) st_elem(dst, dst_dtype, r * ldd + c, v);
    }
  }
  if (!dst_t) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t c = c0 + ty * 16 + k, r = r0 + tx;   // dst_t[c][r] = src[r][c]
    if (r < rows && c < cols) st_elem(dst_t, dst_dtype, c * ldt + r, tile[tx][ty * 16 + k]);
  }
}

// Several weights in ONE launch (the per-step bf16 copies of every dense kernel, refreshed right behind the optimizer
// step: six launch-bound 15 us kernels per FeatureCross stack become one).  Contiguous sources and outputs.
constexpr int kCastMax = 32;
struct CastManyArgs {
  const void* src[kCastMax];
  void* dst[kCastMax];
  void* dst_t[kCastMax];
  int32_t rows[kCastMax], cols[kCastMax];
  int32_t tile_end[kCastMax];    // inclusive prefix of the tensors' 64 x 64 tile counts
  int count, src_dtype, dst_dtype;
};
__global__ __launch_bounds__(256) void cast_transpose_many_kernel(const CastManyArgs a) {
  __shared__ float tile[64][65];
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.tile_end[t]) ++t;
  const void* src = nullptr; void* dst = nullptr; void* dst_t = nullptr; int64_t rows = 0, cols = 0; int first = 0;
#pragma unroll
  for (int i = 0; i < kCastMax; ++i)   // static kernarg indices
    if (i == t) { src = a.src[i]; dst = a.dst[i]; dst_t = a.dst_t[i]; rows = a.rows[i]; cols = a.cols[i]; first = i ? a.tile_end[i - 1] : 0; }
  const int tiles_x = (int)ceil_div(cols, 64);
  const int tid = (int)blockIdx.x - first;
  const int64_t r0 = (int64_t)(tid / tiles_x) * 64, c0 = (int64_t)(tid % tiles_x) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t r = r0 + ty * 16 + k, c = c0 + tx;
    if (r < rows && c < cols) {
      const float v = ld_elem(src, a.src_dtype, r * cols + c);
      tile[ty * 16 + k][tx] = v;
      if (dst) st_elem(dst, a.dst_dtype, r * cols + c, v);
    }
  }
  if (!dst_t) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t c = c0 + ty * 16 + k, r = r0 + tx;   // dst_t[c][r] = src[r][c]
    if (r < rows && c < cols) st_elem(dst_t, a.dst_dtype, c * rows + r, tile[tx][ty * 16 + k]);
  }
}

extern "C" int krs_cast_transpose_many(int count, const void* const* srcs, const int64_t* rows, const int64_t* cols,
                                       int src_dtype, void* const* dsts, void* const* dst_ts, int dst_dtype, void* stream) {
  KRS_REQUIRE(count >= 0 && (count == 0 || (srcs && rows && cols && dsts && dst_ts)), "cast_transpose_many: null list");
  KRS_REQUIRE((src_dtype == KRS_F32 || src_dtype == KRS_BF16) && (dst_dtype == KRS_F32 || dst_dtype == KRS_BF16),
              "cast_transpose_many: dtype must be f32 or bf16");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int lo = 0; lo < count; lo += kCastMax) {
    CastManyArgs a{};
    a.count = std::min(kCastMax, count - lo);
    a.src_dtype = src_dtype; a.dst_dtype = dst_dtype;
    int tiles = 0;
    for (int i = 0; i < a.count; ++i) {
      KRS_REQUIRE(srcs[lo + i] && (dsts[lo + i] || dst_ts[lo + i]) && rows[lo + i] > 0 && cols[lo + i] > 0 &&
                      rows[lo + i] < 0x7fffffff && cols[lo + i] < 0x7fffffff, "cast_transpose_many: bad tensor %d", lo + i);
      a.src[i] = srcs[lo + i]; a.dst[i] = dsts[lo + i]; a.dst_t[i] = dst_ts[lo + i];
      a.rows[i] = (int32_t)rows[lo + i]; a.cols[i] = (int32_t)cols[lo + i];
      tiles += (int)(ceil_div(rows[lo + i], 64) * ceil_div(cols[lo + i], 64));
      a.tile_end[i] = tiles;
    }
    hipLaunchKernelGGL(cast_transpose_many_kernel, dim3((unsigned)tiles), dim3(256), 0, st, a);
    KRS_CHECK_LAUNCH("cast_transpose_many_kernel");
  }
  return KRS_OK;
}

extern "C" int krs_cast_transpose(const void* src, int64_t rows, int64_t cols, int64_t ld_src, int src_dtype,
                                  void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t, int dst_dtype,
                                  void* stream) {
  KRS_REQUIRE(src && (dst || dst_t), "cast_transpose: null operand");
  KRS_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols, "cast_transpose: bad sizes");
  KRS_REQUIRE((src_dtype == KRS_F32 || src_dtype == KRS_BF16) && (dst_dtype == KRS_F32 || dst_dtype == KRS_BF16),
              "cast_transpose: dtype must be f32 or bf16");
  KRS_REQUIRE((!dst || ld_dst >= cols) && (!dst_t || ld_dst_t >= rows), "cast_transpose: bad output strides");
  if (rows == 0 || cols == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64)), dim3(256),
                     0, st, src, rows, cols, ld_src, src_dtype, dst, ld_dst, dst_t, ld_dst_t, dst_dtype);
  KRS_CHECK_LAUNCH("cast_transpose_kernel");
  return KRS_OK;
}

// Dense Adagrad over a LIST of fp32 tensors in one launch (the FeatureCross / Dense weights of a step):
//   acc += g*g;  p -= lr * g / (sqrt(acc) + eps)      -- torch.optim.Adagrad / keras Adagrad with eps outside the root
// One workgroup per 4096-element chunk of one tensor; the (tensor, chunk) of a workgroup comes from the
// cumulative chunk counts in the argument block (<= 32 tensors per launch).
constexpr int kOptMax = 32, kOptChunk = 4096;
struct DenseOptArgs {
  float* p[kOptMax];
  const float* g[kOptMax];
  float* acc[kOptMax];
  int64_t n[kOptMax];
  int32_t chunk_end[kOptMax];   // inclusive prefix of the tensors' chunk counts
  int count;
  float lr, eps;
};
__global__ __launch_bounds__(256) void dense_adagrad_kernel(const DenseOptArgs a) {
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.chunk_end[t]) ++t;
  float* p = nullptr; const float* g = nullptr; float* acc = nullptr; int64_t n = 0; int first = 0;
#pragma unroll
  for (int i = 0; i < kOptMax; ++i)   // static kernarg indices
    if (i == t) { p = a.p[i]; g = a.g[i]; acc = a.acc[i]; n = a.n[i]; first = i ? a.chunk_end[i - 1] : 0; }
  const int64_t base = (int64_t)(blockIdx.x - first) * kOptChunk;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(acc)) & 15) == 0;
#pragma unroll
  for (int k = 0; k < kOptChunk / 1024; ++k) {
    const int64_t i0 = base + k * 1024 + threadIdx.x * 4;
    if (vec && i0 + 4 <= n) {
      const float4 gv = *reinterpret_cast<const float4*>(g + i0);
      float4 av = *reinterpret_cast<const float4*>(acc + i0), pv = *reinterpret_cast<const float4*>(p + i0);
      av.x = fmaf(gv.x, gv.x, av.x); av.y = fmaf(gv.y, gv.y, av.y); av.z = fmaf(gv.z, gv.z, av.z); av.w = fmaf(gv.w, gv.w, av.w);
      pv.x -= a.lr * gv.x / (sqrtf(av.x) + a.eps); pv.y -= a.lr * gv.y / (sqrtf(av.y) + a.eps);
      pv.z -= a.lr * gv.z / (sqrtf(av.z) + a.eps); pv.w -= a.lr * gv.w / (sqrtf(av.w) + a.eps);
      *reinterpret_cast<float4*>(acc + i0) = av;
      *reinterpret_cast<float4*>(p + i0) = pv;
    } else {
      for (int q = 0; q < 4 && i0 + q < n; ++q) {
        const float gq = g[i0 + q], aq = fmaf(gq, gq, acc[i0 + q]);
        acc[i0 + q] = aq;
        p[i0 + q] -= a.lr * gq / (sqrtf(aq) + a.eps);
      }
    }
  }
}

extern "C" int krs_dense_adagrad(float* const* params, const float* const* grads, float* const* accs,
                                 const int64_t* sizes, int count, float lr, float eps, void* stream) {
  KRS_REQUIRE(count >= 0 && (count == 0 || (params && grads && accs && sizes)), "dense_adagrad: null tensor list");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int lo = 0; lo < count; lo += kOptMax) {
    DenseOptArgs a{};
    a.count = std::min(kOptMax, count - lo);
    a.lr = lr; a.eps = eps;
    int chunks = 0;
    for (int i = 0; i < a.count; ++i) {
      KRS_REQUIRE(sizes[lo + i] >= 0 && (sizes[lo + i] == 0 || (params[lo + i] && grads[lo + i] && accs[lo + i])),
                  "dense_adagrad: null tensor");
      a.p[i] = params[lo + i]; a.g[i] = grads[lo + i]; a.acc[i] = accs[lo + i]; a.n[i] = sizes[lo + i];
      chunks += (int)ceil_div(sizes[lo + i], kOptChunk);
      a.chunk_end[i] = chunks;
    }
    if (chunks == 0) continue;
    hipLaunchKernelGGL(dense_adagrad_kernel, dim3((unsigned)chunks), dim3(256), 0, st, a);
    KRS_CHECK_LAUNCH("dense_adagrad_kernel");
  }
  return KRS_OK;
}
