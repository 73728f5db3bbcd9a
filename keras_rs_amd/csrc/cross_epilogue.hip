// K3, elementwise half -- the cross epilogue for the host-composed path (arbitrary pre_activation callables) and the
// backward's elementwise pass (dz, dL/dx0, bias gradient; feature_cross.py:182-194 and its autodiff), plus the two-stage
// deterministic column sums every bias gradient of the library goes through.  (Part of feature_cross.hip until round 5.)
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

#include "krs_dense_common.h"

namespace krs {
namespace {

// ---- elementwise kernels ------------------------------------------------------
struct CrossParams {
  const void* g; const void* u; const void* x0; const void* x;
  void* y; void* du; void* dx0; void* dxd; float* dbias;
  float* partial;   // [row groups][n] partial column sums (workspace) -> colsum_finish_kernel; null: fp32 atomics on dbias
  int dx0_acc;
  int64_t m, n, ld;
  float diag;
  int act;
  int dtype;
};


template <typename T, int V>
__global__ __launch_bounds__(256) void cross_fwd_vec_kernel(const CrossParams p) {
  const int64_t nv = p.n / V;
  const int64_t total = p.m * nv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t i = idx / nv, o = i * p.ld + (idx - i * nv) * V;
    float u[V], x0[V], x[V], y[V];
    RowVec<T, V>::load(p.u, o, u);
    RowVec<T, V>::load(p.x0, o, x0);
    RowVec<T, V>::load(p.x, o, x);
#pragma unroll
    for (int k = 0; k < V; ++k) y[k] = x0[k] * (u[k] + p.diag * x[k]) + x[k];
    RowVec<T, V>::store(p.y, o, y);
  }
}

__global__ __launch_bounds__(256) void cross_fwd_scalar_kernel(const CrossParams p) {
  const int64_t total = p.m * p.n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t i = idx / p.n, o = i * p.ld + (idx - i * p.n);
    const float xv = ld_elem(p.x, p.dtype, o);
    st_elem(p.y, p.dtype, o, ld_elem(p.x0, p.dtype, o) * (ld_elem(p.u, p.dtype, o) + p.diag * xv) + xv);
  }
}

// Backward: grid = (column strips of 64*V, groups of four row chunks).  Each thread owns V columns
// and walks the rows of its wave's chunk, so the bias gradient is a per-thread register sum; the four
// waves of a workgroup add theirs in LDS and issue one lane-contiguous atomic per column (with one
// atomic per thread and chunk, ~600 chunks queued on the same cache lines of dbias and the kernel took
// 156 us for 8192 rows where the rows themselves need 60).
template <typename T, int V, bool ACC>  // ACC: dx0 already holds the terms of the layers above (dx0_accumulate)
__global__ __launch_bounds__(256) void cross_bwd_vec_kernel(const CrossParams p, int rows_per_block) {
  __shared__ float red[4][64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = ((int64_t)blockIdx.x * 64 + lane) * V < p.n;
  const int64_t col = live ? ((int64_t)blockIdx.x * 64 + lane) * V : 0;  // idle lanes prefetch column 0, store nothing
  const int64_t r0 = ((int64_t)blockIdx.y * 4 + wave) * rows_per_block;
  const int64_t r1 = min(p.m, r0 + rows_per_block);
  const int64_t rend = live ? r1 : r0;
  const bool fold = p.dxd == p.dx0;  // x is x0: the direct term lands in dx0 as well
  float db[V];
#pragma unroll
  for (int k = 0; k < V; ++k) db[k] = 0.0f;
  // two rows of g / x0 / u are kept in flight ahead of the row being processed (clamped row index:
  // the loads are unconditional)
  typedef typename RowVec<T, V>::raw_t raw_t;
  constexpr int AHEAD = 2;      // (3 and 4 rows in flight measured equal, profiles/r4y_cross_bwd_ahead.txt)
  raw_t rg[AHEAD], rx0[AHEAD], ru[AHEAD], racc[AHEAD];
  const void* usrc = p.u ? p.u : p.g;  // without u the value is ignored below
  // the running dL/dx0 of the layers above is read ahead like the other streams (a template parameter, not a
  // branch: a load behind a run-time condition makes hipcc drain the load queue, 454 -> 550 us)
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) {
    const int64_t oa = min(r0 + a, r1 - 1) * p.ld + col;
    rg[a] = RowVec<T, V>::load_raw(p.g, oa);
    rx0[a] = RowVec<T, V>::load_raw(p.x0, oa);
    ru[a] = RowVec<T, V>::load_raw(usrc, oa);
    if constexpr (ACC) racc[a] = RowVec<T, V>::load_raw(p.dx0, oa);
    else racc[a] = rg[a];
  }
  for (int64_t i = r0; i < rend; ++i) {
    const int64_t o = i * p.ld + col;
    float g[V], u[V], x0[V], x[V], gx0[V], dz[V], t[V];
    RowVec<T, V>::unpack(rg[0], g);
    RowVec<T, V>::unpack(rx0[0], x0);
    RowVec<T, V>::unpack(ru[0], u);
    RowVec<T, V>::unpack(racc[0], t);
    if (!p.u) {
#pragma unroll
      for (int k = 0; k < V; ++k) u[k] = 0.0f;
    }
#pragma unroll
    for (int a = 0; a + 1 < AHEAD; ++a) { rg[a] = rg[a + 1]; rx0[a] = rx0[a + 1]; ru[a] = ru[a + 1]; racc[a] = racc[a + 1]; }
    {
      const int64_t on = min(i + AHEAD, r1 - 1) * p.ld + col;
      rg[AHEAD - 1] = RowVec<T, V>::load_raw(p.g, on);
      rx0[AHEAD - 1] = RowVec<T, V>::load_raw(p.x0, on);
      ru[AHEAD - 1] = RowVec<T, V>::load_raw(usrc, on);
      if constexpr (ACC) racc[AHEAD - 1] = RowVec<T, V>::load_raw(p.dx0, on);
      else racc[AHEAD - 1] = rg[AHEAD - 1];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      gx0[k] = g[k] * x0[k];
      dz[k] = gx0[k] * act_grad_from_output(p.act, u[k]);
      db[k] += dz[k];
    }
    if (p.du) RowVec<T, V>::store(p.du, o, dz);
    if (p.dx0) {
      if (p.diag != 0.0f) {  // x only enters through diag_scale: do not read 2 bytes per element for a zero
        RowVec<T, V>::load(p.x, o, x);
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = 0.0f;
      }
#pragma unroll
      for (int k = 0; k < V; ++k) {
        t[k] = __builtin_fmaf(g[k], __builtin_fmaf(p.diag, x[k], u[k]), ACC ? t[k] : 0.0f);
        if (fold) t[k] += g[k] + p.diag * gx0[k];
      }
      RowVec<T, V>::store(p.dx0, o, t);
    }
    if (p.dxd && !fold) {
#pragma unroll
      for (int k = 0; k < V; ++k) t[k] = g[k] + p.diag * gx0[k];
      RowVec<T, V>::store(p.dxd, o, t);
    }
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

__global__ __launch_bounds__(64) void cross_bwd_scalar_kernel(const CrossParams p, int rows_per_block) {
  const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (col >= p.n) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(p.m, r0 + rows_per_block);
  float db = 0.0f;
  for (int64_t i = r0; i < r1; ++i) {
    const int64_t o = i * p.ld + col;
    const float g = ld_elem(p.g, p.dtype, o);
    const float gx0 = g * ld_elem(p.x0, p.dtype, o);
    const float uv = p.u ? ld_elem(p.u, p.dtype, o) : 0.0f;
    const float dz = gx0 * act_grad_from_output(p.act, uv);
    db += dz;
    if (p.du) st_elem(p.du, p.dtype, o, dz);
    const bool fold = p.dxd == p.dx0;
    if (p.dx0) {
      const float uf = uv + p.diag * ld_elem(p.x, p.dtype, o);
      float t = (p.dx0_acc ? ld_elem(p.dx0, p.dtype, o) : 0.0f) + g * uf;
      if (fold) t += g + p.diag * gx0;
      st_elem(p.dx0, p.dtype, o, t);
    }
    if (p.dxd && !fold) st_elem(p.dxd, p.dtype, o, g + p.diag * gx0);
  }
  if (p.dbias) {
    if (p.partial) p.partial[(int64_t)blockIdx.y * p.n + col] = db;
    else atomicAdd(p.dbias + col, db);
  }
}

__global__ __launch_bounds__(64) void colsum_kernel(const void* a, int64_t lda, int64_t m, int64_t n, int dtype,
                                                    float* out, float* partial, int rows_per_block) {
  const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (col >= n) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(m, r0 + rows_per_block);
  float s = 0.0f;
  for (int64_t i = r0; i < r1; ++i) s += ld_elem(a, dtype, i * lda + col);
  if (partial) partial[(int64_t)blockIdx.y * n + col] = s;
  else atomicAdd(out + col, s);
}

// Second half of the deterministic column sums (bias gradients): out[c] = sum over the row groups of partial[g][c] in a
// FIXED order -- the grouping depends on (m, n) alone, so the fp32 result is the same bits on every run (with the atomics
// of the workspace-free form the order of the additions, and the last bits, varied from run to run).  A workgroup owns 32
// columns; its 32 slices (one half wave each) sum contiguous runs of groups in ascending order, four independent loads
// in flight, and the slice sums are added in slice order.  (One thread per column walking all groups -- the first
// version -- took 40-500 us: up to 1024 dependent loads per thread and a handful of workgroups.)
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* partial, int64_t groups, int64_t n, float* out) {
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, s = threadIdx.x >> 5;
  const int64_t col = (int64_t)blockIdx.x * 32 + c;
  float acc = 0.0f;
  if (col < n) {
    const int64_t per = (groups + 31) / 32;
    const int64_t g0 = (int64_t)s * per, g1 = min(groups, g0 + per);
    int64_t g = g0;
    for (; g + 4 <= g1; g += 4) {
      const float a0 = partial[g * n + col], a1 = partial[(g + 1) * n + col];
      const float a2 = partial[(g + 2) * n + col], a3 = partial[(g + 3) * n + col];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; g < g1; ++g) acc += partial[g * n + col];
  }
  red[s][c] = acc;
  __syncthreads();
  if (s == 0 && col < n) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][c];
    out[col] = t;
  }
}

bool vec_ok(const CrossParams& p, int v, std::initializer_list<const void*> ptrs) {
  if (p.n % v || p.ld % v) return false;
  for (const void* q : ptrs)
    if (q && (reinterpret_cast<uintptr_t>(q) & 15)) return false;
  return true;
}

}  // namespace

// row chunks of the column-sum walks: enough to fill the chip, few enough to keep the second stage cheap
ColChunks col_chunks(int64_t m, int64_t cols) {
  const int64_t strips = ceil_div(cols, 64);
  int64_t chunks = ceil_div(4096, strips);
  if (chunks > m) chunks = m;
  ColChunks c;
  c.rows_per_block = (int)ceil_div(m, chunks);
  c.chunks = ceil_div(m, c.rows_per_block);
  c.groups4 = ceil_div(c.chunks, 4);
  return c;
}
// groups of partial sums the launch will write for an [m, n] operand walked V columns per thread
int64_t colsum_groups(int64_t m, int64_t n, int v) {
  if (m <= 0 || n <= 0) return 0;
  const ColChunks c = col_chunks(m, v > 1 ? n / v : n);
  return v > 1 ? c.groups4 : c.chunks;
}
int finish_colsum(float* partial, int64_t groups, int64_t n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)ceil_div(n, 32)), dim3(1024), 0, st, partial, groups, n, out);
  KRS_CHECK_LAUNCH("colsum_finish_kernel");
  return KRS_OK;
}

}  // namespace krs

using namespace krs;

extern "C" int krs_cross_epilogue_fwd(const void* u, const void* x0, const void* x, void* y, int64_t m,
                                      int64_t n, int64_t ld, float diag_scale, int dtype, void* stream) {
  KRS_REQUIRE(u && x0 && x && y, "cross_epilogue_fwd: null operand");
  KRS_REQUIRE(m >= 0 && n >= 0 && ld >= n, "cross_epilogue_fwd: bad sizes");
  if (m == 0 || n == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  CrossParams p{};
  p.u = u; p.x0 = x0; p.x = x; p.y = y; p.m = m; p.n = n; p.ld = ld; p.diag = diag_scale; p.dtype = dtype;
  const int v = dtype == KRS_BF16 ? 8 : 4;
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(m * n / (vec_ok(p, v, {u, x0, x, y}) ? v : 1), 256), 16384);
  if (vec_ok(p, v, {u, x0, x, y})) {
    if (dtype == KRS_BF16)
      hipLaunchKernelGGL((cross_fwd_vec_kernel<uint16_t, 8>), dim3(blocks), dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((cross_fwd_vec_kernel<float, 4>), dim3(blocks), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(cross_fwd_scalar_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
  KRS_CHECK_LAUNCH("cross_fwd_kernel");
  return KRS_OK;
}

extern "C" size_t krs_colsum_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  int64_t g = colsum_groups(m, n, 1);
  if (n % 4 == 0) g = std::max(g, colsum_groups(m, n, 4));
  if (n % 8 == 0) g = std::max(g, colsum_groups(m, n, 8));
  return (size_t)g * (size_t)n * sizeof(float);
}

extern "C" int krs_cross_epilogue_bwd(const void* g, const void* u, const void* x0, const void* x, void* du,
                                      void* dx0, int dx0_accumulate, void* dxd, float* dbias, int64_t m,
                                      int64_t n, int64_t ld, float diag_scale, int act, int dtype,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  KRS_REQUIRE(g && x0, "cross_epilogue_bwd: null g/x0");
  KRS_REQUIRE(act == KRS_ACT_NONE || u, "cross_epilogue_bwd: an activation needs the saved u");
  KRS_REQUIRE(!dx0 || (u && x), "cross_epilogue_bwd: dx0 needs u and x");
  KRS_REQUIRE(m >= 0 && n >= 0 && ld >= n, "cross_epilogue_bwd: bad sizes");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool two_stage = dbias && workspace && m > 0 && n > 0;
  if (two_stage) KRS_REQUIRE(workspace_bytes >= krs_colsum_workspace_bytes(m, n), "cross_epilogue_bwd: workspace too small");
  if (dbias && !two_stage) KRS_HIP(hipMemsetAsync(dbias, 0, (size_t)n * sizeof(float), st));
  if (m == 0 || n == 0) return KRS_OK;
  CrossParams p{};
  p.g = g; p.u = u; p.x0 = x0; p.x = x; p.du = du; p.dx0 = dx0; p.dxd = dxd; p.dbias = dbias;
  p.partial = two_stage ? reinterpret_cast<float*>(workspace) : nullptr;
  p.dx0_acc = dx0_accumulate; p.m = m; p.n = n; p.ld = ld; p.diag = diag_scale; p.act = act; p.dtype = dtype;
  const int v = dtype == KRS_BF16 ? 8 : 4;
  const bool vec = vec_ok(p, v, {g, u, x0, x, du, dx0, dxd});
  const ColChunks cc = col_chunks(m, vec ? n / v : n);
  const int64_t strips = ceil_div(vec ? n / v : n, 64);
  const int rows_per_block = cc.rows_per_block;
  if (vec) {
    const dim3 grid4((unsigned)strips, (unsigned)cc.groups4);  // four chunks per workgroup
    const bool acc = p.dx0 && p.dx0_acc;
    if (dtype == KRS_BF16) {
      if (acc) hipLaunchKernelGGL((cross_bwd_vec_kernel<uint16_t, 8, true>), grid4, dim3(256), 0, st, p, rows_per_block);
      else hipLaunchKernelGGL((cross_bwd_vec_kernel<uint16_t, 8, false>), grid4, dim3(256), 0, st, p, rows_per_block);
    } else {
      if (acc) hipLaunchKernelGGL((cross_bwd_vec_kernel<float, 4, true>), grid4, dim3(256), 0, st, p, rows_per_block);
      else hipLaunchKernelGGL((cross_bwd_vec_kernel<float, 4, false>), grid4, dim3(256), 0, st, p, rows_per_block);
    }
  } else {
    hipLaunchKernelGGL(cross_bwd_scalar_kernel, dim3((unsigned)strips, (unsigned)cc.chunks), dim3(64), 0, st, p, rows_per_block);
  }
  KRS_CHECK_LAUNCH("cross_bwd_kernel");
  if (two_stage) return finish_colsum(p.partial, vec ? cc.groups4 : cc.chunks, n, dbias, st);
  return KRS_OK;
}

extern "C" int krs_colsum(const void* a, int64_t lda, int64_t m, int64_t n, int dtype, float* out,
                          void* workspace, size_t workspace_bytes, void* stream) {
  KRS_REQUIRE(a && out, "colsum: null operand");
  KRS_REQUIRE(m >= 0 && n >= 0, "colsum: bad sizes");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n == 0) return KRS_OK;
  const bool two_stage = workspace && m > 0;
  if (two_stage) KRS_REQUIRE(workspace_bytes >= krs_colsum_workspace_bytes(m, n), "colsum: workspace too small");
  if (!two_stage) KRS_HIP(hipMemsetAsync(out, 0, (size_t)n * sizeof(float), st));
  if (m == 0) return KRS_OK;
  const ColChunks cc = col_chunks(m, n);
  float* partial = two_stage ? reinterpret_cast<float*>(workspace) : nullptr;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(n, 64), (unsigned)cc.chunks), dim3(64), 0, st, a, lda, m, n, dtype,
                     out, partial, cc.rows_per_block);
  KRS_CHECK_LAUNCH("colsum_kernel");
  if (two_stage) return finish_colsum(partial, cc.chunks, n, out, st);
  return KRS_OK;
}
