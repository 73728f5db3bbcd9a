// K3 -- FeatureCross (DCN-v2): the dense projection on MFMA with the cross
// epilogue fused, plus the elementwise backward pieces.
//
// Replaces FeatureCross.call (keras_rs/src/layers/feature_interaction/feature_cross.py:182-194):
//   keras Dense (matmul + bias + activation, :134-151) -> cast -> + diag_scale*x
//   -> x0 * u + x, which the reference runs as one GEMM and three separate
//   elementwise passes over [B, d].
//
// krs_gemm: C[M,N] = epilogue(A[M,K] . B[K,N]) for the three operand layouts
// the layer needs (forward, data gradient, weight gradient).  MFMA throughout:
// v_mfma_f32_32x32x16_bf16 (bf16) / v_mfma_f32_32x32x2_f32 (fp32, exact fmaf chain), fp32
// accumulators in registers, a lane's operand = 16 bytes of LDS.  Kernels, by shape:
//   * gemm_pp64_kernel      (round 6) the big K-contiguous bf16 products -- h = x U, dh = dz K^T, the cross / residual /
//                           fused-backward forms -- whenever K is whole 64-k blocks: gemm_pp256_kernel's tile and ping-pong
//                           on 64-k pieces whose rows are whole 128-byte lines, five 32 KB slots, the LDS-DMA instructions
//                           issued among the MFMAs (h 204 -> 189 us, dh 214 -> 193, cross product 455 -> 430; bit-identical);
//   * gemm_pp256_kernel     the weight gradients (K-strided operands) and the K-contiguous shapes gemm_pp64_kernel does not
//                           take: the 256x256 tile (8 waves as
//                           2(M) x 4(N), 128x64 per wave) on a four-stage LDS-DMA ring (global_load_lds, XOR swizzle on
//                           the source side), ping-pong wave groups, counted vmcnt, epilogue operands fetched under the
//                           tail of the main loop; bit-identical to the 128x128 two-stage kernels below, which every
//                           shape takes under pipeline 0 of krs_gemm_set_option.  (The 256x256 two-stage kernels of
//                           round 1, gemm_glds256_kernel / gemm_tn_glds256_kernel, were deleted in round 5: 222-230 ->
//                           204 us and 276 -> 208-220 us against the ring, profiles/archive/r2_gemm_ab.txt.)
//   * gemm_glds_kernel      two 32 KB stages filled by LDS-DMA on a 128x128 tile for smaller M / N (K >= 1024);
//   * gemm_tn_glds_kernel
//                           bf16 weight gradients (both operands K-strided): tiles DMA'd as they
//                           lie in memory, fragments by the transposing ds_read_b64_tr_b16,
//                           split along K into fp32 slabs (one split per XCD at a time) that are
//                           reduced in a fixed order (deterministic, no atomics);
//   * gemm_mfma_kernel      every other aligned shape: global -> register -> LDS staging
//                           (K-strided operands transposed in registers), one 36 KB buffer,
//                           three workgroups per CU;
//   * gemm_thin_kernel      weight gradients with min(M, N) <= 16 (13 dense inputs, 1 unit);
//   * gemm_rowdot_kernel / gemm_smallk_kernel   N <= 8 / K <= 16 with a row-major A (the 1-unit and 13-input Dense layers);
//   * gemm_generic_kernel   anything else (the reference's toy shapes, d = 3).
// Epilogue on the accumulator, staged through LDS so that a lane owns 8 consecutive columns:
// + bias, activation, cross (x0*(v+diag*x)+x), + beta*R, one rounding to the output dtype; the
// cross / residual forms stream x0 / x / R / u / y with non-temporal accesses.
// What bounds these products on MI355X is the L2 -> LDS operand path, not MFMA issue (DESIGN.md
// section 3, scripts/exp/gemm_probe.hip): hence the large tiles.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

#include "krs_dense_common.h"

namespace krs {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;          // K extent of a tile row in bytes (64 bf16 / 32 fp32)
constexpr int LDS_STRIDE = ROW_BYTES + 16;
constexpr int TILE_BYTES = BM * LDS_STRIDE;  // one operand tile (BM == BN)

struct GemmParams {
  const char* a; int64_t lda; int a_km;
  const char* b; int64_t ldb; int b_nk;
  char* c; int64_t ldc;
  int64_t m, n, k;
  int out_dtype;
  krs_gemm_epilogue ep;
  int has_ep;
  // split-K
  int splits; int64_t k_per_split; float* slabs;
  int ep_vec;  // every epilogue operand allows 8-wide vector access
  // fused cross-backward epilogue (EPI 3 .. 8 of gemm_pp256_kernel, krs_gemm_cross_bwd): operands of the layer below
  const void* f_x0; const void* f_u; const void* f_uup; void* f_dz; void* f_dx0; float* f_partial; int64_t f_ld; int f_act; int f_fold;
};

// epilogue of one element; `odt` = dtype of C, x0, x, u_out, r
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int64_t i, int64_t j, float v) {
  const int odt = p.out_dtype;
  if (p.has_ep) {
    const krs_gemm_epilogue& e = p.ep;
    if (e.bias) v += e.bias[j];
    v = apply_act(e.act, v);
    if (e.x0) {
      const float xv = ld_elem(e.x, odt, i * e.ldx + j);
      if (e.u_out) st_elem(e.u_out, odt, i * e.ldu + j, v);
      const float u = v + e.diag_scale * xv;
      v = ld_elem(e.x0, odt, i * e.ldx + j) * u + xv;
    }
    if (e.r) v += e.beta * ld_elem(e.r, odt, i * e.ldr + j);
  }
  st_elem(p.c, odt, i * p.ldc + j, v);
}

// 8 consecutive columns of one row; every pointer 16-byte aligned, every ld a multiple of 8
__device__ __forceinline__ void load8(const void* base, int dt, int64_t off, float (&f)[8]) {
  if (dt == KRS_BF16) {
    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off);
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  } else {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
}
__device__ __forceinline__ void store8(void* base, int dt, int64_t off, const float (&f)[8]) {
  if (dt == KRS_BF16) {
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + off) =
        make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                   pack_bf16x2(f[6], f[7]));
  } else {
    float* d = reinterpret_cast<float*>(base) + off;
    *reinterpret_cast<float4*>(d) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
}

// The streams of the specialised epilogues (x0, x, R in; u, y out) are touched once per launch and
// are two orders of magnitude larger than L2: non-temporal accesses keep them from evicting the
// operand panels the other N tiles of the same M panel are about to re-read.
__device__ __forceinline__ void store8_bf16_nt(void* base, int64_t off, const float (&f)[8]) {
  const u32x4 v = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(base) + off));
}
__device__ __forceinline__ uint4 load8_bf16_nt(const void* base, int64_t off) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + off));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& r, float (&f)[8]) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}

__device__ __forceinline__ void epilogue_store_vec8(const GemmParams& p, int64_t i, int64_t j, float (&v)[8]) {
  const int odt = p.out_dtype;
  if (p.has_ep) {
    const krs_gemm_epilogue& e = p.ep;
    if (e.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(e.bias + j);
      const float4 b1 = *reinterpret_cast<const float4*>(e.bias + j + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
      v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (e.act != KRS_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = apply_act(e.act, v[q]);
    }
    if (e.x0) {
      float xv[8], x0v[8];
      load8(e.x, odt, i * e.ldx + j, xv);
      load8(e.x0, odt, i * e.ldx + j, x0v);
      if (e.u_out) store8(e.u_out, odt, i * e.ldu + j, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = x0v[q] * (v[q] + e.diag_scale * xv[q]) + xv[q];
    }
    if (e.r) {
      float rv[8];
      load8(e.r, odt, i * e.ldr + j, rv);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += e.beta * rv[q];
    }
  }
  store8(p.c, odt, i * p.ldc + j, v);
}

// ---- staging: HBM -> registers -> LDS ([row][k] with 144-byte rows) ---------
// ES = element size (2 | 4).  `row0` tile origin along the operand's row axis
// (m for A, n for B), `k0` along K.  rows/kk are the operand extents.

// operand stored K-contiguous: elem(row, k) at base + (row*ld + k)*ES
template <int ES>
__device__ __forceinline__ void load_kcontig(const char* base, int64_t ld, int64_t row0, int64_t k0,
                                             int64_t rows, int64_t kk, u32x4 (&r)[4]) {
  const int t = threadIdx.x;
  const int c = t & 7;
  const int64_t kel = k0 + c * (16 / ES);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = row0 + (t >> 3) + 32 * i;
    u32x4 v = {0, 0, 0, 0};
    if (row < rows && kel < kk) v = *reinterpret_cast<const u32x4*>(base + (row * ld + kel) * ES);
    r[i] = v;
  }
}
// LDS address of 16-byte chunk `c` (0..7) of tile row `row`.
//   K-contiguous staged operands: 144-byte rows, no swizzle: fragment reads (ds_read_b128) and
//     the row-wise ds_write_b128 stores are both conflict-free;
//   transposed-staged operands: 128-byte rows, chunk XOR (row >> 3): the column-wise
//     ds_write_b64 / b128 stores drop from 16-way to the 2-way minimum of that access shape
//     and fragment reads cost 2 cycles per lane group (scripts/lds_conflicts.py).
template <bool KS>
__device__ __forceinline__ int lds_chunk_off(int row, int c) {
  if constexpr (KS) return row * ROW_BYTES + ((c ^ ((row >> 3) & 7)) << 4);
  else return row * LDS_STRIDE + (c << 4);
}

__device__ __forceinline__ void store_kcontig(char* tile, const u32x4 (&r)[4]) {
  const int t = threadIdx.x;
  const int c = t & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<u32x4*>(tile + lds_chunk_off<false>((t >> 3) + 32 * i, c)) = r[i];
}

// operand stored K-strided: elem(row, k) at base + (k*ld + row)*ES.  A thread
// takes a 4(k) x (16/ES)(row) block: four 16-byte loads along `row`.
template <int ES>
__device__ __forceinline__ void load_kstrided(const char* base, int64_t ld, int64_t row0, int64_t k0,
                                              int64_t rows, int64_t kk, u32x4 (&r)[4]) {
  constexpr int RPB = 16 / ES;          // rows per block: 8 (bf16) | 4 (fp32)
  constexpr int NRB = BM / RPB;         // row blocks per tile: 16 | 32
  const int t = threadIdx.x;
  const int rb = t % NRB;
  const int kb = t / NRB;
  const int64_t row = row0 + rb * RPB;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t kel = k0 + kb * 4 + i;
    u32x4 v = {0, 0, 0, 0};
    if (row < rows && kel < kk) v = *reinterpret_cast<const u32x4*>(base + (kel * ld + row) * ES);
    r[i] = v;
  }
}
template <int ES>
__device__ __forceinline__ void store_kstrided(char* tile, const u32x4 (&r)[4]) {
  constexpr int RPB = 16 / ES;
  constexpr int NRB = BM / RPB;
  const int t = threadIdx.x;
  const int rb = t % NRB;
  const int kb = t / NRB;
  if constexpr (ES == 4) {
    // 4x4 fp32 transpose is pure register renaming
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = jj;
      u32x4 o = {r[0][j], r[1][j], r[2][j], r[3][j]};
      *reinterpret_cast<u32x4*>(tile + lds_chunk_off<true>(rb * 4 + j, kb)) = o;
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = jj;
      const int w = j >> 1;
      uint32_t lo, hi;
      if (j & 1) {
        lo = (r[0][w] >> 16) | (r[1][w] & 0xffff0000u);
        hi = (r[2][w] >> 16) | (r[3][w] & 0xffff0000u);
      } else {
        lo = (r[0][w] & 0xffffu) | (r[1][w] << 16);
        hi = (r[2][w] & 0xffffu) | (r[3][w] << 16);
      }
      *reinterpret_cast<uint2*>(tile + lds_chunk_off<true>(rb * 8 + j, kb >> 1) + (kb & 1) * 8) = make_uint2(lo, hi);
    }
  }
}

// Epilogue shared by the GEMM kernels.  The accumulators go through LDS (the operand tiles are
// dead by now) so that each lane ends up with 8 consecutive columns of one row: x0 / x / R are
// read and y / u written as 16-byte (bf16) or 2x16-byte (fp32) vectors.
// ---- epilogue, in chunks of 32 rows x 64 columns per wave --------------------------------------
// The accumulators go through LDS (the operand tiles are dead: every wave passed the loop's last
// barrier) so that each lane ends up with 8 consecutive columns of one row: x0 / x / R are read
// and y / u written as 16-byte (bf16) or 2x16-byte (fp32) vectors.
// MFMA C/D layout: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
// epi_prefetch issues a chunk's x0 / x / R loads (specialised forms EPI 1 / 2), epi_process stages
// the chunk's accumulators and writes it; callers order them so that a chunk's loads are in
// flight while the previous chunk is processed.
// After a loop that waited for its LDS-DMA with opaque assembly, hipcc still believes the DMA may be in flight
// and fences EVERY later LDS read with `s_waitcnt vmcnt(0)` -- in the epilogue that made each 16-byte store wait
// for the acknowledgement of all earlier stores of the wave.  A counted wait it can see (free at run time: the
// loop has retired everything but the ALLOW youngest loads) tells it that the DMA has landed.
template <int ALLOW>
__device__ __forceinline__ void lds_dma_retired() {
  static_assert(ALLOW >= 0 && ALLOW < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((ALLOW & 15) | ((ALLOW >> 4) << 14) | (7 << 4) | (15 << 8));  // vmcnt only
}

struct EpiOperands {
  uint4 ex[4], ex0[4];
};

template <int EPI>
__device__ __forceinline__ void epi_prefetch(const GemmParams& p, EpiOperands& o, int64_t row0, int64_t wn0) {
  if constexpr (EPI == 1 || EPI == 2) {
    const int lane = threadIdx.x & 63;
    const int64_t gnc = min(wn0 + (lane & 7) * 8, p.n - 8);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t gmc = min(row0 + it * 8 + (lane >> 3), p.m - 1);
      if constexpr (EPI == 1) {
        o.ex[it] = load8_bf16_nt(p.ep.x, gmc * p.ep.ldx + gnc);
        // (unconditional even when x is x0: a predicated load makes hipcc drain the whole load queue)
        o.ex0[it] = load8_bf16_nt(p.ep.x0, gmc * p.ep.ldx + gnc);
      } else {
        o.ex[it] = load8_bf16_nt(p.ep.r, gmc * p.ep.ldr + gnc);
      }
    }
  }
}

// acc2: the chunk's two 32x32 accumulator fragments (columns 0..31 / 32..63); `stage`: the wave's
// private 32 x SST floats of LDS; (row0, wn0): origin of the chunk in C.
template <int EPI>
__device__ __forceinline__ void epi_process(const GemmParams& p, f32x16 (&acc2)[2], const EpiOperands& o, float* stage,
                                            int64_t row0, int64_t wn0, int split) {
  const int lane = threadIdx.x & 63;
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  constexpr int SST = 68;  // staging row stride in floats (64 + pad)
  const int ec = (lane & 7) * 8;
  const int64_t gn = wn0 + ec;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      stage[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * SST + j * 32 + frow] = acc2[j][r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-private staging: no workgroup barrier
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int er = it * 8 + (lane >> 3);
    const int64_t gm = row0 + er;
    if (gm >= p.m || gn >= p.n) continue;
    float v[8];
    const float4 v0 = *reinterpret_cast<const float4*>(stage + er * SST + ec);
    const float4 v1 = *reinterpret_cast<const float4*>(stage + er * SST + ec + 4);
    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    if constexpr (EPI == 1) {
      if (p.ep.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.ep.bias + gn);
        const float4 b1 = *reinterpret_cast<const float4*>(p.ep.bias + gn + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (p.ep.act != KRS_ACT_NONE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = apply_act(p.ep.act, v[q]);
      }
      if (p.ep.u_out) store8_bf16_nt(p.ep.u_out, gm * p.ep.ldu + gn, v);
      float xv[8], x0v[8];
      unpack_bf16x8(o.ex[it], xv);
      unpack_bf16x8(o.ex0[it], x0v);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = x0v[q] * (v[q] + p.ep.diag_scale * xv[q]) + xv[q];
      store8_bf16_nt(p.c, gm * p.ldc + gn, v);
    } else if constexpr (EPI == 2) {
      float rv[8];
      unpack_bf16x8(o.ex[it], rv);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += p.ep.beta * rv[q];
      store8_bf16_nt(p.c, gm * p.ldc + gn, v);
    } else if (p.splits > 1) {
      float* dst = p.slabs + ((int64_t)split * p.m + gm) * p.n + gn;
      if (p.ep_vec) {
        *reinterpret_cast<float4*>(dst) = v0;
        *reinterpret_cast<float4*>(dst + 4) = v1;
      } else {
        for (int q = 0; q < 8 && gn + q < p.n; ++q) dst[q] = v[q];
      }
    } else if (p.ep_vec) {
      epilogue_store_vec8(p, gm, gn, v);
    } else {
      for (int q = 0; q < 8 && gn + q < p.n; ++q) epilogue_store(p, gm, gn + q, v[q]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// a wave's 64x64 block of C (two chunks)
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_wave(const GemmParams& p, f32x16 (&acc)[2][2], float* stage, int64_t wm0,
                                                   int64_t wn0, int split) {
  EpiOperands o0, o1;
  epi_prefetch<EPI>(p, o0, wm0, wn0);
  epi_prefetch<EPI>(p, o1, wm0 + 32, wn0);
  epi_process<EPI>(p, acc[0], o0, stage, wm0, wn0, split);
  epi_process<EPI>(p, acc[1], o1, stage, wm0 + 32, wn0, split);
}

// a wave's 128x64 block of C (four chunks), two chunks of operands in flight
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_wave128(const GemmParams& p, f32x16 (&acc)[2][2][2], float* stage,
                                                      int64_t wm0, int64_t wn0, int split) {
  EpiOperands o0, o1;
  epi_prefetch<EPI>(p, o0, wm0, wn0);
  epi_prefetch<EPI>(p, o1, wm0 + 32, wn0);
  epi_process<EPI>(p, acc[0][0], o0, stage, wm0, wn0, split);
  epi_prefetch<EPI>(p, o0, wm0 + 64, wn0);
  epi_process<EPI>(p, acc[0][1], o1, stage, wm0 + 32, wn0, split);
  epi_prefetch<EPI>(p, o1, wm0 + 96, wn0);
  epi_process<EPI>(p, acc[1][0], o0, stage, wm0 + 64, wn0, split);
  epi_process<EPI>(p, acc[1][1], o1, stage, wm0 + 96, wn0, split);
}

// the same with the operands of the first NPFC chunks already requested by the caller (under the tail of its
// main loop): 4 = all of them (residual form: one operand), 2 = the first two (cross form: x0 and x)
template <int EPI, int NPFC>
__device__ __forceinline__ void gemm_epilogue_wave128_pre(const GemmParams& p, f32x16 (&acc)[2][2][2], float* stage,
                                                          int64_t wm0, int64_t wn0, int split, EpiOperands (&o)[4]) {
  static_assert(NPFC == 2 || NPFC == 4, "two or four chunks of operands are fetched ahead");
  epi_process<EPI>(p, acc[0][0], o[0], stage, wm0, wn0, split);
  if constexpr (NPFC == 2) epi_prefetch<EPI>(p, o[2], wm0 + 64, wn0);
  epi_process<EPI>(p, acc[0][1], o[1], stage, wm0 + 32, wn0, split);
  if constexpr (NPFC == 2) epi_prefetch<EPI>(p, o[3], wm0 + 96, wn0);
  epi_process<EPI>(p, acc[1][0], o[2], stage, wm0 + 64, wn0, split);
  epi_process<EPI>(p, acc[1][1], o[3], stage, wm0 + 96, wn0, split);
}

// 128x128 workgroup tile, 4 waves as 2x2
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[2][2], char* smem, int64_t m0,
                                              int64_t n0, int split) {
  const int wave = threadIdx.x >> 6;
  gemm_epilogue_wave<EPI>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * 68), m0 + (wave >> 1) * 64,
                          n0 + (wave & 1) * 64, split);
}

// EPI: 0 = general epilogue, 1 = cross (bf16, vector access), 2 = residual add (bf16, vector access).
// The specialised forms issue all their x0 / x / R loads for a 32-row half before the accumulators
// are staged, so the epilogue is one memory latency deep instead of one per row group.
template <int ES, bool A_KM, bool B_NK, int EPI>
__global__ __launch_bounds__(256, 3) void gemm_mfma_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = ROW_BYTES / ES;  // K elements per tile
  // ONE operand buffer (A tile, then B tile): 36 KB of LDS per workgroup, so three workgroups
  // share a CU and one's staging / epilogue phases hide under the others' MFMA phases.
  auto tile_a = [&](int) -> char* { return smem; };
  auto tile_b = [&](int) -> char* { return smem + TILE_BYTES; };

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroup ids are dealt round-robin to the 8 XCDs (observed: id % 8),
  // each with a private L2.  The N tiles of one M panel re-read the same A rows, so they are
  // mapped to ONE XCD, back to back: (xcd, slot) -> panel (slot / Nt) * 8 + xcd, tile slot % Nt.
  // A wrong placement guess only costs speed.  grid.x = ceil(Mt / 8) * 8 * Nt; surplus panels exit.
  const int64_t nt = (p.n + BN - 1) / BN;
  int64_t m_tile, n_tile;
  if constexpr (A_KM) {  // weight-gradient shapes: few tiles, split along K instead
    m_tile = blockIdx.x / nt;
    n_tile = blockIdx.x % nt;
  } else {
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    m_tile = (slot / nt) * 8 + xcd;
    n_tile = slot % nt;
  }
  if (m_tile * BM >= p.m) return;
  const int64_t m0 = m_tile * BM;
  const int64_t n0 = n_tile * BN;
  const int split = blockIdx.z;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  const int64_t kend = min(p.k, kbeg + p.k_per_split);
  const int64_t ntiles = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  u32x4 ra[4], rb[4];
  auto load_tiles = [&](int64_t t) {
    const int64_t k0 = kbeg + t * BK;
    if constexpr (A_KM) load_kstrided<ES>(p.a, p.lda, m0, k0, p.m, kend, ra);
    else load_kcontig<ES>(p.a, p.lda, m0, k0, p.m, kend, ra);
    if constexpr (B_NK) load_kcontig<ES>(p.b, p.ldb, n0, k0, p.n, kend, rb);
    else load_kstrided<ES>(p.b, p.ldb, n0, k0, p.n, kend, rb);
  };
  auto store_tiles = [&](int buf) {
    if constexpr (A_KM) store_kstrided<ES>(tile_a(buf), ra); else store_kcontig(tile_a(buf), ra);
    if constexpr (B_NK) store_kcontig(tile_b(buf), rb); else store_kstrided<ES>(tile_b(buf), rb);
  };

  if (ntiles > 0) load_tiles(0);

  const int frow = lane & 31;          // fragment row (m for A, n for B)
  const int fhalf = lane >> 5;         // which 16-byte half of a 32-byte K step this lane feeds
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = 0;
    __syncthreads();          // every wave is done reading the previous tile
    store_tiles(0);
    __syncthreads();
    if (t + 1 < ntiles) load_tiles(t + 1);  // in flight under the MFMAs below
    const char* ta = tile_a(cur);
    const char* tb = tile_b(cur);
#pragma unroll
    for (int ks = 0; ks < ROW_BYTES / 32; ++ks) {  // 32 bytes of K per step (2 lane halves x 16 B)
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const u32x4*>(ta + lds_chunk_off<A_KM>(wm * 64 + i * 32 + frow, ks * 2 + fhalf));
        fb[i] = *reinterpret_cast<const u32x4*>(tb + lds_chunk_off<!B_NK>(wn * 64 + i * 32 + frow, ks * 2 + fhalf));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ES == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i][q]),
                                                               __uint_as_float(fb[j][q]), acc[i][j], 0, 0, 0);
          }
        }
    }
  }
  __syncthreads();  // operand tiles are dead from here on

  gemm_epilogue<EPI>(p, acc, smem, m0, n0, split);
}

// Forward / data-gradient shapes (both operands K-contiguous, K a multiple of the 64-element tile):
// the operand tiles are written straight into LDS by the memory pipeline
// (global_load_lds_dwordx4: no staging registers, no ds_write traffic), two stages deep:
//   issue tile t+1 -> MFMA on tile t -> vmcnt(0) + one barrier.
// An LDS-DMA instruction writes 64 lanes x 16 B = 1 KB contiguous (8 tile rows of 128 B), so the
// tile cannot be padded; the conflict-free read layout is produced on the SOURCE side instead:
// lane (row r, physical chunk pc) fetches the logical chunk pc ^ ((r >> 1) & 7), and fragment
// reads apply the same XOR (scripts/lds_conflicts.py: 4 cycles per ds_read_b128, the minimum).
// Rows beyond M / N are clamped (their results are never stored).
template <int ES, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = ROW_BYTES / ES;
  constexpr int OP_BYTES = BM * ROW_BYTES;   // 16 KB per operand tile
  constexpr int STAGE_BYTES = 2 * OP_BYTES;  // A then B
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t nt = (p.n + BN - 1) / BN;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m_tile = (slot / nt) * 8 + xcd;
  if (m_tile * BM >= p.m) return;
  const int64_t m0 = m_tile * BM;
  const int64_t n0 = (slot % nt) * BN;
  const int64_t ntiles = p.k / BK;

  // per-lane source rows / chunks of the 4 DMA pieces this wave issues per operand and tile
  const char* asrc[4];
  const char* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);            // tile row this lane fills
    const int c = (lane & 7) ^ ((r >> 1) & 7);                  // logical 16-byte chunk it must fetch
    const int64_t ar = min(m0 + r, p.m - 1);
    const int64_t br = min(n0 + r, p.n - 1);
    asrc[i] = p.a + (ar * p.lda) * ES + c * 16;
    bsrc[i] = p.b + (br * p.ldb) * ES + c * 16;
  }
  auto issue = [&](int64_t t, int stage) {
    char* sa = smem + stage * STAGE_BYTES + (wave * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr)(asrc[i] + t * ROW_BYTES), (lptr)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(bsrc[i] + t * ROW_BYTES), (lptr)(sa + OP_BYTES + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (ntiles > 0) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  // fragment row offsets (bytes) and swizzle keys are loop invariant
  int aoff[2], boff[2], akey[2], bkey[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + frow, rb = wn * 64 + i * 32 + frow;
    aoff[i] = ra * ROW_BYTES; akey[i] = (ra >> 1) & 7;
    boff[i] = OP_BYTES + rb * ROW_BYTES; bkey[i] = (rb >> 1) & 7;
  }
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < ntiles) issue(t + 1, cur ^ 1);
    const char* st = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < ROW_BYTES / 32; ++ks) {
      const int c = ks * 2 + fhalf;
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const u32x4*>(st + aoff[i] + ((c ^ akey[i]) << 4));
        fb[i] = *reinterpret_cast<const u32x4*>(st + boff[i] + ((c ^ bkey[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ES == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i][q]),
                                                               __uint_as_float(fb[j][q]), acc[i][j], 0, 0, 0);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed
    __syncthreads();                                   // and every wave is done with tile t
  }
  lds_dma_retired<0>();
  gemm_epilogue<EPI>(p, acc, smem, m0, n0, 0);
}

// Weight-gradient shapes in bf16 (C = A^T B with BOTH operands K-strided in HBM: activations
// contracted over the batch).  The operand tiles go to LDS by DMA exactly as they lie in memory --
// rows of K, 16-byte chunks of 8 columns -- and the MFMA fragments (8 consecutive K per lane) are
// read with gfx950's transposing ds_read_b64_tr_b16: no staging registers, no in-register
// transposes, no ds_write traffic (the register-staged path spends half its LDS cycles on the
// bank conflicts of its transposed writes).
// LDS image of one operand tile (64 k x 128 columns): 16 blocks of 1 KB, block kb = 4 k-rows;
// inside a block the 64-byte unit (cq, kr) (32-column quarter cq, k-row kr) sits at (cq*4 + kr)*64,
// which is the lane-linear order of one DMA instruction whose lanes 4u..4u+3 fetch the 64
// contiguous bytes of unit u (so the memory side still sees whole 64-byte quads).
// A transposing read hands lane i of a 16-lane group column i of the 4 x 16 matrix whose row r is
// the 4 x 8 bytes addressed by lanes 4r..4r+3: a 32-lane half addresses the four k-rows of one
// 32-column quarter, i.e. 256 contiguous bytes -- every bank once.  Two reads (blocks kb, kb+1)
// give a lane its 8 k-values.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(const GemmParams p, int mt, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = 64;
  constexpr int OP_BYTES = BK * 128 * 2;     // 16 KB
  constexpr int STAGE_BYTES = 2 * OP_BYTES;
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* trptr;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // One K split = one XCD at a time (workgroup id % 8 picks the XCD): all mt*nt tiles of a split walk
  // the same K rows together, so a row of A / B is consumed whole (by the tiles side by side) while
  // its DRAM page and TLB entry are hot, and the operand tiles are shared through that XCD's L2.
  const int64_t tiles = (int64_t)mt * nt;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = (int)((slot / tiles) * 8 + xcd);
  if (split >= p.splits) return;
  const int64_t tile = slot % tiles;
  const int64_t m0 = (tile % mt) * BM;
  const int64_t n0 = (tile / mt) * BN;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  const int64_t kend = min(p.k, kbeg + p.k_per_split);
  const int64_t ntiles = (kend - kbeg) / BK;   // K extents are multiples of 64 (checked on the host)

  // DMA: this wave fills blocks kb = wave*4 + j of each operand; lane = (quarter*4 + k-row)*4 + chunk
  const int dkr = (lane >> 2) & 3, dcol = (lane >> 4) * 32 + (lane & 3) * 8;
  const int64_t acol = min(m0 + dcol, p.m - 8);   // clamped: surplus columns are never stored
  const int64_t bcol = min(n0 + dcol, p.n - 8);
  const char* asrc = p.a + ((kbeg + wave * 16 + dkr) * p.lda + acol) * 2;
  const char* bsrc = p.b + ((kbeg + wave * 16 + dkr) * p.ldb + bcol) * 2;
  const int64_t astep = p.lda * 8, bstep = p.ldb * 8;   // 4 k-rows, in bytes
  auto issue = [&](int64_t t, int stage) {
    char* sa = smem + stage * STAGE_BYTES + (wave * 4) * 1024;
    const char* at = asrc + t * BK * p.lda * 2;
    const char* bt = bsrc + t * BK * p.ldb * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr)(at + j * astep), (lptr)(sa + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(bt + j * bstep), (lptr)(sa + OP_BYTES + j * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (ntiles > 0) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // transposing-read addresses: group g = lane >> 4 (columns (g & 1) * 16 .., k half g >> 1),
  // lane-in-group ii: k-row ii >> 2, 4-column quad ii & 3
  const int g = lane >> 4, ii = lane & 15;
  const int lane_off = (ii >> 2) * 64 + (g & 1) * 32 + (ii & 3) * 8 + (g >> 1) * 2048;
  int aoff[2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // a fragment's 32 columns are one quarter of the tile
    aoff[i] = (wm * 2 + i) * 256 + lane_off;
    boff[i] = OP_BYTES + (wn * 2 + i) * 256 + lane_off;
  }
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < ntiles) issue(t + 1, cur ^ 1);
    char* st = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr)(st + aoff[i] + ks * 4096));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr)(st + aoff[i] + ks * 4096 + 1024));
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr)(st + boff[i] + ks * 4096));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr)(st + boff[i] + ks * 4096 + 1024));
        const uint2 ua0 = __builtin_bit_cast(uint2, a0), ua1 = __builtin_bit_cast(uint2, a1);
        const uint2 ub0 = __builtin_bit_cast(uint2, b0), ub1 = __builtin_bit_cast(uint2, b1);
        fa[i] = u32x4{ua0.x, ua0.y, ua1.x, ua1.y};
        fb[i] = u32x4{ub0.x, ub0.y, ub1.x, ub1.y};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                              __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed
    __syncthreads();                                   // and every wave is done with tile t
  }
  lds_dma_retired<0>();
  gemm_epilogue<EPI>(p, acc, smem, m0, n0, split);
}

// ---- fused epilogue of krs_gemm_cross_bwd (round 4) ----------------------------------------------------------------
// The data-gradient product of a cross layer, G = A B^T + beta R, is dL/dy of the layer BELOW it in a stack on one x0,
// whose elementwise backward (cross_bwd_vec_kernel) starts by re-reading G.  This epilogue does that pass on the tile
// while it is in registers: G is rounded and stored (the layer below needs it again as the residual of ITS data-gradient
// product), then from the ROUNDED value -- what the separate pass would have read --
//   dz = G x0 act'(u),   dx0 = [dx0 +] G u,   column sums of dz -> partial[group][n]  (bias gradient, fixed order)
// with the arithmetic of cross_bwd_vec_kernel for diag_scale = 0: G, dz and dx0 are bit-identical to the two calls.
// A wave's 128 x 64 block in four chunks of 32 rows; a chunk's R / x0 / u / dx0 vectors are requested before its
// accumulators are staged.  ACC: dx0 already holds the terms of the layers above (a template parameter: a load behind
// a run-time branch makes hipcc drain the load queue).
// DX0: where the running dL/dx0 comes from -- 0: nothing yet, 1: the dx0 buffer (terms of the layers above), 2: R * u_upper,
// the term of the layer ABOVE computed here from its dL/dy (= R, already loaded) and its saved u, so that the top layer of a
// stack neither writes nor this launch reads a [M, N] matrix for it (that one sum is rounded once instead of twice);
// 3: no dL/dx0 at all from this launch (the caller hands u to the NEXT launch as its u_upper: a Dense layer above a stack).
// X0 = false: the DENSE form (krs_gemm_cross_bwd with x0 = NULL, round 6) -- the layer below is a Dense layer, dz = G act'(y)
// with y in the place of u, no x0 stream, no dL/dx0, and G itself is not stored (nobody reads the raw data gradient of a
// Dense output): two streams (y in, dz out) instead of krs_dense_act_bwd's three behind a stored and re-read G.
template <int DX0, bool HAS_R, bool X0 = true>
__device__ __forceinline__ void gemm_epilogue_wave128_crossbwd(const GemmParams& p, f32x16 (&acc)[2][2][2], float* stage,
                                                               int64_t wm0, int64_t wn0, int64_t group) {
  static_assert(X0 || (DX0 == 3 && !HAS_R), "the dense form has no R and no dL/dx0");
  const int lane = threadIdx.x & 63;
  const int frow = lane & 31, fhalf = lane >> 5;
  constexpr int SST = 68;
  const int ec = (lane & 7) * 8;
  const int64_t gn = wn0 + ec;
  const int64_t gnc = min(gn, p.n - 8);
  float db[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) db[q] = 0.0f;
  // operand vectors of a chunk, two sets: chunk c + 1's are requested while chunk c is processed -- x0 and u as soon as
  // chunk c's accumulators are staged (their registers are free from then on), R and the dL/dx0 source too from the second
  // chunk on.  (Requesting them earlier -- right behind the staging of chunk c's accumulators -- measured equal,
  // profiles/r4y_cross_bwd_fused_prefetch.txt: every stream of chunk c + 1 is requested behind chunk c's stores.)
  uint4 vr2[2][4], vx02[2][4], vu2[2][4], vd2[2][4];
  auto ld_x0u = [&](int c, int b) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t gmc = min(wm0 + c * 32 + it * 8 + (lane >> 3), p.m - 1);
      if constexpr (X0) vx02[b][it] = load8_bf16_nt(p.f_x0, gmc * p.f_ld + gnc);
      vu2[b][it] = load8_bf16_nt(p.f_u, gmc * p.f_ld + gnc);
    }
  };
  auto ld_rd = [&](int c, int b) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t gmc = min(wm0 + c * 32 + it * 8 + (lane >> 3), p.m - 1);
      if constexpr (HAS_R) vr2[b][it] = load8_bf16_nt(p.ep.r, gmc * p.ep.ldr + gnc);
      if constexpr (DX0 == 1) vd2[b][it] = load8_bf16_nt(p.f_dx0, gmc * p.f_ld + gnc);
      else if constexpr (DX0 == 2) vd2[b][it] = load8_bf16_nt(p.f_uup, gmc * p.f_ld + gnc);
    }
  };
  ld_x0u(0, 0);
  ld_rd(0, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t row0 = wm0 + c * 32;
    uint4(&er)[4] = vr2[c & 1];
    uint4(&ex0)[4] = vx02[c & 1];
    uint4(&eu)[4] = vu2[c & 1];
    uint4(&ed)[4] = vd2[c & 1];
    f32x16(&acc2)[2] = acc[c >> 1][c & 1];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * SST + j * 32 + frow] = acc2[j][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int er_ = it * 8 + (lane >> 3);
      const int64_t gm = row0 + er_;
      if (gm >= p.m || gn >= p.n) continue;
      float v[8], rv[8], x0v[8], uv[8], tv[8], g[8], dz[8];
      const float4 v0 = *reinterpret_cast<const float4*>(stage + er_ * SST + ec);
      const float4 v1 = *reinterpret_cast<const float4*>(stage + er_ * SST + ec + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      if constexpr (HAS_R) {
        unpack_bf16x8(er[it], rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += p.ep.beta * rv[q];
      }
      // G as it is stored (one rounding) is what everything below sees
      const uint4 gq = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                  pack_bf16x2(v[6], v[7]));
      if constexpr (X0) {
        const u32x4 gs = {gq.x, gq.y, gq.z, gq.w};
        __builtin_nontemporal_store(gs, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.c) + gm * p.ldc + gn));
      }
      unpack_bf16x8(gq, g);
      if constexpr (X0) unpack_bf16x8(ex0[it], x0v);
      unpack_bf16x8(eu[it], uv);
      if constexpr (DX0 == 1 || DX0 == 2) unpack_bf16x8(ed[it], tv);
      else {
#pragma unroll
        for (int q = 0; q < 8; ++q) tv[q] = 0.0f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float gx0 = X0 ? g[q] * x0v[q] : g[q];
        dz[q] = gx0 * act_grad_from_output(p.f_act, uv[q]);
        db[q] += dz[q];
        const float told = DX0 == 1 ? tv[q] : (DX0 == 2 ? rv[q] * tv[q] : 0.0f);
        tv[q] = __builtin_fmaf(g[q], uv[q], told);
        if (p.f_fold) tv[q] += g[q];     // the layer below is the bottom of its stack (x is x0): the direct term too
      }
      store8_bf16_nt(p.f_dz, gm * p.f_ld + gn, dz);   // (nt like the other streams: 657-660 us against 664-667 with a plain store;
                                                       //  the dh / dK products that read dz next measure the same either way)
      if constexpr (DX0 != 3) store8_bf16_nt(p.f_dx0, gm * p.f_ld + gn, tv);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (c < 3) {
      ld_x0u(c + 1, (c + 1) & 1);
      ld_rd(c + 1, (c + 1) & 1);
    }
  }
  if (p.f_partial) {
    // column sums over the wave's 128 rows: the eight lanes with one (lane & 7) hold the same 8 columns
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float t = db[q];
      t += __shfl_xor(t, 8);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      db[q] = t;
    }
    if ((lane >> 3) == 0 && gn < p.n) {
      float* dst = p.f_partial + group * p.n + gn;
      *reinterpret_cast<float4*>(dst) = make_float4(db[0], db[1], db[2], db[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(db[4], db[5], db[6], db[7]);
    }
  }
}

// ---- ping-pong ring pipeline: the 256x256 bf16 tile of both operand layouts -------------------------
// The two kernels above run "issue tile t+1 -> 32 MFMA on tile t -> vmcnt(0) -> barrier": the DMA queue is
// filled in one burst and drained to empty once per tile, all eight waves read LDS at the same time and
// feed the matrix pipes at the same time (profiles/r1j_gemm_pmc.md: waves parked 44-53 % of their cycles).
// This kernel keeps the tile, the LDS images and the epilogue and replaces the loop:
//   * K advances in blocks of 32; a block is two 16 KB PIECES (A: 256 rows x 64 B, B likewise; for the
//     K-strided layout 32 k-rows x 256 columns), 2 LDS-DMA instructions per thread each; the pieces live in
//     a ring of NSTG stages (4 = 128 KB, 5 = all 160 KB of the CU);
//   * a block is consumed in two PHASES of 16 k (6 ds_read_b128 / 12 transposing reads + 8 MFMA per wave),
//     each phase = a LOAD segment (fragment reads of this phase) and a COMPUTE segment (the 8 MFMA with the
//     phase's ONE piece of DMA issue -- two instructions per wave -- behind its 2nd and 5th MFMA, round 6),
//     every segment closed by s_barrier;
//   * waves 0-3 (rows 0..127 of the tile: one wave per SIMD) and waves 4-7 (rows 128..255: the other wave
//     of every SIMD) run one segment apart -- waves 4-7 pass one extra barrier first -- so on every SIMD
//     one wave computes while its partner reads LDS and issues DMA;
//   * nothing is ever drained: pieces A(j), B(j) are issued in phases 2(j-NSTG)+3 / +4 (a slot is refilled
//     two phases after the last read of its previous content: that read was retired by the reader's
//     lgkmcnt(0) one barrier earlier), and the only wait is a COUNTED vmcnt at the end of every odd phase
//     2kb+1, which retires the two pieces of block kb+1 and leaves the younger pieces in flight (that phase's
//     own piece is issued behind the wait, in its compute segment: 2*NSTG-6 pieces = STEADY - 2 instructions).
//     A wave's vmcnt covers its own DMA writes; the barrier that follows publishes them.
// Results are bit-identical to the two-stage kernels (same MFMA chain per accumulator: k ascending).
// Measured and not kept (profiles/archive/r2_gemm_ab.txt): 128-byte rows with a row-wise walk of the wave's block (the
// "half-tile" ring of the 8-phase template: 205 us against 207 for h = x U, 342 against 338 for dx -- round 6's
// gemm_pp64_kernel is that idea on 64-k pieces with the DMA among the MFMAs: 189 us); in ROUND 2 issuing the
// phase's DMA between the MFMAs of the compute segment instead of beside the fragment reads measured worse (212 / 319 us
// against 205 / 274 for the K-contiguous / K-strided forms of that day's kernel) -- re-measured in round 6 on today's
// kernels it is the better place (215 -> 212 us here, 195 -> 189 in gemm_pp64_kernel:
// profiles/r6_gemm_k64_dma_in_compute_ab.txt); five stages instead of four (equal); for the short-K
// products with heavy epilogues, 256 x 128 tiles on a three-stage ring at TWO workgroups per CU, so that one's
// epilogue runs under the other's main loop (467 / 344 us against 450 / 323: 1.5x the operand bytes per flop cost
// more than the overlap returns).
// (Rounds 2-4 carried timing-only build switches in this kernel -- KRS_PP_PROBE: DMA stream alone / LDS reads + MFMA alone /
//  epilogue alone; KRS_PP_LAYOUT_EXP: operands read as if pre-tiled; KRS_PP_{A,B}_AUX: cache policy of the LDS-DMA loads -- and
//  two more schedules behind krs_gemm_set_option: a five-stage ring and a prefetch schedule.  What they measured is in
//  profiles/archive/r2_pp_probe_dma_lds_mfma.txt, r4_gemm_layout_waves_clock_probe.txt, r2_gemm_ab.txt; round 5 removed them from
//  the product source: the default cache policy, four stages and the ping-pong schedule are what ships.)
namespace pp {
constexpr int PIECE = 16384;
constexpr int STAGE = 2 * PIECE;
}  // namespace pp

__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// Transposing LDS read as opaque assembly: hipcc treats the intrinsic as a read of memory an LDS-DMA may have
// written and puts `s_waitcnt vmcnt(0)` in front of every group of them -- which drained the ring's DMA queue
// once per phase in the K-strided form.  The caller orders these reads against the DMA itself (counted vmcnt +
// barrier) and waits for them with pp_lgkm_wait().
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2 pp_read_tr16(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// s_waitcnt lgkmcnt(N) that the fragments' consumers depend on (the registers pass through the statement)
template <int N>
__device__ __forceinline__ void pp_lgkm_wait(u32x4 (&fa)[4], u32x4 (&fb)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1])
               : "n"(N));
}
template <int N>
__device__ __forceinline__ void pp_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// (Round 4 also built the 128x128 tile on a deep ring -- gemm_ring128_kernel, KRS_GEMM_OPT_PIPELINE = 7 -- for the per-rank
//  shapes of a strongly-scaled job; it measured equal or slower than gemm_glds_kernel (47.2 against 44.9 us at M = 8192,
//  profiles/archive/r4_gemm_ring128_b8192.txt: one 128x128 tile per CU already draws the ~40 GB/s per CU of the L2 -> LDS path) and
//  was deleted in round 5.)
// The steady state's LDS-DMA instructions are issued INSIDE the compute segment, behind its 2nd and 5th MFMA, not in the load
// segment (round 6: the load segment -- six fragment reads + two DMA instructions at 60-185 cycles each -- was the longer
// of the two and set the phase length; weight gradients 215 -> 212 us, profiles/r6_gemm_k64_dma_in_compute_ab.txt)
template <bool TN, int NSTG, int EPI>
__global__ __launch_bounds__(512) void gemm_pp256_kernel(const GemmParams p, int mt, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = 256, TN_ = 256;
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;
  const int wm = grp, wn = wave & 3;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int64_t m0, n0, kbeg = 0, nkb;
  int split = 0;
  if constexpr (TN) {
    // work items (split, tile), split-major, are dealt to the XCDs in contiguous runs of q = ceil(W / 8): the
    // tiles of one K split run side by side on one XCD (at most two splits meet on an XCD), so a k-row of
    // A / B is consumed whole while its DRAM page is open and the operand tiles are shared through that L2
    const int64_t tiles = (int64_t)mt * nt, items = tiles * p.splits, q = (items + 7) / 8;
    const int64_t w = xcd * q + slot;
    if (slot >= q || w >= items) return;
    split = (int)(w / tiles);
    const int64_t tile = w % tiles;
    m0 = (tile % mt) * TM;
    n0 = (tile / mt) * TN_;
    kbeg = (int64_t)split * p.k_per_split;
    nkb = (min(p.k, kbeg + p.k_per_split) - kbeg) / 32;
  } else {
    // the N tiles of one M panel run back to back on one XCD (workgroup id % 8) and share A in its L2.  With split-K
    // (outputs too small to fill the chip with 256x256 tiles: the per-rank products of a strongly-scaled job) the
    // splits of a tile are neighbours on that XCD too: slot = ((m group * nt) + n tile) * splits + split
    const int64_t tslot = slot / p.splits;
    split = (int)(slot - tslot * p.splits);
    const int64_t m_tile = (tslot / nt) * 8 + xcd;
    if (m_tile * TM >= p.m) return;
    m0 = m_tile * TM;
    n0 = (tslot % nt) * TN_;
    kbeg = (int64_t)split * p.k_per_split;
    nkb = (min(p.k, kbeg + p.k_per_split) - kbeg) / 32;
  }

  // DMA sources of this thread's two instructions per piece; `astep` / `bstep` = bytes per K block
  const char* ap[2];
  const char* bp[2];
  int64_t astep, bstep;
  if constexpr (!TN) {
    // instruction q = wave*2 + i fills rows q*16 .. q*16+15 (64 B each): lane = row*4 + physical chunk,
    // which holds the logical 16-byte chunk pc ^ ((row >> 2) & 3) (conflict-free ds_read_b128, see below)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      ap[i] = p.a + (min(m0 + r, p.m - 1) * p.lda + kbeg) * 2 + c * 16;
      bp[i] = p.b + (min(n0 + r, p.n - 1) * p.ldb + kbeg) * 2 + c * 16;
    }
    astep = bstep = 64;
  } else {
    // instruction q fills the 1 KB block (128-column half q >> 3, k-rows (q & 7)*4 .. +3) with the image of
    // gemm_tn_glds_kernel: lane = (quarter*4 + k-row)*4 + chunk
    const int kr = (lane >> 2) & 3, col = (wave >> 2) * 128 + (lane >> 4) * 32 + (lane & 3) * 8;
    const int64_t acol = min(m0 + col, p.m - 8), bcol = min(n0 + col, p.n - 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t krow = kbeg + ((wave & 3) * 2 + i) * 4 + kr;
      ap[i] = p.a + (krow * p.lda + acol) * 2;
      bp[i] = p.b + (krow * p.ldb + bcol) * 2;
    }
    astep = p.lda * 64;
    bstep = p.ldb * 64;
  }
  const int dma_off = wave * 2048;  // + i*1024: where instruction q = wave*2 + i lands inside a piece
  auto issue_a = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr)ap[i], (lptr)(smem + stage * pp::STAGE + dma_off + i * 1024), 16, 0, 0);
      ap[i] += astep;
    }
  };
  auto issue_b = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr)bp[i], (lptr)(smem + stage * pp::STAGE + pp::PIECE + dma_off + i * 1024), 16,
                                       0, 0);
      bp[i] += bstep;
    }
  };
  auto issue_one = [&](bool is_a, int stage, int i) {
    if (is_a) {
      __builtin_amdgcn_global_load_lds((gptr)ap[i], (lptr)(smem + stage * pp::STAGE + dma_off + i * 1024), 16, 0, 0);
      ap[i] += astep;
    } else {
      __builtin_amdgcn_global_load_lds((gptr)bp[i], (lptr)(smem + stage * pp::STAGE + pp::PIECE + dma_off + i * 1024), 16, 0, 0);
      bp[i] += bstep;
    }
  };

  // fragment addresses inside a stage (phase hk = 1: ^ 32 for the K-contiguous image, + 4096 for the other)
  int a_lane, b_lane;
  if constexpr (!TN) {
    // 64-byte rows: lane (frow, fhalf) wants chunk hk*2 + fhalf of row frow; a ds_read_b128 is served in
    // groups of 16 lanes whose rows are {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): with the chunk
    // XOR (row >> 2) & 3 every group covers the sixteen 16-byte slots of the 256-byte bank row once
    const int frow = lane & 31, fhalf = lane >> 5, key = (frow >> 2) & 3;
    a_lane = (wm * 128 + frow) * 64 + ((fhalf ^ key) << 4);
    b_lane = pp::PIECE + (wn * 64 + frow) * 64 + ((fhalf ^ key) << 4);
  } else {
    const int g = lane >> 4, ii = lane & 15;
    const int lane_off = (ii >> 2) * 64 + (g & 1) * 32 + (ii & 3) * 8 + (g >> 1) * 2048;
    a_lane = wm * 8192 + lane_off;
    b_lane = pp::PIECE + (wn >> 1) * 8192 + (wn & 1) * 512 + lane_off;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto load_frags = [&](int stage, int hk, u32x4(&fa)[4], u32x4(&fb)[2]) {
    const char* st = smem + stage * pp::STAGE;
    if constexpr (!TN) {
      const int x = hk * 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x4*>(st + (b_lane ^ x) + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + (a_lane ^ x) + i * 2048);
    } else {
      const uint32_t sa = lds_base + stage * pp::STAGE + hk * 4096 + a_lane, sb = lds_base + stage * pp::STAGE + hk * 4096 + b_lane;
#define KRS_TR2(dst, base, off)                                                            \
  {                                                                                        \
    const u32x2 lo_ = pp_read_tr16<(off)>(base), hi_ = pp_read_tr16<(off) + 1024>(base);   \
    dst = u32x4{lo_.x, lo_.y, hi_.x, hi_.y};                                               \
  }
      KRS_TR2(fb[0], sb, 0)
      KRS_TR2(fb[1], sb, 256)
      KRS_TR2(fa[0], sa, 0)
      KRS_TR2(fa[1], sa, 256)
      KRS_TR2(fa[2], sa, 512)
      KRS_TR2(fa[3], sa, 768)
#undef KRS_TR2
    }
  };

  f32x16 acc[2][2][2];  // [upper / lower 64 rows][m fragment][n fragment]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.0f;
  // epilogue operands fetched under the tail of the main loop (ping-pong schedule): chunks, load instructions
  constexpr int NPFC = EPI == 2 ? 4 : EPI == 1 ? 2 : 0;
  constexpr int NPF = NPFC * (EPI == 1 ? 8 : 4);
  EpiOperands opf[4];
  auto mfma8 = [&](const u32x4(&fa)[4], const u32x4(&fb)[2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16& d = acc[i >> 1][i & 1][j];
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), d, 0,
                                                   0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto mfma8_dma = [&](const u32x4(&fa)[4], const u32x4(&fb)[2], bool is_a, int stage) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16& d = acc[i >> 1][i & 1][j];
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), d, 0,
                                                   0, 0);
        if (i * 2 + j == 1) { __builtin_amdgcn_sched_barrier(0); issue_one(is_a, stage, 0); __builtin_amdgcn_sched_barrier(0); }
        if (i * 2 + j == 4) { __builtin_amdgcn_sched_barrier(0); issue_one(is_a, stage, 1); __builtin_amdgcn_sched_barrier(0); }
      }
    __builtin_amdgcn_s_setprio(0);
  };

  {
    // prologue (the host guarantees nkb >= NSTG): A(0), B(0), ..., A(NSTG-3), B(NSTG-3), A(NSTG-2)
  #pragma unroll
    for (int j = 0; j < NSTG - 2; ++j) {
      issue_a(j);
      issue_b(j);
    }
    issue_a(NSTG - 2);
    constexpr int STEADY = 4 * NSTG - 10;  // DMA instructions of the 2*NSTG-5 pieces that may stay in flight
    pp_vmcnt<STEADY>();
    pp_barrier();
    if (grp == 1) pp_barrier();  // rows 128..255 run one segment behind rows 0..127

    int rd = 0, wa = NSTG - 1, wb = NSTG - 2;  // stages of block kb, of the A piece issued in its odd / the B piece in its even phase
    const int64_t last = nkb - 1;
    auto advance = [&]() {
      rd = rd + 1 == NSTG ? 0 : rd + 1;
      wa = wa + 1 == NSTG ? 0 : wa + 1;
      wb = wb + 1 == NSTG ? 0 : wb + 1;
    };
    int64_t kb = 0;
    for (; kb + NSTG - 1 <= last; ++kb) {  // steady state: one piece issued per phase
      u32x4 fa[4], fb[2];
      // ---- phase 2kb ----
      load_frags(rd, 0, fa, fb);
      pp_barrier();
      if constexpr (TN) pp_lgkm_wait<0>(fa, fb);
      mfma8_dma(fa, fb, false, wb);
      pp_barrier();
      // ---- phase 2kb+1 ----
      load_frags(rd, 1, fa, fb);
      pp_vmcnt<STEADY - 2>();    // (this phase's piece is issued behind the wait, among the MFMAs)
      pp_barrier();
      if constexpr (TN) pp_lgkm_wait<0>(fa, fb);
      mfma8_dma(fa, fb, true, wa);
      pp_barrier();
      advance();
    }
    // tail: the last NSTG-1 blocks; only B(last) is still to be issued.  The epilogue's operands (residual form:
    // all four 32-row chunks of R; cross form: x0 and x of the first two) are requested here, behind the last
    // piece, so that their HBM latency runs under the remaining 2*(NSTG-1) phases instead of in front of the
    // epilogue; they are younger than every piece, so the counted waits simply allow NPF more loads in flight.
#pragma unroll
    for (int t = 0; t < NSTG - 1; ++t) {
      u32x4 fa[4], fb[2];
      load_frags(rd, 0, fa, fb);
      if (t == 0) issue_b(wb);
      pp_barrier();
      if constexpr (TN) pp_lgkm_wait<0>(fa, fb);
      mfma8(fa, fb);
      pp_barrier();
      load_frags(rd, 1, fa, fb);
      if constexpr (NPFC > 0) {
        if (t == 0) {
#pragma unroll
          for (int c = 0; c < NPFC; ++c) epi_prefetch<EPI>(p, opf[c], m0 + wm * 128 + c * 32, n0 + wn * 64);
        }
      }
      // block kb+1 must have landed; behind it only the pieces of blocks kb+2 .. last (rem = NSTG-3-t of them)
      // and the epilogue operands are still in flight
      if (NSTG - 3 - t >= 2) pp_vmcnt<8 + NPF>();
      else if (NSTG - 3 - t == 1) pp_vmcnt<4 + NPF>();
      else if (NSTG - 3 - t == 0) pp_vmcnt<NPF>();
      pp_barrier();
      if constexpr (TN) pp_lgkm_wait<0>(fa, fb);
      mfma8(fa, fb);
      pp_barrier();
      advance();
    }
    if (grp == 0) pp_barrier();
  }
  lds_dma_retired<NPF>();
  float* stage_f = reinterpret_cast<float*>(smem) + wave * (32 * 68);
  if constexpr (EPI >= 3) {   // 3: + R, 4: + R, dx0 accumulates, 5: no R, 6: no R, dx0 accumulates, 7: + R, dx0 = R u_upper + ...,
                              // 8: no R, no dx0, 9: the dense form (no x0, no R, no dx0, G not stored)
    gemm_epilogue_wave128_crossbwd<(EPI == 4 || EPI == 6) ? 1 : (EPI == 7 ? 2 : ((EPI == 8 || EPI == 9) ? 3 : 0)),
                                   EPI == 3 || EPI == 4 || EPI == 7, EPI != 9>(
        p, acc, stage_f, m0 + wm * 128, n0 + wn * 64, (m0 >> 8) * 2 + wm);
    return;
  }
  if constexpr (NPFC > 0)
    gemm_epilogue_wave128_pre<EPI, NPFC>(p, acc, stage_f, m0 + wm * 128, n0 + wn * 64, split, opf);
  else
    gemm_epilogue_wave128<EPI>(p, acc, stage_f, m0 + wm * 128, n0 + wn * 64, split);
}

// ---- gemm_pp64_kernel: the ring on 64-k blocks with WHOLE 128-byte lines per LDS-DMA row (round 6) --------------------------
// Same tile (256 x 256, eight waves as 2 x 4, 128 x 64 per wave, v_mfma_f32_32x32x16_bf16), same ping-pong of the two wave
// groups, same k order per accumulator (bit-identical to gemm_pp256_kernel) -- what changes is the operand stream: a piece is
// 256 rows x 64 k = 32 KB and every row of it is one whole 128-byte line (gemm_pp256_kernel's 32-k pieces fetch each line of
// A / B in two 64-byte halves, one K block apart: twice the requests on the per-CU L2 -> LDS path, which is what bounds the long-K
// products -- 1920 cycles per 32-k block against 1024 of MFMA; scripts/exp/gemm_probe3 measured the stream alone at 137 us with
// whole lines against 174).
//   LDS: FIVE 32 KB slots (160 KB, the whole CU), piece i (A of block kb = 2 kb, B = 2 kb + 1) in slot i % 5: while block kb
//   is read from two slots, pieces 2kb+2 .. 2kb+4 are landed / in flight in the other three.  Block kb issues piece 2kb+3 in its
//   phases 0, 1 and piece 2kb+4 in its phases 2, 3 (two instructions per wave per phase, as the 32-k ring); the slot of 2kb+3 held
//   A(kb-1), read last in phase 3 of block kb-1 -- that phase waits for its fragment reads before its first barrier, so the
//   restage is ordered behind them by a barrier whichever wave group issues it.  One counted wait per block (phase 3: vmcnt(4) =
//   piece 2kb+4 may stay in flight), the reads of block kb+1 come a phase later.
//   Row image: 128-byte pitch, 16-byte chunk c of row r at physical chunk c ^ ((r >> 1) & 7): the four 16-lane groups of a
//   ds_read_b128 (rows {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) then cover the sixteen slots of the 256-byte bank row once.
//   The DMA writes lanes linearly (8 lanes = one row), so the swizzle is applied to the SOURCE chunk: the eight lanes of a row
//   still fetch one whole line.
namespace pp64 {
constexpr int SLOT = 32768;
constexpr int NSLOT = 5;
}  // namespace pp64

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pp64_kernel(const GemmParams p, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = 256, TN_ = 256;
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;
  const int wm = grp, wn = wave & 3;
  const int64_t xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  // (work-item order of gemm_pp256_kernel's K-contiguous form: the N tiles of an M panel, and the K splits of a tile, are
  //  neighbours on one XCD)
  const int64_t tslot = slot_id / p.splits;
  const int split = (int)(slot_id - tslot * p.splits);
  const int64_t m_tile = (tslot / nt) * 8 + xcd;
  if (m_tile * TM >= p.m) return;
  const int64_t m0 = m_tile * TM, n0 = (tslot % nt) * TN_;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  const int64_t nb = (min(p.k, kbeg + p.k_per_split) - kbeg) / 64;     // 64-k blocks (the host guarantees >= 3, whole)

  // DMA sources: instruction q = wave*4 + i fills rows q*8 .. q*8+7 of a piece, lane = row*8 + physical chunk
  const char* ap[4];
  const char* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    ap[i] = p.a + (min(m0 + r, p.m - 1) * p.lda + kbeg) * 2 + c * 16;
    bp[i] = p.b + (min(n0 + r, p.n - 1) * p.ldb + kbeg) * 2 + c * 16;
  }
  const int dma_off = wave * 4096;      // + i*1024
  auto issue_one = [&](const char* (&src)[4], int slot, int i) {
    __builtin_amdgcn_global_load_lds((gptr)src[i], (lptr)(smem + slot * pp64::SLOT + dma_off + i * 1024), 16, 0, 0);
    src[i] += 128;
  };
  auto issue_half = [&](const char* (&src)[4], int slot, int half) {
    issue_one(src, slot, half * 2);
    issue_one(src, slot, half * 2 + 1);
  };
  // fragment addresses inside a slot (sub-phase hk: ^ hk*32)
  const int frow = lane & 31, fhalf = lane >> 5, key = (frow >> 1) & 7;
  const int a_lane = (wm * 128 + frow) * 128 + ((fhalf ^ key) << 4);
  const int b_lane = (wn * 64 + frow) * 128 + ((fhalf ^ key) << 4);
  auto load_frags = [&](int sa, int sb, int hk, u32x4(&fa)[4], u32x4(&fb)[2]) {
    const char* pa = smem + sa * pp64::SLOT;
    const char* pb = smem + sb * pp64::SLOT;
    const int x = hk * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x4*>(pb + (b_lane ^ x) + j * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(pa + (a_lane ^ x) + i * 4096);
  };

  f32x16 acc[2][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.0f;
  constexpr int NPFC = EPI == 2 ? 4 : EPI == 1 ? 2 : 0;
  constexpr int NPF = NPFC * (EPI == 1 ? 8 : 4);
  EpiOperands opf[4];
  auto mfma8 = [&](const u32x4(&fa)[4], const u32x4(&fb)[2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16& d = acc[i >> 1][i & 1][j];
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), d, 0,
                                                   0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // the compute segment with the two LDS-DMA instructions of this phase's half piece (`src` / `slot`) behind its 2nd and 5th
  // MFMA: in the load segment they (60-185 cycles each) made it the longer of the two segments -- h = x U 195 -> 189 us, the
  // cross product 440 -> 430; where among the MFMAs they sit does not matter (profiles/r6_gemm_k64_dma_in_compute_ab.txt)
  auto mfma8_dma = [&](const u32x4(&fa)[4], const u32x4(&fb)[2], const char* (&src)[4], int slot, int half) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16& d = acc[i >> 1][i & 1][j];
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), d, 0,
                                                   0, 0);
        if (i * 2 + j == 1) { __builtin_amdgcn_sched_barrier(0); issue_one(src, slot, half * 2); __builtin_amdgcn_sched_barrier(0); }
        if (i * 2 + j == 4) { __builtin_amdgcn_sched_barrier(0); issue_one(src, slot, half * 2 + 1); __builtin_amdgcn_sched_barrier(0); }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // one phase of a block that issues half a piece
  auto phase = [&](int sa, int sb, int hk, const char* (&src)[4], int slot, int half, bool last_of_block) {
    u32x4 fa[4], fb[2];
    load_frags(sa, sb, hk, fa, fb);
    if (last_of_block) {
      // pieces 2kb+2, 2kb+3 have landed (read from the next phase on); the first half of piece 2kb+4 may stay in flight (its
      // second half is issued behind this wait)
      pp_vmcnt<2>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this block's last reads are done before the barrier: its A slot is restaged next
    }
    pp_barrier();
    mfma8_dma(fa, fb, src, slot, half);
    pp_barrier();
  };
  auto step5 = [](int v, int by) { v += by; return v >= pp64::NSLOT ? v - pp64::NSLOT : v; };

  {
    // prologue: pieces 0 (A0), 1 (B0), 2 (A1) whole; blocks 0's operands must have landed, piece 2 may fly on
    issue_half(ap, 0, 0); issue_half(ap, 0, 1);
    issue_half(bp, 1, 0); issue_half(bp, 1, 1);
    issue_half(ap, 2, 0); issue_half(ap, 2, 1);
    pp_vmcnt<4>();
    pp_barrier();
    if (grp == 1) pp_barrier();  // rows 128..255 run one segment behind rows 0..127
    int sa = 0, sb = 1, s3 = 3, s4 = 4;   // slots of pieces 2kb, 2kb+1, 2kb+3, 2kb+4
    int64_t kb = 0;
    for (; kb + 2 < nb; ++kb) {   // steady state: pieces 2kb+3 (B of kb+1) and 2kb+4 (A of kb+2) are issued
      phase(sa, sb, 0, bp, s3, 0, false);
      phase(sa, sb, 1, bp, s3, 1, false);
      phase(sa, sb, 2, ap, s4, 0, false);
      phase(sa, sb, 3, ap, s4, 1, true);
      sa = step5(sa, 2); sb = step5(sb, 2); s3 = step5(s3, 2); s4 = step5(s4, 2);
    }
    // block nb-2: only B(nb-1) = piece 2kb+3 is left to issue; the epilogue's operands are requested behind it
    {
      phase(sa, sb, 0, bp, s3, 0, false);
      phase(sa, sb, 1, bp, s3, 1, false);
      u32x4 fa[4], fb[2];
      load_frags(sa, sb, 2, fa, fb);
      if constexpr (NPFC > 0) {
#pragma unroll
        for (int c = 0; c < NPFC; ++c) epi_prefetch<EPI>(p, opf[c], m0 + wm * 128 + c * 32, n0 + wn * 64);
      }
      pp_barrier();
      mfma8(fa, fb);
      pp_barrier();
      load_frags(sa, sb, 3, fa, fb);
      pp_vmcnt<NPF>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_barrier();
      mfma8(fa, fb);
      pp_barrier();
      sa = step5(sa, 2); sb = step5(sb, 2);
    }
    // block nb-1: nothing in flight but the epilogue's operands
#pragma unroll
    for (int hk = 0; hk < 4; ++hk) {
      u32x4 fa[4], fb[2];
      load_frags(sa, sb, hk, fa, fb);
      pp_barrier();
      mfma8(fa, fb);
      pp_barrier();
    }
    if (grp == 0) pp_barrier();
  }
  lds_dma_retired<NPF>();
  float* stage_f = reinterpret_cast<float*>(smem) + wave * (32 * 68);
  if constexpr (EPI >= 3) {
    gemm_epilogue_wave128_crossbwd<(EPI == 4 || EPI == 6) ? 1 : (EPI == 7 ? 2 : ((EPI == 8 || EPI == 9) ? 3 : 0)),
                                   EPI == 3 || EPI == 4 || EPI == 7, EPI != 9>(
        p, acc, stage_f, m0 + wm * 128, n0 + wn * 64, (m0 >> 8) * 2 + wm);
    return;
  }
  if constexpr (NPFC > 0)
    gemm_epilogue_wave128_pre<EPI, NPFC>(p, acc, stage_f, m0 + wm * 128, n0 + wn * 64, split, opf);
  else
    gemm_epilogue_wave128<EPI>(p, acc, stage_f, m0 + wm * 128, n0 + wn * 64, split);
}

// Weight gradients with one tiny dimension (C = A^T B, both operands K-strided, min(M, N) <= 16, long K):
// the first Dense of the bottom MLP (13 dense features) and the last of the top MLP (1 unit).  The
// MFMA tiles do not apply and one thread per output would walk K = batch alone; here a thread owns
// one column of the WIDE operand (coalesced across the workgroup), the THIN operand's rows are
// staged in LDS and broadcast, K is split over blockIdx.y into fp32 slabs (fixed-order reduce).
constexpr int kThinMax = 16;
// ES: operand element size (2 = bf16, 4 = fp32); NT: thin extent rounded up to 1 / 4 / 8 / 16 (the LDS rows are
// read as float4).  Eight rows of the wide operand are requested before they are consumed (the first version
// walked its 1024 rows one dependent load at a time behind a run-time dtype switch: 394 us for the 13 x 512 gradient).
template <int ES, int NT>
__global__ __launch_bounds__(256) void gemm_thin_kernel(const GemmParams p, int thin_is_a) {
  typedef typename std::conditional<ES == 2, uint16_t, float>::type elem_t;
  __shared__ __attribute__((aligned(16))) float thin[128][NT];
  const elem_t* wide_p = reinterpret_cast<const elem_t*>(thin_is_a ? p.b : p.a);
  const elem_t* thin_p = reinterpret_cast<const elem_t*>(thin_is_a ? p.a : p.b);
  const int64_t ldw = thin_is_a ? p.ldb : p.lda, ldt = thin_is_a ? p.lda : p.ldb;
  const int64_t n_wide = thin_is_a ? p.n : p.m;
  const int n_thin = (int)(thin_is_a ? p.m : p.n);
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ccol = col < n_wide ? col : n_wide - 1;   // idle threads load a valid column and store nothing
  const int split = blockIdx.y;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  const int64_t kend = min(p.k, kbeg + p.k_per_split);
  auto cvt = [](elem_t v) -> float {
    if constexpr (ES == 2) return bf16_to_f32(v);
    else return v;
  };
  float acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = 0.0f;
  for (int64_t k0 = kbeg; k0 < kend; k0 += 128) {
    const int rows = (int)min<int64_t>(128, kend - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < 128 * NT; e += 256) {
      const int r = e / NT, i = e % NT;
      thin[r][i] = (r < rows && i < n_thin) ? cvt(thin_p[(k0 + r) * ldt + i]) : 0.0f;
    }
    __syncthreads();
    for (int r = 0; r < rows; r += 8) {
      elem_t wv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) wv[q] = wide_p[min(k0 + r + q, kend - 1) * ldw + ccol];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float w = r + q < rows ? cvt(wv[q]) : 0.0f;
        if constexpr (NT == 1) {
          acc[0] = fmaf(thin[r + q][0], w, acc[0]);
        } else {
#pragma unroll
          for (int i4 = 0; i4 < NT / 4; ++i4) {
            const float4 t = *reinterpret_cast<const float4*>(&thin[(r + q) & 127][i4 * 4]);
            acc[i4 * 4 + 0] = fmaf(t.x, w, acc[i4 * 4 + 0]);
            acc[i4 * 4 + 1] = fmaf(t.y, w, acc[i4 * 4 + 1]);
            acc[i4 * 4 + 2] = fmaf(t.z, w, acc[i4 * 4 + 2]);
            acc[i4 * 4 + 3] = fmaf(t.w, w, acc[i4 * 4 + 3]);
          }
        }
      }
    }
  }
  if (col >= n_wide) return;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (i < n_thin) {
      const int64_t m = thin_is_a ? i : col, n = thin_is_a ? col : i;
      if (p.splits > 1) p.slabs[((int64_t)split * p.m + m) * p.n + n] = acc[i];
      else epilogue_store(p, m, n, acc[i]);
    }
  }
}

// Products with ONE tiny dimension on the output side or in the contraction, row-major A (the last Dense of the
// DLRM top MLP: 256 -> 1 unit, forward and data gradient; one thread per output element with 64-bit divisions
// took 235 us for either):
//   gemm_rowdot_kernel  N <= 8, any K: 16 lanes per row of A, each lane walks its 16-byte chunks of the row and
//                       keeps N partial dots; butterfly reduction; lane 0 stores through the epilogue
//   gemm_smallk_kernel  K <= 16, N a multiple of 8: a thread owns 8 consecutive outputs of one row
template <int ES>
__global__ __launch_bounds__(256) void gemm_rowdot_kernel(const GemmParams p) {
  typedef typename std::conditional<ES == 2, uint16_t, float>::type elem_t;
  constexpr int VE = 16 / ES;
  const int sub = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t r = row < p.m ? row : p.m - 1;
  const elem_t* a = reinterpret_cast<const elem_t*>(p.a) + r * p.lda;
  const elem_t* b = reinterpret_cast<const elem_t*>(p.b);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  for (int64_t k0 = (int64_t)sub * VE; k0 < p.k; k0 += 16 * VE) {
    float av[VE];
    if (k0 + VE <= p.k && (reinterpret_cast<uintptr_t>(a + k0) & 15) == 0) {
      const uint4 raw = *reinterpret_cast<const uint4*>(a + k0);
      if constexpr (ES == 2) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[2 * q] = __uint_as_float(w[q] << 16); av[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
      } else {
        av[0] = __uint_as_float(raw.x); av[1] = __uint_as_float(raw.y); av[2] = __uint_as_float(raw.z); av[3] = __uint_as_float(raw.w);
      }
    } else {
#pragma unroll
      for (int q = 0; q < VE; ++q) {
        if constexpr (ES == 2) av[q] = k0 + q < p.k ? bf16_to_f32(a[k0 + q]) : 0.0f;
        else av[q] = k0 + q < p.k ? a[k0 + q] : 0.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < p.n) {
#pragma unroll
        for (int q = 0; q < VE; ++q) {
          if (k0 + q < p.k) {
            const elem_t bv = p.b_nk ? b[(int64_t)j * p.ldb + k0 + q] : b[(k0 + q) * p.ldb + j];
            float bf;
            if constexpr (ES == 2) bf = bf16_to_f32(bv); else bf = bv;
            acc[j] = fmaf(av[q], bf, acc[j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int o = 8; o; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
  if (sub == 0 && row < p.m)
    for (int j = 0; j < p.n; ++j) epilogue_store(p, row, j, acc[j]);
}

template <int ES>
__global__ __launch_bounds__(256) void gemm_smallk_kernel(const GemmParams p, int rows_per_wg) {
  typedef typename std::conditional<ES == 2, uint16_t, float>::type elem_t;
  extern __shared__ __attribute__((aligned(16))) float bs[];   // B as fp32 [K][N]: a thread reads its 8 columns as 2 x float4
  auto f = [](elem_t v) -> float {
    if constexpr (ES == 2) return bf16_to_f32(v);
    else return v;
  };
  const elem_t* b = reinterpret_cast<const elem_t*>(p.b);
  const int K = (int)p.k, N = (int)p.n;
  for (int e = threadIdx.x; e < K * N; e += 256) {
    const int kk = e / N, j = e - kk * N;
    bs[e] = f(p.b_nk ? b[(int64_t)j * p.ldb + kk] : b[(int64_t)kk * p.ldb + j]);
  }
  __syncthreads();
  const int n8 = N / 8;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t items = (int64_t)min<int64_t>(rows_per_wg, p.m - row0) * n8;
  for (int64_t it = threadIdx.x; it < items; it += 256) {
    const int64_t i = row0 + it / n8;
    const int j0 = (int)(it % n8) * 8;
    const elem_t* a = reinterpret_cast<const elem_t*>(p.a) + i * p.lda;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    for (int kk = 0; kk < K; ++kk) {
      const float av = f(a[kk]);
      const float4 b0 = *reinterpret_cast<const float4*>(bs + kk * N + j0);
      const float4 b1 = *reinterpret_cast<const float4*>(bs + kk * N + j0 + 4);
      v[0] = fmaf(av, b0.x, v[0]); v[1] = fmaf(av, b0.y, v[1]); v[2] = fmaf(av, b0.z, v[2]); v[3] = fmaf(av, b0.w, v[3]);
      v[4] = fmaf(av, b1.x, v[4]); v[5] = fmaf(av, b1.y, v[5]); v[6] = fmaf(av, b1.z, v[6]); v[7] = fmaf(av, b1.w, v[7]);
    }
    if (p.ep_vec) {
      epilogue_store_vec8(p, i, j0, v);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) epilogue_store(p, i, j0 + j, v[j]);
    }
  }
}

// fixed-order reduction of the split-K slabs + epilogue
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const GemmParams p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.m * p.n) return;
  float v = 0.0f;
  for (int s = 0; s < p.splits; ++s) v += p.slabs[(int64_t)s * p.m * p.n + idx];
  epilogue_store(p, idx / p.n, idx % p.n, v);
}
// the common weight-gradient case (fp32 C, no epilogue, N % 4 == 0): four columns per thread, same order
__global__ __launch_bounds__(256) void gemm_slab_reduce_vec4_kernel(const GemmParams p) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;   // group of 4 consecutive columns
  const int64_t nq = p.n / 4;
  if (q >= p.m * nq) return;
  const int64_t i = q / nq, j = (q - i * nq) * 4;
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int s = 0; s < p.splits; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(p.slabs + ((int64_t)s * p.m + i) * p.n + j);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.c) + i * p.ldc + j) = v;
}

// eight columns per thread + the vector epilogue (bf16 or fp32 output, bias / activation / cross / residual forms): the
// reduce of the K-contiguous split products (round 5), same slab order
__global__ __launch_bounds__(256) void gemm_slab_reduce_vec8_kernel(const GemmParams p) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;   // group of 8 consecutive columns
  const int64_t nq = p.n / 8;
  if (q >= p.m * nq) return;
  const int64_t i = q / nq, j = (q - i * nq) * 8;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 0.0f;
  for (int s = 0; s < p.splits; ++s) {
    const float* src = p.slabs + ((int64_t)s * p.m + i) * p.n + j;
    const float4 t0 = *reinterpret_cast<const float4*>(src), t1 = *reinterpret_cast<const float4*>(src + 4);
    v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
  }
  epilogue_store_vec8(p, i, j, v);
}

// any shape / alignment: one thread per output element
__global__ __launch_bounds__(256) void gemm_generic_kernel(const GemmParams p, int in_dtype) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.m * p.n) return;
  const int64_t i = idx / p.n, j = idx % p.n;
  float acc = 0.0f;
  for (int64_t kk = 0; kk < p.k; ++kk) {
    const float av = p.a_km ? ld_elem(p.a, in_dtype, kk * p.lda + i) : ld_elem(p.a, in_dtype, i * p.lda + kk);
    const float bv = p.b_nk ? ld_elem(p.b, in_dtype, j * p.ldb + kk) : ld_elem(p.b, in_dtype, kk * p.ldb + j);
    acc = fmaf(av, bv, acc);
  }
  epilogue_store(p, i, j, acc);
}

// Main loop of the big bf16 shapes: 4 = the ping-pong rings on 256x256 tiles (default: gemm_pp64_kernel where K is whole 64-k
// blocks, gemm_pp256_kernel otherwise and for the weight gradients); 5 = gemm_pp256_kernel for all of them (A/B of the two);
// 0 = the two-stage 128x128 kernels for every shape (gemm_glds_kernel / gemm_tn_glds_kernel / gemm_mfma_kernel) and the
// two-call form of krs_gemm_cross_bwd -- the reference schedule for A/B and bit-for-bit tests.
// krs_gemm_set_option(KRS_GEMM_OPT_PIPELINE, v) / environment KRS_GEMM_PIPE (read once).
int g_pipe = -1;
int gemm_pipe() {
  if (g_pipe < 0) {
    const char* e = getenv("KRS_GEMM_PIPE");
    g_pipe = e ? atoi(e) : 4;
    if (g_pipe != 0 && g_pipe != 5) g_pipe = 4;
  }
  return g_pipe;
}

// Split-K factor.  Weight-gradient shapes that take the 256x256 tiles (a_is_km, M, N >= 256, long K): the
// factor that minimises  rounds over the 256 CUs x (K per split + per-workgroup overhead) + slab traffic
// -- 2 x 14 tiles of the C3 weight gradients: 9 splits = 252 workgroups in ONE round (16 splits were 448
// workgroups = 1.75 rounds, and 113 MB of slabs instead of 64).
// `ring_ok`: the product can take the ring kernel's split-K form (bf16, B as [N, K]) -- the third branch hands
// out splits for that kernel only; the workspace query leaves it true (an upper bound for every dtype / layout).
int pick_splits(int64_t m, int64_t n, int64_t k, int a_is_km, bool ring_ok = true) {
  if (a_is_km && k >= 1024 && std::min(m, n) <= 16) {   // gemm_thin_kernel: >= 512 rows of K per split, <= 128 splits
    const int64_t s = k / 512;                            // (256 splits made the slab reduction the longer kernel,
    return (int)(s > 128 ? 128 : s);                      //  64 left half of the CUs without a workgroup)
  }
  if (a_is_km && m >= 256 && n >= 256 && k % 64 == 0 && k >= 4096) {
    const int64_t t256 = ceil_div(m, 256) * ceil_div(n, 256);
    const double slab = (double)m * (double)n * 6.45e-5;   // slab write + read of one split, in units of one k step of a tile
    double best = 0;
    int best_s = 1;
    for (int s = 1; s <= 64 && k / s >= 512; ++s) {
      const int64_t kps = ceil_div(ceil_div(k, s), 64) * 64;
      if ((int64_t)(s - 1) * kps >= k) continue;          // the last split would be empty
      const double cost = (double)ceil_div(t256 * s, 256) * (double)(kps + 256) + (s > 1 ? slab * s : 0.0);
      if (s == 1 || cost < best) { best = cost; best_s = s; }
    }
    return best_s;
  }
  // K-contiguous products whose output is too small for 256x256 tiles to fill the chip (M = 8192 against N = 512: 64 tiles)
  // but whose K is long: the ring kernel with the K range dealt to s workgroups per tile -- half the operand bytes per flop
  // of the 128x128 kernel that ran these shapes until round 5, at the price of s fp32 slabs (h = x U at M = 8192: 59 -> 4x us,
  // DESIGN.md section 4).  s * tiles ~ 256 workgroups, >= 512 of K per split, whole 32-k blocks.
  if (ring_ok && !a_is_km && m >= 256 && n >= 256 && k >= 2048 && k % 32 == 0) {
    const int64_t t256 = ceil_div(m, 256) * ceil_div(n, 256);
    if (t256 < 192) {
      // (krs_gemm rounds the K per split up to whole 64-k tiles: the last split takes what is left, and must still hold
      //  a ring's depth of 32-k blocks)
      int64_t s = 256 / t256;
      auto fits = [&](int64_t q) {
        const int64_t kps = ceil_div(ceil_div(k, q), 64) * 64, last = k - (q - 1) * kps;
        return k / q >= 512 && last >= 128 && last % 32 == 0;
      };
      while (s > 1 && !fits(s)) --s;
      if (s > 1) return (int)s;
    }
  }
  const int64_t tiles = ceil_div(m, BM) * ceil_div(n, BN);
  if (tiles >= 256 || k < 4096) return 1;
  int64_t s = ceil_div(1024, tiles);       // aim at ~4 workgroups per CU
  const int64_t max_s = k / 1024;          // keep >= 1024 of K per split
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  if (s > 8) s = (s + 7) / 8 * 8;          // whole rounds over the 8 XCDs (gemm_tn_glds_kernel deals splits to XCDs)
  if (s > max_s) s = max_s / 8 * 8;
  return s < 1 ? 1 : (int)s;
}

bool mfma_eligible(const GemmParams& p, int es) {
  const int64_t va = 16 / es;  // elements per 16-byte vector
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al(p.a) || !al(p.b)) return false;
  if (p.lda % va || p.ldb % va) return false;
  // the vector-loaded axis must be a multiple of the vector length
  if (p.a_km ? (p.m % va) : (p.k % va)) return false;
  if (p.b_nk ? (p.k % va) : (p.n % va)) return false;
  return p.m >= 1 && p.n >= 1 && p.k >= 1;
}

template <int ES>
int launch_mfma(const GemmParams& p, hipStream_t st) {
  const int64_t mt = p.a_km ? ceil_div(p.m, BM) : ceil_div(ceil_div(p.m, BM), 8) * 8;
  const dim3 grid((unsigned)(mt * ceil_div(p.n, BN)), 1, (unsigned)p.splits);
  const size_t lds = 2 * TILE_BYTES;
  // specialised epilogues: bf16 output, vector access everywhere, no split-K
  int epi = 0;
  if (p.has_ep && p.ep_vec && p.splits == 1 && p.out_dtype == KRS_BF16 && p.n >= 8) {
    if (p.ep.x0 && !p.ep.r) epi = 1;
    else if (p.ep.r && !p.ep.x0 && !p.ep.bias && p.ep.act == KRS_ACT_NONE) epi = 2;
  }
#define KRS_GEMM_LAUNCH(AK, BK_, EP)                                                                \
  {                                                                                                 \
    auto kern = gemm_mfma_kernel<ES, AK, BK_, EP>;                                                  \
    static bool attr_set = false;                                                                   \
    if (!attr_set) {                                                                                \
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
      attr_set = true;                                                                              \
    }                                                                                               \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);                                          \
  }
#define KRS_GEMM_CASE(AK, BK_)                                                                      \
  {                                                                                                 \
    if (epi == 1) KRS_GEMM_LAUNCH(AK, BK_, 1)                                                       \
    else if (epi == 2) KRS_GEMM_LAUNCH(AK, BK_, 2)                                                  \
    else KRS_GEMM_LAUNCH(AK, BK_, 0)                                                                \
  }
  if (p.a_km && p.b_nk) return fail(KRS_ERR_UNSUPPORTED, "krs_gemm: A^T . B^T layout is not used by the layer");
  // Long contractions: the LDS-DMA pipeline (no staging registers, no ds_write traffic).  Short ones
  // (K = 512: eight tiles, then a heavy epilogue) run better on the register-staged kernel, whose
  // 36 KB of LDS lets three workgroups share a CU and hide each other's epilogues.
  const bool dma_ok = !p.a_km && p.b_nk && p.splits == 1 && p.k % (ROW_BYTES / ES) == 0;
  // split-K on the ring kernel (round 5): K-contiguous bf16 operands, every split a whole number of 32-k blocks and at
  // least a ring's depth of them (pick_splits' nt branch hands out exactly such splits)
  const bool ring_split = ES == 2 && !p.a_km && p.b_nk && p.splits > 1 && p.k % 32 == 0 && p.k_per_split % 32 == 0 &&
                          p.k - (int64_t)(p.splits - 1) * p.k_per_split >= 128 && gemm_pipe() != 0;
  // development switch: 128 x 128 tiles at two workgroups per CU for every LDS-DMA shape (y = cross(h V): 474 us against 429)
  static const bool force128 = getenv("KRS_GEMM_FORCE128") != nullptr;
  const bool use_glds = dma_ok && (p.k >= 1024 || force128);
  // ... provided its 4x larger tiles still cover most of the 256 CUs (a per-rank batch of 8192 rows against
  // N = 512 is 64 such tiles: the 128x128 kernels below launch 256 workgroups instead)
  const bool fills256 = ceil_div(p.m, 256) * ceil_div(p.n, 256) >= 192;
  if constexpr (ES == 2) {
    // the ring kernel (gemm_pp256_kernel); krs_gemm_set_option(KRS_GEMM_OPT_PIPELINE, 0) sends these shapes to the 128x128
    // two-stage kernels below instead (same fragment layout, same k order per accumulator: bit-identical results -- the
    // reference schedule of tests/test_dense_ops_gpu.py and scripts/exp/gemm_bench)
    if (((dma_ok && fills256) || ring_split) && gemm_pipe() != 0 && p.k >= 256 && p.k % 32 == 0 && p.m >= 256 && p.n >= 256 &&
        !force128) {
      const dim3 grid256((unsigned)(ceil_div(ceil_div(p.m, 256), 8) * 8 * ceil_div(p.n, 256) * p.splits));
      const int nt_ = (int)ceil_div(p.n, 256);
      // (the residual-add form with a short K -- dx = dh U^T + g -- was 4 % faster on a two-stage loop until its R
      // operands were fetched under the ring's tail: 340 -> 301 us)
#define KRS_PP_LAUNCH(EP)                                                                            \
  {                                                                                                  \
    auto kern = gemm_pp256_kernel<false, 4, EP>;                                                     \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pp::STAGE));       \
      attr_set = true;                                                                               \
    }                                                                                                \
    hipLaunchKernelGGL(kern, grid256, dim3(512), 4 * pp::STAGE, st, p, 0, nt_);                      \
  }
      // whole 64-k blocks of both operands and of every split, at least three of them -> the 64-k ring (pipeline 5: never)
      const bool k64 = gemm_pipe() == 4 && p.k % 64 == 0 && p.k_per_split % 64 == 0 &&
                       p.k - (int64_t)(p.splits - 1) * p.k_per_split >= 192;
#define KRS_PP64_LAUNCH(EP)                                                                          \
  {                                                                                                  \
    auto kern = gemm_pp64_kernel<EP>;                                                                \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pp64::NSLOT * pp64::SLOT)); \
      attr_set = true;                                                                               \
    }                                                                                                \
    hipLaunchKernelGGL(kern, grid256, dim3(512), pp64::NSLOT * pp64::SLOT, st, p, nt_);              \
  }
      if (k64) {
        if (epi == 1) KRS_PP64_LAUNCH(1)
        else if (epi == 2) KRS_PP64_LAUNCH(2)
        else KRS_PP64_LAUNCH(0)
        KRS_CHECK_LAUNCH("gemm_pp64_kernel");
        return KRS_OK;
      }
#undef KRS_PP64_LAUNCH
      if (epi == 1) KRS_PP_LAUNCH(1)
      else if (epi == 2) KRS_PP_LAUNCH(2)
      else KRS_PP_LAUNCH(0)
#undef KRS_PP_LAUNCH
      KRS_CHECK_LAUNCH("gemm_pp256_kernel");
      return KRS_OK;
    }
  }
  if (use_glds) {
    const size_t glds_lds = 4 * BM * ROW_BYTES;  // 2 stages x (A + B) x 16 KB
#define KRS_GLDS_LAUNCH(EP)                                                                          \
  {                                                                                                  \
    auto kern = gemm_glds_kernel<ES, EP>;                                                            \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds_lds));       \
      attr_set = true;                                                                               \
    }                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), glds_lds, st, p);                                      \
  }
    if (epi == 1) KRS_GLDS_LAUNCH(1)
    else if (epi == 2) KRS_GLDS_LAUNCH(2)
    else KRS_GLDS_LAUNCH(0)
#undef KRS_GLDS_LAUNCH
    KRS_CHECK_LAUNCH("gemm_glds_kernel");
    return KRS_OK;
  }
  // bf16 weight gradients: LDS-DMA + transposing reads (K extents in whole 64-row tiles, >= 8 columns)
  if (ES == 2 && p.a_km && !p.b_nk && p.k % 64 == 0 && p.k_per_split % 64 == 0 && p.m >= 8 && p.n >= 8 &&
      p.m % 8 == 0 && p.n % 8 == 0) {
    // development switch: the 128 x 128 twin (4 waves, <= 128 VGPRs, 64 KB of LDS: half a CU) for every shape
    static const bool tn128 = getenv("KRS_GEMM_TN128") != nullptr;
    if (p.m >= 256 && p.n >= 256 && !tn128 && gemm_pipe() != 0 && p.k_per_split >= 256) {
      const int mt_ = (int)ceil_div(p.m, 256), nt_ = (int)ceil_div(p.n, 256);
      const dim3 grid_tn((unsigned)(ceil_div((int64_t)p.splits * mt_ * nt_, 8) * 8));
      auto kern = gemm_pp256_kernel<true, 4, 0>;
      static bool attr_set = false;
      if (!attr_set) {
        KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * pp::STAGE));
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, grid_tn, dim3(512), 4 * pp::STAGE, st, p, mt_, nt_);
      KRS_CHECK_LAUNCH("gemm_pp256_kernel (K-strided operands)");
      return KRS_OK;
    }
    const int mt_ = (int)ceil_div(p.m, BM), nt_ = (int)ceil_div(p.n, BN);
    const dim3 grid_tn((unsigned)(ceil_div(p.splits, 8) * 8 * mt_ * nt_));
    const size_t lds_tn = 4 * 64 * 128 * 2;  // 2 stages x (A + B) x 16 KB
    auto kern = gemm_tn_glds_kernel<0>;
    static bool attr_set = false;
    if (!attr_set) {
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_tn));
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid_tn, dim3(256), lds_tn, st, p, mt_, nt_);
    KRS_CHECK_LAUNCH("gemm_tn_glds_kernel");
    return KRS_OK;
  }
  if (p.a_km) KRS_GEMM_CASE(true, false)
  else if (p.b_nk) KRS_GEMM_CASE(false, true)
  else KRS_GEMM_CASE(false, false)
#undef KRS_GEMM_LAUNCH
#undef KRS_GEMM_CASE
  KRS_CHECK_LAUNCH("gemm_mfma_kernel");
  return KRS_OK;
}

}  // namespace
}  // namespace krs

using namespace krs;

extern "C" int krs_gemm_set_option(int key, int value) {
  if (key == KRS_GEMM_OPT_PIPELINE) {
    KRS_REQUIRE(value == 0 || value == 4 || value == 5, "krs_gemm_set_option: pipeline must be 0, 4 or 5");
    g_pipe = value;
    return KRS_OK;
  }
  return fail(KRS_ERR_INVALID, "krs_gemm_set_option: unknown key %d", key);
}

extern "C" size_t krs_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k, int a_is_km) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const int s = pick_splits(m, n, k, a_is_km);
  return s > 1 ? (size_t)s * (size_t)m * (size_t)n * sizeof(float) : 0;
}

namespace krs {
namespace {
// krs_gemm's body.  `allow_split` false = one pass over K whatever pick_splits would choose (the two-call form of
// krs_gemm_cross_bwd, whose workspace is sized for the column sums only).
int gemm_run(const void* a, int64_t lda, int a_is_km, const void* b, int64_t ldb, int b_is_nk,
             void* c, int64_t ldc, int64_t m, int64_t n, int64_t k, int in_dtype, int out_dtype,
             const krs_gemm_epilogue* epilogue, void* workspace, size_t workspace_bytes,
             void* stream, bool allow_split) {
  KRS_REQUIRE(a && b && c, "krs_gemm: null operand");
  KRS_REQUIRE(m >= 0 && n >= 0 && k >= 0, "krs_gemm: negative size");
  KRS_REQUIRE((in_dtype == KRS_F32 || in_dtype == KRS_BF16) && (out_dtype == KRS_F32 || out_dtype == KRS_BF16),
              "krs_gemm: bad dtype");
  if (epilogue && epilogue->x0) KRS_REQUIRE(epilogue->x, "krs_gemm: cross epilogue needs x with x0");
  if (m == 0 || n == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GemmParams p{};
  p.a = reinterpret_cast<const char*>(a); p.lda = lda; p.a_km = a_is_km != 0;
  p.b = reinterpret_cast<const char*>(b); p.ldb = ldb; p.b_nk = b_is_nk != 0;
  p.c = reinterpret_cast<char*>(c); p.ldc = ldc; p.m = m; p.n = n; p.k = k; p.out_dtype = out_dtype;
  p.has_ep = epilogue != nullptr;
  if (epilogue) p.ep = *epilogue; else memset(&p.ep, 0, sizeof(p.ep));
  p.splits = 1; p.k_per_split = k; p.slabs = nullptr;
  {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool v = n % 8 == 0 && ldc % 8 == 0 && al16(c);
    if (epilogue) {
      if (epilogue->bias) v = v && al16(epilogue->bias);
      if (epilogue->x0) v = v && al16(epilogue->x0) && al16(epilogue->x) && epilogue->ldx % 8 == 0;
      if (epilogue->u_out) v = v && al16(epilogue->u_out) && epilogue->ldu % 8 == 0;
      if (epilogue->r) v = v && al16(epilogue->r) && epilogue->ldr % 8 == 0;
    }
    p.ep_vec = v;
  }
  const int es = in_dtype == KRS_BF16 ? 2 : 4;
  if (k > 0 && mfma_eligible(p, es) && !(p.a_km && p.b_nk)) {
    const int s = allow_split ? pick_splits(m, n, k, a_is_km, es == 2 && b_is_nk) : 1;
    if (s > 1) {
      const size_t need = (size_t)s * m * n * sizeof(float);
      if (!workspace || workspace_bytes < need)
        return fail(KRS_ERR_WORKSPACE, "krs_gemm: split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
      const int64_t bk = ROW_BYTES / es;
      p.splits = s;
      p.k_per_split = ceil_div(ceil_div(k, s), bk) * bk;
      p.slabs = reinterpret_cast<float*>(workspace);
    }
    const int rc = es == 2 ? launch_mfma<2>(p, st) : launch_mfma<4>(p, st);
    if (rc != KRS_OK) return rc;
    if (p.splits > 1) {
      const bool vec4 = !p.has_ep && out_dtype == KRS_F32 && n % 4 == 0 && ldc % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(c) & 15) == 0;
      if (vec4)
        hipLaunchKernelGGL(gemm_slab_reduce_vec4_kernel, dim3((unsigned)ceil_div(m * (n / 4), 256)), dim3(256), 0, st, p);
      else if (p.ep_vec && n % 8 == 0)
        hipLaunchKernelGGL(gemm_slab_reduce_vec8_kernel, dim3((unsigned)ceil_div(m * (n / 8), 256)), dim3(256), 0, st, p);
      else
        hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)ceil_div(m * n, 256)), dim3(256), 0, st, p);
      KRS_CHECK_LAUNCH("gemm_slab_reduce_kernel");
    }
    return KRS_OK;
  }
  if (p.a_km && !p.b_nk && k >= 1024 && std::min(m, n) <= kThinMax) {
    const int s = pick_splits(m, n, k, a_is_km);
    if (s > 1 && workspace && workspace_bytes >= (size_t)s * m * n * sizeof(float)) {
      p.splits = s;
      p.k_per_split = ceil_div(k, s);
      p.slabs = reinterpret_cast<float*>(workspace);
    }
    const int thin_is_a = m <= n;
    const int64_t n_wide = thin_is_a ? n : m;
    const int64_t n_thin = thin_is_a ? m : n;
    const dim3 tgrid((unsigned)ceil_div(n_wide, 256), (unsigned)p.splits);
#define KRS_THIN(ES_, NT_) hipLaunchKernelGGL((gemm_thin_kernel<ES_, NT_>), tgrid, dim3(256), 0, st, p, thin_is_a)
#define KRS_THIN_NT(ES_)                                          \
  {                                                               \
    if (n_thin <= 1) KRS_THIN(ES_, 1);                            \
    else if (n_thin <= 4) KRS_THIN(ES_, 4);                       \
    else if (n_thin <= 8) KRS_THIN(ES_, 8);                       \
    else KRS_THIN(ES_, 16);                                       \
  }
    if (in_dtype == KRS_BF16) KRS_THIN_NT(2) else KRS_THIN_NT(4)
#undef KRS_THIN_NT
#undef KRS_THIN
    KRS_CHECK_LAUNCH("gemm_thin_kernel");
    if (p.splits > 1) {
      hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)ceil_div(m * n, 256)), dim3(256), 0, st, p);
      KRS_CHECK_LAUNCH("gemm_slab_reduce_kernel");
    }
    return KRS_OK;
  }
  if (!p.a_km && p.splits == 1 && n <= 8 && k >= 32 && m >= 1024) {
    if (in_dtype == KRS_BF16) hipLaunchKernelGGL(gemm_rowdot_kernel<2>, dim3((unsigned)ceil_div(m, 16)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gemm_rowdot_kernel<4>, dim3((unsigned)ceil_div(m, 16)), dim3(256), 0, st, p);
    KRS_CHECK_LAUNCH("gemm_rowdot_kernel");
    return KRS_OK;
  }
  if (!p.a_km && p.splits == 1 && k <= 16 && n % 8 == 0 && n <= 1024 && m * n >= (1 << 16)) {
    const int rows_per_wg = 64;
    const unsigned blocks = (unsigned)ceil_div(m, rows_per_wg);
    const size_t lds = (size_t)k * n * sizeof(float);   // <= 64 KB
    if (in_dtype == KRS_BF16) hipLaunchKernelGGL(gemm_smallk_kernel<2>, dim3(blocks), dim3(256), lds, st, p, rows_per_wg);
    else hipLaunchKernelGGL(gemm_smallk_kernel<4>, dim3(blocks), dim3(256), lds, st, p, rows_per_wg);
    KRS_CHECK_LAUNCH("gemm_smallk_kernel");
    return KRS_OK;
  }
  hipLaunchKernelGGL(gemm_generic_kernel, dim3((unsigned)ceil_div(m * n, 256)), dim3(256), 0, st, p, in_dtype);
  KRS_CHECK_LAUNCH("gemm_generic_kernel");
  return KRS_OK;
}
}  // namespace
}  // namespace krs

extern "C" int krs_gemm(const void* a, int64_t lda, int a_is_km, const void* b, int64_t ldb, int b_is_nk,
                        void* c, int64_t ldc, int64_t m, int64_t n, int64_t k, int in_dtype, int out_dtype,
                        const krs_gemm_epilogue* epilogue, void* workspace, size_t workspace_bytes,
                        void* stream) {
  return gemm_run(a, lda, a_is_km, b, ldb, b_is_nk, c, ldc, m, n, k, in_dtype, out_dtype, epilogue, workspace,
                  workspace_bytes, stream, true);
}

// ---- krs_gemm_cross_bwd: data-gradient product + the elementwise backward of the layer below, one launch ---------------
extern "C" size_t krs_gemm_cross_bwd_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  return std::max(krs_colsum_workspace_bytes(m, n), (size_t)(2 * ceil_div(m, 256)) * (size_t)n * sizeof(float));
}

extern "C" int krs_gemm_cross_bwd(const void* a, int64_t lda, const void* bt, int64_t ldb, const void* r, int64_t ldr,
                                  float beta, void* g_out, int64_t ldg, const void* x0, const void* u, void* dz,
                                  void* dx0, int64_t ld, int dx0_accumulate, const void* u_upper, int fold_direct,
                                  float* dbias, int64_t m, int64_t n, int64_t k, int act, int dtype, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  const bool dense_form = x0 == nullptr;    // the layer below is a Dense layer: dz = G act'(u) (u = its saved output), dbias
  KRS_REQUIRE(a && bt && u && dz && (dense_form || g_out), "krs_gemm_cross_bwd: null operand");
  KRS_REQUIRE(!dense_form || (!r && !dx0 && !dx0_accumulate && !u_upper && !fold_direct),
              "krs_gemm_cross_bwd: the dense form (x0 = NULL) takes no R, dx0, u_upper or fold_direct");
  if (dense_form && !g_out) { g_out = dz; ldg = ld; }    // (never written by the fused form; the two-call form passes through it)
  if (!r) { ldr = n; beta = 0.0f; }
  KRS_REQUIRE(dx0 || (!r && !dx0_accumulate && !u_upper && !fold_direct),
              "krs_gemm_cross_bwd: dx0 = NULL (the term is left to the next launch's u_upper) is the form without R");
  KRS_REQUIRE(!u_upper || (r && !dx0_accumulate && beta == 1.0f),
              "krs_gemm_cross_bwd: u_upper (dx0 = R * u_upper + ...) needs R with beta = 1 and no dx0 to accumulate into");
  KRS_REQUIRE(dtype == KRS_BF16 || dtype == KRS_F32, "krs_gemm_cross_bwd: bad dtype");
  KRS_REQUIRE(m >= 0 && n >= 0 && k > 0 && ld >= n && ldg >= n && ldr >= n, "krs_gemm_cross_bwd: bad sizes");
  if (dbias) KRS_REQUIRE(workspace && workspace_bytes >= krs_gemm_cross_bwd_workspace_bytes(m, n),
                         "krs_gemm_cross_bwd: workspace too small (krs_gemm_cross_bwd_workspace_bytes)");
  if (m == 0 || n == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool fused = dtype == KRS_BF16 && gemm_pipe() != 0 && m >= 256 && n >= 256 && k >= 256 && k % 64 == 0 &&
                     ceil_div(m, 256) * ceil_div(n, 256) >= 192 && n % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                     ldr % 8 == 0 && ldg % 8 == 0 && ld % 8 == 0 && al16(a) && al16(bt) && (!r || al16(r)) && al16(g_out) &&
                     (!x0 || al16(x0)) && al16(u) && al16(dz) && (!dx0 || al16(dx0)) && (!u_upper || al16(u_upper));
  if (!fused) {
    // any other shape / dtype: the two calls this entry stands for
    krs_gemm_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    ep.r = r; ep.ldr = ldr; ep.beta = beta;
    // (one pass over K: this entry's workspace holds the column sums, not split-K slabs)
    if (int rc = gemm_run(a, lda, 0, bt, ldb, 1, g_out, ldg, m, n, k, dtype, dtype, r ? &ep : nullptr, nullptr, 0, stream,
                          false))
      return rc;
    KRS_REQUIRE(ldg == ld, "krs_gemm_cross_bwd: the two-call form needs one row stride for G, x0, u, dz and dx0");
    if (dense_form)    // G lands in (or is) dz's buffer, the activation derivative is applied in place
      return krs_dense_act_bwd(g_out, ldg, u, ld, dz, ld, dbias, m, n, act, dtype, workspace, workspace_bytes, stream);
    if (u_upper) {   // the upper layer's term first: dx0 = R * u_upper (its own rounding here), then accumulate
      KRS_REQUIRE(ldr == ld, "krs_gemm_cross_bwd: the two-call form needs R on the common row stride");
      if (int rc = krs_cross_epilogue_bwd(r, u_upper, x0, x0, nullptr, dx0, 0, nullptr, nullptr, m, n, ld, 0.0f, KRS_ACT_NONE,
                                          dtype, nullptr, 0, stream)) return rc;
      dx0_accumulate = 1;
    }
    return krs_cross_epilogue_bwd(g_out, u, x0, x0, dz, dx0, dx0_accumulate, fold_direct ? dx0 : nullptr, dbias, m, n,
                                  ld, 0.0f, act, dtype, workspace, workspace_bytes, stream);
  }
  GemmParams p{};
  p.a = reinterpret_cast<const char*>(a); p.lda = lda; p.a_km = 0;
  p.b = reinterpret_cast<const char*>(bt); p.ldb = ldb; p.b_nk = 1;
  p.c = reinterpret_cast<char*>(g_out); p.ldc = ldg; p.m = m; p.n = n; p.k = k; p.out_dtype = KRS_BF16;
  p.has_ep = 1;
  memset(&p.ep, 0, sizeof(p.ep));
  p.ep.r = r; p.ep.ldr = ldr; p.ep.beta = beta;
  p.splits = 1; p.k_per_split = k; p.slabs = nullptr; p.ep_vec = 1;
  p.f_x0 = x0; p.f_u = u; p.f_uup = u_upper; p.f_dz = dz; p.f_dx0 = dx0; p.f_ld = ld; p.f_act = act; p.f_fold = fold_direct != 0;
  p.f_partial = dbias ? reinterpret_cast<float*>(workspace) : nullptr;
  const int nt_ = (int)ceil_div(n, 256);
  const dim3 grid256((unsigned)(ceil_div(ceil_div(m, 256), 8) * 8 * nt_));
#define KRS_CB_LAUNCH(EP)                                                                              \
  {                                                                                                    \
    auto kern = gemm_pp256_kernel<false, 4, EP>;                                                       \
    static bool attr_set = false;                                                                      \
    if (!attr_set) {                                                                                   \
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pp::STAGE));         \
      attr_set = true;                                                                                 \
    }                                                                                                  \
    hipLaunchKernelGGL(kern, grid256, dim3(512), 4 * pp::STAGE, st, p, 0, nt_);                        \
  }
  if (gemm_pipe() == 4 && k % 64 == 0 && k >= 192) {   // the 64-k ring, as krs_gemm's K-contiguous products (level with the 32-k
                                                        // ring on this epilogue-bound form: 640-650 us either way)
#define KRS_CB64_LAUNCH(EP)                                                                            \
  {                                                                                                    \
    auto kern = gemm_pp64_kernel<EP>;                                                                  \
    static bool attr_set = false;                                                                      \
    if (!attr_set) {                                                                                   \
      KRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pp64::NSLOT * pp64::SLOT)); \
      attr_set = true;                                                                                 \
    }                                                                                                  \
    hipLaunchKernelGGL(kern, grid256, dim3(512), pp64::NSLOT * pp64::SLOT, st, p, nt_);                \
  }
    if (dense_form) KRS_CB64_LAUNCH(9)
    else if (u_upper) KRS_CB64_LAUNCH(7)
    else if (r) {
      if (dx0_accumulate) KRS_CB64_LAUNCH(4)
      else KRS_CB64_LAUNCH(3)
    } else {
      if (!dx0) KRS_CB64_LAUNCH(8)
      else if (dx0_accumulate) KRS_CB64_LAUNCH(6)
      else KRS_CB64_LAUNCH(5)
    }
#undef KRS_CB64_LAUNCH
    KRS_CHECK_LAUNCH("gemm_pp64_kernel (fused cross backward)");
    if (dbias) return finish_colsum(p.f_partial, 2 * ceil_div(m, 256), n, dbias, st);
    return KRS_OK;
  }
  if (dense_form) KRS_CB_LAUNCH(9)
  else if (u_upper) KRS_CB_LAUNCH(7)
  else if (r) {
    if (dx0_accumulate) KRS_CB_LAUNCH(4)
    else KRS_CB_LAUNCH(3)
  } else {
    if (!dx0) KRS_CB_LAUNCH(8)
    else if (dx0_accumulate) KRS_CB_LAUNCH(6)
    else KRS_CB_LAUNCH(5)
  }
#undef KRS_CB_LAUNCH
  KRS_CHECK_LAUNCH("gemm_pp256_kernel (fused cross backward)");
  if (dbias) return finish_colsum(p.f_partial, 2 * ceil_div(m, 256), n, dbias, st);
  return KRS_OK;
}
