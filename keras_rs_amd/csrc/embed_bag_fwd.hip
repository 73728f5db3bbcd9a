// K1 -- fused multi-table embedding gather + weighted segment pool (forward).
//
// One launch covers every feature of a DistributedEmbedding; it replaces the
// reference's per-feature chain  ops.take -> multiply -> sum(axis=-2) ->
// divide_no_nan  (keras_rs/src/layers/embedding/embed_reduce.py:178,253,261-274;
// loop at base_distributed_embedding.py:910-928).
//
// CDNA4 mapping (HBM-bound, no data reuse, so no LDS and no MFMA):
//   * a table row is read as 16-byte pieces, one piece per lane: a bf16 D=128
//     row is 256 B = 16 lanes, so a wave64 carries G = 4 independent "groups";
//     every lane owns a fixed column slice of the row and accumulates it in
//     fp32 registers, hence the segment-sum needs no cross-lane traffic at all;
//   * a wave works on one feature (descriptors come in through scalar loads)
//     and each of its groups on `bpg` consecutive bags, walked as ONE flat
//     stream of lookups with bag boundaries handled by a flush test, so the
//     four row loads of an unrolled step are in flight together whatever the
//     bag lengths are, and the ids of the next step are fetched under them;
//   * ids / offsets are read by all lanes of a group from one address
//     (a single broadcast transaction per group), sequentially along the bag.
// Algorithmic bytes per lookup: D*s_t + 4 (+4 with weights); per bag D*s_o + 4.
#include <cstdlib>

#include "krs_common.h"

#ifndef KRS_K1_UNROLL
#define KRS_K1_UNROLL 4
#endif
#ifndef KRS_K1_NT
#define KRS_K1_NT 0  // bit0: nontemporal row loads, bit1: nontemporal output stores
#endif

namespace krs {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kUnroll = KRS_K1_UNROLL;
constexpr int NT = KRS_K1_NT;

struct EmbedFwdParams {
  const krs_table* tables;
  const krs_feature* feats;
  int n_feats;
  const void* ids;
  int id64;
  const void* offsets;  // null = dense mode
  int off64;
  const float* weights;
  int batch;
  int dim;
  void* out;
  int64_t out_ld;
  float* bag_scale;
  int* err_flag;
  int bpg;  // bags per group
};

template <typename T>
struct Vec16;  // 16 bytes of table elements -> fp32 lanes
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
    f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
  }
};
template <>
struct Vec16<uint16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&f)[8]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
};

// store N fp32 values as OT starting at element pointer `dst` (N*sizeof(OT)-byte
// aligned when `aligned`)
template <typename OT, int N>
__device__ __forceinline__ void store_row_piece(OT* dst, const float (&v)[N], bool aligned) {
  if constexpr (sizeof(OT) == 4) {
    if (aligned) {
#pragma unroll
      for (int i = 0; i < N; i += 4)
      {
        f32x4 t4 = {v[i], v[i + 1], v[i + 2], v[i + 3]};
        if constexpr (NT & 2) __builtin_nontemporal_store(t4, reinterpret_cast<f32x4*>(dst + i));
        else *reinterpret_cast<f32x4*>(dst + i) = t4;
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) dst[i] = v[i];
    }
  } else {
    if (aligned) {
      if constexpr (N == 8) {
        u32x4 t4 = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                    pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if constexpr (NT & 2) __builtin_nontemporal_store(t4, reinterpret_cast<u32x4*>(dst));
        else *reinterpret_cast<u32x4*>(dst) = t4;
      } else {
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) dst[i] = f32_to_bf16(v[i]);
    }
  }
}


// TT: table element (float | uint16_t=bf16), OT: output element, LPR: lanes per row.
template <typename TT, typename OT, int LPR, bool HAS_W>
__global__ __launch_bounds__(256) void embed_bag_fwd_vec(const EmbedFwdParams p) {
  constexpr int G = 64 / LPR;
  constexpr int N = Vec16<TT>::N;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR;
  const int sub = lane % LPR;

  const int bags_per_wave = G * p.bpg;
  const int waves_per_feat = (p.batch + bags_per_wave - 1) / bags_per_wave;
  const int64_t wave_unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave_unit >= (int64_t)waves_per_feat * p.n_feats) return;
  const int f = __builtin_amdgcn_readfirstlane((int)(wave_unit / waves_per_feat));
  const int wave_b0 = __builtin_amdgcn_readfirstlane((int)(wave_unit - (int64_t)f * waves_per_feat)) * bags_per_wave;

  // wave-uniform descriptors (scalar loads)
  const krs_feature ft = p.feats[f];
  const krs_table tb = p.tables[ft.table];
  const int vocab = tb.vocab;
  const int comb = ft.combiner;
  const int64_t row_bytes = (int64_t)p.dim * sizeof(TT);
  const int row_pieces = (int)(row_bytes >> 4);
  const char* table = reinterpret_cast<const char*>(tb.weights);
  const bool tab_aligned = (reinterpret_cast<uintptr_t>(table) & 15) == 0;

  const int b_lo = wave_b0 + g * p.bpg;
  const int b_hi = min(b_lo + p.bpg, p.batch);
  if (b_lo >= b_hi || sub >= row_pieces) return;  // no cross-lane ops below: safe to leave

  const bool dense = p.offsets == nullptr;
  const int64_t bag0 = (int64_t)f * p.batch;
  auto bag_end = [&](int j) -> int64_t {
    return dense ? ft.ids_base + (int64_t)(j + 1) * ft.hot : ld_index(p.offsets, p.off64, bag0 + j + 1);
  };
  int64_t q = dense ? ft.ids_base + (int64_t)b_lo * ft.hot : ld_index(p.offsets, p.off64, bag0 + b_lo);
  const int64_t qe = bag_end(b_hi - 1);

  OT* out = reinterpret_cast<OT*>(p.out) + ft.out_col + sub * N;
  constexpr unsigned kStoreAlign = N * sizeof(OT) < 16 ? N * sizeof(OT) : 16;
  const bool out_aligned =
      ((reinterpret_cast<uintptr_t>(out) | (uintptr_t)(p.out_ld * sizeof(OT))) & (kStoreAlign - 1)) == 0;

  int j = b_lo;
  int64_t endj = bag_end(j);
  float acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.0f;
  float sw = 0.0f, sw2 = 0.0f;
  int oob = 0;

  auto flush = [&]() {
    float den = comb == KRS_MEAN ? sw : (comb == KRS_SQRTN ? sqrtf(sw2) : 1.0f);
    float o[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float v = acc[i];
      if (comb != KRS_SUM) v = den == 0.0f ? 0.0f : v / den;
      o[i] = v;
      acc[i] = 0.0f;
    }
    store_row_piece<OT, N>(out + (int64_t)j * p.out_ld, o, out_aligned);
    if (p.bag_scale && sub == 0)
      p.bag_scale[bag0 + j] = comb == KRS_SUM ? 1.0f : (den == 0.0f ? 0.0f : 1.0f / den);
    sw = 0.0f;
    sw2 = 0.0f;
  };

  int idn[kUnroll];
  float wn[kUnroll];
  auto fetch_ids = [&](int64_t q0) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int64_t pos = q0 + k;
      int id = -2;  // -2: past the end of this group's stream
      float w = 1.0f;
      if (pos < qe) {
        const int64_t raw = ld_index(p.ids, p.id64, pos);
        id = (raw >= 0 && raw < vocab) ? (int)raw : -1;  // -1: out of range, flagged, never clamped
        if constexpr (HAS_W) w = p.weights[pos];
      }
      idn[k] = id;
      wn[k] = w;
    }
  };

  fetch_ids(q);
  while (q < qe) {
    int idc[kUnroll];
    float wc[kUnroll];
    uint4 raw[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      idc[k] = idn[k];
      wc[k] = wn[k];
      raw[k] = make_uint4(0, 0, 0, 0);
      if (idc[k] >= 0) {
        const char* src = table + (int64_t)idc[k] * row_bytes + sub * 16;
        if (tab_aligned) {
          u32x4 t4;
          if constexpr (NT & 1) t4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
          else t4 = *reinterpret_cast<const u32x4*>(src);
          raw[k] = make_uint4(t4.x, t4.y, t4.z, t4.w);
        } else {
          const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
          if constexpr (sizeof(TT) == 4) {
            raw[k] = make_uint4(s4[0], s4[1], s4[2], s4[3]);
          } else {
            const uint16_t* s2 = reinterpret_cast<const uint16_t*>(src);
            raw[k] = make_uint4(s2[0] | ((uint32_t)s2[1] << 16), s2[2] | ((uint32_t)s2[3] << 16),
                                s2[4] | ((uint32_t)s2[5] << 16), s2[6] | ((uint32_t)s2[7] << 16));
          }
        }
      }
    }
    fetch_ids(q + kUnroll);  // next step's ids travel under this step's rows
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      if (idc[k] != -2) {
        const int64_t pos = q + k;
        while (pos >= endj && j < b_hi - 1) {  // bounded even for malformed offsets
          flush();
          ++j;
          endj = bag_end(j);
        }
        sw += wc[k];
        sw2 = fmaf(wc[k], wc[k], sw2);
        if (idc[k] == -1) {
          oob = 1;
        } else {
          float fv[N];
          Vec16<TT>::unpack(raw[k], fv);
#pragma unroll
          for (int i = 0; i < N; ++i) acc[i] = fmaf(wc[k], fv[i], acc[i]);
        }
      }
    }
    q += kUnroll;
  }
  while (j < b_hi) {  // the last bag, and any trailing empty ones
    flush();
    ++j;
  }
  if (oob && p.err_flag && sub == 0) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
}

// Any dim / any alignment / any dtype pair: one LPR-lane group per bag, one
// column per lane per pass.  Used when dim*sizeof(T) is not a multiple of 16 B
// (the reference's toy shapes: D = 6, 7, 11, 20).
__global__ __launch_bounds__(256) void embed_bag_fwd_generic(const EmbedFwdParams p, int table_dtype,
                                                             int out_dtype, int lpr) {
  const int groups_per_block = 256 / lpr;
  const int64_t bag = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / lpr;
  const int sub = threadIdx.x % lpr;
  if (bag >= (int64_t)p.n_feats * p.batch) return;
  const int f = (int)(bag / p.batch);
  const int b = (int)(bag - (int64_t)f * p.batch);
  const krs_feature ft = p.feats[f];
  const krs_table tb = p.tables[ft.table];
  int64_t s, e;
  if (p.offsets) {
    s = ld_index(p.offsets, p.off64, bag);
    e = ld_index(p.offsets, p.off64, bag + 1);
  } else {
    s = ft.ids_base + (int64_t)b * ft.hot;
    e = s + ft.hot;
  }
  float sw = 0.0f, sw2 = 0.0f;
  int oob = 0;
  for (int64_t q = s; q < e; ++q) {
    const float w = p.weights ? p.weights[q] : 1.0f;
    sw += w;
    sw2 = fmaf(w, w, sw2);
  }
  const int comb = ft.combiner;
  const float den = comb == KRS_MEAN ? sw : (comb == KRS_SQRTN ? sqrtf(sw2) : 1.0f);
  for (int c = sub; c < p.dim; c += lpr) {
    float acc = 0.0f;
    for (int64_t q = s; q < e; ++q) {
      const int64_t id = ld_index(p.ids, p.id64, q);
      if (id < 0 || id >= tb.vocab) {
        oob = 1;
        continue;
      }
      const float w = p.weights ? p.weights[q] : 1.0f;
      acc = fmaf(w, ld_elem(tb.weights, table_dtype, id * p.dim + c), acc);
    }
    if (comb != KRS_SUM) acc = den == 0.0f ? 0.0f : acc / den;
    st_elem(p.out, out_dtype, (int64_t)b * p.out_ld + ft.out_col + c, acc);
  }
  if (p.bag_scale && sub == 0)
    p.bag_scale[bag] = comb == KRS_SUM ? 1.0f : (den == 0.0f ? 0.0f : 1.0f / den);
  if (oob && p.err_flag && sub == 0) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
}

template <typename TT, typename OT, int LPR>
int launch_vec(const EmbedFwdParams& p, hipStream_t stream) {
  constexpr int G = 64 / LPR;
  const int64_t waves_per_feat = ceil_div(p.batch, (int64_t)G * p.bpg);
  const int64_t blocks = ceil_div(waves_per_feat * p.n_feats, 4);
  if (blocks == 0) return KRS_OK;
  if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_fwd: grid too large");
  if (p.weights)
    hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, true>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, false>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
  KRS_CHECK_LAUNCH("embed_bag_fwd_vec");
  return KRS_OK;
}

template <typename TT, typename OT>
int dispatch_lpr(const EmbedFwdParams& p, int row_pieces, hipStream_t stream) {
  if (row_pieces <= 8) return launch_vec<TT, OT, 8>(p, stream);
  if (row_pieces <= 16) return launch_vec<TT, OT, 16>(p, stream);
  if (row_pieces <= 32) return launch_vec<TT, OT, 32>(p, stream);
  return launch_vec<TT, OT, 64>(p, stream);
}

}  // namespace
}  // namespace krs

extern "C" int krs_embed_bag_fwd(const krs_table* tables, const krs_feature* feats, int n_feats,
                                 const void* ids, int id_type, const void* offsets, int off_type,
                                 const float* weights, int64_t nnz, int batch, int dim,
                                 int table_dtype, void* out, int out_dtype, int64_t out_ld,
                                 float* bag_scale, int* err_flag, void* stream) {
  using namespace krs;
  KRS_REQUIRE(tables && feats && out, "embed_bag_fwd: null tables/feats/out");
  KRS_REQUIRE(ids || nnz == 0, "embed_bag_fwd: null ids");
  KRS_REQUIRE(n_feats >= 0 && batch >= 0 && dim > 0 && nnz >= 0, "embed_bag_fwd: negative size");
  KRS_REQUIRE((table_dtype == KRS_F32 || table_dtype == KRS_BF16) &&
                  (out_dtype == KRS_F32 || out_dtype == KRS_BF16),
              "embed_bag_fwd: bad dtype");
  KRS_REQUIRE((id_type == KRS_I32 || id_type == KRS_I64) && (off_type == KRS_I32 || off_type == KRS_I64),
              "embed_bag_fwd: bad index type");
  if (n_feats == 0 || batch == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  EmbedFwdParams p;
  p.tables = tables; p.feats = feats; p.n_feats = n_feats;
  p.ids = ids; p.id64 = id_type == KRS_I64;
  p.offsets = offsets; p.off64 = off_type == KRS_I64;
  p.weights = weights; p.batch = batch; p.dim = dim;
  p.out = out; p.out_ld = out_ld; p.bag_scale = bag_scale; p.err_flag = err_flag;
  // ~16-32 lookups per group: enough row loads in flight per wave without
  // making one group's stream long (nnz / bags = mean bag length).
  const int64_t n_bags = (int64_t)n_feats * batch;
  const int64_t mean_hot = nnz / n_bags > 0 ? nnz / n_bags : 1;
  int bpg = (int)(16 / mean_hot);
  p.bpg = bpg < 1 ? 1 : (bpg > 16 ? 16 : bpg);
  if (const char* e = getenv("KRS_BPG")) p.bpg = atoi(e);  // development override

  const int64_t row_bytes = (int64_t)dim * (table_dtype == KRS_BF16 ? 2 : 4);
  if (row_bytes % 16 == 0 && row_bytes <= 1024) {
    const int pieces = (int)(row_bytes / 16);
    if (table_dtype == KRS_F32)
      return out_dtype == KRS_F32 ? dispatch_lpr<float, float>(p, pieces, st)
                                  : dispatch_lpr<float, uint16_t>(p, pieces, st);
    return out_dtype == KRS_F32 ? dispatch_lpr<uint16_t, float>(p, pieces, st)
                                : dispatch_lpr<uint16_t, uint16_t>(p, pieces, st);
  }
  int lpr = 1;
  while (lpr < dim && lpr < 64) lpr <<= 1;
  const int64_t blocks = ceil_div(n_bags, 256 / lpr);
  if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_fwd: grid too large");
  hipLaunchKernelGGL(embed_bag_fwd_generic, dim3((unsigned)blocks), dim3(256), 0, st, p, table_dtype,
                     out_dtype, lpr);
  KRS_CHECK_LAUNCH("embed_bag_fwd_generic");
  return KRS_OK;
}
