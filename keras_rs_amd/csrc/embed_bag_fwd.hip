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

namespace krs {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 2;  // bit 1: non-temporal output stores (pooled rows are written once, read by the next kernel)

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
//
// Two phases per window of a group's lookup stream, so that the hot loop has no type / format
// branches and no predicated loads (those made the compiler fence every load with vmcnt(0)):
//   A. the group's lanes load the next WIN ids (+ weights) of the stream coalesced (clamped
//      addresses, no predication), validate them and park them in LDS as int32 row numbers
//      (-1 = out of range); every slot also gets a flag: 0, or bag+1 when it is the LAST lookup
//      of a bag.  Empty bags are written (zeros) here, so the hot loop never sees them.
//   B. the flat loop: id / flag / weight come from LDS, kUnroll row loads are issued
//      unconditionally (row 0 stands in for invalid and padding slots and is masked after the
//      load) and consumed in order; a set flag flushes the finished bag.
// STREAM: bags of ~1 lookup -> no row reuse to protect: 8 row loads in flight, non-temporal.
// HR > 0 (krs_embed_set_option(KRS_EMBED_OPT_HOTROWS, HR); "LDS staging of hot embedding rows"): when the four waves of
// the workgroup work on ONE feature, rows 0 .. HR-1 of its table are copied to LDS first (ids relabelled hot-first put
// the most frequent rows there), and a lookup of such a row is served from LDS.  No branch and no second access form:
// the row address is SELECTED between the LDS copy and the table and read with one generic (flat) 16-byte load, which
// the hardware routes to LDS or to memory by the address -- so the loop keeps its unconditional loads.  What it costs:
// flat loads count on both the memory and the LDS counter, and HR x row bytes of LDS per workgroup lower the occupancy.
template <typename TT, typename OT, int LPR, bool HAS_W, bool STREAM, int HR = 0>
__global__ __launch_bounds__(256) void embed_bag_fwd_vec(const EmbedFwdParams p) {
  constexpr int G = 64 / LPR;
  constexpr int N = Vec16<TT>::N;
  constexpr int kUnroll = STREAM ? 8 : 4;   // (8 for the pooled form too measured slower, round 5: 602-620 -> 632-646 us standalone, 0.72 -> 0.80 ms in-step)
  constexpr int WIN = 8 * LPR;  // ids per window per group (8 coalesced loads per lane)
  constexpr int MAXB = 16;      // bags per group (host keeps bpg <= 16)
  __shared__ int s_ids[4 * G * WIN];
  __shared__ int s_flag[4 * G * WIN];
  __shared__ float s_w[HAS_W ? 4 * G * WIN : 1];
  __shared__ int s_end[4 * G * (MAXB + 1)];
  __shared__ __attribute__((aligned(16))) char s_hot[HR > 0 ? HR * LPR * 16 : 16];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / LPR;
  const int sub = lane % LPR;

  const int bags_per_wave = G * p.bpg;
  const int waves_per_feat = (p.batch + bags_per_wave - 1) / bags_per_wave;
  int hot_rows = 0;   // rows of this workgroup's table that sit in s_hot (workgroup-uniform)
  if constexpr (HR > 0) {
    const int64_t total = (int64_t)waves_per_feat * p.n_feats;
    const int64_t first = (int64_t)blockIdx.x * 4, last = min(first + 3, total - 1);
    if (first < total && first / waves_per_feat == last / waves_per_feat) {
      const krs_feature ft0 = p.feats[first / waves_per_feat];
      const krs_table tb0 = p.tables[ft0.table];
      hot_rows = min(HR, tb0.vocab);
      const int pieces = hot_rows * LPR;      // 16-byte pieces (the host takes this path for row bytes == LPR * 16)
      typedef const __attribute__((address_space(1))) u32x4* gsrc_ptr;
      for (int i = threadIdx.x; i < pieces; i += 256)
        reinterpret_cast<u32x4*>(s_hot)[i] = ((gsrc_ptr)tb0.weights)[i];
    }
    __syncthreads();
  }
  const int64_t wave_unit = (int64_t)blockIdx.x * 4 + wave;
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
  const bool col_live = sub < row_pieces;  // LPR is the next power of two >= row_pieces
  const char* table = reinterpret_cast<const char*>(tb.weights) + (col_live ? sub : 0) * 16;

  const int b_lo = wave_b0 + g * p.bpg;
  const int b_hi = min(b_lo + p.bpg, p.batch);
  if (b_lo >= b_hi) return;  // whole groups only: the lanes of a live group stay together
  const int nb = b_hi - b_lo;

  int* ids_l = s_ids + (wave * G + g) * WIN;
  int* flag_l = s_flag + (wave * G + g) * WIN;
  float* w_l = s_w + (HAS_W ? (wave * G + g) * WIN : 0);
  int* end_l = s_end + (wave * G + g) * (MAXB + 1);  // end_l[b] = start of bag b, end_l[b+1] = its end

  // ---- stream bounds and bag boundaries (relative to the stream start) ----
  const bool dense = p.offsets == nullptr;
  const int64_t bag0 = (int64_t)f * p.batch + b_lo;
  int64_t qs;
  if (dense) {
    qs = ft.ids_base + (int64_t)b_lo * ft.hot;
    for (int b = sub; b <= nb; b += LPR) end_l[b] = b * ft.hot;
  } else {
    qs = ld_index(p.offsets, p.off64, bag0);
    for (int b = sub; b <= nb; b += LPR) end_l[b] = (int)(ld_index(p.offsets, p.off64, bag0 + b) - qs);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int len = end_l[nb];  // lookups in this group's stream

  OT* out = reinterpret_cast<OT*>(p.out) + ft.out_col + sub * N;
  constexpr unsigned kStoreAlign = N * sizeof(OT) < 16 ? N * sizeof(OT) : 16;
  const bool out_aligned =
      ((reinterpret_cast<uintptr_t>(out) | (uintptr_t)(p.out_ld * sizeof(OT))) & (kStoreAlign - 1)) == 0;

  // empty bags: zeros (and scale 1 / 0) straight away -- the stream below never reaches them
  if (!dense || ft.hot == 0) {
    for (int b = 0; b < nb; ++b) {
      if (end_l[b + 1] == end_l[b]) {
        float z[N];
#pragma unroll
        for (int i = 0; i < N; ++i) z[i] = 0.0f;
        if (col_live) store_row_piece<OT, N>(out + (int64_t)(b_lo + b) * p.out_ld, z, out_aligned);
        if (p.bag_scale && sub == 0) p.bag_scale[bag0 + b] = comb == KRS_SUM ? 1.0f : 0.0f;
      }
    }
  }

  float acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.0f;
  float sw = 0.0f, sw2 = 0.0f;
  int oob = 0;

  auto flush = [&](int b) {
    const float den = comb == KRS_MEAN ? sw : (comb == KRS_SQRTN ? sqrtf(sw2) : 1.0f);
    float o[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float v = acc[i];
      if (comb != KRS_SUM) v = den == 0.0f ? 0.0f : v / den;
      o[i] = v;
      acc[i] = 0.0f;
    }
    if (col_live) store_row_piece<OT, N>(out + (int64_t)(b_lo + b) * p.out_ld, o, out_aligned);
    if (p.bag_scale && sub == 0)
      p.bag_scale[bag0 + b] = comb == KRS_SUM ? 1.0f : (den == 0.0f ? 0.0f : 1.0f / den);
    sw = 0.0f;
    sw2 = 0.0f;
  };

  for (int w0 = 0; w0 < len; w0 += WIN) {
    const int wn = min(WIN, len - w0);
    // ---- phase A: stage ids, flags (+ weights) of this window; addresses clamped, no predication ----
    int staged[8];
    if (p.id64) {
      const int64_t* src = reinterpret_cast<const int64_t*>(p.ids) + qs + w0;
      int64_t raw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = src[min(sub + LPR * i, wn - 1)];
#pragma unroll
      for (int i = 0; i < 8; ++i) staged[i] = (raw[i] >= 0 && raw[i] < vocab) ? (int)raw[i] : -1;
    } else {
      const int32_t* src = reinterpret_cast<const int32_t*>(p.ids) + qs + w0;
      int raw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = src[min(sub + LPR * i, wn - 1)];
#pragma unroll
      for (int i = 0; i < 8; ++i) staged[i] = (raw[i] >= 0 && raw[i] < vocab) ? raw[i] : -1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ids_l[sub + LPR * i] = staged[i];
      flag_l[sub + LPR * i] = 0;
      oob |= (staged[i] < 0) & (sub + LPR * i < wn);
    }
    if constexpr (HAS_W) {
      const float* src = p.weights + qs + w0;
      float raw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = src[min(sub + LPR * i, wn - 1)];
#pragma unroll
      for (int i = 0; i < 8; ++i) w_l[sub + LPR * i] = raw[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // last lookup of every non-empty bag that ends inside this window
    for (int b = sub; b < nb; b += LPR) {
      const int e = end_l[b + 1];
      if (e > end_l[b] && e - 1 >= w0 && e - 1 < w0 + wn) flag_l[e - 1 - w0] = b + 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // ---- phase B: the flat loop over the staged window ----
    for (int wq = 0; wq < wn; wq += kUnroll) {
      int idc[kUnroll], flg[kUnroll];
      float wc[kUnroll];
      u32x4 raw[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const int slot = min(wq + k, WIN - 1);
        const bool live = wq + k < wn;
        idc[k] = live ? ids_l[slot] : -1;
        flg[k] = live ? flag_l[slot] : 0;
        if constexpr (HAS_W) wc[k] = live ? w_l[slot] : 0.0f; else wc[k] = live ? 1.0f : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const int row = idc[k] < 0 ? 0 : idc[k];
        // global address space made explicit: a generic pointer would become flat_load (+ lgkmcnt waits)
        typedef const __attribute__((address_space(1))) u32x4* gvec_ptr;
        if constexpr (HR > 0) {
          const char* from_lds = s_hot + row * (LPR * 16) + sub * 16;      // generic pointer into the LDS aperture
          const char* from_mem = table + (int64_t)row * row_bytes;
          raw[k] = *reinterpret_cast<const u32x4*>(row < hot_rows ? from_lds : from_mem);   // one flat load
        } else {
          gvec_ptr src = (gvec_ptr)(table + (int64_t)row * row_bytes);
          if constexpr (STREAM) raw[k] = __builtin_nontemporal_load(src); else raw[k] = *src;
        }
      }
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        sw += wc[k];
        sw2 = fmaf(wc[k], wc[k], sw2);
        float fv[N];
        Vec16<TT>::unpack(make_uint4(raw[k].x, raw[k].y, raw[k].z, raw[k].w), fv);
        const float wk = idc[k] >= 0 ? wc[k] : 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = fmaf(wk, idc[k] >= 0 ? fv[i] : 0.0f, acc[i]);
        if (flg[k]) flush(flg[k] - 1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // window consumed before it is restaged
  }
  if (oob && p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);  // any lane that staged a bad id
}

// Pure row gather: dense bags of exactly one lookup, no weights (the L = 1 headline shape).
// out[b] = table[ids[b]] for every combiner (sum, x/1, x/sqrt(1)), so the rows are moved as raw
// 16-byte pieces when TT == OT.  A wave takes 64 consecutive bags of one feature: one coalesced
// id load per lane, ids broadcast with ds_bpermute (__shfl), 8 non-temporal row loads in flight
// per lane, non-temporal stores.  Features whose `hot` is not 1 take the slow loop at the end.
template <typename TT, typename OT, int LPR, int UREQ>
__global__ __launch_bounds__(256) void embed_gather_hot1(const EmbedFwdParams p) {
  constexpr int G = 64 / LPR;
  constexpr int N = Vec16<TT>::N;
  constexpr int STEPS = 64 / G;  // = LPR
  constexpr int U = UREQ < STEPS ? UREQ : STEPS;
  typedef const __attribute__((address_space(1))) u32x4* gvec_ptr;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR;
  const int sub = lane % LPR;
  const int waves_per_feat = (p.batch + 63) / 64;
  const int64_t wave_unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave_unit >= (int64_t)waves_per_feat * p.n_feats) return;
  const int f = __builtin_amdgcn_readfirstlane((int)(wave_unit / waves_per_feat));
  const int b0 = __builtin_amdgcn_readfirstlane((int)(wave_unit - (int64_t)f * waves_per_feat)) * 64;
  const krs_feature ft = p.feats[f];
  const krs_table tb = p.tables[ft.table];
  const int64_t row_bytes = (int64_t)p.dim * sizeof(TT);
  const int row_pieces = (int)(row_bytes >> 4);
  const bool col_live = sub < row_pieces;
  const char* table = reinterpret_cast<const char*>(tb.weights) + (col_live ? sub : 0) * 16;
  OT* out = reinterpret_cast<OT*>(p.out) + ft.out_col + sub * N;
  constexpr unsigned kStoreAlign = N * sizeof(OT) < 16 ? N * sizeof(OT) : 16;
  const bool out_aligned =
      ((reinterpret_cast<uintptr_t>(out) | (uintptr_t)(p.out_ld * sizeof(OT))) & (kStoreAlign - 1)) == 0;

  if (ft.hot == 1) {
    const int nb = min(64, p.batch - b0);
    const int64_t q = ft.ids_base + b0 + min(lane, nb - 1);
    const int64_t rawid = p.id64 ? reinterpret_cast<const int64_t*>(p.ids)[q] : (int64_t)reinterpret_cast<const int32_t*>(p.ids)[q];
    const int myid = (rawid >= 0 && rawid < tb.vocab) ? (int)rawid : -1;
    if (lane < nb && myid < 0 && p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
    if (p.bag_scale && lane < nb) p.bag_scale[(int64_t)f * p.batch + b0 + lane] = 1.0f;
#pragma unroll 1
    for (int s0 = 0; s0 < STEPS; s0 += U) {
      int id[U];
      u32x4 raw[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        id[k] = __shfl(myid, (s0 + k) * G + g, 64);
        gvec_ptr src = (gvec_ptr)(table + (int64_t)(id[k] < 0 ? 0 : id[k]) * row_bytes);
        raw[k] = __builtin_nontemporal_load(src);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int b = b0 + (s0 + k) * G + g;
        if (b < p.batch && col_live) {
          if (id[k] < 0) raw[k] = u32x4{0, 0, 0, 0};
          OT* dst = out + (int64_t)b * p.out_ld;
          if constexpr (sizeof(TT) == sizeof(OT)) {
            if (out_aligned) {
              __builtin_nontemporal_store(raw[k], reinterpret_cast<u32x4*>(dst));
              continue;
            }
          }
          float fv[N];
          Vec16<TT>::unpack(make_uint4(raw[k].x, raw[k].y, raw[k].z, raw[k].w), fv);
          store_row_piece<OT, N>(dst, fv, out_aligned);
        }
      }
    }
    return;
  }
  // slow loop for the odd feature whose hot != 1 in an otherwise one-hot call
  for (int b = b0 + g; b < min(b0 + 64, p.batch); b += G) {
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.0f;
    int cnt = 0;
    for (int l = 0; l < ft.hot; ++l) {
      const int64_t raw = ld_index(p.ids, p.id64, ft.ids_base + (int64_t)b * ft.hot + l);
      ++cnt;
      if (raw < 0 || raw >= tb.vocab) {
        if (p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
        continue;
      }
      gvec_ptr src = (gvec_ptr)(table + raw * row_bytes);
      const u32x4 r = *src;
      float fv[N];
      Vec16<TT>::unpack(make_uint4(r.x, r.y, r.z, r.w), fv);
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] += fv[i];
    }
    const int comb = ft.combiner;
    const float den = comb == KRS_MEAN ? (float)cnt : (comb == KRS_SQRTN ? sqrtf((float)cnt) : 1.0f);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (comb != KRS_SUM) acc[i] = den == 0.0f ? 0.0f : acc[i] / den;
    if (col_live) store_row_piece<OT, N>(out + (int64_t)b * p.out_ld, acc, out_aligned);
    if (p.bag_scale && sub == 0)
      p.bag_scale[(int64_t)f * p.batch + b] = comb == KRS_SUM ? 1.0f : (den == 0.0f ? 0.0f : 1.0f / den);
  }
}

// The same pure row gather walked SAMPLE-major: unit u = sample * n_feats + feature, a wave takes 64 consecutive
// units, so its stores cover 64 x row_bytes CONTIGUOUS bytes of the output slab whenever the features' column
// slots are adjacent (a sample's 26 outputs are 6.6 KB of one slab row; consecutive samples follow) instead of
// 64 pieces one slab row apart.  Reads are random either way.  Per-feature constants sit in LDS; a lane fetches
// the id of its own unit (the ids of one feature are consecutive in memory, so the 64 strided 4-byte reads hit
// lines that the neighbouring waves use too) and the groups take (id, feature, sample) by ds_bpermute.
constexpr int kRowsMaxFeats = 128;
template <typename TT, int LPR, int UREQ>
__global__ __launch_bounds__(256) void embed_gather_hot1_rows(const EmbedFwdParams p) {
  constexpr int G = 64 / LPR;
  constexpr int N = Vec16<TT>::N;
  constexpr int STEPS = 64 / G;
  constexpr int U = UREQ < STEPS ? UREQ : STEPS;
  typedef const __attribute__((address_space(1))) u32x4* gvec_ptr;
  __shared__ const char* s_table[kRowsMaxFeats];
  __shared__ int64_t s_ids_base[kRowsMaxFeats];
  __shared__ int s_vocab[kRowsMaxFeats], s_out_col[kRowsMaxFeats], s_hot[kRowsMaxFeats], s_comb[kRowsMaxFeats];
  for (int i = threadIdx.x; i < p.n_feats; i += 256) {
    const krs_feature ft = p.feats[i];
    const krs_table tb = p.tables[ft.table];
    s_table[i] = reinterpret_cast<const char*>(tb.weights);
    s_ids_base[i] = ft.ids_base;
    s_vocab[i] = tb.vocab;
    s_out_col[i] = ft.out_col;
    s_hot[i] = ft.hot;
    s_comb[i] = ft.combiner;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, sub = lane % LPR;
  const int64_t units = (int64_t)p.batch * p.n_feats;
  const int64_t u0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (u0 >= units) return;
  const int64_t row_bytes = (int64_t)p.dim * sizeof(TT);
  const bool col_live = sub < (int)(row_bytes >> 4);
  // this lane's own unit: feature, sample, validated id
  const int64_t mu = min(u0 + lane, units - 1);
  const int mb = (int)(mu / p.n_feats), mf = (int)(mu - (int64_t)mb * p.n_feats);
  // (a call is "one lookup per bag on average" when nnz == bags; a feature may still have hot != 1 -- 2 and 0,
  // say: those units take the slow loop below and their id is not read here)
  const bool one = s_hot[mf] == 1;
  const int64_t q = one ? s_ids_base[mf] + mb : 0;
  const int64_t rawid = p.id64 ? reinterpret_cast<const int64_t*>(p.ids)[q] : (int64_t)reinterpret_cast<const int32_t*>(p.ids)[q];
  const int myid = (one && rawid >= 0 && rawid < s_vocab[mf]) ? (int)rawid : -1;
  const bool mine = u0 + lane < units;
  if (mine && one && myid < 0 && p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
  if (p.bag_scale && mine && one) p.bag_scale[(int64_t)mf * p.batch + mb] = 1.0f;
  TT* out = reinterpret_cast<TT*>(p.out) + sub * N;
#pragma unroll 1
  for (int s0 = 0; s0 < STEPS; s0 += U) {
    int id[U], f[U], b[U];
    u32x4 raw[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int src_lane = (s0 + k) * G + g;
      id[k] = __shfl(myid, src_lane, 64);
      f[k] = __shfl(mf, src_lane, 64);
      b[k] = __shfl(mb, src_lane, 64);
      gvec_ptr src = (gvec_ptr)(s_table[f[k]] + (int64_t)(id[k] < 0 ? 0 : id[k]) * row_bytes + (col_live ? sub : 0) * 16);
      raw[k] = __builtin_nontemporal_load(src);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (u0 + (s0 + k) * G + g < units && s_hot[f[k]] != 1) {
        // the odd feature whose bags do not hold exactly one id: plain loop over the bag
        const int hot = s_hot[f[k]], comb = s_comb[f[k]];
        float acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.0f;
        for (int l = 0; l < hot; ++l) {
          const int64_t r = ld_index(p.ids, p.id64, s_ids_base[f[k]] + (int64_t)b[k] * hot + l);
          if (r < 0 || r >= s_vocab[f[k]]) {
            if (p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
            continue;
          }
          const u32x4 rr = *(gvec_ptr)(s_table[f[k]] + r * row_bytes + (col_live ? sub : 0) * 16);
          float fv[N];
          Vec16<TT>::unpack(make_uint4(rr.x, rr.y, rr.z, rr.w), fv);
#pragma unroll
          for (int i = 0; i < N; ++i) acc[i] += fv[i];
        }
        const float den = comb == KRS_MEAN ? (float)hot : (comb == KRS_SQRTN ? sqrtf((float)hot) : 1.0f);
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (comb != KRS_SUM) acc[i] = den == 0.0f ? 0.0f : acc[i] / den;
        if (col_live) store_row_piece<TT, N>(out + (int64_t)b[k] * p.out_ld + s_out_col[f[k]], acc, false);
        if (p.bag_scale && sub == 0)
          p.bag_scale[(int64_t)f[k] * p.batch + b[k]] = comb == KRS_SUM ? 1.0f : (den == 0.0f ? 0.0f : 1.0f / den);
        continue;
      }
      if (u0 + (s0 + k) * G + g < units && col_live) {
        if (id[k] < 0) raw[k] = u32x4{0, 0, 0, 0};
        TT* dst = out + (int64_t)b[k] * p.out_ld + s_out_col[f[k]];
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          __builtin_nontemporal_store(raw[k], reinterpret_cast<u32x4*>(dst));
        } else {   // a column slot that is not 16-byte aligned: element stores
          const TT* e = reinterpret_cast<const TT*>(&raw[k]);
#pragma unroll
          for (int i = 0; i < N; ++i) dst[i] = e[i];
        }
      }
    }
  }
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

// One-hot gather: 16 row loads in flight per lane, sample-major walk (embed_gather_hot1_rows) where the output rows allow it.
// Round 2 kept four variants behind KRS_EMBED_OPT_HOT1 (8 / 16 loads x feature-major / sample-major walk: 172.7 / 170.1 /
// 164.8 / 159.3 us at the C3 L = 1 launch, profiles/archive/r2_k1_hot1_variants_and_lds_hotrows.txt); round 5 keeps the winner only.
int g_hot_rows = 0;   // krs_embed_set_option(KRS_EMBED_OPT_HOTROWS, rows): 0 = no LDS staging (default), 64, 128

template <typename TT, typename OT, int LPR>
int launch_vec(const EmbedFwdParams& p, bool one_hot, bool stream, hipStream_t st) {
  constexpr int G = 64 / LPR;
  if (one_hot) {
    if constexpr (sizeof(TT) == sizeof(OT)) {
      const bool rows_ok = p.n_feats <= kRowsMaxFeats &&
                           ((reinterpret_cast<uintptr_t>(p.out) | (uintptr_t)(p.out_ld * sizeof(OT))) & 15) == 0;
      if (rows_ok) {
        const int64_t blocks = ceil_div(ceil_div((int64_t)p.batch * p.n_feats, 64), 4);
        if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_fwd: grid too large");
        hipLaunchKernelGGL((embed_gather_hot1_rows<TT, LPR, 16>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        KRS_CHECK_LAUNCH("embed_gather_hot1_rows");
        return KRS_OK;
      }
    }
    const int64_t blocks = ceil_div(ceil_div(p.batch, 64) * p.n_feats, 4);
    if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_fwd: grid too large");
    hipLaunchKernelGGL((embed_gather_hot1<TT, OT, LPR, 16>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    KRS_CHECK_LAUNCH("embed_gather_hot1");
    return KRS_OK;
  }
  const int64_t waves_per_feat = ceil_div(p.batch, (int64_t)G * p.bpg);
  const int64_t blocks = ceil_div(waves_per_feat * p.n_feats, 4);
  if (blocks == 0) return KRS_OK;
  if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_fwd: grid too large");
#define KRS_K1_LAUNCH(W, S) \
  hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, W, S>), dim3((unsigned)blocks), dim3(256), 0, st, p)
  if constexpr (sizeof(TT) == 2 && sizeof(OT) == 2 && LPR == 16) {
    // LDS staging of hot rows (opt-in: krs_embed_set_option(KRS_EMBED_OPT_HOTROWS, 64 | 128)): bf16 rows of 256 bytes
    if (g_hot_rows > 0 && !p.weights && (int64_t)p.dim * 2 == LPR * 16) {
      if (g_hot_rows >= 128) {
        if (stream) hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, false, true, 128>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, false, false, 128>), dim3((unsigned)blocks), dim3(256), 0, st, p);
      } else {
        if (stream) hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, false, true, 64>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((embed_bag_fwd_vec<TT, OT, LPR, false, false, 64>), dim3((unsigned)blocks), dim3(256), 0, st, p);
      }
      KRS_CHECK_LAUNCH("embed_bag_fwd_vec (hot rows in LDS)");
      return KRS_OK;
    }
  }
  if (p.weights) { if (stream) KRS_K1_LAUNCH(true, true); else KRS_K1_LAUNCH(true, false); }
  else { if (stream) KRS_K1_LAUNCH(false, true); else KRS_K1_LAUNCH(false, false); }
#undef KRS_K1_LAUNCH
  KRS_CHECK_LAUNCH("embed_bag_fwd_vec");
  return KRS_OK;
}

template <typename TT, typename OT>
int dispatch_lpr(const EmbedFwdParams& p, int row_pieces, bool one_hot, bool stream_mode, hipStream_t stream) {
  if (row_pieces <= 8) return launch_vec<TT, OT, 8>(p, one_hot, stream_mode, stream);
  if (row_pieces <= 16) return launch_vec<TT, OT, 16>(p, one_hot, stream_mode, stream);
  if (row_pieces <= 32) return launch_vec<TT, OT, 32>(p, one_hot, stream_mode, stream);
  return launch_vec<TT, OT, 64>(p, one_hot, stream_mode, stream);
}

}  // namespace
}  // namespace krs

namespace krs { extern int g_plan_variant; }   // embed_bag_bwd.hip

extern "C" int krs_embed_set_option(int key, int value) {
  if (key == KRS_EMBED_OPT_PLAN) {
    KRS_REQUIRE(value == 0 || value == 1, "krs_embed_set_option: plan variant must be 0 or 1");
    krs::g_plan_variant = value;
    return KRS_OK;
  }
  if (key == KRS_EMBED_OPT_HOTROWS) {
    KRS_REQUIRE(value == 0 || value == 64 || value == 128, "krs_embed_set_option: hot rows must be 0, 64 or 128");
    krs::g_hot_rows = value;
    return KRS_OK;
  }
  return krs::fail(KRS_ERR_INVALID, "krs_embed_set_option: unknown key %d", key);
}

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
  int bpg = (int)(32 / mean_hot);
  p.bpg = bpg < 1 ? 1 : (bpg > 16 ? 16 : bpg);
  p.bpg = p.bpg < 1 ? 1 : (p.bpg > 16 ? 16 : p.bpg);          // the kernel stages at most 16 bag ends per group

  const int64_t row_bytes = (int64_t)dim * (table_dtype == KRS_BF16 ? 2 : 4);
  if (row_bytes % 16 == 0 && row_bytes <= 1024) {
    const int pieces = (int)(row_bytes / 16);
    // every bag has exactly one lookup (dense, unweighted): the pure-gather kernel
    const bool one_hot = offsets == nullptr && weights == nullptr && nnz == n_bags;
    const bool stream_mode = nnz < 2 * n_bags;
    if (table_dtype == KRS_F32)
      return out_dtype == KRS_F32 ? dispatch_lpr<float, float>(p, pieces, one_hot, stream_mode, st)
                                  : dispatch_lpr<float, uint16_t>(p, pieces, one_hot, stream_mode, st);
    return out_dtype == KRS_F32 ? dispatch_lpr<uint16_t, float>(p, pieces, one_hot, stream_mode, st)
                                : dispatch_lpr<uint16_t, uint16_t>(p, pieces, one_hot, stream_mode, st);
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
