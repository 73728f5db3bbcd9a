// K6 -- the id side of the row-sharded lookup as ONE call: route, segment, pack (home rank), unpack (owner),
// combine (home rank).
//
// The reference's accelerated path shards rows MOD-N (sharding_strategy="MOD",
// keras_rs/src/layers/embedding/jax/embedding_utils.py:187-197) and hands the per-partition id lists to the
// SparseCore library (preprocess_sparse_dense_matmul_input: `jax/embedding_utils.py:144-217`); the lookup itself
// is `tpu_sparse_dense_matmul` (jax/embedding_lookup.py:134-147).  On MI355X the exchange is an all-to-all of ONE
// PARTIALLY POOLED vector per (bag, owner) pair (keras_rs_amd/sharded.py); everything the home rank has to
// derive from the ids for that -- in round 1 a chain of ~40 ATen kernels (argsort, cumsum x3, scatter, index_add,
// repeat_interleave ...) and three host syncs -- is integer work with data-independent shapes:
//   krs_shard_route    per lookup: feature / bag, range check (ids outside [0, vocab) are dropped and flagged,
//                      never clamped), composite id -> owner (c % N) and stacked local row (c / N); stable
//                      counting sort by owner; SEGMENTS = runs of one bag inside an owner's bucket (the sort is
//                      stable, so bags ascend inside a bucket); per lookup weight = user weight x combiner scale
//                      (mean: 1 / sum w, sqrtn: 1 / sqrt(sum w^2), divide_no_nan: embed_reduce.py:255-274);
//                      output: the PACKED send buffer [per owner: rows | weights | segment lengths], the
//                      (bag, owner) -> segment table the home-side combine reads, the gradient row of every
//                      segment for the backward, and the per-owner counts (lookups, segments, packed words)
//   krs_shard_unpack   owner side: the packed blocks of all sources -> rows, weights, CSR segment offsets
//                      (what krs_embed_bag_fwd / the fused K2 take)
//   krs_shard_combine  home side: out[b, f] = sum over owners of the returned partial of (bag, owner), fp32 in
//                      ascending owner order, one rounding
//   krs_publish_i64    copies a few counters to page-locked host memory and sets a sequence flag behind them
//                      (the host polls the flag: one wait per lookup, no stream synchronisation)
// All integer outputs are bit-exact against oracle/krs_oracle.c (krs_oracle_shard_*).
#include "krs_common.h"
#include "krs_scan.h"

namespace krs {
namespace {

constexpr int kRB = 1024;            // positions per block: 256 threads x 4 rounds (route) / x 4 consecutive (segments)
constexpr int kMaxShards = 16;       // + 1 bucket for the dropped lookups
constexpr int kMaxFeats = 1024;

struct RouteWs {
  int32_t* bag_of_pos;   // [nnz]      CSR mode: bag of every lookup position
  float* scale;          // [n_bags]   combiner scale of every bag (only when some combiner is not "sum")
  uint8_t* dest;         // [nnz]      owner of the lookup (n_shards = dropped)
  int32_t* row;          // [nnz]      stacked local row at the owner
  int32_t* bagp;         // [nnz]      bag (feature * batch + sample)
  int32_t* blk_cnt;      // [n_shards + 1][n_blocks]  histogram, then exclusive offsets (bucket-major = final order)
  int32_t* rows_b;       // [nnz]      bucket order
  int32_t* bag_b;        // [nnz]
  float* w_b;            // [nnz]
  int32_t* blk_heads;    // [n_blocks] segment heads per block, then exclusive offsets
  int32_t* seg_first;    // [nnz + 1]  first bucket-order position of every segment
  int32_t* scan_sums;    // workspace of the histogram scan
  int64_t* meta;         // [128]: start[0..N] (N+1 entries: bucket starts, start[N] = valid lookups), 17: n_seg,
                         //       18..18+N: seg_start[d], 36..36+N: packed base of owner d; static-capacity form:
                         //       64..64+N: lookups kept for owner d, 81..81+N: segments kept for owner d
};
constexpr int kMetaNseg = 17, kMetaSegStart = 18, kMetaPackBase = 36, kMetaKeepCnt = 64, kMetaKeepSeg = 81;
constexpr int kMetaWords = 128;
constexpr int kStaticHeader = 4;     // words in front of every static block: lookups, segments, need_l, need_s

struct RouteParams {
  const krs_shard_feature* feats;
  int n_feats;
  const void* ids;
  int id64;
  const void* offsets;
  int off64;
  const float* weights;
  int64_t nnz;
  int batch;
  int n_shards;
  int64_t n_bags;
  int emit_w;
  int any_scale;
  int n_blocks;
  RouteWs ws;
  int32_t* packed;
  int32_t* seg_bag;
  int32_t* seg_grow;
  int32_t* bag_seg;
  int64_t* counts;
  int32_t* err_flag;
  // static-capacity form (krs_shard_route_static): every owner's block has cap_l lookup and cap_s segment slots
  int64_t cap_l, cap_s;     // 0 = exact form
  int64_t block_words;      // kStaticHeader + cap_l * (1 + emit_w) + cap_s
};

__device__ __forceinline__ int64_t bag_lo(const RouteParams& p, const krs_shard_feature& f, int64_t bag, int b) {
  return p.offsets ? ld_index(p.offsets, p.off64, bag) : f.ids_base + (int64_t)b * f.hot;
}

// CSR mode: bag of every lookup position
__global__ __launch_bounds__(256) void route_expand_kernel(const RouteParams p) {
  const int64_t bag = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (bag >= p.n_bags) return;
  const int64_t lo = ld_index(p.offsets, p.off64, bag), hi = ld_index(p.offsets, p.off64, bag + 1);
  for (int64_t q = lo; q < hi && q < p.nnz; ++q) p.ws.bag_of_pos[q] = (int32_t)bag;
}

// combiner scale per bag: 1 (sum), 1 / sum w (mean), 1 / sqrt(sum w^2) (sqrtn); 0 where the divisor is 0
__global__ __launch_bounds__(256) void route_scale_kernel(const RouteParams p) {
  const int64_t bag = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (bag >= p.n_bags) return;
  const int f = (int)(bag / p.batch), b = (int)(bag % p.batch);
  const krs_shard_feature ft = p.feats[f];
  float s = 1.0f;
  if (ft.combiner != KRS_SUM) {
    const int64_t lo = bag_lo(p, ft, bag, b);
    const int64_t hi = p.offsets ? ld_index(p.offsets, p.off64, bag + 1) : lo + ft.hot;
    float s1 = 0.0f, s2 = 0.0f;
    for (int64_t q = lo; q < hi; ++q) {          // ascending position, as the forward kernel sums them
      const float w = p.weights ? p.weights[q] : 1.0f;
      s1 += w;
      s2 = fmaf(w, w, s2);
    }
    const float d = ft.combiner == KRS_MEAN ? s1 : sqrtf(s2);
    s = d != 0.0f ? 1.0f / d : 0.0f;
  }
  p.ws.scale[bag] = s;
}

// per lookup: bag, range check, owner, local row; per-block histogram of the owners
__global__ __launch_bounds__(256) void route_classify_kernel(const RouteParams p) {
  __shared__ int64_t fbase[kMaxFeats + 1];
  __shared__ int cnt[kMaxShards + 1];
  if (!p.offsets)
    for (int i = threadIdx.x; i <= p.n_feats; i += 256)
      fbase[i] = i < p.n_feats ? p.feats[i].ids_base : p.nnz;
  if (threadIdx.x <= kMaxShards) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRB;
  const int lane = threadIdx.x & 63;
  bool bad = false;
  for (int it = 0; it < 4; ++it) {
    const int64_t q = base + it * 256 + threadIdx.x;
    int d = -1;
    if (q < p.nnz) {
      int f;
      int64_t bag;
      if (p.offsets) {
        bag = p.ws.bag_of_pos[q];
        f = (int)(bag / p.batch);
      } else {
        int lo = 0, hi = p.n_feats;          // last feature whose base is <= q
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (fbase[mid] <= q) lo = mid; else hi = mid;
        }
        f = lo;
        const int hot = p.feats[f].hot;
        bag = (int64_t)f * p.batch + (q - fbase[f]) / hot;
      }
      const krs_shard_feature ft = p.feats[f];
      const int64_t id = ld_index(p.ids, p.id64, q);
      if (id < 0 || id >= ft.vocab) {
        d = p.n_shards;
        bad = true;
        p.ws.row[q] = 0;
      } else {
        const int64_t c = ft.comp_off + id;
        d = (int)(c % p.n_shards);
        p.ws.row[q] = (int32_t)(c / p.n_shards);
      }
      p.ws.dest[q] = (uint8_t)d;
      p.ws.bagp[q] = (int32_t)bag;
    }
    for (int t = 0; t <= p.n_shards; ++t) {
      const unsigned long long m = __ballot(d == t);
      if (lane == 0 && m) atomicAdd(&cnt[t], __popcll(m));
    }
  }
  if (__ballot(bad) && lane == 0 && p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
  __syncthreads();
  if ((int)threadIdx.x <= p.n_shards) p.ws.blk_cnt[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

// bucket starts from the scanned [bucket][block] offsets
__global__ void route_starts_kernel(const RouteParams p) {
  const int d = threadIdx.x;
  if (d > p.n_shards) return;
  // start of bucket d = offset of its first block; start[n_shards] = number of valid lookups
  p.ws.meta[d] = p.ws.blk_cnt[(int64_t)d * p.n_blocks];
  if (d <= p.n_shards) p.ws.meta[kMetaSegStart + d] = -1;
  if (d == 0) p.ws.meta[kMetaNseg] = 0;
}

// stable scatter into bucket order
__global__ __launch_bounds__(256) void route_scatter_kernel(const RouteParams p) {
  __shared__ int run[kMaxShards + 1];
  __shared__ int wcnt[4][kMaxShards + 1];
  if ((int)threadIdx.x <= p.n_shards) run[threadIdx.x] = p.ws.blk_cnt[(int64_t)threadIdx.x * gridDim.x + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < 4; ++it) {
    const int64_t q = base + it * 256 + threadIdx.x;
    const int d = q < p.nnz ? (int)p.ws.dest[q] : -1;
    int rank = 0;
    for (int t = 0; t <= p.n_shards; ++t) {
      const unsigned long long m = __ballot(d == t);
      if (d == t) rank = __popcll(m & ((1ULL << lane) - 1ULL));
      if (lane == 0) wcnt[wave][t] = __popcll(m);
    }
    __syncthreads();
    if (d >= 0 && d < p.n_shards) {
      int pos = run[d] + rank;
      for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
      const int bag = p.ws.bagp[q];
      p.ws.rows_b[pos] = p.ws.row[q];
      p.ws.bag_b[pos] = bag;
      if (p.emit_w) p.ws.w_b[pos] = (p.weights ? p.weights[q] : 1.0f) * (p.any_scale ? p.ws.scale[bag] : 1.0f);
    }
    __syncthreads();
    if ((int)threadIdx.x <= p.n_shards)
      run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    __syncthreads();
  }
}

__device__ __forceinline__ bool is_head(const RouteParams& p, const int64_t* start, int64_t q) {
  if (q == 0) return true;
  for (int d = 1; d < p.n_shards; ++d)
    if (q == start[d]) return true;
  return p.ws.bag_b[q] != p.ws.bag_b[q - 1];
}

// segment heads per block of kRB bucket-order positions
__global__ __launch_bounds__(256) void route_heads_kernel(const RouteParams p) {
  __shared__ int64_t start[kMaxShards + 1];
  __shared__ int total;
  if ((int)threadIdx.x <= p.n_shards) start[threadIdx.x] = p.ws.meta[threadIdx.x];
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const int64_t n_valid = start[p.n_shards];
  const int64_t q0 = (int64_t)blockIdx.x * kRB + threadIdx.x * 4;
  int c = 0;
  for (int k = 0; k < 4; ++k)
    if (q0 + k < n_valid && is_head(p, start, q0 + k)) ++c;
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) p.ws.blk_heads[blockIdx.x] = total;
}

// segment ids: first position, bag, gradient row, (bag, owner) -> segment
__global__ __launch_bounds__(256) void route_segments_kernel(const RouteParams p) {
  __shared__ int64_t start[kMaxShards + 1];
  __shared__ int wsum[4];
  if ((int)threadIdx.x <= p.n_shards) start[threadIdx.x] = p.ws.meta[threadIdx.x];
  __syncthreads();
  const int64_t n_valid = start[p.n_shards];
  const int64_t q0 = (int64_t)blockIdx.x * kRB + threadIdx.x * 4;
  bool h[4];
  int c = 0;
  for (int k = 0; k < 4; ++k) {
    h[k] = q0 + k < n_valid && is_head(p, start, q0 + k);
    c += h[k];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int s = p.ws.blk_heads[blockIdx.x] + x - c;
  for (int w = 0; w < wave; ++w) s += wsum[w];
  const int n_feats = p.n_feats;
  for (int k = 0; k < 4; ++k) {
    if (!h[k]) continue;
    const int64_t q = q0 + k;
    const int bag = p.ws.bag_b[q];
    int d = 0;
    while (d + 1 < p.n_shards && q >= start[d + 1]) ++d;
    p.ws.seg_first[s] = (int32_t)q;
    p.seg_bag[s] = bag;
    if (p.cap_l == 0) {     // (the static form numbers the segments by their slot: route_pack_static_kernel)
      p.seg_grow[s] = (bag % p.batch) * n_feats + bag / p.batch;
      p.bag_seg[(int64_t)bag * p.n_shards + d] = s;
    }
    if (q == start[d]) p.ws.meta[kMetaSegStart + d] = s;
    ++s;
  }
}

// per-owner counts, packed layout; counts = [lookups[N] | segments[N] | packed words[N]]
__global__ void route_finalize_kernel(const RouteParams p) {
  if (threadIdx.x != 0) return;
  const int n = p.n_shards;
  const int64_t n_seg = p.ws.meta[kMetaNseg];
  const int64_t n_valid = p.ws.meta[n];
  int64_t next = n_seg;
  for (int d = n - 1; d >= 0; --d) {      // an empty bucket starts where the next one does
    const int64_t cnt = p.ws.meta[d + 1] - p.ws.meta[d];
    if (cnt == 0) p.ws.meta[kMetaSegStart + d] = next;
    else next = p.ws.meta[kMetaSegStart + d];
  }
  p.ws.meta[kMetaSegStart + n] = n_seg;
  int64_t base = 0;
  for (int d = 0; d < n; ++d) {
    const int64_t cnt = p.ws.meta[d + 1] - p.ws.meta[d];
    const int64_t segs = p.ws.meta[kMetaSegStart + d + 1] - p.ws.meta[kMetaSegStart + d];
    const int64_t words = cnt * (1 + p.emit_w) + segs;
    p.counts[d] = cnt;
    p.counts[n + d] = segs;
    p.counts[2 * n + d] = words;
    p.ws.meta[kMetaPackBase + d] = base;
    base += words;
  }
  p.ws.seg_first[n_seg] = (int32_t)n_valid;
  if (p.cap_l > 0) {
    // Static-capacity form: owner d's block sits at d * block_words and holds the FIRST cap_s segments of its
    // bucket and of those the first cap_l lookups; what does not fit is dropped and flagged (the reference's
    // allow_id_dropping=True, jax/embedding_utils.py:187-197).  need_l / need_s = the capacities this call would
    // have needed; they travel in every block header so that all ranks see the same maxima (update_stats).
    int64_t need_l = 0, need_s = 0;
    bool over = false;
    for (int d = 0; d < n; ++d) {
      const int64_t cnt = p.counts[d], segs = p.counts[n + d];
      need_l = cnt > need_l ? cnt : need_l;
      need_s = segs > need_s ? segs : need_s;
      int64_t keep_s = segs < p.cap_s ? segs : p.cap_s;
      int64_t keep_l = cnt;
      if (keep_s < segs) keep_l = (int64_t)p.ws.seg_first[p.ws.meta[kMetaSegStart + d] + keep_s] - p.ws.meta[d];
      if (keep_l > p.cap_l) keep_l = p.cap_l;
      over = over || keep_l < cnt || keep_s < segs;
      p.ws.meta[kMetaKeepCnt + d] = keep_l;
      p.ws.meta[kMetaKeepSeg + d] = keep_s;
    }
    for (int d = 0; d < n; ++d) {
      int32_t* h = p.packed + d * p.block_words;
      h[0] = (int32_t)p.ws.meta[kMetaKeepCnt + d];
      h[1] = (int32_t)p.ws.meta[kMetaKeepSeg + d];
      h[2] = (int32_t)need_l;
      h[3] = (int32_t)need_s;
    }
    if (over && p.err_flag) atomicOr(p.err_flag, KRS_FLAG_CAPACITY_OVERFLOW);
  }
}

// static-capacity form of the send buffer: per owner a block of block_words words
//   [lookups kept, segments kept, need_l, need_s | rows[cap_l] | weights[cap_l] (if emitted) | segment lengths[cap_s]]
// (the buffer was zeroed), the gradient row of every segment SLOT (0 for empty slots) and (bag, owner) -> slot.
__global__ __launch_bounds__(256) void route_pack_static_kernel(const RouteParams p) {
  __shared__ int64_t start[kMaxShards + 1], sstart[kMaxShards + 1], keep_l[kMaxShards], keep_s[kMaxShards];
  if ((int)threadIdx.x <= p.n_shards) {
    start[threadIdx.x] = p.ws.meta[threadIdx.x];
    sstart[threadIdx.x] = p.ws.meta[kMetaSegStart + threadIdx.x];
  }
  if ((int)threadIdx.x < p.n_shards) {
    keep_l[threadIdx.x] = p.ws.meta[kMetaKeepCnt + threadIdx.x];
    keep_s[threadIdx.x] = p.ws.meta[kMetaKeepSeg + threadIdx.x];
  }
  __syncthreads();
  const int64_t n_valid = start[p.n_shards], n_seg = sstart[p.n_shards];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_valid) {
    int d = 0;
    while (d + 1 < p.n_shards && i >= start[d + 1]) ++d;
    const int64_t k = i - start[d];
    if (k < keep_l[d]) {
      int32_t* blk = p.packed + d * p.block_words + kStaticHeader;
      blk[k] = p.ws.rows_b[i];
      if (p.emit_w) blk[p.cap_l + k] = __float_as_int(p.ws.w_b[i]);
    }
  }
  if (i < n_seg) {
    int d = 0;
    while (d + 1 < p.n_shards && i >= sstart[d + 1]) ++d;
    const int64_t j = i - sstart[d];
    const int bag = p.seg_bag[i];
    if (j < keep_s[d]) {
      const int64_t end_kept = start[d] + keep_l[d];
      int64_t hi = p.ws.seg_first[i + 1], lo = p.ws.seg_first[i];
      hi = hi < end_kept ? hi : end_kept;
      p.packed[d * p.block_words + kStaticHeader + p.cap_l * (1 + p.emit_w) + j] = (int32_t)(hi > lo ? hi - lo : 0);
      p.seg_grow[d * p.cap_s + j] = (bag % p.batch) * p.n_feats + bag / p.batch;
      p.bag_seg[(int64_t)bag * p.n_shards + d] = (int32_t)(d * p.cap_s + j);
    }
  }
}

// the send buffer: per owner [rows | weights (bit patterns) | segment lengths]
__global__ __launch_bounds__(256) void route_pack_kernel(const RouteParams p) {
  __shared__ int64_t start[kMaxShards + 1], sstart[kMaxShards + 1], pbase[kMaxShards + 1];
  if ((int)threadIdx.x <= p.n_shards) {
    start[threadIdx.x] = p.ws.meta[threadIdx.x];
    sstart[threadIdx.x] = p.ws.meta[kMetaSegStart + threadIdx.x];
    pbase[threadIdx.x] = p.ws.meta[kMetaPackBase + threadIdx.x];
  }
  __syncthreads();
  const int64_t n_valid = start[p.n_shards], n_seg = sstart[p.n_shards];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_valid) {
    int d = 0;
    while (d + 1 < p.n_shards && i >= start[d + 1]) ++d;
    const int64_t cnt = start[d + 1] - start[d];
    const int64_t o = pbase[d] + (i - start[d]);
    p.packed[o] = p.ws.rows_b[i];
    if (p.emit_w) p.packed[o + cnt] = __float_as_int(p.ws.w_b[i]);
  }
  if (i < n_seg) {
    int d = 0;
    while (d + 1 < p.n_shards && i >= sstart[d + 1]) ++d;
    const int64_t cnt = start[d + 1] - start[d];
    p.packed[pbase[d] + cnt * (1 + p.emit_w) + (i - sstart[d])] = p.ws.seg_first[i + 1] - p.ws.seg_first[i];
  }
}

// ---- owner side ------------------------------------------------------------------------------------
struct UnpackParams {
  const int32_t* packed;
  int n_src;
  int weighted;
  int64_t cnt_start[kMaxShards + 1];    // rows of source s go to rows[cnt_start[s] ..)
  int64_t seg_start[kMaxShards + 1];
  int64_t pack_base[kMaxShards + 1];
  int32_t* rows;
  float* w;
  int32_t* offsets;                     // [total segments + 1]
};

__global__ __launch_bounds__(256) void unpack_kernel(const UnpackParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_cnt = p.cnt_start[p.n_src], n_seg = p.seg_start[p.n_src];
  if (i < n_cnt) {
    int s = 0;
    while (s + 1 < p.n_src && i >= p.cnt_start[s + 1]) ++s;
    const int64_t cnt = p.cnt_start[s + 1] - p.cnt_start[s];
    const int64_t o = p.pack_base[s] + (i - p.cnt_start[s]);
    p.rows[i] = p.packed[o];
    if (p.weighted) p.w[i] = __int_as_float(p.packed[o + cnt]);
  }
  if (i < n_seg) {
    int s = 0;
    while (s + 1 < p.n_src && i >= p.seg_start[s + 1]) ++s;
    const int64_t cnt = p.cnt_start[s + 1] - p.cnt_start[s];
    p.offsets[i] = p.packed[p.pack_base[s] + cnt * (1 + p.weighted) + (i - p.seg_start[s])];   // lengths, scanned next
  }
  if (i == n_seg) p.offsets[n_seg] = 0;
}

// static-capacity form: the received blocks (one per source, block_words apart, header in front) -> compact rows /
// weights (source order; the tail up to n_src * cap_l is padded with row -1, weight 0: no segment refers to it and K2's
// plan sorts it behind every valid row), segment LENGTHS at the segment's slot s * cap_s + j (0 for empty slots;
// scanned to CSR offsets next), stats = [max need_l, max need_s, lookups, segments] over the sources.
struct UnpackStaticParams {
  const int32_t* packed;
  int n_src;
  int weighted;
  int64_t cap_l, cap_s, block_words;
  int32_t* rows;
  float* w;
  int32_t* offsets;      // [n_src * cap_s + 1]
  int64_t* stats;        // [4] or null
};

__global__ __launch_bounds__(256) void unpack_static_kernel(const UnpackStaticParams p) {
  __shared__ int64_t cnt[kMaxShards], segs[kMaxShards], pre[kMaxShards + 1];
  if (threadIdx.x == 0) {
    int64_t run = 0, nl = 0, ns = 0, tot_s = 0;
    for (int s = 0; s < p.n_src; ++s) {
      const int32_t* h = p.packed + s * p.block_words;
      int64_t c = h[0], g = h[1];
      c = c < 0 ? 0 : (c > p.cap_l ? p.cap_l : c);
      g = g < 0 ? 0 : (g > p.cap_s ? p.cap_s : g);
      cnt[s] = c; segs[s] = g; pre[s] = run;
      run += c; tot_s += g;
      nl = h[2] > nl ? h[2] : nl;
      ns = h[3] > ns ? h[3] : ns;
    }
    pre[p.n_src] = run;
    if (blockIdx.x == 0 && p.stats) { p.stats[0] = nl; p.stats[1] = ns; p.stats[2] = run; p.stats[3] = tot_s; }
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t slots_l = (int64_t)p.n_src * p.cap_l, slots_s = (int64_t)p.n_src * p.cap_s;
  if (i < slots_l) {
    const int s = (int)(i / p.cap_l);
    const int64_t k = i - (int64_t)s * p.cap_l;
    if (k < cnt[s]) {
      const int32_t* blk = p.packed + s * p.block_words + kStaticHeader;
      p.rows[pre[s] + k] = blk[k];
      if (p.weighted) p.w[pre[s] + k] = __int_as_float(blk[p.cap_l + k]);
    }
    if (i >= pre[p.n_src]) {
      p.rows[i] = -1;
      if (p.weighted) p.w[i] = 0.0f;
    }
  }
  if (i < slots_s) {
    const int s = (int)(i / p.cap_s);
    const int64_t j = i - (int64_t)s * p.cap_s;
    p.offsets[i] = j < segs[s] ? p.packed[s * p.block_words + kStaticHeader + p.cap_l * (1 + p.weighted) + j] : 0;
  }
  if (i == slots_s) p.offsets[slots_s] = 0;
}

// ---- home side -------------------------------------------------------------------------------------
// out[b, f*dim ..] = sum_d partials[bag_seg[(f*batch + b)*N + d]]; one thread per VEC columns of one bag
template <typename T, int VEC>
__global__ __launch_bounds__(256) void combine_kernel(const T* partials, const int32_t* bag_seg, int batch, int n_feats,
                                                      int n_shards, int dim, T* out, int64_t out_ld) {
  const int vpr = dim / VEC;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)batch * n_feats * vpr;
  if (i >= total) return;
  const int64_t bag = i / vpr;                 // feature-major bag index
  const int v = (int)(i - bag * vpr);
  const int f = (int)(bag / batch), b = (int)(bag - (int64_t)f * batch);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
  for (int d = 0; d < n_shards; ++d) {
    const int s = bag_seg[bag * n_shards + d];
    if (s < 0) continue;
    const T* src = partials + (int64_t)s * dim + v * VEC;
    if constexpr (sizeof(T) == 2) {
      if constexpr (VEC == 8) {
        const uint4 r = *reinterpret_cast<const uint4*>(src);
        const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[2 * k] += __uint_as_float(u[k] << 16);
          acc[2 * k + 1] += __uint_as_float(u[k] & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += bf16_to_f32(reinterpret_cast<const uint16_t*>(src)[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += reinterpret_cast<const float*>(src)[k];
    }
  }
  T* dst = out + (int64_t)b * out_ld + (int64_t)f * dim + v * VEC;
  if constexpr (sizeof(T) == 2) {
    if constexpr (VEC == 8) {
      *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                                                  pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) reinterpret_cast<uint16_t*>(dst)[k] = f32_to_bf16(acc[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) reinterpret_cast<float*>(dst)[k] = acc[k];
  }
}

__global__ void publish_kernel(const int64_t* src, int n, volatile int64_t* host_dst, int64_t seq) {
  if (threadIdx.x < (unsigned)n) host_dst[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    host_dst[n] = seq;
  }
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

RouteWs carve(void* workspace, int64_t nnz, int64_t n_bags, int n_shards, int n_blocks, size_t* bytes) {
  char* base = reinterpret_cast<char*>(workspace);
  size_t off = 0;
  auto take = [&](size_t n) {
    char* q = base ? base + off : nullptr;
    off += align_up(n);
    return q;
  };
  RouteWs w;
  w.meta = reinterpret_cast<int64_t*>(take(kMetaWords * sizeof(int64_t)));
  w.bag_of_pos = reinterpret_cast<int32_t*>(take((size_t)nnz * 4));
  w.scale = reinterpret_cast<float*>(take((size_t)n_bags * 4));
  w.dest = reinterpret_cast<uint8_t*>(take((size_t)nnz));
  w.row = reinterpret_cast<int32_t*>(take((size_t)nnz * 4));
  w.bagp = reinterpret_cast<int32_t*>(take((size_t)nnz * 4));
  w.blk_cnt = reinterpret_cast<int32_t*>(take((size_t)(n_shards + 1) * n_blocks * 4));
  w.rows_b = reinterpret_cast<int32_t*>(take((size_t)nnz * 4));
  w.bag_b = reinterpret_cast<int32_t*>(take((size_t)nnz * 4));
  w.w_b = reinterpret_cast<float*>(take((size_t)nnz * 4));
  w.blk_heads = reinterpret_cast<int32_t*>(take((size_t)n_blocks * 4));
  w.seg_first = reinterpret_cast<int32_t*>(take((size_t)(nnz + 1) * 4));
  w.scan_sums = reinterpret_cast<int32_t*>(take(scan::workspace_bytes((int64_t)(n_shards + 1) * n_blocks)));
  *bytes = off;
  return w;
}

}  // namespace
}  // namespace krs

using namespace krs;

extern "C" size_t krs_shard_route_workspace_bytes(int64_t nnz, int64_t n_bags, int n_shards) {
  if (nnz < 0 || n_bags < 0 || n_shards <= 0 || n_shards > kMaxShards) return 0;
  size_t bytes = 0;
  carve(nullptr, nnz > 0 ? nnz : 1, n_bags > 0 ? n_bags : 1, n_shards, (int)ceil_div(nnz > 0 ? nnz : 1, kRB), &bytes);
  return bytes;
}

static int shard_route_impl(const krs_shard_feature* feats, const krs_shard_feature* feats_host, int n_feats,
                            const void* ids, int id_type, const void* offsets, int offset_type,
                            const float* weights, int64_t nnz, int batch, int n_shards, int emit_weights,
                            int64_t cap_l, int64_t cap_s,
                            int32_t* packed, int32_t* seg_bag, int32_t* seg_grow, int32_t* bag_seg,
                            int64_t* counts, int32_t* err_flag, void* workspace, size_t workspace_bytes,
                            void* stream) {
  KRS_REQUIRE(n_shards >= 1 && n_shards <= kMaxShards, "shard_route: n_shards must be in [1, %d]", kMaxShards);
  KRS_REQUIRE(n_feats >= 1 && n_feats <= kMaxFeats, "shard_route: n_feats must be in [1, %d]", kMaxFeats);
  KRS_REQUIRE(nnz >= 0 && nnz < 0x7fffffffLL && batch >= 0, "shard_route: nnz must fit int32");
  KRS_REQUIRE(feats && feats_host && counts && bag_seg, "shard_route: null argument");
  KRS_REQUIRE(id_type == KRS_I32 || id_type == KRS_I64, "shard_route: bad id type");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n_bags = (int64_t)batch * n_feats;
  KRS_REQUIRE(n_bags * n_shards < 0x7fffffffLL, "shard_route: batch * n_feats * n_shards must fit int32");
  KRS_HIP(hipMemsetAsync(bag_seg, 0xff, (size_t)(n_bags > 0 ? n_bags : 1) * n_shards * sizeof(int32_t), st));
  const int64_t block_words = kStaticHeader + cap_l * (1 + (emit_weights != 0)) + cap_s;
  if (cap_l > 0) {
    KRS_REQUIRE(cap_l % 4 == 0 && cap_s % 4 == 0 && cap_s > 0, "shard_route_static: capacities must be multiples of 4");
    KRS_REQUIRE(block_words * n_shards < 0x7fffffffLL && cap_s * n_shards < 0x7fffffffLL,
                "shard_route_static: the packed buffer must fit int32 indexing");
    KRS_REQUIRE(packed && seg_grow, "shard_route_static: null argument");
    KRS_HIP(hipMemsetAsync(packed, 0, (size_t)block_words * n_shards * sizeof(int32_t), st));
    KRS_HIP(hipMemsetAsync(seg_grow, 0, (size_t)cap_s * n_shards * sizeof(int32_t), st));
  }
  if (nnz == 0) {
    KRS_HIP(hipMemsetAsync(counts, 0, (size_t)3 * n_shards * sizeof(int64_t), st));
    return KRS_OK;
  }
  KRS_REQUIRE(ids && packed && seg_bag && seg_grow && workspace, "shard_route: null argument");
  RouteParams p;
  p.feats = feats; p.n_feats = n_feats; p.ids = ids; p.id64 = id_type == KRS_I64;
  p.offsets = offsets; p.off64 = offset_type == KRS_I64; p.weights = weights; p.nnz = nnz; p.batch = batch;
  p.n_shards = n_shards; p.n_bags = n_bags;
  p.any_scale = 0;
  for (int f = 0; f < n_feats; ++f) {
    KRS_REQUIRE(offsets || feats_host[f].hot >= 1, "shard_route: dense bags need hot >= 1");
    if (feats_host[f].combiner != KRS_SUM) p.any_scale = 1;
  }
  p.emit_w = emit_weights != 0;
  KRS_REQUIRE(p.emit_w || (!weights && !p.any_scale), "shard_route: weights / mean / sqrtn bags need emit_weights");
  p.n_blocks = (int)ceil_div(nnz, kRB);
  size_t need = 0;
  p.ws = carve(workspace, nnz, n_bags, n_shards, p.n_blocks, &need);
  if (workspace_bytes < need) return fail(KRS_ERR_WORKSPACE, "shard_route: workspace too small (%zu < %zu)", workspace_bytes, need);
  p.packed = packed; p.seg_bag = seg_bag; p.seg_grow = seg_grow; p.bag_seg = bag_seg; p.counts = counts;
  p.err_flag = err_flag;
  p.cap_l = cap_l; p.cap_s = cap_s; p.block_words = block_words;
  const unsigned bag_blocks = (unsigned)ceil_div(n_bags, 256);
  if (offsets) hipLaunchKernelGGL(route_expand_kernel, dim3(bag_blocks), dim3(256), 0, st, p);
  if (p.any_scale) hipLaunchKernelGGL(route_scale_kernel, dim3(bag_blocks), dim3(256), 0, st, p);
  hipLaunchKernelGGL(route_classify_kernel, dim3(p.n_blocks), dim3(256), 0, st, p);
  scan::exclusive(p.ws.blk_cnt, p.ws.blk_cnt, (int64_t)(n_shards + 1) * p.n_blocks, p.ws.scan_sums, nullptr, st);
  hipLaunchKernelGGL(route_starts_kernel, dim3(1), dim3(64), 0, st, p);
  hipLaunchKernelGGL(route_scatter_kernel, dim3(p.n_blocks), dim3(256), 0, st, p);
  hipLaunchKernelGGL(route_heads_kernel, dim3(p.n_blocks), dim3(256), 0, st, p);
  hipLaunchKernelGGL(scan::block_kernel, dim3(1), dim3(1024), 0, st, p.ws.blk_heads, (int64_t)p.n_blocks,
                     p.ws.meta + kMetaNseg);
  hipLaunchKernelGGL(route_segments_kernel, dim3(p.n_blocks), dim3(256), 0, st, p);
  hipLaunchKernelGGL(route_finalize_kernel, dim3(1), dim3(64), 0, st, p);
  if (cap_l > 0) hipLaunchKernelGGL(route_pack_static_kernel, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(route_pack_kernel, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, p);
  KRS_CHECK_LAUNCH("krs_shard_route");
  return KRS_OK;
}

extern "C" int krs_shard_route(const krs_shard_feature* feats, const krs_shard_feature* feats_host, int n_feats,
                               const void* ids, int id_type, const void* offsets, int offset_type,
                               const float* weights, int64_t nnz, int batch, int n_shards, int emit_weights,
                               int32_t* packed, int32_t* seg_bag, int32_t* seg_grow, int32_t* bag_seg,
                               int64_t* counts, int32_t* err_flag, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return shard_route_impl(feats, feats_host, n_feats, ids, id_type, offsets, offset_type, weights, nnz, batch, n_shards,
                          emit_weights, 0, 0, packed, seg_bag, seg_grow, bag_seg, counts, err_flag, workspace,
                          workspace_bytes, stream);
}

extern "C" int64_t krs_shard_static_block_words(int64_t cap_lookups, int64_t cap_segments, int emit_weights) {
  return kStaticHeader + cap_lookups * (1 + (emit_weights != 0)) + cap_segments;
}

extern "C" int krs_shard_route_static(const krs_shard_feature* feats, const krs_shard_feature* feats_host, int n_feats,
                                      const void* ids, int id_type, const void* offsets, int offset_type,
                                      const float* weights, int64_t nnz, int batch, int n_shards, int emit_weights,
                                      int64_t cap_lookups, int64_t cap_segments,
                                      int32_t* packed, int32_t* seg_bag, int32_t* seg_grow, int32_t* bag_seg,
                                      int64_t* counts, int32_t* err_flag, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  KRS_REQUIRE(cap_lookups > 0 && cap_segments > 0, "shard_route_static: capacities must be positive");
  return shard_route_impl(feats, feats_host, n_feats, ids, id_type, offsets, offset_type, weights, nnz, batch, n_shards,
                          emit_weights, cap_lookups, cap_segments, packed, seg_bag, seg_grow, bag_seg, counts, err_flag,
                          workspace, workspace_bytes, stream);
}

extern "C" size_t krs_shard_unpack_workspace_bytes(int64_t total_segments) {
  return scan::workspace_bytes(total_segments + 1);
}

extern "C" int krs_shard_unpack(const int32_t* packed, int n_sources, const int64_t* lookups, const int64_t* segments,
                                int weighted, int32_t* rows, float* w, int32_t* offsets, void* workspace,
                                size_t workspace_bytes, void* stream) {
  KRS_REQUIRE(n_sources >= 1 && n_sources <= kMaxShards, "shard_unpack: n_sources must be in [1, %d]", kMaxShards);
  KRS_REQUIRE(lookups && segments && offsets, "shard_unpack: null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  UnpackParams p;
  p.packed = packed; p.n_src = n_sources; p.weighted = weighted != 0; p.rows = rows; p.w = w; p.offsets = offsets;
  p.cnt_start[0] = p.seg_start[0] = p.pack_base[0] = 0;
  for (int s = 0; s < n_sources; ++s) {
    KRS_REQUIRE(lookups[s] >= 0 && segments[s] >= 0, "shard_unpack: negative count");
    p.cnt_start[s + 1] = p.cnt_start[s] + lookups[s];
    p.seg_start[s + 1] = p.seg_start[s] + segments[s];
    p.pack_base[s + 1] = p.pack_base[s] + lookups[s] * (1 + p.weighted) + segments[s];
  }
  const int64_t n_cnt = p.cnt_start[n_sources], n_seg = p.seg_start[n_sources];
  KRS_REQUIRE(n_cnt < 0x7fffffffLL, "shard_unpack: lookups must fit int32");
  if (n_cnt > 0) KRS_REQUIRE(packed && rows && (!p.weighted || w), "shard_unpack: null buffer");
  if (workspace_bytes < krs_shard_unpack_workspace_bytes(n_seg) || !workspace)
    return fail(KRS_ERR_WORKSPACE, "shard_unpack: workspace too small");
  const int64_t span = std::max<int64_t>(std::max(n_cnt, n_seg + 1), 1);
  hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)ceil_div(span, 256)), dim3(256), 0, st, p);
  // lengths -> exclusive offsets (offsets[n_seg] = total)
  scan::exclusive(offsets, offsets, n_seg + 1, reinterpret_cast<int32_t*>(workspace), nullptr, st);
  KRS_CHECK_LAUNCH("krs_shard_unpack");
  return KRS_OK;
}

extern "C" int krs_shard_unpack_static(const int32_t* packed, int n_sources, int64_t cap_lookups, int64_t cap_segments,
                                       int weighted, int32_t* rows, float* w, int32_t* offsets, int64_t* stats,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  KRS_REQUIRE(n_sources >= 1 && n_sources <= kMaxShards, "shard_unpack_static: n_sources must be in [1, %d]", kMaxShards);
  KRS_REQUIRE(cap_lookups > 0 && cap_segments > 0 && cap_lookups % 4 == 0 && cap_segments % 4 == 0,
              "shard_unpack_static: capacities must be positive multiples of 4");
  KRS_REQUIRE(packed && rows && offsets && (!weighted || w), "shard_unpack_static: null argument");
  KRS_REQUIRE(cap_lookups * n_sources < 0x7fffffffLL, "shard_unpack_static: lookups must fit int32");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  UnpackStaticParams p;
  p.packed = packed; p.n_src = n_sources; p.weighted = weighted != 0; p.cap_l = cap_lookups; p.cap_s = cap_segments;
  p.block_words = kStaticHeader + cap_lookups * (1 + p.weighted) + cap_segments;
  p.rows = rows; p.w = w; p.offsets = offsets; p.stats = stats;
  const int64_t n_seg = cap_segments * n_sources;
  if (workspace_bytes < krs_shard_unpack_workspace_bytes(n_seg) || !workspace)
    return fail(KRS_ERR_WORKSPACE, "shard_unpack_static: workspace too small");
  const int64_t span = std::max<int64_t>(cap_lookups * n_sources, n_seg + 1);
  hipLaunchKernelGGL(unpack_static_kernel, dim3((unsigned)ceil_div(span, 256)), dim3(256), 0, st, p);
  scan::exclusive(offsets, offsets, n_seg + 1, reinterpret_cast<int32_t*>(workspace), nullptr, st);
  KRS_CHECK_LAUNCH("krs_shard_unpack_static");
  return KRS_OK;
}

extern "C" int krs_shard_combine(const void* partials, const int32_t* bag_seg, int batch, int n_feats, int n_shards,
                                 int dim, int dtype, void* out, int64_t out_ld, void* stream) {
  KRS_REQUIRE(bag_seg && out, "shard_combine: null argument");
  KRS_REQUIRE(dtype == KRS_F32 || dtype == KRS_BF16, "shard_combine: bad dtype");
  KRS_REQUIRE(n_shards >= 1 && n_shards <= kMaxShards && dim >= 1, "shard_combine: bad sizes");
  if (batch == 0 || n_feats == 0) return KRS_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const auto al = [](const void* q, int64_t a) { return (reinterpret_cast<uintptr_t>(q) % a) == 0; };
  if (dtype == KRS_BF16) {
    if (dim % 8 == 0 && out_ld % 8 == 0 && al(partials, 16) && al(out, 16)) {
      const int64_t total = (int64_t)batch * n_feats * (dim / 8);
      hipLaunchKernelGGL((combine_kernel<uint16_t, 8>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                         reinterpret_cast<const uint16_t*>(partials), bag_seg, batch, n_feats, n_shards, dim,
                         reinterpret_cast<uint16_t*>(out), out_ld);
    } else {
      const int64_t total = (int64_t)batch * n_feats * dim;
      hipLaunchKernelGGL((combine_kernel<uint16_t, 1>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                         reinterpret_cast<const uint16_t*>(partials), bag_seg, batch, n_feats, n_shards, dim,
                         reinterpret_cast<uint16_t*>(out), out_ld);
    }
  } else {
    if (dim % 4 == 0 && out_ld % 4 == 0 && al(partials, 16) && al(out, 16)) {
      const int64_t total = (int64_t)batch * n_feats * (dim / 4);
      hipLaunchKernelGGL((combine_kernel<float, 4>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                         reinterpret_cast<const float*>(partials), bag_seg, batch, n_feats, n_shards, dim,
                         reinterpret_cast<float*>(out), out_ld);
    } else {
      const int64_t total = (int64_t)batch * n_feats * dim;
      hipLaunchKernelGGL((combine_kernel<float, 1>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                         reinterpret_cast<const float*>(partials), bag_seg, batch, n_feats, n_shards, dim,
                         reinterpret_cast<float*>(out), out_ld);
    }
  }
  KRS_CHECK_LAUNCH("krs_shard_combine");
  return KRS_OK;
}

extern "C" int krs_publish_i64(const int64_t* src, int n, int64_t* host_dst, int64_t seq, void* stream) {
  KRS_REQUIRE(src && host_dst && n >= 1 && n <= 255, "publish: bad arguments");
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, n,
                     (volatile int64_t*)host_dst, seq);
  KRS_CHECK_LAUNCH("krs_publish_i64");
  return KRS_OK;
}
