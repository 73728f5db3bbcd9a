// K2 -- index-scatter gradient of the fused embedding bag (backward of K1).
//
// Replaces the autodiff of ops.take/multiply/sum (the dense [V, D] scatter-add
// restated by the reference at keras_rs/src/layers/embedding/jax/test_utils.py:395-417,
// summed per table over the features that share it, :450-468) and, in the fused
// forms, the per-table optimizer step of jax/test_utils.py:474-497.
//
// Plan (once per batch of ids, independent of the gradient values):
//   keys[p] = tables[t(f)].row_base + ids[p]   (u32; invalid ids -> 0xffffffff)
//   vals[p] = (bag(p) << 32) | p               (u64)
//   stable LSD radix sort of (keys, vals) over ceil(log2(total_rows)) bits (namespace rs below: own kernels;
//   dense bags whose tables form contiguous runs of positions are sorted per table, over the id bits alone).
//   head flags -> exclusive scan -> segment list (first sorted position of every run of equal
//   keys), and the work items of the segments longer than kLongSeg lookups.
// Apply: a group of LPR lanes (one 16-byte piece of the gradient row per lane, as in K1) per
//   SEGMENT = per touched table row: it issues the loads of the row (and optimizer slots) it will
//   update, gathers and sums the segment's gradient rows, and writes the finished row once:
//   no atomics, one owner per row, contributions summed in ascending p
//   => run-to-run bit-identical.  The row write is the dense gradient row, the compact
//   (unique_rows, grads) entry, or the SGD / Adagrad / Adam / FTRL update of the table row in place.
//   Hot rows (segments longer than kLongSeg) are summed by whole workgroups in chunks of kChunk
//   lookups; rows spanning several chunks are finished from fp32 partial rows in chunk order.
//   Table / slot rows are read and written non-temporally (each is touched once per launch).
// Algorithmic bytes: bags*D*s_g + nnz*(4+8) + U*(2*D*s_t [+ 2*D*4 per slot plane]).
#include <cstdlib>
#include <cstring>

#include "krs_common.h"
#include "krs_scan.h"

// The apply kernels are instantiated per (gradient type, table type, lanes per row, optimizer, weights, scale): 3 dtype
// pairs x 4 widths x 7 kernels x 7 modes = ~590 kernels (round 4: ~1000 -- the round-1 per-segment kernel and the
// fp32-gradient / bf16-table pair are gone).  keras_rs_amd/build.py compiles this file four times, in parallel:
// KRS_BWD_PART 0 = the plan + the dense / sparse / SGD forms, 1 = Adagrad and row-wise Adagrad, 2 = Adam, 3 = FTRL
// (entry points outside a part are left out of it).  Undefined = the whole file in one object.
#ifndef KRS_BWD_PART
#define KRS_BWD_PART -1
#endif
#define KRS_BWD_HAS(part) (KRS_BWD_PART < 0 || KRS_BWD_PART == (part))

namespace krs {
#if KRS_BWD_HAS(0)
// krs_embed_set_option(KRS_EMBED_OPT_PLAN, v): 0 = table-segmented sort where the layout allows it (default), 1 = always the global sort
int g_plan_variant = 0;
#else
extern int g_plan_variant;
#endif
namespace {

constexpr uint32_t kInvalidKey = 0xffffffffu;
constexpr int kLongSeg = 128;   // segments longer than this are summed by whole workgroups
constexpr int kChunk = 2048;    // ... in chunks of this many lookups, one workgroup each
constexpr int kPartialBytes = 2048;  // fp32 partial row of a chunk (row bytes <= 1024 on the vector path)

// one workgroup's share of a long segment
struct LongItem {
  uint32_t seg;       // segment index
  uint32_t chunk;     // which kChunk-sized piece of it
  uint32_t partial;   // slot in the partial-row buffer, or ~0u when the segment is a single chunk
};
// a long segment that spans several chunks: its partial rows are summed in chunk order afterwards
struct MultiSeg {
  uint32_t seg, partial_base, n_chunks;
};

struct PlanLayout {
  uint32_t* keys_in;      // dead after the sort -> reused as head flags
  uint32_t* keys_sorted;
  uint64_t* vals_in;      // dead after the sort -> its second half is reused as seg_start
  uint64_t* vals_sorted;
  uint32_t* seg_start;    // = vals_in, next n words: first sorted position of every segment
  uint32_t* n_seg;        // number of segments (a trailing run of invalid keys counts as one)
  uint32_t* n_long;       // number of LongItems (device scalar); [1] partial rows handed out; [2] MultiSegs
  uint32_t* sort_mode;    // 0 = global sort (out-of-range lookups form ONE trailing run), != 0 = table-segmented sort
  LongItem* long_list;    // work items of the segments longer than kLongSeg (any order)
  MultiSeg* multi_list;   // segments longer than kChunk
  float* partials;        // [<= 2 * nnz / kChunk + 2] fp32 partial rows, kPartialBytes apart
  void* temp;
  size_t temp_bytes;
  size_t total_bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- plan: the sort (LSD radix sort of (row key, bag << 32 | position) pairs, written for this plan) -----------
// Digits of up to 10 bits, ceil(bits / 10) passes over ceil(log2(total_rows + 1)) key bits (C3: 25 bits = 3 passes,
// against 4 passes of 8 bits over the same bits in a general-purpose device sort).  One pass = per-tile digit
// histogram -> exclusive scan of the [digit][tile] counts (krs_scan.h) -> stable scatter.  A tile is 4096 consecutive
// lookups, one workgroup; wave w owns the w-th quarter, so the (wave, round, lane) order IS the input order and
// the ranks below keep equal keys in input order (stability = ascending position inside a row's segment = a fixed
// summation order in the apply kernels).  Dense bags: the first pass computes keys and values from the ids on the
// fly (no key generation kernel, no round trip of 12 bytes per lookup through HBM).
namespace rs {
#ifndef KRS_SORT_TILE
#define KRS_SORT_TILE 4096       // (development builds vary the tile / workgroup shape: scripts/exp/build_variants.sh)
#endif
#ifndef KRS_SORT_THREADS
#define KRS_SORT_THREADS 512
#endif
constexpr int kTile = KRS_SORT_TILE, kMaxBits = 10, kMaxBins = 1 << kMaxBits;
constexpr int kHistThreads = 256, kHistItems = kTile / kHistThreads;
constexpr int kThreads = KRS_SORT_THREADS, kWaves = kThreads / 64, kItems = kTile / kThreads;   // scatter: 8 waves x 512 lookups
constexpr int kGenFeats = 256;   // features whose descriptors the generating pass caches in LDS
constexpr int kMaxProb = 128;    // tables (problems) of the table-segmented sort

struct Gen {   // key generation for dense bags (first pass)
  const krs_table* tables;
  const krs_feature* feats;
  int n_feats;
  const void* ids;
  int id64;
  int batch;
  int* err_flag;
};

struct Pass {
  const uint32_t* keys_in;     // null in a generating pass
  const uint64_t* vals_in;
  uint32_t* keys_out;
  uint64_t* vals_out;
  int32_t* counts;             // [bins][tiles]: histogram, then exclusive offsets
  int64_t nnz;
  int n_tiles;
  int shift, bits;
  Gen gen;
};

// per-feature constants of the generating pass, in LDS
struct GenLds {
  uint32_t base[kGenFeats + 1];   // first lookup position of the feature (nnz < 2^31)
  uint32_t hot[kGenFeats];
  uint32_t row_base[kGenFeats];   // total_rows < 2^32 (checked by the plan)
  uint32_t vocab[kGenFeats];
};
__device__ __forceinline__ void load_gen(const Pass& p, GenLds& g, int n_threads) {
  for (int i = threadIdx.x; i <= p.gen.n_feats; i += n_threads) {
    if (i < p.gen.n_feats) {
      const krs_feature ft = p.gen.feats[i];
      const krs_table tb = p.gen.tables[ft.table];
      g.base[i] = (uint32_t)ft.ids_base;
      g.hot[i] = (uint32_t)ft.hot;
      g.row_base[i] = (uint32_t)tb.row_base;
      g.vocab[i] = (uint32_t)tb.vocab;
    } else {
      g.base[i] = (uint32_t)p.nnz;
    }
  }
}
// (key, value) of lookup position q; dense bags, features laid out one after the other
__device__ __forceinline__ void generate(const Pass& p, const GenLds& g, uint32_t q, uint32_t& key, uint64_t& val,
                                         bool& bad) {
  int lo = 0, hi = p.gen.n_feats;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (g.base[mid] <= q) lo = mid; else hi = mid;
  }
  const uint32_t bag = (uint32_t)lo * (uint32_t)p.gen.batch + (q - g.base[lo]) / g.hot[lo];
  const int64_t id = ld_index(p.gen.ids, p.gen.id64, q);
  key = kInvalidKey;
  if (id >= 0 && id < (int64_t)g.vocab[lo]) key = g.row_base[lo] + (uint32_t)id;
  else bad = true;
  val = ((uint64_t)bag << 32) | (uint64_t)q;
}

template <bool GEN>
__global__ __launch_bounds__(kHistThreads) void hist_kernel(const Pass p) {
  __shared__ int h[kMaxBins];
  __shared__ GenLds g;
  const int bins = 1 << p.bits;
  for (int i = threadIdx.x; i < bins; i += kHistThreads) h[i] = 0;
  if constexpr (GEN) load_gen(p, g, kHistThreads);
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kTile;
  bool bad = false;
  for (int it = 0; it < kHistItems; ++it) {
    const int64_t q = base + it * kHistThreads + threadIdx.x;
    if (q < p.nnz) {
      uint32_t key;
      if constexpr (GEN) {
        uint64_t v;
        generate(p, g, (uint32_t)q, key, v, bad);
      } else {
        key = p.keys_in[q];
      }
      atomicAdd(&h[(key >> p.shift) & (bins - 1)], 1);
    }
  }
  if constexpr (GEN)
    if (bad && p.gen.err_flag) atomicOr(p.gen.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += kHistThreads) p.counts[(int64_t)i * p.n_tiles + blockIdx.x] = h[i];
}

template <bool GEN>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const Pass p) {
  // 74 KB of LDS: two workgroups (16 waves) per CU
  __shared__ uint16_t cnt[kWaves][kMaxBins];   // per wave: elements of each digit seen so far -> (wave, digit) start
  __shared__ uint16_t tile_excl[kMaxBins];     // first slot of every digit in the tile's sorted image
  __shared__ int gbase[kMaxBins];              // first output slot of the tile's elements of every digit
  __shared__ uint32_t skey[kTile];             // the tile, sorted by digit (stable): consecutive threads then
  __shared__ uint64_t sval[kTile];             // write consecutive output slots inside a digit's run
  __shared__ int wtot[kWaves];
  __shared__ GenLds g;
  const int bins = 1 << p.bits;
  for (int i = threadIdx.x; i < kWaves * kMaxBins; i += kThreads) (&cnt[0][0])[i] = 0;
  if constexpr (GEN) load_gen(p, g, kThreads);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile0 = (int64_t)blockIdx.x * kTile;
  const int64_t base = tile0 + (int64_t)wave * (kTile / kWaves);
  uint32_t key[kItems];
  uint64_t val[kItems];
  int rank[kItems];
  bool bad = false;
  volatile uint16_t* mine = cnt[wave];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t q = base + it * 64 + lane;
    const bool live = q < p.nnz;
    key[it] = kInvalidKey;
    val[it] = 0;
    if (live) {
      if constexpr (GEN) {
        generate(p, g, (uint32_t)q, key[it], val[it], bad);
      } else {
        key[it] = p.keys_in[q];
        val[it] = p.vals_in[q];
      }
    }
    const int d = (int)((key[it] >> p.shift) & (bins - 1));
    // lanes of this round with the same digit
    unsigned long long peers = __ballot(live);
    for (int b = 0; b < p.bits; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    int r = 0, c = 0;
    if (live) {
      r = __popcll(peers & ((1ULL << lane) - 1ULL));
      const int leader = __ffsll((long long)peers) - 1;
      if (lane == leader) {
        c = mine[d];
        mine[d] = (uint16_t)(c + __popcll(peers));
      }
      c = __shfl(c, leader, 64);
    }
    rank[it] = c + r;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // digit totals of the tile -> exclusive scan over the digits (bins <= 1024 = 2 per thread)
  int tot[2], run = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x * 2 + k;
    tot[k] = 0;
    if (i < bins)
      for (int w = 0; w < kWaves; ++w) tot[k] += cnt[w][i];
    run += tot[k];
  }
  int x = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wtot[wave] = x;
  __syncthreads();
  int excl = x - run;
  for (int w = 0; w < wave; ++w) excl += wtot[w];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x * 2 + k;
    if (i < bins) {
      tile_excl[i] = (uint16_t)excl;
      gbase[i] = p.counts[(int64_t)i * p.n_tiles + blockIdx.x];
      int o = excl;
      for (int w = 0; w < kWaves; ++w) {
        const int t = cnt[w][i];
        cnt[w][i] = (uint16_t)o;
        o += t;
      }
      excl += tot[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t q = base + it * 64 + lane;
    if (q < p.nnz) {
      const int d = (int)((key[it] >> p.shift) & (bins - 1));
      const int lp = cnt[wave][d] + rank[it];
      skey[lp] = key[it];
      sval[lp] = val[it];
    }
  }
  __syncthreads();
  const int n_here = (int)min<int64_t>(kTile, p.nnz - tile0);
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int i = it * kThreads + threadIdx.x;
    if (i < n_here) {
      const uint32_t k = skey[i];
      const int d = (int)((k >> p.shift) & (bins - 1));
      const int pos = gbase[d] + (i - (int)tile_excl[d]);
      p.keys_out[pos] = k;
      p.vals_out[pos] = sval[i];
    }
  }
  (void)bad;   // the histogram pass of the same data reports out-of-range ids
}

// ---- the same sort, SEGMENTED BY TABLE (round 3) ------------------------------------------------------------
// Dense bags arrive feature-major, so the lookups of one table already form ONE contiguous run of positions (when
// the features of a table are neighbours and tables ascend with the features -- what DistributedEmbedding builds).
// Sorting by the global row then needs no pass over the table bits: every table's run is an independent PROBLEM,
// sorted in place by the id alone -- ceil(log2(max vocab + 1)) bits, 20 at C3 = TWO passes instead of three -- and
// the problems' sorted runs, one after the other, are exactly the global order.  Tiles never straddle problems
// (a problem's last tile may be partial); the count matrix is laid out [problem][digit][tile of the problem], so
// ONE exclusive scan over it still yields every tile's output offsets (a problem's lookups stay inside its run).
// Intermediate passes carry (local key, position) = 8 bytes per lookup instead of (key, bag << 32 | position) = 12;
// the LAST pass writes what the apply kernels read: the global key (row_base + id; all ones for an invalid id) and
// the 64-bit value, whose bag is recomputed from the position (feature constants in LDS, as the first pass does).
// Per lookup 40 bytes move instead of 76.  Out-of-range ids sort to the END OF THEIR TABLE'S RUN (sentinel = all
// ones in the key bits), not to the end of the array: the apply kernels skip invalid segments wherever they are;
// the compact (sparse) form, whose output is indexed by segment, keeps the global sort.
struct Seg {
  int n;                                  // problems
  uint32_t lookup_start[kMaxProb + 1];    // first lookup position of problem i; [n] = nnz
  uint32_t tile_start[kMaxProb + 1];      // first tile of problem i; [n] = tiles in all
  uint32_t row_base[kMaxProb];            // global row of the table's row 0
};
struct SegPass {
  const uint32_t* keys_in;     // FIRST pass: null (keys come from the ids)
  const uint32_t* pos_in;
  uint32_t* keys_out;          // LAST pass: global keys
  uint32_t* pos_out;           // intermediate passes
  uint64_t* vals_out;          // LAST pass
  int32_t* counts;
  int64_t nnz;
  int shift, bits, key_bits;
  Gen gen;
  Seg seg;
};
__device__ __forceinline__ int seg_problem(const Seg& sg, uint32_t tile) {
  int lo = 0, hi = sg.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sg.tile_start[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ void load_gen_seg(const SegPass& p, GenLds& g, int n_threads) {
  for (int i = threadIdx.x; i <= p.gen.n_feats; i += n_threads) {
    if (i < p.gen.n_feats) {
      const krs_feature ft = p.gen.feats[i];
      const krs_table tb = p.gen.tables[ft.table];
      g.base[i] = (uint32_t)ft.ids_base;
      g.hot[i] = (uint32_t)ft.hot;
      g.row_base[i] = (uint32_t)tb.row_base;
      g.vocab[i] = (uint32_t)tb.vocab;
    } else {
      g.base[i] = (uint32_t)p.nnz;
    }
  }
}
__device__ __forceinline__ int gen_feature(const GenLds& g, int n_feats, uint32_t q) {
  int lo = 0, hi = n_feats;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (g.base[mid] <= q) lo = mid; else hi = mid;
  }
  return lo;
}
// local key of lookup position q: its id, or all ones in the key bits when the id is out of range
__device__ __forceinline__ uint32_t seg_local_key(const SegPass& p, const GenLds& g, uint32_t q, bool& bad) {
  const int f = gen_feature(g, p.gen.n_feats, q);
  const int64_t id = ld_index(p.gen.ids, p.gen.id64, q);
  if (id >= 0 && id < (int64_t)g.vocab[f]) return (uint32_t)id;
  bad = true;
  return (1u << p.key_bits) - 1u;
}

// (Round 5 measured one workgroup per EIGHT consecutive tiles here, so that a digit's counters leave as 32 contiguous bytes:
//  the write counter of this kernel fell from 121 MB to 14 MB per launch -- one tile's 1024 counters are 1024 scattered
//  4-byte stores into the [digit][tile] matrix -- and the plan got SLOWER, 558 -> 582 us with 1024-thread groups (1815 us
//  with 256-thread ones): the partial-sector writes are absorbed by L2, the kernel wants its 3400 independent workgroups.)
template <bool FIRST>
__global__ __launch_bounds__(kHistThreads) void hist_seg_kernel(const SegPass p) {
  __shared__ int h[kMaxBins];
  __shared__ GenLds g;
  const int bins = 1 << p.bits;
  for (int i = threadIdx.x; i < bins; i += kHistThreads) h[i] = 0;
  if constexpr (FIRST) load_gen_seg(p, g, kHistThreads);
  __syncthreads();
  const int pr = seg_problem(p.seg, blockIdx.x);
  const uint32_t tl = blockIdx.x - p.seg.tile_start[pr], nt = p.seg.tile_start[pr + 1] - p.seg.tile_start[pr];
  const int64_t base = (int64_t)p.seg.lookup_start[pr] + (int64_t)tl * kTile;
  const int64_t end = min<int64_t>(base + kTile, p.seg.lookup_start[pr + 1]);
  bool bad = false;
  uint32_t key[kHistItems];
  if constexpr (FIRST) {
    int64_t idv[kHistItems];
#pragma unroll
    for (int it = 0; it < kHistItems; ++it)
      idv[it] = ld_index(p.gen.ids, p.gen.id64, min<int64_t>(base + it * kHistThreads + threadIdx.x, end - 1));
    // (a tile lies inside ONE feature almost always -- tiles never straddle tables, and a table's features are few: the
    //  per-key search over the feature bases is then one search per tile; round 5)
    const int f_lo = gen_feature(g, p.gen.n_feats, (uint32_t)base), f_hi = gen_feature(g, p.gen.n_feats, (uint32_t)(end - 1));
#pragma unroll
    for (int it = 0; it < kHistItems; ++it) {
      const int64_t q = min<int64_t>(base + it * kHistThreads + threadIdx.x, end - 1);
      const int f = f_lo == f_hi ? f_lo : gen_feature(g, p.gen.n_feats, (uint32_t)q);
      const bool valid = idv[it] >= 0 && idv[it] < (int64_t)g.vocab[f];
      key[it] = valid ? (uint32_t)idv[it] : (1u << p.key_bits) - 1u;
      bad = bad || (!valid && base + it * kHistThreads + threadIdx.x < end);
    }
  } else {
#pragma unroll
    for (int it = 0; it < kHistItems; ++it)
      key[it] = p.keys_in[min<int64_t>(base + it * kHistThreads + threadIdx.x, end - 1)];
  }
#pragma unroll
  for (int it = 0; it < kHistItems; ++it)
    if (base + it * kHistThreads + threadIdx.x < end) atomicAdd(&h[(key[it] >> p.shift) & (bins - 1)], 1);
  if constexpr (FIRST)
    if (bad && p.gen.err_flag) atomicOr(p.gen.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
  __syncthreads();
  int32_t* dst = p.counts + (int64_t)bins * p.seg.tile_start[pr];
  for (int i = threadIdx.x; i < bins; i += kHistThreads) dst[(int64_t)i * nt + tl] = h[i];
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kThreads) void scatter_seg_kernel(const SegPass p) {
  __shared__ uint16_t cnt[kWaves][kMaxBins];
  __shared__ uint16_t tile_excl[kMaxBins];
  __shared__ int gbase[kMaxBins];
  __shared__ uint32_t skey[kTile];
  __shared__ uint32_t spos[kTile];
  __shared__ int wtot[kWaves];
  __shared__ GenLds g;
  const int bins = 1 << p.bits;
  for (int i = threadIdx.x; i < kWaves * kMaxBins; i += kThreads) (&cnt[0][0])[i] = 0;
  if constexpr (FIRST || LAST) load_gen_seg(p, g, kThreads);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pr = seg_problem(p.seg, blockIdx.x);
  const uint32_t tl = blockIdx.x - p.seg.tile_start[pr], nt = p.seg.tile_start[pr + 1] - p.seg.tile_start[pr];
  const int64_t tile0 = (int64_t)p.seg.lookup_start[pr] + (int64_t)tl * kTile;
  const int64_t tile_end = min<int64_t>(tile0 + kTile, p.seg.lookup_start[pr + 1]);
  const int64_t base = tile0 + (int64_t)wave * (kTile / kWaves);
  uint32_t key[kItems], pos[kItems];
  int rank[kItems];
  bool bad = false;
  volatile uint16_t* mine = cnt[wave];
  // all of the thread's keys are requested before the first is ranked (addresses clamped to the tile, dead slots
  // masked afterwards): one exposed memory latency per tile instead of one per round of the ranking loop below,
  // whose LDS counter updates form a dependent chain the loads could not be scheduled across
  if constexpr (FIRST) {
    int64_t idv[kItems];
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int64_t q = min<int64_t>(base + it * 64 + lane, tile_end - 1);
      idv[it] = ld_index(p.gen.ids, p.gen.id64, q);
    }
    const int f_lo = gen_feature(g, p.gen.n_feats, (uint32_t)tile0), f_hi = gen_feature(g, p.gen.n_feats, (uint32_t)(tile_end - 1));
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int64_t q = min<int64_t>(base + it * 64 + lane, tile_end - 1);
      const int f = f_lo == f_hi ? f_lo : gen_feature(g, p.gen.n_feats, (uint32_t)q);   // (one search per tile, see hist_seg_kernel)
      const bool valid = idv[it] >= 0 && idv[it] < (int64_t)g.vocab[f];
      key[it] = valid ? (uint32_t)idv[it] : (1u << p.key_bits) - 1u;
      pos[it] = (uint32_t)q;
    }
  } else {
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int64_t q = min<int64_t>(base + it * 64 + lane, tile_end - 1);
      key[it] = p.keys_in[q];
      pos[it] = p.pos_in[q];
    }
  }
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t q = base + it * 64 + lane;
    const bool live = q < tile_end;
    if (!live) { key[it] = 0xffffffffu; pos[it] = 0; }
    const int d = (int)((key[it] >> p.shift) & (bins - 1));
    unsigned long long peers = __ballot(live);
    for (int b = 0; b < p.bits; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    int r = 0, c = 0;
    if (live) {
      r = __popcll(peers & ((1ULL << lane) - 1ULL));
      const int leader = __ffsll((long long)peers) - 1;
      if (lane == leader) {
        c = mine[d];
        mine[d] = (uint16_t)(c + __popcll(peers));
      }
      c = __shfl(c, leader, 64);
    }
    rank[it] = c + r;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  int tot[2], run = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x * 2 + k;
    tot[k] = 0;
    if (i < bins)
      for (int w = 0; w < kWaves; ++w) tot[k] += cnt[w][i];
    run += tot[k];
  }
  int x = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wtot[wave] = x;
  __syncthreads();
  int excl = x - run;
  for (int w = 0; w < wave; ++w) excl += wtot[w];
  const int32_t* cbase = p.counts + (int64_t)bins * p.seg.tile_start[pr];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x * 2 + k;
    if (i < bins) {
      tile_excl[i] = (uint16_t)excl;
      gbase[i] = cbase[(int64_t)i * nt + tl];
      int o = excl;
      for (int w = 0; w < kWaves; ++w) {
        const int t = cnt[w][i];
        cnt[w][i] = (uint16_t)o;
        o += t;
      }
      excl += tot[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int64_t q = base + it * 64 + lane;
    if (q < tile_end) {
      const int d = (int)((key[it] >> p.shift) & (bins - 1));
      const int lp = cnt[wave][d] + rank[it];
      skey[lp] = key[it];
      spos[lp] = pos[it];
    }
  }
  __syncthreads();
  const int n_here = (int)(tile_end - tile0);
  const uint32_t sentinel = (1u << p.key_bits) - 1u, rb = p.seg.row_base[pr];
  // (last pass: a key's original position lies anywhere in its TABLE's run; a table looked up by one feature -- the usual
  //  case -- needs one search per tile instead of one per key)
  int pf_lo = 0, pf_hi = 1;
  if constexpr (LAST) {
    pf_lo = gen_feature(g, p.gen.n_feats, p.seg.lookup_start[pr]);
    pf_hi = gen_feature(g, p.gen.n_feats, p.seg.lookup_start[pr + 1] - 1);
  }
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int i = it * kThreads + threadIdx.x;
    if (i < n_here) {
      const uint32_t k = skey[i], q = spos[i];
      const int d = (int)((k >> p.shift) & (bins - 1));
      const int o = gbase[d] + (i - (int)tile_excl[d]);
      if constexpr (LAST) {
        const int f = pf_lo == pf_hi ? pf_lo : gen_feature(g, p.gen.n_feats, q);
        const uint32_t bag = (uint32_t)f * (uint32_t)p.gen.batch + (q - g.base[f]) / g.hot[f];
        p.keys_out[o] = k == sentinel ? kInvalidKey : rb + k;
        p.vals_out[o] = ((uint64_t)bag << 32) | (uint64_t)q;
      } else {
        p.keys_out[o] = k;
        p.pos_out[o] = q;
      }
    }
  }
  (void)bad;
}

inline int n_passes(unsigned bits) { return (int)((bits + kMaxBits - 1) / kMaxBits); }
inline size_t temp_bytes(int64_t nnz) {
  const int64_t tiles = ceil_div(nnz > 0 ? nnz : 1, kTile) + kMaxProb;   // (every problem may end in a partial tile)
  const size_t counts = (size_t)kMaxBins * tiles * sizeof(int32_t);
  return align_up(counts, 256) + align_up(scan::workspace_bytes((int64_t)kMaxBins * tiles), 256) +
         align_up(scan::workspace_bytes(nnz), 256);
}
}  // namespace rs

PlanLayout plan_layout(void* ws, int64_t nnz, bool need_temp = false) {
  PlanLayout l;
  char* p = reinterpret_cast<char*>(ws);
  size_t o = 0;
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
  l.keys_in = reinterpret_cast<uint32_t*>(p + o); o += align_up(n * 4, 256);
  l.keys_sorted = reinterpret_cast<uint32_t*>(p + o); o += align_up(n * 4, 256);
  l.vals_in = reinterpret_cast<uint64_t*>(p + o); o += align_up(n * 8 + 8, 256);
  l.vals_sorted = reinterpret_cast<uint64_t*>(p + o); o += align_up(n * 8, 256);
  l.n_seg = reinterpret_cast<uint32_t*>(p + o);
  l.n_long = l.n_seg + 1;
  l.sort_mode = l.n_seg + 8; o += 256;
  // every long segment has <= len / kChunk + 1 items; there are <= n / kLongSeg long segments
  l.long_list = reinterpret_cast<LongItem*>(p + o); o += align_up((n / kLongSeg + n / kChunk + 2) * sizeof(LongItem), 256);
  l.multi_list = reinterpret_cast<MultiSeg*>(p + o); o += align_up((n / kChunk + 2) * sizeof(MultiSeg), 256);
  l.partials = reinterpret_cast<float*>(p + o); o += align_up((2 * (n / kChunk) + 2) * (size_t)kPartialBytes, 256);
  l.seg_start = reinterpret_cast<uint32_t*>(l.vals_in) + n;
  l.temp = p + o;
  l.temp_bytes = need_temp ? rs::temp_bytes(nnz) : 0;
  l.total_bytes = o + l.temp_bytes;
  return l;
}

// ---- plan: key generation ---------------------------------------------------
struct KeyParams {
  const krs_table* tables;
  const krs_feature* feats;
  int n_feats;
  const void* ids;
  int id64;
  const void* offsets;
  int off64;
  int batch;
  uint32_t* keys;
  uint64_t* vals;
  int* err_flag;
};

// 16 lanes per bag; lanes stride over the bag's positions (coalesced writes).
__global__ __launch_bounds__(256) void bag_keys_kernel(const KeyParams p) {
  const int64_t bag = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
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
  int oob = 0;
  for (int64_t q = s + sub; q < e; q += 16) {
    const int64_t id = ld_index(p.ids, p.id64, q);
    uint32_t key = kInvalidKey;
    if (id >= 0 && id < tb.vocab)
      key = (uint32_t)(tb.row_base + id);
    else
      oob = 1;
    p.keys[q] = key;
    p.vals[q] = ((uint64_t)bag << 32) | (uint64_t)(uint32_t)q;
  }
  if (oob && p.err_flag) atomicOr(p.err_flag, KRS_FLAG_ID_OUT_OF_RANGE);
}

// ---- apply ------------------------------------------------------------------
enum ApplyMode { kDense = 0, kSgd = 1, kAdagrad = 2, kSparse = 3, kAdam = 4, kFtrl = 5, kAdagradRow = 6 };
constexpr bool mode_is_fused(int m) { return m == kSgd || m == kAdagrad || m == kAdam || m == kFtrl || m == kAdagradRow; }
// full-size fp32 slot planes [V, D] of a mode (row-wise Adagrad keeps ONE fp32 per row instead: slot = [V])
constexpr int mode_slots(int m) { return m == kAdagrad ? 1 : ((m == kAdam || m == kFtrl) ? 2 : 0); }

// Row-wise Adagrad (opt-in, NOT the reference's rule: the FBGEMM / TorchRec "rowwise_adagrad" form without
// epsilon): acc[row] += mean_j g_j^2;  w_j -= lr * g_j / sqrt(acc[row]).  The exact form moves 2 x D x 4
// accumulator bytes per touched row (two thirds of K2's traffic at C3), this one 8.  `ss` = this lane's share of
// sum_j g_j^2; the group's lanes (LPR, a power of two, all active) add theirs by butterfly shuffles.
template <int LPR>
__device__ __forceinline__ float row_sumsq(float ss) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  return ss;
}

// optimizer constants shared by every table of a call (the learning rate is per table)
struct Hyper {
  float a, b, c, d;  // Adam: beta_1, beta_2, epsilon, bias-correction factor; FTRL: lr_power, l1, l2, beta
};

// One element of the fused row update; g = the row's summed gradient.  s0 / s1 = slot planes.
//   SGD / Adagrad: jax/test_utils.py:474-497.
//   Adam (lazy: touched rows only; keras.optimizers.Adam.update_step as named by
//     jax/config_conversion.py:256-265): alpha = lr * sqrt(1 - b2^t) / (1 - b1^t) (h.d carries the
//     factor), m += (g - m)(1 - b1), v += (g^2 - v)(1 - b2), w -= alpha * m / (sqrt(v) + eps).
//   FTRL (keras.optimizers.Ftrl.update_step, options of jax/config_conversion.py:266-283; no
//     l2 shrinkage): n' = n + g^2; z += g - (n'^-p - n^-p) / lr * w;
//     w = (clip(z, -l1, l1) - z) / (n'^-p / lr + 2 (l2 + beta / (2 lr))); n = n'.
template <int MODE>
__device__ __forceinline__ void row_update(float& w, float& s0, float& s1, float g, float lr, const Hyper& h) {
  if constexpr (MODE == kSgd) {
    w = w - lr * g;
  } else if constexpr (MODE == kAdagrad) {
    s0 = fmaf(g, g, s0);
    w = w - lr * g / sqrtf(s0);
  } else if constexpr (MODE == kAdam) {
    s0 = s0 + (g - s0) * (1.0f - h.a);
    s1 = s1 + (g * g - s1) * (1.0f - h.b);
    w = w - (lr * h.d) * s0 / (sqrtf(s1) + h.c);
  } else if constexpr (MODE == kFtrl) {
    const float n_new = s0 + g * g;
    // the default power -0.5 is an exactly rounded square root on both sides of the parity check
    const float pn = h.a == -0.5f ? sqrtf(n_new) : powf(n_new, -h.a);
    const float po = h.a == -0.5f ? sqrtf(s0) : powf(s0, -h.a);
    s1 = s1 + g - (pn - po) / lr * w;
    const float quad = pn / lr + 2.0f * (h.c + h.d / (2.0f * lr));
    const float zc = fminf(fmaxf(s1, -h.b), h.b);
    w = (zc - s1) / quad;
    s0 = n_new;
  }
}

struct ApplyParams {
  const krs_table* tables;  // dense: gradient buffers; fused: the tables themselves
  int n_tables;
  const krs_feature* feats;
  int n_feats;
  const float* weights;
  const float* bag_scale;
  const void* grad;
  int64_t grad_ld;
  int batch;
  int dim;
  int64_t nnz;
  const uint32_t* keys;
  const uint64_t* vals;
  const uint32_t* seg_start;   // first sorted position of every segment
  const uint32_t* n_seg;       // device scalar
  const uint32_t* n_long;      // device scalars: items, partial rows, multi-chunk segments
  const LongItem* long_list;
  const MultiSeg* multi_list;
  float* partials;
  int64_t* unique_rows;        // sparse
  float* row_grads;            // sparse
  Hyper hyper;                 // Adam / FTRL
  const float* hyper_d_dev;    // Adam: bias-correction factor read from device memory at run time (NULL: hyper.d)
};

// The constants a launch runs with: Adam's bias-correction factor comes from device memory when the caller keeps it there
// (krs_embed_bag_bwd_fused_adam_dyn: a step replayed from a HIP graph reads the value of THIS replay, not the capture's).
template <int MODE>
__device__ __forceinline__ Hyper live_hyper(const ApplyParams& p) {
  Hyper h = p.hyper;
  if constexpr (MODE == kAdam) {
    if (p.hyper_d_dev) h.d = *p.hyper_d_dev;
  }
  return h;
}

template <typename T>
struct Piece;  // 16 bytes of gradient elements
template <>
struct Piece<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
    f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
  }
};
template <>
struct Piece<uint16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&f)[8]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
};

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));

// N consecutive elements <-> fp32; `aligned` = the address is a multiple of N*sizeof(TT)
// (<= 32 bytes), in which case the access is one or two wide vector instructions.
template <typename TT, int N>
__device__ __forceinline__ void load_elems(const TT* src, float (&f)[N], bool aligned = false) {
  if constexpr (sizeof(TT) == 4) {
    if (aligned) {
#pragma unroll
      for (int i = 0; i < N; i += 4) {
        const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(src + i));
        f[i] = v[0]; f[i + 1] = v[1]; f[i + 2] = v[2]; f[i + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) f[i] = src[i];
    }
  } else {
    if (aligned) {
      if constexpr (N == 8) {
        const u32x4n rr = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(src));
        const uint4 r = make_uint4(rr[0], rr[1], rr[2], rr[3]);
        Piece<uint16_t>::unpack(r, f);
      } else {
        const uint2 r = *reinterpret_cast<const uint2*>(src);
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) f[i] = bf16_to_f32(src[i]);
    }
  }
}
template <typename TT, int N>
__device__ __forceinline__ void store_elems(TT* dst, const float (&f)[N], bool aligned = false) {
  if constexpr (sizeof(TT) == 4) {
    if (aligned) {
#pragma unroll
      for (int i = 0; i < N; i += 4)
        __builtin_nontemporal_store(f32x4n{f[i], f[i + 1], f[i + 2], f[i + 3]}, reinterpret_cast<f32x4n*>(dst + i));
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) dst[i] = f[i];
    }
  } else {
    if (aligned) {
      if constexpr (N == 8)
        __builtin_nontemporal_store(u32x4n{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                                           pack_bf16x2(f[6], f[7])}, reinterpret_cast<u32x4n*>(dst));
      else
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) dst[i] = f32_to_bf16(f[i]);
    }
  }
}

// table index of a global row: tables are few and row_base ascending
__device__ __forceinline__ int find_table(const krs_table* tables, int n_tables, int64_t row) {
  int lo = 0, hi = n_tables - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tables[mid].row_base <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

constexpr int kLongUnroll = 4;   // ... and per group in the hot-row kernel, whose chunks are long
constexpr int kSegsPerGroup = 4;  // segments each group walks (amortises the descriptor prologue)
// ... per mode: the slot-less modes (SGD, dense, compact) keep fewer rows in flight per lane and run better with TWO
// (round 5, variant builds at the C3 shape: fused SGD 1297-1315 -> 1119 us multi-hot, 273 -> 260 us at L = 1; Adagrad is flat
// over 1 / 2 / 3 / 4 -- 2266 / 2301 / 2314 / 2282 us -- and 22 % slower with 8)
constexpr int segs_per_group(int mode) { return mode_slots(mode) == 0 && mode != kAdagradRow ? 2 : kSegsPerGroup; }
constexpr int kMaxLdsDesc = 512;  // features / tables whose descriptors are cached in LDS

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x4* gvec_ptr;

// ---- the per-segment kernel, written for memory-level parallelism (round 3) ---------------------------------
// GT: gradient element, TT: table element (fused modes), LPR lanes per row.  One group of LPR lanes per SEGMENT (run of
// equal keys = one table row); the plan's segment list makes the work dense.  Feature / table descriptors are cached in
// LDS per workgroup, so the dependent chain is seg_start -> vals -> (LDS) -> gradient rows; consecutive segments are
// consecutive table rows (the stream is sorted by row), so a workgroup's row updates are near-sequential in HBM.
// The round-1 kernel (`bag_apply_kernel`, deleted in round 5 after two rounds of A/B: 2336 -> 2305-2326 us Adagrad,
// 1383 -> 1312 us SGD, profiles/r3z_k1_k2_multihot.txt; bit-identical) compiled into a chain of dependent round trips: its run-time branches
// (aligned / unaligned access forms, optional weights and bag scales, descriptors in LDS or in memory) sit
// AROUND loads, and hipcc closes every such branch with `s_waitcnt vmcnt(0)` -- the ISA of the C3 instance had a
// full drain between the table-row load, each accumulator load and each gradient row: seven to eight serial
// round trips per segment, hidden only by occupancy (4.1 TB/s on the algorithmic bytes).  This kernel does the
// same arithmetic in the same order with NO branch around a load:
//   * every access is an under-aligned wide vector access (typedefs below: the HSA ABI runs the memory pipeline
//     in unaligned-access mode, a `global_load_dwordx4` needs no 16-byte alignment), so there is one access form;
//   * weights / bag scales / LDS descriptors are compile-time cases (the host picks the instance);
//   * invalid or out-of-range work is CLAMPED to a valid address and masked at the store, never skipped;
//   * the metadata of the group's segs_per_group(MODE) segments is fetched in two trips for all of them (bounds; key and
//     the first two values), and the segments are software-pipelined: the table row, accumulator row and the
//     first two gradient rows of segment i+1 are requested before segment i is consumed.
// Per group that is 2 + 1 trips for four segments instead of ~8 each.  Segments longer than two lookups finish
// in a loop of four gradient rows per trip.  Results are bit-identical to the round-1 kernel (same fmaf chain in
// ascending position, same row_update).  Descriptor counts beyond the LDS cache (> 512 features or tables) and the
// one dtype pair no Keras policy produces (fp32 gradients into bf16 tables) take bag_apply_generic.
typedef u32x4 u32x4_ua __attribute__((aligned(2)));
typedef u32x2 u32x2_ua __attribute__((aligned(2)));

template <int W>
struct RawRow {   // W dwords of a row piece, as loaded
  uint32_t r[W];
};
// (table / slot pointers come out of descriptors, i.e. as generic pointers: accessed as such they become FLAT
//  instructions, which count on the LDS counter too and make hipcc fence them against every ds_read -- the
//  descriptors live in LDS -- so every access below goes through an explicit global-address-space pointer)
#define KRS_AS1 __attribute__((address_space(1)))
template <int W, bool NT>
__device__ __forceinline__ RawRow<W> raw_load(const void* src) {
  RawRow<W> o;
  if constexpr (W == 2) {
    const KRS_AS1 u32x2_ua* q = (const KRS_AS1 u32x2_ua*)src;
    const u32x2 v = NT ? __builtin_nontemporal_load(q) : *q;
    o.r[0] = v[0]; o.r[1] = v[1];
  } else {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) {
      const KRS_AS1 u32x4_ua* q = (const KRS_AS1 u32x4_ua*)src + i;
      const u32x4 v = NT ? __builtin_nontemporal_load(q) : *q;
      o.r[4 * i] = v[0]; o.r[4 * i + 1] = v[1]; o.r[4 * i + 2] = v[2]; o.r[4 * i + 3] = v[3];
    }
  }
  return o;
}
template <int W>
__device__ __forceinline__ void raw_store_nt(void* dst, const RawRow<W>& o) {
  if constexpr (W == 2) {
    __builtin_nontemporal_store(u32x2{o.r[0], o.r[1]}, (KRS_AS1 u32x2_ua*)dst);
  } else {
#pragma unroll
    for (int i = 0; i < W / 4; ++i)
      __builtin_nontemporal_store(u32x4{o.r[4 * i], o.r[4 * i + 1], o.r[4 * i + 2], o.r[4 * i + 3]},
                                  (KRS_AS1 u32x4_ua*)dst + i);
  }
}
// N elements of type TT <-> fp32
template <typename TT, int N>
__device__ __forceinline__ void raw_to_f32(const RawRow<N * (int)sizeof(TT) / 4>& o, float (&f)[N]) {
  if constexpr (sizeof(TT) == 4) {
#pragma unroll
    for (int k = 0; k < N; ++k) f[k] = __uint_as_float(o.r[k]);
  } else {
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      f[2 * k] = __uint_as_float(o.r[k] << 16);
      f[2 * k + 1] = __uint_as_float(o.r[k] & 0xffff0000u);
    }
  }
}
template <typename TT, int N>
__device__ __forceinline__ RawRow<N * (int)sizeof(TT) / 4> f32_to_raw(const float (&f)[N]) {
  RawRow<N * (int)sizeof(TT) / 4> o;
  if constexpr (sizeof(TT) == 4) {
#pragma unroll
    for (int k = 0; k < N; ++k) o.r[k] = __float_as_uint(f[k]);
  } else {
#pragma unroll
    for (int k = 0; k < N / 2; ++k) o.r[k] = pack_bf16x2(f[2 * k], f[2 * k + 1]);
  }
  return o;
}

constexpr int kFastFirst = 2;   // gradient rows requested with the row itself
constexpr int kFastMore = 4;    // ... and per trip of the remainder loop

template <typename GT, typename TT, int LPR, int MODE, bool HAS_W, bool HAS_SCALE>
__global__ __launch_bounds__(256) void bag_apply_fast_kernel(const ApplyParams p) {
  constexpr int N = Piece<GT>::N;
  constexpr int S = segs_per_group(MODE);
  constexpr int WT = N * (int)sizeof(TT) / 4;   // dwords of a lane's table piece
  constexpr int WS = N;                         // ... of its fp32 slot piece
  constexpr bool kFused = mode_is_fused(MODE);
  constexpr int kSlots = mode_slots(MODE);
  __shared__ int s_fcol[kMaxLdsDesc];
  __shared__ int s_ftab[kMaxLdsDesc];
  __shared__ krs_table s_tab[kMaxLdsDesc];
  constexpr int GPB = 256 / LPR;
  const uint32_t n_seg = *p.n_seg;
  const Hyper hy = live_hyper<MODE>(p);
  const int64_t u_base = (int64_t)blockIdx.x * (GPB * S);
  if (u_base >= n_seg) return;
  for (int f = threadIdx.x; f < p.n_feats; f += 256) {
    s_fcol[f] = p.feats[f].out_col;
    s_ftab[f] = p.feats[f].table;
  }
  if constexpr (MODE != kSparse)
    for (int t = threadIdx.x; t < p.n_tables; t += 256) s_tab[t] = p.tables[t];
  __syncthreads();

  const int sub = threadIdx.x % LPR;
  const int row_pieces = (int)(((int64_t)p.dim * sizeof(GT)) >> 4);
  const bool col_live = sub < row_pieces;
  const int csub = col_live ? sub : 0;
  const char* grad = reinterpret_cast<const char*>(p.grad) + (int64_t)csub * 16;
  const uint32_t batch = (uint32_t)p.batch;

  // ---- trip 1: the bounds of the group's segments; trip 2: key and the first two values of each ----
  int64_t s0[S], e0[S];
  bool ok[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int64_t u = u_base + (int64_t)i * GPB + threadIdx.x / LPR;
    ok[i] = u < n_seg;
    const int64_t uc = ok[i] ? u : (int64_t)n_seg - 1;
    const bool has_next = uc + 1 < n_seg;
    s0[i] = p.seg_start[uc];
    const int64_t nx = p.seg_start[has_next ? uc + 1 : uc];
    e0[i] = has_next ? nx : p.nnz;
  }
  uint32_t key[S];
  uint64_t va[S][kFastFirst];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    key[i] = p.keys[s0[i]];
#pragma unroll
    for (int q = 0; q < kFastFirst; ++q) va[i][q] = p.vals[min(s0[i] + q, e0[i] - 1)];
  }
#pragma unroll
  for (int i = 0; i < S; ++i) {
    // the trailing run of invalid keys (out-of-range ids; the padded tail of a static-capacity exchange, whose
    // values were never written): nothing of it may become an address -- bag 0 / position 0 stand in
    if (key[i] == kInvalidKey) {
#pragma unroll
      for (int q = 0; q < kFastFirst; ++q) va[i][q] = 0;
    }
    ok[i] = ok[i] && key[i] != kInvalidKey && e0[i] - s0[i] <= kLongSeg;
  }

  // what a segment has in flight
  struct InFlight {
    krs_table tb;
    int64_t off, row;
    RawRow<WT> w;
    RawRow<WS> a, b;
    float a_row;
    u32x4 g[kFastFirst];
    float c[kFastFirst];
  };
  auto grad_src = [&](uint32_t bag) {
    const uint32_t f = bag / batch;
    const uint32_t b = bag - f * batch;
    // (timing-only builds of round 4 read sample 0 everywhere / a feature-major slab here: gathers free -10 %, layout -1 %,
    //  profiles/r4y_k2_gather_cost.txt)
    return grad + ((int64_t)b * p.grad_ld + s_fcol[f]) * (int64_t)sizeof(GT);
  };
  auto coef_of = [&](uint64_t v) {
    float c = 1.0f;
    if constexpr (HAS_W) c = p.weights[(uint32_t)v];
    if constexpr (HAS_SCALE) c *= p.bag_scale[(uint32_t)(v >> 32)];
    return c;
  };
  auto issue = [&](int i, InFlight& x) {
    x.tb = krs_table{};
    x.off = 0;
    x.row = 0;
    if constexpr (MODE != kSparse) {
      const uint32_t f0 = (uint32_t)(va[i][0] >> 32) / batch;
      x.tb = s_tab[s_ftab[f0]];
      // (a segment that is not this kernel's to finish still names a table: its row 0 stands in)
      x.row = ok[i] ? (int64_t)key[i] - x.tb.row_base : 0;
      x.off = x.row * p.dim + csub * N;
      if constexpr (kFused) x.w = raw_load<WT, true>(reinterpret_cast<const TT*>(x.tb.weights) + x.off);
      if constexpr (kSlots >= 1) x.a = raw_load<WS, true>(x.tb.slot + x.off);
      if constexpr (kSlots == 2) x.b = raw_load<WS, true>(x.tb.slot + (int64_t)x.tb.vocab * p.dim + x.off);
      if constexpr (MODE == kAdagradRow) x.a_row = *((const KRS_AS1 float*)x.tb.slot + x.row);
    }
#pragma unroll
    for (int q = 0; q < kFastFirst; ++q) {
      x.g[q] = *(const KRS_AS1 u32x4_ua*)grad_src((uint32_t)(va[i][q] >> 32));
      x.c[q] = coef_of(va[i][q]);
    }
  };
  auto consume = [&](int i, const InFlight& x) {
    const int64_t u = u_base + (int64_t)i * GPB + threadIdx.x / LPR;
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int q = 0; q < kFastFirst; ++q) {
      if (s0[i] + q < e0[i]) {
        float gv[N];
        Piece<GT>::unpack(make_uint4(x.g[q].x, x.g[q].y, x.g[q].z, x.g[q].w), gv);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = fmaf(x.c[q], gv[k], acc[k]);
      }
    }
    // the rest of a longer segment, four gradient rows per trip (positions clamped, contributions masked)
    if (ok[i]) {
      for (int64_t j0 = s0[i] + kFastFirst; j0 < e0[i]; j0 += kFastMore) {
        uint64_t vv[kFastMore];
#pragma unroll
        for (int q = 0; q < kFastMore; ++q) vv[q] = p.vals[min(j0 + q, e0[i] - 1)];
        u32x4 raw[kFastMore];
        float cf[kFastMore];
#pragma unroll
        for (int q = 0; q < kFastMore; ++q) {
          raw[q] = *(const KRS_AS1 u32x4_ua*)grad_src((uint32_t)(vv[q] >> 32));
          cf[q] = coef_of(vv[q]);
        }
#pragma unroll
        for (int q = 0; q < kFastMore; ++q) {
          if (j0 + q < e0[i]) {
            float gv[N];
            Piece<GT>::unpack(make_uint4(raw[q].x, raw[q].y, raw[q].z, raw[q].w), gv);
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = fmaf(cf[q], gv[k], acc[k]);
          }
        }
      }
    }
    if constexpr (MODE == kAdagradRow) {
      float ss = 0.0f;
      if (col_live) {
#pragma unroll
        for (int k = 0; k < N; ++k) ss = fmaf(acc[k], acc[k], ss);
      }
      ss = row_sumsq<LPR>(ss);      // every lane of the group takes part, whatever `ok` says
      const float a_new = x.a_row + ss / (float)p.dim;
      if (ok[i] && col_live) {
        float wv[N];
        raw_to_f32<TT, N>(x.w, wv);
        const float inv = a_new > 0.0f ? x.tb.lr / sqrtf(a_new) : 0.0f;  // untouched accumulator + zero gradient: leave the row
#pragma unroll
        for (int k = 0; k < N; ++k) wv[k] = wv[k] - inv * acc[k];
        raw_store_nt<WT>(reinterpret_cast<TT*>(x.tb.weights) + x.off, f32_to_raw<TT, N>(wv));
        if (sub == 0) *((KRS_AS1 float*)x.tb.slot + x.row) = a_new;
      }
      return;
    }
    if (!ok[i] || !col_live) return;
    if constexpr (MODE == kSparse) {
      if (sub == 0) p.unique_rows[u] = (int64_t)key[i];
      float* dst = p.row_grads + u * p.dim + sub * N;
#pragma unroll
      for (int k = 0; k < N; ++k) dst[k] = acc[k];
    } else if constexpr (MODE == kDense) {
      raw_store_nt<WS>(reinterpret_cast<float*>(x.tb.weights) + x.off, f32_to_raw<float, N>(acc));
    } else {
      float wv[N], av[N], bv[N];
      raw_to_f32<TT, N>(x.w, wv);
#pragma unroll
      for (int k = 0; k < N; ++k) { av[k] = 0.0f; bv[k] = 0.0f; }
      if constexpr (kSlots >= 1) raw_to_f32<float, N>(x.a, av);
      if constexpr (kSlots == 2) raw_to_f32<float, N>(x.b, bv);
#pragma unroll
      for (int k = 0; k < N; ++k) row_update<MODE>(wv[k], av[k], bv[k], acc[k], x.tb.lr, hy);
      if constexpr (kSlots >= 1) raw_store_nt<WS>(x.tb.slot + x.off, f32_to_raw<float, N>(av));
      if constexpr (kSlots == 2) raw_store_nt<WS>(x.tb.slot + (int64_t)x.tb.vocab * p.dim + x.off, f32_to_raw<float, N>(bv));
      raw_store_nt<WT>(reinterpret_cast<TT*>(x.tb.weights) + x.off, f32_to_raw<TT, N>(wv));
    }
  };

  InFlight fl[2];
  issue(0, fl[0]);
#pragma unroll
  for (int i = 0; i < S; ++i) {
    if (i + 1 < S) issue(i + 1, fl[(i + 1) & 1]);
    consume(i, fl[i & 1]);
  }
}

// Writes one finished row (summed gradient `tot` of segment u, this lane's N columns): dense
// gradient row, compact (unique_rows, grads) entry, or the fused optimizer update in place.
// Row-wise Adagrad: EVERY lane of the row's group calls (live = the lane holds columns of the row).
template <typename GT, typename TT, int MODE, int LPR = 1>
__device__ __forceinline__ void finish_row(const ApplyParams& p, uint32_t u, uint32_t key, int64_t s0, int sub,
                                           const float (&tot)[Piece<GT>::N], bool live = true) {
  constexpr int N = Piece<GT>::N;
  if constexpr (MODE == kAdagradRow) {
    const uint64_t v0 = p.vals[s0];
    const int f0 = (int)((uint32_t)(v0 >> 32) / (uint32_t)p.batch);
    const krs_table tb = p.tables[p.feats[f0].table];
    const int64_t row = (int64_t)key - tb.row_base;
    float ss = 0.0f;
    if (live) {
#pragma unroll
      for (int k = 0; k < N; ++k) ss = fmaf(tot[k], tot[k], ss);
    }
    ss = row_sumsq<LPR>(ss);
    const float a_new = tb.slot[row] + ss / (float)p.dim;
    if (live) {
      const int64_t off = row * p.dim + sub * N;
      const bool t_al = (reinterpret_cast<uintptr_t>(tb.weights) & 15) == 0;
      float wv[N];
      load_elems<TT, N>(reinterpret_cast<const TT*>(tb.weights) + off, wv, t_al);
      const float inv = a_new > 0.0f ? tb.lr / sqrtf(a_new) : 0.0f;  // untouched accumulator + zero gradient: leave the row
#pragma unroll
      for (int k = 0; k < N; ++k) wv[k] = wv[k] - inv * tot[k];
      store_elems<TT, N>(reinterpret_cast<TT*>(tb.weights) + off, wv, t_al);
      if (sub == 0) tb.slot[row] = a_new;
    }
  } else if constexpr (MODE == kSparse) {
    if (sub == 0) p.unique_rows[u] = (int64_t)key;
    float* dst = p.row_grads + (int64_t)u * p.dim + sub * N;
#pragma unroll
    for (int k = 0; k < N; ++k) dst[k] = tot[k];
  } else {
    const uint64_t v0 = p.vals[s0];
    const int f0 = (int)((uint32_t)(v0 >> 32) / (uint32_t)p.batch);
    const krs_table tb = p.tables[p.feats[f0].table];
    const int64_t off = ((int64_t)key - tb.row_base) * p.dim + sub * N;
    const bool t_al = ((reinterpret_cast<uintptr_t>(tb.weights) | reinterpret_cast<uintptr_t>(tb.slot)) & 15) == 0;
    if constexpr (MODE == kDense) {
      store_elems<float, N>(reinterpret_cast<float*>(tb.weights) + off, tot, t_al);
    } else {
      float wv[N], av[N], bv[N];
#pragma unroll
      for (int k = 0; k < N; ++k) { av[k] = 0.0f; bv[k] = 0.0f; }
      const int64_t plane = tb.vocab * p.dim;
      load_elems<TT, N>(reinterpret_cast<const TT*>(tb.weights) + off, wv, t_al);
      if constexpr (mode_slots(MODE) >= 1) load_elems<float, N>(tb.slot + off, av, t_al);
      if constexpr (mode_slots(MODE) == 2) load_elems<float, N>(tb.slot + plane + off, bv, t_al && plane % 4 == 0);
      const Hyper hy = live_hyper<MODE>(p);
#pragma unroll
      for (int k = 0; k < N; ++k) row_update<MODE>(wv[k], av[k], bv[k], tot[k], tb.lr, hy);
      if constexpr (mode_slots(MODE) >= 1) store_elems<float, N>(tb.slot + off, av, t_al);
      if constexpr (mode_slots(MODE) == 2) store_elems<float, N>(tb.slot + plane + off, bv, t_al && plane % 4 == 0);
      store_elems<TT, N>(reinterpret_cast<TT*>(tb.weights) + off, wv, t_al);
    }
  }
}

// Hot rows (segments longer than kLongSeg, e.g. power-law ids or tiny vocabularies): one
// workgroup per segment.  Its 256/LPR groups sum interleaved positions of the segment (four
// gradient rows in flight each), the partial rows meet in LDS and are added in a fixed order
// (group 0, 1, 2, ...), so the result stays run-to-run bit-identical.
template <typename GT, typename TT, int LPR, int MODE, bool HAS_W>
__global__ __launch_bounds__(256) void bag_apply_long_kernel(const ApplyParams p) {
  constexpr int N = Piece<GT>::N;
  constexpr int GPB = 256 / LPR;
  extern __shared__ __attribute__((aligned(16))) char smem_long[];
  float* part = reinterpret_cast<float*>(smem_long);  // [GPB][dim]
  const uint32_t n_long = *p.n_long;
  const uint32_t n_seg = *p.n_seg;
  const int g = threadIdx.x / LPR;
  const int sub = threadIdx.x % LPR;
  const int row_pieces = (int)(((int64_t)p.dim * sizeof(GT)) >> 4);
  const bool col_live = sub < row_pieces;
  const int csub = col_live ? sub : 0;
  const char* grad = reinterpret_cast<const char*>(p.grad) + (int64_t)csub * 16;
  const bool g_aligned = ((reinterpret_cast<uintptr_t>(p.grad) | (uintptr_t)(p.grad_ld * sizeof(GT))) & 15) == 0;
  for (uint32_t li = blockIdx.x; li < n_long; li += gridDim.x) {
    const LongItem item = p.long_list[li];
    const uint32_t u = item.seg;
    const int64_t s0 = p.seg_start[u];
    const int64_t seg_end = u + 1 < n_seg ? (int64_t)p.seg_start[u + 1] : p.nnz;
    const uint32_t key = p.keys[s0];
    if (key == kInvalidKey) continue;
    // this workgroup's piece of the segment
    const int64_t c0 = s0 + (int64_t)item.chunk * kChunk;
    const int64_t e0 = min(seg_end, c0 + kChunk);
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.0f;
    for (int64_t j0 = c0 + g; j0 < e0; j0 += (int64_t)GPB * kLongUnroll) {
      uint64_t vv[kLongUnroll];
#pragma unroll
      for (int q = 0; q < kLongUnroll; ++q) vv[q] = p.vals[min(j0 + (int64_t)q * GPB, e0 - 1)];
      float coef[kLongUnroll];
      u32x4 raw[kLongUnroll];
#pragma unroll
      for (int q = 0; q < kLongUnroll; ++q) {
        const uint32_t bag = (uint32_t)(vv[q] >> 32);
        const uint32_t pos = (uint32_t)vv[q];
        const int f = (int)(bag / (uint32_t)p.batch);
        const int b = (int)(bag - (uint32_t)f * (uint32_t)p.batch);
        float c = 1.0f;
        if constexpr (HAS_W) c = p.weights[pos];
        if (p.bag_scale) c *= p.bag_scale[bag];
        coef[q] = c;
        const char* src = grad + ((int64_t)b * p.grad_ld + p.feats[f].out_col) * (int64_t)sizeof(GT);
        if (g_aligned) {
          raw[q] = *(gvec_ptr)src;
        } else {
          const GT* e = reinterpret_cast<const GT*>(src);
          if constexpr (sizeof(GT) == 4) {
            raw[q] = u32x4{__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2]), __float_as_uint(e[3])};
          } else {
            raw[q] = u32x4{e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16),
                           e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)};
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kLongUnroll; ++q) {
        if (j0 + (int64_t)q * GPB < e0) {
          float gv[N];
          Piece<GT>::unpack(make_uint4(raw[q].x, raw[q].y, raw[q].z, raw[q].w), gv);
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = fmaf(coef[q], gv[k], acc[k]);
        }
      }
    }
    if (col_live) {
#pragma unroll
      for (int k = 0; k < N; ++k) part[g * p.dim + sub * N + k] = acc[k];
    }
    __syncthreads();
    if (g == 0 && (col_live || MODE == kAdagradRow)) {
      float tot[N];
#pragma unroll
      for (int k = 0; k < N; ++k) tot[k] = 0.0f;
      if (col_live)
        for (int gg = 0; gg < GPB; ++gg)
#pragma unroll
          for (int k = 0; k < N; ++k) tot[k] += part[gg * p.dim + sub * N + k];
      if (item.partial != 0xffffffffu) {  // one of several chunks: the row is finished by bag_apply_finish_kernel
        if (col_live) {
          float* dst = p.partials + (int64_t)item.partial * (kPartialBytes / 4) + sub * N;
#pragma unroll
          for (int k = 0; k < N; ++k) dst[k] = tot[k];
        }
      } else {
        finish_row<GT, TT, MODE, LPR>(p, u, key, s0, sub, tot, col_live);
      }
    }
    __syncthreads();
  }
}

// Segments longer than kChunk: their chunks' partial rows are added in chunk order (fixed, so the
// result stays run-to-run bit-identical) by one group of LPR lanes each, which then writes the row.
template <typename GT, typename TT, int LPR, int MODE>
__global__ __launch_bounds__(256) void bag_apply_finish_kernel(const ApplyParams p) {
  constexpr int N = Piece<GT>::N;
  constexpr int GPB = 256 / LPR;
  const uint32_t n_multi = p.n_long[2];
  const int sub = threadIdx.x % LPR;
  const int row_pieces = (int)(((int64_t)p.dim * sizeof(GT)) >> 4);
  const bool live = sub < row_pieces;
  if (!live && MODE != kAdagradRow) return;
  for (uint32_t mi = blockIdx.x * GPB + threadIdx.x / LPR; mi < n_multi; mi += gridDim.x * GPB) {
    const MultiSeg ms = p.multi_list[mi];
    const int64_t s0 = p.seg_start[ms.seg];
    const uint32_t key = p.keys[s0];
    // a trailing run of invalid keys longer than kChunk (out-of-range ids, the padded tail of a static-capacity
    // exchange) is listed here too: bag_apply_long_kernel wrote no partial rows for it and it names no table row
    if (key == kInvalidKey) continue;
    float tot[N];
#pragma unroll
    for (int k = 0; k < N; ++k) tot[k] = 0.0f;
    if (live)
      for (uint32_t c = 0; c < ms.n_chunks; ++c) {
        const float* src = p.partials + (int64_t)(ms.partial_base + c) * (kPartialBytes / 4) + sub * N;
#pragma unroll
        for (int k = 0; k < N; ++k) tot[k] += src[k];
      }
    finish_row<GT, TT, MODE, LPR>(p, ms.seg, key, s0, sub, tot, live);
  }
}

// Any dim / dtype: LPR lanes per segment, one column per lane per pass.
template <int MODE>
__global__ __launch_bounds__(256) void bag_apply_generic(const ApplyParams p, int grad_dtype, int table_dtype,
                                                         int lpr) {
  const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpr;
  const int sub = threadIdx.x % lpr;
  const uint32_t n_seg = *p.n_seg;
  if (u >= n_seg) return;
  const int64_t s0 = p.seg_start[u];
  const int64_t e0 = u + 1 < n_seg ? (int64_t)p.seg_start[u + 1] : p.nnz;
  const uint32_t key = p.keys[s0];
  if (key == kInvalidKey) return;
  const int t = MODE == kSparse ? 0 : find_table(p.tables, p.n_tables, (int64_t)key);
  auto column_grad = [&](int c) {
    float acc = 0.0f;
    for (int64_t j = s0; j < e0; ++j) {
      const uint64_t v = p.vals[j];
      const uint32_t bag = (uint32_t)(v >> 32);
      const int f = (int)(bag / (uint32_t)p.batch);
      const int b = (int)(bag - (uint32_t)f * (uint32_t)p.batch);
      float coef = p.weights ? p.weights[(uint32_t)v] : 1.0f;
      if (p.bag_scale) coef *= p.bag_scale[bag];
      acc = fmaf(coef, ld_elem(p.grad, grad_dtype, (int64_t)b * p.grad_ld + p.feats[f].out_col + c), acc);
    }
    return acc;
  };
  if constexpr (MODE == kAdagradRow) {
    // two passes over the row's columns: sum of squares (group-wide), then the update
    const krs_table tb = p.tables[t];
    const int64_t row = (int64_t)key - tb.row_base;
    float ss = 0.0f;
    for (int c = sub; c < p.dim; c += lpr) {
      const float gc = column_grad(c);
      ss = fmaf(gc, gc, ss);
    }
    for (int o = lpr / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float a_new = tb.slot[row] + ss / (float)p.dim;
    const float inv = a_new > 0.0f ? tb.lr / sqrtf(a_new) : 0.0f;  // untouched accumulator + zero gradient: leave the row
    for (int c = sub; c < p.dim; c += lpr) {
      const int64_t off = row * p.dim + c;
      st_elem(tb.weights, table_dtype, off, ld_elem(tb.weights, table_dtype, off) - inv * column_grad(c));
    }
    // every lane of the group has read the old accumulator before lane 0 overwrites it (lanes of a wave run in
    // lock step through the shuffle above)
    if (sub == 0) tb.slot[row] = a_new;
    return;
  }
  for (int c = sub; c < p.dim; c += lpr) {
    const float acc = column_grad(c);
    if (MODE == kSparse) {
      if (c == 0) p.unique_rows[u] = (int64_t)key;
      p.row_grads[(int64_t)u * p.dim + c] = acc;
    } else {
      const krs_table tb = p.tables[t];
      const int64_t off = ((int64_t)key - tb.row_base) * p.dim + c;
      if (MODE == kDense) {
        reinterpret_cast<float*>(tb.weights)[off] = acc;
      } else {
        float w = ld_elem(tb.weights, table_dtype, off);
        const int64_t plane = tb.vocab * p.dim;
        float s0 = mode_slots(MODE) >= 1 ? tb.slot[off] : 0.0f;
        float s1 = mode_slots(MODE) == 2 ? tb.slot[plane + off] : 0.0f;
        row_update<MODE>(w, s0, s1, acc, tb.lr, live_hyper<MODE>(p));
        if (mode_slots(MODE) >= 1) tb.slot[off] = s0;
        if (mode_slots(MODE) == 2) tb.slot[plane + off] = s1;
        st_elem(tb.weights, table_dtype, off, w);
      }
    }
  }
}

// ---- plan: segment list ---------------------------------------------------------
// (Measured and not kept: sorting each feature's lookups by row id alone when every feature has its own table -- the
//  lookups arrive grouped by feature, so two 10-bit passes replace the three 9/8/8-bit ones at 26 x 1 M rows: 602 us
//  either way; the scatter pass pays per key BIT (ballots), 20 against 25, and the per-feature tile bookkeeping and the
//  wider count matrix took the difference back.)
// Segment list (first sorted position of every run of equal keys) by block-wise compaction: heads are counted
// per block of 4096 keys, the block counts are scanned by one workgroup, and a second pass over the keys writes
// every head's position at (block offset + rank inside the block).  Two reads of the sorted keys and one
// compact write -- the flag / index arrays of a flags -> device-wide scan -> scatter pipeline (five launches,
// 0.5 GB of traffic at 14 M lookups) are not materialised.  A trailing run of invalid keys is a segment too.
constexpr int kSegTile = 4096, kSegItems = 16;   // 256 threads x 16 consecutive keys
__device__ __forceinline__ uint32_t seg_head_mask(const uint32_t* keys, int64_t nnz, int64_t t0) {
  if (t0 >= nnz) return 0u;
  uint32_t prev = t0 > 0 ? keys[t0 - 1] : ~keys[0];   // position 0 is a head
  uint32_t mask = 0u;
  if (t0 + kSegItems <= nnz) {
    const uint4* v = reinterpret_cast<const uint4*>(keys + t0);   // t0 is a multiple of 16: 64-byte aligned
#pragma unroll
    for (int q = 0; q < kSegItems / 4; ++q) {
      const uint4 k = v[q];
      mask |= (uint32_t)(k.x != prev) << (4 * q);
      mask |= (uint32_t)(k.y != k.x) << (4 * q + 1);
      mask |= (uint32_t)(k.z != k.y) << (4 * q + 2);
      mask |= (uint32_t)(k.w != k.z) << (4 * q + 3);
      prev = k.w;
    }
  } else {
    for (int k = 0; t0 + k < nnz; ++k) {
      const uint32_t cur = keys[t0 + k];
      mask |= (uint32_t)(cur != prev) << k;
      prev = cur;
    }
  }
  return mask;
}
__global__ __launch_bounds__(256) void seg_count_kernel(const uint32_t* keys, int64_t nnz, int32_t* block_heads) {
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  int c = __popc(seg_head_mask(keys, nnz, (int64_t)blockIdx.x * kSegTile + threadIdx.x * kSegItems));
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);   // integer: order does not matter
  __syncthreads();
  if (threadIdx.x == 0) block_heads[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void seg_emit_kernel(const uint32_t* keys, int64_t nnz, const int32_t* block_off,
                                                       uint32_t* seg_start, uint32_t* n_seg) {
  __shared__ int wsum[4];
  const int64_t t0 = (int64_t)blockIdx.x * kSegTile + threadIdx.x * kSegItems;
  const uint32_t mask = seg_head_mask(keys, nnz, t0);
  const int c = __popc(mask), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint32_t off = (uint32_t)block_off[blockIdx.x] + (uint32_t)(x - c);
  for (int w = 0; w < wave; ++w) off += (uint32_t)wsum[w];
  uint32_t m = mask;
  while (m) {
    const int k = __ffs(m) - 1;
    m &= m - 1;
    seg_start[off++] = (uint32_t)(t0 + k);
  }
  if (t0 < nnz && t0 + kSegItems >= nnz) *n_seg = off;   // the thread that holds the last key
}
__global__ void long_list_kernel(const uint32_t* seg_start, const uint32_t* n_seg, int64_t nnz, uint32_t* counters,
                                 LongItem* items, MultiSeg* multi) {
  // counters[0] = work items, [1] = partial rows handed out, [2] = multi-chunk segments
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ns = *n_seg;
  if (u >= ns) return;
  const int64_t e = u + 1 < ns ? (int64_t)seg_start[u + 1] : nnz;
  const int64_t len = e - (int64_t)seg_start[u];
  if (len <= kLongSeg) return;
  const uint32_t nch = (uint32_t)((len + kChunk - 1) / kChunk);
  const uint32_t base = atomicAdd(counters, nch);
  uint32_t pb = 0xffffffffu;
  if (nch > 1) {
    pb = atomicAdd(counters + 1, nch);
    multi[atomicAdd(counters + 2, 1u)] = MultiSeg{(uint32_t)u, pb, nch};
  }
  for (uint32_t c = 0; c < nch; ++c) items[base + c] = LongItem{(uint32_t)u, c, nch > 1 ? pb + c : 0xffffffffu};
}
__global__ void count_unique_kernel(const uint32_t* keys, const uint32_t* n_seg, const uint32_t* sort_mode, int64_t nnz,
                                    int64_t* n_unique) {
  // segments minus the trailing run of invalid keys, if any.  A table-segmented plan leaves the out-of-range lookups at
  // the end of every TABLE's run: the compact form cannot be built from it (-1, which
  // the caller turns into an error: embedding_ops.backward_sparse)
  if (*sort_mode != 0) { *n_unique = -1; return; }
  *n_unique = (int64_t)*n_seg - (nnz > 0 && keys[nnz - 1] == kInvalidKey ? 1 : 0);
}

template <typename GT, typename TT, int MODE>
int launch_apply_lpr(const ApplyParams& p, int pieces, hipStream_t st) {
  const int lpr = pieces <= 8 ? 8 : (pieces <= 16 ? 16 : (pieces <= 32 ? 32 : 64));
  const int64_t groups = p.nnz;  // upper bound of the segment count (device-side n_seg trims it)
  const int64_t blocks = ceil_div(groups, (256 / lpr) * segs_per_group(MODE));
  if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_bwd: grid too large");
#define KRS_LAUNCH_FAST(L, W, SC) \
  hipLaunchKernelGGL((bag_apply_fast_kernel<GT, TT, L, MODE, W, SC>), dim3((unsigned)blocks), dim3(256), 0, st, p)
#define KRS_LAUNCH_APPLY(L)                                                                             \
  if (p.weights) { if (p.bag_scale) KRS_LAUNCH_FAST(L, true, true); else KRS_LAUNCH_FAST(L, true, false); }   \
  else { if (p.bag_scale) KRS_LAUNCH_FAST(L, false, true); else KRS_LAUNCH_FAST(L, false, false); }
  if (lpr == 8) { KRS_LAUNCH_APPLY(8) }
  else if (lpr == 16) { KRS_LAUNCH_APPLY(16) }
  else if (lpr == 32) { KRS_LAUNCH_APPLY(32) }
  else { KRS_LAUNCH_APPLY(64) }
#undef KRS_LAUNCH_APPLY
#undef KRS_LAUNCH_FAST
  KRS_CHECK_LAUNCH("bag_apply_fast_kernel");
  // hot rows: upper bound of the item count is nnz / kLongSeg + nnz / kChunk; surplus workgroups leave at once
  const int64_t max_long = p.nnz / kLongSeg;
  if (max_long > 0) {
    const unsigned lb = (unsigned)(max_long < 8192 ? max_long : 8192);
    const size_t lds = (size_t)(256 / lpr) * p.dim * sizeof(float);
#define KRS_LAUNCH_LONG(L)                                                                              \
  if (p.weights)                                                                                        \
    hipLaunchKernelGGL((bag_apply_long_kernel<GT, TT, L, MODE, true>), dim3(lb), dim3(256), lds, st, p); \
  else                                                                                                  \
    hipLaunchKernelGGL((bag_apply_long_kernel<GT, TT, L, MODE, false>), dim3(lb), dim3(256), lds, st, p);
    if (lpr == 8) { KRS_LAUNCH_LONG(8) }
    else if (lpr == 16) { KRS_LAUNCH_LONG(16) }
    else if (lpr == 32) { KRS_LAUNCH_LONG(32) }
    else { KRS_LAUNCH_LONG(64) }
#undef KRS_LAUNCH_LONG
    KRS_CHECK_LAUNCH("bag_apply_long_kernel");
    const int64_t max_multi = p.nnz / kChunk;
    if (max_multi > 0) {
      const unsigned fb = (unsigned)std::min<int64_t>(ceil_div(max_multi, 256 / lpr), 1024);
      if (lpr == 8) hipLaunchKernelGGL((bag_apply_finish_kernel<GT, TT, 8, MODE>), dim3(fb), dim3(256), 0, st, p);
      else if (lpr == 16) hipLaunchKernelGGL((bag_apply_finish_kernel<GT, TT, 16, MODE>), dim3(fb), dim3(256), 0, st, p);
      else if (lpr == 32) hipLaunchKernelGGL((bag_apply_finish_kernel<GT, TT, 32, MODE>), dim3(fb), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((bag_apply_finish_kernel<GT, TT, 64, MODE>), dim3(fb), dim3(256), 0, st, p);
      KRS_CHECK_LAUNCH("bag_apply_finish_kernel");
    }
  }
  return KRS_OK;
}

template <int MODE>
int run_apply(ApplyParams p, int grad_dtype, int table_dtype, hipStream_t st) {
  if (p.nnz == 0) return KRS_OK;
  const int64_t gbytes = (int64_t)p.dim * (grad_dtype == KRS_BF16 ? 2 : 4);
  // the per-lane piece must also map to whole table / accumulator elements: N elements each
  // (vector kernels: descriptors cached in LDS -- up to kMaxLdsDesc features / tables -- and the dtype pairs the Keras
  //  policies produce: float32, bfloat16, mixed_bfloat16 = bf16 gradients into fp32 tables; fp32 gradients into bf16
  //  tables and wider descriptor lists take the any-shape kernel below)
  const bool vec_pair = !(grad_dtype == KRS_F32 && table_dtype == KRS_BF16 && mode_is_fused(MODE));
  if (gbytes % 16 == 0 && gbytes <= 1024 && vec_pair && p.n_feats <= kMaxLdsDesc && p.n_tables <= kMaxLdsDesc) {
    const int pieces = (int)(gbytes / 16);
    if (grad_dtype == KRS_F32) return launch_apply_lpr<float, float, MODE>(p, pieces, st);
    return table_dtype == KRS_F32 ? launch_apply_lpr<uint16_t, float, MODE>(p, pieces, st)
                                  : launch_apply_lpr<uint16_t, uint16_t, MODE>(p, pieces, st);
  }
  int lpr = 1;
  while (lpr < p.dim && lpr < 64) lpr <<= 1;
  const int64_t blocks = ceil_div(p.nnz * lpr, 256);
  if (blocks > 0x7fffffffLL) return fail(KRS_ERR_UNSUPPORTED, "embed_bag_bwd: grid too large");
  hipLaunchKernelGGL(bag_apply_generic<MODE>, dim3((unsigned)blocks), dim3(256), 0, st, p, grad_dtype,
                     table_dtype, lpr);
  KRS_CHECK_LAUNCH("bag_apply_generic");
  return KRS_OK;
}

int check_apply_args(const void* tables_or_null, int need_tables, const krs_feature* feats, const void* grad,
                     int grad_dtype, int batch, int dim, int64_t nnz, const void* workspace) {
  KRS_REQUIRE(!need_tables || tables_or_null, "embed_bag_bwd: null tables");
  KRS_REQUIRE(feats && grad && (workspace || nnz == 0), "embed_bag_bwd: null feats/grad/workspace");
  KRS_REQUIRE(grad_dtype == KRS_F32 || grad_dtype == KRS_BF16, "embed_bag_bwd: bad grad dtype");
  KRS_REQUIRE(batch > 0 && dim > 0 && nnz >= 0, "embed_bag_bwd: bad sizes");
  return KRS_OK;
}

ApplyParams make_apply(const krs_table* tables, int n_tables, const krs_feature* feats, int n_feats,
                       const float* weights,
                       const float* bag_scale, const void* grad, int64_t grad_ld, int batch, int dim, int64_t nnz,
                       const void* workspace) {
  ApplyParams p;
  const PlanLayout l = plan_layout(const_cast<void*>(workspace), nnz);
  p.tables = tables; p.n_tables = n_tables; p.feats = feats; p.n_feats = n_feats; p.weights = weights;
  p.bag_scale = bag_scale;
  p.grad = grad; p.grad_ld = grad_ld; p.batch = batch; p.dim = dim; p.nnz = nnz;
  p.keys = l.keys_sorted; p.vals = l.vals_sorted; p.seg_start = l.seg_start; p.n_seg = l.n_seg;
  p.n_long = l.n_long; p.long_list = l.long_list; p.multi_list = l.multi_list; p.partials = l.partials;
  p.unique_rows = nullptr; p.row_grads = nullptr;
  p.hyper = Hyper{0.0f, 0.0f, 0.0f, 0.0f};
  p.hyper_d_dev = nullptr;
  return p;
}

}  // namespace
}  // namespace krs

using namespace krs;

#if KRS_BWD_HAS(0)
extern "C" size_t krs_embed_bag_bwd_workspace_bytes(int64_t nnz) {
  if (nnz < 0) return 0;
  return plan_layout(nullptr, nnz, true).total_bytes;
}
#endif

// Which sort produced the plan is recorded IN the workspace (PlanLayout::sort_mode, written by the plan call on its
// stream): krs_embed_bag_bwd_sparse reports n_unique = -1 for a table-segmented plan wherever the workspace has been
// copied to.  (Round 3 also kept a host map keyed by the workspace address; an allocator hands freed addresses out
// again, so a stale entry could refuse a good plan -- removed)

// Table-segmented sort (rs::scatter_seg_kernel) when the host descriptors are given and the lookups are laid out for
// it: dense bags, the features of a table neighbours, tables (and their row bases) ascending with the features.
// Returns true when it has enqueued the sort; false = the caller runs the global sort.
static bool plan_sort_by_table(const PlanLayout& l, const krs_table* tables, const krs_table* tables_host, int n_tables,
                               const krs_feature* feats, const krs_feature* feats_host, int n_feats, const void* ids,
                               int id_type, int batch, int64_t nnz, int* err_flag, hipStream_t st) {
  if (!tables_host || !feats_host || n_feats > rs::kGenFeats || n_tables <= 0) return false;
  rs::SegPass sp;
  rs::Seg& sg = sp.seg;
  sg.n = 0;
  int64_t pos = 0, max_vocab = 0;
  int prev_table = -1;
  for (int f = 0; f < n_feats; ++f) {
    const krs_feature& ft = feats_host[f];
    if (ft.hot < 1 || ft.ids_base != pos || ft.table < prev_table || ft.table >= n_tables) return false;
    if (ft.table != prev_table) {
      if (sg.n == rs::kMaxProb) return false;
      const krs_table& tb = tables_host[ft.table];
      if (prev_table >= 0 && tb.row_base < tables_host[prev_table].row_base + tables_host[prev_table].vocab) return false;
      sg.lookup_start[sg.n] = (uint32_t)pos;
      sg.row_base[sg.n] = (uint32_t)tb.row_base;
      max_vocab = std::max<int64_t>(max_vocab, tb.vocab);
      ++sg.n;
      prev_table = ft.table;
    }
    pos += (int64_t)batch * ft.hot;
  }
  if (pos != nnz || sg.n == 0) return false;
  sg.lookup_start[sg.n] = (uint32_t)nnz;
  uint32_t tiles = 0;
  for (int i = 0; i < sg.n; ++i) {
    sg.tile_start[i] = tiles;
    tiles += (uint32_t)ceil_div((int64_t)sg.lookup_start[i + 1] - sg.lookup_start[i], rs::kTile);
  }
  sg.tile_start[sg.n] = tiles;
  if (tiles == 0) return false;
  // key = id, plus one pattern (all ones) for an out-of-range id
  unsigned bits = 1;
  while (bits < 32 && (1ULL << bits) <= (uint64_t)max_vocab) ++bits;
  if (bits > 31) return false;
  const int passes = rs::n_passes(bits);
  char* tp = reinterpret_cast<char*>(l.temp);
  int32_t* counts = reinterpret_cast<int32_t*>(tp);
  tp += align_up((size_t)rs::kMaxBins * (ceil_div(nnz, rs::kTile) + rs::kMaxProb) * sizeof(int32_t), 256);
  int32_t* sums = reinterpret_cast<int32_t*>(tp);
  sp.counts = counts; sp.nnz = nnz; sp.key_bits = (int)bits;
  sp.gen.tables = tables; sp.gen.feats = feats; sp.gen.n_feats = n_feats; sp.gen.ids = ids;
  sp.gen.id64 = id_type == KRS_I64; sp.gen.batch = batch; sp.gen.err_flag = err_flag;
  // intermediate (key, position) pairs ping-pong between I0 = (keys_in, vals_in as u32) and I1 = (keys_sorted,
  // vals_sorted as u32); the LAST pass reads I0 and writes the final (keys_sorted, vals_sorted): the pass before it
  // writes I0, the one before that I1, ...
  uint32_t* k0 = l.keys_in; uint32_t* q0 = reinterpret_cast<uint32_t*>(l.vals_in);
  uint32_t* k1 = l.keys_sorted; uint32_t* q1 = reinterpret_cast<uint32_t*>(l.vals_sorted);
  unsigned done = 0;
  for (int ps = 0; ps < passes; ++ps) {
    const bool first = ps == 0, last = ps == passes - 1;
    sp.bits = (int)((bits - done + (passes - ps) - 1) / (passes - ps));
    sp.shift = (int)done;
    const bool out_is_i0 = ((passes - 2 - ps) % 2) == 0;        // (meaningless for the last pass)
    const bool in_is_i0 = ((passes - 1 - ps) % 2) == 0;         // the last pass reads I0
    sp.keys_in = first ? nullptr : (in_is_i0 ? k0 : k1);
    sp.pos_in = first ? nullptr : (in_is_i0 ? q0 : q1);
    sp.keys_out = last ? l.keys_sorted : (out_is_i0 ? k0 : k1);
    sp.pos_out = last ? nullptr : (out_is_i0 ? q0 : q1);
    sp.vals_out = last ? l.vals_sorted : nullptr;
    if (first) hipLaunchKernelGGL(rs::hist_seg_kernel<true>, dim3(tiles), dim3(rs::kHistThreads), 0, st, sp);
    else hipLaunchKernelGGL(rs::hist_seg_kernel<false>, dim3(tiles), dim3(rs::kHistThreads), 0, st, sp);
    scan::exclusive(counts, counts, (int64_t)(1 << sp.bits) * tiles, sums, nullptr, st);
    if (first && last) hipLaunchKernelGGL((rs::scatter_seg_kernel<true, true>), dim3(tiles), dim3(rs::kThreads), 0, st, sp);
    else if (first) hipLaunchKernelGGL((rs::scatter_seg_kernel<true, false>), dim3(tiles), dim3(rs::kThreads), 0, st, sp);
    else if (last) hipLaunchKernelGGL((rs::scatter_seg_kernel<false, true>), dim3(tiles), dim3(rs::kThreads), 0, st, sp);
    else hipLaunchKernelGGL((rs::scatter_seg_kernel<false, false>), dim3(tiles), dim3(rs::kThreads), 0, st, sp);
    done += (unsigned)sp.bits;
  }
  return true;
}

static int plan_impl(const krs_table* tables, const krs_table* tables_host, int n_tables, const krs_feature* feats,
                     const krs_feature* feats_host, int n_feats, const void* ids, int id_type, const void* offsets,
                     int off_type, int batch, int64_t nnz, int64_t total_rows, void* workspace,
                     size_t workspace_bytes, int* err_flag, void* stream) {
  KRS_REQUIRE(tables && feats && (ids || nnz == 0), "embed_bag_bwd_plan: null argument");
  KRS_REQUIRE(n_feats > 0 && batch > 0 && nnz >= 0, "embed_bag_bwd_plan: bad sizes");
  KRS_REQUIRE(total_rows > 0 && total_rows < 0xffffffffLL, "embed_bag_bwd_plan: total_rows must fit 32-bit keys");
  KRS_REQUIRE(nnz < 0x7fffffffLL && (int64_t)n_feats * batch < 0xffffffffLL,
              "embed_bag_bwd_plan: nnz must stay below 2^31 and the bag count below 2^32");
  if (nnz == 0) return KRS_OK;
  KRS_REQUIRE(workspace, "embed_bag_bwd_plan: null workspace");
  const PlanLayout l = plan_layout(workspace, nnz, true);
  if (workspace_bytes < l.total_bytes)
    return fail(KRS_ERR_WORKSPACE, "embed_bag_bwd_plan: workspace %zu < %zu bytes", workspace_bytes, l.total_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int n_tiles = (int)ceil_div(nnz, rs::kTile);
  char* tp = reinterpret_cast<char*>(l.temp);
  int32_t* counts = reinterpret_cast<int32_t*>(tp);
  tp += align_up((size_t)rs::kMaxBins * (n_tiles + rs::kMaxProb) * sizeof(int32_t), 256);
  int32_t* sums = reinterpret_cast<int32_t*>(tp);
  tp += align_up(scan::workspace_bytes((int64_t)rs::kMaxBins * (n_tiles + rs::kMaxProb)), 256);
  int32_t* sums2 = reinterpret_cast<int32_t*>(tp);
  const bool by_table = offsets == nullptr && g_plan_variant == 0 &&
                        plan_sort_by_table(l, tables, tables_host, n_tables, feats, feats_host, n_feats, ids, id_type,
                                           batch, nnz, err_flag, st);
  if (!by_table) {
  KeyParams kp;
  kp.tables = tables; kp.feats = feats; kp.n_feats = n_feats; kp.ids = ids; kp.id64 = id_type == KRS_I64;
  kp.offsets = offsets; kp.off64 = off_type == KRS_I64; kp.batch = batch; kp.keys = l.keys_in; kp.vals = l.vals_in;
  kp.err_flag = err_flag;
  // Sort only the significant key bits.  2^bits - 1 > every valid row id, so the invalid key (all ones) still
  // sorts last.
  unsigned bits = 1;
  while (bits < 32 && (1ULL << bits) <= (uint64_t)total_rows) ++bits;
  const int passes = rs::n_passes(bits);
  // ping-pong between (keys_in, vals_in) and (keys_sorted, vals_sorted); the LAST pass must write the sorted pair
  // dense bags: keys are generated inside the first pass (feature constants cached in LDS)
  const bool gen = offsets == nullptr && n_feats <= rs::kGenFeats;
  bool to_sorted = (passes % 2) == 1;    // where the first pass writes
  if (!gen) {
    // CSR bags (or very many features): the key kernel walks the bags
    kp.keys = to_sorted ? l.keys_in : l.keys_sorted;
    kp.vals = to_sorted ? l.vals_in : l.vals_sorted;
    const int64_t n_bags = (int64_t)n_feats * batch;
    // lookup positions outside every bag (offsets[n_bags] < nnz: the padded tail of a static-capacity exchange)
    // must not carry stale keys: all ones = the invalid key, which sorts behind every row and is skipped
    KRS_HIP(hipMemsetAsync(kp.keys, 0xff, (size_t)nnz * sizeof(uint32_t), st));
    hipLaunchKernelGGL(bag_keys_kernel, dim3((unsigned)ceil_div(n_bags, 16)), dim3(256), 0, st, kp);
    KRS_CHECK_LAUNCH("bag_keys_kernel");
  }
  unsigned done = 0;
  for (int ps = 0; ps < passes; ++ps) {
    rs::Pass p;
    p.bits = (int)((bits - done + (passes - ps) - 1) / (passes - ps));
    p.shift = (int)done;
    p.nnz = nnz;
    p.n_tiles = n_tiles;
    p.counts = counts;
    const bool generating = gen && ps == 0;
    p.keys_in = generating ? nullptr : (to_sorted ? l.keys_in : l.keys_sorted);
    p.vals_in = generating ? nullptr : (to_sorted ? l.vals_in : l.vals_sorted);
    p.keys_out = to_sorted ? l.keys_sorted : l.keys_in;
    p.vals_out = to_sorted ? l.vals_sorted : l.vals_in;
    p.gen.tables = tables; p.gen.feats = feats; p.gen.n_feats = n_feats; p.gen.ids = ids;
    p.gen.id64 = id_type == KRS_I64; p.gen.batch = batch; p.gen.err_flag = err_flag;
    if (generating) hipLaunchKernelGGL(rs::hist_kernel<true>, dim3(n_tiles), dim3(rs::kHistThreads), 0, st, p);
    else hipLaunchKernelGGL(rs::hist_kernel<false>, dim3(n_tiles), dim3(rs::kHistThreads), 0, st, p);
    scan::exclusive(counts, counts, (int64_t)(1 << p.bits) * n_tiles, sums, nullptr, st);
    if (generating) hipLaunchKernelGGL(rs::scatter_kernel<true>, dim3(n_tiles), dim3(rs::kThreads), 0, st, p);
    else hipLaunchKernelGGL(rs::scatter_kernel<false>, dim3(n_tiles), dim3(rs::kThreads), 0, st, p);
    done += (unsigned)p.bits;
    to_sorted = !to_sorted;
  }
  }
  KRS_CHECK_LAUNCH("embed_bag_bwd_plan: radix sort");
  // segment list: heads per block -> one-workgroup scan of the block counts -> head positions
  const unsigned nb = (unsigned)ceil_div(nnz, 256);
  const unsigned nsb = (unsigned)ceil_div(nnz, kSegTile);
  hipLaunchKernelGGL(seg_count_kernel, dim3(nsb), dim3(256), 0, st, l.keys_sorted, nnz, sums2);
  hipLaunchKernelGGL(scan::block_kernel, dim3(1), dim3(1024), 0, st, sums2, (int64_t)nsb, (int64_t*)nullptr);
  hipLaunchKernelGGL(seg_emit_kernel, dim3(nsb), dim3(256), 0, st, l.keys_sorted, nnz, sums2, l.seg_start, l.n_seg);
  KRS_CHECK_LAUNCH("seg_emit_kernel");
  // segments too long for one lane group (at most nnz / kLongSeg of them)
  KRS_HIP(hipMemsetAsync(l.n_long, 0, 3 * sizeof(uint32_t), st));
  KRS_HIP(hipMemsetAsync(l.sort_mode, by_table ? 1 : 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(long_list_kernel, dim3(nb), dim3(256), 0, st, l.seg_start, l.n_seg, nnz, l.n_long, l.long_list,
                     l.multi_list);
  KRS_CHECK_LAUNCH("long_list_kernel");
  return KRS_OK;
}

#if KRS_BWD_HAS(0)
extern "C" int krs_embed_bag_bwd_plan(const krs_table* tables, const krs_feature* feats, int n_feats,
                                      const void* ids, int id_type, const void* offsets, int off_type,
                                      int batch, int64_t nnz, int64_t total_rows, void* workspace,
                                      size_t workspace_bytes, int* err_flag, void* stream) {
  return plan_impl(tables, nullptr, 0, feats, nullptr, n_feats, ids, id_type, offsets, off_type, batch, nnz, total_rows,
                   workspace, workspace_bytes, err_flag, stream);
}
#endif

#if KRS_BWD_HAS(0)
extern "C" int krs_embed_bag_bwd_plan_tables(const krs_table* tables, const krs_table* tables_host, int n_tables,
                                             const krs_feature* feats, const krs_feature* feats_host, int n_feats,
                                             const void* ids, int id_type, int batch, int64_t nnz, int64_t total_rows,
                                             void* workspace, size_t workspace_bytes, int* err_flag, void* stream) {
  KRS_REQUIRE(tables_host && feats_host && n_tables > 0, "embed_bag_bwd_plan_tables: null host descriptors");
  return plan_impl(tables, tables_host, n_tables, feats, feats_host, n_feats, ids, id_type, nullptr, KRS_I32, batch, nnz,
                   total_rows, workspace, workspace_bytes, err_flag, stream);
}
#endif

#if KRS_BWD_HAS(0)
extern "C" int krs_embed_bag_bwd_dense(const krs_table* grad_tables, int n_tables, const krs_feature* feats,
                                       int n_feats, const float* weights, const float* bag_scale,
                                       const void* grad, int grad_dtype, int64_t grad_ld, int batch, int dim,
                                       int64_t nnz, const void* workspace, void* stream) {
  if (int rc = check_apply_args(grad_tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  ApplyParams p = make_apply(grad_tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  return run_apply<kDense>(p, grad_dtype, KRS_F32, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(0)
extern "C" int krs_embed_bag_bwd_fused_sgd(const krs_table* tables, int n_tables, const krs_feature* feats,
                                           int n_feats, const float* weights, const float* bag_scale,
                                           const void* grad, int grad_dtype, int64_t grad_ld, int batch, int dim,
                                           int table_dtype, int64_t nnz, const void* workspace, void* stream) {
  if (int rc = check_apply_args(tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  ApplyParams p = make_apply(tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  return run_apply<kSgd>(p, grad_dtype, table_dtype, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(1)
extern "C" int krs_embed_bag_bwd_fused_adagrad(const krs_table* tables, int n_tables, const krs_feature* feats,
                                               int n_feats, const float* weights, const float* bag_scale,
                                               const void* grad, int grad_dtype, int64_t grad_ld, int batch,
                                               int dim, int table_dtype, int64_t nnz, const void* workspace,
                                               void* stream) {
  if (int rc = check_apply_args(tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  ApplyParams p = make_apply(tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  return run_apply<kAdagrad>(p, grad_dtype, table_dtype, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(1)
extern "C" int krs_embed_bag_bwd_fused_adagrad_rowwise(const krs_table* tables, int n_tables,
                                                       const krs_feature* feats, int n_feats, const float* weights,
                                                       const float* bag_scale, const void* grad, int grad_dtype,
                                                       int64_t grad_ld, int batch, int dim, int table_dtype,
                                                       int64_t nnz, const void* workspace, void* stream) {
  if (int rc = check_apply_args(tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  ApplyParams p = make_apply(tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  return run_apply<kAdagradRow>(p, grad_dtype, table_dtype, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(2)
extern "C" int krs_embed_bag_bwd_fused_adam(const krs_table* tables, int n_tables, const krs_feature* feats,
                                            int n_feats, const float* weights, const float* bag_scale,
                                            const void* grad, int grad_dtype, int64_t grad_ld, int batch,
                                            int dim, int table_dtype, int64_t nnz, float beta_1, float beta_2,
                                            float epsilon, float bias_correction, const void* workspace,
                                            void* stream) {
  if (int rc = check_apply_args(tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  ApplyParams p = make_apply(tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  p.hyper = Hyper{beta_1, beta_2, epsilon, bias_correction};
  return run_apply<kAdam>(p, grad_dtype, table_dtype, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(2)
extern "C" int krs_embed_bag_bwd_fused_adam_dyn(const krs_table* tables, int n_tables, const krs_feature* feats,
                                                int n_feats, const float* weights, const float* bag_scale,
                                                const void* grad, int grad_dtype, int64_t grad_ld, int batch,
                                                int dim, int table_dtype, int64_t nnz, float beta_1, float beta_2,
                                                float epsilon, const float* bias_correction_dev, const void* workspace,
                                                void* stream) {
  if (int rc = check_apply_args(tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  KRS_REQUIRE(bias_correction_dev, "krs_embed_bag_bwd_fused_adam_dyn: null bias_correction_dev");
  ApplyParams p = make_apply(tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  p.hyper = Hyper{beta_1, beta_2, epsilon, 1.0f};
  p.hyper_d_dev = bias_correction_dev;
  return run_apply<kAdam>(p, grad_dtype, table_dtype, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(3)
extern "C" int krs_embed_bag_bwd_fused_ftrl(const krs_table* tables, int n_tables, const krs_feature* feats,
                                            int n_feats, const float* weights, const float* bag_scale,
                                            const void* grad, int grad_dtype, int64_t grad_ld, int batch,
                                            int dim, int table_dtype, int64_t nnz, float learning_rate_power,
                                            float l1, float l2, float beta, const void* workspace, void* stream) {
  if (int rc = check_apply_args(tables, 1, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  ApplyParams p = make_apply(tables, n_tables, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  p.hyper = Hyper{learning_rate_power, l1, l2, beta};
  return run_apply<kFtrl>(p, grad_dtype, table_dtype, reinterpret_cast<hipStream_t>(stream));
}
#endif

#if KRS_BWD_HAS(0)
extern "C" int krs_embed_bag_bwd_sparse(const krs_feature* feats, int n_feats, const float* weights,
                                        const float* bag_scale, const void* grad, int grad_dtype,
                                        int64_t grad_ld, int batch, int dim, int64_t nnz, const void* workspace,
                                        int64_t* unique_rows, float* row_grads, int64_t* n_unique, void* stream) {
  if (int rc = check_apply_args(nullptr, 0, feats, grad, grad_dtype, batch, dim, nnz, workspace)) return rc;
  KRS_REQUIRE(unique_rows && row_grads && n_unique, "embed_bag_bwd_sparse: null outputs");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (nnz == 0) {
    KRS_HIP(hipMemsetAsync(n_unique, 0, sizeof(int64_t), st));
    return KRS_OK;
  }
  const PlanLayout l = plan_layout(const_cast<void*>(workspace), nnz);
  hipLaunchKernelGGL(count_unique_kernel, dim3(1), dim3(1), 0, st, l.keys_sorted, l.n_seg, l.sort_mode, nnz, n_unique);
  KRS_CHECK_LAUNCH("count_unique_kernel");
  ApplyParams p = make_apply(nullptr, 0, feats, n_feats, weights, bag_scale, grad, grad_ld, batch, dim, nnz, workspace);
  p.unique_rows = unique_rows;
  p.row_grads = row_grads;
  return run_apply<kSparse>(p, grad_dtype, KRS_F32, st);
}
#endif
