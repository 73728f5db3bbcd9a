// K5 -- MOD bucketise of ids for row-sharded tables (stable counting sort by shard).
//
// The reference shards embedding rows MOD-N on its accelerated path
// (sharding_strategy="MOD", keras_rs/src/layers/embedding/jax/embedding_utils.py:194;
// layout documented at tensorflow/distributed_embedding.py:316-328): global row r
// lives on shard r % N at local row r / N.  This is the id-side half of that
// exchange: group the ids by destination shard (keeping their order), convert
// them to local rows and remember where each came from.  Pure integer work,
// bit-exact against the oracle.
//   pass 1  per-block histogram (wave ballots, no atomics on global memory)
//   pass 2  one-block exclusive scan over [shard][block]
//   pass 3  stable scatter: rank inside the block from wave ballots + LDS prefix
#include "krs_common.h"

namespace krs {
namespace {

constexpr int kItems = 8;                 // sub-rounds of 256 consecutive ids per block
constexpr int kChunk = 256 * kItems;      // ids per block
constexpr int kMaxShards = 64;

__device__ __forceinline__ int shard_of(int64_t id, int n) {
  int64_t s = id % n;
  return (int)(s < 0 ? s + n : s);
}

__global__ __launch_bounds__(256) void bucket_hist_kernel(const void* ids, int id64, int64_t nnz, int n_shards,
                                                          int* block_counts) {
  __shared__ int cnt[kMaxShards];
  if (threadIdx.x < kMaxShards) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < kItems; ++it) {
    const int64_t q = base + it * 256 + threadIdx.x;
    const int s = q < nnz ? shard_of(ld_index(ids, id64, q), n_shards) : -1;
    for (int t = 0; t < n_shards; ++t) {
      const unsigned long long m = __ballot(s == t);
      if (lane == 0 && m) atomicAdd(&cnt[t], __popcll(m));
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_shards) block_counts[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of block_counts laid out [shard][block] (shard-major = final order); also bucket totals
__global__ __launch_bounds__(256) void bucket_scan_kernel(int* block_counts, int64_t n_entries, int n_blocks,
                                                          int n_shards, int64_t* bucket_counts) {
  __shared__ long long carry;
  __shared__ long long wsum[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_entries; base += 256) {
    const int64_t i = base + threadIdx.x;
    const int v = i < n_entries ? block_counts[i] : 0;
    long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(x, o, 64);
      if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    long long off = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < n_entries) block_counts[i] = (int)(off + x - v);  // exclusive; fits int (nnz < 2^31 checked by host)
    __syncthreads();
    if (threadIdx.x == 255) carry = off + x;
    __syncthreads();
  }
  (void)n_blocks;
  (void)n_shards;
  (void)bucket_counts;
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(const void* ids, int id64, int64_t nnz, int n_shards,
                                                             const int* block_offsets, void* local_ids,
                                                             int32_t* perm) {
  __shared__ int run[kMaxShards];        // next free slot of each shard for this block
  __shared__ int wcnt[4][kMaxShards];    // per-wave counts of the current sub-round
  if ((int)threadIdx.x < n_shards) run[threadIdx.x] = block_offsets[(int64_t)threadIdx.x * gridDim.x + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < kItems; ++it) {
    const int64_t q = base + it * 256 + threadIdx.x;
    int64_t id = 0;
    int s = -1;
    if (q < nnz) {
      id = ld_index(ids, id64, q);
      s = shard_of(id, n_shards);
    }
    int rank = 0;
    for (int t = 0; t < n_shards; ++t) {
      const unsigned long long m = __ballot(s == t);
      if (s == t) rank = __popcll(m & ((1ULL << lane) - 1ULL));
      if (lane == 0) wcnt[wave][t] = __popcll(m);
    }
    __syncthreads();
    if (s >= 0) {
      int pos = run[s] + rank;
      for (int w = 0; w < wave; ++w) pos += wcnt[w][s];
      const int64_t loc = (id - s) / n_shards;
      if (id64) reinterpret_cast<int64_t*>(local_ids)[pos] = loc;
      else reinterpret_cast<int32_t*>(local_ids)[pos] = (int32_t)loc;
      perm[pos] = (int32_t)q;
    }
    __syncthreads();
    if ((int)threadIdx.x < n_shards)
      run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    __syncthreads();
  }
}

__global__ void bucket_totals_kernel(const int* block_offsets, int n_blocks, int n_shards, int64_t nnz,
                                     int64_t* bucket_counts) {
  const int s = threadIdx.x;
  if (s >= n_shards) return;
  const int64_t beg = block_offsets[(int64_t)s * n_blocks];
  const int64_t end = s + 1 < n_shards ? block_offsets[(int64_t)(s + 1) * n_blocks] : nnz;
  bucket_counts[s] = end - beg;
}

}  // namespace
}  // namespace krs

using namespace krs;

extern "C" size_t krs_mod_bucketize_workspace_bytes(int64_t nnz, int n_shards) {
  if (nnz < 0 || n_shards <= 0) return 0;
  const int64_t blocks = ceil_div(nnz > 0 ? nnz : 1, kChunk);
  return (size_t)blocks * (size_t)n_shards * sizeof(int) + 256;
}

extern "C" int krs_mod_bucketize(const void* ids, int id_type, int64_t nnz, int n_shards, void* local_ids,
                                 int32_t* perm, int64_t* bucket_counts, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  KRS_REQUIRE(n_shards > 0 && n_shards <= kMaxShards, "mod_bucketize: n_shards must be in [1, 64]");
  KRS_REQUIRE(nnz >= 0 && nnz < 0x7fffffffLL, "mod_bucketize: nnz must fit int32");
  KRS_REQUIRE(bucket_counts, "mod_bucketize: null bucket_counts");
  KRS_REQUIRE(id_type == KRS_I32 || id_type == KRS_I64, "mod_bucketize: bad id type");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (nnz == 0) {
    KRS_HIP(hipMemsetAsync(bucket_counts, 0, (size_t)n_shards * sizeof(int64_t), st));
    return KRS_OK;
  }
  KRS_REQUIRE(ids && local_ids && perm && workspace, "mod_bucketize: null argument");
  if (workspace_bytes < krs_mod_bucketize_workspace_bytes(nnz, n_shards))
    return fail(KRS_ERR_WORKSPACE, "mod_bucketize: workspace too small");
  const int blocks = (int)ceil_div(nnz, kChunk);
  int* counts = reinterpret_cast<int*>(workspace);
  const int id64 = id_type == KRS_I64;
  hipLaunchKernelGGL(bucket_hist_kernel, dim3(blocks), dim3(256), 0, st, ids, id64, nnz, n_shards, counts);
  KRS_CHECK_LAUNCH("bucket_hist_kernel");
  hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(256), 0, st, counts, (int64_t)blocks * n_shards, blocks,
                     n_shards, bucket_counts);
  KRS_CHECK_LAUNCH("bucket_scan_kernel");
  hipLaunchKernelGGL(bucket_totals_kernel, dim3(1), dim3(64), 0, st, counts, blocks, n_shards, nnz, bucket_counts);
  KRS_CHECK_LAUNCH("bucket_totals_kernel");
  hipLaunchKernelGGL(bucket_scatter_kernel, dim3(blocks), dim3(256), 0, st, ids, id64, nnz, n_shards, counts,
                     local_ids, perm);
  KRS_CHECK_LAUNCH("bucket_scatter_kernel");
  return KRS_OK;
}
