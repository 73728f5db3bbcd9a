// Pieces shared by the dense translation units (gemm.hip, cross_epilogue.hip, dense_aux.hip; one file until round 5):
// activation helpers, the row-vector access forms of the elementwise passes, and the host side of the two-stage
// deterministic column sums (defined once, in cross_epilogue.hip).
#ifndef KRS_DENSE_COMMON_H_
#define KRS_DENSE_COMMON_H_

#include "krs_common.h"

namespace krs {

// row chunks of the column-sum walks: enough to fill the chip, few enough to keep the second stage cheap
struct ColChunks {
  int rows_per_block;
  int64_t chunks;      // = grid.y of the scalar kernels
  int64_t groups4;     // = grid.y of the vector kernels (four chunks, one per wave, per workgroup)
};
ColChunks col_chunks(int64_t m, int64_t cols);
// groups of partial sums the launch will write for an [m, n] operand walked V columns per thread
int64_t colsum_groups(int64_t m, int64_t n, int v);
// out[c] = sum over the row groups of partial[g][c] in a fixed order (colsum_finish_kernel)
int finish_colsum(float* partial, int64_t groups, int64_t n, float* out, hipStream_t st);

namespace {

__device__ __forceinline__ float apply_act(int act, float v) {
  switch (act) {
    case KRS_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case KRS_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case KRS_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ float act_grad_from_output(int act, float u) {
  switch (act) {
    case KRS_ACT_RELU: return u > 0.0f ? 1.0f : 0.0f;
    case KRS_ACT_SIGMOID: return u * (1.0f - u);
    case KRS_ACT_TANH: return 1.0f - u * u;
    default: return 1.0f;
  }
}

// one thread per 8 columns; rows strided by gridDim.y*ROWS_PER_BLOCK
template <typename T, int V>
struct RowVec;  // V contiguous elements <-> fp32
template <>
struct RowVec<float, 4> {
  typedef float4 raw_t;
  static __device__ __forceinline__ raw_t load_raw(const void* p, int64_t o) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + o);
  }
  static __device__ __forceinline__ void unpack(const raw_t& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  static __device__ __forceinline__ void load(const void* p, int64_t o, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + o);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ void store(void* p, int64_t o, const float (&f)[4]) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + o) = make_float4(f[0], f[1], f[2], f[3]);
  }
};
template <>
struct RowVec<uint16_t, 8> {
  typedef uint4 raw_t;
  static __device__ __forceinline__ raw_t load_raw(const void* p, int64_t o) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p) + o);
  }
  static __device__ __forceinline__ void unpack(const raw_t& r, float (&f)[8]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void load(const void* p, int64_t o, float (&f)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p) + o);
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  // (plain accesses: non-temporal ones made these streaming passes 10-15 % slower)
  static __device__ __forceinline__ void store(void* p, int64_t o, const float (&f)[8]) {
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p) + o) =
        make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                   pack_bf16x2(f[6], f[7]));
  }
};

}  // namespace
}  // namespace krs
#endif
