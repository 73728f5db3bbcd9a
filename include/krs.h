/*
 * krs.h -- C ABI of libkrs_hip.so: the MI355X (gfx950) hot path behind the
 * keras-rs layer API (embedding gather+pool, its index-scatter gradient,
 * DotInteraction, FeatureCross).
 *
 * The reference (keras-team/keras-rs) has NO FFI of its own: its boundary for
 * this path is the Keras layer protocol plus the DistributedEmbedding backend
 * hook set (keras_rs/src/layers/embedding/base_distributed_embedding.py:990-1042).
 * Every entry point below therefore names the reference *Python* statement(s)
 * whose arithmetic it replaces; the Python layers in keras_rs_amd/layers call
 * these through ctypes from inside torch.autograd.Function objects.
 *
 * Conventions
 *   - plain C, no torch / HIP types: device pointers are `void*`, the stream is
 *     a `hipStream_t` passed as `void*` (NULL = the null stream);
 *   - every function returns 0 on success or a negative krs_status; the text of
 *     the last failure on the calling thread is krs_last_error();
 *   - no allocation, no ownership transfer, no hidden state: the caller owns
 *     inputs, outputs and workspaces (sizes from *_workspace_bytes()); calls are
 *     asynchronous and ordered by `stream`; safe from any host thread;
 *   - all matrices are row-major; `ld*` are row strides in ELEMENTS;
 *   - index work is bit-exact; floating point accumulates in fp32.
 */
#ifndef KRS_H_
#define KRS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KRS_VERSION 100 /* 0.1.0 */

typedef enum krs_status {
  KRS_OK = 0,
  KRS_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, bad enum) */
  KRS_ERR_UNSUPPORTED = -2, /* valid but not implemented for this shape/dtype */
  KRS_ERR_LAUNCH = -3,      /* HIP launch / runtime failure */
  KRS_ERR_WORKSPACE = -4    /* workspace too small */
} krs_status;

typedef enum krs_dtype { KRS_F32 = 0, KRS_BF16 = 1 } krs_dtype;
typedef enum krs_itype { KRS_I32 = 0, KRS_I64 = 1 } krs_itype;

/* EmbedReduce combiners, embed_reduce.py:10 (SUPPORTED_COMBINERS). */
typedef enum krs_combiner { KRS_SUM = 0, KRS_MEAN = 1, KRS_SQRTN = 2 } krs_combiner;

/* keras activations FeatureCross(pre_activation=...) that are fused into the
 * GEMM epilogue; anything else is composed by the host layer. */
typedef enum krs_activation {
  KRS_ACT_NONE = 0,
  KRS_ACT_RELU = 1,
  KRS_ACT_SIGMOID = 2,
  KRS_ACT_TANH = 3
} krs_activation;

/* Bits OR'ed into the optional device-side error word of the embedding calls.
 * Out-of-range ids are never clamped: the row contributes nothing and the bit
 * is raised (SURVEY.md section 8c decision on keras ops.take out-of-range). */
#define KRS_FLAG_ID_OUT_OF_RANGE 1
#define KRS_FLAG_BAD_OFFSETS 2
#define KRS_FLAG_CAPACITY_OVERFLOW 4   /* krs_shard_route_static: lookups beyond the static capacity were dropped */

/* One embedding table ([vocab, dim] row-major).  Lives in DEVICE memory as an
 * array indexed by krs_feature.table.  Replaces the `embeddings` variable of
 * one EmbedReduce sublayer (base_distributed_embedding.py:836-852). */
typedef struct krs_table {
  void* weights;    /* [vocab, dim] fp32 or bf16 (table dtype of the call) */
  float* slot;      /* optimizer slot: Adagrad accumulator [vocab, dim] fp32, or NULL */
  int64_t row_base; /* global row id of row 0: tables of one call get disjoint
                       [row_base, row_base+vocab) ranges (sort keys of the backward) */
  int32_t vocab;
  float lr;         /* learning rate of this table's fused optimizer */
} krs_table;

/* One feature = one (ids, table) pair; several features may name one table
 * (distributed_embedding_test.py:601-652).  DEVICE memory array.
 * Bags are numbered feature-major: bag = feature * batch + sample. */
typedef struct krs_feature {
  int64_t ids_base; /* dense mode (offsets == NULL): index of this feature's first
                       id in `ids`; bag (f, b) owns ids[ids_base + b*hot .. +hot) */
  int32_t table;    /* index into the krs_table array */
  int32_t hot;      /* dense mode: ids per bag (the L of a [batch, L] input) */
  int32_t combiner; /* krs_combiner */
  int32_t out_col;  /* first column of this feature's dim-wide slot in an output row */
} krs_feature;

int krs_version(void);
const char* krs_last_error(void);

/* ------------------------------------------------------------------------- *
 * K1  fused multi-table gather + weighted segment pool (forward)
 *
 * Replaces, for ALL features of a DistributedEmbedding in one launch, the
 * per-feature chain of the reference:
 *   keras.layers.Embedding.call -> ops.take(table, ids, axis=0)   embed_reduce.py:178
 *   x * w (w = ones when weights is None)                         embed_reduce.py:224-253
 *   ops.sum(axis=-2), divide_no_nan by sum(w) / sqrt(sum(w^2))     embed_reduce.py:255-274
 *   python loop over features                                     base_distributed_embedding.py:910-928
 *
 *   out[b*out_ld + feats[f].out_col + c] =
 *        scale(f,b) * sum_{p in bag(f,b)} w[p] * table_f[ids[p], c]
 *   scale = 1 (sum) | 1/sum w (mean) | 1/sqrt(sum w^2) (sqrtn); 0 where the
 *   divisor is 0 (divide_no_nan).  Accumulation is fp32 in ascending p.
 *
 * ids      [nnz] int32/int64, feature-major concatenation of all features' ids
 * offsets  CSR mode: [n_feats*batch + 1] (int32/int64) bag boundaries into ids;
 *          NULL = dense mode (krs_feature.ids_base / hot describe the bags)
 * weights  [nnz] fp32 per-id weights or NULL (= all ones)
 * nnz      number of ids (CSR: offsets[last]); also sizes the work split
 * bag_scale optional [n_feats*batch] fp32: receives scale(f,b) for the backward
 * err_flag optional device int, OR'ed with KRS_FLAG_* bits
 * ------------------------------------------------------------------------- */
int krs_embed_bag_fwd(const krs_table* tables, const krs_feature* feats, int n_feats,
                      const void* ids, int id_type,
                      const void* offsets, int off_type,
                      const float* weights, int64_t nnz,
                      int batch, int dim, int table_dtype,
                      void* out, int out_dtype, int64_t out_ld,
                      float* bag_scale, int* err_flag, void* stream);

/* ------------------------------------------------------------------------- *
 * K2  index-scatter gradient of K1
 *
 * Replaces the autodiff of ops.take/multiply/sum (restated by the reference at
 * keras_rs/src/layers/embedding/jax/test_utils.py:395-417, accumulated per
 * table over features :450-468) and, in the fused forms, the per-table
 * optimizer step of jax/test_utils.py:474-497 (SGD: t -= lr*g;
 * Adagrad: acc += g*g; t -= lr*g/sqrt(acc), no epsilon).
 *
 *   dE_table[r, :] = sum over {p : ids[p] == r, feature(p) uses table}
 *                       w[p] * scale(bag(p)) * grad[b(p)*grad_ld + out_col + :]
 *
 * Step 1 (plan): stable sort of the nnz lookups by global row id
 * (tables[t].row_base + id).  Within one row the contributions are then
 * summed in ascending p: the result is deterministic (run-to-run bit-exact).
 * Step 2 (apply): one pass over the sorted segments that either writes the
 * dense gradient rows, or applies SGD / Adagrad to the touched rows in place.
 * ------------------------------------------------------------------------- */
size_t krs_embed_bag_bwd_workspace_bytes(int64_t nnz);

/* total_rows = max over tables of row_base+vocab (bounds the sort key bits). */
int krs_embed_bag_bwd_plan(const krs_table* tables, const krs_feature* feats, int n_feats,
                           const void* ids, int id_type,
                           const void* offsets, int off_type,
                           int batch, int64_t nnz, int64_t total_rows,
                           void* workspace, size_t workspace_bytes,
                           int* err_flag, void* stream);

/* The same plan for DENSE bags (no offsets) when the caller also holds the descriptors on the HOST
 * (tables_host / feats_host: host copies of the device arrays).  When the features of every table are
 * neighbours and tables (and their row bases) ascend with the features, each table's lookups are one
 * contiguous run and the sort runs per table on the id alone (fewer passes, 8-byte intermediate pairs);
 * any other layout takes the global sort of krs_embed_bag_bwd_plan.  The workspace it leaves is
 * consumed by the same apply calls, with one difference: out-of-range ids end their TABLE'S run
 * instead of the whole array -- which krs_embed_bag_bwd_dense and the fused forms skip wherever they
 * are; krs_embed_bag_bwd_sparse (output indexed by segment) needs the plan of krs_embed_bag_bwd_plan. */
int krs_embed_bag_bwd_plan_tables(const krs_table* tables, const krs_table* tables_host, int n_tables,
                                  const krs_feature* feats, const krs_feature* feats_host, int n_feats,
                                  const void* ids, int id_type, int batch, int64_t nnz, int64_t total_rows,
                                  void* workspace, size_t workspace_bytes, int* err_flag, void* stream);

/* Dense parity form: grad_tables[t].weights is the [vocab, dim] fp32 gradient
 * buffer of table t (krs_table array in device memory, same indexing and
 * row_base as `tables`).  Rows that are touched are OVERWRITTEN with their sum;
 * untouched rows are not written (caller zero-fills once). */
int krs_embed_bag_bwd_dense(const krs_table* grad_tables, int n_tables,
                            const krs_feature* feats, int n_feats,
                            const float* weights, const float* bag_scale,
                            const void* grad, int grad_dtype, int64_t grad_ld,
                            int batch, int dim, int64_t nnz,
                            const void* workspace, void* stream);

/* Fused SGD on the touched rows of tables[t].weights (dtype table_dtype). */
int krs_embed_bag_bwd_fused_sgd(const krs_table* tables, int n_tables,
                                const krs_feature* feats, int n_feats,
                                const float* weights, const float* bag_scale,
                                const void* grad, int grad_dtype, int64_t grad_ld,
                                int batch, int dim, int table_dtype, int64_t nnz,
                                const void* workspace, void* stream);

/* Fused Adagrad on the touched rows (tables[t].slot = fp32 accumulator). */
int krs_embed_bag_bwd_fused_adagrad(const krs_table* tables, int n_tables,
                                    const krs_feature* feats, int n_feats,
                                    const float* weights, const float* bag_scale,
                                    const void* grad, int grad_dtype, int64_t grad_ld,
                                    int batch, int dim, int table_dtype, int64_t nnz,
                                    const void* workspace, void* stream);

/* Row-wise Adagrad on the touched rows -- an OPT-IN variant, not the reference's rule (the FBGEMM / TorchRec
 * "rowwise_adagrad" form, no epsilon): acc[row] += mean_j g[row, j]^2;  w[row, j] -= lr * g[row, j] / sqrt(acc[row]).
 * tables[t].slot = fp32 [vocab]: ONE accumulator per row (the exact form keeps [vocab, dim], which is two
 * thirds of K2's HBM traffic at C3: 2 x dim x 4 bytes per touched row against 8 here). */
int krs_embed_bag_bwd_fused_adagrad_rowwise(const krs_table* tables, int n_tables,
                                            const krs_feature* feats, int n_feats,
                                            const float* weights, const float* bag_scale,
                                            const void* grad, int grad_dtype, int64_t grad_ld,
                                            int batch, int dim, int table_dtype, int64_t nnz,
                                            const void* workspace, void* stream);

/* Fused Adam on the touched rows ("lazy": a row that is not looked up keeps its moments and its
 * value).  The reference names keras.optimizers.Adam for 'sparsecore' tables and hands
 * (learning_rate, beta_1, beta_2, epsilon) to the SparseCore library
 * (embedding/jax/config_conversion.py:256-265; amsgrad unsupported); the arithmetic restated here
 * is Keras' Adam.update_step:
 *   m += (g - m)(1 - beta_1);  v += (g*g - v)(1 - beta_2);
 *   w -= lr * bias_correction * m / (sqrt(v) + epsilon),
 *   bias_correction = sqrt(1 - beta_2^t) / (1 - beta_1^t), t = 1-based step count (host side).
 * tables[t].slot = fp32 [2][vocab][dim]: m then v.  tables[t].lr = learning rate. */
int krs_embed_bag_bwd_fused_adam(const krs_table* tables, int n_tables,
                                 const krs_feature* feats, int n_feats,
                                 const float* weights, const float* bag_scale,
                                 const void* grad, int grad_dtype, int64_t grad_ld,
                                 int batch, int dim, int table_dtype, int64_t nnz,
                                 float beta_1, float beta_2, float epsilon, float bias_correction,
                                 const void* workspace, void* stream);

/* Fused FTRL on the touched rows: keras.optimizers.Ftrl.update_step with the options the
 * reference supports (embedding/jax/config_conversion.py:266-283: learning_rate_power, l1, l2,
 * beta, initial_accumulator_value; no l2 shrinkage):
 *   n' = n + g*g;  z += g - (n'^-p - n^-p) / lr * w;
 *   w = (clip(z, -l1, l1) - z) / (n'^-p / lr + 2*(l2 + beta / (2 lr)));  n = n'.
 * tables[t].slot = fp32 [2][vocab][dim]: accumulator n then linear z. */
int krs_embed_bag_bwd_fused_ftrl(const krs_table* tables, int n_tables,
                                 const krs_feature* feats, int n_feats,
                                 const float* weights, const float* bag_scale,
                                 const void* grad, int grad_dtype, int64_t grad_ld,
                                 int batch, int dim, int table_dtype, int64_t nnz,
                                 float learning_rate_power, float l1, float l2, float beta,
                                 const void* workspace, void* stream);

/* krs_embed_bag_bwd_fused_adam with the bias-correction factor read from DEVICE memory when the kernel runs
 * (*bias_correction_dev, one float): the form a step replayed from a HIP graph needs -- a by-value argument is frozen
 * into the captured launch, the device float is rewritten before every replay (krs_store_f32).  The layers use this
 * entry for eager steps as well, so both kinds of step run one code path. */
int krs_embed_bag_bwd_fused_adam_dyn(const krs_table* tables, int n_tables,
                                     const krs_feature* feats, int n_feats,
                                     const float* weights, const float* bag_scale,
                                     const void* grad, int grad_dtype, int64_t grad_ld,
                                     int batch, int dim, int table_dtype, int64_t nnz,
                                     float beta_1, float beta_2, float epsilon, const float* bias_correction_dev,
                                     const void* workspace, void* stream);

/* Step-dependent optimizer constants (scheduled learning rates, Adam's bias correction) -> device memory without a copy
 * from the host: values_host[0..count) are read NOW (on the calling thread) and travel as kernel arguments; the kernel
 * stores value i at (char*)dst + i * stride_bytes.  stride_bytes = sizeof(krs_table) with dst = &tables[0].lr rewrites the
 * learning rates of a descriptor array in place; stride 4 fills a float array.  No page-locked buffer the host could
 * overwrite too early, no wait: ordered on `stream` like any launch, so an eager call placed before a graph replay on the
 * same stream is seen by that replay and not by the one before it.  The reference evaluates schedules on the host every
 * step too (jax/config_conversion.py:136-176: callable learning rates). */
int krs_store_f32(void* dst, int64_t stride_bytes, const float* values_host, int count, void* stream);

/* Sparse form: unique global rows and their summed gradients.
 *   unique_rows [nnz] int64 (first *n_unique valid), row_grads [nnz, dim] fp32,
 *   n_unique device int64.
 * Needs the plan of krs_embed_bag_bwd_plan (global sort).  A workspace holding the plan of
 * krs_embed_bag_bwd_plan_tables yields *n_unique = -1 (unique_rows / row_grads are then meaningless; the
 * workspace itself records which sort filled it, so the answer follows a plan copied to another address). */
int krs_embed_bag_bwd_sparse(const krs_feature* feats, int n_feats,
                             const float* weights, const float* bag_scale,
                             const void* grad, int grad_dtype, int64_t grad_ld,
                             int batch, int dim, int64_t nnz,
                             const void* workspace,
                             int64_t* unique_rows, float* row_grads, int64_t* n_unique,
                             void* stream);

/* ------------------------------------------------------------------------- *
 * K3  FeatureCross (DCN-v2 cross layer)
 *
 * Replaces FeatureCross.call, feature_cross.py:182-194:
 *   u = act(h @ kernel + bias)            keras Dense inside the layer (:134-151)
 *   u = cast(u, compute dtype); u += diag_scale * x   (:189-192)
 *   y = x0 * u + x                         (:194)
 * with h = x (full rank) or h = x @ down_kernel (low rank, :185-187).
 *
 * krs_gemm: C[M,N] = epilogue(A[M,K] @ B), the MFMA kernel that owns the dense
 * projection.  B is given K-contiguous ("B transposed", Bt[N,K]) when
 * b_is_nk != 0, else as the keras kernel layout B[K,N].  Likewise a_is_km != 0
 * means A is given as At[K,M] (the weight-gradient contractions over the
 * batch).  Epilogue, applied on the fp32 accumulator v of element (m,n):
 *      v += bias[n]                (bias != NULL)
 *      v  = act(v)
 *      v  = x0[m,n] * (v + diag_scale * x[m,n]) + x[m,n]   (x0 != NULL: cross)
 *      v += beta * R[m,n]          (R != NULL: residual / gradient accumulate)
 *   u_out (optional, cross form only) receives v = act(A@B + bias), the
 *   activation output the backward needs (dx0 = g*(v + diag_scale*x), act'(v)).
 * ------------------------------------------------------------------------- */
typedef struct krs_gemm_epilogue {
  const float* bias;  /* [N] fp32 or NULL */
  int32_t act;        /* krs_activation */
  float diag_scale;
  const void* x0;     /* [M,N] (dtype of C) or NULL */
  const void* x;      /* [M,N] (dtype of C), required when x0 != NULL */
  int64_t ldx;        /* row stride of x0 and x */
  void* u_out;        /* [M,N] (dtype of C) or NULL */
  int64_t ldu;
  const void* r;      /* [M,N] (dtype of C) or NULL */
  int64_t ldr;
  float beta;
} krs_gemm_epilogue;

int krs_gemm(const void* a, int64_t lda, int a_is_km,
             const void* b, int64_t ldb, int b_is_nk,
             void* c, int64_t ldc,
             int64_t m, int64_t n, int64_t k,
             int in_dtype, int out_dtype,
             const krs_gemm_epilogue* epilogue,
             void* workspace, size_t workspace_bytes, void* stream);
/* Bytes of `workspace` a krs_gemm call of this shape needs (0 = none): split-K products keep one fp32 [M, N] slab per
 * split and reduce them in a fixed order (deterministic, no atomics) -- the weight-gradient contractions over the batch
 * (a_is_km), and since round 5 K-contiguous products whose output is too small for 256 x 256 tiles to fill the chip but
 * whose K is long (M = 8192 against N = 512, K = 3456: the per-rank products of a strongly-scaled job).  A call with less
 * workspace than this returns KRS_ERR_WORKSPACE. */
size_t krs_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k, int a_is_km);

/* Data-gradient product of a cross layer FUSED with the elementwise backward of the cross layer below it in a stack
 * on one x0 (`xl = layer(x0, xl)` repeated, examples/ml_perf/model.py:332-336; gradients of feature_cross.py:182-194):
 *      G   = A[M,K] @ Bt[N,K]^T [+ beta * R]          dL/dx of the upper layer = dL/dy of the lower one (stored;
 *                                                      R == NULL: no residual -- the layer above is a Dense layer)
 *      dz  = G * x0 * act'(u)                          d(pre-activation) of the lower layer   (act' from its saved output u)
 *      dx0 = [dx0 +] G * u [+ G]                       its term of dL/dx0 (dx0_accumulate != 0: added to what dx0 holds;
 *                                                      fold_direct != 0: the lower layer's x IS x0, so the direct term
 *                                                      joins, the `dxd == dx0` rule of krs_cross_epilogue_bwd)
 *                                                      u_upper != NULL (needs R, beta = 1, dx0_accumulate = 0): the term of
 *                                                      the layer ABOVE joins here, dx0 = R * u_upper + G * u [+ G] -- R is its
 *                                                      dL/dy, u_upper its saved activation output; that layer then writes
 *                                                      no dL/dx0 of its own and this sum is rounded once, not twice)
 *                                                      dx0 == NULL (R == NULL form only): nothing is written -- the caller
 *                                                      passes u as the u_upper of the NEXT launch, whose R is this G
 *      dbias[n] = sum_m dz[m,n]                        fp32, fixed summation order (NULL: not wanted)
 * i.e. exactly krs_gemm(A, Bt, epilogue{r = R, beta}) followed by krs_cross_epilogue_bwd(g = G, u, x0, diag_scale = 0),
 * with G, dz and dx0 BIT-IDENTICAL to those two calls (dz and dx0 are computed from G as it is stored, after its one
 * rounding) -- without the second pass reading G, x0 and u back and with the streams written by the product's epilogue.
 * Row stride `ld` for x0, u, dz and dx0; workspace: krs_gemm_cross_bwd_workspace_bytes(m, n) when dbias is wanted.
 * bf16 tiles the 256x256 ring kernel covers run fused; every other shape / dtype runs the two calls (needs ldg == ld).
 *
 * DENSE form (round 6): x0 == NULL -- the layer below is a Dense layer (examples/ml_perf/model.py:214-262), u its saved
 * output y:  dz = G * act'(y),  dbias = column sums of dz;  R, dx0, u_upper, fold_direct must be absent and g_out may be NULL:
 * G itself is NOT stored (the raw data gradient of a Dense output has no other reader), so the launch streams y in and dz out
 * where krs_gemm + krs_dense_act_bwd write G, read G and y and write dz.  dz is bit-identical to those two calls (one
 * rounding of G, then the derivative). */
size_t krs_gemm_cross_bwd_workspace_bytes(int64_t m, int64_t n);
int krs_gemm_cross_bwd(const void* a, int64_t lda, const void* bt, int64_t ldb,
                       const void* r, int64_t ldr, float beta,
                       void* g_out, int64_t ldg,
                       const void* x0, const void* u, void* dz, void* dx0, int64_t ld, int dx0_accumulate,
                       const void* u_upper, int fold_direct, float* dbias, int64_t m, int64_t n, int64_t k, int act, int dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Tuning / diagnostic switches of krs_gemm (process-wide; results never depend on them).
 *   KRS_GEMM_OPT_PIPELINE: 4 = the big bf16 shapes run the four-stage ping-pong ring on 256x256 tiles (default, or
 *   the environment variable KRS_GEMM_PIPE at first use); 0 = every shape runs the two-stage 128x128 kernels and
 *   krs_gemm_cross_bwd its two-call form -- the reference schedule of the bit-for-bit tests and A/B harnesses.
 *   5 = the 32-k ring (gemm_pp256_kernel) also for the K-contiguous products that take the 64-k ring (gemm_pp64_kernel) by
 *   default: the A/B switch of round 6.
 *   (Rounds 2-4 used 5 / 6 / 7 for a five-stage ring, a prefetch schedule and a deep ring on 128x128 tiles; measured no
 *   faster and deleted in round 5.) */
enum { KRS_GEMM_OPT_PIPELINE = 0 };
int krs_gemm_set_option(int key, int value);

/* Tuning switches of krs_embed_bag_fwd / krs_embed_bag_bwd_* (process-wide; results never depend on them).
 *   KRS_EMBED_OPT_PLAN: krs_embed_bag_bwd_plan_tables -- 0 = table-segmented sort where the layout allows it
 *   (default), 1 = always the global sort (A/B).
 *   KRS_EMBED_OPT_HOTROWS: LDS staging of hot embedding rows in the pooled gather -- 0 = off (default), 64 / 128 =
 *   rows 0 .. n-1 of a workgroup's table are copied to LDS and lookups of them are served from there (one flat
 *   16-byte load per lookup, routed to LDS or memory by its address); pays only when ids are relabelled
 *   hot-first and those rows are NOT already cache hits (profiles/r3_k1_hot_rows_lds.txt: they are).
 *   (Keys 0 and 1 -- the one-hot gather variants and the round-1 per-segment backward kernel -- were retired in
 *   round 5 with the kernels they selected; they are refused.) */
enum { KRS_EMBED_OPT_PLAN = 2, KRS_EMBED_OPT_HOTROWS = 3 };
int krs_embed_set_option(int key, int value);

/* Elementwise halves of FeatureCross for the host-composed path (arbitrary
 * pre_activation callables) and for the backward:
 *   fwd: y = x0 * (u + diag_scale*x) + x                     feature_cross.py:191-194
 *   bwd: given g = dL/dy and u = act(z) (z = h@K+b):
 *        dx0 (+)= g * (u + diag_scale*x)
 *        dxd = g + diag_scale * g*x0      (the non-GEMM part of dL/dx)
 *              dxd == dx0 (same pointer) is the case "x is x0": dx0 then
 *              receives the sum of both lines and nothing else is written
 *        du  = dz = g*x0 * act'(z), act' written through u (relu: u>0,
 *              sigmoid: u(1-u), tanh: 1-u^2)
 *        dbias[n] = sum_m dz[m,n]         (optional)
 * Bias gradients (here, krs_dense_act_bwd, krs_colsum) are column sums over all m rows.  With a workspace of
 * krs_colsum_workspace_bytes(m, n) bytes they are formed in two stages -- per row group, then over the groups in
 * order -- and come out with the same bits on every run; with workspace == NULL the groups are added with fp32
 * atomics, whose order (and so the last bits) varies from run to run. */
size_t krs_colsum_workspace_bytes(int64_t m, int64_t n);
int krs_cross_epilogue_fwd(const void* u, const void* x0, const void* x, void* y,
                           int64_t m, int64_t n, int64_t ld, float diag_scale,
                           int dtype, void* stream);
int krs_cross_epilogue_bwd(const void* g, const void* u, const void* x0, const void* x,
                           void* du, void* dx0, int dx0_accumulate, void* dxd,
                           float* dbias,
                           int64_t m, int64_t n, int64_t ld, float diag_scale, int act,
                           int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Weight preparation for the two GEMM layouts of a Dense / FeatureCross step: dst [rows, cols] = cast(src) and
 * dst_t [cols, rows] = cast(src)^T, either may be NULL.  What `ops.cast(kernel, compute_dtype)` +
 * a transposed copy do on the host side of feature_cross.py:182-194 under a mixed-precision policy, in one
 * launch (bf16 rounding: round-to-nearest-even, as everywhere in this library). */
int krs_cast_transpose(const void* src, int64_t rows, int64_t cols, int64_t ld_src, int src_dtype,
                       void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t, int dst_dtype,
                       void* stream);
/* The same for several contiguous weights in ONE launch (the bf16 copies of every dense kernel of a model, refreshed
 * behind the optimizer step): srcs[i] is [rows[i], cols[i]] row-major, dsts[i] (same shape) and / or dst_ts[i]
 * ([cols[i], rows[i]]) receive the cast copy / its transpose; either output pointer of a tensor may be NULL. */
int krs_cast_transpose_many(int count, const void* const* srcs, const int64_t* rows, const int64_t* cols,
                            int src_dtype, void* const* dsts, void* const* dst_ts, int dst_dtype, void* stream);

/* Backward of the bias + activation epilogue of a Dense layer (keras.layers.Dense inside the DLRM MLP blocks,
 * examples/ml_perf/model.py:214-262): dz [m,n] = g * act'(y) with the derivative taken from the saved output y
 * (relu: y > 0; sigmoid: y(1-y); tanh: 1-y^2; none: 1, y may be NULL) and dbias [n] = column sums of dz (fp32).
 * dz or dbias may be NULL. */
int krs_dense_act_bwd(const void* g, int64_t ld_g, const void* y, int64_t ld_y, void* dz, int64_t ld_dz,
                      float* dbias, int64_t m, int64_t n, int act, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Adagrad step on a list of dense fp32 weights (the FeatureCross / Dense kernels and biases of one training step) in
 * one launch: acc += g*g; p -= lr * g / (sqrt(acc) + eps).  params / grads / accs / sizes: HOST arrays of `count`
 * device pointers and element counts.  The optimizer the ml_perf example attaches to the dense part
 * (examples/ml_perf/main.py: keras.optimizers.Adagrad, learning rate of configs/v6e_8.py); the fused table
 * optimizers are K2's. */
int krs_dense_adagrad(float* const* params, const float* const* grads, float* const* accs,
                      const int64_t* sizes, int count, float lr, float eps, void* stream);

/* Column sum: out[n] = sum_m a[m,n] (fp32 out).  Dense bias gradient. */
int krs_colsum(const void* a, int64_t lda, int64_t m, int64_t n, int dtype,
               float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * K4  DotInteraction
 *
 * Replaces DotInteraction.call, dot_interaction.py:170-203: stack F features
 * [B,D] -> X[B,F,D]; P = X X^T; then either the row-major strictly/inclusive
 * lower triangle gather (indices of :118-132) or the masked flatten (:96-116).
 * feats: HOST array of F device pointers, feature f is [batch, dim] with row
 * stride ld[f] (views into one concat buffer are fine).
 *   out [batch, out_cols], out_cols = F*F (skip_gather) | F(F+1)/2 | F(F-1)/2
 * Backward: dX[b,i,:] = sum_j (G[b,i,j] + G[b,j,i]) X[b,j,:], G = the gradient
 * scattered back to [F,F] (zero above / on the masked part).
 * Any F (the reference has no limit): the MFMA path covers F <= 32 with 16-B aligned rows, one plain launch
 * F <= 64, beyond that one launch per (32 features i, 128 features j) block pair.  accumulate_mask addresses the
 * first 64 features.
 * ------------------------------------------------------------------------- */
int krs_dot_interaction_fwd(const void* const* feats, const int64_t* ld, int n_feats,
                            int64_t batch, int dim, int dtype,
                            int self_interaction, int skip_gather,
                            void* out, int64_t out_ld, void* stream);
int krs_dot_interaction_bwd(const void* const* feats, const int64_t* ld, int n_feats,
                            int64_t batch, int dim, int dtype,
                            int self_interaction, int skip_gather,
                            const void* grad_out, int64_t grad_ld,
                            void* const* grad_feats, const int64_t* grad_feat_ld,
                            void* stream);
/* The same with gradients that are already there: feature f with bit f of accumulate_mask set receives
 * grad_feats[f] + dX[f] (fp32 sum of the stored value and the new term, one rounding); the others are
 * overwritten.  Lets the gradient of the interaction join a gradient buffer the features' other consumer has
 * already written (the concat of the same features: examples/ml_perf/model.py:204-207) without a separate add. */
int krs_dot_interaction_bwd_accumulate(const void* const* feats, const int64_t* ld, int n_feats,
                                       int64_t batch, int dim, int dtype,
                                       int self_interaction, int skip_gather,
                                       const void* grad_out, int64_t grad_ld,
                                       void* const* grad_feats, const int64_t* grad_feat_ld,
                                       uint64_t accumulate_mask, void* stream);

/* ------------------------------------------------------------------------- *
 * K5  MOD bucketise for row-sharded tables
 *
 * The reference shards rows MOD-N (sharding_strategy="MOD",
 * jax/embedding_utils.py:194; layout tensorflow/distributed_embedding.py:316-328):
 * global row r lives on shard r % n_shards at local row r / n_shards.
 * Stable counting sort of the nnz ids by destination shard:
 *   bucket_counts [n_shards] int64: ids per shard
 *   local_ids     [nnz] int32/int64 (id_type): ids[perm[i]] / n_shards, grouped by shard
 *   perm          [nnz] int32: source position of each bucketed id
 * ------------------------------------------------------------------------- */
size_t krs_mod_bucketize_workspace_bytes(int64_t nnz, int n_shards);
int krs_mod_bucketize(const void* ids, int id_type, int64_t nnz, int n_shards,
                      void* local_ids, int32_t* perm, int64_t* bucket_counts,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * K7  Binary cross-entropy of the DLRM head, forward + backward in one pass
 *
 * keras.losses.BinaryCrossentropy() as examples/ml_perf/main.py:201-210 compiles it (from_logits=False,
 * mean reduction) on the sigmoid output of the top MLP (examples/ml_perf/model.py:105-163):
 *   p = clip(pred, epsilon, 1 - epsilon);  loss = mean_i -(y_i log p_i + (1 - y_i) log(1 - p_i))
 *   dpred_i = grad_scale / n * ((1 - y_i) / (1 - p_i) - y_i / p_i), 0 where the clip is active
 * pred [n] fp32 / bf16 (contiguous), labels [n] fp32, loss = device scalar, dpred [n] in pred's dtype
 * (may be NULL: forward only).  partials: KRS_BCE_MAX_BLOCKS floats of scratch (device), or NULL = one workgroup
 * does all of it.  fp32 arithmetic, fixed summation order either way (deterministic for a given scratch choice).
 * ------------------------------------------------------------------------- */
#define KRS_BCE_MAX_BLOCKS 64
int krs_bce_fwd_bwd(const void* pred, int pred_dtype, const float* labels, int64_t n, float epsilon,
                    float grad_scale, float* loss, void* dpred, float* partials, void* stream);

/* ------------------------------------------------------------------------- *
 * K6  Row-sharded lookup: route / unpack / combine (the id side of the exchange)
 *
 * The reference's accelerated path hands per-partition id lists to the SparseCore library
 * (embedding.preprocess_sparse_dense_matmul_input, sharding_strategy="MOD":
 * jax/embedding_utils.py:144-217) and looks them up with tpu_sparse_dense_matmul
 * (jax/embedding_lookup.py:134-147).  Here the ranks exchange one PARTIALLY POOLED vector per
 * (bag, owner) pair (keras_rs_amd/sharded.py); these three calls are everything around the two
 * all-to-alls, each a fixed sequence of kernels on `stream` (no allocation, no host sync).
 *
 * krs_shard_route (home rank).  Lookups are given as for krs_embed_bag_fwd: ONE flat id buffer,
 * feature-major, dense bags (feats[f].hot ids per bag, feats[f].ids_base = first id of the feature)
 * or CSR (offsets[n_feats*batch + 1]).  Per lookup: id outside [0, feats[f].vocab) -> dropped,
 * KRS_FLAG_ID_OUT_OF_RANGE raised (never clamped); composite c = comp_off + id, owner = c % n_shards,
 * stacked local row = c / n_shards; stable grouping by owner; a SEGMENT = a run of one bag inside an
 * owner's bucket.  Outputs:
 *   packed   [<= nnz*(1 + emit_weights) + nnz] int32: per owner d, contiguously,
 *            [rows(lookups_d) | weights as fp32 bit patterns (lookups_d, only with emit_weights) |
 *             segment lengths (segments_d)]; weight = user weight x combiner scale of the bag
 *            (mean 1/sum w, sqrtn 1/sqrt(sum w^2), 0 where the divisor is 0: embed_reduce.py:255-274)
 *   seg_bag  [nnz] bag of every segment (first n_seg valid; segments are numbered in bucket order)
 *   seg_grow [nnz] row of the [batch*n_feats, dim] view of the output gradient the segment's partial
 *            takes in the backward: (bag % batch)*n_feats + bag / batch
 *   bag_seg  [n_feats*batch, n_shards] segment of (bag, owner), -1 where the bag has no lookup there
 *   counts   [3*n_shards] int64: lookups, segments, packed words per owner
 * emit_weights must be set when weights != NULL or any combiner is not KRS_SUM.
 * feats is the DEVICE copy of the descriptors, feats_host the same array on the host.
 *
 * krs_shard_unpack (owner rank): the packed blocks received from n_sources ranks (their `lookups` /
 * `segments` counts are HOST arrays), concatenated in source order -> rows[sum lookups],
 * w[sum lookups] (weighted != 0), offsets[sum segments + 1] (exclusive scan of the lengths): the CSR
 * form krs_embed_bag_fwd and the fused backward take.
 *
 * krs_shard_combine (home rank): out[b, f*dim + j] = sum over owners d of
 * partials[bag_seg[(f*batch + b)*n_shards + d]][j] (skipping -1), fp32 accumulation in ascending d,
 * one rounding to `dtype` (the dtype of partials and out).
 *
 * krs_publish_i64: copies src[0..n) to page-locked host memory (host_dst, device-visible) and then
 * writes `seq` to host_dst[n]; the host polls host_dst[n] instead of synchronising the stream.
 *
 * STATIC-CAPACITY form (krs_shard_route_static / krs_shard_unpack_static).  The reference makes every
 * buffer of the SparseCore exchange static with TableConfig.max_ids_per_partition /
 * max_unique_ids_per_partition (distributed_embedding_config.py:54-61), drops what does not fit
 * (allow_id_dropping=True, jax/embedding_utils.py:187-197) and learns better limits from running statistics
 * (update_stats, jax/distributed_embedding.py:657-664).  Same contract here: every (home, owner) pair
 * exchanges a block of fixed size, so the all-to-alls have equal splits and the host needs no count:
 *   block = [lookups kept, segments kept, need_l, need_s | rows[cap_lookups] |
 *            weights[cap_lookups] (only with emit_weights) | segment lengths[cap_segments]]   (int32 words;
 *            krs_shard_static_block_words() of them; unused slots are 0)
 * krs_shard_route_static writes n_shards such blocks into `packed`: owner d keeps the first cap_segments
 * segments of its bucket and of those the first cap_lookups lookups (a segment cut by the limit keeps its
 * head); anything dropped raises KRS_FLAG_CAPACITY_OVERFLOW in err_flag.  need_l / need_s = the largest
 * per-owner lookup / segment count of THIS call, repeated in every header (each rank learns every rank's
 * needs from the blocks it receives).  seg_grow is [n_shards*cap_segments] (gradient row of every segment
 * SLOT d*cap_segments + j; 0 for unused slots), bag_seg holds slots (-1 for no / dropped segment),
 * counts as in the exact form (before dropping).  Capacities must be multiples of 4.  When nothing
 * overflows, the kept lookups, their order and the segments are those of krs_shard_route.
 * krs_shard_unpack_static (owner): n_sources received blocks -> rows / w [n_sources*cap_lookups] (compact,
 * source order; the unused tail is row -1 / weight 0: outside every segment, and an invalid id for the
 * backward plan), offsets[n_sources*cap_segments + 1] (CSR over the segment SLOTS; unused slots are empty),
 * stats[4] (optional) = max need_l, max need_s over the sources, lookups, segments received.
 * ------------------------------------------------------------------------- */
typedef struct krs_shard_feature {
  int64_t ids_base;   /* dense bags: position of the feature's first id in `ids` */
  int64_t comp_off;   /* stacked local row offset of the feature's table, times n_shards */
  int32_t hot;        /* dense bags: ids per bag; ignored with CSR offsets */
  int32_t combiner;   /* krs_combiner */
  int32_t vocab;      /* valid ids are [0, vocab) */
  int32_t reserved;
} krs_shard_feature;

size_t krs_shard_route_workspace_bytes(int64_t nnz, int64_t n_bags, int n_shards);
int krs_shard_route(const krs_shard_feature* feats, const krs_shard_feature* feats_host, int n_feats,
                    const void* ids, int id_type, const void* offsets, int offset_type,
                    const float* weights, int64_t nnz, int batch, int n_shards, int emit_weights,
                    int32_t* packed, int32_t* seg_bag, int32_t* seg_grow, int32_t* bag_seg,
                    int64_t* counts, int32_t* err_flag,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t krs_shard_unpack_workspace_bytes(int64_t total_segments);
int krs_shard_unpack(const int32_t* packed, int n_sources, const int64_t* lookups, const int64_t* segments,
                     int weighted, int32_t* rows, float* w, int32_t* offsets,
                     void* workspace, size_t workspace_bytes, void* stream);
int krs_shard_combine(const void* partials, const int32_t* bag_seg, int batch, int n_feats, int n_shards,
                      int dim, int dtype, void* out, int64_t out_ld, void* stream);
int64_t krs_shard_static_block_words(int64_t cap_lookups, int64_t cap_segments, int emit_weights);
int krs_shard_route_static(const krs_shard_feature* feats, const krs_shard_feature* feats_host, int n_feats,
                           const void* ids, int id_type, const void* offsets, int offset_type,
                           const float* weights, int64_t nnz, int batch, int n_shards, int emit_weights,
                           int64_t cap_lookups, int64_t cap_segments,
                           int32_t* packed, int32_t* seg_bag, int32_t* seg_grow, int32_t* bag_seg,
                           int64_t* counts, int32_t* err_flag,
                           void* workspace, size_t workspace_bytes, void* stream);
int krs_shard_unpack_static(const int32_t* packed, int n_sources, int64_t cap_lookups, int64_t cap_segments,
                            int weighted, int32_t* rows, float* w, int32_t* offsets, int64_t* stats,
                            void* workspace, size_t workspace_bytes, void* stream);
int krs_publish_i64(const int64_t* src, int n, int64_t* host_dst, int64_t seq, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KRS_H_ */
