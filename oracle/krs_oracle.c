/*
 * krs_oracle.c -- CPU restatement of the keras-rs hot path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle for libkrs_hip.so.  It is imported only by
 * tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg, and
 * only as the checker / the timed CPU baseline -- never by the product path
 * (keras_rs_amd/ fails loudly when its HIP library is missing).
 *
 * Each function restates, in plain C on HOST pointers, the arithmetic of the
 * reference statement(s) cited above it, with the same signature as its
 * device twin in include/krs.h (the `stream` argument is ignored).
 *
 * Pinning: the reference (pure Python on Keras 3) cannot be imported in the
 * build container (keras/jax/tensorflow are absent), so this oracle is pinned
 * by the reference's own known-answer tests, transcribed as data in
 * tests/golden/kat.json (see tests/golden/make_golden.py for file:line of each
 * vector) and checked by tests/test_oracle_golden.py.  Beyond those KATs parity
 * is "unpinned" in the sense of SURVEY.md section 8c.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/krs.h"

/* ---- bf16 <-> f32 (round to nearest even, NaN kept quiet) ---------------- */
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float ld(const void* p, int dtype, int64_t i) {
  return dtype == KRS_BF16 ? bf16_to_f32(((const uint16_t*)p)[i]) : ((const float*)p)[i];
}
static inline void st(void* p, int dtype, int64_t i, float v) {
  if (dtype == KRS_BF16)
    ((uint16_t*)p)[i] = f32_to_bf16(v);
  else
    ((float*)p)[i] = v;
}
static inline int64_t ldi(const void* p, int itype, int64_t i) {
  return itype == KRS_I64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}

int krs_oracle_version(void) { return KRS_VERSION; }

/* bag (f, b) -> [start, end) in ids */
static inline void bag_range(const krs_feature* feats, int f, int64_t b, int batch,
                             const void* offsets, int off_type, int64_t* s, int64_t* e) {
  if (offsets) {
    int64_t bag = (int64_t)f * batch + b;
    *s = ldi(offsets, off_type, bag);
    *e = ldi(offsets, off_type, bag + 1);
  } else {
    *s = feats[f].ids_base + b * (int64_t)feats[f].hot;
    *e = *s + feats[f].hot;
  }
}

/* divide_no_nan reciprocal form: 0 where the divisor is 0.
 * keras ops.divide_no_nan as used at embed_reduce.py:267-274. */
static inline float bag_scale_of(int combiner, float sw, float sw2) {
  if (combiner == KRS_MEAN) return sw == 0.0f ? 0.0f : 1.0f / sw;
  if (combiner == KRS_SQRTN) {
    float d = sqrtf(sw2);
    return d == 0.0f ? 0.0f : 1.0f / d;
  }
  return 1.0f;
}

/*
 * K1.  EmbedReduce.call for every feature of a DistributedEmbedding:
 *   gather      keras.layers.Embedding.call == ops.take(table, ids, axis=0)   embed_reduce.py:178
 *   weights     ones when None                                                embed_reduce.py:224-231
 *   x * w       embed_reduce.py:253
 *   sum(axis=-2) and the mean / sqrtn divisors (divide_no_nan)                embed_reduce.py:261-274
 *   per-feature loop                                     base_distributed_embedding.py:910-928
 * The numpy restatement the reference tests use is
 * keras_rs/src/layers/embedding/test_utils.py:245-267 (weights @ table[ids]).
 * fp32 accumulation in ascending position, then one division.
 */
int krs_oracle_embed_bag_fwd(const krs_table* tables, const krs_feature* feats, int n_feats,
                             const void* ids, int id_type, const void* offsets, int off_type,
                             const float* weights, int batch, int dim, int table_dtype,
                             void* out, int out_dtype, int64_t out_ld, float* bag_scale,
                             int* err_flag, void* stream) {
  (void)stream;
  if (!tables || !feats || !out || n_feats < 0 || batch < 0 || dim <= 0) return KRS_ERR_INVALID;
  int flags = 0;
#pragma omp parallel for collapse(2) schedule(static) reduction(| : flags)
  for (int f = 0; f < n_feats; ++f) {
    for (int64_t b = 0; b < batch; ++b) {
      const krs_table* tb = &tables[feats[f].table];
      int64_t s, e;
      bag_range(feats, f, b, batch, offsets, off_type, &s, &e);
      float acc[dim];
      for (int c = 0; c < dim; ++c) acc[c] = 0.0f;
      float sw = 0.0f, sw2 = 0.0f;
      for (int64_t p = s; p < e; ++p) {
        float w = weights ? weights[p] : 1.0f;
        sw += w;
        sw2 = fmaf(w, w, sw2);
        int64_t id = ldi(ids, id_type, p);
        if (id < 0 || id >= tb->vocab) {
          flags |= KRS_FLAG_ID_OUT_OF_RANGE;
          continue;
        }
        for (int c = 0; c < dim; ++c)
          acc[c] = fmaf(w, ld(tb->weights, table_dtype, id * dim + c), acc[c]);
      }
      int comb = feats[f].combiner;
      float den = comb == KRS_MEAN ? sw : (comb == KRS_SQRTN ? sqrtf(sw2) : 1.0f);
      for (int c = 0; c < dim; ++c) {
        float v = acc[c];
        if (comb != KRS_SUM) v = den == 0.0f ? 0.0f : v / den;
        st(out, out_dtype, b * out_ld + feats[f].out_col + c, v);
      }
      if (bag_scale) bag_scale[(int64_t)f * batch + b] = bag_scale_of(comb, sw, sw2);
    }
  }
  if (err_flag && flags) *err_flag |= flags;
  return KRS_OK;
}

/*
 * K2 (dense form).  Gradient of K1 w.r.t. the tables:
 *   grad.at[cols].add(vals * activation_gradients[rows])   jax/test_utils.py:395-417
 *   per-table accumulation over features sharing it          jax/test_utils.py:450-468
 * with vals = w (sum), w/sum w (mean), w/sqrt(sum w^2) (sqrtn) -- the autodiff
 * of embed_reduce.py:253-274.  Contributions are added in ascending position p
 * (feature-major), fp32.  grad_tables[t].weights = dE_t [vocab, dim] fp32,
 * accumulated INTO (caller zero-fills).
 */
int krs_oracle_embed_bag_bwd_dense(const krs_table* grad_tables, int n_tables,
                                   const krs_feature* feats, int n_feats, const void* ids,
                                   int id_type, const void* offsets, int off_type,
                                   const float* weights, const float* bag_scale,
                                   const void* grad, int grad_dtype, int64_t grad_ld, int batch,
                                   int dim) {
  (void)n_tables;
  for (int f = 0; f < n_feats; ++f) {
    const krs_table* tb = &grad_tables[feats[f].table];
    float* de = (float*)tb->weights;
    for (int64_t b = 0; b < batch; ++b) {
      int64_t s, e;
      bag_range(feats, f, b, batch, offsets, off_type, &s, &e);
      float sc = bag_scale ? bag_scale[(int64_t)f * batch + b] : 1.0f;
      for (int64_t p = s; p < e; ++p) {
        int64_t id = ldi(ids, id_type, p);
        if (id < 0 || id >= tb->vocab) continue;
        float coef = (weights ? weights[p] : 1.0f) * sc;
        for (int c = 0; c < dim; ++c)
          de[id * dim + c] =
              fmaf(coef, ld(grad, grad_dtype, b * grad_ld + feats[f].out_col + c), de[id * dim + c]);
      }
    }
  }
  return KRS_OK;
}

/*
 * Per-table optimizer step on a dense gradient.
 *   SGD      table - lr * grad                                   jax/test_utils.py:474-497
 *   Adagrad  acc += grad*grad ; table - lr / sqrt(acc) * grad    (no epsilon; same lines)
 *   Adam     keras.optimizers.Adam.update_step (named at jax/config_conversion.py:256-265):
 *            m += (g-m)(1-b1); v += (g*g-v)(1-b2); w -= lr*corr*m/(sqrt(v)+eps),
 *            corr = sqrt(1-b2^t)/(1-b1^t).   hyper = {b1, b2, eps, corr}; acc = [2][vocab][dim]
 *   FTRL     keras.optimizers.Ftrl.update_step (options of jax/config_conversion.py:266-283):
 *            n' = n+g*g; z += g-(n'^-p-n^-p)/lr*w; w = (clip(z,-l1,l1)-z)/(n'^-p/lr+2(l2+beta/(2lr)));
 *            hyper = {p, l1, l2, beta}; acc = [2][vocab][dim] (n, z)
 * Adam / FTRL arithmetic is a restatement of the published Keras formulas: the SparseCore
 * library the reference hands them to is not in /root/reference (parity unpinned).
 * Only rows with touched[r] != 0 are updated (the fused device kernels touch
 * only looked-up rows; for SGD an untouched row has grad 0 so the result is
 * identical, for Adagrad acc stays and 0/sqrt(acc) = 0 as long as acc > 0;
 * Adam / FTRL are "lazy": untouched rows keep value and slots).
 * kind: 0 = SGD, 1 = Adagrad, 2 = Adam, 3 = FTRL, 4 = row-wise Adagrad (opt-in variant, acc = [vocab]).
 */
int krs_oracle_apply_optimizer2(void* table, int table_dtype, float* acc, const float* grad,
                                const uint8_t* touched, int64_t vocab, int dim, float lr,
                                int kind, const float* hyper) {
  const int64_t plane = vocab * dim;
  if (kind == 4) {
    /* row-wise Adagrad (opt-in variant, include/krs.h krs_embed_bag_bwd_fused_adagrad_rowwise): acc is [vocab];
     * the row's sum of squares is taken in the kernel's order -- 16-byte pieces of the GRADIENT dtype per lane
     * (N = 8 columns for bf16 gradients, 4 for fp32: `hyper[0]` carries N), fmaf chain inside a piece,
     * butterfly over the pieces -- so that the accumulator matches bit for bit. */
    const int n = hyper && hyper[0] > 0 ? (int)hyper[0] : 4;
    for (int64_t r = 0; r < vocab; ++r) {
      if (touched && !touched[r]) continue;
      int lanes = 1;
      while (lanes * n < dim) lanes <<= 1;
      float part[64];
      for (int l = 0; l < lanes; ++l) {
        float ss = 0.0f;
        for (int k = 0; k < n && l * n + k < dim; ++k) ss = fmaf(grad[r * dim + l * n + k], grad[r * dim + l * n + k], ss);
        part[l] = ss;
      }
      for (int o = lanes / 2; o > 0; o >>= 1)          /* every lane ends with the same total: lane 0's */
        for (int l = 0; l < lanes; ++l) {
          if ((l & o) == 0) { float a = part[l], b = part[l | o]; part[l] = a + b; part[l | o] = b + a; }
        }
      const float a_new = acc[r] + part[0] / (float)dim;
      const float inv = a_new > 0.0f ? lr / sqrtf(a_new) : 0.0f;   /* zero accumulator + zero gradient: row unchanged */
      for (int c = 0; c < dim; ++c) {
        int64_t i = r * dim + c;
        st(table, table_dtype, i, ld(table, table_dtype, i) - inv * grad[i]);
      }
      acc[r] = a_new;
    }
    return KRS_OK;
  }
  for (int64_t r = 0; r < vocab; ++r) {
    if (touched && !touched[r]) continue;
    for (int c = 0; c < dim; ++c) {
      int64_t i = r * dim + c;
      float g = grad[i];
      float t = ld(table, table_dtype, i);
      if (kind == 1) {
        float a = fmaf(g, g, acc[i]);
        acc[i] = a;
        t = t - lr * g / sqrtf(a);
      } else if (kind == 2) {
        float m = acc[i], v = acc[plane + i];
        m = m + (g - m) * (1.0f - hyper[0]);
        v = v + (g * g - v) * (1.0f - hyper[1]);
        t = t - (lr * hyper[3]) * m / (sqrtf(v) + hyper[2]);
        acc[i] = m;
        acc[plane + i] = v;
      } else if (kind == 3) {
        float n = acc[i], z = acc[plane + i];
        float n_new = n + g * g;
        float pn = hyper[0] == -0.5f ? sqrtf(n_new) : powf(n_new, -hyper[0]);
        float po = hyper[0] == -0.5f ? sqrtf(n) : powf(n, -hyper[0]);
        z = z + g - (pn - po) / lr * t;
        float quad = pn / lr + 2.0f * (hyper[2] + hyper[3] / (2.0f * lr));
        float zc = fminf(fmaxf(z, -hyper[1]), hyper[1]);
        t = (zc - z) / quad;
        acc[i] = n_new;
        acc[plane + i] = z;
      } else {
        t = t - lr * g;
      }
      st(table, table_dtype, i, t);
    }
  }
  return KRS_OK;
}

int krs_oracle_apply_optimizer(void* table, int table_dtype, float* acc, const float* grad,
                               const uint8_t* touched, int64_t vocab, int dim, float lr,
                               int kind) {
  return krs_oracle_apply_optimizer2(table, table_dtype, acc, grad, touched, vocab, dim, lr, kind, 0);
}

static inline float act_apply(int act, float v) {
  switch (act) {
    case KRS_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case KRS_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case KRS_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

/*
 * K3.  The dense contraction + fused epilogue of FeatureCross.call:
 *   keras Dense: matmul(inputs, kernel) + bias, then activation  feature_cross.py:134-151,187
 *   u += diag_scale * x                                           feature_cross.py:191-192
 *   y = x0 * u + x                                                feature_cross.py:194
 * fp32 accumulation over ascending k; inputs of dtype in_dtype, one rounding
 * to out_dtype at the end (SURVEY.md section 8c bf16 policy).
 */
int krs_oracle_gemm(const void* a, int64_t lda, int a_is_km, const void* b, int64_t ldb,
                    int b_is_nk, void* c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                    int in_dtype, int out_dtype, const krs_gemm_epilogue* ep) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; ++i) {
    for (int64_t j = 0; j < n; ++j) {
      float acc = 0.0f;
      for (int64_t kk = 0; kk < k; ++kk) {
        float av = a_is_km ? ld(a, in_dtype, kk * lda + i) : ld(a, in_dtype, i * lda + kk);
        float bv = b_is_nk ? ld(b, in_dtype, j * ldb + kk) : ld(b, in_dtype, kk * ldb + j);
        acc = fmaf(av, bv, acc);
      }
      float v = acc;
      if (ep) {
        if (ep->bias) v += ep->bias[j];
        v = act_apply(ep->act, v);
        if (ep->x0) {
          float xv = ld(ep->x, out_dtype, i * ep->ldx + j);
          if (ep->u_out) st(ep->u_out, out_dtype, i * ep->ldu + j, v);
          float u = v + ep->diag_scale * xv;
          v = ld(ep->x0, out_dtype, i * ep->ldx + j) * u + xv;
        }
        if (ep->r) v += ep->beta * ld(ep->r, out_dtype, i * ep->ldr + j);
      }
      st(c, out_dtype, i * ldc + j, v);
    }
  }
  return KRS_OK;
}

/* y = x0 * (u + diag*x) + x   feature_cross.py:191-194 */
int krs_oracle_cross_epilogue_fwd(const void* u, const void* x0, const void* x, void* y,
                                  int64_t m, int64_t n, int64_t ldm, float diag_scale,
                                  int dtype) {
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j) {
      int64_t o = i * ldm + j;
      float xv = ld(x, dtype, o);
      st(y, dtype, o, ld(x0, dtype, o) * (ld(u, dtype, o) + diag_scale * xv) + xv);
    }
  return KRS_OK;
}

static inline float act_grad_from_output(int act, float u) {
  switch (act) {
    case KRS_ACT_RELU: return u > 0.0f ? 1.0f : 0.0f;
    case KRS_ACT_SIGMOID: return u * (1.0f - u);
    case KRS_ACT_TANH: return 1.0f - u * u;
    default: return 1.0f;
  }
}

/* Autodiff of y = x0 * (act(z) + diag*x) + x w.r.t. z, x0 and the direct x path;
 * u = act(z) as saved by the forward. */
int krs_oracle_cross_epilogue_bwd(const void* g, const void* u, const void* x0, const void* x,
                                  void* du, void* dx0, int dx0_accumulate, void* dxd,
                                  float* dbias, int64_t m, int64_t n, int64_t ldm,
                                  float diag_scale, int act, int dtype) {
  if (dbias)
    for (int64_t j = 0; j < n; ++j) dbias[j] = 0.0f;
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j) {
      int64_t o = i * ldm + j;
      float gv = ld(g, dtype, o);
      float gx0 = gv * ld(x0, dtype, o);
      float uv = u ? ld(u, dtype, o) : 0.0f;
      float dz = gx0 * act_grad_from_output(act, uv);
      if (du) st(du, dtype, o, dz);
      int fold = dxd && dxd == dx0; /* x is x0: both terms of dL/dx0 in one buffer */
      if (dx0) {
        float t = (dx0_accumulate ? ld(dx0, dtype, o) : 0.0f) + gv * (uv + diag_scale * ld(x, dtype, o));
        if (fold) t += gv + diag_scale * gx0;
        st(dx0, dtype, o, t);
      }
      if (dxd && !fold) st(dxd, dtype, o, gv + diag_scale * gx0);
      if (dbias) dbias[j] += dz;
    }
  return KRS_OK;
}

int krs_oracle_colsum(const void* a, int64_t lda, int64_t m, int64_t n, int dtype, float* out) {
  for (int64_t j = 0; j < n; ++j) out[j] = 0.0f;
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j) out[j] += ld(a, dtype, i * lda + j);
  return KRS_OK;
}

/* Column index of pair (i, j) in the output row, or -1 when (i, j) is not kept.
 * Row-major lower-triangle order of dot_interaction.py:118-132; the masked
 * flatten of :96-116 keeps position i*F+j. */
static inline int64_t di_col(int i, int j, int F, int self_interaction, int skip_gather) {
  int keep = self_interaction ? (j <= i) : (j < i);
  if (skip_gather) return keep ? (int64_t)i * F + j : -2; /* -2: written as zero */
  if (!keep) return -1;
  return self_interaction ? (int64_t)i * (i + 1) / 2 + j : (int64_t)i * (i - 1) / 2 + j;
}

/*
 * K4.  DotInteraction.call, dot_interaction.py:170-203:
 *   features = stack(inputs, axis=1); P = matmul(features, features^T)
 *   skip_gather: P * tril(ones, k) reshaped to [B, F*F];  else take(P.flat, tril indices)
 */
int krs_oracle_dot_interaction_fwd(const void* const* feats, const int64_t* ldf, int n_feats,
                                   int64_t batch, int dim, int dtype, int self_interaction,
                                   int skip_gather, void* out, int64_t out_ld) {
  int F = n_feats;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; ++b)
    for (int i = 0; i < F; ++i)
      for (int j = 0; j < F; ++j) {
        int64_t col = di_col(i, j, F, self_interaction, skip_gather);
        if (col == -1) continue;
        if (col == -2) {
          st(out, dtype, b * out_ld + (int64_t)i * F + j, 0.0f);
          continue;
        }
        float acc = 0.0f;
        for (int c = 0; c < dim; ++c)
          acc = fmaf(ld(feats[i], dtype, b * ldf[i] + c), ld(feats[j], dtype, b * ldf[j] + c), acc);
        st(out, dtype, b * out_ld + col, acc);
      }
  return KRS_OK;
}

/* Autodiff of K4: dX[b,i,:] = sum_j (G[b,i,j] + G[b,j,i]) X[b,j,:]. */
int krs_oracle_dot_interaction_bwd_accumulate(const void* const* feats, const int64_t* ldf, int n_feats,
                                              int64_t batch, int dim, int dtype, int self_interaction,
                                              int skip_gather, const void* grad_out, int64_t grad_ld,
                                              void* const* grad_feats, const int64_t* gld,
                                              uint64_t accumulate_mask);

int krs_oracle_dot_interaction_bwd(const void* const* feats, const int64_t* ldf, int n_feats,
                                   int64_t batch, int dim, int dtype, int self_interaction,
                                   int skip_gather, const void* grad_out, int64_t grad_ld,
                                   void* const* grad_feats, const int64_t* gld) {
  return krs_oracle_dot_interaction_bwd_accumulate(feats, ldf, n_feats, batch, dim, dtype, self_interaction,
                                                   skip_gather, grad_out, grad_ld, grad_feats, gld, 0);
}

/* include/krs.h krs_dot_interaction_bwd_accumulate: features whose mask bit is set receive
 * stored + dX (the stored value widened to fp32, one rounding of the sum); the gradient meeting the one of the
 * concat of the same features, which autodiff would add (examples/ml_perf/model.py:204-207). */
int krs_oracle_dot_interaction_bwd_accumulate(const void* const* feats, const int64_t* ldf, int n_feats,
                                              int64_t batch, int dim, int dtype, int self_interaction,
                                              int skip_gather, const void* grad_out, int64_t grad_ld,
                                              void* const* grad_feats, const int64_t* gld,
                                              uint64_t accumulate_mask) {
  int F = n_feats;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; ++b)
    for (int i = 0; i < F; ++i)
      for (int c = 0; c < dim; ++c) {
        float acc = 0.0f;
        for (int j = 0; j < F; ++j) {
          int64_t cij = di_col(i, j, F, self_interaction, skip_gather);
          int64_t cji = di_col(j, i, F, self_interaction, skip_gather);
          float gs = 0.0f;
          if (cij >= 0) gs += ld(grad_out, dtype, b * grad_ld + cij);
          if (cji >= 0) gs += ld(grad_out, dtype, b * grad_ld + cji);
          acc = fmaf(gs, ld(feats[j], dtype, b * ldf[j] + c), acc);
        }
        if (i < 64 && ((accumulate_mask >> i) & 1)) acc += ld(grad_feats[i], dtype, b * gld[i] + c);
        st(grad_feats[i], dtype, b * gld[i] + c, acc);
      }
  return KRS_OK;
}

/*
 * K5.  MOD sharding (sharding_strategy="MOD", jax/embedding_utils.py:194;
 * row r -> shard r % S, local row r / S, tensorflow/distributed_embedding.py:316-328).
 * Stable counting sort by destination shard.
 */
int krs_oracle_mod_bucketize(const void* ids, int id_type, int64_t nnz, int n_shards,
                             void* local_ids, int32_t* perm, int64_t* bucket_counts) {
  if (n_shards <= 0) return KRS_ERR_INVALID;
  int64_t* start = (int64_t*)calloc((size_t)n_shards + 1, sizeof(int64_t));
  for (int s = 0; s < n_shards; ++s) bucket_counts[s] = 0;
  for (int64_t p = 0; p < nnz; ++p) {
    int64_t id = ldi(ids, id_type, p);
    int64_t s = ((id % n_shards) + n_shards) % n_shards;
    bucket_counts[s]++;
  }
  for (int s = 0; s < n_shards; ++s) start[s + 1] = start[s] + bucket_counts[s];
  for (int64_t p = 0; p < nnz; ++p) {
    int64_t id = ldi(ids, id_type, p);
    int64_t s = ((id % n_shards) + n_shards) % n_shards;
    int64_t q = start[s]++;
    int64_t loc = (id - s) / n_shards;
    if (id_type == KRS_I64)
      ((int64_t*)local_ids)[q] = loc;
    else
      ((int32_t*)local_ids)[q] = (int32_t)loc;
    perm[q] = (int32_t)p;
  }
  free(start);
  return KRS_OK;
}

/* ---- K6: row-sharded lookup, id side (include/krs.h: krs_shard_route / unpack / combine) --------------
 * Sequential restatement: MOD ownership and local rows as in the reference's layout
 * (tensorflow/distributed_embedding.py:316-328, jax/embedding_utils.py:187-197), combiner scales as
 * embed_reduce.py:255-274, segments = runs of one bag inside an owner's bucket in lookup order. */
int krs_oracle_shard_route(const krs_shard_feature* feats, int n_feats, const void* ids, int id_type,
                           const void* offsets, int offset_type, const float* weights, int64_t nnz, int batch,
                           int n_shards, int emit_weights, int32_t* packed, int32_t* seg_bag, int32_t* seg_grow,
                           int32_t* bag_seg, int64_t* counts, int32_t* err_flag) {
  const int64_t n_bags = (int64_t)batch * n_feats;
  for (int64_t i = 0; i < n_bags * n_shards; ++i) bag_seg[i] = -1;
  for (int i = 0; i < 3 * n_shards; ++i) counts[i] = 0;
  if (nnz == 0) return 0;
  int32_t* dest = (int32_t*)malloc((size_t)nnz * 4);
  int32_t* row = (int32_t*)malloc((size_t)nnz * 4);
  int32_t* bagp = (int32_t*)malloc((size_t)nnz * 4);
  float* wl = (float*)malloc((size_t)nnz * 4);
  /* per lookup: bag, validity, owner, local row, effective weight */
  for (int f = 0; f < n_feats; ++f)
    for (int b = 0; b < batch; ++b) {
      const int64_t bag = (int64_t)f * batch + b;
      int64_t lo, hi;
      if (offsets) { lo = ldi(offsets, offset_type, bag); hi = ldi(offsets, offset_type, bag + 1); }
      else { lo = feats[f].ids_base + (int64_t)b * feats[f].hot; hi = lo + feats[f].hot; }
      float s1 = 0.0f, s2 = 0.0f;
      for (int64_t q = lo; q < hi; ++q) {
        const float w = weights ? weights[q] : 1.0f;
        s1 += w;
        s2 = fmaf(w, w, s2);
      }
      float scale = 1.0f;
      if (feats[f].combiner != KRS_SUM) {
        const float d = feats[f].combiner == KRS_MEAN ? s1 : sqrtf(s2);
        scale = d != 0.0f ? 1.0f / d : 0.0f;
      }
      for (int64_t q = lo; q < hi; ++q) {
        const int64_t id = ldi(ids, id_type, q);
        bagp[q] = (int32_t)bag;
        wl[q] = (weights ? weights[q] : 1.0f) * scale;
        if (id < 0 || id >= feats[f].vocab) {
          dest[q] = n_shards;
          row[q] = 0;
          if (err_flag) *err_flag |= KRS_FLAG_ID_OUT_OF_RANGE;
        } else {
          const int64_t c = feats[f].comp_off + id;
          dest[q] = (int32_t)(c % n_shards);
          row[q] = (int32_t)(c / n_shards);
        }
      }
    }
  /* counts, then the packed blocks in owner order (stable inside an owner) */
  for (int64_t q = 0; q < nnz; ++q)
    if (dest[q] < n_shards) counts[dest[q]]++;
  int64_t n_seg = 0;
  for (int d = 0; d < n_shards; ++d) {
    int32_t prev = -1;
    int first = 1;
    for (int64_t q = 0; q < nnz; ++q)
      if (dest[q] == d) {
        if (first || bagp[q] != prev) counts[n_shards + d]++;
        prev = bagp[q];
        first = 0;
      }
  }
  int64_t base = 0;
  for (int d = 0; d < n_shards; ++d) {
    const int64_t cnt = counts[d], segs = counts[n_shards + d];
    counts[2 * n_shards + d] = cnt * (1 + (emit_weights != 0)) + segs;
    int64_t k = 0, sk = -1;
    int32_t prev = -1;
    for (int64_t q = 0; q < nnz; ++q)
      if (dest[q] == d) {
        if (k == 0 || bagp[q] != prev) {
          ++sk;
          const int32_t bag = bagp[q];
          seg_bag[n_seg] = bag;
          seg_grow[n_seg] = (bag % batch) * n_feats + bag / batch;
          bag_seg[(int64_t)bag * n_shards + d] = (int32_t)n_seg;
          packed[base + cnt * (1 + (emit_weights != 0)) + sk] = 0;
          ++n_seg;
        }
        prev = bagp[q];
        packed[base + k] = row[q];
        if (emit_weights) memcpy(&packed[base + cnt + k], &wl[q], 4);
        packed[base + cnt * (1 + (emit_weights != 0)) + sk] += 1;
        ++k;
      }
    base += counts[2 * n_shards + d];
  }
  free(dest); free(row); free(bagp); free(wl);
  return 0;
}

int krs_oracle_shard_unpack(const int32_t* packed, int n_sources, const int64_t* lookups, const int64_t* segments,
                            int weighted, int32_t* rows, float* w, int32_t* offsets) {
  int64_t base = 0, r = 0, s = 0;
  int32_t run = 0;
  for (int src = 0; src < n_sources; ++src) {
    const int64_t cnt = lookups[src], segs = segments[src];
    for (int64_t k = 0; k < cnt; ++k) {
      rows[r + k] = packed[base + k];
      if (weighted) memcpy(&w[r + k], &packed[base + cnt + k], 4);
    }
    for (int64_t k = 0; k < segs; ++k) {
      offsets[s + k] = run;
      run += packed[base + cnt * (1 + (weighted != 0)) + k];
    }
    r += cnt;
    s += segs;
    base += cnt * (1 + (weighted != 0)) + segs;
  }
  offsets[s] = run;
  return 0;
}

int krs_oracle_shard_combine(const void* partials, const int32_t* bag_seg, int batch, int n_feats, int n_shards,
                             int dim, int dtype, void* out, int64_t out_ld) {
  for (int f = 0; f < n_feats; ++f)
    for (int b = 0; b < batch; ++b)
      for (int j = 0; j < dim; ++j) {
        float acc = 0.0f;
        for (int d = 0; d < n_shards; ++d) {
          const int32_t s = bag_seg[((int64_t)f * batch + b) * n_shards + d];
          if (s >= 0) acc += ld(partials, dtype, (int64_t)s * dim + j);
        }
        st(out, dtype, (int64_t)b * out_ld + (int64_t)f * dim + j, acc);
      }
  return 0;
}
