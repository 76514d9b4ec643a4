/*
 * sae_oracle.c -- CPU restatement of the reference SAE hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (multimodal-sae_amd/) never links, imports or calls anything in oracle/.
 *
 * What is restated (reference = /root/reference, EvolvingLMMs-Lab/multimodal-sae):
 *   msae_oracle_pre_acts   Sae.pre_acts            sae_auto_interp/sae/sae.py:172-177
 *   msae_oracle_topk       Sae.select_topk         sae_auto_interp/sae/sae.py:179-181
 *                          torch.topk in the cache sae_auto_interp/features/cache.py:210-212
 *   msae_oracle_decode     Sae.decode              sae_auto_interp/sae/sae.py:187-191
 *                          k-sparse gather matmul  sae_auto_interp/sae/kernels.py:222-284
 *   msae_oracle_sparsify   scatter_ + Cache.add / get_nonzeros
 *                                                  sae_auto_interp/features/cache.py:214-217,42-92
 *   msae_oracle_decode_bwd TritonDecoder.backward  sae_auto_interp/sae/kernels.py:411-429
 *
 * Arithmetic contract (the reference leaves these to torch/BLAS, so they are *defined* here and
 * pinned against reference outputs by tests/golden/ within the tolerances stated in DESIGN.md):
 *   - dot products are ONE f32 fused-multiply-add chain in ascending k, starting from +0.0f:
 *       acc = fmaf(a[k], w[k], acc),  a[k] = x[k] - b_dec[k]  (f32 subtract)
 *     then `acc + b_enc[n]` (one f32 add) and ReLU.  This is exactly what gfx950's
 *     v_mfma_f32_32x32x2_f32 computes when k is walked in order, so the HIP path is compared
 *     BIT-EXACTLY against this file.
 *   - top-k order is canonical: value descending, then feature index ascending (ties / zeros).
 *   - decode is one fmaf chain over the k selected rows in their given order, entries with
 *     act == 0 skipped (kernels.py:277), then `+ b_dec`.
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TB 16 /* tokens per SIMD block */

int msae_oracle_abi_version(void) { return 1; }

int msae_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* aT[k][t] = x[t][k] - b_dec[k], token dimension padded to a multiple of TB with zeros. */
static float *make_sae_in_T(const float *x, const float *b_dec, int T, int d, int Tp) {
  float *aT = (float *)aligned_alloc(64, (size_t)d * Tp * sizeof(float));
  if (!aT) return NULL;
  memset(aT, 0, (size_t)d * Tp * sizeof(float));
#pragma omp parallel for schedule(static)
  for (int k = 0; k < d; ++k) {
    const float bd = b_dec ? b_dec[k] : 0.0f;
    for (int t = 0; t < T; ++t) aT[(size_t)k * Tp + t] = x[(size_t)t * d + k] - bd;
  }
  return aT;
}

/* One feature row against a block of TB tokens: TB independent k-ordered fmaf chains. */
static inline void dot_block(const float *aT, int Tp, int t0, const float *w, int d, float *acc) {
  for (int i = 0; i < TB; ++i) acc[i] = 0.0f;
  for (int k = 0; k < d; ++k) {
    const float wk = w[k];
    const float *a = aT + (size_t)k * Tp + t0;
    for (int i = 0; i < TB; ++i) acc[i] = __builtin_fmaf(a[i], wk, acc[i]);
  }
}

/* out[t][n] = relu( chain_k (x[t][k]-b_dec[k]) * W[n][k]  + b_enc[n] )      sae.py:172-177 */
int msae_oracle_pre_acts(const float *x, const float *W_enc, const float *b_enc,
                         const float *b_dec, int T, int d, int N, int relu, float *out) {
  const int Tp = (T + TB - 1) / TB * TB;
  float *aT = make_sae_in_T(x, b_dec, T, d, Tp);
  if (!aT) return -1;
#pragma omp parallel for schedule(dynamic, 16)
  for (int n = 0; n < N; ++n) {
    const float *w = W_enc + (size_t)n * d;
    const float bn = b_enc ? b_enc[n] : 0.0f;
    float acc[TB];
    for (int t0 = 0; t0 < Tp; t0 += TB) {
      dot_block(aT, Tp, t0, w, d, acc);
      for (int i = 0; i < TB && t0 + i < T; ++i) {
        float v = acc[i] + bn;
        if (relu && !(v > 0.0f)) v = 0.0f;
        out[(size_t)(t0 + i) * N + n] = v;
      }
    }
  }
  free(aT);
  return 0;
}

/* ---- canonical top-k: (value desc, index asc) ------------------------------------------- */
typedef struct {
  float v;
  int32_t i;
} cand_t;

/* "a ranks strictly before b" */
static inline int before(cand_t a, cand_t b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }

/* min-heap on rank: root = the WORST of the kept k */
static void sift_down(cand_t *h, int n, int p) {
  for (;;) {
    int l = 2 * p + 1, r = l + 1, m = p;
    if (l < n && before(h[m], h[l])) m = l;
    if (r < n && before(h[m], h[r])) m = r;
    if (m == p) return;
    cand_t t = h[p];
    h[p] = h[m];
    h[m] = t;
    p = m;
  }
}

static int cmp_rank(const void *pa, const void *pb) {
  cand_t a = *(const cand_t *)pa, b = *(const cand_t *)pb;
  return before(a, b) ? -1 : (before(b, a) ? 1 : 0);
}

static void topk_row(const float *row, int N, int k, float *vals, int32_t *idx) {
  cand_t *h = (cand_t *)malloc((size_t)k * sizeof(cand_t));
  int n = 0;
  for (int i = 0; i < N; ++i) {
    cand_t c = {row[i], i};
    if (n < k) {
      h[n++] = c;
      if (n == k)
        for (int p = k / 2 - 1; p >= 0; --p) sift_down(h, k, p);
    } else if (before(c, h[0])) {
      h[0] = c;
      sift_down(h, k, 0);
    }
  }
  qsort(h, (size_t)n, sizeof(cand_t), cmp_rank);
  for (int j = 0; j < n; ++j) {
    vals[j] = h[j].v;
    idx[j] = h[j].i;
  }
  free(h);
}

/* latents [T][N] -> vals [T][k] (descending), idx [T][k]                     sae.py:179-181 */
int msae_oracle_topk(const float *latents, int T, int N, int k, float *vals, int32_t *idx) {
  if (k > N) return -2;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < T; ++t)
    topk_row(latents + (size_t)t * N, N, k, vals + (size_t)t * k, idx + (size_t)t * k);
  return 0;
}

/* Optional per-token edit of the dense latents before TopK, as the reference's hooks do:
 *   set_feature >= 0 : latents[:, set_feature] = set_value      features/steering.py:113-114
 *   zero_feature >= 0: latents[:, zero_feature] *= 0             features/patching/utils.py:43-48
 * Fused encode: pre_acts + edit + topk without keeping [T][N].                sae.py:183-185 */
int msae_oracle_encode_topk(const float *x, const float *W_enc, const float *b_enc,
                            const float *b_dec, int T, int d, int N, int k, int set_feature,
                            float set_value, int zero_feature, float *vals, int32_t *idx) {
  if (k > N) return -2;
  const int Tp = (T + TB - 1) / TB * TB;
  float *aT = make_sae_in_T(x, b_dec, T, d, Tp);
  if (!aT) return -1;
  int rc = 0;
  for (int t0 = 0; t0 < Tp && rc == 0; t0 += TB) {
    float *lat = (float *)malloc((size_t)TB * N * sizeof(float));
    if (!lat) {
      rc = -1;
      break;
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int n = 0; n < N; ++n) {
      float acc[TB];
      dot_block(aT, Tp, t0, W_enc + (size_t)n * d, d, acc);
      const float bn = b_enc ? b_enc[n] : 0.0f;
      for (int i = 0; i < TB; ++i) {
        float v = acc[i] + bn;
        if (!(v > 0.0f)) v = 0.0f;
        if (n == set_feature) v = set_value;
        if (n == zero_feature) v = v * 0.0f;
        lat[(size_t)i * N + n] = v;
      }
    }
    const int nt = (T - t0) < TB ? (T - t0) : TB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < nt; ++i)
      topk_row(lat + (size_t)i * N, N, k, vals + (size_t)(t0 + i) * k, idx + (size_t)(t0 + i) * k);
    free(lat);
  }
  free(aT);
  return rc;
}

/* out[a][:] = chain_j acts[a][j] * W_dec[idx[a][j]][:]  + b_dec     sae.py:187-191, kernels.py:222-284 */
int msae_oracle_decode(const int32_t *idx, const float *acts, const float *W_dec,
                       const float *b_dec, int A, int k, int N, int d, float *out) {
  int bad = 0;
#pragma omp parallel for schedule(static)
  for (int a = 0; a < A; ++a) {
    float *o = out + (size_t)a * d;
    for (int c = 0; c < d; ++c) o[c] = 0.0f;
    for (int j = 0; j < k; ++j) {
      const float v = acts[(size_t)a * k + j];
      const int32_t i = idx[(size_t)a * k + j];
      if (i < 0 || i >= N) { /* tl.device_assert(i < N), kernels.py:276 */
#pragma omp atomic write
        bad = 1;
        continue;
      }
      if (v == 0.0f) continue; /* kernels.py:277 */
      const float *w = W_dec + (size_t)i * d;
      for (int c = 0; c < d; ++c) o[c] = __builtin_fmaf(v, w[c], o[c]);
    }
    if (b_dec)
      for (int c = 0; c < d; ++c) o[c] = o[c] + b_dec[c];
  }
  return bad ? -3 : 0;
}

/* grad wrt the top-k activations: g_acts[a][j] = chain_c grad_out[a][c] * W_dec[idx[a][j]][c]
 *                                                             kernels.py:421-425, 341-400 */
int msae_oracle_decode_bwd_acts(const int32_t *idx, const float *grad_out, const float *W_dec,
                                int A, int k, int N, int d, float *g_acts) {
#pragma omp parallel for schedule(static)
  for (int a = 0; a < A; ++a)
    for (int j = 0; j < k; ++j) {
      const int32_t i = idx[(size_t)a * k + j];
      const float *w = W_dec + (size_t)i * d;
      const float *g = grad_out + (size_t)a * d;
      float acc = 0.0f;
      for (int c = 0; c < d; ++c) acc = __builtin_fmaf(g[c], w[c], acc);
      g_acts[(size_t)a * k + j] = acc;
    }
  (void)N;
  return 0;
}

/* ---- cache sparsify ------------------------------------------------------------------------
 * Reference: result = zeros_like(latents).scatter_(-1, topk.indices, topk.values)
 *            locations = nonzero(|result| > 1e-5) ; activations = result[|result| > 1e-5]
 *            optional mask = isin(locations[:,2], filters[module])
 *            locations[:,0] += batch_number*batch_size + shard_size
 * (features/cache.py:214-217, 80-92, 55).  Row-major nonzero order == (row, pos, feature asc).
 * Input here: per token the k (val, idx) pairs in any order, tokens laid out [B][S].
 * `filter_bitmap` (N bytes, 1 = keep) or NULL.  Returns nnz; writes at most cap entries. */
static int cmp_i32(const void *a, const void *b) {
  int32_t x = ((const cand_t *)a)->i, y = ((const cand_t *)b)->i;
  return x < y ? -1 : (x > y);
}

int64_t msae_oracle_sparsify(const float *vals, const int32_t *idx, int B, int S, int k,
                             int64_t row_base, float thresh, const uint8_t *filter_bitmap,
                             int64_t cap, int64_t *locations, float *activations) {
  int64_t nnz = 0;
  cand_t *tmp = (cand_t *)malloc((size_t)k * sizeof(cand_t));
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      const size_t t = (size_t)b * S + s;
      for (int j = 0; j < k; ++j) {
        tmp[j].v = vals[t * k + j];
        tmp[j].i = idx[t * k + j];
      }
      qsort(tmp, (size_t)k, sizeof(cand_t), cmp_i32);
      for (int j = 0; j < k; ++j) {
        if (!(fabsf(tmp[j].v) > thresh)) continue;
        if (filter_bitmap && !filter_bitmap[tmp[j].i]) continue;
        if (nnz < cap) {
          locations[nnz * 3 + 0] = row_base + b;
          locations[nnz * 3 + 1] = s;
          locations[nnz * 3 + 2] = tmp[j].i;
          activations[nnz] = tmp[j].v;
        }
        ++nnz;
      }
    }
  free(tmp);
  return nnz;
}
